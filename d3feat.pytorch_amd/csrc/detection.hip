// Detector (saliency) score, forward + backward.
//
// Replaces reference KPFCNN.detection_scores (models/architectures.py:322-368):
//   f      = feat / (max(feat) + 1e-6)                                     (:342, global max over the stacked pair)
//   mean_n = sum_h f[idx[n,h]] / max(1, #{h : sum_c f[idx[n,h],c] != 0})   (:345-349, shadow rows are zero)
//   alpha  = softplus(f_n - mean_n)      beta = f_n / (1e-6 + max_c f_n)   (:350-354)
//   score  = max_c(alpha * beta)                                           (:356-358)
//   eval:  score *= any_c( f_n[c] == max_h f[idx[n,h],c] )                 (:361-366)
// The reference gathers [N,H,C] (207 MB at N = 38.6k); here one wave serves one point, lanes <-> channels, 64/CP
// neighbors per step, and nothing is materialised.
#include "common.hpp"

namespace {

template <int W>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = W >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int W>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
  for (int o = W >> 1; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ float softplus_t(float x) { return x > 20.0f ? x : log1pf(expf(x)); }  // torch threshold=20

struct PointStats {
  float fself, mean, lmax, num, dmax, alpha, beta, sc;
};

// shared by forward and backward: everything a lane (channel c, neighbor group g) needs about point n
template <int CP>
__device__ __forceinline__ PointStats point_stats(const float* __restrict__ feat, int N, int C,
                                                  const int32_t* __restrict__ row, int H, float denom, int c, int g) {
  constexpr int G = 64 / CP;
  PointStats s;
  float msum = 0.0f, lmax = -INFINITY;
  int cnt = 0;
  for (int h0 = 0; h0 < H; h0 += G) {
    const int h = h0 + g;
    const int m = h < H ? row[h] : N;
    const bool real = m >= 0 && m < N;
    const float v = (real && c < C) ? feat[(size_t)m * C + c] / denom : 0.0f;
    const float rs = group_sum<CP>(v);
    if (h < H) {
      cnt += rs != 0.0f;
      msum += v;
      lmax = fmaxf(lmax, v);
    }
  }
#pragma unroll
  for (int o = CP; o < 64; o <<= 1) {
    msum += __shfl_xor(msum, o, 64);
    cnt += __shfl_xor(cnt, o, 64);
    lmax = fmaxf(lmax, __shfl_xor(lmax, o, 64));
  }
  s.num = (float)(cnt > 1 ? cnt : 1);
  s.mean = msum / s.num;
  s.lmax = lmax;
  return s;
}

template <int CP>
__global__ __launch_bounds__(256) void det_fwd_kernel(const float* __restrict__ feat, int N, int C,
                                                      const int32_t* __restrict__ idx, int H,
                                                      const float* __restrict__ fmax, int training,
                                                      float* __restrict__ scores) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const int c = lane % CP, g = lane / CP;
  const float denom = fmaxf(*fmax, 0.0f) + 1e-6f;  // the reference's max includes its zero shadow row (:336-342)
  PointStats s = point_stats<CP>(feat, N, C, idx + (size_t)n * H, H, denom, c, g);
  const float fself = c < C ? feat[(size_t)n * C + c] / denom : -INFINITY;
  const float dmax = group_max<CP>(fself);
  const float alpha = softplus_t(fself - s.mean);
  const float beta = fself / (1e-6f + dmax);
  float sc = c < C ? alpha * beta : -INFINITY;
  float score = group_max<CP>(sc);
  if (!training) {
    const float is = (c < C && fself == s.lmax) ? 1.0f : 0.0f;
    score *= group_max<CP>(is);
  }
  if (lane == 0) scores[n] = score;
}

// df (gradient wrt the NORMALISED features f) is accumulated with atomics: a point receives from itself and from
// every point that lists it as a neighbor.
template <int CP>
__global__ __launch_bounds__(256) void det_bwd_kernel(const float* __restrict__ feat, int N, int C,
                                                      const int32_t* __restrict__ idx, int H,
                                                      const float* __restrict__ fmax,
                                                      const float* __restrict__ gscore, float* __restrict__ df) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const int c = lane % CP, g = lane / CP;
  const float denom = fmaxf(*fmax, 0.0f) + 1e-6f;
  const int32_t* row = idx + (size_t)n * H;
  PointStats s = point_stats<CP>(feat, N, C, row, H, denom, c, g);
  const float fself = c < C ? feat[(size_t)n * C + c] / denom : -INFINITY;
  const float dmax = group_max<CP>(fself);
  const float u = fself - s.mean;
  const float alpha = softplus_t(u);
  const float beta = fself / (1e-6f + dmax);
  const float sc = c < C ? alpha * beta : -INFINITY;
  const float best = group_max<CP>(sc);
  // first channel attaining the max (torch.max returns the first maximal index)
  const uint64_t bm = __ballot(sc == best && lane < CP);
  const int cstar = __ffsll((unsigned long long)bm) - 1;
  const uint64_t dm = __ballot(fself == dmax && lane < CP);
  const int cprime = __ffsll((unsigned long long)dm) - 1;
  const float ds = gscore[n];
  const float f_star = __shfl(fself, cstar, 64), a_star = __shfl(alpha, cstar, 64), b_star = __shfl(beta, cstar, 64),
              u_star = __shfl(u, cstar, 64);
  const float sig = u_star > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-u_star));
  const float du = ds * b_star * sig;
  const float inv = 1.0f / (1e-6f + dmax);
  if (lane == 0) {
    atomicAdd(&df[(size_t)n * C + cstar], du + ds * a_star * inv);
    atomicAdd(&df[(size_t)n * C + cprime], -ds * a_star * f_star * inv * inv);
  }
  const float gn = -du / s.num;
  for (int h = lane; h < H; h += 64) {
    const int m = row[h];
    if (m >= 0 && m < N) atomicAdd(&df[(size_t)m * C + cstar], gn);
  }
}

__device__ __forceinline__ uint32_t f2ord(float f) {
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

// one contended atomic per WORKGROUP (a per-wave atomic on a single word serialises: ~8 ns each, 4096 of them)
__device__ __forceinline__ float block_max256(float m) {
  __shared__ float sh[4];
  m = d3f::wave_max(m);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

__global__ __launch_bounds__(256) void gmax_partial_kernel(const float* __restrict__ x, size_t n,
                                                           uint32_t* __restrict__ enc) {
  float m = -INFINITY;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    m = fmaxf(m, x[i]);
  m = block_max256(m);
  if (threadIdx.x == 0) atomicMax(enc, f2ord(m));
}
__global__ __launch_bounds__(256) void gmax_rows_kernel(const float* __restrict__ x, int cap_rows, int C,
                                                        const int32_t* __restrict__ len, int B,
                                                        uint32_t* __restrict__ enc) {
  int rows = d3f::batch_offset(len, B);
  if (rows > cap_rows) rows = cap_rows;
  const size_t n = (size_t)rows * C;
  float m = -INFINITY;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    m = fmaxf(m, x[i]);
  m = block_max256(m);
  if (threadIdx.x == 0) atomicMax(enc, f2ord(m));
}
__global__ void gmax_final_kernel(const uint32_t* __restrict__ enc, float* __restrict__ out) { *out = ord2f(*enc); }

// S = sum(df * f), ties = #{feat == fmax}
__global__ __launch_bounds__(256) void det_reduce_kernel(const float* __restrict__ feat, const float* __restrict__ df,
                                                         size_t n, const float* __restrict__ fmax,
                                                         float* __restrict__ acc /*[2]*/) {
  const float mx = fmaxf(*fmax, 0.0f), denom = mx + 1e-6f;
  float s = 0.0f, t = (blockIdx.x == 0 && threadIdx.x == 0 && mx == 0.0f) ? 1.0f : 0.0f;  // the zero shadow row ties at 0
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = feat[i];
    s += df[i] * (v / denom);
    t += v == mx ? 1.0f : 0.0f;
  }
  __shared__ float sh[2][4];
  s = d3f::wave_sum(s);
  t = d3f::wave_sum(t);
  if ((threadIdx.x & 63) == 0) {
    sh[0][threadIdx.x >> 6] = s;
    sh[1][threadIdx.x >> 6] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&acc[0], (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]));
    atomicAdd(&acc[1], (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]));
  }
}

// grad_feat = df / g  +  [feat == max] * ( -S / g ) / ties        (g = max + 1e-6; d g / d feat flows to the arg-max)
__global__ void det_finalize_kernel(const float* __restrict__ feat, size_t n, const float* __restrict__ fmax,
                                    const float* __restrict__ acc, float* __restrict__ gfeat) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float mx = fmaxf(*fmax, 0.0f), denom = mx + 1e-6f;
  float gval = gfeat[i] / denom;
  if (feat[i] == mx) gval += (-acc[0] / denom) / fmaxf(acc[1], 1.0f);
  gfeat[i] = gval;
}

}  // namespace

extern "C" {

int d3f_global_max(const float* x, size_t n, float* out_max, void* ws, size_t ws_bytes, void* stream_) {
  if (!x || !out_max || !ws || ws_bytes < 4 || n == 0) return D3F_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  if (d3f::zero_async(ws, 4, stream) != hipSuccess) return D3F_ELAUNCH;
  int blocks = d3f::cdiv((long long)n, 256 * 8);
  if (blocks > 512) blocks = 512;
  gmax_partial_kernel<<<blocks, 256, 0, stream>>>(x, n, (uint32_t*)ws);
  gmax_final_kernel<<<1, 1, 0, stream>>>((const uint32_t*)ws, out_max);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

int d3f_global_max_rows(const float* x, int cap_rows, int C, const int32_t* len, int B, float* out_max, void* ws,
                        size_t ws_bytes, void* stream_) {
  if (!x || !out_max || !ws || !len || ws_bytes < 4 || cap_rows < 1 || C < 1 || B < 1) return D3F_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  if (d3f::zero_async(ws, 4, stream) != hipSuccess) return D3F_ELAUNCH;
  int blocks = d3f::cdiv((long long)cap_rows * C, 256 * 8);
  if (blocks > 512) blocks = 512;
  gmax_rows_kernel<<<blocks, 256, 0, stream>>>(x, cap_rows, C, len, B, (uint32_t*)ws);
  gmax_final_kernel<<<1, 1, 0, stream>>>((const uint32_t*)ws, out_max);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

int d3f_detection_scores_forward(const float* feat, int N, int C, const int32_t* idx, int H, const float* feat_max,
                                 int training, float* scores, void* stream_) {
  if (!feat || !idx || !feat_max || !scores || N < 0 || C < 1 || C > 64 || H < 1) return D3F_EINVAL;
  if (N == 0) return D3F_OK;
  hipStream_t stream = (hipStream_t)stream_;
  const int grid = d3f::cdiv(N, 4);
  if (C <= 16) det_fwd_kernel<16><<<grid, 256, 0, stream>>>(feat, N, C, idx, H, feat_max, training, scores);
  else if (C <= 32) det_fwd_kernel<32><<<grid, 256, 0, stream>>>(feat, N, C, idx, H, feat_max, training, scores);
  else det_fwd_kernel<64><<<grid, 256, 0, stream>>>(feat, N, C, idx, H, feat_max, training, scores);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

size_t d3f_detection_scores_ws_bytes(int N, int C) { (void)N; (void)C; return 256; }

int d3f_detection_scores_backward(const float* feat, int N, int C, const int32_t* idx, int H, const float* feat_max,
                                  const float* grad_scores, float* grad_feat, void* ws, size_t ws_bytes,
                                  void* stream_) {
  if (!feat || !idx || !feat_max || !grad_scores || !grad_feat || !ws || ws_bytes < 8 || N < 0 || C < 1 || C > 64 ||
      H < 1)
    return D3F_EINVAL;
  if (N == 0) return D3F_OK;
  hipStream_t stream = (hipStream_t)stream_;
  const size_t n = (size_t)N * C;
  if (d3f::zero_async(grad_feat, sizeof(float) * n, stream) != hipSuccess) return D3F_ELAUNCH;
  if (d3f::zero_async(ws, 8, stream) != hipSuccess) return D3F_ELAUNCH;
  const int grid = d3f::cdiv(N, 4);
  if (C <= 16) det_bwd_kernel<16><<<grid, 256, 0, stream>>>(feat, N, C, idx, H, feat_max, grad_scores, grad_feat);
  else if (C <= 32) det_bwd_kernel<32><<<grid, 256, 0, stream>>>(feat, N, C, idx, H, feat_max, grad_scores, grad_feat);
  else det_bwd_kernel<64><<<grid, 256, 0, stream>>>(feat, N, C, idx, H, feat_max, grad_scores, grad_feat);
  int blocks = d3f::cdiv((long long)n, 256 * 8);
  if (blocks > 512) blocks = 512;
  det_reduce_kernel<<<blocks, 256, 0, stream>>>(feat, grad_feat, n, feat_max, (float*)ws);
  det_finalize_kernel<<<d3f::cdiv((long long)n, 256), 256, 0, stream>>>(feat, n, feat_max, (const float*)ws, grad_feat);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

}  // extern "C"
