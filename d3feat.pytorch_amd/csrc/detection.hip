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
                                                      float* __restrict__ scores, const int32_t* __restrict__ width,
                                                      d3f::RowGroups rg) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const int c = lane % CP, g = lane / CP;
  const int gi = d3f::group_of_row(rg, n);   // stacked pairs: the normaliser and the table width of the row's own pair
  fmax += gi;
  if (width) width += gi;
  const float denom = fmaxf(*fmax, 0.0f) + 1e-6f;  // the reference's max includes its zero shadow row (:336-342)
  // columns of the table the reference would have built: min(H, max neighbor count) (dataloader.py:64-66); the columns
  // past it are all shadow and only matter to the local-maximum gate (a zero candidate the reference does not have)
  const int Hw = width ? min(H, max(1, __builtin_amdgcn_readfirstlane(*width))) : H;
  PointStats s = point_stats<CP>(feat, N, C, idx + (size_t)n * H, Hw, denom, c, g);
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
                                                      const float* __restrict__ gscore, float* __restrict__ df,
                                                      d3f::RowGroups rg) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const int c = lane % CP, g = lane / CP;
  fmax += d3f::group_of_row(rg, n);
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

// ---- 16-byte gathers: lane = (4 channels, neighbor group); LP = C/4 lanes serve one neighbor row, 64/LP rows per step.
// Training additionally leaves 8 scalars per point behind (aux) so that the backward pass needs no feature gather:
//   {f*, alpha*, beta*, u*, dmax, num, c* (int bits), c' (int bits)}   (* = winning channel, ' = channel of max_c f)
template <int LP>
__global__ __launch_bounds__(256) void det_fwd_v4_kernel(const float* __restrict__ feat, int N,
                                                         const int32_t* __restrict__ idx, int H,
                                                         const float* __restrict__ fmax, int training,
                                                         float* __restrict__ scores, float* __restrict__ aux,
                                                         const int32_t* __restrict__ width, d3f::RowGroups rg) {
  constexpr int C = 4 * LP, G = 64 / LP;
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const int c4 = lane % LP, g = lane / LP;
  const int gi = d3f::group_of_row(rg, n);   // see det_fwd_kernel
  fmax += gi;
  if (width) width += gi;
  const float denom = fmaxf(*fmax, 0.0f) + 1e-6f;  // the reference's max includes its zero shadow row (:336-342)
  const int32_t* row = idx + (size_t)n * H;
  float4 msum = make_float4(0.f, 0.f, 0.f, 0.f), lmax = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  int cnt = 0;
  // the whole index row in one coalesced load (H <= 64), then all gathers of a batch of steps issued back to back:
  // one memory latency per batch instead of two dependent ones per step
  const int mrow = lane < H ? row[lane] : N;
  const int Hw = width ? min(H, max(1, __builtin_amdgcn_readfirstlane(*width))) : H;  // see det_fwd_kernel
  constexpr int SB = 4;  // steps per batch
  for (int h0 = 0; h0 < Hw; h0 += G * SB) {
    float4 raw[SB];
    bool live[SB];
#pragma unroll
    for (int s = 0; s < SB; ++s) {
      const int h = h0 + s * G + g;
      const int m = __shfl(mrow, h & 63, 64);
      live[s] = h < Hw && m >= 0 && m < N;
      raw[s] = live[s] ? *(const float4*)(feat + (size_t)m * C + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int s = 0; s < SB; ++s) {
      const int h = h0 + s * G + g;
      // un-normalised: max_i (x_i / d) = (max_i x_i) / d exactly (a correctly rounded division is monotonic) and
      // sum_i (x_i / d) = (sum_i x_i) / d up to rounding -- eight divisions per lane after the loop instead of four per
      // neighbor (160 per lane at 40 neighbors): 75.8 -> 62 us at level 0
      const float4 v = live[s] ? raw[s] : make_float4(0.f, 0.f, 0.f, 0.f);
      const float rs = group_sum<LP>((v.x + v.y) + (v.z + v.w));
      if (h < Hw) {
        cnt += rs != 0.0f;
        msum.x += v.x; msum.y += v.y; msum.z += v.z; msum.w += v.w;
        lmax.x = fmaxf(lmax.x, v.x); lmax.y = fmaxf(lmax.y, v.y); lmax.z = fmaxf(lmax.z, v.z); lmax.w = fmaxf(lmax.w, v.w);
      }
    }
  }
#pragma unroll
  for (int o = LP; o < 64; o <<= 1) {
    msum.x += __shfl_xor(msum.x, o, 64); msum.y += __shfl_xor(msum.y, o, 64);
    msum.z += __shfl_xor(msum.z, o, 64); msum.w += __shfl_xor(msum.w, o, 64);
    cnt += __shfl_xor(cnt, o, 64);
    lmax.x = fmaxf(lmax.x, __shfl_xor(lmax.x, o, 64)); lmax.y = fmaxf(lmax.y, __shfl_xor(lmax.y, o, 64));
    lmax.z = fmaxf(lmax.z, __shfl_xor(lmax.z, o, 64)); lmax.w = fmaxf(lmax.w, __shfl_xor(lmax.w, o, 64));
  }
  const float num = (float)(cnt > 1 ? cnt : 1);
  const float4 t = *(const float4*)(feat + (size_t)n * C + 4 * c4);
  const float fs[4] = {t.x / denom, t.y / denom, t.z / denom, t.w / denom};
  const float mean[4] = {msum.x / denom / num, msum.y / denom / num, msum.z / denom / num, msum.w / denom / num};
  const float lm[4] = {lmax.x / denom, lmax.y / denom, lmax.z / denom, lmax.w / denom};
  const float dmax = group_max<LP>(fmaxf(fmaxf(fs[0], fs[1]), fmaxf(fs[2], fs[3])));
  float u[4], al[4], be[4], sc[4];
  float best = -INFINITY;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    u[k] = fs[k] - mean[k];
    al[k] = softplus_t(u[k]);
    be[k] = fs[k] / (1e-6f + dmax);
    sc[k] = al[k] * be[k];
    best = fmaxf(best, sc[k]);
  }
  best = group_max<LP>(best);
  float score = best;
  if (!training) {
    float is = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) is = fmaxf(is, fs[k] == lm[k] ? 1.0f : 0.0f);
    score *= group_max<LP>(is);
  }
  if (lane == 0) scores[n] = score;
  if (aux) {
    // first channel attaining the max (torch.max returns the first maximal index), and first channel of max_c f
    int cs = 0x7fffffff, cp = 0x7fffffff;
#pragma unroll
    for (int k = 3; k >= 0; --k) {
      if (sc[k] == best) cs = 4 * c4 + k;
      if (fs[k] == dmax) cp = 4 * c4 + k;
    }
#pragma unroll
    for (int o = LP >> 1; o > 0; o >>= 1) {
      cs = min(cs, __shfl_xor(cs, o, 64));
      cp = min(cp, __shfl_xor(cp, o, 64));
    }
    // the lane that owns channel c* publishes its values
    if (g == 0 && (cs >> 2) == c4) {
      const int k = cs & 3;
      float* a = aux + (size_t)n * 8;
      a[0] = fs[k]; a[1] = al[k]; a[2] = be[k]; a[3] = u[k];
      a[4] = dmax; a[5] = num; a[6] = __int_as_float(cs); a[7] = __int_as_float(cp);
    }
  }
}

// backward from aux: no feature gather; one THREAD per (point, neighbor slot) -- the index table is read fully
// coalesced and a wave's 64 atomics belong to 1-2 points (one wave per point is dispatch-rate bound: 80 us).
// (Round 3: a GATHER form over the exact-form transposed table -- one wave per point, (channel, value) pairs of the
// reverse neighbors handed round with readlane, no atomics, no zero fill, S and the tie count folded in -- was built,
// passed the parity tests and took 123 us against 81 + 15 + 5 here: again one short dependent chain per wave.)
__global__ __launch_bounds__(256) void det_bwd_aux_kernel(const float* __restrict__ aux, int N, int C,
                                                          const int32_t* __restrict__ idx, int H,
                                                          const float* __restrict__ gscore, float* __restrict__ df) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)N * H) return;
  int n, h;
  if ((size_t)N * H < (1ull << 32)) {   // 32-bit division (the 64-bit one is a long software routine per thread)
    const uint32_t q = (uint32_t)t / (uint32_t)H;
    n = (int)q;
    h = (int)((uint32_t)t - q * (uint32_t)H);
  } else {
    n = (int)(t / H);
    h = (int)(t % H);
  }
  // The detector loss reads the scores of the correspondences only (reference utils/loss.py:140-158 on
  // scores[corr], trainer.py:96-101): the incoming gradient is zero at all but a few hundred of the N points, and a point
  // without gradient would add +-0 to 1 + H addresses -- nothing, bit for bit (grad_feat starts at 0).  Whole waves leave
  // here: 233 -> 12 us per 3-pair stack.
  const float ds = gscore[n];
  if (ds == 0.0f) return;
  const float4 a0 = *(const float4*)(aux + (size_t)n * 8);      // f*, alpha*, beta*, u*
  const float4 a1 = *(const float4*)(aux + (size_t)n * 8 + 4);  // dmax, num, c*, c'
  const int cstar = __float_as_int(a1.z);
  const float sig = a0.w > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-a0.w));
  const float du = ds * a0.z * sig;
  if (h == 0) {
    const float inv = 1.0f / (1e-6f + a1.x);
    atomicAdd(&df[(size_t)n * C + cstar], du + ds * a0.y * inv);
    atomicAdd(&df[(size_t)n * C + __float_as_int(a1.w)], -ds * a0.y * a0.x * inv * inv);
  }
  const int m = idx[t];
  if (m >= 0 && m < N) atomicAdd(&df[(size_t)m * C + cstar], -du / a1.y);
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
// blockIdx.y = group of `group` consecutive clouds (group <= 0: all B clouds are one group)
__global__ __launch_bounds__(256) void gmax_rows_kernel(const float* __restrict__ x, int cap_rows, int C,
                                                        const int32_t* __restrict__ len, int B, int group,
                                                        uint32_t* __restrict__ enc) {
  const int b0 = group > 0 ? blockIdx.y * group : 0, b1 = group > 0 ? min(B, b0 + group) : B;
  const int r0 = min(cap_rows, d3f::batch_offset(len, b0)), r1 = min(cap_rows, d3f::batch_offset(len, b1));
  const size_t beg = (size_t)r0 * C, n = (size_t)r1 * C;
  float m = -INFINITY;
  for (size_t i = beg + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    m = fmaxf(m, x[i]);
  m = block_max256(m);
  if (threadIdx.x == 0) atomicMax(enc + blockIdx.y, f2ord(m));
}
__global__ void gmax_final_kernel(const uint32_t* __restrict__ enc, float* __restrict__ out, int n = 1) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = ord2f(enc[i]);
}

// Element range of group blockIdx.y (all rows when there are no groups; the last group also owns the padding rows of a
// capacity-shaped batch, whose gradient is zero)
__device__ __forceinline__ void group_range(const d3f::RowGroups& rg, int cap_rows, int C, size_t& beg, size_t& end) {
  if (!rg.len || rg.group <= 0) { beg = 0; end = (size_t)cap_rows * C; return; }
  const int b0 = blockIdx.y * rg.group, b1 = min(rg.B, b0 + rg.group);
  const int r0 = min(cap_rows, d3f::batch_offset(rg.len, b0));
  const int r1 = (b1 >= rg.B) ? cap_rows : min(cap_rows, d3f::batch_offset(rg.len, b1));
  beg = (size_t)r0 * C;
  end = (size_t)r1 * C;
}

// S = sum(df * f), ties = #{feat == fmax}; per group of clouds (blockIdx.y) and block: part[(g * DET_RED_BLOCKS + block) * 2
// + {0, 1}] -- partial sums, added up in block order by the finalize kernel (no float atomics: the tie term has the same bits
// on every replay)
constexpr int DET_RED_BLOCKS = 128;
__global__ __launch_bounds__(256) void det_reduce_kernel(const float* __restrict__ feat, const float* __restrict__ df,
                                                         int cap_rows, int C, const float* __restrict__ fmax,
                                                         float* __restrict__ part, d3f::RowGroups rg) {
  size_t beg, n;
  group_range(rg, cap_rows, C, beg, n);
  fmax += blockIdx.y;
  float* acc = part + ((size_t)blockIdx.y * DET_RED_BLOCKS + blockIdx.x) * 2;
  const float mx = fmaxf(*fmax, 0.0f), denom = mx + 1e-6f;
  float s = 0.0f, t = (blockIdx.x == 0 && threadIdx.x == 0 && mx == 0.0f) ? 1.0f : 0.0f;  // the zero shadow row ties at 0
  for (size_t i = beg + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
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
    acc[0] = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
    acc[1] = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
  }
}

// grad_feat = df / g  +  [feat == max] * ( -S / g ) / ties        (g = max + 1e-6; d g / d feat flows to the arg-max)
__global__ void det_finalize_kernel(const float* __restrict__ feat, int cap_rows, int C, const float* __restrict__ fmax,
                                    const float* __restrict__ part, int red_blocks, float* __restrict__ gfeat,
                                    d3f::RowGroups rg) {
  size_t beg, n;
  group_range(rg, cap_rows, C, beg, n);
  fmax += blockIdx.y;
  part += (size_t)blockIdx.y * DET_RED_BLOCKS * 2;
  // the reduce kernel's per-block partials in block order: the first wave sums them (two per lane, butterfly), every
  // workgroup the same way
  __shared__ float tot[2];
  if (threadIdx.x < 64) {
    float s = 0.0f, t = 0.0f;
    for (int b = threadIdx.x; b < red_blocks; b += 64) {
      s += part[2 * b];
      t += part[2 * b + 1];
    }
    s = d3f::wave_sum(s);
    t = d3f::wave_sum(t);
    if (threadIdx.x == 0) {
      tot[0] = s;
      tot[1] = t;
    }
  }
  __syncthreads();
  const float mx = fmaxf(*fmax, 0.0f), denom = mx + 1e-6f;
  const float tie_term = (-tot[0] / denom) / fmaxf(tot[1], 1.0f);
  for (size_t i = beg + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float gval = gfeat[i] / denom;
    if (feat[i] == mx) gval += tie_term;
    gfeat[i] = gval;
  }
}

}  // namespace

extern "C" {

int d3f_global_max(const float* x, size_t n, float* out_max, void* ws, size_t ws_bytes, void* stream_) {
  if (!x || !out_max || !ws || ws_bytes < 4 || n == 0) return D3F_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  if (d3f::zero_async(ws, 4, stream) != hipSuccess) return D3F_ELAUNCH;
  int blocks = d3f::cdiv((long long)n, 256 * 8);
  if (blocks > 128) blocks = 128;  // one same-address atomic per block: ~14 ns each when they pile up
  gmax_partial_kernel<<<blocks, 256, 0, stream>>>(x, n, (uint32_t*)ws);
  gmax_final_kernel<<<1, 1, 0, stream>>>((const uint32_t*)ws, out_max);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

int d3f_global_max_groups(const float* x, int cap_rows, int C, const int32_t* len, int B, int group, float* out_max,
                          void* ws, size_t ws_bytes, void* stream_) {
  if (!x || !out_max || !ws || !len || cap_rows < 1 || C < 1 || B < 1 || B > D3F_MAX_BATCH || group < 0)
    return D3F_EINVAL;
  const int G = group > 0 ? d3f::cdiv(B, group) : 1;
  if (ws_bytes < 4 * (size_t)G) return D3F_EWORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  if (d3f::zero_async(ws, 4 * (size_t)G, stream) != hipSuccess) return D3F_ELAUNCH;
  int blocks = d3f::cdiv((long long)cap_rows * C, 256 * 8 * G);
  if (blocks > 128) blocks = 128;
  if (blocks < 1) blocks = 1;
  gmax_rows_kernel<<<dim3(blocks, G), 256, 0, stream>>>(x, cap_rows, C, len, B, group, (uint32_t*)ws);
  gmax_final_kernel<<<d3f::cdiv(G, 64), 64, 0, stream>>>((const uint32_t*)ws, out_max, G);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

int d3f_global_max_rows(const float* x, int cap_rows, int C, const int32_t* len, int B, float* out_max, void* ws,
                        size_t ws_bytes, void* stream_) {
  if (ws_bytes < 4) return D3F_EINVAL;
  return d3f_global_max_groups(x, cap_rows, C, len, B, 0, out_max, ws, ws_bytes, stream_);
}

int d3f_detection_scores_aux_floats(int C) { return (C == 16 || C == 32 || C == 64) ? 8 : 0; }  /* and H <= 64 */

int d3f_detection_scores_forward(const float* feat, int N, int C, const int32_t* idx, int H, const float* feat_max,
                                 int training, float* scores, float* aux, const int32_t* width, const int32_t* len,
                                 int B, int group, void* stream_) {
  if (!feat || !idx || !feat_max || !scores || N < 0 || C < 1 || C > 64 || H < 1) return D3F_EINVAL;
  if (len && (B < 1 || B > D3F_MAX_BATCH || group < 1)) return D3F_EINVAL;
  const d3f::RowGroups rg = {len, B, group};
  if (aux && (!training || !d3f_detection_scores_aux_floats(C) || H > 64)) return D3F_EINVAL;
  if (N == 0) return D3F_OK;
  hipStream_t stream = (hipStream_t)stream_;
  const int grid = d3f::cdiv(N, 4);
  if ((C == 16 || C == 32 || C == 64) && H <= 64) {
    if (C == 16) det_fwd_v4_kernel<4><<<grid, 256, 0, stream>>>(feat, N, idx, H, feat_max, training, scores, aux, width, rg);
    else if (C == 32) det_fwd_v4_kernel<8><<<grid, 256, 0, stream>>>(feat, N, idx, H, feat_max, training, scores, aux, width, rg);
    else det_fwd_v4_kernel<16><<<grid, 256, 0, stream>>>(feat, N, idx, H, feat_max, training, scores, aux, width, rg);
    D3F_LAUNCH_CHECK();
    return D3F_OK;
  }
  if (C <= 16) det_fwd_kernel<16><<<grid, 256, 0, stream>>>(feat, N, C, idx, H, feat_max, training, scores, width, rg);
  else if (C <= 32) det_fwd_kernel<32><<<grid, 256, 0, stream>>>(feat, N, C, idx, H, feat_max, training, scores, width, rg);
  else det_fwd_kernel<64><<<grid, 256, 0, stream>>>(feat, N, C, idx, H, feat_max, training, scores, width, rg);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

size_t d3f_detection_scores_ws_bytes(int N, int C) {
  (void)N;
  (void)C;
  return 8 * (size_t)DET_RED_BLOCKS * D3F_MAX_BATCH;   // per-block partial sums of every group of clouds
}

static int det_backward_impl(const float* feat, int N, int C, const int32_t* idx, int H, const float* feat_max,
                             const float* grad_scores, const float* aux, float* grad_feat, const int32_t* len, int B,
                             int group, void* ws, size_t ws_bytes, void* stream_) {
  if (!feat || !idx || !feat_max || !grad_scores || !grad_feat || !ws || N < 0 || C < 1 || C > 64 || H < 1)
    return D3F_EINVAL;
  if (len && (B < 1 || B > D3F_MAX_BATCH || group < 1)) return D3F_EINVAL;
  const int G = len ? d3f::cdiv(B, group) : 1;
  if (ws_bytes < 8 * (size_t)G * DET_RED_BLOCKS) return D3F_EWORKSPACE;
  if (N == 0) return D3F_OK;
  const d3f::RowGroups rg = {len, B, group};
  hipStream_t stream = (hipStream_t)stream_;
  const size_t n = (size_t)N * C;
  if (d3f::zero_async(grad_feat, sizeof(float) * n, stream) != hipSuccess) return D3F_ELAUNCH;
  const int grid = d3f::cdiv(N, 4);
  if (aux) det_bwd_aux_kernel<<<d3f::cdiv((long long)N * H, 256), 256, 0, stream>>>(aux, N, C, idx, H, grad_scores, grad_feat);
  else if (C <= 16) det_bwd_kernel<16><<<grid, 256, 0, stream>>>(feat, N, C, idx, H, feat_max, grad_scores, grad_feat, rg);
  else if (C <= 32) det_bwd_kernel<32><<<grid, 256, 0, stream>>>(feat, N, C, idx, H, feat_max, grad_scores, grad_feat, rg);
  else det_bwd_kernel<64><<<grid, 256, 0, stream>>>(feat, N, C, idx, H, feat_max, grad_scores, grad_feat, rg);
  int blocks = d3f::cdiv((long long)n, 256 * 8 * G);
  if (blocks > DET_RED_BLOCKS) blocks = DET_RED_BLOCKS;
  if (blocks < 1) blocks = 1;
  det_reduce_kernel<<<dim3(blocks, G), 256, 0, stream>>>(feat, grad_feat, N, C, feat_max, (float*)ws, rg);
  int fblocks = d3f::cdiv((long long)n, 256 * G);
  if (fblocks < 1) fblocks = 1;
  det_finalize_kernel<<<dim3(fblocks, G), 256, 0, stream>>>(feat, N, C, feat_max, (const float*)ws, blocks, grad_feat, rg);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

int d3f_detection_scores_backward(const float* feat, int N, int C, const int32_t* idx, int H, const float* feat_max,
                                  const float* grad_scores, const float* aux, float* grad_feat, void* ws,
                                  size_t ws_bytes, void* stream_) {
  return det_backward_impl(feat, N, C, idx, H, feat_max, grad_scores, aux, grad_feat, nullptr, 0, 0, ws, ws_bytes,
                           stream_);
}

int d3f_detection_scores_backward_groups(const float* feat, int N, int C, const int32_t* idx, int H,
                                         const float* feat_max, const float* grad_scores, const float* aux,
                                         float* grad_feat, const int32_t* len, int B, int group, void* ws,
                                         size_t ws_bytes, void* stream_) {
  if (!len) return D3F_EINVAL;
  return det_backward_impl(feat, N, C, idx, H, feat_max, grad_scores, aux, grad_feat, len, B, group, ws, ws_bytes,
                           stream_);
}

}  // extern "C"
