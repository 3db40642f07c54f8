// Fused bias / residual / LeakyReLU epilogue of the unary and KPConv blocks, forward and backward.
//
// Replaces, per block of the reference network (models/blocks.py): BatchNormBlock's bias add (:473, use_bn=False),
// nn.LeakyReLU(0.1) (:497,:598,:676), the residual add of the bottleneck (:686) and -- in backward -- the two
// elementwise kernels plus the column reduction PyTorch runs for the bias gradient.  ~150 tiny launches per training
// step in the reference graph become one launch forward and one backward per block.
//
//   forward : out[n,c] = act( x[n,c]/d[n] + b1[c] + (add[n,c] + b2[c]) ),  act(v) = v > 0 ? v : slope*v   (slope = 1: identity;
//             d = optional per-row divisor: the KPConv neighbor count when x is the raw (wf @ W) product)
//   backward: gx[n,c]  = go[n,c] * (out[n,c] > 0 ? 1 : slope)     (also the gradient of `add`)
//             gb[c]    = sum_n gx[n,c]                               (gradient of b1 and of b2)
#include "common.hpp"

namespace {

__global__ __launch_bounds__(256) void bias_act_fwd_kernel(const float* __restrict__ x, const float* __restrict__ b1,
                                                           const float* __restrict__ add,
                                                           const float* __restrict__ b2, float slope, size_t n4,
                                                           int C, float* __restrict__ out, float* __restrict__ zinit,
                                                           int zn, const float* __restrict__ row_div,
                                                           const int32_t* __restrict__ add_idx, int idx_stride,
                                                           int add_rows, const float* __restrict__ s_pts = nullptr,
                                                           float4* __restrict__ spack_out = nullptr,
                                                           float* __restrict__ zero_like_out = nullptr) {
  // C % 4 == 0: one float4 per thread, columns of a float4 are c .. c+3
  __shared__ float rowpart[4];
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (zinit)  // side job: clear the caller's backward targets (a few bias rows up to a pooled-gradient matrix)
    for (size_t t = i; t < (size_t)zn; t += (size_t)gridDim.x * blockDim.x) zinit[t] = 0.0f;
  if (spack_out) {
    // the output is the feature matrix of a KPConv: leave its packed supports {x, y, z, [row sum > 0]} behind (what
    // pack_supports_kernel would compute in a launch of its own) and clear that KPConv's scatter target.  A row is
    // TPR = C/4 consecutive threads (a power of two <= 128); whole rows never straddle a workgroup.
    const int TPR = C >> 2;
    const bool live = i < n4;
    const size_t row = live ? (size_t)((uint32_t)i / (uint32_t)TPR) : 0;
    const int c = live ? (int)(((uint32_t)i - (uint32_t)row * (uint32_t)TPR) * 4u) : 0;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
      v = ((const float4*)x)[i];
      if (row_div) { const float d = row_div[row]; v.x /= d; v.y /= d; v.z /= d; v.w /= d; }
      if (b1) { const float4 b = *(const float4*)(b1 + c); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
      if (add) { const float4 a = ((const float4*)add)[i]; v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
      if (b2) { const float4 b = *(const float4*)(b2 + c); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
      v.x = v.x > 0.0f ? v.x : v.x * slope;
      v.y = v.y > 0.0f ? v.y : v.y * slope;
      v.z = v.z > 0.0f ? v.z : v.z * slope;
      v.w = v.w > 0.0f ? v.w : v.w * slope;
      ((float4*)out)[i] = v;
      if (zero_like_out) ((float4*)zero_like_out)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float s = (v.x + v.y) + (v.z + v.w);
    for (int o = 1; o < min(TPR, 64); o <<= 1) s += __shfl_xor(s, o, 64);
    if (TPR == 128) {   // a row spans two waves
      const int wave = threadIdx.x >> 6;
      if ((threadIdx.x & 63) == 0) rowpart[wave] = s;
      __syncthreads();
      s = rowpart[wave & ~1] + rowpart[wave | 1];
    }
    if (live && c == 0)
      spack_out[row] = make_float4(s_pts[3 * row], s_pts[3 * row + 1], s_pts[3 * row + 2], s > 0.0f ? 1.0f : 0.0f);
    return;
  }
  if (i >= n4) return;
  // (row, first column) of this float4; 32-bit arithmetic whenever the matrix has fewer than 2^32 elements -- a 64-bit
  // division by the run-time C costs more than the rest of the thread
  size_t row;
  int c;
  if (n4 <= 0x3fffffffull) {
    const uint32_t e = (uint32_t)i * 4u, r32 = e / (uint32_t)C;
    row = r32;
    c = (int)(e - r32 * (uint32_t)C);
  } else {
    row = (i * 4) / (size_t)C;
    c = (int)((i * 4) % (size_t)C);
  }
  float4 v = ((const float4*)x)[i];
  if (row_div) { const float d = row_div[row]; v.x /= d; v.y /= d; v.z /= d; v.w /= d; }
  if (b1) { const float4 b = *(const float4*)(b1 + c); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
  if (add) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (add_idx) {  // add is a COARSE matrix, row n takes the row of its nearest coarse point (shadow -> zeros)
      const int m = add_idx[row * (size_t)idx_stride];
      if (m >= 0 && m < add_rows) a = *(const float4*)(add + (size_t)m * C + c);
    } else {
      a = ((const float4*)add)[i];
    }
    v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
  }
  if (b2) { const float4 b = *(const float4*)(b2 + c); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
  v.x = v.x > 0.0f ? v.x : v.x * slope;
  v.y = v.y > 0.0f ? v.y : v.y * slope;
  v.z = v.z > 0.0f ? v.z : v.z * slope;
  v.w = v.w > 0.0f ? v.w : v.w * slope;
  ((float4*)out)[i] = v;
}

// scalar fallback for C % 4 != 0
__global__ __launch_bounds__(256) void bias_act_fwd_scalar_kernel(const float* __restrict__ x,
                                                                  const float* __restrict__ b1,
                                                                  const float* __restrict__ add,
                                                                  const float* __restrict__ b2, float slope, size_t n,
                                                                  int C, float* __restrict__ out,
                                                                  float* __restrict__ zinit, int zn,
                                                                  const float* __restrict__ row_div,
                                                                  const int32_t* __restrict__ add_idx,
                                                                  int idx_stride, int add_rows) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (zinit)
    for (size_t t = i; t < (size_t)zn; t += (size_t)gridDim.x * blockDim.x) zinit[t] = 0.0f;
  if (i >= n) return;
  const int c = (int)(i % (size_t)C);
  float v = x[i];
  if (row_div) v /= row_div[i / (size_t)C];
  if (b1) v += b1[c];
  if (add) {
    if (add_idx) {
      const int m = add_idx[(i / (size_t)C) * (size_t)idx_stride];
      if (m >= 0 && m < add_rows) v += add[(size_t)m * C + c];
    } else {
      v += add[i];
    }
  }
  if (b2) v += b2[c];
  out[i] = v > 0.0f ? v : v * slope;
}

// Workgroup (bx, by) owns rows [bx*rows_per_block, ...) x 64 columns [64*by, ...): thread t handles column t & 63 of
// row lane t >> 6 (4 lanes), so every access is a coalesced 256-B row segment; the 4 lane partials are combined in
// LDS and flushed with one atomic per column.  rows_per_block is chosen by the host so that the launch has >= ~1000
// workgroups also for the few-point / 2048-channel layers at the bottom of the U-Net.
__global__ __launch_bounds__(256) void bias_act_bwd_kernel(const float* __restrict__ go, const float* __restrict__ out,
                                                           float slope, int N, int C, int rows_per_block,
                                                           float* __restrict__ gx, float* __restrict__ gb,
                                                           float* __restrict__ gb2,
                                                           const float* __restrict__ row_div,
                                                           float* __restrict__ part) {
  __shared__ float red[256];
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(N, r0 + rows_per_block);
  const int c = blockIdx.y * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  float s = 0.0f;
  if (c < C) {
#pragma unroll 4
    for (int r = r0 + rl; r < r1; r += 4) {
      const size_t i = (size_t)r * C + c;
      const float g = go[i] * (out[i] > 0.0f ? 1.0f : slope);
      if (gx) gx[i] = row_div ? g / row_div[r] : g;  // the bias sums stay undivided
      s += g;
    }
  }
  if (gb) {
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < 64 && c < C) {
      const float v = red[threadIdx.x] + red[threadIdx.x + 64] + red[threadIdx.x + 128] + red[threadIdx.x + 192];
      if (part) {
        part[(size_t)blockIdx.x * C + c] = v;  // two-pass mode: summed by bias_sum_kernel in a fixed order
      } else {
        atomicAdd(&gb[c], v);
        if (gb2) atomicAdd(&gb2[c], v);
      }
    }
  }
}

// Many-row form for C = 16..256 (two-pass mode only): a thread owns one float4 of a row (TPR = C/4 threads per row,
// 256/TPR rows per workgroup step), so a 32-column matrix keeps every lane busy (the column-per-thread kernel above
// idles half of them) and every access is 16 bytes; the per-column partial sums are combined over the row lanes by
// shuffles, over the 4 waves through LDS, and land in part[block][C] for bias_sum_kernel.
template <int TPR>
__global__ __launch_bounds__(256) void bias_act_bwd_v4_kernel(const float4* __restrict__ go,
                                                              const float4* __restrict__ out, float slope, int N,
                                                              int rows_per_block, float4* __restrict__ gx,
                                                              const float* __restrict__ row_div,
                                                              float4* __restrict__ part) {
  constexpr int RL = 256 / TPR;
  __shared__ float4 red[4][TPR];
  const int cl = threadIdx.x % TPR, rl = threadIdx.x / TPR;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(N, r0 + rows_per_block);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int r = r0 + rl; r < r1; r += RL) {
    const size_t i = (size_t)r * TPR + cl;
    const float4 g0 = go[i], o = out[i];
    float4 g;
    g.x = g0.x * (o.x > 0.0f ? 1.0f : slope);
    g.y = g0.y * (o.y > 0.0f ? 1.0f : slope);
    g.z = g0.z * (o.z > 0.0f ? 1.0f : slope);
    g.w = g0.w * (o.w > 0.0f ? 1.0f : slope);
    if (gx) {
      if (row_div) {
        const float d = row_div[r];
        gx[i] = make_float4(g.x / d, g.y / d, g.z / d, g.w / d);  // the bias sums stay undivided
      } else {
        gx[i] = g;
      }
    }
    s.x += g.x; s.y += g.y; s.z += g.z; s.w += g.w;
  }
  if (!part) return;
#pragma unroll
  for (int o = TPR; o < 64; o <<= 1) {
    s.x += __shfl_xor(s.x, o, 64); s.y += __shfl_xor(s.y, o, 64);
    s.z += __shfl_xor(s.z, o, 64); s.w += __shfl_xor(s.w, o, 64);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane < TPR) red[wave][lane] = s;
  __syncthreads();
  if (threadIdx.x < TPR) {
    float4 v = red[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      v.x += red[w][threadIdx.x].x; v.y += red[w][threadIdx.x].y;
      v.z += red[w][threadIdx.x].z; v.w += red[w][threadIdx.x].w;
    }
    part[(size_t)blockIdx.x * TPR + threadIdx.x] = v;
  }
}

// gb[c] = sum_b part[b][c]: with thousands of row blocks the per-column atomics of the one-pass form all hit the same C
// addresses and serialise (38k x 32: 32 us for 15 MB); 4 waves each sum a quarter of the blocks, combined through LDS
__global__ __launch_bounds__(1024) void bias_sum_kernel(const float* __restrict__ part, int nblocks, int C,
                                                        float* __restrict__ gb, float* __restrict__ gb2) {
  // 64 columns x 16 row lanes; every lane sums its share with 4 independent chains, then a fixed-order LDS combine
  __shared__ float sh[16][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int w = threadIdx.x >> 6;
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
  if (c < C) {
    int b = w;
    for (; b + 48 < nblocks; b += 64) {
      s0 += part[(size_t)b * C + c];
      s1 += part[(size_t)(b + 16) * C + c];
      s2 += part[(size_t)(b + 32) * C + c];
      s3 += part[(size_t)(b + 48) * C + c];
    }
    for (; b < nblocks; b += 16) s0 += part[(size_t)b * C + c];
  }
  sh[w][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (w == 0 && c < C) {
    float v = sh[0][threadIdx.x];
#pragma unroll
    for (int j = 1; j < 16; ++j) v += sh[j][threadIdx.x];
    gb[c] = v;
    if (gb2) gb2[c] = v;
  }
}

}  // namespace

extern "C" {

int d3f_bias_act_forward(const float* x, const float* bias1, const float* add, const float* bias2, float slope, int N,
                         int C, float* out, float* zero_init, int zero_n, const float* row_div, const int32_t* add_idx,
                         int idx_stride, int add_rows, void* stream) {
  if (!x || !out || N < 0 || C < 1 || (zero_init && zero_n < 1)) return D3F_EINVAL;
  const size_t n = (size_t)N * C;
  if (n == 0) {
    if (zero_init && d3f::zero_async(zero_init, sizeof(float) * (size_t)zero_n, (hipStream_t)stream) != hipSuccess)
      return D3F_ELAUNCH;
    return D3F_OK;
  }
  if (C % 4 == 0)
    bias_act_fwd_kernel<<<d3f::cdiv((long long)(n / 4), 256), 256, 0, (hipStream_t)stream>>>(
        x, bias1, add, bias2, slope, n / 4, C, out, zero_init, zero_n, row_div, add_idx, idx_stride, add_rows);
  else
    bias_act_fwd_scalar_kernel<<<d3f::cdiv((long long)n, 256), 256, 0, (hipStream_t)stream>>>(
        x, bias1, add, bias2, slope, n, C, out, zero_init, zero_n, row_div, add_idx, idx_stride, add_rows);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

int d3f_bias_act_packs(int C) { return (C == 16 || C == 32 || C == 64 || C == 128 || C == 256 || C == 512) ? 1 : 0; }

int d3f_bias_act_forward_pack(const float* x, const float* bias1, const float* add, const float* bias2, float slope,
                              int N, int C, float* out, float* zero_init, int zero_n, const float* row_div,
                              const int32_t* add_idx, int idx_stride, int add_rows, const float* s_pts,
                              void* spack_out, float* zero_like_out, void* stream) {
  if (!s_pts || !spack_out) return D3F_EINVAL;
  if (!x || !out || N < 1 || !d3f_bias_act_packs(C) || (zero_init && zero_n < 1) || add_idx ||
      (long long)N * C >= (1ll << 32))
    return D3F_EINVAL;
  (void)idx_stride; (void)add_rows;
  const size_t n = (size_t)N * C;
  bias_act_fwd_kernel<<<d3f::cdiv((long long)(n / 4), 256), 256, 0, (hipStream_t)stream>>>(
      x, bias1, add, bias2, slope, n / 4, C, out, zero_init, zero_n, row_div, nullptr, 0, 0, s_pts, (float4*)spack_out,
      zero_like_out);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

/* grad_x (optional) [N,C]; grad_bias / grad_bias2 (optional) [C] receive the SAME column sums (the two biases of a
 * unary block are distinct parameters with identical gradients).  With bias_prezeroed = 0 they are zeroed here;
 * with 1 the caller guarantees zeros (d3f_bias_act_forward's zero_init). */
size_t d3f_bias_act_backward_ws_bytes(int N, int C) {
  return N >= 4096 ? d3f::align_up(sizeof(float) * (size_t)d3f::cdiv(N, 64) * (size_t)C, 256) : 0;
}

/* number of partial-sum rows the two-pass form writes into ws ([blocks, C] floats); 0: the one-pass form applies */
int d3f_bias_act_backward_blocks(int N, int C) { return (N >= 4096 && C >= 1) ? d3f::cdiv(N, 64) : 0; }

static int bias_act_backward_impl(const float* grad_out, const float* out, float slope, int N, int C, float* grad_x,
                                  float* grad_bias, float* grad_bias2, int bias_prezeroed, const float* row_div,
                                  void* ws, size_t ws_bytes, void* stream, bool partial_only);

/* First pass only: grad_x (optional) and the per-block partial column sums in ws [d3f_bias_act_backward_blocks, C];
 * the caller finishes the bias gradient with d3f_linear_grad_weight_bias (inside the weight gradient's second-stage
 * launch) or d3f_bias_sum. */
int d3f_bias_act_backward_partial(const float* grad_out, const float* out, float slope, int N, int C, float* grad_x,
                                  const float* row_div, void* ws, size_t ws_bytes, void* stream) {
  if (!grad_out || !out || !ws || N < 4096 || C < 1 || ws_bytes < d3f_bias_act_backward_ws_bytes(N, C))
    return D3F_EINVAL;
  return bias_act_backward_impl(grad_out, out, slope, N, C, grad_x, (float*)ws, nullptr, 1, row_div, ws, ws_bytes,
                                stream, true);
}

int d3f_bias_sum(const float* part, int blocks, int C, float* grad_bias, float* grad_bias2, void* stream) {
  if (!part || !grad_bias || blocks < 1 || C < 1) return D3F_EINVAL;
  bias_sum_kernel<<<d3f::cdiv(C, 64), 1024, 0, (hipStream_t)stream>>>(part, blocks, C, grad_bias, grad_bias2);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

int d3f_bias_act_backward(const float* grad_out, const float* out, float slope, int N, int C, float* grad_x,
                          float* grad_bias, float* grad_bias2, int bias_prezeroed, const float* row_div, void* ws,
                          size_t ws_bytes, void* stream) {
  return bias_act_backward_impl(grad_out, out, slope, N, C, grad_x, grad_bias, grad_bias2, bias_prezeroed, row_div, ws,
                                ws_bytes, stream, false);
}

static int bias_act_backward_impl(const float* grad_out, const float* out, float slope, int N, int C, float* grad_x,
                                  float* grad_bias, float* grad_bias2, int bias_prezeroed, const float* row_div,
                                  void* ws, size_t ws_bytes, void* stream, bool partial_only) {
  if (!grad_out || !out || N < 0 || C < 1 || (!grad_x && !grad_bias) || (grad_bias2 && !grad_bias)) return D3F_EINVAL;
  if (grad_bias && N >= 4096 && ws && ws_bytes >= d3f_bias_act_backward_ws_bytes(N, C)) {
    // many rows: per-block partial column sums + a second, tiny launch instead of contended atomics (deterministic)
    const int rows = 64;
    dim3 grid(d3f::cdiv(N, rows), d3f::cdiv(C, 64));
    const int tpr = C / 4;
    const bool v4 = C % 4 == 0 && (tpr == 4 || tpr == 8 || tpr == 16 || tpr == 32 || tpr == 64) &&
                    ((uintptr_t)grad_out | (uintptr_t)out | (uintptr_t)grad_x | (uintptr_t)ws) % 16 == 0;
#define D3F_BAB(T)                                                                                                  \
  bias_act_bwd_v4_kernel<T><<<grid.x, 256, 0, (hipStream_t)stream>>>((const float4*)grad_out, (const float4*)out, slope, \
                                                                     N, rows, (float4*)grad_x, row_div, (float4*)ws)
    if (v4 && tpr == 4) D3F_BAB(4);
    else if (v4 && tpr == 8) D3F_BAB(8);
    else if (v4 && tpr == 16) D3F_BAB(16);
    else if (v4 && tpr == 32) D3F_BAB(32);
    else if (v4 && tpr == 64) D3F_BAB(64);
    else
#undef D3F_BAB
    bias_act_bwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(grad_out, out, slope, N, C, rows, grad_x, grad_bias,
                                                               grad_bias2, row_div, (float*)ws);
    if (!partial_only)
      bias_sum_kernel<<<d3f::cdiv(C, 64), 1024, 0, (hipStream_t)stream>>>((const float*)ws, (int)grid.x, C, grad_bias,
                                                                         grad_bias2);
    D3F_LAUNCH_CHECK();
    return D3F_OK;
  }
  if (partial_only) return D3F_EINVAL;
  if (!bias_prezeroed) {
    if (grad_bias && d3f::zero_async(grad_bias, sizeof(float) * (size_t)C, (hipStream_t)stream) != hipSuccess)
      return D3F_ELAUNCH;
    if (grad_bias2 && d3f::zero_async(grad_bias2, sizeof(float) * (size_t)C, (hipStream_t)stream) != hipSuccess)
      return D3F_ELAUNCH;
  }
  if (N == 0) return D3F_OK;
  const int cblocks = d3f::cdiv(C, 64);
  int rows = 256;  // fewer atomics per column when there are plenty of rows
  while (rows > 16 && (long long)d3f::cdiv(N, rows) * cblocks < 1024) rows >>= 1;
  dim3 grid(d3f::cdiv(N, rows), cblocks);
  bias_act_bwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(grad_out, out, slope, N, C, rows, grad_x, grad_bias,
                                                             grad_bias2, row_div, nullptr);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

}  // extern "C"
