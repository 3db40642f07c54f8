// KPConv forward / backward (rigid kernel points, 'linear' influence, 'sum' aggregation).
//
// Replaces reference models/blocks.py:237-382 (KPConv.forward) and the autograd graph PyTorch builds for it.
//   wf[n,k,c] = sum_h w[n,h,k] * x[idx[n,h],c]            w = max(0, 1 - sqrt(|(s[idx]-q[n]) - kp[k]|^2)/extent)
//   out[n,:]  = ( sum_k wf[n,k,:] @ W[k] ) / nn[n]         nn = max(1, #{h : sum_c x[idx[n,h],c] > 0})
// The reference materialises [N,H,K,3], [N,H,K], [N,K,H], [N,H,Cin], [N,K,Cin], [K,N,Cout]
// (blocks.py:280-374); here the neighbor gather, the influence weights and the K-way aggregation are one kernel
// and the (K*Cin)->Cout contraction runs on the f32 matrix cores (v_mfma_f32_16x16x4_f32: exact f32 FMA chains).
//
// This file holds the GENERAL path (any Cin/Cout/H/K<=16): wave-per-query aggregation into a [Nq,K*Cin] scratch +
// MFMA GEMMs.  kpconv_fused.hip holds the LDS-tiled fused kernels used for the channel widths of the D3Feat net.
#include "common.hpp"
#include "kpconv_modes.hpp"

namespace d3f {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// Aggregation: one wave per query, lanes <-> input channels (CPL channels per lane).
// The influence weight of kernel point k is computed by lane k and broadcast with v_readlane.
// ------------------------------------------------------------------------------------------------
template <int CPL, bool WRITE_NN>
__global__ __launch_bounds__(256) void kpconv_wf_kernel(const float* __restrict__ q_pts,
                                                        const float* __restrict__ s_pts,
                                                        const int32_t* __restrict__ idx,
                                                        const float* __restrict__ x,
                                                        const float* __restrict__ kp, int Nq, int Ns, int H, int Cin,
                                                        int K, float extent, float* __restrict__ wf,
                                                        float* __restrict__ nn, float gauss_denom = 1.0f,
                                                        int mode = 0) {
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= Nq) return;
  const float qx = q_pts[3 * (size_t)q + 0], qy = q_pts[3 * (size_t)q + 1], qz = q_pts[3 * (size_t)q + 2];
  const int kk = lane & 15;
  const bool klive = kk < K;
  const float kx = klive ? kp[3 * kk + 0] : 0.0f, ky = klive ? kp[3 * kk + 1] : 0.0f, kz = klive ? kp[3 * kk + 2] : 0.0f;
  float acc[16][CPL];
#pragma unroll
  for (int k = 0; k < 16; ++k)
#pragma unroll
    for (int j = 0; j < CPL; ++j) acc[k][j] = 0.0f;
  int cnt = 0;
  const int32_t* row = idx + (size_t)q * H;
  for (int h = 0; h < H; ++h) {
    const int n = row[h];
    if (n < 0 || n >= Ns) continue;  // shadow neighbor: zero weight, zero feature (blocks.py:277,356)
    const float rx = s_pts[3 * (size_t)n + 0] - qx, ry = s_pts[3 * (size_t)n + 1] - qy,
                rz = s_pts[3 * (size_t)n + 2] - qz;
    const float dx = rx - kx, dy = ry - ky, dz = rz - kz;
    const float d2 = dx * dx + dy * dy + dz * dz;
    const float w = influence_weight(d2, klive, kk, extent, gauss_denom, mode);
    float xs[CPL];
    float rs = 0.0f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const int c = lane + 64 * j;
      xs[j] = c < Cin ? x[(size_t)n * Cin + c] : 0.0f;
      rs += xs[j];
    }
    if (WRITE_NN) {
      rs = wave_sum(rs);
      cnt += rs > 0.0f;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float wk = __shfl(w, k, 64);
#pragma unroll
      for (int j = 0; j < CPL; ++j) acc[k][j] = fmaf(wk, xs[j], acc[k][j]);
    }
  }
  float* o = wf + (size_t)q * K * Cin;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    if (k < K) {
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        const int c = lane + 64 * j;
        if (c < Cin) o[(size_t)k * Cin + c] = acc[k][j];
      }
    }
  }
  if (WRITE_NN && lane == 0) nn[q] = (float)(cnt > 1 ? cnt : 1);
}

// ------------------------------------------------------------------------------------------------
// grad_x scatter:  grad_x[idx[n,h], c] += sum_k w[n,h,k] * gW[n,k,c]      (gW = (grad_out/nn) @ W^T)
// ------------------------------------------------------------------------------------------------
template <int CPL>
__global__ __launch_bounds__(256) void kpconv_dx_kernel(const float* __restrict__ q_pts,
                                                        const float* __restrict__ s_pts,
                                                        const int32_t* __restrict__ idx,
                                                        const float* __restrict__ kp, int Nq, int Ns, int H, int Cin,
                                                        int K, float extent, const float* __restrict__ gW,
                                                        float* __restrict__ grad_x, float gauss_denom = 1.0f,
                                                        int mode = 0) {
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= Nq) return;
  const float qx = q_pts[3 * (size_t)q + 0], qy = q_pts[3 * (size_t)q + 1], qz = q_pts[3 * (size_t)q + 2];
  const int kk = lane & 15;
  const bool klive = kk < K;
  const float kx = klive ? kp[3 * kk + 0] : 0.0f, ky = klive ? kp[3 * kk + 1] : 0.0f, kz = klive ? kp[3 * kk + 2] : 0.0f;
  float g[16][CPL];
  const float* gq = gW + (size_t)q * K * Cin;
#pragma unroll
  for (int k = 0; k < 16; ++k)
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const int c = lane + 64 * j;
      g[k][j] = (k < K && c < Cin) ? gq[(size_t)k * Cin + c] : 0.0f;
    }
  const int32_t* row = idx + (size_t)q * H;
  for (int h = 0; h < H; ++h) {
    const int n = row[h];
    if (n < 0 || n >= Ns) continue;
    const float rx = s_pts[3 * (size_t)n + 0] - qx, ry = s_pts[3 * (size_t)n + 1] - qy,
                rz = s_pts[3 * (size_t)n + 2] - qz;
    const float dx = rx - kx, dy = ry - ky, dz = rz - kz;
    const float d2 = dx * dx + dy * dy + dz * dz;
    const float w = influence_weight(d2, klive, kk, extent, gauss_denom, mode);
    float e[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) e[j] = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float wk = __shfl(w, k, 64);
#pragma unroll
      for (int j = 0; j < CPL; ++j) e[j] = fmaf(wk, g[k][j], e[j]);
    }
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const int c = lane + 64 * j;
      if (c < Cin) atomicAdd(&grad_x[(size_t)n * Cin + c], e[j]);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// f32 MFMA GEMM with generic strides:  C[i,j] (+)= rscale[i] * sum_k A(i,k) * B(k,j)
//   A(i,k) at A[i*sai + k*sak], B(k,j) at B[k*sbk + j*sbj], C row-major [M,N].
// One wave owns a 16 x 64 tile; blockIdx.z splits the reduction (split > 1 => atomicAdd into zeroed C).
// Lane l feeds A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; D: col = l&15, row = 4*(l>>4) + r.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemm_f32_mfma_kernel(const float* __restrict__ A, long sai, long sak,
                                                            const float* __restrict__ B, long sbk, long sbj,
                                                            float* __restrict__ C, int M, int N, int Kd,
                                                            const float* __restrict__ rscale, int rscale_inv,
                                                            int kchunk, int atomic) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m0 = (blockIdx.x * 4 + wave) * 16;
  const int n0 = blockIdx.y * 64;
  if (m0 >= M) return;
  const int k_begin = blockIdx.z * kchunk;
  const int k_end = min(Kd, k_begin + kchunk);
  const int li = lane & 15, lk = lane >> 4;
  const int ai = m0 + li;
  const bool a_ok = ai < M;
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int k = k_begin; k < k_end; k += 4) {
    const int kk = k + lk;
    const bool k_ok = kk < k_end;
    const float a = (a_ok && k_ok) ? A[(long)ai * sai + (long)kk * sak] : 0.0f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int bj = n0 + 16 * t + li;
      const float b = (k_ok && bj < N) ? B[(long)kk * sbk + (long)bj * sbj] : 0.0f;
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int cj = n0 + 16 * t + li;
    if (cj >= N) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ci = m0 + 4 * lk + r;
      if (ci >= M) continue;
      float v = acc[t][r];
      if (rscale) v = rscale_inv ? v / rscale[ci] : v * rscale[ci];
      if (atomic) atomicAdd(&C[(size_t)ci * N + cj], v);
      else C[(size_t)ci * N + cj] = v;
    }
  }
}

// out[i,:] = in[i,:] / nn[i]
__global__ void row_div_kernel(const float* __restrict__ in, const float* __restrict__ nn, int M, int N,
                               float* __restrict__ out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)M * N) return;
  out[t] = in[t] / nn[t / N];
}

int launch_gemm(const float* A, long sai, long sak, const float* B, long sbk, long sbj, float* C, int M, int N, int Kd,
                const float* rscale, int rscale_inv, int split, hipStream_t stream) {
  if (M <= 0 || N <= 0) return D3F_OK;
  if (split < 1) split = 1;
  int kchunk = (cdiv(Kd, split) + 3) / 4 * 4;
  if (kchunk < 4) kchunk = 4;
  split = cdiv(Kd, kchunk);
  if (split < 1) split = 1;
  if (split > 1) {
    if (d3f::zero_async(C, sizeof(float) * (size_t)M * N, stream) != hipSuccess) return D3F_ELAUNCH;
  }
  dim3 grid(cdiv(M, 64), cdiv(N, 64), split);
  gemm_f32_mfma_kernel<<<grid, 256, 0, stream>>>(A, sai, sak, B, sbk, sbj, C, M, N, Kd, rscale, rscale_inv, kchunk,
                                                 split > 1);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

template <bool NN>
int launch_wf(const float* q_pts, const float* s_pts, const int32_t* idx, const float* x, const float* kp, int Nq,
              int Ns, int H, int Cin, int K, float extent, float* wf, float* nn, hipStream_t stream, int mode = 0) {
  const int grid = cdiv(Nq, 4);
  const int cpl = cdiv(Cin, 64);
  const float gd = gauss_denominator(extent);
#define D3F_WF(CPL)                                                                                                  \
  kpconv_wf_kernel<CPL, NN><<<grid, 256, 0, stream>>>(q_pts, s_pts, idx, x, kp, Nq, Ns, H, Cin, K, extent, wf, nn, gd, \
                                                      mode)
  if (cpl <= 1) D3F_WF(1);
  else if (cpl <= 2) D3F_WF(2);
  else if (cpl <= 4) D3F_WF(4);
  else if (cpl <= 8) D3F_WF(8);
  else return D3F_EINVAL;
#undef D3F_WF
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

// kpconv_fused.hip
bool kpconv_fused_supported(int Cin, int Cout, int K, int H, int Ns);
size_t kpconv_fused_ws_bytes(int Ns);
int kpconv_forward_fused(const float* q_pts, int Nq, const float* s_pts, int Ns, const int32_t* idx, int H,
                         const float* x, int Cin, const float* kp, int K, const float* W, int Cout, float extent,
                         float* out, float* nn_out, float* wf_save, void* spack_keep, float* grad_x_clear, void* ws,
                         hipStream_t stream);
size_t atb_ws_bytes(int R, int M, int N);
int atb_splitk(const float* A, const float* B, const float* row_div, int R, int M, int N, float* C, void* ws,
               hipStream_t stream, int M_out = 0);
int kpconv_grad_input_from_gw(const float* q_pts, int Nq, const float* s_pts, int Ns, const int32_t* idx, int H,
                              const float* x, int Cin, const float* kp, int K, float extent, const float* gwf,
                              const void* spack_kept, int gx_precleared, float* gx, void* ws, hipStream_t stream);
int kpconv_aggregate(const float* q_pts, int Nq, const float* s_pts, int Ns, const int32_t* idx, int H, const float* x,
                     int Cin, const float* kp, int K, float extent, float* wf_out, float* nn_out, void* spack_keep,
                     float* grad_x_clear, void* ws, hipStream_t stream);
int kpconv_aggregate_direct(const float* q_pts, int Nq, const float* s_pts, int Ns, const int32_t* idx, int H,
                            const float* x, int Cin, const float* kp, int K, float extent, float* wf_out, float* nn_out,
                            void* spack_keep, float* grad_x_clear, void* ws, hipStream_t stream);   // kpconv_aggregate.hip
// kpconv_small.hip
bool kpconv_small_supported(int Cin, int Cout, int K, int H);
int kpconv_small_dispatch(bool fwd, const float* q_pts, const float* s_pts, const int32_t* idx, const float* x,
                          const float* kp, const float* W, const float* nn_in, const float* gout, int Nq, int Ns, int H,
                          int Cin, int Cout, int K, float extent, float* out, float* nn_out, float* gW,
                          hipStream_t stream, float* wf_save = nullptr);
int kpconv_backward_fused(const float* q_pts, int Nq, const float* s_pts, int Ns, const int32_t* idx, int H,
                          const float* x, int Cin, const float* kp, int K, const float* W, int Cout, float extent,
                          const float* nn, const float* gout, const float* wf_saved, const void* spack_kept,
                          int gx_precleared, float* gx, float* gw, void* ws, hipStream_t stream);

}  // namespace d3f

using namespace d3f;

namespace d3f {
void kpconv_set_debug_flags(int f);
void set_phase_clock(unsigned long long* p);
int kpconv_timing_begin(int which, int max_launches);
int kpconv_timing_end(float* ms_out, int* shapes_out, int cap);
}

extern "C" {

// profiling aid (not part of the operator contract): run-time ablation switches of the fused forward kernel
void d3f_debug_set_flags(int flags) { d3f::kpconv_set_debug_flags(flags); }
void d3f_debug_set_phase_clock(void* counters) { d3f::set_phase_clock((unsigned long long*)counters); }

// measurement aid: HIP events on the launch stream around every launch of one kernel (which = 1: fused KPConv
// forward kernel, 2: KPConv grad-input kernel) between begin and end; end (after a device synchronisation by the
// caller) returns the launch count and fills per-launch milliseconds + {Nq, Ns, H, Cin, Cout, K}
int d3f_debug_kernel_timing_begin(int which, int max_launches) { return d3f::kpconv_timing_begin(which, max_launches); }
int d3f_debug_kernel_timing_end(float* ms_out, int32_t* shapes_out, int cap) {
  return d3f::kpconv_timing_end(ms_out, shapes_out, cap);
}

size_t d3f_kpconv_ws_bytes(int Nq, int Ns, int H, int K, int Cin, int Cout) {
  (void)Ns; (void)H;
  const size_t n = (size_t)(Nq > 0 ? Nq : 1), kc = (size_t)K * Cin;
  const size_t wf = align_up(sizeof(float) * n * kc, 256);                                  // wf
  const size_t gw = align_up(sizeof(float) * n * (kc > (size_t)Cout ? kc : (size_t)Cout), 256);  // gW / scaled grad
  const size_t generic = wf + gw + 256;
  const size_t fused = kpconv_fused_ws_bytes(Ns) + atb_ws_bytes(Nq, K * Cin, Cout) + 256;
  const size_t small = atb_ws_bytes(Nq, 16 * Cin, Cout) + 256;   // input-layer path: dW from the saved 16-slot wf
  const size_t m = generic > fused ? generic : fused;
  return m > small ? m : small;
}

static int kp_args_ok(const void* q_pts, int Nq, const void* s_pts, int Ns, const void* idx, int H, const void* x,
                      int Cin, const void* kp, int K, const void* w, int Cout, float extent) {
  return q_pts && s_pts && idx && x && kp && w && Nq >= 0 && Ns >= 0 && H >= 1 && Cin >= 1 && Cin <= 512 && K >= 1 &&
         K <= 16 && Cout >= 1 && extent > 0.0f;
}

// 1 when forward/backward for these shapes work on packed supports (spack_keep / spack_kept are honoured)
int d3f_kpconv_packs_supports(int Cin, int Cout, int K, int H, int Ns) {
  return (!kpconv_small_supported(Cin, Cout, K, H) && kpconv_fused_supported(Cin, Cout, K, H, Ns)) ? 1 : 0;
}

// floats per query the forward leaves in wf_save (0: these shapes save nothing): K*Cin, except the tiny-Cin input-layer
// kernels, whose rows are padded to 16 kernel-point slots (the A^T B kernel wants multiples of 16)
int d3f_kpconv_saves_wf(int Cin, int Cout, int K, int H) {
  if (!kpconv_small_supported(Cin, Cout, K, H)) return K * Cin;
  return (Cout % 16 == 0) ? 16 * Cin : 0;
}

int d3f_kpconv_forward(const float* q_pts, int Nq, const float* s_pts, int Ns, const int32_t* idx, int H,
                       const float* x, int Cin, const float* kernel_points, int K, const float* weights, int Cout,
                       float extent, float* out, float* nn_out, float* wf_save, void* spack_keep, float* grad_x_clear,
                       void* ws, size_t ws_bytes, void* stream_) {
  if (!kp_args_ok(q_pts, Nq, s_pts, Ns, idx, H, x, Cin, kernel_points, K, weights, Cout, extent) || !out || !nn_out ||
      !ws)
    return D3F_EINVAL;
  if (ws_bytes < d3f_kpconv_ws_bytes(Nq, Ns, H, K, Cin, Cout)) return D3F_EWORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const bool packs = Nq > 0 && Ns > 0 && !kpconv_small_supported(Cin, Cout, K, H) && kpconv_fused_supported(Cin, Cout, K, H, Ns);
  if (grad_x_clear == D3F_SPACK_READY && !packs) return D3F_EINVAL;   // nothing to be "ready" on the other paths
  if (grad_x_clear && !packs &&  // only the fused path clears it while packing
      d3f::zero_async(grad_x_clear, sizeof(float) * (size_t)Ns * Cin, stream) != hipSuccess)
    return D3F_ELAUNCH;
  if (Nq == 0) return D3F_OK;
  if (kpconv_small_supported(Cin, Cout, K, H))
    return kpconv_small_dispatch(true, q_pts, s_pts, idx, x, kernel_points, weights, nullptr, nullptr, Nq, Ns, H, Cin,
                                 Cout, K, extent, out, nn_out, nullptr, stream,
                                 d3f_kpconv_saves_wf(Cin, Cout, K, H) ? wf_save : nullptr);
  if (kpconv_fused_supported(Cin, Cout, K, H, Ns))
    return kpconv_forward_fused(q_pts, Nq, s_pts, Ns, idx, H, x, Cin, kernel_points, K, weights, Cout, extent, out,
                                nn_out, wf_save, spack_keep, grad_x_clear, ws, stream);
  float* wf = wf_save ? wf_save : (float*)ws;
  int rc = launch_wf<true>(q_pts, s_pts, idx, x, kernel_points, Nq, Ns, H, Cin, K, extent, wf, nn_out, stream);
  if (rc) return rc;
  // out = (wf [Nq, K*Cin] @ W [K*Cin, Cout]) / nn
  return launch_gemm(wf, (long)K * Cin, 1, weights, Cout, 1, out, Nq, Cout, K * Cin, nn_out, 1, 1, stream);
}

int d3f_kpconv_backward(const float* q_pts, int Nq, const float* s_pts, int Ns, const int32_t* idx, int H,
                        const float* x, int Cin, const float* kernel_points, int K, const float* weights, int Cout,
                        float extent, const float* nn, const float* grad_out, const float* wf_saved,
                        const void* spack_kept, int grad_x_precleared, float* grad_x, float* grad_w, void* ws,
                        size_t ws_bytes, void* stream_) {
  if (!kp_args_ok(q_pts, Nq, s_pts, Ns, idx, H, x, Cin, kernel_points, K, weights, Cout, extent) || !nn || !grad_out ||
      !ws || (!grad_x && !grad_w))
    return D3F_EINVAL;
  if (ws_bytes < d3f_kpconv_ws_bytes(Nq, Ns, H, K, Cin, Cout)) return D3F_EWORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const int KC = K * Cin;
  const size_t wf_elems = align_up(sizeof(float) * (size_t)(Nq > 0 ? Nq : 1) * KC, 256) / sizeof(float);
  float* wf = (float*)ws;
  float* gW = wf + wf_elems;
  const bool fused = Nq > 0 && Ns > 0 && kpconv_fused_supported(Cin, Cout, K, H, Ns);  // clears grad_x while packing
  if (grad_x && !fused && !grad_x_precleared &&
      d3f::zero_async(grad_x, sizeof(float) * (size_t)Ns * Cin, stream) != hipSuccess)
    return D3F_ELAUNCH;
  if (Nq == 0) {
    if (grad_w && d3f::zero_async(grad_w, sizeof(float) * (size_t)KC * Cout, stream) != hipSuccess)
      return D3F_ELAUNCH;
    return D3F_OK;
  }
  if (kpconv_fused_supported(Cin, Cout, K, H, Ns))
    return kpconv_backward_fused(q_pts, Nq, s_pts, Ns, idx, H, x, Cin, kernel_points, K, weights, Cout, extent, nn,
                                 grad_out, wf_saved, spack_kept, grad_x_precleared, grad_x, grad_w, ws, stream);
  int rc;
  if (grad_w && kpconv_small_supported(Cin, Cout, K, H) && wf_saved && d3f_kpconv_saves_wf(Cin, Cout, K, H)) {
    // grad_W = wf^T (g/nn) over the saved 16-slot rows; only the K live slots are written (deterministic, no atomics)
    rc = atb_splitk(wf_saved, grad_out, nn, Nq, 16 * Cin, Cout, grad_w, ws, stream, KC);
    if (rc) return rc;
    grad_w = nullptr;
  }
  if (grad_w && kpconv_small_supported(Cin, Cout, K, H)) {
    if (d3f::zero_async(grad_w, sizeof(float) * (size_t)KC * Cout, stream) != hipSuccess) return D3F_ELAUNCH;
    rc = kpconv_small_dispatch(false, q_pts, s_pts, idx, x, kernel_points, weights, nn, grad_out, Nq, Ns, H, Cin, Cout,
                               K, extent, nullptr, nullptr, grad_w, stream);
    if (rc) return rc;
    grad_w = nullptr;  // done; grad_x (rarely needed for the input layer) continues on the general path
  }
  if (grad_w) {
    // grad_W [KC, Cout] = wf^T [KC, Nq] @ (grad_out / nn) [Nq, Cout]
    if (wf_saved) {
      wf = const_cast<float*>(wf_saved);
    } else {
      rc = launch_wf<false>(q_pts, s_pts, idx, x, kernel_points, Nq, Ns, H, Cin, K, extent, wf, nullptr, stream);
      if (rc) return rc;
    }
    float* gbuf = gW;  // second scratch doubles as the scaled gradient [Nq, Cout] (sized max(KC, Cout) per row)
    row_div_kernel<<<cdiv((long long)Nq * Cout, 256), 256, 0, stream>>>(grad_out, nn, Nq, Cout, gbuf);
    D3F_LAUNCH_CHECK();
    const int split = cdiv(Nq, 256);
    rc = launch_gemm(wf, 1, KC, gbuf, Cout, 1, grad_w, KC, Cout, Nq, nullptr, 0, split, stream);
    if (rc) return rc;
  }
  if (grad_x) {
    // gW [Nq, KC] = (grad_out / nn) [Nq, Cout] @ W^T [Cout, KC]
    rc = launch_gemm(grad_out, Cout, 1, weights, 1, Cout, gW, Nq, KC, Cout, nn, 1, 1, stream);
    if (rc) return rc;
    const int grid = cdiv(Nq, 4);
    const int cpl = cdiv(Cin, 64);
#define D3F_DX(CPL) \
  kpconv_dx_kernel<CPL><<<grid, 256, 0, stream>>>(q_pts, s_pts, idx, kernel_points, Nq, Ns, H, Cin, K, extent, gW, grad_x)
    if (cpl <= 1) D3F_DX(1);
    else if (cpl <= 2) D3F_DX(2);
    else if (cpl <= 4) D3F_DX(4);
    else D3F_DX(8);
#undef D3F_DX
    D3F_LAUNCH_CHECK();
  }
  return D3F_OK;
}

// wf [Nq, K*Cin] and nn [Nq] only (the caller contracts wf with W by a GEMM and divides by nn); shapes as
// d3f_kpconv_grad_input_supported
int d3f_kpconv_aggregate(const float* q_pts, int Nq, const float* s_pts, int Ns, const int32_t* idx, int H,
                         const float* x, int Cin, const float* kernel_points, int K, float extent, float* wf_out,
                         float* nn_out, void* spack_keep, float* grad_x_clear, void* ws, size_t ws_bytes,
                         void* stream_) {
  if (!q_pts || !s_pts || !idx || !x || !kernel_points || !wf_out || !nn_out || !ws || Nq < 1 || Ns < 1 || H < 1 ||
      !(extent > 0.0f) || !kpconv_fused_supported(Cin, 64, K, H, Ns))
    return D3F_EINVAL;
  if (ws_bytes < d3f_kpconv_ws_bytes(Nq, Ns, H, K, Cin, 64)) return D3F_EWORKSPACE;
  // registers -> HBM (kpconv_aggregate.hip); tunables().agg_through_lds keeps the round-1 form (phase A of the fused
  // kernel, through its LDS tile) for A/B measurements
  const bool through_lds = d3f::tunables().agg_through_lds != 0;
  if (!through_lds)
    return kpconv_aggregate_direct(q_pts, Nq, s_pts, Ns, idx, H, x, Cin, kernel_points, K, extent, wf_out, nn_out,
                                   spack_keep, grad_x_clear, ws, (hipStream_t)stream_);
  return kpconv_aggregate(q_pts, Nq, s_pts, Ns, idx, H, x, Cin, kernel_points, K, extent, wf_out, nn_out, spack_keep,
                          grad_x_clear, ws, (hipStream_t)stream_);
}

// grad_x [Ns, Cin] (OVERWRITTEN) from gwf = (grad_out / nn) @ W^T  [Nq, K*Cin] computed by the caller (a plain GEMM)
int d3f_kpconv_grad_input_supported(int Cin, int K, int H, int Ns) {
  return kpconv_fused_supported(Cin, 64, K, H, Ns) ? 1 : 0;
}

int d3f_kpconv_grad_input(const float* q_pts, int Nq, const float* s_pts, int Ns, const int32_t* idx, int H,
                          const float* x, int Cin, const float* kernel_points, int K, float extent, const float* gwf,
                          const void* spack_kept, int grad_x_precleared, float* grad_x, void* ws, size_t ws_bytes,
                          void* stream_) {
  if (!q_pts || !s_pts || !idx || !x || !kernel_points || !gwf || !grad_x || !ws || Nq < 0 || Ns < 0 || H < 1 ||
      !(extent > 0.0f) || !kpconv_fused_supported(Cin, 64, K, H, Ns))
    return D3F_EINVAL;
  if (ws_bytes < d3f_kpconv_ws_bytes(Nq, Ns, H, K, Cin, 64)) return D3F_EWORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  if (Nq == 0 || Ns == 0) {
    if (!grad_x_precleared && d3f::zero_async(grad_x, sizeof(float) * (size_t)Ns * Cin, stream) != hipSuccess)
      return D3F_ELAUNCH;
    return D3F_OK;
  }
  return kpconv_grad_input_from_gw(q_pts, Nq, s_pts, Ns, idx, H, x, Cin, kernel_points, K, extent, gwf, spack_kept,
                                   grad_x_precleared, grad_x, ws, stream);
}

// ---- non-default influence / aggregation modes (blocks.py:327-352): general path only ---------------------------
// mode = influence (0 'linear', 1 'constant', 2 'gaussian') | 4 when aggregation_mode == 'closest'.
static bool mode_ok(int mode) { return mode >= 0 && mode < 8 && (mode & 3) != 3; }

// wf [Nq, K*Cin] = sum_h w_mode[n,h,k] x[idx[n,h], :],  nn [Nq] = max(1, #neighbors with a positive feature sum)
int d3f_kpconv_aggregate_modes(const float* q_pts, int Nq, const float* s_pts, int Ns, const int32_t* idx, int H,
                               const float* x, int Cin, const float* kernel_points, int K, float extent, int mode,
                               float* wf_out, float* nn_out, void* stream_) {
  if (!q_pts || !s_pts || !idx || !x || !kernel_points || !wf_out || !nn_out || Nq < 0 || Ns < 1 || H < 1 ||
      Cin < 1 || Cin > 512 || K < 1 || K > 16 || !(extent > 0.0f) || !mode_ok(mode))
    return D3F_EINVAL;
  if (Nq == 0) return D3F_OK;
  return launch_wf<true>(q_pts, s_pts, idx, x, kernel_points, Nq, Ns, H, Cin, K, extent, wf_out, nn_out,
                         (hipStream_t)stream_, mode);
}

// grad_x [Ns, Cin] (OVERWRITTEN) = scatter_h sum_k w_mode[n,h,k] gwf[n,k,:]   with gwf = (grad_out/nn) @ W^T
int d3f_kpconv_grad_input_modes(const float* q_pts, int Nq, const float* s_pts, int Ns, const int32_t* idx, int H,
                                int Cin, const float* kernel_points, int K, float extent, int mode, const float* gwf,
                                float* grad_x, void* stream_) {
  if (!q_pts || !s_pts || !idx || !kernel_points || !gwf || !grad_x || Nq < 0 || Ns < 1 || H < 1 || Cin < 1 ||
      Cin > 512 || K < 1 || K > 16 || !(extent > 0.0f) || !mode_ok(mode))
    return D3F_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  if (d3f::zero_async(grad_x, sizeof(float) * (size_t)Ns * Cin, stream) != hipSuccess) return D3F_ELAUNCH;
  if (Nq == 0) return D3F_OK;
  const int grid = cdiv(Nq, 4);
  const int cpl = cdiv(Cin, 64);
  const float gd = gauss_denominator(extent);
#define D3F_DX(CPL)                                                                                                 \
  kpconv_dx_kernel<CPL><<<grid, 256, 0, stream>>>(q_pts, s_pts, idx, kernel_points, Nq, Ns, H, Cin, K, extent, gwf, \
                                                  grad_x, gd, mode)
  if (cpl <= 1) D3F_DX(1);
  else if (cpl <= 2) D3F_DX(2);
  else if (cpl <= 4) D3F_DX(4);
  else D3F_DX(8);
#undef D3F_DX
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

}  // extern "C"
