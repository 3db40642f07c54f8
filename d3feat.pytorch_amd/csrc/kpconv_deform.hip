// Deformable KPConv: aggregation with PER-QUERY kernel points, and its gradients.
//
// Replaces the deformable=True branch of the reference's KPConv.forward (models/blocks.py:243-257,286-324) and the
// autograd graph behind it.  The kernel-point offsets come from a rigid KPConv (the `offset_conv`, :190-199,244) which
// runs on the ordinary kernels; this file takes the deformed kernel points kp_def [Nq,K,3] = offsets + kernel_points
// (:287) and computes
//   d2[n,h,k]   = |(s[idx[n,h]] - q[n]) - kp_def[n,k]|^2                                   (:293-297)
//   min_d2[n,k] = min_h d2[n,h,k]  (+ the support index that attains it)                   (:301, the fitting loss input)
//   live[n,h]   = idx[n,h] is a real support  AND  any_k d2[n,h,k] < extent^2               (:304-321: out-of-range
//                 neighbors are replaced by the shadow index, so they add neither weight nor neighbor count)
//   wf[n,k,c]   = sum_{h live} w_mode(d2[n,h,k]) x[idx[n,h],c]                              (:327-362)
//   nn[n]       = max(1, #{h live : sum_c x[idx[n,h],c] > 0})                               (:376-379)
// The caller applies the modulations (:365-366), contracts wf with the kernel weights and divides by nn (plain tensor
// ops / GEMMs).  Backward, for gwf = d loss / d wf:
//   grad_x[s,c]      += sum_{(n,h) live, idx = s} sum_k w[n,h,k] gwf[n,k,c]
//   grad_kp[n,k,:]    = sum_{h live} dw/dkp (n,h,k) * <gwf[n,k,:], x[idx[n,h],:]>
//       'linear'   w = max(0, 1 - d/extent):  dw/dkp = diff / (d extent)  where 0 < d < extent   (diff = (s-q) - kp)
//       'gaussian' w = exp(-d2/g):            dw/dkp = w * 2 diff / g
//       'constant' 0;  'closest' keeps only the selected kernel point's term (the one-hot mask has no gradient).
// General path: one wave per query, lanes <-> channels, the K influence weights live in lanes 0..15 of every 16-lane
// group (like kpconv.hip).  D3Feat's configuration never enables it (config.py:45-46).
#include "common.hpp"
#include "kpconv_modes.hpp"

namespace d3f {

template <int CPL>
__global__ __launch_bounds__(256) void kpconv_deform_wf_kernel(
    const float* __restrict__ q_pts, const float* __restrict__ s_pts, const int32_t* __restrict__ idx,
    const float* __restrict__ x, const float* __restrict__ kp_def, int Nq, int Ns, int H, int Cin, int K, float extent,
    float extent_sq, float gauss_denom, int mode, float* __restrict__ wf, float* __restrict__ nn,
    float* __restrict__ min_d2, int32_t* __restrict__ min_idx) {
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= Nq) return;
  const float qx = q_pts[3 * (size_t)q + 0], qy = q_pts[3 * (size_t)q + 1], qz = q_pts[3 * (size_t)q + 2];
  const int kk = lane & 15;
  const bool klive = kk < K;
  const float* kq = kp_def + ((size_t)q * K + (klive ? kk : 0)) * 3;
  const float kx = klive ? kq[0] : 0.0f, ky = klive ? kq[1] : 0.0f, kz = klive ? kq[2] : 0.0f;
  float acc[16][CPL];
#pragma unroll
  for (int k = 0; k < 16; ++k)
#pragma unroll
    for (int j = 0; j < CPL; ++j) acc[k][j] = 0.0f;
  int cnt = 0;
  float best = __builtin_huge_valf();
  int bidx = Ns;
  const int32_t* row = idx + (size_t)q * H;
  for (int h = 0; h < H; ++h) {
    const int n = row[h];
    if (n < 0 || n >= Ns) continue;  // shadow neighbor (the 1e6 point): never nearest unless the row is all shadow
    const float rx = s_pts[3 * (size_t)n + 0] - qx, ry = s_pts[3 * (size_t)n + 1] - qy,
                rz = s_pts[3 * (size_t)n + 2] - qz;
    const float dx = rx - kx, dy = ry - ky, dz = rz - kz;
    const float d2 = dx * dx + dy * dy + dz * dz;
    if (klive && d2 < best) { best = d2; bidx = n; }
    if ((__ballot(klive && d2 < extent_sq) & 0xffffull) == 0ull) continue;  // out of every kernel point's range
    const float w = influence_weight(d2, klive, kk, extent, gauss_denom, mode);
    float xs[CPL];
    float rs = 0.0f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const int c = lane + 64 * j;
      xs[j] = c < Cin ? x[(size_t)n * Cin + c] : 0.0f;
      rs += xs[j];
    }
    rs = wave_sum(rs);
    cnt += rs > 0.0f;
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
  if (lane == 0) nn[q] = (float)(cnt > 1 ? cnt : 1);
  if (lane < 16 && klive) {
    if (bidx == Ns) {  // no real neighbor: the reference's minimum is the distance to its shadow point at 1e6 (:277)
      const float dx = (1e6f - qx) - kx, dy = (1e6f - qy) - ky, dz = (1e6f - qz) - kz;
      best = dx * dx + dy * dy + dz * dz;
    }
    if (min_d2) min_d2[(size_t)q * K + kk] = best;
    if (min_idx) min_idx[(size_t)q * K + kk] = bidx;
  }
}

template <int CPL>
__global__ __launch_bounds__(256) void kpconv_deform_grad_kernel(
    const float* __restrict__ q_pts, const float* __restrict__ s_pts, const int32_t* __restrict__ idx,
    const float* __restrict__ x, const float* __restrict__ kp_def, int Nq, int Ns, int H, int Cin, int K, float extent,
    float extent_sq, float gauss_denom, int mode, const float* __restrict__ gwf, float* __restrict__ grad_x,
    float* __restrict__ grad_kp) {
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= Nq) return;
  const float qx = q_pts[3 * (size_t)q + 0], qy = q_pts[3 * (size_t)q + 1], qz = q_pts[3 * (size_t)q + 2];
  const int kk = lane & 15;
  const bool klive = kk < K;
  const float* kq = kp_def + ((size_t)q * K + (klive ? kk : 0)) * 3;
  const float kx = klive ? kq[0] : 0.0f, ky = klive ? kq[1] : 0.0f, kz = klive ? kq[2] : 0.0f;
  float g[16][CPL];
  const float* gq = gwf + (size_t)q * K * Cin;
#pragma unroll
  for (int k = 0; k < 16; ++k)
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const int c = lane + 64 * j;
      g[k][j] = (k < K && c < Cin) ? gq[(size_t)k * Cin + c] : 0.0f;
    }
  float ax = 0.0f, ay = 0.0f, az = 0.0f;
  const int32_t* row = idx + (size_t)q * H;
  for (int h = 0; h < H; ++h) {
    const int n = row[h];
    if (n < 0 || n >= Ns) continue;
    const float rx = s_pts[3 * (size_t)n + 0] - qx, ry = s_pts[3 * (size_t)n + 1] - qy,
                rz = s_pts[3 * (size_t)n + 2] - qz;
    const float dx = rx - kx, dy = ry - ky, dz = rz - kz;
    const float d2 = dx * dx + dy * dy + dz * dz;
    if ((__ballot(klive && d2 < extent_sq) & 0xffffull) == 0ull) continue;
    const float w = influence_weight(d2, klive, kk, extent, gauss_denom, mode);
    if (grad_x) {
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
    if (grad_kp) {
      float xs[CPL];
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        const int c = lane + 64 * j;
        xs[j] = c < Cin ? x[(size_t)n * Cin + c] : 0.0f;
      }
      float t = 0.0f;  // lane kk keeps <gwf[n,kk,:], x[idx,:]>
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        float p = 0.0f;
#pragma unroll
        for (int j = 0; j < CPL; ++j) p = fmaf(g[k][j], xs[j], p);
        p = wave_sum(p);
        if (kk == k) t = p;
      }
      float coef = 0.0f;  // dw/dkp = coef * diff
      switch (mode & 3) {
        case 1: break;
        case 2: coef = w * (2.0f / gauss_denom); break;
        default: {
          const float d = sqrtf(d2);
          if (w > 0.0f && d > 0.0f) coef = 1.0f / (d * extent);
        }
      }
      coef *= t;
      ax = fmaf(coef, dx, ax);
      ay = fmaf(coef, dy, ay);
      az = fmaf(coef, dz, az);
    }
  }
  if (grad_kp && lane < 16 && klive) {
    float* o = grad_kp + ((size_t)q * K + kk) * 3;
    o[0] = ax; o[1] = ay; o[2] = az;
  }
}

}  // namespace d3f

using namespace d3f;

namespace {
bool args_ok(const float* q_pts, int Nq, const float* s_pts, int Ns, const int32_t* idx, int H, const float* x, int Cin,
             const float* kp_def, int K, float extent, float extent_sq, int mode) {
  return q_pts && s_pts && idx && x && kp_def && Nq >= 0 && Ns >= 1 && H >= 1 && Cin >= 1 && Cin <= 512 && K >= 1 &&
         K <= 16 && extent > 0.0f && extent_sq > 0.0f && kpconv_mode_ok(mode);
}
}  // namespace

extern "C" {

int d3f_kpconv_deform_aggregate(const float* q_pts, int Nq, const float* s_pts, int Ns, const int32_t* idx, int H,
                                const float* x, int Cin, const float* kp_def, int K, float extent, float extent_sq,
                                int mode, float* wf_out, float* nn_out, float* min_d2_out, int32_t* min_idx_out,
                                void* stream_) {
  if (!args_ok(q_pts, Nq, s_pts, Ns, idx, H, x, Cin, kp_def, K, extent, extent_sq, mode) || !wf_out || !nn_out)
    return D3F_EINVAL;
  if (Nq == 0) return D3F_OK;
  hipStream_t stream = (hipStream_t)stream_;
  const int grid = cdiv(Nq, 4), cpl = cdiv(Cin, 64);
  const float gd = gauss_denominator(extent);
#define D3F_DWF(CPL)                                                                                                   \
  kpconv_deform_wf_kernel<CPL><<<grid, 256, 0, stream>>>(q_pts, s_pts, idx, x, kp_def, Nq, Ns, H, Cin, K, extent,      \
                                                         extent_sq, gd, mode, wf_out, nn_out, min_d2_out, min_idx_out)
  if (cpl <= 1) D3F_DWF(1);
  else if (cpl <= 2) D3F_DWF(2);
  else if (cpl <= 4) D3F_DWF(4);
  else D3F_DWF(8);
#undef D3F_DWF
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

int d3f_kpconv_deform_grad(const float* q_pts, int Nq, const float* s_pts, int Ns, const int32_t* idx, int H,
                           const float* x, int Cin, const float* kp_def, int K, float extent, float extent_sq, int mode,
                           const float* gwf, float* grad_x, float* grad_kp, void* stream_) {
  if (!args_ok(q_pts, Nq, s_pts, Ns, idx, H, x, Cin, kp_def, K, extent, extent_sq, mode) || !gwf ||
      (!grad_x && !grad_kp))
    return D3F_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  if (grad_x && d3f::zero_async(grad_x, sizeof(float) * (size_t)Ns * Cin, stream) != hipSuccess) return D3F_ELAUNCH;
  if (Nq == 0) return D3F_OK;
  const int grid = cdiv(Nq, 4), cpl = cdiv(Cin, 64);
  const float gd = gauss_denominator(extent);
#define D3F_DGR(CPL)                                                                                                   \
  kpconv_deform_grad_kernel<CPL><<<grid, 256, 0, stream>>>(q_pts, s_pts, idx, x, kp_def, Nq, Ns, H, Cin, K, extent,    \
                                                           extent_sq, gd, mode, gwf, grad_x, grad_kp)
  if (cpl <= 1) D3F_DGR(1);
  else if (cpl <= 2) D3F_DGR(2);
  else if (cpl <= 4) D3F_DGR(4);
  else D3F_DGR(8);
#undef D3F_DGR
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

}  // extern "C"
