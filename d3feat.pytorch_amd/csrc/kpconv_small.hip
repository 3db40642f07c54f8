// KPConv for very small input widths (Cin <= 4: the first layer of the network has Cin = 1, blocks.py:557-571 with
// in_features_dim = 1) -- forward and grad-weights.  With 1..4 channels there is no matrix shape worth an MFMA:
// one wave serves one query at a time with the same (kernel point = lane & 15, neighbor sub-slot = lane >> 4) lane
// layout as the fused kernels, but on the VALU:
//   wf[k, c]  = sum_h w[q,h,k] * x[idx[q,h], c]      lane (k, j) accumulates neighbors h = 4g + j, then a 2-step
//                                                    shuffle reduction over j leaves wf[k, :] on the 16 lanes li = k
//   out[q, o] = ( sum_k sum_c wf[k,c] * W[k,c,o] ) / nn[q]        lane <-> output channel o (Cout <= 128),
//                                                    W (K*Cin*Cout floats, <= 30 KB) lives in registers per lane
// Supports and index rows are fetched once per query (coalesced row load + one gather, then shuffles).
// grad-weights: dW[k,c,o] = sum_q wf[q,k,c] * g[q,o]/nn[q] accumulated in registers over the wave's queries and
// flushed with one atomic per element per wave.
#include "kpconv_tile.hpp"

namespace d3f {

template <int CIN, int OPL>  // OPL = output channels per lane (Cout <= 64*OPL)
struct SmallAgg {
  // returns wf[k = li][c] (valid on every lane, replicated over lg) and nn.  `rec` = this wave's 64-entry LDS record
  // (float4 {s.x, s.y, s.z, x[.,0]} per neighbor, + CIN - 1 further feature words in `recx`): every lane publishes the
  // neighbor it fetched, then lane (k, j) reads neighbor 4 g + j of every group as ONE 16-byte LDS read (4 distinct
  // addresses per wave: a broadcast) -- the 5 ds_bpermute per group this replaces kept the LDS pipe 0.86 busy
  // (profiles/r03_pmc_kpconv.txt).  A shadow neighbor reads x = 0 (bounds-checked buffer), so it contributes nothing
  // whatever its weight.
  __device__ static __forceinline__ void run(const float* __restrict__ q_pts, const int32_t* __restrict__ idx, int q,
                                             int H, int Ns, __amdgpu_buffer_rsrc_t rs_s, __amdgpu_buffer_rsrc_t rs_x,
                                             float kx, float ky, float kz, float inv_extent, int lane,
                                             float4* __restrict__ rec, float* __restrict__ recx, float (&wf)[CIN],
                                             float& nn) {
    const int lg = lane >> 4;
    const int n_own = (int)min((unsigned)(lane < H ? idx[(size_t)q * H + lane] : Ns), (unsigned)Ns);
    // s_pts [Ns,3] and x [Ns,CIN] are read through bounds-checked buffers: the shadow index returns zeros
    const float sx = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_s, (unsigned)n_own * 12u + 0u, 0, 0));
    const float sy = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_s, (unsigned)n_own * 12u + 4u, 0, 0));
    const float sz = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_s, (unsigned)n_own * 12u + 8u, 0, 0));
    float xo[CIN];
    float rowsum = 0.0f;
#pragma unroll
    for (int c = 0; c < CIN; ++c) {
      xo[c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_x, ((unsigned)n_own * CIN + c) * 4u, 0, 0));
      rowsum += xo[c];
    }
    nn = fmaxf((float)__popcll(__ballot(rowsum > 0.0f)), 1.0f);   // (#neighbors with a positive feature sum, blocks.py:377)
    rec[lane] = make_float4(sx, sy, sz, xo[0]);
#pragma unroll
    for (int c = 1; c < CIN; ++c) recx[(c - 1) * 64 + lane] = xo[c];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const float cx = q_pts[3 * (size_t)q + 0] + kx, cy = q_pts[3 * (size_t)q + 1] + ky, cz = q_pts[3 * (size_t)q + 2] + kz;
#pragma unroll
    for (int c = 0; c < CIN; ++c) wf[c] = 0.0f;
    const int ng = (H + 3) >> 2;
    for (int g = 0; g < ng; ++g) {
      const int src = 4 * g + lg;  // < 64
      const float4 sp = rec[src];
      const float w = kp_influence(sp, cx, cy, cz, inv_extent);
      wf[0] = fmaf(w, sp.w, wf[0]);
#pragma unroll
      for (int c = 1; c < CIN; ++c) wf[c] = fmaf(w, recx[(c - 1) * 64 + src], wf[c]);
    }
    __builtin_amdgcn_wave_barrier();   // (the record is rewritten for the wave's next query)
#pragma unroll
    for (int c = 0; c < CIN; ++c) {
      wf[c] += __shfl_xor(wf[c], 16, 64);
      wf[c] += __shfl_xor(wf[c], 32, 64);
    }
  }
};

template <int CIN, int OPL>
__global__ __launch_bounds__(256) void kpconv_small_fwd_kernel(const float* __restrict__ q_pts,
                                                               const float* __restrict__ s_pts,
                                                               const int32_t* __restrict__ idx,
                                                               const float* __restrict__ x,
                                                               const float* __restrict__ kp,
                                                               const float* __restrict__ W, int Nq, int Ns, int H,
                                                               int Cout, int K, float extent, float* __restrict__ out,
                                                               float* __restrict__ nn_out,
                                                               float* __restrict__ wf_save) {
  const int lane = threadIdx.x & 63;
  const int li = lane & 15;
  const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
  const bool klive = li < K;
  const float kx = klive ? kp[3 * li + 0] : kFarKernelPoint, ky = klive ? kp[3 * li + 1] : kFarKernelPoint,
              kz = klive ? kp[3 * li + 2] : kFarKernelPoint;
  const __amdgpu_buffer_rsrc_t rs_s = make_rsrc(s_pts, (unsigned)Ns * 12u);
  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(x, (unsigned)Ns * CIN * 4u);
  const float inv_extent = 1.0f / extent;
  __shared__ __attribute__((aligned(16))) float4 rec_all[4][64];
  __shared__ float recx_all[4][CIN > 1 ? (CIN - 1) * 64 : 1];
  float4* rec = rec_all[threadIdx.x >> 6];
  float* recx = recx_all[threadIdx.x >> 6];
  float wreg[16][CIN][OPL];  // W[k][c][o = lane + 64*j]
#pragma unroll
  for (int k = 0; k < 16; ++k)
#pragma unroll
    for (int c = 0; c < CIN; ++c)
#pragma unroll
      for (int j = 0; j < OPL; ++j) {
        const int o = lane + 64 * j;
        wreg[k][c][j] = (k < K && o < Cout) ? W[((size_t)k * CIN + c) * Cout + o] : 0.0f;
      }
  for (int q = gw; q < Nq; q += nw) {
    float wf[CIN], nn;
    SmallAgg<CIN, OPL>::run(q_pts, idx, q, H, Ns, rs_s, rs_x, kx, ky, kz, inv_extent, lane, rec, recx, wf, nn);
    float acc[OPL];
#pragma unroll
    for (int j = 0; j < OPL; ++j) acc[j] = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k)
#pragma unroll
      for (int c = 0; c < CIN; ++c) {
        // wf[k][c] lives on lane k: a scalar broadcast (v_readlane), not a trip through the LDS crossbar
        const float v = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(wf[c]), k));
#pragma unroll
        for (int j = 0; j < OPL; ++j) acc[j] = fmaf(v, wreg[k][c][j], acc[j]);
      }
#pragma unroll
    for (int j = 0; j < OPL; ++j) {
      const int o = lane + 64 * j;
      if (o < Cout) out[(size_t)q * Cout + o] = acc[j] / nn;
    }
    if (lane == 0) nn_out[q] = nn;
    // training: the weighted features [16 kernel-point slots x CIN] of the query (slot k lives on lane k; slots >= K are
    // zero) stay behind, so the weight gradient is the reduction-parallel A^T B kernel instead of a second aggregation
    if (wf_save && lane < 16) {
#pragma unroll
      for (int c = 0; c < CIN; ++c) wf_save[((size_t)q * 16 + lane) * CIN + c] = klive ? wf[c] : 0.0f;
    }
  }
}

template <int CIN, int OPL>
__global__ __launch_bounds__(256) void kpconv_small_dw_kernel(const float* __restrict__ q_pts,
                                                              const float* __restrict__ s_pts,
                                                              const int32_t* __restrict__ idx,
                                                              const float* __restrict__ x,
                                                              const float* __restrict__ kp,
                                                              const float* __restrict__ nn_in,
                                                              const float* __restrict__ gout, int Nq, int Ns, int H,
                                                              int Cout, int K, float extent, float* __restrict__ gW) {
  const int lane = threadIdx.x & 63;
  const int li = lane & 15;
  const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
  const bool klive = li < K;
  const float kx = klive ? kp[3 * li + 0] : kFarKernelPoint, ky = klive ? kp[3 * li + 1] : kFarKernelPoint,
              kz = klive ? kp[3 * li + 2] : kFarKernelPoint;
  const __amdgpu_buffer_rsrc_t rs_s = make_rsrc(s_pts, (unsigned)Ns * 12u);
  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(x, (unsigned)Ns * CIN * 4u);
  const float inv_extent = 1.0f / extent;
  __shared__ __attribute__((aligned(16))) float4 rec_all[4][64];
  __shared__ float recx_all[4][CIN > 1 ? (CIN - 1) * 64 : 1];
  float4* rec = rec_all[threadIdx.x >> 6];
  float* recx = recx_all[threadIdx.x >> 6];
  float dw[16][CIN][OPL];
#pragma unroll
  for (int k = 0; k < 16; ++k)
#pragma unroll
    for (int c = 0; c < CIN; ++c)
#pragma unroll
      for (int j = 0; j < OPL; ++j) dw[k][c][j] = 0.0f;
  for (int q = gw; q < Nq; q += nw) {
    float wf[CIN], nn;
    SmallAgg<CIN, OPL>::run(q_pts, idx, q, H, Ns, rs_s, rs_x, kx, ky, kz, inv_extent, lane, rec, recx, wf, nn);
    float g[OPL];
    const float inv_nn = 1.0f / nn_in[q];
#pragma unroll
    for (int j = 0; j < OPL; ++j) {
      const int o = lane + 64 * j;
      g[j] = o < Cout ? gout[(size_t)q * Cout + o] * inv_nn : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k)
#pragma unroll
      for (int c = 0; c < CIN; ++c) {
        const float v = __shfl(wf[c], k, 64);
#pragma unroll
        for (int j = 0; j < OPL; ++j) dw[k][c][j] = fmaf(v, g[j], dw[k][c][j]);
      }
  }
  // combine the 4 waves of the workgroup in LDS, then one global atomic per weight per workgroup
  __shared__ float red[16 * CIN * OPL * 64];
  for (int i = threadIdx.x; i < 16 * CIN * OPL * 64; i += blockDim.x) red[i] = 0.0f;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 16; ++k)
#pragma unroll
    for (int c = 0; c < CIN; ++c)
#pragma unroll
      for (int j = 0; j < OPL; ++j) atomicAdd(&red[((k * CIN + c) * OPL + j) * 64 + lane], dw[k][c][j]);
  __syncthreads();
  for (int i = threadIdx.x; i < 16 * CIN * OPL * 64; i += blockDim.x) {
    const int l = i & 63, j = (i >> 6) % OPL, kc = (i >> 6) / OPL;
    const int k = kc / CIN, c = kc % CIN, o = l + 64 * j;
    if (k < K && o < Cout) atomicAdd(&gW[((size_t)k * CIN + c) * Cout + o], red[i]);
  }
}

bool kpconv_small_supported(int Cin, int Cout, int K, int H) {
  return Cin >= 1 && Cin <= 4 && Cout >= 1 && Cout <= 128 && K >= 1 && K <= 16 && H >= 1 && H <= 64 &&
         (Cin * ((Cout + 63) / 64) <= 4);  // register budget: 16*Cin*OPL weights per lane
}

template <int CIN, int OPL>
static int launch_small(bool fwd, const float* q_pts, const float* s_pts, const int32_t* idx, const float* x,
                        const float* kp, const float* W, const float* nn_in, const float* gout, int Nq, int Ns, int H,
                        int Cout, int K, float extent, float* out, float* nn_out, float* gW, hipStream_t stream,
                        float* wf_save) {
  // persistent waves: 4 per workgroup, ~8 workgroups per CU so the per-wave weight registers are loaded once per ~5 queries
  int blocks = cdiv(Nq, 4 * 4);
  if (blocks > 2048) blocks = 2048;
  if (!fwd && blocks > 512) blocks = 512;  // dW: every workgroup ends with K*Cin*Cout global atomics
  if (blocks < 1) blocks = 1;
  if (fwd)
    kpconv_small_fwd_kernel<CIN, OPL><<<blocks, 256, 0, stream>>>(q_pts, s_pts, idx, x, kp, W, Nq, Ns, H, Cout, K, extent,
                                                                  out, nn_out, wf_save);
  else
    kpconv_small_dw_kernel<CIN, OPL><<<blocks, 256, 0, stream>>>(q_pts, s_pts, idx, x, kp, nn_in, gout, Nq, Ns, H, Cout, K,
                                                                 extent, gW);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

int kpconv_small_dispatch(bool fwd, const float* q_pts, const float* s_pts, const int32_t* idx, const float* x,
                          const float* kp, const float* W, const float* nn_in, const float* gout, int Nq, int Ns, int H,
                          int Cin, int Cout, int K, float extent, float* out, float* nn_out, float* gW,
                          hipStream_t stream, float* wf_save) {
  const int opl = (Cout + 63) / 64;
#define D3F_S(C, O) \
  return launch_small<C, O>(fwd, q_pts, s_pts, idx, x, kp, W, nn_in, gout, Nq, Ns, H, Cout, K, extent, out, nn_out, gW, stream, \
                            wf_save)
  if (Cin == 1 && opl == 1) D3F_S(1, 1);
  if (Cin == 1 && opl == 2) D3F_S(1, 2);
  if (Cin == 2 && opl == 1) D3F_S(2, 1);
  if (Cin == 2 && opl == 2) D3F_S(2, 2);
  if (Cin == 3 && opl == 1) D3F_S(3, 1);
  if (Cin == 4 && opl == 1) D3F_S(4, 1);
#undef D3F_S
  return D3F_EINVAL;
}

}  // namespace d3f
