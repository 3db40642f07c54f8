// KPConv neighbor aggregation as a kernel of its own, straight from registers to HBM -- both directions:
//
//   forward     wf[q, k, c] = sum_h w(q, idx[q,h], k) x[idx[q,h], c]                 (reference models/blocks.py:359-375)
//   transposed  A [s, k, o] = sum_{q in rev(s)} w(q, s, k) g[q, o] (/ nn[q])         (autograd of the same lines, with the
//                                                                                     sums over q and (k, o) exchanged)
//
// after which BOTH contractions with the kernel weights are plain GEMMs over many rows:
//   out    = (wf [Nq, K Cin]  @ W  [K Cin, Cout]) / nn              grad_x = A [Ns, K Cout] @ W' [K Cout, Cin],  W'[k,o,c] = W[k,c,o]
//   grad_W =  wf^T @ (g / nn)
//
// Why not fused (kpconv_fused.hip / kpconv_dx_gather.hip keep the fused forms for the narrow layers): the fused kernels
// hold a [16 rows][K x 64 channels] tile in LDS between their two phases -- 66 KB, two workgroups = two waves per SIMD on a
// CU -- and contract it against W fragments streamed from L2 once per 16-row tile (245 KB of weights per 16 rows at
// 64 -> 64 channels).  From 64 channels up the contraction is 60 % and more of the arithmetic and ran at 0.15 - 0.2 of
// the f32 matrix rate there, while the aggregation phase sat on gather latency with nothing to overlap it
// (profiles/r03_kpconv_phase_clock.txt, profiles/r04_step_timeline_stack4.txt: 176 us for 8.2k rows x 128 channels).
// Split, the aggregation needs NO shared memory (a lane owns D[k = 4 lg + j][channel li CV + r] of its query: 16
// lanes write 64 CV contiguous bytes of one (query, kernel point) row), so occupancy is set by registers alone, and the
// contractions run as tall GEMMs with 128-row tiles.  The price is one write + one read of the aggregated matrix
// (K C 4 bytes per row; training keeps wf for the weight gradient anyway).
#include "kpconv_tile.hpp"

namespace d3f {

// measurement aid of bench.py (kpconv_fused.hip)
void* kpconv_timing_open(int which, hipStream_t stream, int Nq, int Ns, int H, int Cin, int Cout, int K);
void kpconv_timing_close(void* rec, hipStream_t stream);

// kpconv_dx_gather.hip (phase A of the gather kernel: one chunk of <= 64 compacted reverse neighbors)
template <int CV, int NSTEPS>
__device__ __forceinline__ void agg_rev_core(int n_c, float qx, float qy, float qz, float inn, __amdgpu_buffer_rsrc_t rs_g,
                                             unsigned row_bytes, unsigned col_off, float cx, float cy, float cz,
                                             float inv_extent, int lg, f32x4 (&acc)[CV]) {
  constexpr int NG = 4 * NSTEPS;
  typename VecT<CV>::type xv[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const unsigned n = (unsigned)__shfl(n_c, 4 * g + lg, 64);
    xv[g] = buf_load_vec<CV>(rs_g, n * row_bytes + col_off);
  }
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    float4 sp;
    sp.x = __shfl(qx, 4 * g + lg, 64);
    sp.y = __shfl(qy, 4 * g + lg, 64);
    sp.z = __shfl(qz, 4 * g + lg, 64);
    sp.w = 0.0f;
    const float w = kp_influence(sp, cx, cy, cz, inv_extent) * __shfl(inn, 4 * g + lg, 64);
#pragma unroll
    for (int r = 0; r < CV; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, vget<CV>(xv[g], r), acc[r], 0, 0, 0);
  }
}

// D tile of one row (lane (li, lg): rows k = 4 lg + j, channels li CV + r) -> out[(row K + k) C + cbase + li CV ..]
template <int CV>
__device__ __forceinline__ void store_agg_row(float* __restrict__ out_row, int K, int C, int cbase, int li, int lg,
                                              const f32x4 (&acc)[CV]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = 4 * lg + j;
    if (k < K) {
      float* dst = out_row + (size_t)k * C + cbase + li * CV;
      if (CV == 1) dst[0] = acc[0][j];
      if (CV == 2) *(float2*)dst = make_float2(acc[0][j], acc[CV > 1 ? 1 : 0][j]);
      if (CV == 4)
        *(float4*)dst = make_float4(acc[0][j], acc[CV > 1 ? 1 : 0][j], acc[CV > 2 ? 2 : 0][j], acc[CV > 3 ? 3 : 0][j]);
    }
  }
}

// forward direction: grid (query tiles of 16, channel chunks of 16 CV); NSTEPS = ceil(H / 16)
// (second launch bound = waves per SIMD the register allocation aims for: 114 instead of 136 VGPRs at 64 channels x 48
// neighbors, four resident waves instead of three; no spills below 49 neighbors.  The 49..64-neighbor body -- 16 gathers
// in flight per lane -- does not fit 128 registers: it asks for three waves per SIMD instead of spilling)
template <int CV, int NSTEPS>
__global__ __launch_bounds__(256, (NSTEPS >= 4 ? 3 : 4)) void kpconv_agg_fwd_kernel(const float* __restrict__ q_pts,
                                                             const float4* __restrict__ spack,
                                                             const int32_t* __restrict__ idx, const float* __restrict__ x,
                                                             const float* __restrict__ kp, int Nq, int Ns, int H, int Cin,
                                                             int K, float extent, float* __restrict__ wf,
                                                             float* __restrict__ nn_out) {
  constexpr int CC = 16 * CV;
  __shared__ float nn_l[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int q0 = blockIdx.x * 16;
  const bool klive = li < K;
  const float kx = klive ? kp[3 * li + 0] : kFarKernelPoint, ky = klive ? kp[3 * li + 1] : kFarKernelPoint,
              kz = klive ? kp[3 * li + 2] : kFarKernelPoint;
  const __amdgpu_buffer_rsrc_t rs_sp = make_rsrc(spack, (unsigned)Ns * 16u);
  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(x, (unsigned)Ns * (unsigned)Cin * 4u);
  const int cbase = blockIdx.y * CC;
  const int qw = q0 + wave * 4;
  aggregate_wave_steps<CV, NSTEPS>(
      q_pts, idx, qw, Nq, H, Ns, rs_sp, rs_x, Cin, cbase, kx, ky, kz, 1.0f / extent, lane,
      blockIdx.y == 0 ? nn_l + wave * 4 : nullptr, [&](int i, const f32x4(&acc)[CV]) {
        if (qw + i < Nq) store_agg_row<CV>(wf + (size_t)(qw + i) * K * Cin, K, Cin, cbase, li, lg, acc);
      });
  if (blockIdx.y == 0 && lane < 4 && qw + lane < Nq) nn_out[qw + lane] = nn_l[wave * 4 + lane];   // (same wave wrote it)
}

// transposed direction over the EXACT-form reverse table (reverse_table.hip: row s = its true reverse neighbors,
// compacted, as float4 {q - s, bits of q}): grid (support tiles of 16, channel chunks of 16 CV)
template <int CV>
__global__ __launch_bounds__(256) void kpconv_agg_rev_kernel(const float4* __restrict__ rev_rel, int W,
                                                             const float* __restrict__ g, const float* __restrict__ nn,
                                                             const float* __restrict__ kp, int Ns, int Nq, int Cout, int K,
                                                             float extent, float* __restrict__ A) {
  constexpr int CC = 16 * CV;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int s0 = blockIdx.x * 16 + wave * 4;
  const bool klive = li < K;
  // |(s - q) - kp| = |(q - s) + kp|: the kernel points seen from s itself are negated
  const float cx = klive ? -kp[3 * li + 0] : kFarKernelPoint, cy = klive ? -kp[3 * li + 1] : kFarKernelPoint,
              cz = klive ? -kp[3 * li + 2] : kFarKernelPoint;
  const __amdgpu_buffer_rsrc_t rs_nn = make_rsrc(nn ? nn : g, (unsigned)Nq * 4u);
  const __amdgpu_buffer_rsrc_t rs_g = make_rsrc(g, (unsigned)Nq * (unsigned)Cout * 4u);
  const float inv_extent = 1.0f / extent;
  const unsigned row_bytes = (unsigned)Cout * 4u;
  const int cbase = blockIdx.y * CC;
  const unsigned col_off = (unsigned)(cbase + li * CV) * 4u;
  // entries + 1/nn of the wave's four rows first: two memory round trips for four rows
  float4 eA[4];
  float nvA[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int s = s0 + i;
    eA[i] = rev_rel[(size_t)min(s, Ns - 1) * W + min(lane, W - 1)];
    if (!(s < Ns && lane < W)) eA[i].w = __int_as_float(Nq);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
    nvA[i] = __uint_as_float(
        __builtin_amdgcn_raw_buffer_load_b32(rs_nn, (unsigned)min(max(__float_as_int(eA[i].w), 0), Nq) * 4u, 0, 0));
#pragma unroll 1
  for (int i = 0; i < 4; ++i) {
    const int s = s0 + i;
    f32x4 acc[CV];
#pragma unroll
    for (int r = 0; r < CV; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (s < Ns) {
      float4 e = eA[0];
      float nv = nvA[0];
      for (int c0 = 0; c0 < W; c0 += 64) {
        if (c0 > 0) {   // rows longer than 64 entries (rare)
          e = rev_rel[(size_t)s * W + min(c0 + lane, W - 1)];
          if (c0 + lane >= W) e.w = __int_as_float(Nq);
          nv = __uint_as_float(
              __builtin_amdgcn_raw_buffer_load_b32(rs_nn, (unsigned)min(max(__float_as_int(e.w), 0), Nq) * 4u, 0, 0));
        }
        const int n = min(max(__float_as_int(e.w), 0), Nq);
        const int cnt = __popcll(__ballot(n < Nq));     // compacted: the live entries are a prefix
        if (cnt == 0) break;
        const float inn = n < Nq ? (nn ? 1.0f / nv : 1.0f) : 0.0f;
        const int steps = (cnt + 15) >> 4;
        if (steps == 1) agg_rev_core<CV, 1>(n, e.x, e.y, e.z, inn, rs_g, row_bytes, col_off, cx, cy, cz, inv_extent, lg, acc);
        else if (steps == 2) agg_rev_core<CV, 2>(n, e.x, e.y, e.z, inn, rs_g, row_bytes, col_off, cx, cy, cz, inv_extent, lg, acc);
        else if (steps == 3) agg_rev_core<CV, 3>(n, e.x, e.y, e.z, inn, rs_g, row_bytes, col_off, cx, cy, cz, inv_extent, lg, acc);
        else agg_rev_core<CV, 4>(n, e.x, e.y, e.z, inn, rs_g, row_bytes, col_off, cx, cy, cz, inv_extent, lg, acc);
        if (cnt < 64) break;
      }
      store_agg_row<CV>(A + (size_t)s * K * Cout, K, Cout, cbase, li, lg, acc);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) { eA[j] = eA[j + 1]; nvA[j] = nvA[j + 1]; }
  }
}

// kpconv_fused.hip
int kpconv_pack_supports(const float* s_pts, const float* x, int Ns, int Cin, float4* spack, hipStream_t stream,
                         float* zero_rows);

int kpconv_aggregate_direct(const float* q_pts, int Nq, const float* s_pts, int Ns, const int32_t* idx, int H,
                            const float* x, int Cin, const float* kp, int K, float extent, float* wf_out, float* nn_out,
                            void* spack_keep, float* grad_x_clear, void* ws, hipStream_t stream) {
  float4* spack = (float4*)(spack_keep ? spack_keep : ws);
  if (!(grad_x_clear == D3F_SPACK_READY && spack_keep)) {   // (else: packed by the epilogue that produced x)
    const int rc = kpconv_pack_supports(s_pts, x, Ns, Cin, spack, stream,
                                        grad_x_clear == D3F_SPACK_READY ? nullptr : grad_x_clear);
    if (rc) return rc;
  }
  const int CV = Cin == 16 ? 1 : (Cin == 32 ? 2 : 4);
  dim3 grid(cdiv(Nq, 16), Cin / (16 * CV));
  const int nsteps = (H + 15) >> 4;   // H <= 64 (kpconv_fused_supported)
  void* timing = kpconv_timing_open(5, stream, Nq, Ns, H, Cin, 0, K);
#define D3F_AGGF(CVV, NS) \
  kpconv_agg_fwd_kernel<CVV, NS><<<grid, 256, 0, stream>>>(q_pts, spack, idx, x, kp, Nq, Ns, H, Cin, K, extent, wf_out, nn_out)
#define D3F_AGGF_STEPS(CVV)            \
  {                                    \
    if (nsteps <= 1) D3F_AGGF(CVV, 1); \
    else if (nsteps == 2) D3F_AGGF(CVV, 2); \
    else if (nsteps == 3) D3F_AGGF(CVV, 3); \
    else D3F_AGGF(CVV, 4);             \
  }
  if (CV == 1) D3F_AGGF_STEPS(1)
  else if (CV == 2) D3F_AGGF_STEPS(2)
  else D3F_AGGF_STEPS(4)
#undef D3F_AGGF_STEPS
#undef D3F_AGGF
  kpconv_timing_close(timing, stream);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

}  // namespace d3f

extern "C" {

int d3f_kpconv_aggregate_transposed_supported(int Cout, int K) {
  return ((Cout == 16 || Cout == 32 || (Cout % 64 == 0 && Cout <= 1024)) && K >= 1 && K <= 16) ? 1 : 0;
}

int d3f_kpconv_aggregate_transposed(const float* rev_rel, int rev_width, int Ns, int Nq, const float* kernel_points,
                                    int K, float extent, const float* nn, const float* grad_out, int Cout,
                                    float* agg_out, void* stream) {
  if (!rev_rel || !kernel_points || !grad_out || !agg_out || Ns < 0 || Nq < 1 || rev_width < 1 ||
      !d3f_kpconv_aggregate_transposed_supported(Cout, K) || !(extent > 0.0f) || ((uintptr_t)rev_rel & 15u))
    return D3F_EINVAL;
  if ((double)Nq * Cout * 4.0 >= 4294967295.0 || (double)Ns * rev_width >= 2147483647.0) return D3F_EINVAL;
  if (Ns == 0) return D3F_OK;
  hipStream_t st = (hipStream_t)stream;
  const int CV = Cout == 16 ? 1 : (Cout == 32 ? 2 : 4);
  dim3 grid(d3f::cdiv(Ns, 16), Cout / (16 * CV));
  const float4* rel = (const float4*)rev_rel;
  void* timing = d3f::kpconv_timing_open(6, st, Nq, Ns, rev_width, 0, Cout, K);
  if (CV == 1) d3f::kpconv_agg_rev_kernel<1><<<grid, 256, 0, st>>>(rel, rev_width, grad_out, nn, kernel_points, Ns, Nq, Cout, K, extent, agg_out);
  else if (CV == 2) d3f::kpconv_agg_rev_kernel<2><<<grid, 256, 0, st>>>(rel, rev_width, grad_out, nn, kernel_points, Ns, Nq, Cout, K, extent, agg_out);
  else d3f::kpconv_agg_rev_kernel<4><<<grid, 256, 0, st>>>(rel, rev_width, grad_out, nn, kernel_points, Ns, Nq, Cout, K, extent, agg_out);
  d3f::kpconv_timing_close(timing, st);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

}  // extern "C"
