// Fused KPConv kernels for gfx950: neighbor gather + influence weights + K-way aggregation + (K*Cin)->Cout
// contraction + neighbor-count normalisation in ONE launch (forward), and the matching grad-input / grad-weight
// kernels; every matrix product runs on v_mfma_f32_16x16x4_f32 (exact f32 FMA chains).
//
// Reference semantics: models/blocks.py:277-380 (see kpconv.hip).  Mapping to the hardware:
//
//   workgroup = 4 waves = one tile of 16 queries.  Input channels are walked in chunks of CC = 16*CV (<= 64).
//
//   FORWARD
//   phase A (aggregation, per wave: 4 of the 16 queries, one after the other; kpconv_tile.hpp)
//     MFMA  D[k, c] += A[k, h] * B[h, c]   with  A = influence weights w[q, h, k]  (16 kernel-point rows x 4 neighbors)
//                                                  B = gathered features x[idx[q,h], c] (4 neighbors x 16 channels)
//     lane l = (k = l & 15, hh = l >> 4) computes ITS OWN A element (one sqrt per lane, the kernel point lives in
//     3 VGPRs) and loads ITS OWN B elements as one CV-wide vector, so the weights never touch LDS.
//     Supports come from a packed float4 {x, y, z, [sum_c x > 0]} array (one 16-B gather instead of four loads).
//   the 16 x (K*CC) tile of weighted features goes to LDS (row stride K*CC + 4 floats)
//   phase B (contraction): out[16 x Cout] += wf[16 x K*CC] @ W[k*Cin + c, :]; A fragments are 16-B LDS reads
//     (4 MFMA k-steps each), B fragments stream from L2; waves split the Cout blocks (and the reduction range when
//     Cout < 64, combined through LDS float atomics).  Accumulators stay in registers across channel chunks.
//   epilogue: divide by nn (count of neighbors with positive feature sum) and store.
//   Few-point / wide layers (the bottom of the U-Net: 150..600 points, 256..512 channels) would leave the chip
//   empty with one workgroup per tile, so the launch additionally splits Cout into slabs (grid.y) and the channel
//   chunks (grid.z, partial sums combined with global float atomics into a zeroed output).
//
//   GRAD INPUT   gx[idx[q,h], c] += sum_k w[q,h,k] * gW[q,k,c],   gW = (g/nn) @ W^T
//   phase 1: gW tile [16 x K*CC] by MFMA (A = g tile from LDS, B = 16 rows x 64 B of W per lane group) -> LDS
//   phase 2: per query, E[16 neighbors x 16 channels] = w^T[16 x K] @ gW[q][K x 16] by MFMA, rows scattered with
//            global float atomics (64-B contiguous per row).
//
//   GRAD WEIGHTS dW[k*Cin + c, o] = sum_q wf[q,k,c] * g[q,o]/nn[q]
//   persistent workgroups own a (32-channel, 32-output) block of dW in registers, walk their share of query tiles
//   (phase A again + a 16 x 32 gradient tile in LDS, MFMA with the QUERY as reduction index) and flush once with
//   atomics.
#include "kpconv_tile.hpp"

namespace d3f {

// ablation switches for profiling (profiles/ablate_kpconv*.py): forward bit0 skip phase A, bit1 skip phase B, bit2 skip
// stores; grad-input bit3 skip the scatter atomics, bit4 skip phase 1 (gW tile), bit5 skip phase 2
static int g_debug_flags = 0;
void kpconv_set_debug_flags(int f) { g_debug_flags = f; }
static unsigned long long* g_phase_clock = nullptr;
unsigned long long* phase_clock_ptr() { return g_phase_clock; }
void set_phase_clock(unsigned long long* p) { g_phase_clock = p; }

// Measurement aid (bench.py's roofline leg): HIP events recorded on the launch stream right around ONE kernel of
// this file (1 = fused forward, 2 = grad input), one event pair per launch, read back after a synchronisation.
struct TimedLaunch { hipEvent_t e0, e1; int shape[6]; };
static TimedLaunch* g_timed = nullptr;
static int g_timed_mask = 0, g_timed_cap = 0, g_timed_n = 0;

// which: 1 fused forward, 2 scatter-form grad input, 3 gather-form grad input (kpconv_dx_gather.hip), 4 the A^T B weight
// gradient (linear.hip: partial + reduce launches; shape = {R, 0, 0, M, N, 0}), 5 / 6 the forward / transposed aggregation
// kernels (kpconv_aggregate.hip; 6: shape = {Nq, Ns, table width, 0, Cout, K}), 7 a GROUPED A^T B launch pair
// (linear.hip: shape = {problems, MiFLOP, KiB, workgroups, 0, 0}).  The kernel id is kept in the record as
// shape[5] = K | which << 8.
void* kpconv_timing_open(int which, hipStream_t stream, int Nq, int Ns, int H, int Cin, int Cout, int K) {
  if (!g_timed || !(g_timed_mask & (1 << (which - 1))) || g_timed_n >= g_timed_cap) return nullptr;
  TimedLaunch* t = &g_timed[g_timed_n++];
  const int sh[6] = {Nq, Ns, H, Cin, Cout, K | (which << 8)};
  for (int i = 0; i < 6; ++i) t->shape[i] = sh[i];
  (void)hipEventRecord(t->e0, stream);
  return t;
}
void kpconv_timing_close(void* rec, hipStream_t stream) {
  if (rec) (void)hipEventRecord(((TimedLaunch*)rec)->e1, stream);
}

struct TimingScope {
  hipStream_t st;
  void* t;
  TimingScope(int which, hipStream_t stream, int Nq, int Ns, int H, int Cin, int Cout, int K)
      : st(stream), t(kpconv_timing_open(which, stream, Nq, Ns, H, Cin, Cout, K)) {}
  ~TimingScope() { kpconv_timing_close(t, st); }
};

// which: one kernel id (1..7) or, negative, a mask of ids: -(bit0 | bit1 | ... | bit6)
int kpconv_timing_begin(int which, int max_launches) {
  const int mask = which < 0 ? -which : (which >= 1 && which <= 7 ? 1 << (which - 1) : 0);
  if (g_timed || mask == 0 || mask > 127 || max_launches < 1) return D3F_EINVAL;
  g_timed = new TimedLaunch[max_launches];
  for (int i = 0; i < max_launches; ++i)
    if (hipEventCreate(&g_timed[i].e0) != hipSuccess || hipEventCreate(&g_timed[i].e1) != hipSuccess) return D3F_ELAUNCH;
  g_timed_cap = max_launches;
  g_timed_n = 0;
  g_timed_mask = mask;
  return D3F_OK;
}

// the caller has synchronised the device; returns the number of launches recorded (<= cap written)
int kpconv_timing_end(float* ms_out, int* shapes_out, int cap) {
  if (!g_timed) return D3F_EINVAL;
  const int n = g_timed_n;
  for (int i = 0; i < n && i < cap; ++i) {
    float ms = 0.0f;
    (void)hipEventElapsedTime(&ms, g_timed[i].e0, g_timed[i].e1);
    ms_out[i] = ms;
    for (int j = 0; j < 6; ++j) shapes_out[6 * i + j] = g_timed[i].shape[j];
  }
  for (int i = 0; i < g_timed_cap; ++i) { (void)hipEventDestroy(g_timed[i].e0); (void)hipEventDestroy(g_timed[i].e1); }
  delete[] g_timed;
  g_timed = nullptr;
  g_timed_mask = g_timed_cap = g_timed_n = 0;
  return n;
}

// spack[n] = {s.x, s.y, s.z, (sum_c x[n,c] > 0) ? 1 : 0}; 16 lanes cooperate on one support row.
// In backward the same lanes clear row n of grad_x (the scatter target) -- no separate fill launch.
__global__ __launch_bounds__(256) void pack_supports_kernel(const float* __restrict__ s_pts,
                                                            const float* __restrict__ x, int Ns, int Cin,
                                                            float4* __restrict__ spack, float* __restrict__ zero_rows) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = t >> 4, l = t & 15;
  float s = 0.0f;
  if (n < Ns)
    for (int c = l; c < Cin; c += 16) {
      s += x[(size_t)n * Cin + c];
      if (zero_rows) zero_rows[(size_t)n * Cin + c] = 0.0f;
    }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (n < Ns && l == 0)
    spack[n] = make_float4(s_pts[3 * (size_t)n], s_pts[3 * (size_t)n + 1], s_pts[3 * (size_t)n + 2],
                           s > 0.0f ? 1.0f : 0.0f);
}

// ================================================================================================ forward
template <int CV, int NBW, int WK>
__global__ __launch_bounds__(256) void kpconv_fwd_fused_kernel(
    const float* __restrict__ q_pts, const float4* __restrict__ spack, const int32_t* __restrict__ idx,
    const float* __restrict__ x, const float* __restrict__ kp, const float* __restrict__ W, int Nq, int Ns, int H,
    int Cin, int Cout, int K, float extent, float* __restrict__ out, float* __restrict__ nn_out,
    float* __restrict__ wf_save, int dbg, unsigned long long* __restrict__ clk) {
  constexpr int CC = 16 * CV;
  constexpr int WN = 4 / WK;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int RS = 16 * CC + 4;  // 16 kernel-point rows per query (row 15 is zero padding when K = 15)
  PhaseClock pc;   // laps: 0 prologue, 1 phase A (4 queries of the wave), 2 barrier wait, 3 wf_save, 4 phase B,
  pc.start(clk);   //       5 second barrier + epilogue
  float* wf = lds;              // [16][RS]
  float* nn_l = lds + 16 * RS;  // [16]
  float* red = nn_l + 16;       // [16][16*NBW*WN] when WK > 1

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int q0 = blockIdx.x * 16;
  const bool klive = li < K;
  const float kx = klive ? kp[3 * li + 0] : kFarKernelPoint, ky = klive ? kp[3 * li + 1] : kFarKernelPoint,
              kz = klive ? kp[3 * li + 2] : kFarKernelPoint;
  const __amdgpu_buffer_rsrc_t rs_sp = make_rsrc(spack, (unsigned)Ns * 16u);
  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(x, (unsigned)Ns * (unsigned)Cin * 4u);
  const float inv_extent = 1.0f / extent;
  const int wn = (WK == 1) ? wave : (WK == 2 ? (wave & 1) : 0);
  const int wk = (WK == 1) ? 0 : (WK == 2 ? (wave >> 1) : wave);
  constexpr int SLAB = 16 * NBW * WN;
  const int n_base = blockIdx.y * SLAB;  // Cout slab of this workgroup
  const bool split = gridDim.z > 1;      // channel chunks shared between workgroups -> atomic epilogue

  // two accumulators per output block when a wave owns a single block: v_mfma_f32_16x16x4_f32 has a 40-cycle
  // dependent-accumulator latency against a 32-cycle issue interval, so back-to-back MFMAs must alternate targets
  constexpr int NACC = NBW == 1 ? 2 : 1;
  f32x4 acc2[NBW][NACC];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
    for (int a = 0; a < NACC; ++a) acc2[nb][a] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (WK > 1)
    for (int t = threadIdx.x; t < 16 * SLAB; t += 256) red[t] = 0.0f;

  const int nchunks = Cin / CC;
  bool first = true;
  pc.lap(0);
  for (int ch = blockIdx.z; ch < nchunks; ch += gridDim.z) {
    const int cbase = ch * CC;
    // ------------------------------------------------------------------ phase A
    if (!(dbg & 1))
    aggregate_wave<CV>(q_pts, idx, q0 + wave * 4, Nq, H, Ns, rs_sp, rs_x, Cin, cbase, kx, ky, kz, inv_extent, lane,
                       first ? nn_l + wave * 4 : nullptr,
                       [&](int i, const f32x4(&acc)[CV]) { store_wf_tile<CV>(wf + (wave * 4 + i) * RS, li, lg, acc); });
    pc.lap(1);
    __syncthreads();
    pc.lap(2);
    if (first && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x < 16 && q0 + (int)threadIdx.x < Nq)
      nn_out[q0 + threadIdx.x] = nn_l[threadIdx.x];
    first = false;
    if (wf_save && blockIdx.y == 0) {
      // training: leave the weighted features behind for the backward pass (grad_W becomes one tall-skinny GEMM and
      // the aggregation is never recomputed) -- K*Cin*4 bytes per query of HBM, plentiful on this part
      constexpr int V = CC / 4;
      for (int t = threadIdx.x; t < 16 * K * V; t += 256) {
        const int v = t % V, k = (t / V) % K, ql = t / (V * K);
        if (q0 + ql < Nq)
          *(float4*)(wf_save + ((size_t)(q0 + ql) * K + k) * Cin + cbase + 4 * v) =
              *(const float4*)(wf + ql * RS + k * CC + 4 * v);
      }
    }
    pc.lap(3);
    // ------------------------------------------------------------------ phase B
    const int steps = (K * CC) >> 4;
    // The W fragments stream from L2 and each 16-row step is only 4*NBW MFMAs, so a step-at-a-time loop pays one
    // L2 round trip per step.  Instead BS steps are fetched back to back (4*NBW*BS loads in flight per lane) and
    // then consumed; BS is sized for ~32 registers of fragments.
    constexpr int BS = NBW >= 8 ? 1 : (NBW == 4 ? 2 : (NBW == 2 ? 4 : 8));
    for (int s0 = wk; s0 < ((dbg & 2) ? 0 : steps); s0 += WK * BS) {
      float b[BS][NBW][4];
#pragma unroll
      for (int j = 0; j < BS; ++j) {
        const int sj = min(s0 + j * WK, steps - 1);  // clamped: the surplus loads of the last batch are discarded
        const int kc0 = sj << 4;
        const int k = kc0 / CC, c0 = kc0 % CC;
        const float* wrow = W + (size_t)(k * Cin + cbase + c0 + 4 * lg) * Cout;
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
          const int col = n_base + (wn + nb * WN) * 16 + li;
          b[j][nb][0] = wrow[col];
          b[j][nb][1] = wrow[Cout + col];
          b[j][nb][2] = wrow[2 * Cout + col];
          b[j][nb][3] = wrow[3 * Cout + col];
        }
      }
#pragma unroll
      for (int j = 0; j < BS; ++j) {
        const int sj = s0 + j * WK;
        if (sj < steps) {
          const float4 a = *(const float4*)(wf + li * RS + (sj << 4) + 4 * lg);
          const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb)
              acc2[nb][t % NACC] =
                  __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], b[j][nb][t], acc2[nb][t % NACC], 0, 0, 0);
        }
      }
    }
    pc.lap(4);
    __syncthreads();
  }
  // ------------------------------------------------------------------ epilogue
  if (dbg & 4) return;
  f32x4 accf[NBW];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) {
    accf[nb] = acc2[nb][0];
    if (NACC == 2) accf[nb] += acc2[nb][NACC - 1];
  }
  if (WK == 1) {
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
      const int col = n_base + (wn + nb * WN) * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rowl = 4 * lg + r;
        if (q0 + rowl < Nq) {
          const float v = accf[nb][r] / nn_l[rowl];
          if (split) atomicAdd(&out[(size_t)(q0 + rowl) * Cout + col], v);
          else out[(size_t)(q0 + rowl) * Cout + col] = v;
        }
      }
    }
  } else {
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
      const int col = (wn + nb * WN) * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(&red[(4 * lg + r) * SLAB + col], accf[nb][r]);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 16 * SLAB; t += 256) {
      const int rowl = t / SLAB, col = n_base + t % SLAB;
      if (q0 + rowl < Nq) {
        const float v = red[t] / nn_l[rowl];
        if (split) atomicAdd(&out[(size_t)(q0 + rowl) * Cout + col], v);
        else out[(size_t)(q0 + rowl) * Cout + col] = v;
      }
    }
  }
  pc.lap(5);
  pc.done();
}

// ================================================================================================ grad input
template <int CV>
__global__ __launch_bounds__(256) void kpconv_bwd_dx_kernel(
    const float* __restrict__ q_pts, const float4* __restrict__ spack, const int32_t* __restrict__ idx,
    const float* __restrict__ kp, const float* __restrict__ W, const float* __restrict__ nn,
    const float* __restrict__ gout, int Nq, int Ns, int H, int Cin, int Cout, int K, float extent,
    float* __restrict__ gx, const float* __restrict__ gwf_in, int dbg, int qsplit = 1, int hsplit = 1) {
  constexpr int CC = 16 * CV;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int RS = 16 * CC + 4;
  const int GS = Cout + 4;
  // few-point layers (GEMM-fed form): blockIdx.z splits the tile's work further -- query i of every wave's four
  // (i % qsplit) and the 16-neighbor chunks (chunk % hsplit) -- so that a 159-point layer runs on ~1000 workgroups
  // instead of 80: the scatter is bound by atomic latency per workgroup, not by their total number
  const int iq = blockIdx.z % qsplit, ih = blockIdx.z / qsplit;
  float* gw = lds;            // [16][RS]   gW tile of this channel chunk
  float* gl = lds + 16 * RS;  // [16][GS]   (grad_out / nn) tile

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int q0 = blockIdx.x * 16;
  const int cbase = blockIdx.y * CC;
  const __amdgpu_buffer_rsrc_t rs_sp = make_rsrc(spack, (unsigned)Ns * 16u);

  if (gwf_in) {
    // few-point / wide layers: gW = (g/nn) W^T was produced by a library GEMM over ALL queries (phase 1 below would
    // run on a handful of workgroups there); just stage this tile's slice
    constexpr int V = CC / 4;
    for (int t = threadIdx.x; t < 16 * K * V; t += 256) {
      const int v = t % V, k = (t / V) % K, ql = t / (V * K);
      if ((ql & 3) % qsplit != iq) continue;   // rows of queries another workgroup of the split handles
      const int q = q0 + ql;
      const float4 val = q < Nq ? *(const float4*)(gwf_in + ((size_t)q * K + k) * Cin + cbase + 4 * v)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
      *(float4*)(gw + ql * RS + k * CC + 4 * v) = val;
    }
  } else {
    for (int t = threadIdx.x; t < 16 * Cout; t += 256) {
      const int r = t / Cout, o = t % Cout;
      const int q = q0 + r;
      gl[r * GS + o] = q < Nq ? gout[(size_t)q * Cout + o] / nn[q] : 0.0f;
    }
  }
  __syncthreads();
  // ---- phase 1: gW[q, kc] = sum_o g[q, o] * W[kc, o]
  const int nkb = (K * CC) >> 4;
  for (int kb = wave; kb < ((gwf_in || (dbg & 16)) ? 0 : nkb); kb += 4) {
    const int kc0 = kb << 4;
    const int k = kc0 / CC, c0 = kc0 % CC;
    const float* wr = W + (size_t)(k * Cin + cbase + c0 + li) * Cout + 4 * lg;
    const float* ar = gl + li * GS + 4 * lg;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int o0 = 0; o0 < Cout; o0 += 16) {
      const float4 a = *(const float4*)(ar + o0);
      const float4 b = *(const float4*)(wr + o0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) gw[(4 * lg + r) * RS + kc0 + li] = acc[r];
  }
  __syncthreads();
  // ---- phase 2: per query, E[h, c] = sum_k w[q,h,k] * gW[q,k,c]; scatter rows to gx
  const float inv_extent = 1.0f / extent;
  float kpx[4], kpy[4], kpz[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int k = 4 * s + lg;
    const bool live = k < K;
    kpx[s] = live ? kp[3 * k + 0] : kFarKernelPoint;
    kpy[s] = live ? kp[3 * k + 1] : kFarKernelPoint;
    kpz[s] = live ? kp[3 * k + 2] : kFarKernelPoint;
  }
#pragma unroll 1
  for (int i = iq; i < 4; i += qsplit) {
    const int ql = wave * 4 + i;
    const int q = q0 + ql;
    if (q >= Nq || (dbg & 32)) continue;
    const float qx = q_pts[3 * (size_t)q + 0], qy = q_pts[3 * (size_t)q + 1], qz = q_pts[3 * (size_t)q + 2];
    float cx[4], cy[4], cz[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) { cx[s] = qx + kpx[s]; cy[s] = qy + kpy[s]; cz[s] = qz + kpz[s]; }
    const int32_t* row = idx + (size_t)q * H;
    const float* gq = gw + ql * RS;
    for (int h0 = 16 * ih; h0 < H; h0 += 16 * hsplit) {
      const int h = h0 + li;
      const int n = (int)min((unsigned)(h < H ? row[h] : Ns), (unsigned)Ns);
      const float4 sp = buf_load_f4(rs_sp, (unsigned)n * 16u);
      float wk[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) wk[s] = n < Ns ? kp_influence(sp, cx[s], cy[s], cz[s], inv_extent) : 0.0f;
      int nrow[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) nrow[r] = __shfl(n, 4 * lg + r, 64);
#pragma unroll
      for (int cb = 0; cb < CV; ++cb) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int k = 4 * s + lg;
          const float b = k < K ? gq[k * CC + cb * 16 + li] : 0.0f;
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wk[s], b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if ((unsigned)nrow[r] < (unsigned)Ns && !(dbg & 8))
            atomicAdd(&gx[(size_t)nrow[r] * Cin + cbase + cb * 16 + li], acc[r]);
      }
    }
  }
}

// ================================================================================================ grad weights
// One workgroup owns the dW block [K x CC channels(cbase..) x 16*SB outputs(obase..)] and walks query tiles
// blockIdx.x, blockIdx.x + gridDim.x, ...   (kc-block, o-block) pairs are dealt round-robin to the 4 waves.
template <int CV, int SB>
__global__ __launch_bounds__(256) void kpconv_bwd_dw_kernel(
    const float* __restrict__ q_pts, const float4* __restrict__ spack, const int32_t* __restrict__ idx,
    const float* __restrict__ x, const float* __restrict__ kp, const float* __restrict__ nn,
    const float* __restrict__ gout, int Nq, int Ns, int H, int Cin, int Cout, int K, float extent,
    float* __restrict__ gW) {
  constexpr int CC = 16 * CV;
  constexpr int SLAB = 16 * SB;
  constexpr int GS = SLAB + 4;
  constexpr int MAXP = (16 * CV * SB + 3) / 4;  // pairs per wave (K <= 16 -> at most 16*CV kc-blocks)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int RS = 16 * CC + 4;
  float* wf = lds;            // [16][RS]
  float* gl = lds + 16 * RS;  // [16][GS]

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int cbase = blockIdx.y * CC, obase = blockIdx.z * SLAB;
  const bool klive = li < K;
  const float kx = klive ? kp[3 * li + 0] : kFarKernelPoint, ky = klive ? kp[3 * li + 1] : kFarKernelPoint,
              kz = klive ? kp[3 * li + 2] : kFarKernelPoint;
  const __amdgpu_buffer_rsrc_t rs_sp = make_rsrc(spack, (unsigned)Ns * 16u);
  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(x, (unsigned)Ns * (unsigned)Cin * 4u);
  const float inv_extent = 1.0f / extent;
  const int nkb = (K * CC) >> 4;
  const int npairs = nkb * SB;

  f32x4 acc2[MAXP];
#pragma unroll
  for (int j = 0; j < MAXP; ++j) acc2[j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int tiles = (Nq + 15) >> 4;
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int q0 = tile * 16;
    for (int t = threadIdx.x; t < 16 * SLAB; t += 256) {
      const int r = t / SLAB, o = t % SLAB;
      const int q = q0 + r;
      gl[r * GS + o] = q < Nq ? gout[(size_t)q * Cout + obase + o] / nn[q] : 0.0f;
    }
    aggregate_wave<CV>(q_pts, idx, q0 + wave * 4, Nq, H, Ns, rs_sp, rs_x, Cin, cbase, kx, ky, kz, inv_extent, lane,
                       (float*)nullptr,
                       [&](int i, const f32x4(&acc)[CV]) { store_wf_tile<CV>(wf + (wave * 4 + i) * RS, li, lg, acc); });
    __syncthreads();
    // dW[kc, o] += sum_q wf[q, kc] * g[q, o]    (A[i = kc][kk = q], B[kk = q][j = o])
#pragma unroll
    for (int j = 0; j < MAXP; ++j) {
      const int p = wave + 4 * j;
      if (p < npairs) {
        const int kb = p / SB, ob = p % SB;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const float a = wf[(4 * s + lg) * RS + kb * 16 + li];
          const float b = gl[(4 * s + lg) * GS + ob * 16 + li];
          acc2[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc2[j], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < MAXP; ++j) {
    const int p = wave + 4 * j;
    if (p < npairs) {
      const int kb = p / SB, ob = p % SB;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kc = kb * 16 + 4 * lg + r;  // D row
        const int k = kc / CC, c = kc % CC;
        atomicAdd(&gW[(size_t)(k * Cin + cbase + c) * Cout + obase + ob * 16 + li], acc2[j][r]);
      }
    }
  }
}

// ================================================================================================ host side
bool kpconv_fused_supported(int Cin, int Cout, int K, int H, int Ns) {
  const bool cin_ok = (Cin == 16 || Cin == 32 || (Cin % 64 == 0 && Cin <= 512));
  const bool cout_ok = (Cout == 16 || Cout == 32 || Cout == 64 || Cout == 128 || Cout == 256 || Cout == 512);
  const bool addr_ok = (double)Ns * Cin * 4.0 < 4294967295.0;  // 32-bit buffer offsets
  return cin_ok && cout_ok && addr_ok && K >= 1 && K <= 16 && H >= 1 && H <= 64;  // one index row per wave load
}

// linear.hip
bool atb_supported(int R, int M, int N);
size_t atb_ws_bytes(int R, int M, int N);
int atb_splitk(const float* A, const float* B, const float* row_div, int R, int M, int N, float* C, void* ws,
               hipStream_t stream, int M_out = 0);

size_t kpconv_fused_ws_bytes(int Ns) { return align_up(sizeof(float4) * (size_t)(Ns > 0 ? Ns : 1), 256); }

static int pack_supports(const float* s_pts, const float* x, int Ns, int Cin, float4* spack, hipStream_t stream,
                         float* zero_rows = nullptr) {
  if (Ns > 0) {
    pack_supports_kernel<<<cdiv((long long)Ns * 16, 256), 256, 0, stream>>>(s_pts, x, Ns, Cin, spack, zero_rows);
    D3F_LAUNCH_CHECK();
  }
  return D3F_OK;
}

int kpconv_pack_supports(const float* s_pts, const float* x, int Ns, int Cin, float4* spack, hipStream_t stream,
                         float* zero_rows) {   // kpconv_aggregate.hip
  return pack_supports(s_pts, x, Ns, Cin, spack, stream, zero_rows);
}

template <int CV>
static int launch_fused_cv(const float* q_pts, const float4* spack, const int32_t* idx, const float* x,
                           const float* kp, const float* W, int Nq, int Ns, int H, int Cin, int Cout, int K,
                           float extent, float* out, float* nn_out, float* wf_save, hipStream_t stream) {
  const int tiles = cdiv(Nq, 16);
  const int CC = 16 * CV;
  const int nchunks = Cin / CC;
  const size_t lds_base = sizeof(float) * (size_t)(16 * (16 * CC + 4) + 16);
  // Work decomposition: the whole Cout in one workgroup when there are plenty of query tiles (phase A is then
  // computed once per tile); for the deep, few-point layers split Cout into slabs (grid.y), then the channel chunks
  // (grid.z, atomic combine into a zeroed output) until the launch covers the 256 CUs a few times over.
  int slab = Cout;
  while (slab > 64 && (long long)tiles * (Cout / slab) < 512) slab >>= 1;
  int zsplit = 1;
  while (zsplit < nchunks && (long long)tiles * (Cout / slab) * zsplit < 512) zsplit <<= 1;
  if (zsplit > nchunks) zsplit = nchunks;
  if (zsplit > 1 && d3f::zero_async(out, sizeof(float) * (size_t)Nq * Cout, stream) != hipSuccess) return D3F_ELAUNCH;
#define D3F_LAUNCH(NBW, WK)                                                                                     \
  {                                                                                                             \
    const size_t lds = lds_base + ((WK) > 1 ? sizeof(float) * 16 * (size_t)slab : 0);                           \
    dim3 grid(tiles, Cout / slab, zsplit);                                                                      \
    kpconv_fwd_fused_kernel<CV, NBW, WK><<<grid, 256, lds, stream>>>(q_pts, spack, idx, x, kp, W, Nq, Ns, H, Cin, \
                                                                      Cout, K, extent, out, nn_out, wf_save,       \
                                                                      g_debug_flags, g_phase_clock);             \
  }
  TimingScope timing(1, stream, Nq, Ns, H, Cin, Cout, K);
  switch (slab) {
    case 16: D3F_LAUNCH(1, 4) break;
    case 32: D3F_LAUNCH(1, 2) break;
    case 64: D3F_LAUNCH(1, 1) break;
    case 128: D3F_LAUNCH(2, 1) break;
    case 256: D3F_LAUNCH(4, 1) break;
    case 512: D3F_LAUNCH(8, 1) break;
    default: return D3F_EINVAL;
  }
#undef D3F_LAUNCH
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

int kpconv_forward_fused(const float* q_pts, int Nq, const float* s_pts, int Ns, const int32_t* idx, int H,
                         const float* x, int Cin, const float* kp, int K, const float* W, int Cout, float extent,
                         float* out, float* nn_out, float* wf_save, void* spack_keep, float* grad_x_clear, void* ws,
                         hipStream_t stream) {
  // spack_keep: the packed supports are written to the caller's buffer (kept for the backward pass, which then skips
  // its own packing launch); grad_x_clear: the backward's scatter target, cleared here on the side
  float4* spack = (float4*)(spack_keep ? spack_keep : ws);
  if (!(grad_x_clear == D3F_SPACK_READY && spack_keep)) {   // (else: packed by the epilogue that produced x)
    int rc = pack_supports(s_pts, x, Ns, Cin, spack, stream, grad_x_clear == D3F_SPACK_READY ? nullptr : grad_x_clear);
    if (rc) return rc;
  }
  if (Cin == 16) return launch_fused_cv<1>(q_pts, spack, idx, x, kp, W, Nq, Ns, H, Cin, Cout, K, extent, out, nn_out, wf_save, stream);
  if (Cin == 32) return launch_fused_cv<2>(q_pts, spack, idx, x, kp, W, Nq, Ns, H, Cin, Cout, K, extent, out, nn_out, wf_save, stream);
  return launch_fused_cv<4>(q_pts, spack, idx, x, kp, W, Nq, Ns, H, Cin, Cout, K, extent, out, nn_out, wf_save, stream);
}

int kpconv_backward_fused(const float* q_pts, int Nq, const float* s_pts, int Ns, const int32_t* idx, int H,
                          const float* x, int Cin, const float* kp, int K, const float* W, int Cout, float extent,
                          const float* nn, const float* gout, const float* wf_saved, const void* spack_kept,
                          int gx_precleared, float* gx, float* gw, void* ws, hipStream_t stream) {
  const float4* spack = (const float4*)spack_kept;
  int rc = D3F_OK;
  if (!spack) {
    rc = pack_supports(s_pts, x, Ns, Cin, (float4*)ws, stream, gx_precleared ? nullptr : gx);  // also clears gx
    if (rc) return rc;
    spack = (const float4*)ws;
  } else if (gx && !gx_precleared) {
    if (d3f::zero_async(gx, sizeof(float) * (size_t)Ns * Cin, stream) != hipSuccess) return D3F_ELAUNCH;
  }
  const int tiles = cdiv(Nq, 16);
  if (gx) {
    const int CV = Cin == 16 ? 1 : (Cin == 32 ? 2 : 4);
    const int CC = 16 * CV;
    const size_t lds = sizeof(float) * (size_t)(16 * (16 * CC + 4) + 16 * (Cout + 4));
    dim3 grid(tiles, Cin / CC);
    TimingScope timing(2, stream, Nq, Ns, H, Cin, Cout, K);
    if (CV == 1) kpconv_bwd_dx_kernel<1><<<grid, 256, lds, stream>>>(q_pts, spack, idx, kp, W, nn, gout, Nq, Ns, H, Cin, Cout, K, extent, gx, nullptr, g_debug_flags);
    else if (CV == 2) kpconv_bwd_dx_kernel<2><<<grid, 256, lds, stream>>>(q_pts, spack, idx, kp, W, nn, gout, Nq, Ns, H, Cin, Cout, K, extent, gx, nullptr, g_debug_flags);
    else kpconv_bwd_dx_kernel<4><<<grid, 256, lds, stream>>>(q_pts, spack, idx, kp, W, nn, gout, Nq, Ns, H, Cin, Cout, K, extent, gx, nullptr, g_debug_flags);
    D3F_LAUNCH_CHECK();
  }
  if (gw && wf_saved) {
    // dW [K*Cin, Cout] = wf^T (g / nn): tall-skinny GEMM over the saved weighted features (linear.hip)
    rc = atb_splitk(wf_saved, gout, nn, Nq, K * Cin, Cout, gw, (char*)ws + kpconv_fused_ws_bytes(Ns), stream);
    if (rc) return rc;
  } else if (gw) {
    if (d3f::zero_async(gw, sizeof(float) * (size_t)K * Cin * Cout, stream) != hipSuccess) return D3F_ELAUNCH;
    const int CV = Cin == 16 ? 1 : 2;
    const int SB = Cout == 16 ? 1 : 2;
    const int CC = 16 * CV, SLAB = 16 * SB;
    const int ny = Cin / CC, nz = Cout / SLAB;
    int G = cdiv(768, ny * nz);
    if (G > tiles) G = tiles;
    if (G < 1) G = 1;
    const size_t lds = sizeof(float) * (size_t)(16 * (16 * CC + 4) + 16 * (SLAB + 4));
    dim3 grid(G, ny, nz);
#define D3F_DW(CVV, SBB)                                                                                          \
  kpconv_bwd_dw_kernel<CVV, SBB><<<grid, 256, lds, stream>>>(q_pts, spack, idx, x, kp, nn, gout, Nq, Ns, H, Cin, Cout, \
                                                             K, extent, gw)
    if (CV == 1 && SB == 1) D3F_DW(1, 1);
    else if (CV == 1) D3F_DW(1, 2);
    else if (SB == 1) D3F_DW(2, 1);
    else D3F_DW(2, 2);
#undef D3F_DW
    D3F_LAUNCH_CHECK();
  }
  return D3F_OK;
}

// Aggregation only: wf [Nq, K*Cin] = sum_h w[q,h,k] x[idx[q,h], c] and the neighbor count nn -- phase A of the fused
// kernel, one workgroup per (16-query tile, channel chunk); the contraction with W is left to a GEMM (few-point layers)
int kpconv_aggregate(const float* q_pts, int Nq, const float* s_pts, int Ns, const int32_t* idx, int H, const float* x,
                     int Cin, const float* kp, int K, float extent, float* wf_out, float* nn_out, void* spack_keep,
                     float* grad_x_clear, void* ws, hipStream_t stream) {
  float4* spack = (float4*)(spack_keep ? spack_keep : ws);
  int rc = D3F_OK;
  if (!(grad_x_clear == D3F_SPACK_READY && spack_keep)) {   // (else: packed by the epilogue that produced x)
    rc = pack_supports(s_pts, x, Ns, Cin, spack, stream, grad_x_clear == D3F_SPACK_READY ? nullptr : grad_x_clear);
    if (rc) return rc;
  }
  const int CV = Cin == 16 ? 1 : (Cin == 32 ? 2 : 4);
  const int CC = 16 * CV;
  const size_t lds = sizeof(float) * (size_t)(16 * (16 * CC + 4) + 16);
  dim3 grid(cdiv(Nq, 16), 1, Cin / CC);
  const int dbg = 2 | 4;  // no contraction, no output store
#define D3F_AGG(CVV)                                                                                                  \
  kpconv_fwd_fused_kernel<CVV, 1, 1><<<grid, 256, lds, stream>>>(q_pts, spack, idx, x, kp, nullptr, Nq, Ns, H, Cin, 64, K, \
                                                                 extent, nullptr, nn_out, wf_out, dbg, nullptr)
  if (CV == 1) D3F_AGG(1);
  else if (CV == 2) D3F_AGG(2);
  else D3F_AGG(4);
#undef D3F_AGG
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

// grad_x from a precomputed gW = (grad_out / nn) @ W^T  [Nq, K*Cin]: staging + phase 2 (scatter) of the kernel above
int kpconv_grad_input_from_gw(const float* q_pts, int Nq, const float* s_pts, int Ns, const int32_t* idx, int H,
                              const float* x, int Cin, const float* kp, int K, float extent, const float* gwf,
                              const void* spack_kept, int gx_precleared, float* gx, void* ws, hipStream_t stream) {
  const float4* spack = (const float4*)spack_kept;
  int rc = D3F_OK;
  if (!spack) {
    rc = pack_supports(s_pts, x, Ns, Cin, (float4*)ws, stream, gx_precleared ? nullptr : gx);  // also clears gx
    if (rc) return rc;
    spack = (const float4*)ws;
  } else if (!gx_precleared) {
    if (d3f::zero_async(gx, sizeof(float) * (size_t)Ns * Cin, stream) != hipSuccess) return D3F_ELAUNCH;
  }
  const int CV = Cin == 16 ? 1 : (Cin == 32 ? 2 : 4);
  const int CC = 16 * CV;
  const size_t lds = sizeof(float) * (size_t)(16 * (16 * CC + 4));
  // fill the chip: split the 16-neighbor chunks first, then the four queries of a wave (measured on the 159 / 581 /
  // 2053-point layers of S1: see DESIGN.md, few-point grad-input)
  const int base = cdiv(Nq, 16) * (Cin / CC);
  const int hchunks = cdiv(H, 16);
  int hsplit = 1, qsplit = 1;
  while (hsplit < hchunks && base * hsplit < 1024) ++hsplit;
  while (qsplit < 4 && base * hsplit * qsplit < 1024) qsplit <<= 1;
  dim3 grid(cdiv(Nq, 16), Cin / CC, hsplit * qsplit);
  TimingScope timing(2, stream, Nq, Ns, H, Cin, 0, K);  // Cout = 0: gW comes from the caller's GEMM
#define D3F_DXG(CVV)                                                                                                  \
  kpconv_bwd_dx_kernel<CVV><<<grid, 256, lds, stream>>>(q_pts, spack, idx, kp, nullptr, nullptr, nullptr, Nq, Ns, H, Cin, \
                                                        0, K, extent, gx, gwf, 0, qsplit, hsplit)
  if (CV == 1) D3F_DXG(1);
  else if (CV == 2) D3F_DXG(2);
  else D3F_DXG(4);
#undef D3F_DXG
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

}  // namespace d3f
