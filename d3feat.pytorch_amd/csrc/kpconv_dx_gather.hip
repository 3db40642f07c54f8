// KPConv grad-input as a GATHER over the reverse neighbor table -- no atomics, bit-reproducible.
//
// Reference: autograd of models/blocks.py:359-380.  With gn[q,:] = grad_out[q,:] / nn[q] and
// w(q,s,k) = max(0, 1 - |(s - q) - kp[k]| / extent):
//     grad_x[s, c] = sum_{(q,h): idx[q,h] = s}  sum_k w(q,s,k) * sum_o gn[q,o] W[k,c,o]
// The scatter form (kpconv_fused.hip: kpconv_bwd_dx_kernel) contracts with W first and adds one row per edge with
// float atomics: 1.6 M edges x Cin atomics at level 0, bound by the L2 atomic unit (25x write amplification measured).
// Here the sums are exchanged,
//     grad_x[s, :] = sum_k ( sum_{q in rev(s)} w(q,s,k) gn[q, :] )  @  W[k]^T ,
// which is the FORWARD operator run on the transposed graph: support s plays the query, rev(s) (the queries that list
// s, a CSR row built by reverse_table.hip) its neighborhood, gn the features, the kernel points are negated
// (|(s - q) - kp| = |(q - s) + kp|) and W[k] is applied transposed.  Same two MFMA phases as the fused forward:
//   workgroup = 4 waves = 16 support rows; output channels o walked in chunks of CC = 16*CV
//   phase A  D[k, o] += A[k, e] B[e, o]: lane (k = l & 15, g = l >> 4) computes its own influence weight
//            (x 1/nn[q] folded in) for reverse neighbor e = 4 j + g and loads its own CV-wide slice of gn[q];
//            CSR rows are walked 64 entries at a time with as many 16-entry MFMA steps as the row needs
//            (level-0 conv rows hold 40 +- 6 entries, pooling rows ~7: one step);
//   phase B  out[16 x Cin] += tile[16 x K*CC] @ W[k, c, o]^T: the B fragment of 4 reduction indices is ONE float4 of
//            W's innermost (o) dimension -- the transposed product needs no transposed copy of the weights.
#include "kpconv_tile.hpp"

namespace d3f {

// measurement aid of bench.py (kpconv_fused.hip)
void* kpconv_timing_open(int which, hipStream_t stream, int Nq, int Ns, int H, int Cin, int Cout, int K);
void kpconv_timing_close(void* rec, hipStream_t stream);

// Table form of the reverse neighborhood (d3f_radius_query_ex): the row of s holds EVERY point q of the query cloud
// within the radius; q really lists s only if s survived q's truncation to the table width, i.e. iff
// key(q,s) = (d2 bits << 32 | s) <= last_key[q] -- d2 evaluated exactly as the search did (no FMA; symmetric in q, s).
struct RevTest {
  const uint64_t* last_key;  // null: CSR form, every entry counts
  float sx, sy, sz;
  unsigned s;
  float r2;                  // > 0: the row comes from a search with a LARGER radius (the upsampling table of a pooling
                             // layer, radius 2r): only its entries with d2 < r2 belong to the transposed table
};

// Feature gathers + MFMAs of one prepared chunk: lane l holds entry l of the (compacted) list -- its query index n_c
// (Nq = none), that query's position and 1/nn (0 = none); `NSTEPS` 16-entry MFMA steps cover the live prefix.
template <int CV, int NSTEPS>
__device__ __forceinline__ void dxg_core(int n_c, float qx, float qy, float qz, float inn, __amdgpu_buffer_rsrc_t rs_g,
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
    for (int r = 0; r < CV; ++r)
      acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, vget<CV>(xv[g], r), acc[r], 0, 0, 0);
  }
}

template <int CV>
__device__ __forceinline__ void dxg_core_n(int steps, int n_c, float qx, float qy, float qz, float inn,
                                           __amdgpu_buffer_rsrc_t rs_g, unsigned row_bytes, unsigned col_off, float cx,
                                           float cy, float cz, float inv_extent, int lg, f32x4 (&acc)[CV]) {
  if (steps == 1) dxg_core<CV, 1>(n_c, qx, qy, qz, inn, rs_g, row_bytes, col_off, cx, cy, cz, inv_extent, lg, acc);
  else if (steps == 2) dxg_core<CV, 2>(n_c, qx, qy, qz, inn, rs_g, row_bytes, col_off, cx, cy, cz, inv_extent, lg, acc);
  else if (steps == 3) dxg_core<CV, 3>(n_c, qx, qy, qz, inn, rs_g, row_bytes, col_off, cx, cy, cz, inv_extent, lg, acc);
  else dxg_core<CV, 4>(n_c, qx, qy, qz, inn, rs_g, row_bytes, col_off, cx, cy, cz, inv_extent, lg, acc);
}

// Table form, one 64-entry chunk of the row of support s: membership of every entry FIRST (position + last key of the
// entry's query: 20 bytes), live entries compacted to the front of the wave through a 64-entry LDS scratch, and only
// then the Cout-wide gradient rows of the LIVE entries are gathered -- a pooling layer's transpose is the d2 < r2 head
// of a row of the (radius 2r) upsampling table: 4x fewer gathers and one MFMA step instead of three.
template <int CV>
__device__ __forceinline__ void dxg_table_chunk(int n, float qx, float qy, float qz, float nnv, uint64_t lk, bool has_nn,
                                                const RevTest& rt, int Nq, int last_lane, int32_t* status,
                                                float4* scr4, float* scr1, __amdgpu_buffer_rsrc_t rs_g,
                                                unsigned row_bytes, unsigned col_off, float cx, float cy, float cz,
                                                float inv_extent, int lane, int lg, f32x4 (&acc)[CV]) {
  const float d2 = sqdist_exact(qx, qy, qz, rt.sx, rt.sy, rt.sz);
  const uint64_t key = ((uint64_t)__float_as_uint(d2) << 32) | rt.s;
  const bool in_r = rt.r2 <= 0.0f || d2 < rt.r2;
  const bool member = n < Nq && key <= lk && in_r;
  // a full row of the wider search whose LAST entry is still within r: members may have been cut off
  if (rt.r2 > 0.0f && status && lane == last_lane && n < Nq && d2 < rt.r2) atomicOr(status, D3F_ST_WIDE_OVERFLOW);
  const uint64_t m = __ballot(member);
  const int cnt = __popcll(m);
  if (cnt == 0) return;
  const int rank = __popcll(m & ((1ull << lane) - 1ull));
  if (member) {
    scr4[rank] = make_float4(qx, qy, qz, __int_as_float(n));
    scr1[rank] = has_nn ? 1.0f / nnv : 1.0f;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const bool mine = lane < cnt;
  const float4 c = scr4[mine ? lane : 0];
  const float inn = mine ? scr1[lane] : 0.0f;
  const int n_c = mine ? __float_as_int(c.w) : Nq;
  __builtin_amdgcn_wave_barrier();
  dxg_core_n<CV>((cnt + 15) >> 4, n_c, c.x, c.y, c.z, inn, rs_g, row_bytes, col_off, cx, cy, cz, inv_extent, lg, acc);
}

template <int CV, int NBW, int WK>
__global__ __launch_bounds__(256) void kpconv_dx_gather_kernel(
    const float* __restrict__ s_pts, const float* __restrict__ q_pts, const int32_t* __restrict__ rev_ptr,
    const int32_t* __restrict__ rev_ent, const float* __restrict__ gout, const float* __restrict__ nn,
    const float* __restrict__ kp, const float* __restrict__ W, int Ns, int Nq, int Cin, int Cout, int K, float extent,
    float* __restrict__ gx, const uint64_t* __restrict__ last_key, int rev_width, float rev_r2,
    int32_t* __restrict__ status, unsigned long long* __restrict__ clk, const float4* __restrict__ rev_rel) {
  PhaseClock pc;   // laps: 0 prologue, 1 index + position loads of the 4 rows, 2 rows (membership, gathers, MFMAs),
  pc.start(clk);   //       3 barrier wait, 4 phase B, 5 second barrier + store
  constexpr int CC = 16 * CV;
  constexpr int WN = 4 / WK;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int RS = 16 * CC + 4;
  float* tile = lds;            // [16 rows][RS]: aggregated gradients A[s][k*CC + o]
  float4* scr4 = (float4*)(lds + 16 * RS) + (threadIdx.x >> 6) * 64;              // per wave: 64 compacted entries
  float* scr1 = lds + 16 * RS + 4 * 256 + (threadIdx.x >> 6) * 64;                // (table form; 5 KB per workgroup)
  float* red = lds + 16 * RS + 5 * 256;   // [16][SLAB] when WK > 1

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int s0 = blockIdx.x * 16;
  const bool klive = li < K;
  // centre of kernel point li as seen from the row's point s: |(s - q) - kp| = |q - (s - kp)|
  const float kx = klive ? kp[3 * li + 0] : kFarKernelPoint, ky = klive ? kp[3 * li + 1] : kFarKernelPoint,
              kz = klive ? kp[3 * li + 2] : kFarKernelPoint;
  const __amdgpu_buffer_rsrc_t rs_q = make_rsrc(q_pts, (unsigned)Nq * 12u);
  const __amdgpu_buffer_rsrc_t rs_nn = make_rsrc(nn ? nn : q_pts, (unsigned)Nq * 4u);
  const __amdgpu_buffer_rsrc_t rs_g = make_rsrc(gout, (unsigned)Nq * (unsigned)Cout * 4u);
  const float inv_extent = 1.0f / extent;
  const int wn = (WK == 1) ? wave : (WK == 2 ? (wave & 1) : 0);
  const int wk = (WK == 1) ? 0 : (WK == 2 ? (wave >> 1) : wave);
  constexpr int SLAB = 16 * NBW * WN;
  const int n_base = blockIdx.y * SLAB;  // Cin slab of this workgroup

  constexpr int NACC = NBW == 1 ? 2 : 1;
  f32x4 acc2[NBW][NACC];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
    for (int a = 0; a < NACC; ++a) acc2[nb][a] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (WK > 1)
    for (int t = threadIdx.x; t < 16 * SLAB; t += 256) red[t] = 0.0f;

  const unsigned row_bytes = (unsigned)Cout * 4u;
  const int nchunks = Cout / CC;
  pc.lap(0);
  for (int ch = 0; ch < nchunks; ++ch) {
    const int cbase = ch * CC;
    const unsigned col_off = (unsigned)(cbase + li * CV) * 4u;
    // ------------------------------------------------------------------ phase A: 4 rows per wave
    if (rev_rel) {
      // exact form (d3f_reverse_table_filter): row s = its true reverse neighbors, compacted, as {q - s, q}: one coalesced
      // kilobyte per row and a 4-byte gather of 1/nn per entry; the kernel points are seen from s itself (centre = -kp).
      // (A software-pipelined variant -- row i + 1's gradient-row gathers in flight under row i's MFMAs -- halved the
      // cycles per wave but needs 204 instead of 160 VGPRs at 32 channels, one occupancy step: 53.1 vs 50.4 us per
      // launch, profiles/r03_kpconv_phase_clock.txt; the kernel has no saturated unit -- MFMA 0.20, VALU 0.31, LDS 0.35,
      // TA 0.31 busy -- it is bound by how many dependent phases the resident waves overlap.)
      const int W = rev_width;
      float4 eA[4];
      float nvA[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int s = s0 + wave * 4 + i;
        // (unconditional load from a clamped address + a select on the VALUE: a conditional load with a constant
        // alternative becomes a select of pointers into scratch)
        eA[i] = rev_rel[(size_t)min(s, Ns - 1) * W + min(lane, W - 1)];
        if (!(s < Ns && lane < W)) eA[i].w = __int_as_float(Nq);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
        nvA[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
            rs_nn, (unsigned)min(max(__float_as_int(eA[i].w), 0), Nq) * 4u, 0, 0));
      if (clk) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        pc.lap(1);
      }
#pragma unroll 1
      for (int i = 0; i < 4; ++i) {
        const int s = s0 + wave * 4 + i;
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
              nv = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                  rs_nn, (unsigned)min(max(__float_as_int(e.w), 0), Nq) * 4u, 0, 0));
            }
            const int n = min(max(__float_as_int(e.w), 0), Nq);
            const int cnt = __popcll(__ballot(n < Nq));     // compacted: the live entries are a prefix
            if (cnt == 0) break;
            const float inn = n < Nq ? (nn ? 1.0f / nv : 1.0f) : 0.0f;
            dxg_core_n<CV>((cnt + 15) >> 4, n, e.x, e.y, e.z, inn, rs_g, row_bytes, col_off, -kx, -ky, -kz, inv_extent,
                           lg, acc);
            if (cnt < 64) break;
          }
        }
        store_wf_tile<CV>(tile + (wave * 4 + i) * RS, li, lg, acc);
#pragma unroll
        for (int j = 0; j < 3; ++j) { eA[j] = eA[j + 1]; nvA[j] = nvA[j + 1]; }
      }
    } else if (last_key) {
      // table form.  The index rows of the wave's four supports (two 64-entry chunks each) are fetched together, then
      // position / last key / 1/nn of every first-chunk entry of all four rows: two memory round trips for four rows.
      const int W = rev_width;
      const __amdgpu_buffer_rsrc_t rs_lk = make_rsrc(last_key, (unsigned)Nq * 8u);
      int nA[4], nB[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int s = s0 + wave * 4 + i;
        const size_t rb = (size_t)min(s, Ns - 1) * W;
        nA[i] = (s < Ns && lane < W) ? rev_ent[rb + lane] : Nq;
        nB[i] = (s < Ns && 64 + lane < W) ? rev_ent[rb + 64 + lane] : Nq;
      }
      float qx[4], qy[4], qz[4], nv[4];
      int lklo[4], lkhi[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        nA[i] = min(max(nA[i], 0), Nq);
        nB[i] = min(max(nB[i], 0), Nq);
        const unsigned n = (unsigned)nA[i];
        qx[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_q, n * 12u, 0, 0));
        qy[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_q, n * 12u + 4u, 0, 0));
        qz[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_q, n * 12u + 8u, 0, 0));
        nv[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_nn, n * 4u, 0, 0));
        const u32x2v k2 = __builtin_amdgcn_raw_buffer_load_b64(rs_lk, n * 8u, 0, 0);
        lklo[i] = (int)k2[0];
        lkhi[i] = (int)k2[1];
      }
      if (clk) {   // (only when measuring: wait for the batched loads so that lap 1 is their latency)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        pc.lap(1);
      }
#pragma unroll 1
      for (int i = 0; i < 4; ++i) {
        const int s = s0 + wave * 4 + i;
        f32x4 acc[CV];
#pragma unroll
        for (int r = 0; r < CV; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (s < Ns) {
          const float sx = s_pts[3 * (size_t)s + 0], sy = s_pts[3 * (size_t)s + 1], sz = s_pts[3 * (size_t)s + 2];
          const float cx = sx - kx, cy = sy - ky, cz = sz - kz;
          const RevTest rt = {last_key, sx, sy, sz, (unsigned)s, rev_r2};
          int32_t* st = ch == 0 ? status : nullptr;
          const uint64_t lk_i = ((uint64_t)(unsigned)lkhi[0] << 32) | (unsigned)lklo[0];
          dxg_table_chunk<CV>(nA[0], qx[0], qy[0], qz[0], nv[0], lk_i,
                              nn != nullptr, rt, Nq, W <= 64 ? W - 1 : -1, st, scr4, scr1, rs_g, row_bytes, col_off,
                              cx, cy, cz, inv_extent, lane, lg, acc);
          // rows longer than 64 entries (rare): further chunks on demand (the second one's indices are already here);
          // entries are ranked with the shadow entries last, so the first all-shadow chunk ends the row
          int nb = nB[0];
          for (int c0 = 64; c0 < W; c0 += 64) {
            if (c0 > 64) nb = (c0 + lane < W) ? min(max(rev_ent[(size_t)s * W + c0 + lane], 0), Nq) : Nq;
            if (__ballot(nb < Nq) == 0ull) break;
            const unsigned n = (unsigned)nb;
            const float bx = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_q, n * 12u, 0, 0));
            const float by = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_q, n * 12u + 4u, 0, 0));
            const float bz = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_q, n * 12u + 8u, 0, 0));
            const float bn = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_nn, n * 4u, 0, 0));
            const u32x2v k2 = __builtin_amdgcn_raw_buffer_load_b64(rs_lk, n * 8u, 0, 0);
            const int last = (W - 1 >= c0 && W - 1 < c0 + 64) ? W - 1 - c0 : -1;
            dxg_table_chunk<CV>(nb, bx, by, bz, bn, ((uint64_t)k2[1] << 32) | k2[0], nn != nullptr, rt, Nq, last, st,
                                scr4, scr1, rs_g, row_bytes, col_off, cx, cy, cz, inv_extent, lane, lg, acc);
          }
        }
        store_wf_tile<CV>(tile + (wave * 4 + i) * RS, li, lg, acc);
        // next row's registers move to slot 0 (a run-time index into the register arrays would put them in scratch)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          nA[j] = nA[j + 1]; nB[j] = nB[j + 1]; qx[j] = qx[j + 1]; qy[j] = qy[j + 1]; qz[j] = qz[j + 1];
          nv[j] = nv[j + 1]; lklo[j] = lklo[j + 1]; lkhi[j] = lkhi[j + 1];
        }
      }
    } else
#pragma unroll 1
    for (int i = 0; i < 4; ++i) {
      const int s = s0 + wave * 4 + i;
      f32x4 acc[CV];
#pragma unroll
      for (int r = 0; r < CV; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (s < Ns) {
        // CSR row: every entry counts
        const int beg = __builtin_amdgcn_readfirstlane(rev_ptr[s]);
        const int end = __builtin_amdgcn_readfirstlane(rev_ptr[s + 1]);
        const float sx = s_pts[3 * (size_t)s + 0], sy = s_pts[3 * (size_t)s + 1], sz = s_pts[3 * (size_t)s + 2];
        const float cx = sx - kx, cy = sy - ky, cz = sz - kz;
        for (int c0 = beg; c0 < end; c0 += 64) {
          const int rem = min(end - c0, 64);
          const int n = lane < rem ? min(max(rev_ent[(size_t)c0 + lane], 0), Nq) : Nq;
          // the lane's own reverse neighbor: position and 1/nn (a shadow lane reads zeros and gets weight 0)
          const float qx = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_q, (unsigned)n * 12u, 0, 0));
          const float qy = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_q, (unsigned)n * 12u + 4u, 0, 0));
          const float qz = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_q, (unsigned)n * 12u + 8u, 0, 0));
          const float nnv = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_nn, (unsigned)n * 4u, 0, 0));
          const float inn = (lane < rem && n < Nq) ? (nn ? 1.0f / nnv : 1.0f) : 0.0f;
          dxg_core_n<CV>((rem + 15) >> 4, n, qx, qy, qz, inn, rs_g, row_bytes, col_off, cx, cy, cz, inv_extent, lg, acc);
        }
      }
      store_wf_tile<CV>(tile + (wave * 4 + i) * RS, li, lg, acc);
    }
    pc.lap(2);
    __syncthreads();
    pc.lap(3);
    // ------------------------------------------------------------------ phase B: out[16 x SLAB] += tile @ W^T
    const int steps = (K * CC) >> 4;
    constexpr int BS = NBW >= 8 ? 1 : (NBW == 4 ? 2 : (NBW == 2 ? 4 : 8));
    for (int st0 = wk; st0 < steps; st0 += WK * BS) {
      float4 b[BS][NBW];
#pragma unroll
      for (int j = 0; j < BS; ++j) {
        const int sj = min(st0 + j * WK, steps - 1);
        const int kc0 = sj << 4;
        const int k = kc0 / CC, c0 = kc0 % CC;
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
          const int col = n_base + (wn + nb * WN) * 16 + li;  // input channel c
          b[j][nb] = *(const float4*)(W + ((size_t)k * Cin + col) * Cout + cbase + c0 + 4 * lg);
        }
      }
#pragma unroll
      for (int j = 0; j < BS; ++j) {
        const int sj = st0 + j * WK;
        if (sj < steps) {
          const float4 a = *(const float4*)(tile + li * RS + (sj << 4) + 4 * lg);
          const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) {
              const float bv = t == 0 ? b[j][nb].x : (t == 1 ? b[j][nb].y : (t == 2 ? b[j][nb].z : b[j][nb].w));
              acc2[nb][t % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bv, acc2[nb][t % NACC], 0, 0, 0);
            }
        }
      }
    }
    pc.lap(4);
    __syncthreads();
  }
  // ------------------------------------------------------------------ store
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
        if (s0 + rowl < Ns) gx[(size_t)(s0 + rowl) * Cin + col] = accf[nb][r];
      }
    }
  } else {
    // the WK waves that share an output block are combined in a FIXED order (wave wk = 0, 1, ...), not with atomics
    for (int turn = 0; turn < WK; ++turn) {
      if (wk == turn) {
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
          const int col = (wn + nb * WN) * 16 + li;
#pragma unroll
          for (int r = 0; r < 4; ++r) red[(4 * lg + r) * SLAB + col] += accf[nb][r];
        }
      }
      __syncthreads();
    }
    for (int t = threadIdx.x; t < 16 * SLAB; t += 256) {
      const int rowl = t / SLAB, col = n_base + t % SLAB;
      if (s0 + rowl < Ns) gx[(size_t)(s0 + rowl) * Cin + col] = red[t];
    }
  }
  pc.lap(5);
  pc.done();
}

bool kpconv_dx_gather_supported(int Cin, int Cout, int K) {
  const bool cout_ok = (Cout == 16 || Cout == 32 || (Cout % 64 == 0 && Cout <= 512));
  const bool cin_ok = (Cin == 16 || Cin == 32 || Cin == 64 || Cin == 128 || Cin == 256 || Cin == 512);
  return cin_ok && cout_ok && K >= 1 && K <= 16;
}

template <int CV>
static int launch_dxg(const float* s_pts, const float* q_pts, const int32_t* rev_ptr, const int32_t* rev_ent,
                      const float* gout, const float* nn, const float* kp, const float* W, int Ns, int Nq, int Cin,
                      int Cout, int K, float extent, float* gx, const uint64_t* last_key, int rev_width, float rev_r2,
                      int32_t* status, const float4* rev_rel, hipStream_t stream) {
  const int tiles = cdiv(Ns, 16);
  constexpr int CC = 16 * CV;
  const size_t lds_base = sizeof(float) * (size_t)(16 * (16 * CC + 4) + 5 * 256);  // tile + compaction scratch
  int slab = Cin;  // few rows: split the input channels over workgroups (deterministic; no reduction split)
  while (slab > 64 && (long long)tiles * (Cin / slab) < 512) slab >>= 1;
#define D3F_DXG(NBW, WK)                                                                                            \
  {                                                                                                                 \
    const size_t lds = lds_base + ((WK) > 1 ? sizeof(float) * 16 * (size_t)slab : 0);                               \
    dim3 grid(tiles, Cin / slab);                                                                                   \
    kpconv_dx_gather_kernel<CV, NBW, WK><<<grid, 256, lds, stream>>>(s_pts, q_pts, rev_ptr, rev_ent, gout, nn, kp, W, \
                                                                      Ns, Nq, Cin, Cout, K, extent, gx, last_key,    \
                                                                      rev_width, rev_r2, status, phase_clock_ptr(), \
                                                                      rev_rel);                                     \
  }
  void* timing = kpconv_timing_open(3, stream, Nq, Ns, 0, Cin, Cout, K);
  switch (slab) {
    case 16: D3F_DXG(1, 4) break;
    case 32: D3F_DXG(1, 2) break;
    case 64: D3F_DXG(1, 1) break;
    case 128: D3F_DXG(2, 1) break;
    case 256: D3F_DXG(4, 1) break;
    case 512: D3F_DXG(8, 1) break;
    default: return D3F_EINVAL;
  }
#undef D3F_DXG
  kpconv_timing_close(timing, stream);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

}  // namespace d3f

extern "C" {

int d3f_kpconv_grad_input_gather_supported(int Cin, int Cout, int K) {
  return d3f::kpconv_dx_gather_supported(Cin, Cout, K) ? 1 : 0;
}

int d3f_kpconv_grad_input_gather(const float* q_pts, int Nq, const float* s_pts, int Ns, const int32_t* rev_ptr,
                                 const int32_t* rev_ent, const uint64_t* rev_last_key, int rev_width, float rev_radius,
                                 const float* rev_rel, const float* kernel_points, int K, const float* weights, int Cin,
                                 int Cout, float extent, const float* nn, const float* grad_out, float* grad_x,
                                 int32_t* status, void* stream) {
  if (!q_pts || !s_pts || !kernel_points || !weights || !grad_out || !grad_x || Nq < 0 || Ns < 0 ||
      !d3f::kpconv_dx_gather_supported(Cin, Cout, K) || !(extent > 0.0f))
    return D3F_EINVAL;
  // exactly one of the three forms: CSR (rev_ptr + rev_ent), search form (rev_ent + rev_last_key), exact form (rev_rel)
  if (rev_rel) {
    if (rev_ptr || rev_ent || rev_last_key || rev_width < 1 || ((uintptr_t)rev_rel & 15u)) return D3F_EINVAL;
  } else {
    if (!rev_ent || (rev_ptr != nullptr) == (rev_last_key != nullptr) || (!rev_ptr && rev_width < 1)) return D3F_EINVAL;
  }
  if ((double)Nq * Cout * 4.0 >= 4294967295.0 || (!rev_ptr && (double)Ns * rev_width >= 2147483647.0)) return D3F_EINVAL;
  if (Ns == 0) return D3F_OK;
  hipStream_t st = (hipStream_t)stream;
  const float rev_r2 = rev_radius > 0.0f ? rev_radius * rev_radius : 0.0f;  // float32 product, like the search
  const float4* rel = (const float4*)rev_rel;
  if (Cout == 16)
    return d3f::launch_dxg<1>(s_pts, q_pts, rev_ptr, rev_ent, grad_out, nn, kernel_points, weights, Ns, Nq, Cin, Cout, K, extent, grad_x, rev_last_key, rev_width, rev_r2, status, rel, st);
  if (Cout == 32)
    return d3f::launch_dxg<2>(s_pts, q_pts, rev_ptr, rev_ent, grad_out, nn, kernel_points, weights, Ns, Nq, Cin, Cout, K, extent, grad_x, rev_last_key, rev_width, rev_r2, status, rel, st);
  return d3f::launch_dxg<4>(s_pts, q_pts, rev_ptr, rev_ent, grad_out, nn, kernel_points, weights, Ns, Nq, Cin, Cout, K, extent, grad_x, rev_last_key, rev_width, rev_r2, status, rel, st);
}

}  // extern "C"
