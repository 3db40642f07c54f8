// f32 GEMM with fused prologue / epilogue for the point-wise ("unary") layers and the few-point KPConv contractions
// of the D3Feat network (reference models/blocks.py:481-541 nn.Linear + BatchNormBlock bias + LeakyReLU, :598,:676,:686
// block epilogues, and the `weighted_features @ weights` contraction of KPConv.forward, blocks.py:375-380).
//
// Why an own kernel: the layers below the first pyramid level are GEMMs with 150..8000 rows against 64..7680 columns.
// In the training step they were 116 library launches (28 % of the f32-MFMA peak on these shapes) plus ~110 separate
// bias / activation / mask / column-sum launches at the ~4.5 us launch floor.  One kernel family here does
//   C = epi( op_A(A) . op_B(B) )
// with the elementwise work folded into the operand loads and the accumulator store:
//   operands   : each of A, B is either "KC" (reduction index contiguous in memory) or "KS" (reduction index = row),
//                so x W^T, g W, g^T x, wf W, g W^T all run without a transposed copy;
//   prologue   : A' = A * (mask > 0 ? 1 : slope)   -- the LeakyReLU backward mask evaluated on the saved activation
//                while the tile is staged (no masked-gradient tensor, no separate launch);
//   by-product : row sums of A' (the bias gradient when A' = masked-gradient^T in the weight-gradient GEMM);
//   epilogue   : (/ row_div[m]) + bias1[n] + add[m,n] (or add[idx[m],n]: nearest-upsampled coarse term) + bias2[n],
//                LeakyReLU, store; a side job clears the caller's scratch (backward accumulators).
// Mapping to gfx950: workgroup = 4 waves = 64 x 64 (or 32 x 64) tile of C, each wave FM x 2 accumulators of
// v_mfma_f32_16x16x4_f32 (independent accumulators hide the 40-cycle dependent latency).  Tiles of 64 reduction
// indices are staged through LDS (double buffered, one barrier per tile, the next tile's buffer loads in flight
// during the MFMAs).  LDS layouts per operand kind:
//   KC operand: [rows][68]  (row = 64 k + 4 pad), staged with ds_write_b128 (16 lanes = one 256-B row), fragments by
//               8-byte reads: lane (i = l & 15, g = l >> 4) reads k = 8 kk + 2 g, +1  -> two MFMA k-steps per read;
//   KS operand: [64 k][rows + 8], staged with ds_write_b128, fragments by two ds_read_b32 (k = 8 kk + 2 g + t): the
//               stride = 8 mod 16 puts the two lane groups of a half-wave 16 banks apart.
// Few-row / deep-reduction shapes (571 x 3072 x 1024, 154 x 7680 x 512) have too few tiles for 256 CUs: the
// reduction is split over grid.z, partial tiles go to a slab and a second launch sums them in a fixed order and
// applies the epilogue (bit-reproducible; no atomics).
#include "kpconv_tile.hpp"

namespace d3f {

constexpr int G_BK = 64;                // reduction indices per staged tile
constexpr int G_KQ = G_BK / 4;          // float4 per row of a KC tile
constexpr int G_KC_LD = G_BK + 2;   // floats per row of a KC tile: row stride 2 mod 32 banks -> the 16 rows of a fragment
                                    // read (8 B per lane) fall on 16 distinct bank pairs, conflict-free also when hipcc
                                    // merges two reads into ds_read2_b64 (banks mod 32); rows are only 8-B aligned,
                                    // so a staged float4 goes in as two 8-byte writes

// ROWS = 64 or 32 output rows/columns of the operand tile; a KS tile row (one reduction index) holds ROWS + 8 floats
template <int ROWS>
struct GTile {
  static constexpr int KS_LD = ROWS + 8;
  static constexpr int FLOATS = (ROWS * G_KC_LD > G_BK * KS_LD) ? ROWS * G_KC_LD : G_BK * KS_LD;  // either layout fits
  static constexpr int NV = ROWS * G_BK / 1024;  // float4 per thread per tile
};

struct GemmP {
  const float* A; const float* B; float* C;
  int M, N, K, lda, ldb, ldc;
  const float* a_mask; float mask_slope;       // prologue (mask indexed exactly like A)
  float* rowsum; float* rowsum2;               // by-product: sum_k A'[m][k] -> [M] (both receive the same values)
  const float* row_div; const float* bias1; const float* bias2;
  const float* add; int ldadd; const int32_t* add_idx; int idx_stride; int add_rows;
  float slope;
  float* zero_init; int zero_n;
  float* slab;                                 // split-K: [S][M*N (+ M)] partial results
};

// Operand tiles are fetched with raw buffer loads: an offset past the operand's extent returns zeros, which is the
// zero padding of the ragged last row tile (KC) / last reduction tile (KS) without a branch around the load -- the
// loads stay unconditional, so the compiler keeps them in flight behind counted vmcnt waits.  The other ragged
// direction (reduction tail of a KC operand, row tail of a KS operand) is a select on the loaded value.
template <bool KS, int ROWS>
__device__ __forceinline__ void g_load(__amdgpu_buffer_rsrc_t rs, int ld, int r0, int rows, int k0, int K, int tid,
                                       float4 (&v)[GTile<ROWS>::NV]) {
#pragma unroll
  for (int j = 0; j < GTile<ROWS>::NV; ++j) {
    const int f = tid + 256 * j;
    unsigned off;
    bool ok;
    if (!KS) {
      const int gr = r0 + f / G_KQ, gk = k0 + 4 * (f % G_KQ);
      ok = gk < K;
      off = ((unsigned)gr * (unsigned)ld + (unsigned)gk) * 4u;
      if (gr >= rows) off = 0xfffffff0u;
    } else {
      constexpr int Q = ROWS / 4;  // float4 per reduction index
      const int gk = k0 + f / Q, gr = r0 + 4 * (f % Q);
      ok = gr < rows;
      off = ((unsigned)gk * (unsigned)ld + (unsigned)gr) * 4u;
      if (gk >= K) off = 0xfffffff0u;
    }
    (void)ok;
    v[j] = buf_load_f4(rs, off);
  }
}

// the select half of g_load, applied when the tile is staged (a use right after the load would wait for it)
template <bool KS, int ROWS>
__device__ __forceinline__ void g_fix(int r0, int rows, int k0, int K, int tid, float4 (&v)[GTile<ROWS>::NV]) {
#pragma unroll
  for (int j = 0; j < GTile<ROWS>::NV; ++j) {
    const int f = tid + 256 * j;
    bool ok;
    if (!KS) ok = k0 + 4 * (f % G_KQ) < K;
    else ok = r0 + 4 * (f % (ROWS / 4)) < rows;
    if (!ok) v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

template <bool KS, int ROWS>
__device__ __forceinline__ void g_store(float* __restrict__ T, int tid, const float4 (&v)[GTile<ROWS>::NV]) {
#pragma unroll
  for (int j = 0; j < GTile<ROWS>::NV; ++j) {
    const int f = tid + 256 * j;
    constexpr int Q = ROWS / 4;
    if (!KS) {
      float* dst = T + (f / G_KQ) * G_KC_LD + 4 * (f % G_KQ);
      *(float2*)dst = make_float2(v[j].x, v[j].y);
      *(float2*)(dst + 2) = make_float2(v[j].z, v[j].w);
    }
    else *(float4*)(T + (f / Q) * GTile<ROWS>::KS_LD + 4 * (f % Q)) = v[j];
  }
}

// fragment values of lane (li, lg) for the two MFMA k-steps of sub-step kk: rows/columns base .. base+15
template <bool KS, int ROWS>
__device__ __forceinline__ void g_frag(const float* __restrict__ T, int base, int kk, int li, int lg, float (&o)[2]) {
  if (!KS) {
    const float2 v = *(const float2*)(T + (base + li) * G_KC_LD + kk * 8 + 2 * lg);
    o[0] = v.x;
    o[1] = v.y;
  } else {
    o[0] = T[(kk * 8 + 2 * lg) * GTile<ROWS>::KS_LD + base + li];
    o[1] = T[(kk * 8 + 2 * lg + 1) * GTile<ROWS>::KS_LD + base + li];
  }
}

__device__ __forceinline__ float g_epilogue(const GemmP& p, float v, int row, int col) {
  if (p.row_div) v /= p.row_div[row];
  if (p.bias1) v += p.bias1[col];
  if (p.add) {
    if (p.add_idx) {
      const int m = p.add_idx[(size_t)row * p.idx_stride];
      if (m >= 0 && m < p.add_rows) v += p.add[(size_t)m * p.ldadd + col];
    } else {
      v += p.add[(size_t)row * p.ldadd + col];
    }
  }
  if (p.bias2) v += p.bias2[col];
  return v > 0.0f ? v : v * p.slope;
}

// FM = accumulator fragments per wave along M: 2 -> 64 x 64 tile per workgroup, 1 -> 32 x 64 (twice the workgroups
// for the few-row levels).  Waves are laid out 2 x 2; every wave owns FM x 2 fragments (16*FM rows x 32 columns).
template <bool AKS, bool BKS, bool MASK, int FM>
__global__ __launch_bounds__(256) void gemm_tile_kernel(const GemmP p) {
  constexpr int BM = 32 * FM;
  typedef GTile<BM> TA;
  typedef GTile<64> TB;
  __shared__ __attribute__((aligned(16))) float lds[2][TA::FLOATS + TB::FLOATS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;
  const int n0 = blockIdx.x * 64, m0 = blockIdx.y * BM;
  const int S = gridDim.z, z = blockIdx.z;
  if (p.zero_init && blockIdx.x == 0 && blockIdx.y == 0 && z == 0)
    for (int t = tid; t < p.zero_n; t += 256) p.zero_init[t] = 0.0f;
  const int ktiles = (p.K + G_BK - 1) / G_BK;
  const int t_begin = (int)((long long)ktiles * z / S), t_end = (int)((long long)ktiles * (z + 1) / S);
  const int T = t_end - t_begin;

  f32x4 acc[FM][2];
#pragma unroll
  for (int a = 0; a < FM; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 asum = make_float4(0.f, 0.f, 0.f, 0.f);  // AKS: running sums of this thread's 4 A' rows
  const bool want_sum = AKS && p.rowsum != nullptr && blockIdx.x == 0;

  // extents in bytes: (outer - 1) * ld + inner elements
  const unsigned a_bytes = (unsigned)(((size_t)((AKS ? p.K : p.M) - 1) * p.lda + (AKS ? p.M : p.K)) * 4);
  const unsigned b_bytes = (unsigned)(((size_t)((BKS ? p.K : p.N) - 1) * p.ldb + (BKS ? p.N : p.K)) * 4);
  const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(p.A, a_bytes);
  const __amdgpu_buffer_rsrc_t rs_m = make_rsrc(MASK ? p.a_mask : p.A, a_bytes);
  const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(p.B, b_bytes);

  // two register sets: the loads of tile t+2 are issued before the MFMAs of tile t and consumed (staged into the
  // other LDS buffer) after the MFMAs of tile t+1 -- two compute phases of latency cover
  // one register set: the loads of tile t+1 are issued before the MFMAs of tile t (64 per wave = 2048 cycles, which
  // is what covers their latency) and staged into the other LDS buffer after them.  (A second register set for a
  // two-tile look-ahead was tried: hipcc drains vmcnt at the loop header, so it bought nothing.)
  constexpr int NM = MASK ? TA::NV : 1;
  float4 va0[TA::NV], vb0[TB::NV], vm0[NM];
  auto fetch = [&](int t, float4 (&va)[TA::NV], float4 (&vb)[TB::NV], float4 (&vm)[NM]) {
    const int k0 = (t_begin + t) * G_BK;
    g_load<AKS, BM>(rs_a, p.lda, m0, p.M, k0, p.K, tid, va);
    if (MASK) {
      float4 (&vmm)[TA::NV] = reinterpret_cast<float4 (&)[TA::NV]>(vm);
      g_load<AKS, BM>(rs_m, p.lda, m0, p.M, k0, p.K, tid, vmm);
    }
    g_load<BKS, 64>(rs_b, p.ldb, n0, p.N, k0, p.K, tid, vb);
  };
  auto stage = [&](int t, float4 (&va)[TA::NV], float4 (&vb)[TB::NV], const float4 (&vm)[NM]) {
    const int k0 = (t_begin + t) * G_BK, buf = t & 1;
    if (MASK) {
#pragma unroll
      for (int j = 0; j < TA::NV; ++j) {
        va[j].x *= vm[j % NM].x > 0.0f ? 1.0f : p.mask_slope;
        va[j].y *= vm[j % NM].y > 0.0f ? 1.0f : p.mask_slope;
        va[j].z *= vm[j % NM].z > 0.0f ? 1.0f : p.mask_slope;
        va[j].w *= vm[j % NM].w > 0.0f ? 1.0f : p.mask_slope;
      }
    }
    g_fix<AKS, BM>(m0, p.M, k0, p.K, tid, va);
    g_fix<BKS, 64>(n0, p.N, k0, p.K, tid, vb);
    g_store<AKS, BM>(lds[buf], tid, va);
    g_store<BKS, 64>(lds[buf] + TA::FLOATS, tid, vb);
    if (want_sum) {
#pragma unroll
      for (int j = 0; j < TA::NV; ++j) { asum.x += va[j].x; asum.y += va[j].y; asum.z += va[j].z; asum.w += va[j].w; }
    }
  };
  // Fragments are double-buffered in registers: the LDS reads of step ps+1 are issued BEFORE the MFMAs of step ps
  // (pinned with sched_barrier: hipcc otherwise sinks them behind the MFMAs), so their latency runs under 8*FM matrix
  // instructions.  `mid` runs after the first step's MFMAs are issued: the staging stores of the NEXT tile go there,
  // into the other LDS buffer, and execute in the LDS pipe while the matrix pipe works.
  auto compute = [&](int buf, auto&& mid) {
    const float* As = lds[buf];
    const float* Bs = lds[buf] + TA::FLOATS;
    constexpr int NP = G_BK / 16;         // pipeline steps of two sub-steps (16 reduction indices) each
    float a[2][2][FM][2], b[2][2][2][2];  // [register set][sub-step][fragment][k-step]
    auto read = [&](int pstep, int slot) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int f = 0; f < FM; ++f) g_frag<AKS, BM>(As, wr * 16 * FM + f * 16, 2 * pstep + h, li, lg, a[slot][h][f]);
#pragma unroll
        for (int f = 0; f < 2; ++f) g_frag<BKS, 64>(Bs, wc * 32 + f * 16, 2 * pstep + h, li, lg, b[slot][h][f]);
      }
    };
    read(0, 0);
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      const int cur = ps & 1;
      if (ps + 1 < NP) read(ps + 1, cur ^ 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int fa = 0; fa < FM; ++fa)
#pragma unroll
            for (int fb = 0; fb < 2; ++fb)
              acc[fa][fb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][h][fa][s], b[cur][h][fb][s], acc[fa][fb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (ps == 0) mid();
    }
  };
  // Schedule (one barrier per tile; tile t is computed from LDS buffer t & 1):
  //   the registers hold tile t+1, loaded during the previous iteration (a whole compute phase of latency cover);
  //   they are staged into the buffer tile t-1 has left, behind the first MFMAs of tile t, and refilled with the
  //   loads of tile t+2.
  if (T > 0) {
    fetch(0, va0, vb0, vm0);
    stage(0, va0, vb0, vm0);
    if (T > 1) fetch(1, va0, vb0, vm0);
  }
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    compute(t & 1, [&]() {
      if (t + 1 < T) stage(t + 1, va0, vb0, vm0);
      if (t + 2 < T) fetch(t + 2, va0, vb0, vm0);
    });
    __syncthreads();
  }

  // ---- by-product: row sums of A' over this workgroup's reduction range (fixed summation order)
  if (want_sum) {
    constexpr int Q = BM / 4, KL = 256 / Q;  // threads per reduction index, reduction lanes
    float* red = lds[0];                     // [KL][BM]; every fragment read is behind the loop's last barrier
    *(float4*)(red + (tid / Q) * BM + 4 * (tid % Q)) = asum;
    __syncthreads();
    if (tid < BM && m0 + tid < p.M) {
      float s = 0.0f;
#pragma unroll
      for (int j = 0; j < KL; ++j) s += red[j * BM + tid];
      if (S == 1) {
        p.rowsum[m0 + tid] = s;
        if (p.rowsum2) p.rowsum2[m0 + tid] = s;
      } else {
        p.slab[(size_t)z * ((size_t)p.M * p.N + p.M) + (size_t)p.M * p.N + m0 + tid] = s;
      }
    }
  }

  // ---- accumulator store: D[4 lg + r][li] of fragment (fa, fb).  Every epilogue operand of the thread's FM*8
  // outputs is loaded up front (independent loads in flight together), then combined and stored.
  if (S > 1) {
    float* slab = p.slab + (size_t)z * ((size_t)p.M * p.N + (p.rowsum ? p.M : 0));
#pragma unroll
    for (int fa = 0; fa < FM; ++fa)
#pragma unroll
      for (int fb = 0; fb < 2; ++fb) {
        const int col = n0 + wc * 32 + fb * 16 + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + wr * 16 * FM + fa * 16 + 4 * lg + r;
          if (row < p.M && col < p.N) slab[(size_t)row * p.N + col] = acc[fa][fb][r];
        }
      }
    return;
  }
  float bia1[2], bia2[2], dv[FM][4], ad[FM][2][4];
  int arow[FM][4];
#pragma unroll
  for (int fb = 0; fb < 2; ++fb) {
    const int col = min(n0 + wc * 32 + fb * 16 + li, p.N - 1);
    bia1[fb] = p.bias1 ? p.bias1[col] : 0.0f;
    bia2[fb] = p.bias2 ? p.bias2[col] : 0.0f;
  }
#pragma unroll
  for (int fa = 0; fa < FM; ++fa)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = min(m0 + wr * 16 * FM + fa * 16 + 4 * lg + r, p.M - 1);
      dv[fa][r] = p.row_div ? p.row_div[row] : 1.0f;
      arow[fa][r] = p.add_idx ? p.add_idx[(size_t)row * p.idx_stride] : row;
    }
#pragma unroll
  for (int fa = 0; fa < FM; ++fa)
#pragma unroll
    for (int fb = 0; fb < 2; ++fb) {
      const int col = min(n0 + wc * 32 + fb * 16 + li, p.N - 1);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = arow[fa][r];
        const bool live = p.add && (!p.add_idx || (m >= 0 && m < p.add_rows));
        ad[fa][fb][r] = live ? p.add[(size_t)m * p.ldadd + col] : 0.0f;
      }
    }
#pragma unroll
  for (int fa = 0; fa < FM; ++fa)
#pragma unroll
    for (int fb = 0; fb < 2; ++fb) {
      const int col = n0 + wc * 32 + fb * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wr * 16 * FM + fa * 16 + 4 * lg + r;
        float v = acc[fa][fb][r];
        if (p.row_div) v /= dv[fa][r];
        if (p.bias1) v += bia1[fb];
        if (p.add) v += ad[fa][fb][r];
        if (p.bias2) v += bia2[fb];
        v = v > 0.0f ? v : v * p.slope;
        if (row < p.M && col < p.N) p.C[(size_t)row * p.ldc + col] = v;
      }
    }
}

// C = epilogue(sum_z slab[z]) and rowsum = sum_z partial row sums, slabs added in index order
__global__ __launch_bounds__(256) void gemm_reduce_kernel(const GemmP p, int S) {
  const size_t MN = (size_t)p.M * p.N;
  const size_t stride = MN + (p.rowsum ? p.M : 0);
  const size_t i4 = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i4 * 4 < MN) {
    const size_t e = i4 * 4;
    float4 s = *(const float4*)(p.slab + e);
    for (int z = 1; z < S; ++z) {
      const float4 v = *(const float4*)(p.slab + (size_t)z * stride + e);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const int row = (int)(e / p.N), col = (int)(e % p.N);
    float* dst = p.C + (size_t)row * p.ldc + col;
    dst[0] = g_epilogue(p, s.x, row, col);
    dst[1] = g_epilogue(p, s.y, row, col + 1);
    dst[2] = g_epilogue(p, s.z, row, col + 2);
    dst[3] = g_epilogue(p, s.w, row, col + 3);
  } else if (p.rowsum) {
    const size_t m = i4 - (MN + 3) / 4;
    if (m < (size_t)p.M) {
      float s = 0.0f;
      for (int z = 0; z < S; ++z) s += p.slab[(size_t)z * stride + MN + m];
      p.rowsum[m] = s;
      if (p.rowsum2) p.rowsum2[m] = s;
    }
  }
}

// Work decomposition.  FM = 2 (64-row tiles) when that alone fills the chip; otherwise 32-row tiles; the reduction is
// split (second launch for the fixed-order sum + epilogue: ~4.5 us) only when the launch would still leave most
// CUs idle AND each slice keeps >= 4 reduction tiles (256 indices).
struct GemmPlan { int fm, split; };
static GemmPlan gemm_plan(int M, int N, int K) {
  const long long nt = cdiv(N, 64);
  const int ktiles = cdiv(K, G_BK);
  GemmPlan pl;
  pl.fm = ((long long)cdiv(M, 64) * nt >= 384) ? 2 : 1;
  const long long tiles = (long long)cdiv(M, 32 * pl.fm) * nt;
  pl.split = 1;
  if (tiles < 160 && ktiles >= 8) {
    long long s = (512 + tiles - 1) / tiles;
    if (s > ktiles / 4) s = ktiles / 4;
    if (s > 32) s = 32;
    if (s >= 2) pl.split = (int)s;
  }
  return pl;
}

size_t gemm_ws_bytes(int M, int N, int K, bool rowsum) {
  const int S = gemm_plan(M, N, K).split;
  if (S == 1) return 0;
  return align_up(sizeof(float) * (size_t)S * ((size_t)M * N + (rowsum ? M : 0)), 256);
}

static bool aligned16(const void* p) { return ((uintptr_t)p & 15u) == 0; }

int gemm_launch(GemmP p, bool a_ks, bool b_ks, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (!p.A || !p.B || !p.C || p.M < 0 || p.N < 1 || p.K < 1) return D3F_EINVAL;
  // float4 granularity along every contiguous dimension
  const int a_cont = a_ks ? p.M : p.K, b_cont = b_ks ? p.N : p.K;
  if (a_cont % 4 || b_cont % 4 || p.lda % 4 || p.ldb % 4 || p.N % 4 || p.ldc < p.N) return D3F_EINVAL;
  if (!aligned16(p.A) || !aligned16(p.B) || (p.a_mask && !aligned16(p.a_mask))) return D3F_EINVAL;
  if (p.rowsum && !a_ks) return D3F_EINVAL;  // row sums ride on the KS staging pattern (weight-gradient GEMM)
  if (p.rowsum2 && !p.rowsum) return D3F_EINVAL;
  if (p.add && p.ldadd < p.N) return D3F_EINVAL;
  if (p.M == 0) {
    if (p.zero_init && p.zero_n > 0 && zero_async(p.zero_init, sizeof(float) * (size_t)p.zero_n, stream) != hipSuccess)
      return D3F_ELAUNCH;
    return D3F_OK;
  }
  const GemmPlan pl = gemm_plan(p.M, p.N, p.K);
  const int S = pl.split;
  if (S > 1) {
    if (!ws || ws_bytes < gemm_ws_bytes(p.M, p.N, p.K, p.rowsum != nullptr)) return D3F_EWORKSPACE;
    p.slab = (float*)ws;
  }
  dim3 grid(cdiv(p.N, 64), cdiv(p.M, 32 * pl.fm), S);
  const bool mask = p.a_mask != nullptr;
#define D3F_G3(AK, BK, MK)                                                        \
  {                                                                               \
    if (pl.fm == 2) gemm_tile_kernel<AK, BK, MK, 2><<<grid, 256, 0, stream>>>(p); \
    else gemm_tile_kernel<AK, BK, MK, 1><<<grid, 256, 0, stream>>>(p);            \
  }
#define D3F_G(AK, BK) { if (mask) D3F_G3(AK, BK, true) else D3F_G3(AK, BK, false) }
  if (!a_ks && !b_ks) D3F_G(false, false)
  else if (!a_ks && b_ks) D3F_G(false, true)
  else if (a_ks && !b_ks) D3F_G(true, false)
  else D3F_G(true, true)
#undef D3F_G
#undef D3F_G3
  D3F_LAUNCH_CHECK();
  if (S > 1) {
    const long long work = (long long)((size_t)p.M * p.N + 3) / 4 + (p.rowsum ? p.M : 0);
    gemm_reduce_kernel<<<cdiv(work, 256), 256, 0, stream>>>(p, S);
    D3F_LAUNCH_CHECK();
  }
  return D3F_OK;
}

}  // namespace d3f

extern "C" {

size_t d3f_gemm_ws_bytes(int M, int N, int K, int with_rowsum) { return d3f::gemm_ws_bytes(M, N, K, with_rowsum != 0); }

int d3f_gemm(const d3f_gemm_args* a, void* ws, size_t ws_bytes, void* stream) {
  if (!a) return D3F_EINVAL;
  d3f::GemmP p;
  p.A = a->A; p.B = a->B; p.C = a->C;
  p.M = a->M; p.N = a->N; p.K = a->K; p.lda = a->lda; p.ldb = a->ldb; p.ldc = a->ldc;
  p.a_mask = a->a_mask; p.mask_slope = a->mask_slope;
  p.rowsum = a->rowsum; p.rowsum2 = a->rowsum2;
  p.row_div = a->row_div; p.bias1 = a->bias1; p.bias2 = a->bias2;
  p.add = a->add; p.ldadd = a->ldadd; p.add_idx = a->add_idx; p.idx_stride = a->idx_stride; p.add_rows = a->add_rows;
  p.slope = a->slope;
  p.zero_init = a->zero_init; p.zero_n = a->zero_n;
  p.slab = nullptr;
  if (p.add_idx && (p.idx_stride < 1 || p.add_rows < 0)) return D3F_EINVAL;
  return d3f::gemm_launch(p, a->a_layout == D3F_GEMM_KS, a->b_layout == D3F_GEMM_KS, ws, ws_bytes, (hipStream_t)stream);
}

}  // extern "C"
