// Y [R,N] = epilogue( X [R,Kd] . B ) in f32 on the matrix cores, for the contractions of the network that are neither
// tall-skinny weight gradients (linear.hip: A^T B) nor many-row / narrow unary blocks (linear.hip: rowgemm_kernel):
//   * KPConv from 64 channels up: raw = wf [Nq, K Cin] . W [K Cin, Cout], out = act(raw / nn + bias)   (reference
//     models/blocks.py:362-374 followed by :473,:598,:676) -- `NN`, B stored [Kd, N];
//   * unary blocks from 128 channels up / below 4096 rows: y = act(x W^T + b1 + add + b2)   (blocks.py:481-541,686) --
//     `NT`, B = nn.Linear's weight [N, Kd];
//   * their grad-input: g W (`NN`), (g / nn) W^T (`NT`), and the transposed-aggregation form A [Ns, K Cout] . W' with
//     W'[k, o, c] = W[k, c, o] read IN PLACE from W [K, Cin, Cout] (`NT` with a block structure along the reduction: the
//     reduction index k Cout + o of column c lives at W + (k Cin + c) Cout + o -- no permuted copy of the weights).
// The library's GEMM runs these at 0.5-0.6 of the f32 matrix rate (profiles/r06_step_timeline_stack3.txt) and needs a
// separate bias / activation launch behind every one of them.
//
// Structure = the grouped weight-gradient kernel's (linear.hip, round 6), turned by 90 degrees: a workgroup owns a
// (16 TI) x (16 TJ) block of Y, its 4 waves interleave groups of 16 reduction indices, every wave streams ITS groups of
// both operand panels through a private two-slot LDS ring by LDS-DMA (buffer_load_dwordx4 ... lds: a lane's 16 bytes
// land at ring + 16 lane; the per-lane offset of a piece is loop-invariant, the group's offset rides in the SGPR
// offset), counted s_waitcnt vmcnt, no workgroup barrier in the loop, software-pipelined (the fragments of the next slot
// are read and the freed slot is refilled inside the MFMA stream).  Operand layouts:
//   X (and the NT form of B) are reduction-MINOR: piece i of a slot is 16 rows x 16 reduction indices, lane (li, lk) loads
//   the float4 X[row li][16 g + 4 lk ..], the MFMA k-step ks of the slot takes reduction index 4 lk + ks from lane group lk
//   -- so ONE ds_read_b128 at ring + 16 lane (conflict-free by construction) feeds the four k-steps of a 16-row block;
//   the NN form of B is reduction-MAJOR (the weight-gradient kernel's panels): row 4 lk + ks of the [16][16 TJ] panel.
// Output columns are dealt TJ-interleaved (tile u, lane column li = column n0 + TJ li + u) so a lane owns TJ CONSECUTIVE
// columns of a row: `add` is read and Y written as one 16 / 8 / 4-byte vector, 16 lanes = one contiguous run.
// The 4 waves' partial blocks are combined through LDS in a fixed order and every wave finishes a quarter of the block's
// rows: epilogue out = act(v / d + b1 + add + b2) in the arithmetic of bias_act_fwd_kernel (elementwise.hip), one store.
// Few rows against a long reduction (the bottom levels: 512 rows x 7680 -> 512): the reduction is split into 8
// partitions, partition p runs on XCD p (its slices of X and W come through that L2 once), raw partial blocks go to
// slabs and xw_reduce_epilogue_kernel sums them in a fixed order and applies the epilogue.  No atomics: bit-reproducible.
#include <math.h>

#include "kpconv_tile.hpp"

namespace d3f {

namespace {

constexpr int XW_RING_BYTES = 16 * 1024;      // LDS ring of one wave (two slots of the widest tile)
constexpr int XW_LDS_BYTES = 4 * XW_RING_BYTES;

struct XwArgs {
  const float* X;        // [R, lda]
  const float* W;        // NT: [N, ldb] (kblock > 0: [Kd / kblock][N][kblock]); NN: [Kd, ldb]
  float* Y;              // [R, ldy]
  float* part;           // P > 1: slabs [P][R][N]
  const float* row_div;  // [R] or null
  const float* b1;       // [N] or null
  const float* add;      // [R, ldadd] or null
  const float* b2;       // [N] or null
  float* zinit;          // side job of workgroup 0: zn floats cleared (a caller's bias-gradient accumulators)
  int R, Kd, N;
  int lda, ldb, ldy, ldadd;
  int kblock;
  int P, kpp;            // partitions of the reduction, reduction indices per partition (a multiple of 16)
  int nrb, nbj;          // row blocks, column blocks
  int zn;
  float slope;
  unsigned bytesX, bytesW;
};

__device__ __forceinline__ void xw_dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, unsigned lds_addr) {
  unsigned keep;   // M0 = LDS base of the wave-instruction (compiler-reserved: saved and restored in the statement)
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %1, %2, %4 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(rsrc), "s"(lds_addr), "s"(soff)
      : "memory");
}
template <int N>
__device__ __forceinline__ void xw_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

extern __shared__ __attribute__((aligned(1024))) unsigned char xw_smem[];

// act(v / d + b1 + add + b2) on the TJ consecutive columns a lane owns of one row; same operation order as
// bias_act_fwd_kernel
template <int TJ>
__device__ __forceinline__ void xw_epilogue_store(float (&v)[TJ], const XwArgs& g, int row, int col) {
  typedef typename VecT<TJ>::type VO;
  if (g.row_div) {
    const float d = g.row_div[row];
#pragma unroll
    for (int u = 0; u < TJ; ++u) v[u] /= d;
  }
  if (g.b1) {
    const VO b = *(const VO*)(g.b1 + col);
#pragma unroll
    for (int u = 0; u < TJ; ++u) v[u] += vget<TJ>(b, u);
  }
  if (g.add) {
    const VO a = *(const VO*)(g.add + (size_t)row * g.ldadd + col);
#pragma unroll
    for (int u = 0; u < TJ; ++u) v[u] += vget<TJ>(a, u);
  }
  if (g.b2) {
    const VO b = *(const VO*)(g.b2 + col);
#pragma unroll
    for (int u = 0; u < TJ; ++u) v[u] += vget<TJ>(b, u);
  }
  const float slope = g.slope;
#pragma unroll
  for (int u = 0; u < TJ; ++u) v[u] = v[u] > 0.0f ? v[u] : v[u] * slope;
  float* dst = g.Y + (size_t)row * g.ldy + col;
  if constexpr (TJ == 4) *(float4*)dst = make_float4(v[0], v[TJ > 1 ? 1 : 0], v[TJ > 2 ? 2 : 0], v[TJ > 3 ? 3 : 0]);
  else if constexpr (TJ == 2) *(float2*)dst = make_float2(v[0], v[TJ > 1 ? 1 : 0]);
  else dst[0] = v[0];
}

// MODE 0: B reduction-minor (NT, and its block form), MODE 1: B reduction-major (NN)
template <int TI, int TJ, int MODE>
__global__ __launch_bounds__(256, 2) void xw_gemm_kernel(const XwArgs g) {
  constexpr int BM = 16 * TI, BN = 16 * TJ;
  constexpr int A_BYTES = TI * 1024, B_BYTES = TJ * 1024, SB = A_BYTES + B_BYTES;
  constexpr int G = TI + TJ;                        // LDS-DMA pieces of a slot (1 KiB each)
  constexpr int NRB = (MODE == 1) ? 4 : TJ;         // fragment reads of the B side of a slot
  constexpr int NRD = TI + NRB;
  constexpr int NM = 4 * TI * TJ, EXTRAS = G + NRD; // MFMAs of a slot; DMA pieces + fragment reads dealt into them
  static_assert(2 * SB <= XW_RING_BYTES, "two slots per wave");
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int li = lane & 15, lk = lane >> 4;
  if (g.zinit && blockIdx.x == 0)
    for (int t = threadIdx.x; t < g.zn; t += 256) g.zinit[t] = 0.0f;
  // workgroup L runs on XCD L % 8.  Undivided reduction: every XCD takes a CONTIGUOUS run of output blocks (row block
  // major), so the column blocks that share an X panel follow each other through one L2.  Split reduction: partition
  // p = L % 8 (+ 8 ...) -- an XCD sees only its slices of X and W.
  const int L = blockIdx.x;
  const int ntiles = g.nrb * g.nbj;
  int p = 0, tile;
  if (g.P == 1) {
    const int per = (ntiles + 7) >> 3;
    tile = (L & 7) * per + (L >> 3);
    if ((L >> 3) >= per || tile >= ntiles) return;
  } else {
    p = (L & 7) + 8 * ((L >> 3) / ntiles);
    tile = (L >> 3) % ntiles;
    if (p >= g.P) return;
  }
  const int rb = tile / g.nbj, cb = tile - rb * g.nbj;
  const int m0 = rb * BM, n0 = cb * BN;
  const int k0 = p * g.kpp, k1 = min(g.Kd, k0 + g.kpp);
  const int ngroups = (k1 - k0) >> 4;
  const int n_my = (ngroups - wave + 3) >> 2;           // the 4 waves interleave groups of 16 reduction indices
  unsigned char* ring = xw_smem + wave * XW_RING_BYTES;
  const unsigned ring_addr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)ring);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.X), 0, g.bytesX, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.W), 0, g.bytesW, 0x00020000);
  // per-lane byte offsets of the pieces (loop-invariant); rows past R re-read the last row (their results are not stored)
  unsigned voA[TI], voB[TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i) voA[i] = (unsigned)(((size_t)min(m0 + 16 * i + li, g.R - 1) * g.lda + 4 * lk) * 4);
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    if constexpr (MODE == 0) {
      voB[j] = (unsigned)(((size_t)(n0 + TJ * li + j) * g.ldb + 4 * lk) * 4);
    } else {
      const int e = j * 64 + lane, row = e / (4 * TJ), col = (e % (4 * TJ)) * 4;
      voB[j] = (unsigned)(((size_t)row * g.ldb + n0 + col) * 4);
    }
  }
  // the wave's NEXT group to load: reduction index kq = kb kblk + ko (block form of NT; otherwise ONE block that never
  // ends: kb = 0).  Branch-free, all scalar.
  const int kblk = (MODE == 0 && g.kblock > 0) ? g.kblock : 0x40000000;
  const unsigned blk_bytes = (MODE == 0 && g.kblock > 0) ? (unsigned)g.N * (unsigned)g.kblock * 4u : 0u;
  int kq = k0 + 16 * wave;
  int kb = kq / kblk;
  int ko = kq - kb * kblk;
  unsigned sA = 0, sB = 0;
  auto group_offsets = [&]() {
    sA = (unsigned)kq * 4u;
    if constexpr (MODE == 0) sB = (unsigned)kb * blk_bytes + (unsigned)ko * 4u;
    else sB = (unsigned)kq * (unsigned)g.ldb * 4u;
  };
  auto next_group = [&]() {
    kq += 64;
    ko += 64;                   // (kblock is a multiple of 64: at most one block boundary per step)
    const bool wrap = ko >= kblk;
    ko = wrap ? ko - kblk : ko;
    kb = wrap ? kb + 1 : kb;
    group_offsets();
  };
  group_offsets();

  f32x4 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int u = 0; u < TJ; ++u) acc[i][u] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // piece x (0 .. G - 1: the A pieces, then the B pieces) of the next group to load, into `slot`
  auto dma_piece = [&](int x, int slot) {
    const unsigned sa = ring_addr + (unsigned)slot * SB;
#pragma unroll
    for (int j = 0; j < TI; ++j)
      if (x == j) xw_dma16(rsA, voA[j], __builtin_amdgcn_readfirstlane(sA), __builtin_amdgcn_readfirstlane(sa + j * 1024));
#pragma unroll
    for (int j = 0; j < TJ; ++j)
      if (x == TI + j)
        xw_dma16(rsB, voB[j], __builtin_amdgcn_readfirstlane(sB), __builtin_amdgcn_readfirstlane(sa + A_BYTES + j * 1024));
    if (x == G - 1) next_group();
  };
  // fragment read y (0 .. TI - 1: A piece y; then the B side) of `slot`: a[i][ks], b[ks][u]
  auto read_frag = [&](int y, int slot, float (&a)[TI][4], float (&b)[4][TJ]) {
    const unsigned char* s0 = ring + slot * SB;
#pragma unroll
    for (int i = 0; i < TI; ++i)
      if (y == i) {
        const float4 v = *(const float4*)(s0 + i * 1024 + lane * 16);
        a[i][0] = v.x; a[i][1] = v.y; a[i][2] = v.z; a[i][3] = v.w;
      }
    if constexpr (MODE == 0) {
#pragma unroll
      for (int u = 0; u < TJ; ++u)
        if (y == TI + u) {
          const float4 v = *(const float4*)(s0 + A_BYTES + u * 1024 + lane * 16);
          b[0][u] = v.x; b[1][u] = v.y; b[2][u] = v.z; b[3][u] = v.w;
        }
    } else {
      const float* sBf = (const float*)(s0 + A_BYTES);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        if (y == TI + ks) {
          const typename VecT<TJ>::type v = *(const typename VecT<TJ>::type*)(sBf + (4 * lk + ks) * BN + TJ * li);
#pragma unroll
          for (int u = 0; u < TJ; ++u) b[ks][u] = vget<TJ>(v, u);
        }
    }
  };
  // one slot's MFMAs on (a, b); dealt into them: the refill of `slot` (do_dma) and the next slot's fragments into (a2, b2)
  auto block = [&](float (&a)[TI][4], float (&b)[4][TJ], float (&a2)[TI][4], float (&b2)[4][TJ], int slot, bool do_dma) {
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      const int ks = m / (TI * TJ), u = (m / TI) % TJ, i = m % TI;
      acc[i][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][ks], b[ks][u], acc[i][u], 0, 0, 0);
#pragma unroll
      for (int e = 0; e < (EXTRAS + NM - 1) / NM; ++e) {      // (constant trip count: the MFMA loop unrolls completely)
        const int x = (m * EXTRAS) / NM + e;
        if (x >= ((m + 1) * EXTRAS) / NM) continue;
        if (x < G) {
          if (do_dma) {
            if (x == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every fragment of `slot` is in registers
            dma_piece(x, slot);
          }
        } else {
          // (unconditional: behind the wave's last group this reads a slot nobody refilled and nobody uses -- a branch here
          // makes every fragment a phi of "old or new" and the compiler waits for each read on the spot to copy it)
          read_frag(x - G, slot ^ 1, a2, b2);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  if (n_my > 0) {
    float a0[TI][4], b0[4][TJ], a1[TI][4], b1[4][TJ];
#pragma unroll
    for (int x = 0; x < G; ++x) dma_piece(x, 0);
    if (n_my > 1) {
#pragma unroll
      for (int x = 0; x < G; ++x) dma_piece(x, 1);
      xw_wait_vmcnt<G>();                                // group 0 has landed
    } else {
      xw_wait_vmcnt<0>();
    }
#pragma unroll
    for (int y = 0; y < NRD; ++y) read_frag(y, 0, a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    for (int it = 0; it < n_my; it += 2) {
      xw_wait_vmcnt<0>();                                // slot 1 (group it + 1), issued a block ago, has landed
      block(a0, b0, a1, b1, 0, it + 2 < n_my);
      if (it + 1 < n_my) {
        xw_wait_vmcnt<0>();                              // slot 0 (group it + 2)
        block(a1, b1, a0, b0, 1, it + 3 < n_my);
      }
    }
  }

  // combine: every wave parks its partial block in LDS (the rings are free), wave w finishes the (tile row block i, D row
  // r) slots s = w TI .. w TI + TI - 1 (s = 4 i + r) for all lanes: sum in the fixed order wave 0 + 1 + 2 + 3
  constexpr int NT = TI * TJ;
  float* red = (float*)xw_smem;                         // [4][NT * 256]
  __syncthreads();
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int u = 0; u < TJ; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave * (NT * 256) + ((i * TJ + u) * 4 + r) * 64 + lane] = acc[i][u][r];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < TI; ++q) {
    const int s = wave * TI + q, i = s >> 2, r = s & 3;
    const int row = m0 + 16 * i + 4 * lk + r;
    float v[TJ];
#pragma unroll
    for (int u = 0; u < TJ; ++u) {
      const int e = ((i * TJ + u) * 4 + r) * 64 + lane;
      v[u] = ((red[e] + red[NT * 256 + e]) + red[2 * NT * 256 + e]) + red[3 * NT * 256 + e];
    }
    if (row < g.R) {
      const int col = n0 + TJ * li;
      if (g.P == 1) {
        xw_epilogue_store<TJ>(v, g, row, col);
      } else {
        float* dst = g.part + ((size_t)p * g.R + row) * g.N + col;
        if constexpr (TJ == 4) *(float4*)dst = make_float4(v[0], v[TJ > 1 ? 1 : 0], v[TJ > 2 ? 2 : 0], v[TJ > 3 ? 3 : 0]);
        else if constexpr (TJ == 2) *(float2*)dst = make_float2(v[0], v[TJ > 1 ? 1 : 0]);
        else dst[0] = v[0];
      }
    }
  }
}

// second stage of a split reduction: Y = epilogue(sum_p part[p]) -- one float4 per thread, slabs summed in order
__global__ __launch_bounds__(256) void xw_reduce_epilogue_kernel(const XwArgs g) {
  const size_t e4 = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total4 = (size_t)g.R * g.N / 4;
  if (e4 >= total4) return;
  const float4* part4 = (const float4*)g.part;
  float4 s = part4[e4];
  for (int p = 1; p < g.P; ++p) {
    const float4 w = part4[(size_t)p * total4 + e4];
    s.x += w.x; s.y += w.y; s.z += w.z; s.w += w.w;
  }
  const uint32_t e = (uint32_t)e4 * 4u, row = e / (uint32_t)g.N, col = e - row * (uint32_t)g.N;
  float v[4] = {s.x, s.y, s.z, s.w};
  xw_epilogue_store<4>(v, g, (int)row, (int)col);
}

static inline int xw_tile_width(int n) { return n % 64 == 0 ? 4 : (n % 32 == 0 ? 2 : (n % 16 == 0 ? 1 : 0)); }

struct XwPlan {
  int ti, tj, nrb, nbj, P, kpp;
  unsigned grid;
  size_t slab_floats;
};

}  // namespace

bool xw_supported(int R, int Kd, int N, int mode, int kblock) {
  if (R < 1 || Kd < 16 || Kd % 16 || N < 16 || !xw_tile_width(N) || mode < 0 || mode > 1) return false;
  if (kblock != 0 && (mode != 0 || kblock < 64 || kblock % 64 || Kd % kblock)) return false;
  // byte offsets are 32-bit (buffer addressing); slabs are float4-summed
  if (((size_t)R + 64) * (size_t)Kd * 4 >= 0xfff00000ull || ((size_t)N + 64) * (size_t)Kd * 4 >= 0xfff00000ull) return false;
  if ((size_t)R * N >= 0x3fffffffull) return false;
  return true;
}

static XwPlan xw_plan(int R, int Kd, int N) {
  XwPlan a;
  const d3f_tunables& tn = tunables();
  a.tj = xw_tile_width(N);
  a.nbj = N / (16 * a.tj);
  // 64-row blocks while they fill the chip's 512 workgroup slots, 32-row blocks below
  a.ti = 4;
  if ((long long)cdiv(R, 64) * a.nbj < 512) a.ti = 2;
  if (tn.xw_rows == 2 || tn.xw_rows == 4) a.ti = tn.xw_rows;
  a.nrb = cdiv(R, 16 * a.ti);
  const long long ntiles = (long long)a.nrb * a.nbj;
  a.P = 1;
  // few output blocks against a long reduction: 8 partitions, one per XCD
  if (ntiles <= 160 && Kd >= 1024 && Kd % 128 == 0) a.P = 8;
  if (tn.xw_split == 1) a.P = 1;
  if (tn.xw_split >= 8 && tn.xw_split % 8 == 0 && Kd % (16 * tn.xw_split) == 0) a.P = tn.xw_split;
  a.kpp = a.P == 1 ? Kd : Kd / a.P;
  a.grid = a.P == 1 ? (unsigned)(8 * ((ntiles + 7) / 8)) : (unsigned)(a.P * ntiles);
  a.slab_floats = a.P == 1 ? 0 : (size_t)a.P * R * N;
  return a;
}

size_t xw_ws_bytes(int R, int Kd, int N) {
  return 256 + align_up(sizeof(float) * xw_plan(R, Kd, N).slab_floats, 256);
}

int xw_gemm(const float* X, int lda, const float* W, int ldb, int mode, int kblock, int R, int Kd, int N,
            const float* row_div, const float* b1, const float* add, int ldadd, const float* b2, float slope, float* Y,
            int ldy, float* zinit, int zn, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (!X || !W || !Y || !xw_supported(R, Kd, N, mode, kblock)) return D3F_EINVAL;
  if (lda < Kd || ldy < N || (add && ldadd < N) || (zinit && zn < 1)) return D3F_EINVAL;
  if (mode == 0 && kblock == 0 && ldb < Kd) return D3F_EINVAL;
  if (mode == 1 && ldb < N) return D3F_EINVAL;
  if ((((uintptr_t)X | (uintptr_t)W | (uintptr_t)Y | (uintptr_t)add | (uintptr_t)b1 | (uintptr_t)b2) & 15) != 0) return D3F_EINVAL;
  if ((lda | ldb | ldy | (add ? ldadd : 0)) & 3) return D3F_EINVAL;
  const XwPlan pl = xw_plan(R, Kd, N);
  if (pl.P > 1 && (!ws || ws_bytes < xw_ws_bytes(R, Kd, N))) return D3F_EWORKSPACE;
  XwArgs g;
  g.X = X; g.W = W; g.Y = Y;
  g.part = pl.P > 1 ? (float*)ws : nullptr;
  g.row_div = row_div; g.b1 = b1; g.add = add; g.b2 = b2;
  g.zinit = zinit; g.zn = zinit ? zn : 0;
  g.R = R; g.Kd = Kd; g.N = N;
  g.lda = lda; g.ldb = (mode == 0 && kblock > 0) ? kblock : ldb; g.ldy = ldy; g.ldadd = add ? ldadd : 0;
  g.kblock = kblock;
  g.P = pl.P; g.kpp = pl.kpp;
  g.nrb = pl.nrb; g.nbj = pl.nbj;
  g.slope = slope;
  const size_t bx = (size_t)R * lda * 4;
  const size_t bw = mode == 1 ? (size_t)Kd * ldb * 4 : (kblock > 0 ? (size_t)Kd * N * 4 : (size_t)N * ldb * 4);
  if (bx >= 0xfff00000ull || bw >= 0xfff00000ull) return D3F_EINVAL;
  g.bytesX = (unsigned)bx;
  g.bytesW = (unsigned)bw;
#define D3F_XW(I, J, M) \
  case (M) * 64 + (I) * 8 + (J): xw_gemm_kernel<I, J, M><<<pl.grid, 256, XW_LDS_BYTES, stream>>>(g); break
  switch (mode * 64 + pl.ti * 8 + pl.tj) {
    D3F_XW(2, 1, 0); D3F_XW(2, 2, 0); D3F_XW(2, 4, 0); D3F_XW(4, 1, 0); D3F_XW(4, 2, 0); D3F_XW(4, 4, 0);
    D3F_XW(2, 1, 1); D3F_XW(2, 2, 1); D3F_XW(2, 4, 1); D3F_XW(4, 1, 1); D3F_XW(4, 2, 1); D3F_XW(4, 4, 1);
    default: return D3F_EINVAL;
  }
#undef D3F_XW
  D3F_LAUNCH_CHECK();
  if (pl.P > 1) {
    xw_reduce_epilogue_kernel<<<cdiv((long long)R * N / 4, 256), 256, 0, stream>>>(g);
    D3F_LAUNCH_CHECK();
  }
  return D3F_OK;
}

}  // namespace d3f

extern "C" {

int d3f_gemm_epilogue_supported(int R, int K, int N, int mode, int kblock) {
  return d3f::xw_supported(R, K, N, mode, kblock) ? 1 : 0;
}

size_t d3f_gemm_epilogue_ws_bytes(int R, int K, int N) {
  if (R < 1 || K < 16 || N < 16) return 0;
  return d3f::xw_ws_bytes(R, K, N);
}

int d3f_gemm_epilogue(const float* x, int ldx, const float* w, int ldw, int mode, int kblock, int R, int K, int N,
                      const float* row_div, const float* bias1, const float* add, int ldadd, const float* bias2,
                      float slope, float* y, int ldy, float* zero_init, int zero_n, void* ws, size_t ws_bytes,
                      void* stream) {
  return d3f::xw_gemm(x, ldx, w, ldw, mode, kblock, R, K, N, row_div, bias1, add, ldadd, bias2, slope, y, ldy, zero_init,
                      zero_n, ws, ws_bytes, (hipStream_t)stream);
}

}  // extern "C"
