// C[M,N] = A^T B for tall-skinny operands A [R,M], B [R,N] (R = number of points >> M, N): the weight gradient of
// every 1x1 "unary" convolution (reference models/blocks.py:481-515, nn.Linear inside UnaryBlock) and -- with the
// weighted-feature matrix the fused KPConv forward leaves behind -- of every KPConv (blocks.py:375-380).
//
// A library GEMM sees M x N = 64 x 128 outputs and a 38 000-long reduction and launches 8 workgroups (measured:
// ~100 us).  Here the REDUCTION is what is spread over the chip:
//   workgroup g owns rows [g*rpw, (g+1)*rpw) and one (16*TI) x (16*TJ) block of C; each of its 4 waves accumulates it with
//   v_mfma_f32_16x16x4_f32 (each lane feeds the MFMA straight from ONE TI-wide and ONE TJ-wide vector load: the
//   A fragment t of lane (i, k) is A[row k][m0 + TI*i + t], so 16 lanes read 64*TI contiguous bytes of a row), and
//   the 4 waves are summed through LDS and the partial block goes to a scratch slab; a second launch sums the
//   slabs in a fixed order.
// No atomics: the result is bit-reproducible run to run.
// fp32-MFMA bound when M*N is large, HBM bound (R*(M+N)*4 bytes, each read once per column/row block) otherwise.
#include <math.h>

#include <vector>

#include "kpconv_tile.hpp"

namespace d3f {

// measurement aid of bench.py (kpconv_fused.hip)
void* kpconv_timing_open(int which, hipStream_t stream, int Nq, int Ns, int H, int Cin, int Cout, int K);
void kpconv_timing_close(void* rec, hipStream_t stream);

template <int TI, int TJ, int U>
__global__ __launch_bounds__(256, 4) void atb_partial_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                             const float* __restrict__ row_div, int R, int M, int N,
                                                             int rows_per_wg, float* __restrict__ part) {
  typedef typename VecT<TI>::type VA;
  typedef typename VecT<TJ>::type VB;
  constexpr int HALVES = (TI * TJ >= 16) ? 2 : 1, TH = TI / HALVES;
  __shared__ float red[3][TH * TJ * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int nbj = N / (16 * TJ);
  const int m0 = (blockIdx.y / nbj) * 16 * TI, n0 = (blockIdx.y % nbj) * 16 * TJ;
  const int r0 = blockIdx.x * rows_per_wg, r1 = min(R, r0 + rows_per_wg);
  const float* ap = A + m0 + TI * li;
  const float* bp = B + n0 + TJ * li;
  f32x4 acc[TI][TJ];
#pragma unroll
  for (int t = 0; t < TI; ++t)
#pragma unroll
    for (int u = 0; u < TJ; ++u) acc[t][u] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // U = k-steps whose loads are issued together
  // the 4 waves interleave groups of 4*U rows
  for (int base = r0 + wave * 4 * U; base < r1; base += 16 * U) {
    VA a[U];
    VB b[U];
    float sc[U];
#pragma unroll
    for (int s = 0; s < U; ++s) {
      const int row = base + 4 * s + lk;
      const bool ok = row < r1;
      const size_t rr = (size_t)(ok ? row : r0);
      a[s] = *(const VA*)(ap + rr * M);
      b[s] = *(const VB*)(bp + rr * N);
      sc[s] = ok ? (row_div ? 1.0f / row_div[rr] : 1.0f) : 0.0f;
    }
#pragma unroll
    for (int s = 0; s < U; ++s)
#pragma unroll
      for (int u = 0; u < TJ; ++u) {
        const float bv = vget<TJ>(b[s], u) * sc[s];
#pragma unroll
        for (int t = 0; t < TI; ++t)
          acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(vget<TI>(a[s], t), bv, acc[t][u], 0, 0, 0);
      }
  }
  // combine the 4 waves in a fixed order (wave 0 + 1 + 2 + 3) through LDS; 64 x 64 blocks in two rounds of half the
  // A-side tiles each, so that the staging area is 24 KB instead of 48 (round 4: with the second launch bound -- 126
  // instead of 184 registers -- a CU then holds four workgroups of this kernel instead of two)
  // D[i][j] (i = 4*lk + r, j = li) is C[m0 + TI*i + t][n0 + TJ*j + u]
  float* pp = part + (size_t)blockIdx.x * M * N;
#pragma unroll
  for (int h = 0; h < HALVES; ++h) {
    if (h > 0) __syncthreads();   // wave 0 has read round h - 1
    if (wave > 0) {
#pragma unroll
      for (int tt = 0; tt < TH; ++tt)
#pragma unroll
        for (int u = 0; u < TJ; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[wave - 1][((tt * TJ + u) * 4 + r) * 64 + lane] = acc[h * TH + tt][u][r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int tt = 0; tt < TH; ++tt) {
        const int t = h * TH + tt;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* dst = pp + (size_t)(m0 + TI * (4 * lk + r) + t) * N + n0 + TJ * li;
#pragma unroll
          for (int u = 0; u < TJ; ++u) {
            const int e = ((tt * TJ + u) * 4 + r) * 64 + lane;
            dst[u] = ((acc[t][u][r] + red[0][e]) + red[1][e]) + red[2][e];
          }
        }
      }
    }
  }
}

// C[e] = sum_p part[p][e]: SUBS threads per element each sum a strided share of the slabs, combined in a fixed order
// (SUBS = 16 when the output is small and the slabs are many: 4 threads walking 128 slabs each took 21 us for a
// 32 x 32 gradient).  Only the first MN_out elements are written (a caller whose C holds fewer rows than the padded M).
template <int SUBS>
__global__ __launch_bounds__(64 * SUBS) void atb_reduce_kernel(const float* __restrict__ part, int P, size_t MN,
                                                               float* __restrict__ C, size_t MN_out) {
  const size_t e = (size_t)blockIdx.x * 64 + (threadIdx.x & 63);  // 64 consecutive elements per workgroup
  const int sub = threadIdx.x >> 6;
  __shared__ float sh[SUBS][64];
  float s0 = 0.f, s1 = 0.f;
  if (e < MN) {
    int p = sub;
    for (; p + SUBS < P; p += 2 * SUBS) {
      s0 += part[(size_t)p * MN + e];
      s1 += part[(size_t)(p + SUBS) * MN + e];
    }
    if (p < P) s0 += part[(size_t)p * MN + e];
  }
  sh[sub][threadIdx.x & 63] = s0 + s1;
  __syncthreads();
  if (sub == 0 && e < MN_out) {
    float s = sh[0][threadIdx.x];
#pragma unroll
    for (int k = 1; k < SUBS; ++k) s += sh[k][threadIdx.x];
    C[e] = s;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Grouped form (round 6): EVERY weight gradient of a backward stage in ONE launch, their slabs summed by ONE more.
//
// A weight gradient has no consumer before the optimizer, so nothing forces C_i = A_i^T B_i of layer i to run between
// the grad-input kernels of layers i and i - 1.  Launched one by one (rounds 1-5: 27 launches + 27 second stages per
// stack) every problem paid its own ramp-up, ring prologue, 4-wave combine, slab write and tail on a chip it could not
// fill (256-512 workgroups, ONE round), and the memory-bound problems (32-wide operands) never overlapped with the
// matrix-bound ones.  Here the host side queues the problems of a stage (ops.WeightGradGroup) and
//   * atb_grouped_kernel walks them all: workgroup L -> (problem, row partition, output block) through a prefix table in
//     the kernel argument (captured by value in the hipGraph node: no descriptor upload, no extra launch); partitions
//     are sized by a TIME model (a task ~ atb_task_us of one workgroup slot, the larger of its matrix time and its share
//     of the HBM stream), the chip is filled by the sum of all problems, so a problem takes as few partitions (slabs) as
//     its work needs, and problems are ordered longest task first;
//   * the inner loop is the per-wave LDS ring of round 5: a slot = 16 rows of the wave's A / B column panels filled by
//     LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction; lane l writes ring + 16 l, so a slot is simply the
//     row-major [16][16 TI] / [16][16 TJ] panels), (S - 1) slots in flight behind a counted s_waitcnt vmcnt, no
//     workgroup barrier in the loop (a wave reads only what it loaded itself), MFMA operands by one LDS vector read
//     each; a slot is refilled as soon as its fragments are in registers, i.e. BEFORE its MFMAs issue;
//   * 16 KiB of ring per wave = 64 KiB per workgroup = two workgroups per CU: one wave's fragment reads and DMA issue
//     sit under its SIMD partner's MFMAs;
//   * XCD-aware: every problem's tasks start at a multiple of 8, task t of a problem is partition (t % 8) + 8 (t / 8 /
//     nblk), so all output blocks of one row partition run on ONE XCD (workgroup L runs on XCD L % 8) and the panels
//     they share come out of that L2;
//   * atb_grouped_reduce_kernel sums the slabs of ALL problems (fixed order, float4 per thread) and the bias gradients'
//     partial column sums.
// No atomics anywhere: bit-reproducible.  The single-problem entry points use the same two kernels with one problem.
__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_addr) {
  unsigned keep;   // M0 = LDS base of the wave-instruction (compiler-reserved: saved and restored in the statement)
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_addr)
      : "memory");
}
// the compiler does not count the asm loads: outstanding LDS-DMA requests are waited for by hand
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wait until at most `rem` (<= K) groups of G requests are outstanding
template <int G, int K>
__device__ __forceinline__ void wait_groups(int rem) {
  if constexpr (K == 0) {
    wait_vmcnt<0>();
  } else {
    if (rem >= K) wait_vmcnt<G * K>();
    else wait_groups<G, K - 1>(rem);
  }
}
template <int T>
__device__ __forceinline__ void lds_frag(const float* __restrict__ row, int li, float (&v)[T]) {
  if constexpr (T == 4) {
    const float4 a = *(const float4*)(row + 4 * li);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  } else if constexpr (T == 2) {
    const float2 a = *(const float2*)(row + 2 * li);
    v[0] = a.x; v[1] = a.y;
  } else {
    v[0] = row[li];
  }
}

constexpr int ATB_GROUP_MAX = 60;          // problems per launch (both kernel-argument tables stay below 4 KiB)
constexpr int ATB_RING_BYTES = 16 * 1024;  // LDS ring of one wave
constexpr int ATB_LDS_BYTES = 4 * ATB_RING_BYTES;

struct AtbTask {       // one problem of a grouped launch, first stage
  const float* A;      // [R, M]
  const float* B;      // [R, N]
  float* part;         // slabs [P][M N] -- or, for an undivided reduction (P == 1, `direct`), the result C itself
  int R, M, N;
  int rpw, P;          // rows per partition (a multiple of 64), live partitions
  int tile;            // 16 TI + TJ
  int ldp;             // row stride of `part` (N for slabs, the caller's ldc when direct)
  int direct;          // bit 0: task t = output block t (no partitions, no slab, no second stage); bit 1: the
                       // software-pipelined body (wide tiles, operands below 4 GiB)
};
struct AtbGroup {
  int n, pad;
  int task0[ATB_GROUP_MAX];   // first workgroup of every problem (multiples of 8, ascending)
  AtbTask t[ATB_GROUP_MAX];
};

extern __shared__ __attribute__((aligned(1024))) unsigned char atb_smem[];

// the 4 waves of a workgroup hold partial sums of the same output block: combined in a fixed order (wave 0 + 1 + 2 + 3)
// through LDS -- the rings are free by then -- and stored by wave 0
template <int TI, int TJ>
__device__ __forceinline__ void atb_combine_store(f32x4 (&acc)[TI][TJ], float* __restrict__ pp, int ldp, int m0, int n0) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int li = lane & 15, lk = lane >> 4;
  constexpr int NT = TI * TJ;
  float* red = (float*)atb_smem;                          // [3][NT * 256]
  __syncthreads();                                        // every wave is out of its ring
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int u = 0; u < TJ; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(wave - 1) * (NT * 256) + ((i * TJ + u) * 4 + r) * 64 + lane] = acc[i][u][r];
  }
  __syncthreads();
  if (wave == 0) {
    // D[i][j] (i = 4 lk + r, j = li) of tile (ti, u) is C[m0 + TI i + ti][n0 + TJ j + u]: a lane stores its TJ columns of
    // one row as one 16 / 8 / 4-byte vector
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v[TJ];
#pragma unroll
        for (int u = 0; u < TJ; ++u) {
          const int e = ((i * TJ + u) * 4 + r) * 64 + lane;
          v[u] = ((acc[i][u][r] + red[e]) + red[NT * 256 + e]) + red[2 * NT * 256 + e];
        }
        float* dst = pp + (size_t)(m0 + TI * (4 * lk + r) + i) * ldp + n0 + TJ * li;
        if constexpr (TJ == 4) *(float4*)dst = make_float4(v[0], v[TJ > 1 ? 1 : 0], v[TJ > 2 ? 2 : 0], v[TJ > 3 ? 3 : 0]);
        else if constexpr (TJ == 2) *(float2*)dst = make_float2(v[0], v[TJ > 1 ? 1 : 0]);
        else dst[0] = v[0];
      }
  }
}

// task t of a problem: partition p = (t % 8) + 8 (t / 8 / nblk), output block (t / 8) % nblk
template <int TI, int TJ>
__device__ __forceinline__ void atb_task_body(const float* __restrict__ A, const float* __restrict__ B,
                                              float* __restrict__ part, int R, int M, int N, int rpw, int P, int t,
                                              int ldp, int direct) {
  constexpr int KS = 4, ROWS = 4 * KS, BM = 16 * TI, BN = 16 * TJ;
  constexpr int A_BYTES = ROWS * BM * 4, B_BYTES = ROWS * BN * 4, SB = A_BYTES + B_BYTES;
  constexpr int NA = A_BYTES / 1024, NB = B_BYTES / 1024, G = NA + NB;   // (= TI, TJ, TI + TJ)
  constexpr int S = (ATB_RING_BYTES / SB) > 8 ? 8 : (ATB_RING_BYTES / SB);
  static_assert(S >= 2, "two slots of the widest tile fit a wave's ring");
  static_assert((S - 1) * G <= 63, "vmcnt is a 6-bit counter");
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int li = lane & 15, lk = lane >> 4;
  const int nbj = N / BN, nblk = (M / BM) * nbj;
  // direct: few rows against a large output (the bottom levels' weight gradients): one task per output block, the
  // column block n0 -- hence the B panel -- of consecutive tasks cycles, so with nbj a multiple of 8 every XCD keeps
  // its own column blocks
  const int p = direct ? 0 : (t & 7) + 8 * ((t >> 3) / nblk), blk = direct ? t : (t >> 3) % nblk;
  if (p >= P || blk >= nblk) return;                    // (padding tasks of the last group of 8)
  const int m0 = (blk / nbj) * BM, n0 = (blk % nbj) * BN;
  const int r0 = p * rpw, r1 = min(R, r0 + rpw);
  const int ngroups = (r1 - r0 + ROWS - 1) / ROWS;
  const int n_my = (ngroups - wave + 3) >> 2;           // the 4 waves interleave groups of ROWS rows
  unsigned char* ring = atb_smem + wave * ATB_RING_BYTES;
  const unsigned ring_addr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)ring);
  const float* Ab = A + m0;
  const float* Bb = B + n0;

  f32x4 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int u = 0; u < TJ; ++u) acc[i][u] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto issue = [&](int it, int slot) {
    const int rg = r0 + (wave + 4 * it) * ROWS;
    const unsigned sa = __builtin_amdgcn_readfirstlane(ring_addr + (unsigned)slot * SB);
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int e = j * 64 + lane;                       // 16-byte element of the A panel
      const int row = e / (4 * TI), col = (e % (4 * TI)) * 4;
      const int gr = min(rg + row, r1 - 1);              // rows past the slice re-read its last row (weight 0)
      lds_dma16(Ab + (size_t)gr * M + col, sa + j * 1024);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int e = j * 64 + lane;
      const int row = e / (4 * TJ), col = (e % (4 * TJ)) * 4;
      const int gr = min(rg + row, r1 - 1);
      lds_dma16(Bb + (size_t)gr * N + col, sa + A_BYTES + j * 1024);
    }
  };

  // D3F_ATB_PROBE (a MEASUREMENT BUILD of this file, profiles/atb_loop_probe.py; never defined in the product): bit 0 =
  // no LDS-DMA (the MFMA / fragment-read side alone), bit 1 = no MFMAs (the LDS-DMA side alone); results are garbage
#ifndef D3F_ATB_PROBE
#define D3F_ATB_PROBE 0
#endif
  // (tried, round 6: unequal static wave priorities -- s_setprio by a hash of the task index -- so that the two waves a
  // SIMD holds do not sit in their load segments at the same time: 2-3 % SLOWER on every tile shape; not kept)
#pragma unroll
  for (int i = 0; i < S; ++i)
    if (i < n_my && !(D3F_ATB_PROBE & 1)) issue(i, i);
  int slot = 0;
  for (int it = 0; it < n_my; ++it) {
    // groups it .. min(it + S, n_my) - 1 are in flight; group `it` has landed once at most the others are outstanding
    if (!(D3F_ATB_PROBE & 1)) wait_groups<G, S - 1>(min(S - 1, n_my - 1 - it));
    const int rg = r0 + (wave + 4 * it) * ROWS;
    const float* sA = (const float*)(ring + slot * SB);
    const float* sB = (const float*)(ring + slot * SB + A_BYTES);
    // every fragment of the slot is requested before the first MFMA: the LDS latency is paid once per slot
    float a[KS][TI], b[KS][TJ];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int row = 4 * ks + lk;
      lds_frag<TI>(sA + row * BM, li, a[ks]);
      lds_frag<TJ>(sB + row * BN, li, b[ks]);
      if (rg + row >= r1) {                              // (rows past the slice: the re-read row counts 0 times)
#pragma unroll
        for (int u = 0; u < TJ; ++u) b[ks][u] = 0.0f;
      }
    }
    if (it + S < n_my && !(D3F_ATB_PROBE & 1)) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slot's fragments are in registers: refill it now,
      issue(it + S, slot);                                 // under this slot's own MFMAs
    }
    __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise sinks every read back in front of its own MFMAs)
    if (!(D3F_ATB_PROBE & 2))
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int u = 0; u < TJ; ++u)
#pragma unroll
        for (int i = 0; i < TI; ++i)
          acc[i][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks][i], b[ks][u], acc[i][u], 0, 0, 0);
    slot = (slot + 1 == S) ? 0 : slot + 1;
  }

  atb_combine_store<TI, TJ>(acc, part + (size_t)p * M * N, ldp, m0, n0);   // (direct: p = 0)
}

// ---- software-pipelined task body (round 6, the wide tiles: 64 x 64, 64 x 32, 32 x 64) ------------------------------------
// The plain body above runs [wait DMA] [8 fragment reads] [wait] [G DMA issues + their address arithmetic] [MFMAs] one
// after the other: a wave's MFMA stream pauses for ~0.3 of a slot's time, and the probe builds put the loop at 0.66 of the
// matrix rate with the DMA side and 0.83 without (profiles/r06_atb_loop_probe.txt).  Here a slot's MFMAs run on fragments
// read DURING the previous slot's MFMAs (two register sets), and both the refill of the slot just freed and the
// fragment reads of the next slot are dealt into the MFMA stream a piece every few MFMAs:
//   block k:  s_waitcnt vmcnt(0)                          (slot k + 1, issued during block k - 1, has landed)
//             MFMA x STEP, DMA piece 0 of group k + 2 -> slot k % 2 (its fragments went to registers in block k - 1) ...
//             MFMA x STEP, fragment read 0 of slot (k + 1) % 2 -> the other register set ...
// LDS-DMA goes through BUFFER loads (buffer_load_dwordx4 ... offen lds): the per-lane offset of a piece is one VGPR that
// advances by a uniform stride per group -- no 64-bit address arithmetic in the loop --, and rows past the operand's end
// (the ragged last partition) come back as ZEROS by the resource's range check: no clamp, no masking of fragments.
// (The range check covers the VGPR offset only, so the whole offset lives there; operands must stay below 4 GiB.)
__device__ __forceinline__ void lds_dma16_buf(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %1, %2, 0 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(rsrc), "s"(lds_addr)
      : "memory");
}

template <int TI, int TJ>
__device__ __forceinline__ void atb_task_body_pipe(const float* __restrict__ A, const float* __restrict__ B,
                                                   float* __restrict__ part, int R, int M, int N, int rpw, int P, int t,
                                                   int ldp, int direct) {
  constexpr int KS = 4, ROWS = 4 * KS, BM = 16 * TI, BN = 16 * TJ;
  constexpr int A_BYTES = ROWS * BM * 4, B_BYTES = ROWS * BN * 4, SB = A_BYTES + B_BYTES;
  constexpr int NA = A_BYTES / 1024, NB = B_BYTES / 1024, G = NA + NB;   // (= TI, TJ, TI + TJ)
  constexpr int NM = KS * TI * TJ, EXTRAS = G + 2 * KS;     // MFMAs of a slot; DMA pieces + fragment reads dealt into them
  static_assert(2 * SB <= ATB_RING_BYTES, "two slots per wave");
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int li = lane & 15, lk = lane >> 4;
  const int nbj = N / BN, nblk = (M / BM) * nbj;
  const int p = direct ? 0 : (t & 7) + 8 * ((t >> 3) / nblk), blk = direct ? t : (t >> 3) % nblk;
  if (p >= P || blk >= nblk) return;
  const int m0 = (blk / nbj) * BM, n0 = (blk % nbj) * BN;
  const int r0 = p * rpw, r1 = min(R, r0 + rpw);
  const int ngroups = (r1 - r0 + ROWS - 1) / ROWS;
  const int n_my = (ngroups - wave + 3) >> 2;           // the 4 waves interleave groups of ROWS rows
  unsigned char* ring = atb_smem + wave * ATB_RING_BYTES;
  const unsigned ring_addr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)ring);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, (unsigned)((size_t)R * M * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B), 0, (unsigned)((size_t)R * N * 4), 0x00020000);
  // byte offset of every piece of the wave's NEXT group to load (lane's 16-byte element), advanced by 64 rows per group
  unsigned voA[NA], voB[NB];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int e = j * 64 + lane, row = e / (4 * TI), col = (e % (4 * TI)) * 4;
    voA[j] = (unsigned)(((size_t)(r0 + wave * ROWS + row) * M + m0 + col) * 4);
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int e = j * 64 + lane, row = e / (4 * TJ), col = (e % (4 * TJ)) * 4;
    voB[j] = (unsigned)(((size_t)(r0 + wave * ROWS + row) * N + n0 + col) * 4);
  }
  const unsigned strideA = 4u * ROWS * (unsigned)M * 4u, strideB = 4u * ROWS * (unsigned)N * 4u;

  f32x4 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int u = 0; u < TJ; ++u) acc[i][u] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // piece x (0 .. G - 1: the A pieces, then the B pieces) of the next group to load, into `slot`
  auto dma_piece = [&](int x, int slot) {
    const unsigned sa = ring_addr + (unsigned)slot * SB;
#pragma unroll
    for (int j = 0; j < NA; ++j)
      if (x == j) {
        lds_dma16_buf(rsA, voA[j], __builtin_amdgcn_readfirstlane(sa + j * 1024));
        voA[j] += strideA;
      }
#pragma unroll
    for (int j = 0; j < NB; ++j)
      if (x == NA + j) {
        lds_dma16_buf(rsB, voB[j], __builtin_amdgcn_readfirstlane(sa + A_BYTES + j * 1024));
        voB[j] += strideB;
      }
  };
  // fragment read y (0 .. 2 KS - 1: A of k-step y / 2 when even, B when odd) of `slot`
  auto read_frag = [&](int y, int slot, float (&a)[KS][TI], float (&b)[KS][TJ]) {
    const float* sA = (const float*)(ring + slot * SB);
    const float* sB = (const float*)(ring + slot * SB + A_BYTES);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (y == 2 * ks) lds_frag<TI>(sA + (4 * ks + lk) * BM, li, a[ks]);
      if (y == 2 * ks + 1) lds_frag<TJ>(sB + (4 * ks + lk) * BN, li, b[ks]);
    }
  };
  // one slot's MFMAs on (a, b); dealt into them: the refill of `slot` (do_dma) and the next slot's fragments into (a2, b2)
  auto block = [&](float (&a)[KS][TI], float (&b)[KS][TJ], float (&a2)[KS][TI], float (&b2)[KS][TJ], int slot,
                   bool do_dma, bool do_read) {
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      const int ks = m / (TI * TJ), u = (m / TI) % TJ, i = m % TI;
      acc[i][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks][i], b[ks][u], acc[i][u], 0, 0, 0);
      // extras [m EXTRAS / NM, (m + 1) EXTRAS / NM) follow MFMA m: evenly dealt, whatever the ratio (the 16-wide tiles have
      // more extras than MFMAs)
#pragma unroll
      for (int e = 0; e < (EXTRAS + NM - 1) / NM; ++e) {      // (constant trip count: the MFMA loop unrolls completely)
        const int x = (m * EXTRAS) / NM + e;
        if (x >= ((m + 1) * EXTRAS) / NM) continue;
        if (x < G) {
          if (do_dma) {
            if (x == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every fragment of `slot` is in registers
            dma_piece(x, slot);
          }
        } else if (do_read) {
          read_frag(x - G, slot ^ 1, a2, b2);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  if (n_my > 0) {
    float a0[KS][TI], b0[KS][TJ], a1[KS][TI], b1[KS][TJ];
#pragma unroll
    for (int x = 0; x < G; ++x) dma_piece(x, 0);
    if (n_my > 1) {
#pragma unroll
      for (int x = 0; x < G; ++x) dma_piece(x, 1);
      wait_vmcnt<G>();                                   // group 0 has landed
    } else {
      wait_vmcnt<0>();
    }
#pragma unroll
    for (int y = 0; y < 2 * KS; ++y) read_frag(y, 0, a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    for (int it = 0; it < n_my; it += 2) {
      // block `it` on (a0, b0) from slot 0; block `it + 1` on (a1, b1) from slot 1
      wait_vmcnt<0>();                                   // slot 1 (group it + 1), issued a block ago, has landed
      block(a0, b0, a1, b1, 0, it + 2 < n_my, it + 1 < n_my);
      if (it + 1 < n_my) {
        wait_vmcnt<0>();                                 // slot 0 (group it + 2)
        block(a1, b1, a0, b0, 1, it + 3 < n_my, it + 2 < n_my);
      }
    }
  }
  atb_combine_store<TI, TJ>(acc, part + (size_t)p * M * N, ldp, m0, n0);   // (direct: p = 0)
}

__global__ __launch_bounds__(256, 2) void atb_grouped_kernel(const AtbGroup g) {
  const int L = blockIdx.x;
  // problem of this workgroup = number of table entries <= L, minus one (ascending, unused entries INT_MAX): the whole
  // table comes in with three wide scalar loads, no dependent load per step
  int i = -1;
#pragma unroll
  for (int j = 0; j < ATB_GROUP_MAX; ++j) i += (g.task0[j] <= L) ? 1 : 0;
  i = __builtin_amdgcn_readfirstlane(i);
  const AtbTask& k = g.t[i];
  const int t = L - g.task0[i];
#define D3F_ATB_TILE(I, J) \
  case (I) * 16 + (J): atb_task_body<I, J>(k.A, k.B, k.part, k.R, k.M, k.N, k.rpw, k.P, t, k.ldp, k.direct & 1); break
#define D3F_ATB_PIPE(I, J) \
  case 256 + (I) * 16 + (J): atb_task_body_pipe<I, J>(k.A, k.B, k.part, k.R, k.M, k.N, k.rpw, k.P, t, k.ldp, k.direct & 1); break
  switch (k.tile + ((k.direct & 2) ? 256 : 0)) {      // (direct bit 1: the software-pipelined body)
    D3F_ATB_TILE(1, 1); D3F_ATB_TILE(1, 2); D3F_ATB_TILE(1, 4);
    D3F_ATB_TILE(2, 1); D3F_ATB_TILE(2, 2); D3F_ATB_TILE(2, 4);
    D3F_ATB_TILE(4, 1); D3F_ATB_TILE(4, 2); D3F_ATB_TILE(4, 4);
    D3F_ATB_PIPE(1, 1); D3F_ATB_PIPE(1, 2); D3F_ATB_PIPE(1, 4);
    D3F_ATB_PIPE(2, 1); D3F_ATB_PIPE(2, 2); D3F_ATB_PIPE(2, 4);
    D3F_ATB_PIPE(4, 1); D3F_ATB_PIPE(4, 2); D3F_ATB_PIPE(4, 4);
    default: break;
  }
#undef D3F_ATB_TILE
#undef D3F_ATB_PIPE
}

// Second stage of a grouped launch.  Problem i owns blocks [block0[i], block0[i + 1]): its first c_blocks blocks sum the
// slabs -- 256 float4 elements x 4 slab lanes per workgroup, lane s sums slabs s, s + 4, ... in two chains, the four
// lanes combined in a fixed order --, the rest finish the bias gradient gb[c] = sum_b bpart[b][c] (16 row lanes x 4
// chains, fixed-order LDS combine: what bias_sum_kernel of elementwise.hip does, without a launch of its own).
struct AtbReduceTask {
  const float* part;   // [P][M N]
  float* C;            // [M, ldc]
  const float* bpart;  // [nblocks, BC] or null
  float* gb;
  float* gb2;
  int P, MN4, N, ldc;  // MN4 = M N / 4; P = 0: no slabs to sum (a direct problem, present for its bias gradient)
  int nblocks, BC_vec; // BC_vec = 2 BC + (C rows are 16-byte aligned)
};
static_assert(sizeof(AtbReduceTask) == 64, "table entry");
struct AtbReduceGroup {
  int n, pad;
  int block0[ATB_GROUP_MAX];
  AtbReduceTask t[ATB_GROUP_MAX];
};
static_assert(sizeof(AtbGroup) <= 4096 && sizeof(AtbReduceGroup) <= 4096, "kernel arguments are limited to 4 KiB");

__global__ __launch_bounds__(1024) void atb_grouped_reduce_kernel(const AtbReduceGroup g) {
  __shared__ float4 sh4[4][256];
  const int L = blockIdx.x;
  int i = -1;
#pragma unroll
  for (int j = 0; j < ATB_GROUP_MAX; ++j) i += (g.block0[j] <= L) ? 1 : 0;
  i = __builtin_amdgcn_readfirstlane(i);
  const AtbReduceTask& k = g.t[i];
  const int lb = L - g.block0[i];
  const int c_blocks = k.P > 0 ? (k.MN4 + 255) / 256 : 0;
  if (lb < c_blocks) {
    const int el = threadIdx.x & 255, s4 = threadIdx.x >> 8;
    const int e4 = lb * 256 + el;
    const float4* part4 = (const float4*)k.part;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    if (e4 < k.MN4) {
      int p = s4;
      for (; p + 4 < k.P; p += 8) {
        const float4 u = part4[(size_t)p * k.MN4 + e4], w = part4[(size_t)(p + 4) * k.MN4 + e4];
        s0.x += u.x; s0.y += u.y; s0.z += u.z; s0.w += u.w;
        s1.x += w.x; s1.y += w.y; s1.z += w.z; s1.w += w.w;
      }
      if (p < k.P) {
        const float4 u = part4[(size_t)p * k.MN4 + e4];
        s0.x += u.x; s0.y += u.y; s0.z += u.z; s0.w += u.w;
      }
    }
    sh4[s4][el] = make_float4(s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w);
    __syncthreads();
    if (s4 == 0 && e4 < k.MN4) {
      const float4 a = sh4[0][el], b = sh4[1][el], c = sh4[2][el], d = sh4[3][el];
      const float4 v = make_float4(((a.x + b.x) + c.x) + d.x, ((a.y + b.y) + c.y) + d.y, ((a.z + b.z) + c.z) + d.z,
                                   ((a.w + b.w) + c.w) + d.w);
      const int e = 4 * e4, row = e / k.N, col = e % k.N;   // (N is a multiple of 16: the four stay in one row)
      float* dst = k.C + (size_t)row * k.ldc + col;
      if (k.BC_vec & 1) {
        *(float4*)dst = v;
      } else {
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
      }
    }
    return;
  }
  float* sh = (float*)&sh4[0][0];          // [16][64]
  const int col = threadIdx.x & 63, sub = threadIdx.x >> 6;
  const int c = (lb - c_blocks) * 64 + col;
  const float* bpart = k.bpart;
  const int BC = k.BC_vec >> 1, nblocks = k.nblocks;
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
  if (c < BC) {
    int b = sub;
    for (; b + 48 < nblocks; b += 64) {
      s0 += bpart[(size_t)b * BC + c];
      s1 += bpart[(size_t)(b + 16) * BC + c];
      s2 += bpart[(size_t)(b + 32) * BC + c];
      s3 += bpart[(size_t)(b + 48) * BC + c];
    }
    for (; b < nblocks; b += 16) s0 += bpart[(size_t)b * BC + c];
  }
  sh[sub * 64 + col] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sub == 0 && c < BC) {
    float v = sh[col];
#pragma unroll
    for (int j = 1; j < 16; ++j) v += sh[j * 64 + col];
    k.gb[c] = v;
    if (k.gb2) k.gb2[c] = v;
  }
}

// second stage of the weight gradient.  Blocks [0, c_blocks): C[e] = sum_p part[p][e] (as atb_reduce_kernel<16>);
// blocks [c_blocks, ..): the bias gradient's column sums gb[c] = sum_b bpart[b][c] over the per-block partials the
// epilogue's backward kernel left behind -- the launch `bias_sum_kernel` (elementwise.hip) used to be, summed in the same
// order (16 row lanes x 4 chains, fixed-order LDS combine): one launch per layer instead of two.
__global__ __launch_bounds__(1024) void atb_reduce_bias_kernel(const float* __restrict__ part, int P, size_t MN,
                                                               float* __restrict__ C, size_t MN_out, int c_blocks,
                                                               const float* __restrict__ bpart, int nblocks, int BC,
                                                               float* __restrict__ gb, float* __restrict__ gb2,
                                                               int fan16) {
  __shared__ float sh[16][64];
  const int col = threadIdx.x & 63, sub = threadIdx.x >> 6;
  if ((int)blockIdx.x < c_blocks && !fan16) {
    // few slabs (the usual case: 8..32 partitions): 256 elements x 4 row lanes per workgroup, every lane sums a quarter
    // of the slabs in two chains -- 16 lanes per element spent their time in the LDS combine (7.0 us a launch against
    // 4.7 for the plain 4-lane reduce)
    float* sh4 = &sh[0][0];                  // [4][256]
    const int el = threadIdx.x & 255, s4 = threadIdx.x >> 8;
    const size_t e = (size_t)blockIdx.x * 256 + el;
    float s0 = 0.f, s1 = 0.f;
    if (e < MN) {
      int p = s4;
      for (; p + 4 < P; p += 8) {
        s0 += part[(size_t)p * MN + e];
        s1 += part[(size_t)(p + 4) * MN + e];
      }
      if (p < P) s0 += part[(size_t)p * MN + e];
    }
    sh4[s4 * 256 + el] = s0 + s1;
    __syncthreads();
    if (s4 == 0 && e < MN_out) C[e] = ((sh4[el] + sh4[256 + el]) + sh4[512 + el]) + sh4[768 + el];
    return;
  }
  if ((int)blockIdx.x < c_blocks) {
    const size_t e = (size_t)blockIdx.x * 64 + col;
    float s0 = 0.f, s1 = 0.f;
    if (e < MN) {
      int p = sub;
      for (; p + 16 < P; p += 32) {
        s0 += part[(size_t)p * MN + e];
        s1 += part[(size_t)(p + 16) * MN + e];
      }
      if (p < P) s0 += part[(size_t)p * MN + e];
    }
    sh[sub][col] = s0 + s1;
    __syncthreads();
    if (sub == 0 && e < MN_out) {
      float s = sh[0][col];
#pragma unroll
      for (int k = 1; k < 16; ++k) s += sh[k][col];
      C[e] = s;
    }
    return;
  }
  const int c = ((int)blockIdx.x - c_blocks) * 64 + col;
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
  if (c < BC) {
    int b = sub;
    for (; b + 48 < nblocks; b += 64) {
      s0 += bpart[(size_t)b * BC + c];
      s1 += bpart[(size_t)(b + 16) * BC + c];
      s2 += bpart[(size_t)(b + 32) * BC + c];
      s3 += bpart[(size_t)(b + 48) * BC + c];
    }
    for (; b < nblocks; b += 16) s0 += bpart[(size_t)b * BC + c];
  }
  sh[sub][col] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sub == 0 && c < BC) {
    float v = sh[0][col];
#pragma unroll
    for (int j = 1; j < 16; ++j) v += sh[j][col];
    gb[c] = v;
    if (gb2) gb2[c] = v;
  }
}

static inline int tile_width(int n) { return n % 64 == 0 ? 4 : (n % 32 == 0 ? 2 : (n % 16 == 0 ? 1 : 0)); }

bool atb_supported(int R, int M, int N) { return R >= 1 && tile_width(M) && tile_width(N); }

// ---- first form: number of row partitions = workgroups along the reduction ------------------------------------------
static int atb_partitions(int R, int M, int N) {
  const int forced = tunables().atb_first_form_wgs;
  const long long nblocks = (long long)(M / (16 * tile_width(M))) * (N / (16 * tile_width(N)));
  // one workgroup per CU is enough unless the operands are long AND wide (profiles/atb_microbench.py: 512 partitions'
  // worth of slabs cost the small outputs 3 us each in the reduce pass; the 38k x 480 x 32 KPConv gradient wants them)
  int target = forced > 0 ? forced : ((R >= 30000 && nblocks >= 8) ? 512 : 256);
  // the counts above were measured on one S1-class pair (<= 38k rows); several pairs stacked into one batch (round 4)
  // multiply the rows: keep the ROWS per workgroup where they were instead of the workgroup count
  if (forced <= 0 && R > 40000) target = (int)((long long)target * R / 40000);
  long long wgs = (target + nblocks - 1) / nblocks;  // workgroups over the whole launch (256 CUs)
  const long long max_by_rows = (R + 63) / 64;            // >= 16 rows (4 MFMA k-steps) per wave
  if (wgs > max_by_rows) wgs = max_by_rows;
  if (wgs > 512) wgs = 512;
  if (wgs < 1) wgs = 1;
  return (int)wgs;
}

// ---- grouped form: partition plan of one problem ----------------------------------------------------------------------
// A task (one workgroup: one row partition x one output block) should last about atb_task_us on one of the chip's 512
// workgroup slots (2 per CU).  Its duration is modelled as the larger of its matrix time (f32 MFMA peak / 512 per slot)
// and its share of the HBM stream (the A panel of a partition is fetched once for the nbj blocks that read it -- they
// run on one XCD --, the B panel once for nbi; 6.3 TB/s / 512 per slot).
struct AtbPlan {
  int ti, tj, nbi, nbj, P, rpw, direct;
  long long ntasks;         // 8 ceil(P / 8) nblk (direct: nblk rounded up to 8)
  double task_s;            // modelled duration of one task
  size_t slab_floats;       // P M N (direct: 0)
};
// can_direct: the target admits 16-byte row stores (the kernel then writes C itself when the reduction is not split)
static AtbPlan atb_plan(int R, int M, int N, bool can_direct = false) {
  AtbPlan a;
  a.ti = tile_width(M);
  a.tj = tile_width(N);
  const int BM = 16 * a.ti, BN = 16 * a.tj;
  a.nbi = M / BM;
  a.nbj = N / BN;
  const double f_slot = 157.3e12 / 512, w_slot = 6.3e12 / 512;
  const double row_s = fmax(2.0 * BM * BN / f_slot, 4.0 * ((double)BM / a.nbj + (double)BN / a.nbi) / w_slot);
  int us = tunables().atb_task_us;
  if (us < 1) us = 40;      // (profiles/atb_group_bench.py --step: 862 / 845 / 837 us at 20 / 40 / 80 -- fewer slabs)
  double rows = us * 1e-6 / row_s;
  if (rows < 256) rows = 256;
  a.direct = 0;
  // Few rows against a large output (weight gradients of the bottom levels: 462 / 1713 rows x 7680 x 512 ...): the
  // output blocks alone fill the chip, splitting the reduction would only add slab traffic (8 x 15.7 MB for one of
  // them) -- one task per output block over ALL rows, result written straight to C.
  if (can_direct && (long long)a.nbi * a.nbj >= 128 && (double)R <= 2.5 * rows) {
    a.direct = 1;
    a.P = 1;
    a.rpw = (R + 63) / 64 * 64;
    a.ntasks = ((long long)a.nbi * a.nbj + 7) / 8 * 8;
    a.task_s = row_s * (double)R;
    a.slab_floats = 0;
    return a;
  }
  long long q = (long long)((double)R / (8.0 * rows) + 0.5);
  if (q < 1) q = 1;
  if (q > 64) q = 64;
  long long rpw = ((long long)R + 8 * q - 1) / (8 * q);
  rpw = (rpw + 63) / 64 * 64;                           // whole groups of 16 rows for every wave
  a.rpw = (int)rpw;
  a.P = (int)(((long long)R + rpw - 1) / rpw);
  a.ntasks = (long long)((a.P + 7) / 8) * 8 * a.nbi * a.nbj;
  a.task_s = row_s * (double)rpw;
  a.slab_floats = (size_t)a.P * M * N;
  return a;
}

struct AtbProblem {   // host side of one problem: C [M, ldc] = A^T [M, R] B [R, N] (+ the bias gradient of the same block)
  const float* A;
  const float* B;
  float* C;
  int R, M, N, ldc;
  const float* bias_part;
  int bias_blocks, bias_cols;
  float* grad_bias;
  float* grad_bias2;
};

static bool atb_can_direct(const AtbProblem& p) { return (((uintptr_t)p.C) & 15) == 0 && (p.ldc & 3) == 0; }
static bool atb_problem_ok(const AtbProblem& p) {
  if (!p.A || !p.B || !p.C || !atb_supported(p.R, p.M, p.N) || p.ldc < p.N) return false;
  if ((((uintptr_t)p.A | (uintptr_t)p.B) & 15) != 0 || (((uintptr_t)p.C) & 3) != 0) return false;   // (LDS-DMA moves 16 B)
  if (p.bias_part && (p.bias_blocks < 1 || p.bias_cols < 1 || !p.grad_bias)) return false;
  return true;
}

size_t atb_group_ws_bytes(const AtbProblem* probs, int n) {
  size_t total = 256;     // (never 0 for a valid queue: 0 is the C ABI's "unsupported problem")
  for (int i = 0; i < n; ++i) {
    if (!atb_supported(probs[i].R, probs[i].M, probs[i].N)) return 0;
    total += align_up(sizeof(float) * atb_plan(probs[i].R, probs[i].M, probs[i].N, atb_can_direct(probs[i])).slab_floats,
                      256);
  }
  return total;
}

// all problems of `probs` in ceil(n / ATB_GROUP_MAX) x 2 launches; ws >= atb_group_ws_bytes
int atb_group_launch(const AtbProblem* probs, int n, void* ws, hipStream_t stream) {
  if (n < 1) return D3F_OK;
  for (int i = 0; i < n; ++i)
    if (!atb_problem_ok(probs[i])) return D3F_EINVAL;
  // measurement aid of bench.py: ONE record for both launches of a group; shape = {problems, MiFLOP, KiB, tasks, 0, 0}
  double flops = 0.0, bytes = 0.0;
  long long all_tasks = 0;
  for (int i = 0; i < n; ++i) {
    flops += 2.0 * probs[i].R * (double)probs[i].M * probs[i].N;
    bytes += 4.0 * probs[i].R * ((double)probs[i].M + probs[i].N) + 4.0 * (double)probs[i].M * probs[i].N;
    all_tasks += atb_plan(probs[i].R, probs[i].M, probs[i].N, atb_can_direct(probs[i])).ntasks;
  }
  void* timing = kpconv_timing_open(7, stream, n, (int)(flops / 1048576.0), (int)(bytes / 1024.0), (int)all_tasks, 0, 0);
  char* wsp = (char*)ws;
  for (int c0 = 0; c0 < n; c0 += ATB_GROUP_MAX) {
    const int m = (n - c0 < ATB_GROUP_MAX) ? n - c0 : ATB_GROUP_MAX;
    AtbPlan plan[ATB_GROUP_MAX];
    int order[ATB_GROUP_MAX];
    float* part[ATB_GROUP_MAX];
    for (int i = 0; i < m; ++i) {
      const AtbProblem& p = probs[c0 + i];
      plan[i] = atb_plan(p.R, p.M, p.N, atb_can_direct(p));
      part[i] = plan[i].direct ? p.C : (float*)wsp;
      wsp += align_up(sizeof(float) * plan[i].slab_floats, 256);
      order[i] = i;
    }
    // longest task first (insertion sort, stable: equal problems keep their queue order -- the plan is deterministic)
    for (int i = 1; i < m; ++i) {
      const int o = order[i];
      int j = i;
      while (j > 0 && plan[order[j - 1]].task_s < plan[o].task_s) {
        order[j] = order[j - 1];
        --j;
      }
      order[j] = o;
    }
    AtbGroup g;
    AtbReduceGroup rg;
    g.n = rg.n = m;
    g.pad = rg.pad = 0;
    long long task = 0, block = 0;
    for (int s = 0; s < m; ++s) {
      const int i = order[s];
      const AtbProblem& p = probs[c0 + i];
      g.task0[s] = (int)task;
      g.t[s].A = p.A;
      g.t[s].B = p.B;
      g.t[s].part = part[i];
      g.t[s].R = p.R;
      g.t[s].M = p.M;
      g.t[s].N = p.N;
      g.t[s].rpw = plan[i].rpw;
      g.t[s].P = plan[i].P;
      g.t[s].tile = plan[i].ti * 16 + plan[i].tj;
      g.t[s].ldp = plan[i].direct ? p.ldc : p.N;
      // the software-pipelined body: operands a 32-bit buffer offset reaches (tunables().atb_pipe = 1: never; n >= 2:
      // only tiles of at least n 16 x 16 accumulators -- A/B measurements)
      const bool wide = plan[i].ti * plan[i].tj >= (tunables().atb_pipe >= 2 ? tunables().atb_pipe : 1);
      const bool fits = ((size_t)p.R + 256) * p.M * 4 < 0xffffffffull && ((size_t)p.R + 256) * p.N * 4 < 0xffffffffull;
      g.t[s].direct = plan[i].direct | ((wide && fits && tunables().atb_pipe != 1) ? 2 : 0);
      task += plan[i].ntasks;
      AtbReduceTask& r = rg.t[s];
      rg.block0[s] = (int)block;
      r.part = part[i];
      r.C = p.C;
      r.bpart = p.bias_part;
      r.gb = p.grad_bias;
      r.gb2 = p.grad_bias2;
      r.P = plan[i].direct ? 0 : plan[i].P;
      r.MN4 = (int)((size_t)p.M * p.N / 4);
      r.N = p.N;
      r.ldc = p.ldc;
      r.nblocks = p.bias_part ? p.bias_blocks : 0;
      r.BC_vec = 2 * (p.bias_part ? p.bias_cols : 0) + (atb_can_direct(p) ? 1 : 0);
      block += (plan[i].direct ? 0 : cdiv(r.MN4, 256)) + (p.bias_part ? cdiv(p.bias_cols, 64) : 0);
    }
    for (int s = m; s < ATB_GROUP_MAX; ++s) g.task0[s] = rg.block0[s] = 0x7fffffff;
    if (task >= 0x7fffffffLL || block >= 0x7fffffffLL) return D3F_EINVAL;
    atb_grouped_kernel<<<(unsigned)task, 256, ATB_LDS_BYTES, stream>>>(g);
    D3F_LAUNCH_CHECK();
    if (block > 0) {        // (nothing to sum when every problem of the launch wrote its result directly)
      atb_grouped_reduce_kernel<<<(unsigned)block, 1024, 0, stream>>>(rg);
      D3F_LAUNCH_CHECK();
    }
  }
  kpconv_timing_close(timing, stream);
  return D3F_OK;
}

// Which form a SINGLE problem runs (the C-ABI's one-problem entry points and the KPConv kernels' own weight gradient):
// the grouped kernels where there is arithmetic to pipeline (from 0.7 GFLOP, profiles/r05_atb_sweep.txt), the first
// form -- whose workgroups start faster and which applies a row divisor -- on the small launches.
// tunables().atb_form = 1 / 2: always the first form / the grouped kernels.
static bool atb_grouped_wanted(const float* A, const float* B, const float* row_div, int R, int M, int N, int M_out) {
  const int v = tunables().atb_form;
  if (v == 1 || row_div || (M_out > 0 && M_out < M)) return false;
  if ((((uintptr_t)A | (uintptr_t)B) & 15) != 0) return false;
  if (v == 2) return true;
  return 2.0 * R * (double)M * N >= 0.7e9;
}

size_t atb_ws_bytes(int R, int M, int N) {
  if (!atb_supported(R, M, N)) return 0;
  const size_t first = align_up(sizeof(float) * (size_t)atb_partitions(R, M, N) * M * N, 256);
  const size_t grouped = align_up(sizeof(float) * atb_plan(R, M, N).slab_floats, 256);
  return first > grouped ? first : grouped;
}

// C [M,N] = A^T [M,R] (B [R,N] / row_div [R]); ws >= atb_ws_bytes.  bias_*: the second stage also finishes a bias
// gradient from `bias_blocks` rows of partial column sums [bias_blocks, bias_cols] (see atb_reduce_bias_kernel).
int atb_splitk_bias(const float* A, const float* B, const float* row_div, int R, int M, int N, float* C, void* ws,
                    hipStream_t stream, int M_out, const float* bias_part, int bias_blocks, int bias_cols,
                    float* grad_bias, float* grad_bias2) {
  if (!atb_supported(R, M, N)) return D3F_EINVAL;
  if (bias_part && (bias_blocks < 1 || bias_cols < 1 || !grad_bias)) return D3F_EINVAL;
  if (atb_grouped_wanted(A, B, row_div, R, M, N, M_out)) {
    const AtbProblem p = {A, B, C, R, M, N, N, bias_part, bias_blocks, bias_cols, grad_bias, grad_bias2};
    return atb_group_launch(&p, 1, ws, stream);
  }
  float* part = (float*)ws;
  void* timing = kpconv_timing_open(4, stream, R, 0, 0, M, N, 0);   // (both launches: partial sums + their reduction)
  const int ti = tile_width(M), tj = tile_width(N);
  const int P = atb_partitions(R, M, N);
  int rpw = (R + P - 1) / P;
  rpw = (rpw + 3) / 4 * 4;
  dim3 grid(P, (M / (16 * ti)) * (N / (16 * tj)));
  // U = k-steps whose loads are issued together: the measured default per tile shape (2 for >= 8 accumulator tiles)
#define D3F_ATB(I, J) \
  atb_partial_kernel<I, J, ((I) * (J) >= 8) ? 2 : 4><<<grid, 256, 0, stream>>>(A, B, row_div, R, M, N, rpw, part)
  switch (ti * 8 + tj) {
    case 1 * 8 + 1: D3F_ATB(1, 1); break;
    case 1 * 8 + 2: D3F_ATB(1, 2); break;
    case 1 * 8 + 4: D3F_ATB(1, 4); break;
    case 2 * 8 + 1: D3F_ATB(2, 1); break;
    case 2 * 8 + 2: D3F_ATB(2, 2); break;
    case 2 * 8 + 4: D3F_ATB(2, 4); break;
    case 4 * 8 + 1: D3F_ATB(4, 1); break;
    case 4 * 8 + 2: D3F_ATB(4, 2); break;
    default: D3F_ATB(4, 4); break;
  }
#undef D3F_ATB
  D3F_LAUNCH_CHECK();
  const size_t MN = (size_t)M * N;
  const size_t MN_out = (M_out > 0 && M_out < M) ? (size_t)M_out * N : MN;
  if (bias_part) {
    const int fan16 = (P >= 64 && MN <= 65536) ? 1 : 0;
    const int cb = cdiv((long long)MN, fan16 ? 64 : 256);
    atb_reduce_bias_kernel<<<cb + cdiv(bias_cols, 64), 1024, 0, stream>>>(part, P, MN, C, MN_out, cb, bias_part,
                                                                         bias_blocks, bias_cols, grad_bias, grad_bias2,
                                                                         fan16);
  } else if (P >= 64 && MN <= 65536) {
    atb_reduce_kernel<16><<<cdiv((long long)MN, 64), 1024, 0, stream>>>(part, P, MN, C, MN_out);
  } else {
    atb_reduce_kernel<4><<<cdiv((long long)MN, 64), 256, 0, stream>>>(part, P, MN, C, MN_out);
  }
  kpconv_timing_close(timing, stream);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

int atb_splitk(const float* A, const float* B, const float* row_div, int R, int M, int N, float* C, void* ws,
               hipStream_t stream, int M_out = 0) {
  return atb_splitk_bias(A, B, row_div, R, M, N, C, ws, stream, M_out, nullptr, 0, 0, nullptr, nullptr);
}

// ------------------------------------------------------------------------------------------------------------------
// Y [N,M] = epilogue( X [N,K] . B [K,M] ) for MANY rows and a SMALL weight matrix (unary blocks of the upper pyramid
// levels: N = 8k..38k points, K, M <= 256).  A library GEMM is launch/latency bound there (6-20 us for 15-30 MB of
// streaming) and needs a separate bias/activation pass.  Here a wave owns 16*RT rows and ALL M columns:
//   A fragments = one float4 of the row per 16 reduction indices (x is streamed from HBM exactly once),
//   B fragments straight from the weight matrix in L2 (WT: B[k][m] = W[m][k], the forward of nn.Linear -- 4 reduction
//   indices are one contiguous float4 of W's row m; !WT: B = W as stored, grad_x = g W),
//   RT*M/16 accumulators in registers, and the epilogue out = act(acc + b1 + add + b2) is applied before the only store.
template <int MBW, int CS, bool WT, bool EPI>
__global__ __launch_bounds__(256) void rowgemm_kernel(const float* __restrict__ X, const float* __restrict__ W, int N,
                                                      int K, const float* __restrict__ b1,
                                                      const float* __restrict__ add, const float* __restrict__ b2,
                                                      float slope, float* __restrict__ Y, float* __restrict__ zinit,
                                                      int zn) {
  // wave w of the workgroup: row tile w / CS (16 rows), column group w % CS (16*MBW columns); M = 16*MBW*CS.
  // Splitting the columns over waves keeps >= 8 workgroups per CU in flight at 38k rows (one wave per 16 rows and all
  // columns left the chip at ~1 wave per SIMD and was slower than the library GEMM).
  constexpr int M = 16 * MBW * CS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  if (zinit && blockIdx.x == 0)
    for (int t = threadIdx.x; t < zn; t += blockDim.x) zinit[t] = 0.0f;
  const int row0 = (blockIdx.x * (4 / CS) + wave / CS) * 16;
  const int c0 = (wave % CS) * 16 * MBW;
  if (row0 >= N) return;
  f32x4 acc[MBW];
#pragma unroll
  for (int nb = 0; nb < MBW; ++nb) acc[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float* xr = X + (size_t)min(row0 + li, N - 1) * K + 4 * lk;
  // column of fragment nb of lane-column li: c0 + MBW li + nb -- a lane holds MBW CONSECUTIVE columns of its 4 rows, so
  // the epilogue reads `add` and writes Y with one MBW-wide vector per row (16 lanes = one contiguous 64 MBW-byte run;
  // with the columns dealt nb-major every access was a 4-byte one in 64-byte runs and the memory-bound layers -- 114k
  // rows, 32 -> 128 channels + residual -- streamed at 2.2 TB/s)
  for (int k0 = 0; k0 < K; k0 += 16) {
    const float4 a = *(const float4*)(xr + k0);
    float bq[MBW][4];
    if (WT) {
#pragma unroll
      for (int nb = 0; nb < MBW; ++nb) {
        const float4 b = *(const float4*)(W + (size_t)(c0 + MBW * li + nb) * K + k0 + 4 * lk);
        bq[nb][0] = b.x; bq[nb][1] = b.y; bq[nb][2] = b.z; bq[nb][3] = b.w;
      }
    } else {   // B = W as stored [K, M]: the lane's MBW columns of reduction row k0 + 4 lk + t are one vector
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const typename VecT<MBW>::type wv =
            *(const typename VecT<MBW>::type*)(W + (size_t)(k0 + 4 * lk + t) * M + c0 + MBW * li);
#pragma unroll
        for (int nb = 0; nb < MBW; ++nb) bq[nb][t] = vget<MBW>(wv, nb);
      }
    }
#pragma unroll
    for (int nb = 0; nb < MBW; ++nb) {
      acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bq[nb][0], acc[nb], 0, 0, 0);
      acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bq[nb][1], acc[nb], 0, 0, 0);
      acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bq[nb][2], acc[nb], 0, 0, 0);
      acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bq[nb][3], acc[nb], 0, 0, 0);
    }
  }
  // D[i][j]: row = row0 + 4 lk + r, columns = cbase .. cbase + MBW - 1
  typedef typename VecT<MBW>::type VO;
  const int cbase = c0 + MBW * li;
  float bias1[MBW], bias2[MBW];
#pragma unroll
  for (int nb = 0; nb < MBW; ++nb) {
    bias1[nb] = (EPI && b1) ? b1[cbase + nb] : 0.0f;
    bias2[nb] = (EPI && b2) ? b2[cbase + nb] : 0.0f;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = row0 + 4 * lk + r;
    if (row < N) {
      float v[MBW];
#pragma unroll
      for (int nb = 0; nb < MBW; ++nb) v[nb] = acc[nb][r];
      if (EPI) {
        if (b1) {
#pragma unroll
          for (int nb = 0; nb < MBW; ++nb) v[nb] += bias1[nb];
        }
        if (add) {
          const VO av = *(const VO*)(add + (size_t)row * M + cbase);
#pragma unroll
          for (int nb = 0; nb < MBW; ++nb) v[nb] += vget<MBW>(av, nb);
        }
        if (b2) {
#pragma unroll
          for (int nb = 0; nb < MBW; ++nb) v[nb] += bias2[nb];
        }
#pragma unroll
        for (int nb = 0; nb < MBW; ++nb) v[nb] = v[nb] > 0.0f ? v[nb] : v[nb] * slope;
      }
      float* dst = Y + (size_t)row * M + cbase;
      if constexpr (MBW == 4) *(float4*)dst = make_float4(v[0], v[MBW > 1 ? 1 : 0], v[MBW > 2 ? 2 : 0], v[MBW > 3 ? 3 : 0]);
      else if constexpr (MBW == 2) *(float2*)dst = make_float2(v[0], v[MBW > 1 ? 1 : 0]);
      else dst[0] = v[0];
    }
  }
}

// The FORWARD product for the many-row layers with a short reduction (K = 16 KT <= 64): the weight fragments of the wave's
// column group stay in REGISTERS (KT x MBW x 4 values per lane) while the wave walks two consecutive 16-row tiles, and the
// second tile's x fragments and `add` rows are requested before the first tile's MFMAs.  rowgemm_kernel re-reads its
// weights from L2 for every 16 rows with the loads right in front of the MFMAs that need them: bound by L2 latency per
// k-step, not by HBM -- 114624 x 32 -> 128 + residual 32.7 -> 25.8 us (5.1 TB/s), 114624 x 64 -> 32 18.8 -> 15.2,
// 23808 x 64 -> 256 + residual 23.9 -> 18.7 (profiles/rowgemm_bench.py; four tiles per wave: the same; the grad-input
// products, with their longer reductions, gain nothing and keep rowgemm_kernel).
// KT2 > 0: a SECOND product into the same output, Y = act(X W^T + X2 W2^T + b1 + b2 + b3 + b4) -- the last unary block
// of a bottleneck and its shortcut unary (reference models/blocks.py:658-686) as one launch: the [N, M] shortcut tensor is
// never written and read back.
template <int MBW, int CS, int KT, int KT2>
__global__ __launch_bounds__(256) void rowgemm_rt_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                         const float* __restrict__ X2, const float* __restrict__ W2,
                                                         int N, const float* __restrict__ b1,
                                                         const float* __restrict__ add, const float* __restrict__ b2,
                                                         const float* __restrict__ b3, const float* __restrict__ b4,
                                                         float slope, float* __restrict__ Y, float* __restrict__ zinit,
                                                         int zn) {
  constexpr int M = 16 * MBW * CS, K = 16 * KT, K2 = 16 * KT2, RT = 2;
  typedef typename VecT<MBW>::type VO;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  if (zinit && blockIdx.x == 0)
    for (int t = threadIdx.x; t < zn; t += blockDim.x) zinit[t] = 0.0f;
  const int tile0 = (blockIdx.x * (4 / CS) + wave / CS) * RT;     // first of the wave's RT row tiles
  const int c0 = (wave % CS) * 16 * MBW;
  if (tile0 * 16 >= N) return;
  const int cbase = c0 + MBW * li;
  // weight fragments: wq[kt][nb][t] = W[cbase + nb][16 kt + 4 lk + t]
  float wq[KT + KT2][MBW][4];
#pragma unroll
  for (int kt = 0; kt < KT + KT2; ++kt)
#pragma unroll
    for (int nb = 0; nb < MBW; ++nb) {
      const float4 b = kt < KT ? *(const float4*)(W + (size_t)(cbase + nb) * K + 16 * kt + 4 * lk)
                               : *(const float4*)(W2 + (size_t)(cbase + nb) * K2 + 16 * (kt - KT) + 4 * lk);
      wq[kt][nb][0] = b.x; wq[kt][nb][1] = b.y; wq[kt][nb][2] = b.z; wq[kt][nb][3] = b.w;
    }
  float bias1[MBW], bias2[MBW];
#pragma unroll
  for (int nb = 0; nb < MBW; ++nb) {
    bias1[nb] = b1 ? b1[cbase + nb] : 0.0f;
    bias2[nb] = b2 ? b2[cbase + nb] : 0.0f;
    if (KT2 > 0) {   // (b1 + b2) + (b3 + b4): the separate launches add b1, then the shortcut (which carries b3 + b4), then b2
      bias1[nb] = (bias1[nb] + bias2[nb]) + ((b3 ? b3[cbase + nb] : 0.0f) + (b4 ? b4[cbase + nb] : 0.0f));
      bias2[nb] = 0.0f;
    }
  }
  auto load_x = [&](int tile, float4 (&a)[KT + KT2]) {
    const size_t row = (size_t)min(tile * 16 + li, N - 1);
#pragma unroll
    for (int kt = 0; kt < KT + KT2; ++kt)
      a[kt] = kt < KT ? *(const float4*)(X + row * K + 4 * lk + 16 * kt)
                      : *(const float4*)(X2 + row * K2 + 4 * lk + 16 * (kt - KT));
  };
  auto load_add = [&](int tile, VO (&av)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = min(tile * 16 + 4 * lk + r, N - 1);
      av[r] = *(const VO*)(add + (size_t)row * M + cbase);
    }
  };
  float4 a[KT + KT2], an[KT + KT2];
  VO av[4], avn[4];
  load_x(tile0, a);
  if (add) load_add(tile0, av);
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int tile = tile0 + rt;
    if (tile * 16 >= N) break;
    const bool more = rt + 1 < RT && (tile + 1) * 16 < N;
    if (more) {                      // the next tile's operands are on their way during this tile's MFMAs
      load_x(tile + 1, an);
      if (add) load_add(tile + 1, avn);
    }
    f32x4 acc[MBW];
#pragma unroll
    for (int nb = 0; nb < MBW; ++nb) acc[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < KT + KT2; ++kt)
#pragma unroll
      for (int nb = 0; nb < MBW; ++nb) {
        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kt].x, wq[kt][nb][0], acc[nb], 0, 0, 0);
        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kt].y, wq[kt][nb][1], acc[nb], 0, 0, 0);
        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kt].z, wq[kt][nb][2], acc[nb], 0, 0, 0);
        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kt].w, wq[kt][nb][3], acc[nb], 0, 0, 0);
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = tile * 16 + 4 * lk + r;
      if (row < N) {
        float v[MBW];
#pragma unroll
        for (int nb = 0; nb < MBW; ++nb) v[nb] = acc[nb][r];
        if (b1 || KT2 > 0) {
#pragma unroll
          for (int nb = 0; nb < MBW; ++nb) v[nb] += bias1[nb];
        }
        if (add) {
#pragma unroll
          for (int nb = 0; nb < MBW; ++nb) v[nb] += vget<MBW>(av[r], nb);
        }
        if (b2 && KT2 == 0) {
#pragma unroll
          for (int nb = 0; nb < MBW; ++nb) v[nb] += bias2[nb];
        }
#pragma unroll
        for (int nb = 0; nb < MBW; ++nb) v[nb] = v[nb] > 0.0f ? v[nb] : v[nb] * slope;
        float* dst = Y + (size_t)row * M + cbase;
        if constexpr (MBW == 4) *(float4*)dst = make_float4(v[0], v[MBW > 1 ? 1 : 0], v[MBW > 2 ? 2 : 0], v[MBW > 3 ? 3 : 0]);
        else if constexpr (MBW == 2) *(float2*)dst = make_float2(v[0], v[MBW > 1 ? 1 : 0]);
        else dst[0] = v[0];
      }
    }
    if (more) {
#pragma unroll
      for (int kt = 0; kt < KT + KT2; ++kt) a[kt] = an[kt];
#pragma unroll
      for (int r = 0; r < 4; ++r) av[r] = avn[r];
    }
  }
}

bool rowgemm_supported(int N, int K, int M) {
  return N >= 1 && K >= 16 && K % 16 == 0 && K <= 1024 && (M == 32 || M == 64 || M == 128 || M == 256);
}

template <bool WT, bool EPI>
static int rowgemm_launch(const float* X, const float* W, int N, int K, int M, const float* b1, const float* add,
                          const float* b2, float slope, float* Y, float* zinit, int zn, hipStream_t stream) {
#define D3F_RG(MBW, CS)                                                                                  \
  rowgemm_kernel<MBW, CS, WT, EPI><<<cdiv(N, 16 * (4 / CS)), 256, 0, stream>>>(X, W, N, K, b1, add, b2, slope, Y, \
                                                                               zinit, zn)
  // 64 / 128 outputs at the many-row level: a lane owns 4 consecutive columns (16-byte accesses of `add` / Y, half the
  // column groups per row tile) -- 114624 x 32 -> 128 + residual 37.8 -> 32.7 us, grad-input 128 -> 64 37.1 -> 28.2
  // (profiles/rowgemm_bench.py); with fewer rows the launch has too few workgroups and the narrow dealing wins.
  // tunables().rowgemm_wide: 0 = by rows, 1 = never, 2 = always.
  const int rw = tunables().rowgemm_wide;
  const bool wide = rw == 2 || (rw == 0 && N >= 65536);
  // forward with a short reduction: weights in registers over two row tiles per wave (rowgemm_rt_kernel).
  // tunables().rowgemm_rt: 0 = from 4096 rows, 1 = never
  const int kt = K / 16;
  if (WT && EPI && tunables().rowgemm_rt != 1 && N >= 4096 && (kt == 1 || kt == 2 || kt == 4)) {
#define D3F_RGT2(MBW, CS, KT) \
  rowgemm_rt_kernel<MBW, CS, KT, 0><<<cdiv(N, 32 * (4 / CS)), 256, 0, stream>>>(X, W, nullptr, nullptr, N, b1, add, b2, nullptr, nullptr, slope, Y, zinit, zn)
#define D3F_RGT(MBW, CS) \
  do { if (kt == 1) D3F_RGT2(MBW, CS, 1); else if (kt == 2) D3F_RGT2(MBW, CS, 2); else D3F_RGT2(MBW, CS, 4); } while (0)
    switch (M) {
      case 32: D3F_RGT(1, 2); break;
      case 64: if (wide) D3F_RGT(4, 1); else D3F_RGT(2, 2); break;
      case 128: if (wide) D3F_RGT(4, 2); else D3F_RGT(2, 4); break;
      case 256: D3F_RGT(4, 4); break;
      default: return D3F_EINVAL;
    }
#undef D3F_RGT
#undef D3F_RGT2
    D3F_LAUNCH_CHECK();
    return D3F_OK;
  }
  switch (M) {
    case 32: D3F_RG(1, 2); break;
    case 64: if (wide) D3F_RG(4, 1); else D3F_RG(2, 2); break;
    case 128: if (wide) D3F_RG(4, 2); else D3F_RG(2, 4); break;
    case 256: D3F_RG(4, 4); break;
    default: return D3F_EINVAL;
  }
#undef D3F_RG
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

// unary2 + shortcut unary of a bottleneck in one launch (rowgemm_rt_kernel with KT2 > 0): served while both weight
// matrices fit the registers -- the level-0 bottleneck (32 | 64 -> 128 channels) and its half-width kin
bool rowgemm_pair_supported(int N, int K1, int K2, int M) {
  return N >= 4096 && ((M == 128 && K1 == 32 && K2 == 64) || (M == 64 && K1 == 16 && K2 == 32));
}

static int rowgemm_pair_launch(const float* X1, const float* W1, int K1, const float* X2, const float* W2, int K2, int N,
                               int M, const float* b1, const float* b2, const float* b3, const float* b4, float slope,
                               float* Y, float* zinit, int zn, hipStream_t stream) {
  if (!rowgemm_pair_supported(N, K1, K2, M)) return D3F_EINVAL;
  const int rw = tunables().rowgemm_wide;
  const bool wide = rw == 2 || (rw == 0 && N >= 65536);
#define D3F_RGPT(MBW, CS, KA, KB)                                                                                      \
  rowgemm_rt_kernel<MBW, CS, KA, KB><<<cdiv(N, 32 * (4 / CS)), 256, 0, stream>>>(X1, W1, X2, W2, N, b1, nullptr, b2, b3, \
                                                                                  b4, slope, Y, zinit, zn)
  if (M == 128) {
    if (wide) D3F_RGPT(4, 2, 2, 4); else D3F_RGPT(2, 4, 2, 4);
  } else {
    if (wide) D3F_RGPT(4, 1, 1, 2); else D3F_RGPT(2, 2, 1, 2);
  }
#undef D3F_RGPT
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

}  // namespace d3f

extern "C" {

int d3f_linear_grad_weight_supported(int N, int Cin, int Cout) { return d3f::atb_supported(N, Cout, Cin) ? 1 : 0; }

size_t d3f_linear_grad_weight_ws_bytes(int N, int Cin, int Cout) { return d3f::atb_ws_bytes(N, Cout, Cin); }

/* grad_w [Cout, Cin] (nn.Linear layout) = grad_out^T [Cout, N] @ x [N, Cin] */
int d3f_linear_grad_weight(const float* x, const float* grad_out, int N, int Cin, int Cout, float* grad_w, void* ws,
                           size_t ws_bytes, void* stream) {
  if (!x || !grad_out || !grad_w || !ws || !d3f::atb_supported(N, Cout, Cin)) return D3F_EINVAL;
  if (ws_bytes < d3f::atb_ws_bytes(N, Cout, Cin)) return D3F_EWORKSPACE;
  return d3f::atb_splitk(grad_out, x, nullptr, N, Cout, Cin, grad_w, ws, (hipStream_t)stream);
}

/* The same, and the launch that sums the partial weight-gradient slabs also finishes a bias gradient:
 * grad_bias[c] (and grad_bias2[c], optional) = sum_b bias_part[b][c] over the bias_blocks x bias_cols partial column sums
 * d3f_bias_act_backward_partial left behind (one second-stage launch per layer instead of two). */
int d3f_linear_grad_weight_bias(const float* x, const float* grad_out, int N, int Cin, int Cout, float* grad_w, void* ws,
                                size_t ws_bytes, const float* bias_part, int bias_blocks, int bias_cols,
                                float* grad_bias, float* grad_bias2, void* stream) {
  if (!x || !grad_out || !grad_w || !ws || !d3f::atb_supported(N, Cout, Cin)) return D3F_EINVAL;
  if (!bias_part || !grad_bias || bias_blocks < 1 || bias_cols < 1) return D3F_EINVAL;
  if (ws_bytes < d3f::atb_ws_bytes(N, Cout, Cin)) return D3F_EWORKSPACE;
  return d3f::atb_splitk_bias(grad_out, x, nullptr, N, Cout, Cin, grad_w, ws, (hipStream_t)stream, 0, bias_part,
                              bias_blocks, bias_cols, grad_bias, grad_bias2);
}

static bool group_convert(const d3f_atb_problem* in, int n, d3f::AtbProblem* out) {
  for (int i = 0; i < n; ++i) {
    const d3f_atb_problem& q = in[i];
    // grad_w [Cout, Cin] = grad_out^T x: A = grad_out (M = Cout), B = x (N = Cin)
    out[i] = d3f::AtbProblem{q.grad_out, q.x, q.grad_w, q.N, q.Cout, q.Cin, q.ldw, q.bias_part, q.bias_blocks,
                             q.bias_cols, q.grad_bias, q.grad_bias2};
    if (!d3f::atb_problem_ok(out[i])) return false;
  }
  return true;
}

size_t d3f_linear_grad_weight_group_ws_bytes(const d3f_atb_problem* problems_host, int n) {
  if (!problems_host || n < 1 || n > 4096) return 0;
  std::vector<d3f::AtbProblem> p((size_t)n);
  if (!group_convert(problems_host, n, p.data())) return 0;
  return d3f::atb_group_ws_bytes(p.data(), n);
}

int d3f_linear_grad_weight_group(const d3f_atb_problem* problems_host, int n, void* ws, size_t ws_bytes, void* stream) {
  if (n == 0) return D3F_OK;
  if (!problems_host || n < 0 || n > 4096 || !ws) return D3F_EINVAL;
  std::vector<d3f::AtbProblem> p((size_t)n);
  if (!group_convert(problems_host, n, p.data())) return D3F_EINVAL;
  if (ws_bytes < d3f::atb_group_ws_bytes(p.data(), n)) return D3F_EWORKSPACE;
  return d3f::atb_group_launch(p.data(), n, ws, (hipStream_t)stream);
}

int d3f_linear_fused_supported(int N, int Cin, int Cout) {
  return (d3f::rowgemm_supported(N, Cin, Cout) && d3f::rowgemm_supported(N, Cout, Cin)) ? 1 : 0;
}

/* out [N,Cout] = act(x [N,Cin] @ weight[Cout,Cin]^T + bias1 + add + bias2), act = LeakyReLU(slope) (slope = 1: none);
 * bias1 / add [N,Cout] / bias2 optional.  zero_init as in d3f_bias_act_forward. */
int d3f_linear_bias_act_forward(const float* x, const float* weight, int N, int Cin, int Cout, const float* bias1,
                                const float* add, const float* bias2, float slope, float* out, float* zero_init,
                                int zero_n, void* stream) {
  if (!x || !weight || !out || !d3f::rowgemm_supported(N, Cin, Cout) || (zero_init && zero_n < 1)) return D3F_EINVAL;
  return d3f::rowgemm_launch<true, true>(x, weight, N, Cin, Cout, bias1, add, bias2, slope, out, zero_init, zero_n,
                                         (hipStream_t)stream);
}

int d3f_linear_pair_supported(int N, int Cin1, int Cin2, int Cout) {
  return d3f::rowgemm_pair_supported(N, Cin1, Cin2, Cout) ? 1 : 0;
}

/* out [N,Cout] = act(x1 [N,Cin1] @ w1[Cout,Cin1]^T + x2 [N,Cin2] @ w2[Cout,Cin2]^T + bias1 + bias2 + bias3 + bias4) */
int d3f_linear_pair_bias_act_forward(const float* x1, const float* w1, int Cin1, const float* x2, const float* w2, int Cin2,
                                     int N, int Cout, const float* bias1, const float* bias2, const float* bias3,
                                     const float* bias4, float slope, float* out, float* zero_init, int zero_n,
                                     void* stream) {
  if (!x1 || !w1 || !x2 || !w2 || !out || !d3f::rowgemm_pair_supported(N, Cin1, Cin2, Cout) || (zero_init && zero_n < 1))
    return D3F_EINVAL;
  return d3f::rowgemm_pair_launch(x1, w1, Cin1, x2, w2, Cin2, N, Cout, bias1, bias2, bias3, bias4, slope, out, zero_init,
                                  zero_n, (hipStream_t)stream);
}

/* grad_x [N,Cin] = grad_out [N,Cout] @ weight [Cout,Cin] (+ add [N,Cin], optional) */
int d3f_linear_grad_input(const float* grad_out, const float* weight, int N, int Cin, int Cout, const float* add,
                          float* grad_x, void* stream) {
  if (!grad_out || !weight || !grad_x || !d3f::rowgemm_supported(N, Cout, Cin)) return D3F_EINVAL;
  if (add)  // epilogue with no bias and slope 1: acc + add
    return d3f::rowgemm_launch<false, true>(grad_out, weight, N, Cout, Cin, nullptr, add, nullptr, 1.0f, grad_x,
                                            nullptr, 0, (hipStream_t)stream);
  return d3f::rowgemm_launch<false, false>(grad_out, weight, N, Cout, Cin, nullptr, nullptr, nullptr, 1.0f, grad_x,
                                           nullptr, 0, (hipStream_t)stream);
}

}  // extern "C"
