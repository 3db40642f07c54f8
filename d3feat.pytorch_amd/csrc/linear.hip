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
// Second form of the partial-sum kernel (round 5).  The first form feeds every MFMA straight from a global load: with
// the one workgroup per CU the reduction split affords (more partitions = more slab traffic) a SIMD holds ONE wave with
// two k-steps of loads in flight, and the counters show it waiting on memory 45-78 % of the time with the matrix pipe
// <= 0.27 busy (profiles/r04_pmc_kernels.txt).  Here the prefetch depth is decoupled from the registers:
//   * every wave owns a private ring of S slots in LDS; a slot holds KS k-steps (4 KS rows) of the wave's A columns and
//     B columns (+ their row divisors), filled by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction, lane l
//     writes ring + 16 l, so a slot is simply the row-major [4 KS][16 TI] / [4 KS][16 TJ] panels);
//   * S - 1 slots are in flight while one is consumed -- (S - 1)(TI + TJ) KS KiB per wave, no VGPR behind them; the only
//     synchronisation is the wave's own counted s_waitcnt vmcnt (no workgroup barrier inside the reduction loop: a wave
//     reads only what it loaded itself);
//   * the MFMA operands come out of the slot with one 16 / 8 / 4-byte LDS read per operand vector (conflict-free:
//     16 lanes cover one 64 TI-byte row segment), same fragment layout as the first form;
//   * blockIdx -> (row partition, output block) keeps all output blocks of one row partition on ONE XCD (block b runs
//     on XCD b % 8): the A / B panels that those workgroups share are fetched from HBM once and served from that L2.
// 128-wide tiles (TI / TJ = 8 = two 64-column halves) halve the re-reads of the other operand for the wide gradients.
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
__device__ __forceinline__ void lds_dma4(const void* gsrc, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dword %1, off\n\t"
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

// column of fragment element t of lane-column i in a 16 T wide tile: T <= 4: T i + t (one T-wide vector per lane);
// T = 8: two 64-column halves, 4 i + t and 64 + 4 i + (t - 4) (two 16-byte vectors per lane)
template <int T>
__device__ __forceinline__ int frag_col(int i, int t) {
  return T <= 4 ? T * i + t : (t < 4 ? 4 * i + t : 64 + 4 * i + (t - 4));
}
template <int T>
__device__ __forceinline__ void lds_frag(const float* __restrict__ row, int li, float (&v)[T]) {
  if constexpr (T == 8) {
    const float4 a = *(const float4*)(row + 4 * li), b = *(const float4*)(row + 64 + 4 * li);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else if constexpr (T == 4) {
    const float4 a = *(const float4*)(row + 4 * li);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  } else if constexpr (T == 2) {
    const float2 a = *(const float2*)(row + 2 * li);
    v[0] = a.x; v[1] = a.y;
  } else {
    v[0] = row[li];
  }
}

extern __shared__ __attribute__((aligned(1024))) unsigned char atb_smem[];

template <int TI, int TJ, int KS, int S, bool DIV>
__global__ __launch_bounds__(256, (TI * TJ > 16) ? 1 : 2) void atb_partial_kernel2(
    const float* __restrict__ A, const float* __restrict__ B, const float* __restrict__ row_div, int R, int M, int N,
    int rows_per_wg, int P, float* __restrict__ part, int dbg) {
  constexpr int BM = 16 * TI, BN = 16 * TJ, ROWS = 4 * KS;
  constexpr int A_BYTES = ROWS * BM * 4, B_BYTES = ROWS * BN * 4, D_BYTES = DIV ? 256 : 0;
  constexpr int SB = A_BYTES + B_BYTES + D_BYTES;
  constexpr int NA = A_BYTES / 1024, NB = B_BYTES / 1024, G = NA + NB + (DIV ? 1 : 0);
  static_assert(A_BYTES % 1024 == 0 && B_BYTES % 1024 == 0, "a slot panel is a whole number of wave-instructions");
  static_assert((S - 1) * G <= 63, "vmcnt is a 6-bit counter");
  static_assert(ROWS <= 64, "one divisor request per slot");
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int li = lane & 15, lk = lane >> 4;
  const int nbj = N / BN, nblk = (M / BM) * nbj;
  // XCD-aware: workgroup L runs on XCD L % 8; all nblk output blocks of row partition p live on XCD p % 8
  const int L = blockIdx.x;
  const int p = (L & 7) + 8 * ((L >> 3) / nblk), blk = (L >> 3) % nblk;
  if (p >= P) return;
  const int m0 = (blk / nbj) * BM, n0 = (blk % nbj) * BN;
  const int r0 = p * rows_per_wg, r1 = min(R, r0 + rows_per_wg);
  const int ngroups = (r1 - r0 + ROWS - 1) / ROWS;
  const int n_my = (ngroups - wave + 3) >> 2;   // the 4 waves interleave groups of ROWS rows
  unsigned char* ring = atb_smem + wave * (S * SB);
  const unsigned ring_addr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)ring);
  const float* Ab = A + m0;
  const float* Bb = B + n0;

  f32x4 acc[TI][TJ];
#pragma unroll
  for (int t = 0; t < TI; ++t)
#pragma unroll
    for (int u = 0; u < TJ; ++u) acc[t][u] = (f32x4){0.f, 0.f, 0.f, 0.f};

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
    if constexpr (DIV) {
      const int gr = min(rg + (lane & (ROWS - 1)), r1 - 1);
      lds_dma4(row_div + gr, sa + A_BYTES + B_BYTES);
    }
  };

  // dbg (measurement only, D3F_ATB2_DBG): 1 = no loads (the MFMA / LDS-read side alone), 2 = no MFMAs (the LDS-DMA side
  // alone); results are garbage then
#pragma unroll
  for (int i = 0; i < S; ++i)
    if (i < n_my && !(dbg & 1)) issue(i, i);
  int slot = 0;
  for (int it = 0; it < n_my; ++it) {
    if (!(dbg & 1)) wait_groups<G, S - 1>(min(S - 1, n_my - 1 - it));    // group `it` has landed
    const int rg = r0 + (wave + 4 * it) * ROWS;
    const float* sA = (const float*)(ring + slot * SB);
    const float* sB = (const float*)(ring + slot * SB + A_BYTES);
    const float* sD = (const float*)(ring + slot * SB + A_BYTES + B_BYTES);
    // every fragment of the slot is requested before the first MFMA: the LDS latency is paid once per slot, not once
    // per k-step (a wave alone on its SIMD has nobody to hide it behind)
    float a[KS][TI], b[KS][TJ], sc[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int row = 4 * ks + lk;
      lds_frag<TI>(sA + row * BM, li, a[ks]);
      lds_frag<TJ>(sB + row * BN, li, b[ks]);
      sc[ks] = (rg + row < r1) ? 1.0f : 0.0f;
      if constexpr (DIV) sc[ks] = (rg + row < r1) ? 1.0f / sD[row] : 0.0f;
    }
    __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise sinks every read back in front of its own MFMAs)
    if (!(dbg & 2))
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int u = 0; u < TJ; ++u) {
        const float bv = b[ks][u] * sc[ks];
#pragma unroll
        for (int t = 0; t < TI; ++t) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks][t], bv, acc[t][u], 0, 0, 0);
      }
    }
    if (it + S < n_my && !(dbg & 1)) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slot's reads have returned before it is refilled
      issue(it + S, slot);
    }
    slot = (slot + 1 == S) ? 0 : slot + 1;
  }

  // combine the 4 waves in a fixed order (wave 0 + 1 + 2 + 3) through LDS -- the rings are free now --, 16 tiles a round
  constexpr int NT = TI * TJ, TPR = NT < 16 ? NT : 16, ROUNDS = NT / TPR;
  float* red = (float*)atb_smem;                          // [3][TPR * 256]
  float* pp = part + (size_t)p * M * N;
#pragma unroll
  for (int h = 0; h < ROUNDS; ++h) {
    __syncthreads();          // every wave is out of its ring (h = 0) / wave 0 has read round h - 1
    if (wave > 0) {
#pragma unroll
      for (int tt = 0; tt < TPR; ++tt) {
        const int tile = h * TPR + tt, t = tile / TJ, u = tile % TJ;
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(wave - 1) * (TPR * 256) + (tt * 4 + r) * 64 + lane] = acc[t][u][r];
      }
    }
    __syncthreads();
    if (wave == 0) {
      // D[i][j] (i = 4 lk + r, j = li) of tile (t, u) is C[m0 + frag_col<TI>(i, t)][n0 + frag_col<TJ>(j, u)]: a lane
      // stores its TJ columns of one row as 16 / 8 / 4-byte vectors
#pragma unroll
      for (int tl = 0; tl < TPR / TJ; ++tl) {
        const int t = h * (TPR / TJ) + tl;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v[TJ];
#pragma unroll
          for (int u = 0; u < TJ; ++u) {
            const int e = ((tl * TJ + u) * 4 + r) * 64 + lane;
            v[u] = ((acc[t][u][r] + red[e]) + red[TPR * 256 + e]) + red[2 * TPR * 256 + e];
          }
          float* dst = pp + (size_t)(m0 + frag_col<TI>(4 * lk + r, t)) * N + n0;
          if constexpr (TJ == 8) {
            *(float4*)(dst + 4 * li) = make_float4(v[0], v[1], v[2], v[3]);
            *(float4*)(dst + 64 + 4 * li) = make_float4(v[TJ > 4 ? 4 : 0], v[TJ > 5 ? 5 : 0], v[TJ > 6 ? 6 : 0], v[TJ > 7 ? 7 : 0]);
          } else if constexpr (TJ == 4) {
            *(float4*)(dst + 4 * li) = make_float4(v[0], v[TJ > 1 ? 1 : 0], v[TJ > 2 ? 2 : 0], v[TJ > 3 ? 3 : 0]);
          } else if constexpr (TJ == 2) {
            *(float2*)(dst + 2 * li) = make_float2(v[0], v[TJ > 1 ? 1 : 0]);
          } else {
            dst[li] = v[0];
          }
        }
      }
    }
  }
}

#ifndef D3F_ATB_TARGET_WGS
#define D3F_ATB_TARGET_WGS 512
#endif
static inline int tile_width(int n) { return n % 64 == 0 ? 4 : (n % 32 == 0 ? 2 : (n % 16 == 0 ? 1 : 0)); }

bool atb_supported(int R, int M, int N) { return R >= 1 && tile_width(M) && tile_width(N); }

// number of row partitions = workgroups along the reduction
static int atb_tunable(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}
static int atb_partitions(int R, int M, int N) {
  static const int forced = atb_tunable("D3F_ATB_WGS", 0);
  const long long nblocks = (long long)(M / (16 * tile_width(M))) * (N / (16 * tile_width(N)));
  // one workgroup per CU is enough unless the operands are long AND wide (profiles/atb_microbench.py: 512 partitions'
  // worth of slabs cost the small outputs 3 us each in the reduce pass; the 38k x 480 x 32 KPConv gradient wants them)
  int target = forced ? forced : ((R >= 30000 && nblocks >= 8) ? D3F_ATB_TARGET_WGS : 256);
  // the counts above were measured on one S1-class pair (<= 38k rows); several pairs stacked into one batch (round 4)
  // multiply the rows: keep the ROWS per workgroup where they were instead of the workgroup count (stacked x 4, 153k rows
  // x 480 x 32: 198 us on 525 workgroups, profiles/r04_step_timeline_stack4.txt)
  static const int scale_rows = atb_tunable("D3F_ATB_SCALE_ROWS", 40000);
  if (!forced && scale_rows > 0 && R > scale_rows) target = (int)((long long)target * R / scale_rows);
  long long wgs = (target + nblocks - 1) / nblocks;  // workgroups over the whole launch (256 CUs)
  const long long max_by_rows = (R + 63) / 64;            // >= 16 rows (4 MFMA k-steps) per wave
  if (wgs > max_by_rows) wgs = max_by_rows;
  if (wgs > 512) wgs = 512;
  if (wgs < 1) wgs = 1;
  return (int)wgs;
}

// ---- second form: configuration ------------------------------------------------------------------------------
// Tunables of the second form.  Read ONCE per variable (first use) in normal operation; with D3F_ATB_SWEEP set (the sweep
// scripts of profiles/, which change them between launches of one process) at every call.  The cache is keyed by the
// address of the name literal: a call site is one entry (benign race: two threads may both fill an entry with the same value).
static int env_int(const char* name, int dflt) {
  static const bool live = getenv("D3F_ATB_SWEEP") != nullptr;
  if (live) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
  }
  struct Entry { const char* name; int value; };
  static Entry cache[32];
  static int n_cached = 0;
  for (int i = 0; i < n_cached; ++i)
    if (cache[i].name == name) return cache[i].value;
  const char* v = getenv(name);
  const int value = v ? atoi(v) : dflt;
  if (n_cached < 32) {
    cache[n_cached].value = value;
    cache[n_cached].name = name;
    ++n_cached;
  }
  return value;
}
static inline int tile_width2(int n, int tmax) {
  if (tmax >= 8 && n % 128 == 0) return 8;
  return tile_width(n);
}
struct Atb2Cfg {
  int ti, tj, ks, s, P, rpw;
  size_t lds;
  bool ok;
};
static size_t atb2_lds_bytes(int ti, int tj, int ks, int s, bool div) {
  const size_t sb = (size_t)256 * ks * (ti + tj) + (div ? 256 : 0);
  const size_t ring = 4 * s * sb;
  const int nt = ti * tj, tpr = nt < 16 ? nt : 16;
  const size_t red = (size_t)3 * tpr * 1024;
  return ring > red ? ring : red;
}
static Atb2Cfg atb2_config(int R, int M, int N, bool div) {
  Atb2Cfg c;
  const int tmax = div ? 4 : env_int("D3F_ATB2_TMAX", 4);
  c.ti = tile_width2(M, tmax);
  c.tj = tile_width2(N, tmax);
  if (c.ti == 8 && c.tj == 8) c.tj = 4;                  // (64 accumulator tiles per wave spill; 128 x 64 blocks)
  c.ks = env_int("D3F_ATB2_KS", 4);
  c.s = env_int("D3F_ATB2_S", 2);
  if (div || c.ti == 1 || c.tj == 1) c.ks = 4;            // (16-wide panels: 16 rows fill one wave-instruction)
  if (div && c.s > 3) c.s = 3;
  if (c.ks != 2 && c.ks != 4) c.ks = 4;
  if (c.s < 2) c.s = 2;
  if (c.s > 4) c.s = 4;
  while (c.s > 2 && atb2_lds_bytes(c.ti, c.tj, c.ks, c.s, div) > 160 * 1024) --c.s;
  c.lds = atb2_lds_bytes(c.ti, c.tj, c.ks, c.s, div);
  c.ok = c.lds <= 160 * 1024;
  const long long nblk = (long long)(M / (16 * c.ti)) * (N / (16 * c.tj));
  const int rows = 4 * c.ks;
  // Row partitions.  The kernel keeps every output block of partition p on XCD p % 8, so partitions come in multiples
  // of 8 (fewer would leave whole XCDs idle: measured, 6208 x 512 x 512 on P = 4 took 81 us against 41 on P = 8) and
  // one XCD's share of the launch, (P / 8) nblk workgroups, should fit its 32 CUs in ONE round -- a second, partly
  // filled round costs as much as the first (23872 x 960 x 64: P = 34 put 75 workgroups on two XCDs' 64 slots).
  int per_cu = (int)((160 * 1024) / c.lds);
  if (per_cu > 4) per_cu = 4;
  if (c.ti * c.tj > 16) per_cu = 1;                       // (launch bound of the 32-tile instantiations)
  if (per_cu < 1) per_cu = 1;
  const long long slots = env_int("D3F_ATB2_WGS", 0) ? env_int("D3F_ATB2_WGS", 0) / 8 : 32LL * per_cu;   // per XCD
  long long q = slots / nblk;
  const int min_groups = env_int("D3F_ATB2_MIN_GROUPS", c.s);         // groups per WAVE: one ring revolution
  const long long q_rows = (long long)R / ((long long)8 * 4 * rows * min_groups);
  if (q > q_rows) q = q_rows;
  if (q > 128) q = 128;
  if (q < 1) q = 1;
  long long rpw = (R + 8 * q - 1) / (8 * q);
  rpw = (rpw + 4 * rows - 1) / (4 * rows) * (4 * rows);  // whole groups for every wave
  c.rpw = (int)rpw;
  c.P = (int)((R + rpw - 1) / rpw);
  return c;
}
// Which form runs (profiles/r05_atb_sweep.txt, 28 launches of a 3-pair stack's step: first form 762 us, second form
// everywhere 688, the better of the two per shape 648): the second form where there is arithmetic to pipeline -- from
// 1.4 GFLOP per launch, and for the wide-by-narrow KPConv gradients (960 x 64, 480 x 32) from 0.7 --, the first form,
// whose workgroups start faster, on the small launches.  D3F_ATB_V = 1 / 3: always the first / second form.
static bool atb2_wanted(const float* A, const float* B, const float* row_div, int R, int M, int N) {
  const int v = env_int("D3F_ATB_V", 2);
  if (v < 2) return false;
  if ((((uintptr_t)A | (uintptr_t)B) & 15) != 0 || (((uintptr_t)row_div) & 3) != 0) return false;
  if (v >= 3) return true;
  const double flops = 2.0 * R * (double)M * N;
  const int wide = M > N ? M : N, narrow = M > N ? N : M;
  return flops >= 1.4e9 || (wide >= 480 && narrow <= 64 && flops >= 0.7e9);
}

size_t atb_ws_bytes(int R, int M, int N) {
  if (!atb_supported(R, M, N)) return 0;
  size_t P = (size_t)atb_partitions(R, M, N);
  for (int div = 0; div < 2; ++div) {
    const Atb2Cfg c = atb2_config(R, M, N, div != 0);
    if (c.ok && (size_t)c.P > P) P = (size_t)c.P;
  }
  return align_up(sizeof(float) * P * M * N, 256);
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

template <int TI, int TJ, int KS, int S, bool DIV>
static int atb2_launch(const Atb2Cfg& c, const float* A, const float* B, const float* row_div, int R, int M, int N,
                       float* part, hipStream_t stream) {
  auto kern = atb_partial_kernel2<TI, TJ, KS, S, DIV>;
  // more than 64 KB of dynamic LDS needs the opt-in, a per-device function attribute: asked for on every call (a
  // host-side table update, no stream operation; a process may drive several devices and threads)
  if (c.lds > 64 * 1024 &&
      hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.lds) != hipSuccess) {
    (void)hipGetLastError();
    return D3F_EINVAL;
  }
  const long long nblk = (long long)(M / (16 * TI)) * (N / (16 * TJ));
  const long long P8 = ((long long)c.P + 7) / 8 * 8;      // (partitions past P exit at once: see the XCD mapping)
  kern<<<(unsigned)(P8 * nblk), 256, c.lds, stream>>>(A, B, row_div, R, M, N, c.rpw, c.P, part,
                                                      env_int("D3F_ATB2_DBG", 0));
  return D3F_OK;
}

template <int TI, int TJ, bool DIV>
static int atb2_dispatch_ring(const Atb2Cfg& c, const float* A, const float* B, const float* row_div, int R, int M,
                              int N, float* part, hipStream_t stream) {
  constexpr int W = TI + TJ;
  if constexpr (TI >= 2 && TJ >= 2 && !DIV) {
    if (c.ks == 2) {
      if (c.s >= 4) return atb2_launch<TI, TJ, 2, 4, DIV>(c, A, B, row_div, R, M, N, part, stream);
      if (c.s == 3) return atb2_launch<TI, TJ, 2, 3, DIV>(c, A, B, row_div, R, M, N, part, stream);
      return atb2_launch<TI, TJ, 2, 2, DIV>(c, A, B, row_div, R, M, N, part, stream);
    }
  }
  if constexpr (!DIV && W <= 12) {
    if (c.s >= 4) return atb2_launch<TI, TJ, 4, 4, DIV>(c, A, B, row_div, R, M, N, part, stream);
  }
  if constexpr (W <= 12) {
    if (c.s >= 3) return atb2_launch<TI, TJ, 4, 3, DIV>(c, A, B, row_div, R, M, N, part, stream);
  }
  return atb2_launch<TI, TJ, 4, 2, DIV>(c, A, B, row_div, R, M, N, part, stream);
}

template <bool DIV>
static int atb2_dispatch(const Atb2Cfg& c, const float* A, const float* B, const float* row_div, int R, int M, int N,
                         float* part, hipStream_t stream) {
#define D3F_ATB2(I, J) \
  case (I) * 16 + (J): return atb2_dispatch_ring<I, J, DIV>(c, A, B, row_div, R, M, N, part, stream)
  switch (c.ti * 16 + c.tj) {
    D3F_ATB2(1, 1); D3F_ATB2(1, 2); D3F_ATB2(1, 4);
    D3F_ATB2(2, 1); D3F_ATB2(2, 2); D3F_ATB2(2, 4);
    D3F_ATB2(4, 1); D3F_ATB2(4, 2); D3F_ATB2(4, 4);
    default: break;
  }
  if constexpr (!DIV) {
    switch (c.ti * 16 + c.tj) {
      D3F_ATB2(1, 8); D3F_ATB2(2, 8); D3F_ATB2(4, 8);
      D3F_ATB2(8, 1); D3F_ATB2(8, 2); D3F_ATB2(8, 4);
      default: break;
    }
  }
#undef D3F_ATB2
  return D3F_EINVAL;
}

// C [M,N] = A^T [M,R] (B [R,N] / row_div [R]); ws >= atb_ws_bytes.  bias_*: the second stage also finishes a bias
// gradient from `bias_blocks` rows of partial column sums [bias_blocks, bias_cols] (see atb_reduce_bias_kernel).
int atb_splitk_bias(const float* A, const float* B, const float* row_div, int R, int M, int N, float* C, void* ws,
                    hipStream_t stream, int M_out, const float* bias_part, int bias_blocks, int bias_cols,
                    float* grad_bias, float* grad_bias2) {
  if (!atb_supported(R, M, N)) return D3F_EINVAL;
  if (bias_part && (bias_blocks < 1 || bias_cols < 1 || !grad_bias)) return D3F_EINVAL;
  float* part = (float*)ws;
  void* timing = kpconv_timing_open(4, stream, R, 0, 0, M, N, 0);   // (both launches: partial sums + their reduction)
  int P = 0;
  bool launched = false;
  if (atb2_wanted(A, B, row_div, R, M, N)) {
    const Atb2Cfg c = atb2_config(R, M, N, row_div != nullptr);
    if (c.ok) {
      const int rc = row_div ? atb2_dispatch<true>(c, A, B, row_div, R, M, N, part, stream)
                             : atb2_dispatch<false>(c, A, B, row_div, R, M, N, part, stream);
      if (rc == D3F_OK) {
        launched = true;
        P = c.P;
      }
    }
  }
  if (!launched) {
    const int ti = tile_width(M), tj = tile_width(N);
    P = atb_partitions(R, M, N);
    int rpw = (R + P - 1) / P;
    rpw = (rpw + 3) / 4 * 4;
    dim3 grid(P, (M / (16 * ti)) * (N / (16 * tj)));
    static const int deep = atb_tunable("D3F_ATB_U", 0);   // 0: the measured default per tile shape
#define D3F_ATB(I, J)                                                                                      \
  {                                                                                                        \
    const int u = deep ? deep : (((I) * (J) >= 8) ? 2 : 4);                                                \
    if (u >= 8) atb_partial_kernel<I, J, 8><<<grid, 256, 0, stream>>>(A, B, row_div, R, M, N, rpw, part);  \
    else if (u >= 4) atb_partial_kernel<I, J, 4><<<grid, 256, 0, stream>>>(A, B, row_div, R, M, N, rpw, part); \
    else atb_partial_kernel<I, J, 2><<<grid, 256, 0, stream>>>(A, B, row_div, R, M, N, rpw, part);         \
  }
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
  }
  D3F_LAUNCH_CHECK();
  const size_t MN = (size_t)M * N;
  const size_t MN_out = (M_out > 0 && M_out < M) ? (size_t)M_out * N : MN;
  static const int fan = atb_tunable("D3F_ATB_FAN", 0);
  if (bias_part) {
    const int fan16 = (fan ? fan >= 16 : (P >= 64 && MN <= 65536)) ? 1 : 0;
    const int cb = cdiv((long long)MN, fan16 ? 64 : 256);
    atb_reduce_bias_kernel<<<cb + cdiv(bias_cols, 64), 1024, 0, stream>>>(part, P, MN, C, MN_out, cb, bias_part,
                                                                         bias_blocks, bias_cols, grad_bias, grad_bias2,
                                                                         fan16);
  } else if (fan ? fan >= 16 : (P >= 64 && MN <= 65536)) {
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

bool rowgemm_supported(int N, int K, int M) {
  return N >= 1 && K >= 16 && K % 16 == 0 && K <= 1024 && (M == 32 || M == 64 || M == 128 || M == 256);
}

template <bool WT, bool EPI>
static int rowgemm_launch(const float* X, const float* W, int N, int K, int M, const float* b1, const float* add,
                          const float* b2, float slope, float* Y, float* zinit, int zn, hipStream_t stream) {
#define D3F_RG(MBW, CS)                                                                                  \
  rowgemm_kernel<MBW, CS, WT, EPI><<<cdiv(N, 16 * (4 / CS)), 256, 0, stream>>>(X, W, N, K, b1, add, b2, slope, Y, \
                                                                               zinit, zn)
  switch (M) {
    case 32: D3F_RG(1, 2); break;
    case 64: D3F_RG(2, 2); break;
    case 128: D3F_RG(2, 4); break;
    case 256: D3F_RG(4, 4); break;
    default: return D3F_EINVAL;
  }
#undef D3F_RG
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
