// f32 GEMM with fused prologue / epilogue for the point-wise ("unary") layers and the few-point KPConv contractions
// of the D3Feat network (reference models/blocks.py:481-541 nn.Linear + BatchNormBlock bias + LeakyReLU, :598,:676,:686
// block epilogues, and the `weighted_features @ weights` contraction of KPConv.forward, blocks.py:375-380).
//
// Why an own kernel: below the first pyramid level the layers are GEMMs with 150..8000 rows against 64..7680 columns.
// A BLAS heuristic puts them on 100..250 workgroups that each walk the whole reduction (the 159 x 7680 x 512 KPConv
// contraction: 38 us on 192 workgroups, 120 dependent k-iterations each), and every one of them is followed by a
// separate bias / activation / mask / column-sum launch.  One kernel family here does
//   C = epi( op_A(A) . op_B(B) )
// with the elementwise work folded into the operand loads and the accumulator store:
//   operands   : each of A, B is either "KC" (reduction index contiguous in memory) or "KS" (reduction index = row),
//                so x W^T, g W, g^T x, wf W, g W^T all run without a transposed copy;
//   prologue   : A' = A * (mask > 0 ? 1 : slope)   -- the LeakyReLU backward mask evaluated on the saved activation
//                while the operand is loaded (no masked-gradient tensor, no separate launch);
//   by-product : row sums of A' (the bias gradient when A' = masked-gradient^T in the weight-gradient GEMM);
//   epilogue   : (/ row_div[m]) + bias1[n] + add[m,n] (or add[idx[m],n]: nearest-upsampled coarse term) + bias2[n],
//                LeakyReLU, store; a side job clears the caller's scratch (backward accumulators).
//
// Mapping to gfx950 (round 3: register-direct form, no LDS, no barriers).  v_mfma_f32_16x16x4_f32 runs at the f32
// vector rate (32 cycles per instruction per SIMD), i.e. these GEMMs are matrix-pipe bound long before they are
// bandwidth bound: 16 x the time per operand byte of a bf16 GEMM.  What the small shapes lack is PARALLELISM and
// latency cover, not staging bandwidth.  So:
//   * the unit of work is ONE WAVE = a (16 FA) x (16 FB) tile of C over one slice of the reduction.  The 8 waves of a
//     workgroup are KW x MW: KW waves split the reduction of the SAME tile (a 1024-deep reduction is a chain of 256
//     dependent-latency-bound chunk steps for one wave, 32 for eight) and combine their accumulators through LDS in wave
//     order -- no slab, no second launch, bit-reproducible --, MW such groups are stacked along M;
//   * operand fragments go straight from memory to the MFMA operand registers as 16-byte loads, using the freedom to
//     permute the reduction index consistently in A and B: within a 16-deep chunk, MFMA step s (0..3) and k-slot
//     g = lane >> 4 stand for k = 4 g + s.  A KC operand then loads ONE float4 per fragment and chunk (lane (i, g) <-
//     X[i][4g .. 4g+3], 64 contiguous bytes per row) and a KS operand ONE float4 per step (lane (i, g) <- X[4g+s][4i ..
//     4i+3]: four consecutive output indices, which become the lane's element of FOUR fragments -- the tile's output
//     index is permuted, idx = 4 i + f, and so float4 stores come out of the epilogue);
//   * two register sets alternate (loads of chunk c+1 in flight under the 16..64 MFMAs of chunk c), 3-4 waves per SIMD
//     cover the rest of the latency;
//   * only when that still leaves the chip idle (192 rows x 7680 deep) the reduction is also split over grid.z; partial
//     tiles go to a slab and a second launch sums them in slice order and applies the epilogue (no atomics).
#include "kpconv_tile.hpp"

namespace d3f {

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

__device__ __forceinline__ float f4c(const float4& v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); }

// One operand of one wave.  KC: NV = F float4 per chunk (fragment f <- row base + 16 f + li, k = kb + 4 lg ..), the
// per-fragment byte offsets are lane constants and the chunk's k offset rides in the scalar offset of the buffer load.
// KS: NV = 4 float4 per chunk (step s <- row k = kb + 4 lg + s, output indices base + 4 li ..).
template <bool KS, int F>
struct Operand {
  static constexpr int NV = KS ? 4 : F;
  __amdgpu_buffer_rsrc_t rs;
  unsigned voff[KS ? 1 : F];
  unsigned ld4;   // row stride in bytes

  __device__ __forceinline__ void init(const float* p, int ld, int outer, int K, int base, int li, int lg) {
    // outer = number of output indices (rows of a KC operand, columns of a KS operand)
    ld4 = (unsigned)ld * 4u;
    const size_t bytes = KS ? ((size_t)(K - 1) * ld + outer) * 4 : ((size_t)(outer - 1) * ld + K) * 4;
    rs = make_rsrc(p, (unsigned)bytes);
    if (KS) {
      const int col = min(base + 4 * li, outer - 4);   // clamped: a tile past the edge recomputes the last columns
      voff[0] = (unsigned)(4 * lg) * ld4 + (unsigned)col * 4u;
    } else {
#pragma unroll
      for (int f = 0; f < F; ++f) {
        const int row = min(base + 16 * f + li, outer - 1);
        voff[f] = (unsigned)row * ld4 + (unsigned)(4 * lg) * 4u;
      }
    }
  }

  // TAIL: the chunk reaches past K (K % 16 != 0): out-of-range pieces read zeros.  !live (wave-uniform): the chunk lies
  // past the wave's range -- the loads are still ISSUED (every path through the pipelined loop then carries the same
  // number of outstanding loads, which is what lets the compiler count them instead of draining the queue) but their
  // scalar offset points past the operand, so they return zeros without touching memory.
  template <bool TAIL>
  __device__ __forceinline__ void load(int kb, int K, int lg, float4 (&v)[NV], bool live = true) const {
    constexpr unsigned kPast = 0x7ffffff0u;
    if (!KS) {
      const unsigned so = live ? (unsigned)kb * 4u : kPast;
#pragma unroll
      for (int f = 0; f < F; ++f) {
        if (TAIL) {
          const unsigned vo = (kb + 4 * lg < K) ? voff[f] + so : 0xfffffff0u;
          v[f] = buf_load_f4(rs, vo);
        } else {
          const u32x4v r = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[f], so, 0);
          v[f] = make_float4(__uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[2]), __uint_as_float(r[3]));
        }
      }
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const unsigned so = live ? (unsigned)(kb + s) * ld4 : kPast;
        if (TAIL) {
          const unsigned vo = (kb + 4 * lg + s < K) ? voff[0] + so : 0xfffffff0u;
          v[s] = buf_load_f4(rs, vo);
        } else {
          const u32x4v r = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[0], so, 0);
          v[s] = make_float4(__uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[2]), __uint_as_float(r[3]));
        }
      }
    }
  }

  // MFMA operand of fragment f at step s
  __device__ __forceinline__ static float frag(const float4 (&v)[NV], int f, int s) {
    return KS ? f4c(v[s], f) : f4c(v[f], s);
  }
};

// FA / FB = fragments per wave along M / N (a KS operand always spans 4).  Workgroup = NW waves = KW (reduction split of
// one tile) x NW / KW (tiles stacked along M).
constexpr int G_NW = 8;
template <bool AKS, bool BKS, int FA, int FB, bool MASK>
__global__ __launch_bounds__(64 * G_NW) void gemm_direct_kernel(const GemmP p, const int kw_shift, const int per_z,
                                                                const int per_w) {
  // kw_shift = log2(KW); per_z / per_w = chunks per grid slice / per wave of a slice (rounded up) -- computed on the host:
  // integer divisions by run-time values are ~100-instruction sequences each, and this kernel's shortest launches are
  // a few thousand cycles long
  static_assert(!AKS || FA == 4, "a KS operand spans four fragments");
  static_assert(!BKS || FB == 4, "a KS operand spans four fragments");
  constexpr int TM = 16 * FA, TN = 16 * FB;
  typedef Operand<AKS, FA> OA;
  typedef Operand<BKS, FB> OB;
  // (the wave id IS wave-uniform, but anything derived from threadIdx is divergent to the compiler: without the
  // readfirstlane every buffer load whose scalar offset depends on it is wrapped in a waterfall loop)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int KW = 1 << kw_shift, MW = G_NW >> kw_shift, kw = wave & (KW - 1), mw = wave >> kw_shift;
  const int m0 = (blockIdx.y * MW + mw) * TM, n0 = blockIdx.x * TN;
  const int S = gridDim.z, z = blockIdx.z;
  extern __shared__ __attribute__((aligned(16))) float4 red[];   // [NW][FA*FB + 1][64]: accumulators of the waves kw > 0
  constexpr int SLOTS = FA * FB + 1;
  if (p.zero_init && blockIdx.x == 0 && blockIdx.y == 0 && z == 0)
    for (int t = threadIdx.x; t < p.zero_n; t += 64 * G_NW) p.zero_init[t] = 0.0f;
  const bool active = m0 < p.M;  // (a whole wave past the last row idles up to the barrier)

  OA oa;
  OB ob;
  OA om;  // mask, addressed like A
  oa.init(p.A, p.lda, p.M, p.K, m0, li, lg);
  ob.init(p.B, p.ldb, p.N, p.K, n0, li, lg);
  if (MASK) om.init(p.a_mask, p.lda, p.M, p.K, m0, li, lg);

  const int chunks = (p.K + 15) >> 4;
  const int zb = min(z * per_z, chunks), ze = min(zb + per_z, chunks);
  // this wave's part of the slice (an inactive wave gets an empty range)
  const int cb = active ? min(zb + kw * per_w, ze) : 0;
  const int ce = active ? min(cb + per_w, ze) : 0;
  const bool has_tail = (p.K & 15) != 0 && ce == chunks && ce > cb;
  const int full_end = has_tail ? ce - 1 : ce;

  f32x4 acc[FA][FB];
#pragma unroll
  for (int a = 0; a < FA; ++a)
#pragma unroll
    for (int b = 0; b < FB; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 asum = make_float4(0.f, 0.f, 0.f, 0.f);  // AKS: running sums of this lane's 4 A' rows over its k slots
  const bool want_sum = AKS && p.rowsum != nullptr && blockIdx.x == 0;
  const float mslope = p.mask_slope;

  // epilogue operands of the tile are fetched up front (their latency runs under the reduction loop instead of in
  // front of the stores): both biases, the row divisors and the residual's row indices
  float b1v[FB], b2v[FB], dvv[FA][4];
  int arowv[FA][4];
#pragma unroll
  for (int fb = 0; fb < FB; ++fb) {
    const int col = min(n0 + (BKS ? 4 * li + fb : 16 * fb + li), p.N - 1);
    b1v[fb] = (p.bias1 && S == 1) ? p.bias1[col] : 0.0f;
    b2v[fb] = (p.bias2 && S == 1) ? p.bias2[col] : 0.0f;
  }
#pragma unroll
  for (int fa = 0; fa < FA; ++fa)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 4 * lg + r;
      const int row = min(m0 + (AKS ? 4 * i + fa : 16 * fa + i), p.M - 1);
      dvv[fa][r] = (p.row_div && S == 1) ? p.row_div[row] : 1.0f;
      arowv[fa][r] = (p.add_idx && S == 1) ? p.add_idx[(size_t)row * p.idx_stride] : row;
    }

  // register ring: NST chunk-sized operand sets; the loads of chunk c + NST - 1 are issued before the MFMAs of chunk c
  constexpr int NM = MASK ? OA::NV : 1;
  constexpr int STAGE_REGS = 4 * (OA::NV * (MASK ? 2 : 1) + OB::NV);
  constexpr int NST = STAGE_REGS <= 24 ? 4 : (STAGE_REGS <= 32 ? 3 : 2);
  float4 va[NST][OA::NV], vb[NST][OB::NV], vm[NST][NM];

  auto compute = [&](float4 (&va)[OA::NV], const float4 (&vb)[OB::NV], const float4 (&vm)[NM]) {
    if (MASK) {
#pragma unroll
      for (int j = 0; j < OA::NV; ++j) {
        va[j].x *= vm[j % NM].x > 0.0f ? 1.0f : mslope;
        va[j].y *= vm[j % NM].y > 0.0f ? 1.0f : mslope;
        va[j].z *= vm[j % NM].z > 0.0f ? 1.0f : mslope;
        va[j].w *= vm[j % NM].w > 0.0f ? 1.0f : mslope;
      }
    }
    if (AKS && want_sum) {
#pragma unroll
      for (int j = 0; j < OA::NV; ++j) { asum.x += va[j].x; asum.y += va[j].y; asum.z += va[j].z; asum.w += va[j].w; }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int fa = 0; fa < FA; ++fa)
#pragma unroll
        for (int fb = 0; fb < FB; ++fb)
          acc[fa][fb] = __builtin_amdgcn_mfma_f32_16x16x4f32(OA::frag(va, fa, s), OB::frag(vb, fb, s), acc[fa][fb], 0, 0, 0);
  };
#define D3F_LOAD(TAIL, chunk, va, vb, vm)                                                       \
  {                                                                                             \
    const bool live_ = TAIL || (chunk) < full_end;                                              \
    oa.template load<TAIL>((chunk) * 16, p.K, lg, va, live_);                                   \
    if (MASK) om.template load<TAIL>((chunk) * 16, p.K, lg, reinterpret_cast<float4(&)[OA::NV]>(vm), live_); \
    ob.template load<TAIL>((chunk) * 16, p.K, lg, vb, live_);                                   \
  }

  // chunk cb + t lives in set t % NST
#pragma unroll
  for (int j = 0; j < NST - 1; ++j) D3F_LOAD(false, cb + j, va[j], vb[j], vm[j])
  int c = cb;
  for (; c + NST <= full_end; c += NST) {          // steady state: NST chunks per trip, no branch inside
#pragma unroll
    for (int j = 0; j < NST; ++j) {
      D3F_LOAD(false, c + j + NST - 1, va[(j + NST - 1) % NST], vb[(j + NST - 1) % NST], vm[(j + NST - 1) % NST])
      __builtin_amdgcn_sched_barrier(0);           // (hipcc otherwise sinks the loads between later MFMAs: the ring
      compute(va[j], vb[j], vm[j]);                //  would run one to two chunks ahead instead of NST - 1)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int j = 0; j < NST - 1; ++j)                // the last < NST chunks (their loads are already in flight)
    if (c + j < full_end) compute(va[j], vb[j], vm[j]);
  if (has_tail) {
    D3F_LOAD(true, ce - 1, va[0], vb[0], vm[0])
    compute(va[0], vb[0], vm[0]);
  }
#undef D3F_LOAD

  // ---- by-product: row sums of A' over this wave's reduction range.  Lane (li, lg) holds rows m0 + 4 li .. + 3 summed
  // over its k slots; the four slots are combined in a fixed order.
  float4 rs4 = asum;
  if (want_sum) {
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
      rs4.x += __shfl_xor(rs4.x, o, 64); rs4.y += __shfl_xor(rs4.y, o, 64);
      rs4.z += __shfl_xor(rs4.z, o, 64); rs4.w += __shfl_xor(rs4.w, o, 64);
    }
  }
  // ---- the KW waves of a tile: wave kw > 0 parks its accumulators in LDS, wave kw = 0 adds them in wave order
  if (KW > 1) {
    if (kw > 0 && active) {
      float4* dst = red + (size_t)wave * SLOTS * 64 + lane;
#pragma unroll
      for (int fa = 0; fa < FA; ++fa)
#pragma unroll
        for (int fb = 0; fb < FB; ++fb)
          dst[(fa * FB + fb) * 64] = make_float4(acc[fa][fb][0], acc[fa][fb][1], acc[fa][fb][2], acc[fa][fb][3]);
      if (want_sum) dst[FA * FB * 64] = rs4;
    }
    __syncthreads();
    if (kw == 0 && active) {
      for (int k2 = 1; k2 < KW; ++k2) {
        const float4* src = red + (size_t)(wave + k2) * SLOTS * 64 + lane;
#pragma unroll
        for (int fa = 0; fa < FA; ++fa)
#pragma unroll
          for (int fb = 0; fb < FB; ++fb) {
            const float4 v = src[(fa * FB + fb) * 64];
            acc[fa][fb][0] += v.x; acc[fa][fb][1] += v.y; acc[fa][fb][2] += v.z; acc[fa][fb][3] += v.w;
          }
        if (want_sum) {
          const float4 v = src[FA * FB * 64];
          rs4.x += v.x; rs4.y += v.y; rs4.z += v.z; rs4.w += v.w;
        }
      }
    }
  }
  if (kw != 0 || !active) return;
  if (want_sum) {
    const int m = m0 + 4 * li;
    if (lg == 0 && m < p.M) {  // M % 4 == 0 for a KS operand
      if (S == 1) {
        *(float4*)(p.rowsum + m) = rs4;
        if (p.rowsum2) *(float4*)(p.rowsum2 + m) = rs4;
      } else {
        *(float4*)(p.slab + (size_t)z * ((size_t)p.M * p.N + p.M) + (size_t)p.M * p.N + m) = rs4;
      }
    }
  }

  // ---- accumulator store.  acc[fa][fb][r] = D[i = 4 lg + r][li] of fragment (fa, fb):
  //   row = m0 + (AKS ? 4 i + fa : 16 fa + i),  col = n0 + (BKS ? 4 li + fb : 16 fb + li)
  if (S > 1) {
    float* slab = p.slab + (size_t)z * ((size_t)p.M * p.N + (p.rowsum ? p.M : 0));
#pragma unroll
    for (int fa = 0; fa < FA; ++fa)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 4 * lg + r;
        const int row = m0 + (AKS ? 4 * i + fa : 16 * fa + i);
        if (row >= p.M) continue;
        if (BKS) {
          const int col = n0 + 4 * li;
          if (col < p.N)
            *(float4*)(slab + (size_t)row * p.N + col) = make_float4(acc[fa][0][r], acc[fa][FB > 1 ? 1 : 0][r],
                                                                     acc[fa][FB > 2 ? 2 : 0][r], acc[fa][FB > 3 ? 3 : 0][r]);
        } else {
#pragma unroll
          for (int fb = 0; fb < FB; ++fb) {
            const int col = n0 + 16 * fb + li;
            if (col < p.N) slab[(size_t)row * p.N + col] = acc[fa][fb][r];
          }
        }
      }
    return;
  }
  // (bias1 and bias2 enter the reference's sum at different points: kept apart)
#pragma unroll
  for (int fa = 0; fa < FA; ++fa)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 4 * lg + r;
      const int row = m0 + (AKS ? 4 * i + fa : 16 * fa + i);
      if (row >= p.M) continue;
      const float dv = dvv[fa][r];
      const int arow = arowv[fa][r];
      const bool alive = p.add != nullptr && (!p.add_idx || (arow >= 0 && arow < p.add_rows));
      float v[FB];
#pragma unroll
      for (int fb = 0; fb < FB; ++fb) {
        const int col = n0 + (BKS ? 4 * li + fb : 16 * fb + li);
        float t = acc[fa][fb][r];
        if (p.row_div) t /= dv;
        if (p.bias1) t += b1v[fb];
        if (alive && col < p.N) t += p.add[(size_t)arow * p.ldadd + col];
        if (p.bias2) t += b2v[fb];
        v[fb] = t > 0.0f ? t : t * p.slope;
      }
      if (BKS) {
        const int col = n0 + 4 * li;
        if (col < p.N)
          *(float4*)(p.C + (size_t)row * p.ldc + col) = make_float4(v[0], v[FB > 1 ? 1 : 0], v[FB > 2 ? 2 : 0], v[FB > 3 ? 3 : 0]);
      } else {
#pragma unroll
        for (int fb = 0; fb < FB; ++fb) {
          const int col = n0 + 16 * fb + li;
          if (col < p.N) p.C[(size_t)row * p.ldc + col] = v[fb];
        }
      }
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

// Work decomposition: the wave tile (fa x fb fragments), the in-workgroup reduction split kw (1, 2, 4 or 8 of the
// workgroup's 8 waves; the other 8 / kw stack along M) and the grid-level split.  A KS operand fixes its side at 4
// fragments.  Target: ~3 waves per SIMD on 256 CUs with at least 4 chunks (64 reduction indices) per wave; the
// in-workgroup split is free (LDS), the grid split costs a slab round trip and a second launch.
struct GemmPlan { int fa, fb, kw, split; };
static int g_force_fa = 0, g_force_fb = 0, g_force_kw = 0, g_force_split = 0;   // measurement aid (d3f_debug_set_gemm_plan)

static GemmPlan gemm_plan(int M, int N, int K, bool aks, bool bks) {
  GemmPlan pl;
  pl.fb = bks ? 4 : (N <= 32 ? 2 : 4);
  if (!bks && (g_force_fb == 2 || g_force_fb == 4)) pl.fb = g_force_fb;
  if (aks) pl.fa = 4;
  else pl.fa = ((long long)cdiv(M, 32) * cdiv(N, 16 * pl.fb) >= 4096) ? 2 : 1;
  if (!aks && (g_force_fa == 1 || g_force_fa == 2)) pl.fa = g_force_fa;
  const long long tiles = (long long)cdiv(M, 16 * pl.fa) * cdiv(N, 16 * pl.fb);
  const int chunks = cdiv(K, 16);
  long long want = (3072 + tiles - 1) / tiles;          // waves per tile that fill the chip
  if (want > chunks / 4) want = chunks / 4;             // >= 4 chunks per wave
  pl.kw = want >= 8 ? 8 : (want >= 4 ? 4 : (want >= 2 ? 2 : 1));
  if (g_force_kw == 1 || g_force_kw == 2 || g_force_kw == 4 || g_force_kw == 8) pl.kw = g_force_kw;
  pl.split = 1;
  if (pl.kw == 8 && want >= 24) {                       // the second launch has to buy >= 3x the waves
    long long s = want / 8;
    if (s > 4) s = 4;                                   // (measured: beyond 4 slices the slab round trip eats the gain)
    pl.split = (int)s;
  }
  if (g_force_split >= 1) pl.split = g_force_split > chunks ? chunks : g_force_split;
  return pl;
}

size_t gemm_ws_bytes(int M, int N, int K, bool rowsum) {
  int S = 1;
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b) {
      const int s = gemm_plan(M, N, K, a != 0, b != 0).split;
      if (s > S) S = s;
    }
  if (S == 1) return 0;
  return align_up(sizeof(float) * (size_t)S * ((size_t)M * N + (rowsum ? M : 0)), 256);
}

static bool aligned16(const void* p) { return ((uintptr_t)p & 15u) == 0; }

template <bool AKS, bool BKS, int FA, int FB>
static void gemm_launch_t(const GemmP& p, dim3 grid, int kw, bool mask, hipStream_t stream) {
  const int kw_shift = kw >= 8 ? 3 : (kw >= 4 ? 2 : (kw >= 2 ? 1 : 0));
  const int per_z = cdiv(cdiv(p.K, 16), (int)grid.z), per_w = cdiv(per_z, kw);
  const size_t lds = kw > 1 ? sizeof(float4) * (size_t)G_NW * (FA * FB + 1) * 64 : 0;
  if (lds > 65536) {  // (64 x 64 wave tiles: 136 KB) -- the opt-in is per kernel and cheap
    if (mask) (void)hipFuncSetAttribute((const void*)gemm_direct_kernel<AKS, BKS, FA, FB, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    else (void)hipFuncSetAttribute((const void*)gemm_direct_kernel<AKS, BKS, FA, FB, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  if (mask) gemm_direct_kernel<AKS, BKS, FA, FB, true><<<grid, 64 * G_NW, lds, stream>>>(p, kw_shift, per_z, per_w);
  else gemm_direct_kernel<AKS, BKS, FA, FB, false><<<grid, 64 * G_NW, lds, stream>>>(p, kw_shift, per_z, per_w);
}

int gemm_launch(GemmP p, bool a_ks, bool b_ks, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (!p.A || !p.B || !p.C || p.M < 0 || p.N < 1 || p.K < 1) return D3F_EINVAL;
  // float4 granularity along every contiguous dimension
  const int a_cont = a_ks ? p.M : p.K, b_cont = b_ks ? p.N : p.K;
  if (a_cont % 4 || b_cont % 4 || p.lda % 4 || p.ldb % 4 || p.N % 4 || p.ldc < p.N) return D3F_EINVAL;
  if (!aligned16(p.A) || !aligned16(p.B) || (p.a_mask && !aligned16(p.a_mask))) return D3F_EINVAL;
  if (b_ks && (!aligned16(p.C) || p.ldc % 4)) return D3F_EINVAL;                // float4 stores of the permuted columns
  if (p.rowsum && (!a_ks || !aligned16(p.rowsum) || (p.rowsum2 && !aligned16(p.rowsum2)))) return D3F_EINVAL;
  if (p.rowsum2 && !p.rowsum) return D3F_EINVAL;
  if (p.add && p.ldadd < p.N) return D3F_EINVAL;
  // 32-bit byte offsets inside the operands
  const double a_ext = ((double)((a_ks ? p.K : p.M) - 1) * p.lda + (a_ks ? p.M : p.K)) * 4.0;
  const double b_ext = ((double)((b_ks ? p.K : p.N) - 1) * p.ldb + (b_ks ? p.N : p.K)) * 4.0;
  if (a_ext >= 4294967280.0 || b_ext >= 4294967280.0) return D3F_EINVAL;
  if (p.M == 0) {
    if (p.zero_init && p.zero_n > 0 && zero_async(p.zero_init, sizeof(float) * (size_t)p.zero_n, stream) != hipSuccess)
      return D3F_ELAUNCH;
    return D3F_OK;
  }
  const GemmPlan pl = gemm_plan(p.M, p.N, p.K, a_ks, b_ks);
  const int S = pl.split;
  if (S > 1) {
    const size_t need = align_up(sizeof(float) * (size_t)S * ((size_t)p.M * p.N + (p.rowsum ? p.M : 0)), 256);
    if (!ws || ws_bytes < need) return D3F_EWORKSPACE;
    p.slab = (float*)ws;
  }
  const int kw = pl.kw;
  dim3 grid(cdiv(p.N, 16 * pl.fb), cdiv(p.M, 16 * pl.fa * (G_NW / kw)), S);
  const bool mask = p.a_mask != nullptr;
  if (a_ks) {
    if (b_ks) gemm_launch_t<true, true, 4, 4>(p, grid, kw, mask, stream);
    else if (pl.fb == 4) gemm_launch_t<true, false, 4, 4>(p, grid, kw, mask, stream);
    else gemm_launch_t<true, false, 4, 2>(p, grid, kw, mask, stream);
  } else if (pl.fa == 2) {
    if (b_ks) gemm_launch_t<false, true, 2, 4>(p, grid, kw, mask, stream);
    else if (pl.fb == 4) gemm_launch_t<false, false, 2, 4>(p, grid, kw, mask, stream);
    else gemm_launch_t<false, false, 2, 2>(p, grid, kw, mask, stream);
  } else {
    if (b_ks) gemm_launch_t<false, true, 1, 4>(p, grid, kw, mask, stream);
    else if (pl.fb == 4) gemm_launch_t<false, false, 1, 4>(p, grid, kw, mask, stream);
    else gemm_launch_t<false, false, 1, 2>(p, grid, kw, mask, stream);
  }
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

void d3f_debug_set_gemm_plan(int fa, int fb, int kw, int split) {
  d3f::g_force_fa = fa;
  d3f::g_force_fb = fb;
  d3f::g_force_kw = kw;
  d3f::g_force_split = split;
}

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
