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

size_t atb_ws_bytes(int R, int M, int N) {
  if (!atb_supported(R, M, N)) return 0;
  return align_up(sizeof(float) * (size_t)atb_partitions(R, M, N) * M * N, 256);
}

// C [M,N] = A^T [M,R] (B [R,N] / row_div [R]); ws >= atb_ws_bytes
int atb_splitk(const float* A, const float* B, const float* row_div, int R, int M, int N, float* C, void* ws,
               hipStream_t stream, int M_out = 0) {
  if (!atb_supported(R, M, N)) return D3F_EINVAL;
  const int ti = tile_width(M), tj = tile_width(N);
  const int P = atb_partitions(R, M, N);
  int rpw = (R + P - 1) / P;
  rpw = (rpw + 3) / 4 * 4;
  dim3 grid(P, (M / (16 * ti)) * (N / (16 * tj)));
  float* part = (float*)ws;
  void* timing = kpconv_timing_open(4, stream, R, 0, 0, M, N, 0);   // (both launches: partial sums + their reduction)
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
  D3F_LAUNCH_CHECK();
  const size_t MN = (size_t)M * N;
  const size_t MN_out = (M_out > 0 && M_out < M) ? (size_t)M_out * N : MN;
  static const int fan = atb_tunable("D3F_ATB_FAN", 0);
  if (fan ? fan >= 16 : (P >= 64 && MN <= 65536))
    atb_reduce_kernel<16><<<cdiv((long long)MN, 64), 1024, 0, stream>>>(part, P, MN, C, MN_out);
  else
    atb_reduce_kernel<4><<<cdiv((long long)MN, 64), 256, 0, stream>>>(part, P, MN, C, MN_out);
  kpconv_timing_close(timing, stream);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
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
  for (int k0 = 0; k0 < K; k0 += 16) {
    const float4 a = *(const float4*)(xr + k0);
    float bq[MBW][4];
#pragma unroll
    for (int nb = 0; nb < MBW; ++nb) {
      if (WT) {
        const float4 b = *(const float4*)(W + (size_t)(c0 + nb * 16 + li) * K + k0 + 4 * lk);
        bq[nb][0] = b.x; bq[nb][1] = b.y; bq[nb][2] = b.z; bq[nb][3] = b.w;
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) bq[nb][t] = W[(size_t)(k0 + 4 * lk + t) * M + c0 + nb * 16 + li];
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
  // D[i][j]: row = row0 + 4 lk + r, column = c0 + 16 nb + li
#pragma unroll
  for (int nb = 0; nb < MBW; ++nb) {
    const int col = c0 + nb * 16 + li;
    float bias1 = 0.0f, bias2 = 0.0f;
    if (EPI) {
      if (b1) bias1 = b1[col];
      if (b2) bias2 = b2[col];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + 4 * lk + r;
      if (row < N) {
        float v = acc[nb][r];
        if (EPI) {
          if (b1) v += bias1;
          if (add) v += add[(size_t)row * M + col];
          if (b2) v += bias2;
          v = v > 0.0f ? v : v * slope;
        }
        Y[(size_t)row * M + col] = v;
      }
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
