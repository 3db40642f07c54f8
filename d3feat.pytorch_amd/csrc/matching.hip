// Dense mutual-nearest-neighbor descriptor matching on the f32 matrix cores.
//
// Replaces reference geometric_registration/common.py:5-21 (build_correspondence):
//   distance = sqrt(2 - 2 * S @ T.T); source_idx = argmin(axis=1); target_idx = argmin(axis=0); keep mutual pairs.
// At N = 19.3k the reference's distance matrix is 1.5 GB; it is never formed here.
//
//   workgroup = 4 waves x (RT x 16) source rows; the A fragments of a wave's rows stay in registers for the whole
//   sweep; target descriptors stream through in tiles of 64 (fragments loaded straight from L2); every B fragment
//   feeds RT row tiles, i.e. 4*RT v_mfma_f32_16x16x4_f32 (exact f32 FMA chains) per 16-B load.
//   The target range is split over grid.y so that the launch fills the chip; partial winners are merged with ONE
//   64-bit atomicMin per row on the key (order-preserving bits of the distance << 32 | column): ties resolve to the
//   lowest index exactly like np.argmin, in any arrival order.
//   The reference compares sqrt(2 - 2s): the running minimum is kept on v = 2 - 2s (no sqrt in the inner loop) and the
//   correctly rounded sqrt is taken only when a candidate beats the incumbent (a handful of times per row), so two
//   different v that round to the same distance keep the earlier column as numpy does.  A negative v (NaN after the
//   reference's sqrt) wins like NaN does in np.argmin.
// The column arg-min is the same kernel with S and T swapped.
#include "common.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kColsPerTile = 64;

__device__ __forceinline__ uint32_t ord_bits(float f) {  // monotone float -> uint32
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__global__ void fill_u64_kernel(unsigned long long* __restrict__ p, int n, unsigned long long v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

__global__ void unpack_arg_kernel(const unsigned long long* __restrict__ best, int n, int32_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (int32_t)(best[i] & 0xffffffffull);
}

template <int C, int RT, bool PF>
__global__ __launch_bounds__(256) void row_argmin_kernel(const float* __restrict__ S, int Ns,
                                                         const float* __restrict__ T, int Nt, int cols_per_chunk,
                                                         unsigned long long* __restrict__ best) {
  static_assert(C % 16 == 0, "descriptor width must be a multiple of 16");
  constexpr int KS = C / 16;  // float4 chunks per lane: reduction index c = 16*u + 4*(lane>>4) + t
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int row0 = (blockIdx.x * 4 + wave) * (16 * RT);
  if (row0 >= Ns) return;
  const int cbeg = blockIdx.y * cols_per_chunk, cend = min(Nt, cbeg + cols_per_chunk);
  // A fragments: A[i = li][kk = lk] for step (u,t) of row tile rt is S[row0 + 16rt + li][16u + 4lk + t]
  float4 afrag[RT][KS];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int r = min(row0 + 16 * rt + li, Ns - 1);
#pragma unroll
    for (int u = 0; u < KS; ++u) afrag[rt][u] = *(const float4*)(S + (size_t)r * C + 16 * u + 4 * lk);
  }
  // per lane and row: incumbent (rounded distance bs, column bj) among THIS lane's columns, and a filter flt = the
  // smallest v any of the 16 lanes of the row has seen in earlier tiles.  A candidate with v >= flt cannot have a
  // smaller distance than the row's incumbent, and (columns ascend from tile to tile) loses a distance tie to it, so
  // it is dropped by one compare; without the shared filter every lane would rediscover its own minimum over 1/16 of
  // the columns and the update path (a correctly rounded sqrt) would run on most tiles.
  // The hot compare is on the raw dot product: v = 2 - 2a < flt can only hold when a > thr = (2 - flt)/2 - 1e-6
  // (conservative; the update path re-checks v < flt exactly), one v_cmp per product.
  float flt[RT][4], thr[RT][4], bs[RT][4];
  int bj[RT][4];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      flt[rt][r] = INFINITY; thr[rt][r] = -INFINITY; bs[rt][r] = INFINITY; bj[rt][r] = 0x7fffffff;
    }

  float4 bnext[4][KS];
  auto load_tile = [&](int col0) {
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      const int cj = min(col0 + 16 * nb + li, Nt - 1);
#pragma unroll
      for (int u = 0; u < KS; ++u) bnext[nb][u] = *(const float4*)(T + (size_t)cj * C + 16 * u + 4 * lk);
    }
  };
  if (PF && cbeg < cend) load_tile(cbeg);
  for (int col0 = cbeg; col0 < cend; col0 += kColsPerTile) {
    float4 bfrag[4][KS];
    if (!PF) load_tile(col0);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int u = 0; u < KS; ++u) bfrag[nb][u] = bnext[nb][u];
    if (PF && col0 + kColsPerTile < cend) load_tile(col0 + kColsPerTile);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      f32x4 acc[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < KS; ++u) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[rt][u].x, bfrag[nb][u].x, acc[rt], 0, 0, 0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[rt][u].y, bfrag[nb][u].y, acc[rt], 0, 0, 0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[rt][u].z, bfrag[nb][u].z, acc[rt], 0, 0, 0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[rt][u].w, bfrag[nb][u].w, acc[rt], 0, 0, 0);
      }
      // D layout: col = li (target col0 + 16nb + li), row = 4*lk + r of row tile rt.  Columns arrive in ascending
      // order within a lane, so a strict "<" keeps the earliest column among equals.
      const int j = col0 + 16 * nb + li;
      bool any = false;  // one branch per 16 x 16*RT products instead of one per product
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) any |= acc[rt][r] > thr[rt][r];
      if (any && j < cend) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = 2.0f - 2.0f * acc[rt][r];
            if (v < flt[rt][r]) {
              flt[rt][r] = v;
              thr[rt][r] = (2.0f - v) * 0.5f - 1e-6f;
              const float sv = v < 0.0f ? -INFINITY : __fsqrt_rn(v);
              if (sv < bs[rt][r]) { bs[rt][r] = sv; bj[rt][r] = j; }
            }
          }
      }
    }
    // every other tile: share the filter among the 16 lanes of each row
    if (((col0 - cbeg) / kColsPerTile) & 1) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float f = flt[rt][r];
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) f = fminf(f, __shfl_xor(f, o, 64));
          flt[rt][r] = f;
          thr[rt][r] = (2.0f - f) * 0.5f - 1e-6f;
        }
    }
  }
  // merge: the 16 lanes that share a row, then the column chunks (other workgroups) through a 64-bit atomicMin
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      unsigned long long key = ((unsigned long long)ord_bits(bs[rt][r]) << 32) | (uint32_t)bj[rt][r];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) {
        const unsigned long long ok = __shfl_xor(key, o, 64);
        key = ok < key ? ok : key;
      }
      const int row = row0 + 16 * rt + 4 * lk + r;
      if (li == 0 && row < Ns) atomicMin(&best[row], key);
    }
}

__global__ void mutual_kernel(const int32_t* __restrict__ row_arg, const int32_t* __restrict__ col_arg, int Ns,
                              int32_t* __restrict__ mutual) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < Ns) mutual[i] = col_arg[row_arg[i]] == i ? 1 : 0;
}

template <int C>
int one_pass(const float* S, int Ns, const float* T, int Nt, int32_t* out, unsigned long long* best,
             hipStream_t stream) {
  // Measured on MI355X at 19.1k x 19.1k x 32 (profiles/matching_microbench.py): 2 row tiles per wave without software
  // prefetch and ~2048 workgroups (0.60 ms) beat 4 row tiles / prefetch / 1024 workgroups (0.70 ms) and 1 row tile
  // (0.73 ms): more resident waves hide the fragment loads better than the deeper register tiling saves them.
  constexpr int RT = C <= 64 ? 2 : 1;
  const int rows_per_wg = 64 * RT;
  const int gx = d3f::cdiv(Ns, rows_per_wg);
  int chunks = 2048 / gx;  // target range split so that the launch is ~8 workgroups per CU
  const int max_chunks = d3f::cdiv(Nt, 4 * kColsPerTile);
  if (chunks > max_chunks) chunks = max_chunks;
  if (chunks < 1) chunks = 1;
  int cpc = d3f::cdiv(Nt, chunks);
  cpc = d3f::cdiv(cpc, kColsPerTile) * kColsPerTile;
  chunks = d3f::cdiv(Nt, cpc);
  fill_u64_kernel<<<d3f::cdiv(Ns, 256), 256, 0, stream>>>(best, Ns, ~0ull);
  row_argmin_kernel<C, RT, false><<<dim3(gx, chunks), 256, 0, stream>>>(S, Ns, T, Nt, cpc, best);
  unpack_arg_kernel<<<d3f::cdiv(Ns, 256), 256, 0, stream>>>(best, Ns, out);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

template <int C>
int run(const float* S, int Ns, const float* T, int Nt, int32_t* ra, int32_t* ca, int32_t* mu, void* ws,
        hipStream_t stream) {
  unsigned long long* best = (unsigned long long*)ws;
  int rc = one_pass<C>(S, Ns, T, Nt, ra, best, stream);
  if (rc) return rc;
  rc = one_pass<C>(T, Nt, S, Ns, ca, best, stream);
  if (rc) return rc;
  if (mu) mutual_kernel<<<d3f::cdiv(Ns, 256), 256, 0, stream>>>(ra, ca, Ns, mu);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

}  // namespace

extern "C" {

size_t d3f_mutual_nn_ws_bytes(int Ns, int Nt) { return d3f::align_up(8 * (size_t)(Ns > Nt ? Ns : Nt), 256); }

int d3f_mutual_nn(const float* src_desc, int Ns, const float* tgt_desc, int Nt, int C, int32_t* row_argmin,
                  int32_t* col_argmin, int32_t* mutual, void* ws, size_t ws_bytes, void* stream) {
  if (!src_desc || !tgt_desc || !row_argmin || !col_argmin || !ws || Ns < 1 || Nt < 1) return D3F_EINVAL;
  if (ws_bytes < d3f_mutual_nn_ws_bytes(Ns, Nt)) return D3F_EWORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  switch (C) {
    case 16: return run<16>(src_desc, Ns, tgt_desc, Nt, row_argmin, col_argmin, mutual, ws, s);
    case 32: return run<32>(src_desc, Ns, tgt_desc, Nt, row_argmin, col_argmin, mutual, ws, s);
    case 64: return run<64>(src_desc, Ns, tgt_desc, Nt, row_argmin, col_argmin, mutual, ws, s);
    case 128: return run<128>(src_desc, Ns, tgt_desc, Nt, row_argmin, col_argmin, mutual, ws, s);
    default: return D3F_EINVAL;
  }
}

}  // extern "C"
