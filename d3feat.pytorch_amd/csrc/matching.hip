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
// Row AND column arg-min come out of ONE sweep over the S x T tiles (round 4; the reference forms S T^T once,
// common.py:9-13): the products a wave holds for its 32 rows x 16 columns also give, per column, the best of those rows;
// the workgroup's 128 rows meet in an LDS table of (distance bits << 32 | row) keys over its column chunk (ds_min_u64),
// and after the sweep every column of the chunk costs one global 64-bit atomicMin.  The dot product of a (row, column)
// pair is the same bit pattern whichever side asks for it, so both arg-mins see identical distances.
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

// seg != nullptr: blockIdx.z = pair; {S rows, T rows} of the pair are segments of the stacked matrices (seg [P,4] =
// {src_off, src_len, tgt_off, tgt_len}); `best` / `best_col` are indexed by the row of the stacked source / target
// matrix, the winning column / row stays pair-local.
template <int C, int RT, bool PF>
__global__ __launch_bounds__(256) void row_argmin_kernel(const float* __restrict__ S, int Ns,
                                                         const float* __restrict__ T, int Nt, int cols_per_chunk,
                                                         unsigned long long* __restrict__ best,
                                                         unsigned long long* __restrict__ best_col,
                                                         const int32_t* __restrict__ seg = nullptr) {
  static_assert(C % 16 == 0, "descriptor width must be a multiple of 16");
  extern __shared__ unsigned long long colbest[];   // [cols_per_chunk]: best (distance, row) key of this workgroup's rows
  if (seg) {
    const int32_t* e = seg + 4 * blockIdx.z;
    const int so = e[0], sn = e[1], to = e[2], tn = e[3];
    S += (size_t)so * C;
    T += (size_t)to * C;
    best += so;
    best_col += to;
    Ns = sn;
    Nt = tn;
    if (Ns < 1 || Nt < 1) return;
  }
  constexpr int KS = C / 16;  // float4 chunks per lane: reduction index c = 16*u + 4*(lane>>4) + t
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int row0 = (blockIdx.x * 4 + wave) * (16 * RT);
  const int cbeg = blockIdx.y * cols_per_chunk, cend = min(Nt, cbeg + cols_per_chunk);
  if (blockIdx.x * 4 * (16 * RT) >= Ns || cbeg >= cend) return;   // (whole workgroup: no barrier is left waiting)
  for (int t = threadIdx.x; t < cend - cbeg; t += 256) colbest[t] = ~0ull;
  __syncthreads();
  const bool live_wave = row0 < Ns;   // a wave past the last row still takes part in the barriers
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
  for (int col0 = cbeg; live_wave && col0 < cend; col0 += kColsPerTile) {
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
      // (round 4: issuing block nb + 1's products before block nb's row / column bookkeeping was measured slower,
      // 0.443 vs 0.411 ms at 19.1k x 19.1k x 32 -- the second accumulator set costs more than the overlap returns)
      // D layout: col = li (target col0 + 16nb + li), row = 4*lk + r of row tile rt.  Columns arrive in ascending
      // order within a lane, so a strict "<" keeps the earliest column among equals.
      const int j = col0 + 16 * nb + li;
      if (j < cend) {
        // column side: best of this lane's 4*RT rows for column j.  Rows past Ns are copies of row Ns - 1 (clamped
        // loads) with a LARGER index: they tie with the real row and lose to it.  `near` counts products within
        // rounding reach of the maximum: only then can a lower row with a smaller product share its rounded distance.
        float m = -INFINITY;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) m = fmaxf(m, acc[rt][r]);
        // Margin: two products can share the rounded sqrt(2 - 2a) when they differ by less than ~ulp(sqrt v) sqrt v +
        // ulp(v) / 2 <= 3.6e-7 (worst case v ~ 4, a ~ -1: anti-correlated descriptors); 1e-6 leaves a factor of 2.7.
        // `!(a <= mthr)` also counts a NaN product (fmaxf drops NaNs, so m alone does not see one next to finite
        // products): any NaN in the lane takes the slow path, which looks at every candidate on its own.
        const float mthr = m - 1e-6f;
        int ri = 0, near = 0;
#pragma unroll
        for (int rt = RT - 1; rt >= 0; --rt)
#pragma unroll
          for (int r = 3; r >= 0; --r) {
            if (acc[rt][r] == m) ri = 16 * rt + r;
            near += !(acc[rt][r] <= mthr) ? 1 : 0;
          }
        unsigned long long ckey;
        if (near > 1 || !(m == m)) {   // (rare; NaN products: every candidate is looked at)
          ckey = ~0ull;
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (acc[rt][r] > mthr || !(acc[rt][r] == acc[rt][r])) {
                const float v = 2.0f - 2.0f * acc[rt][r];
                const float sv = v < 0.0f ? -INFINITY : __fsqrt_rn(v);
                const unsigned long long k2 =
                    ((unsigned long long)ord_bits(sv) << 32) | (uint32_t)(row0 + 16 * rt + 4 * lk + r);
                ckey = k2 < ckey ? k2 : ckey;
              }
        } else {
          const float v = 2.0f - 2.0f * m;
          const float sv = v < 0.0f ? -INFINITY : __fsqrt_rn(v);
          ckey = ((unsigned long long)ord_bits(sv) << 32) | (uint32_t)(row0 + ri + 4 * lk);
        }
        atomicMin(&colbest[j - cbeg], ckey);   // LDS: the 4 row quads of the wave and the 4 waves meet here
      }
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
      if (live_wave && li == 0 && row < Ns) atomicMin(&best[row], key);
    }
  // the chunk's columns: one global atomicMin each
  __syncthreads();
  for (int t = threadIdx.x; t < cend - cbeg; t += 256) {
    const unsigned long long k2 = colbest[t];
    // most workgroups lose a column to an earlier one: a plain load first (a stale value is an older, LARGER key: it
    // can only cause a redundant atomic, never a skipped one)
    if (k2 != ~0ull && k2 < best_col[cbeg + t]) atomicMin(&best_col[cbeg + t], k2);
  }
}

// best_row [Ns] / best_col [Nt] keys -> arg-mins and the mutual flag (one launch over max(Ns, Nt))
__global__ void unpack_mutual_kernel(const unsigned long long* __restrict__ best_row,
                                     const unsigned long long* __restrict__ best_col, int Ns, int Nt,
                                     int32_t* __restrict__ row_arg, int32_t* __restrict__ col_arg,
                                     int32_t* __restrict__ mutual) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < Nt) col_arg[i] = (int32_t)(best_col[i] & 0xffffffffull);
  if (i < Ns) {
    const int32_t j = (int32_t)(best_row[i] & 0xffffffffull);
    row_arg[i] = j;
    if (mutual) mutual[i] = (j >= 0 && j < Nt && (int32_t)(best_col[j] & 0xffffffffull) == i) ? 1 : 0;
  }
}

// workgroups along the target range: ~2048 in all (8 per CU), column chunks of whole tiles that fit the LDS table
static void chunking(int gx, int Nt, int& chunks, int& cpc) {
  const int target = d3f::tunables().match_wgs > 0 ? d3f::tunables().match_wgs : 2048;
  chunks = target / (gx > 0 ? gx : 1);
  const int max_chunks = d3f::cdiv(Nt, 4 * kColsPerTile);
  if (chunks > max_chunks) chunks = max_chunks;
  const int min_chunks = d3f::cdiv(Nt, 4096);   // 32 KB of keys per workgroup at most
  if (chunks < min_chunks) chunks = min_chunks;
  if (chunks < 1) chunks = 1;
  cpc = d3f::cdiv(Nt, chunks);
  cpc = d3f::cdiv(cpc, kColsPerTile) * kColsPerTile;
  chunks = d3f::cdiv(Nt, cpc);
}

template <int C>
int run(const float* S, int Ns, const float* T, int Nt, int32_t* ra, int32_t* ca, int32_t* mu, void* ws,
        hipStream_t stream) {
  // Measured on MI355X at 19.1k x 19.1k x 32 (profiles/matching_microbench.py): 2 row tiles per wave without software
  // prefetch and ~2048 workgroups beat 4 row tiles / prefetch / 1024 workgroups and 1 row tile: more resident waves
  // hide the fragment loads better than the deeper register tiling saves them.
  constexpr int RT = C <= 64 ? 2 : 1;
  unsigned long long* best_row = (unsigned long long*)ws;
  unsigned long long* best_col = best_row + Ns;
  const int gx = d3f::cdiv(Ns, 64 * RT);
  int chunks, cpc;
  chunking(gx, Nt, chunks, cpc);
  fill_u64_kernel<<<d3f::cdiv(Ns + Nt, 256), 256, 0, stream>>>(best_row, Ns + Nt, ~0ull);
  row_argmin_kernel<C, RT, false><<<dim3(gx, chunks), 256, sizeof(unsigned long long) * (size_t)cpc, stream>>>(
      S, Ns, T, Nt, cpc, best_row, best_col);
  unpack_mutual_kernel<<<d3f::cdiv(Ns > Nt ? Ns : Nt, 256), 256, 0, stream>>>(best_row, best_col, Ns, Nt, ra, ca, mu);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

// which = 0: source segments (seg[4p], seg[4p+1]); 2: target segments
__global__ void fill_seg_kernel(unsigned long long* __restrict__ p, const int32_t* __restrict__ seg, int which,
                                int max_len) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int32_t* e = seg + 4 * blockIdx.y;
  if (i < e[which + 1] && i < max_len) p[e[which] + i] = ~0ull;
}
__global__ void unpack_mutual_seg_kernel(const unsigned long long* __restrict__ best_row,
                                         const unsigned long long* __restrict__ best_col,
                                         const int32_t* __restrict__ seg, int max_s, int max_t,
                                         int32_t* __restrict__ row_arg, int32_t* __restrict__ col_arg,
                                         int32_t* __restrict__ mutual) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int32_t* e = seg + 4 * blockIdx.y;
  if (i < e[3] && i < max_t) col_arg[e[2] + i] = (int32_t)(best_col[e[2] + i] & 0xffffffffull);
  if (i < e[1] && i < max_s) {
    const int32_t j = (int32_t)(best_row[e[0] + i] & 0xffffffffull);
    row_arg[e[0] + i] = j;
    if (mutual) mutual[e[0] + i] = (j >= 0 && j < e[3] && (int32_t)(best_col[e[2] + j] & 0xffffffffull) == i) ? 1 : 0;
  }
}

template <int C>
int run_batched(const float* S, const float* T, const int32_t* seg, int P, int max_s, int max_t, int32_t* ra,
                int32_t* ca, int32_t* mu, void* ws, size_t half, hipStream_t stream) {
  constexpr int RT = C <= 64 ? 2 : 1;
  unsigned long long* best_s = (unsigned long long*)ws;
  unsigned long long* best_t = (unsigned long long*)((char*)ws + half);
  const int gx = d3f::cdiv(max_s, 64 * RT);
  int chunks, cpc;
  chunking(gx * P, max_t, chunks, cpc);   // the P pairs together fill the chip
  fill_seg_kernel<<<dim3(d3f::cdiv(max_s, 256), P), 256, 0, stream>>>(best_s, seg, 0, max_s);
  fill_seg_kernel<<<dim3(d3f::cdiv(max_t, 256), P), 256, 0, stream>>>(best_t, seg, 2, max_t);
  row_argmin_kernel<C, RT, false><<<dim3(gx, chunks, P), 256, sizeof(unsigned long long) * (size_t)cpc, stream>>>(
      S, 0, T, 0, cpc, best_s, best_t, seg);
  unpack_mutual_seg_kernel<<<dim3(d3f::cdiv(max_s > max_t ? max_s : max_t, 256), P), 256, 0, stream>>>(
      best_s, best_t, seg, max_s, max_t, ra, ca, mu);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

// ------------------------------------------------------------------------------------------------
// top-k rows of every cloud by (score, index), ascending: one workgroup per cloud
// ------------------------------------------------------------------------------------------------
constexpr int kTopThreads = 1024;

__device__ __forceinline__ unsigned long long score_key(float s, int i) {
  return ((unsigned long long)ord_bits(s) << 32) | (uint32_t)i;
}

__global__ __launch_bounds__(kTopThreads) void topk_kernel(const float* __restrict__ scores,
                                                           const int32_t* __restrict__ seg, int k,
                                                           int32_t* __restrict__ out) {
  __shared__ unsigned long long keys[D3F_TOPK_MAX];
  __shared__ int hist[256];
  __shared__ unsigned long long prefix_sh;
  __shared__ int want_sh, count_sh;
  const int off = seg[2 * blockIdx.x], n = seg[2 * blockIdx.x + 1];
  const float* sc = scores + off;
  int32_t* o = out + (size_t)blockIdx.x * k;
  const int kk = min(k, max(n, 0));
  for (int i = threadIdx.x; i < k - kk; i += blockDim.x) o[i] = -1;
  if (kk == 0) return;
  // radix select, most significant byte first: the kk-th largest key (keys are distinct: the index is part of them)
  if (threadIdx.x == 0) { prefix_sh = 0ull; want_sh = kk; }
  __syncthreads();
  for (int byte = 7; byte >= 0; --byte) {
    for (int b = threadIdx.x; b < 256; b += blockDim.x) hist[b] = 0;
    __syncthreads();
    const unsigned long long prefix = prefix_sh;
    const unsigned long long hi_mask = byte == 7 ? 0ull : (~0ull << (8 * (byte + 1)));
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const unsigned long long key = score_key(sc[i], i);
      if ((key & hi_mask) == prefix) atomicAdd(&hist[(int)((key >> (8 * byte)) & 0xffull)], 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int want = want_sh, b = 255;
      for (; b > 0; --b) {        // walk down from the largest byte value until the wanted rank falls into a bin
        if (hist[b] >= want) break;
        want -= hist[b];
      }
      want_sh = want;
      prefix_sh = prefix | ((unsigned long long)b << (8 * byte));
    }
    __syncthreads();
  }
  const unsigned long long kth = prefix_sh;   // survivors: key >= kth, exactly kk of them
  if (threadIdx.x == 0) count_sh = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const unsigned long long key = score_key(sc[i], i);
    if (key >= kth) keys[atomicAdd(&count_sh, 1)] = key;
  }
  __syncthreads();
  // rank sort (ascending): kk <= 6144 keys, every thread ranks its own against all (LDS broadcast reads)
  for (int i = threadIdx.x; i < kk; i += blockDim.x) {
    const unsigned long long mine = keys[i];
    int rank = 0;
    for (int j = 0; j < kk; ++j) rank += keys[j] < mine;
    o[k - kk + rank] = (int32_t)(mine & 0xffffffffull);
  }
}

}  // namespace

extern "C" {

size_t d3f_mutual_nn_ws_bytes(int Ns, int Nt) {
  return d3f::align_up(8 * ((size_t)(Ns > 0 ? Ns : 1) + (size_t)(Nt > 0 ? Nt : 1)), 256);
}

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

size_t d3f_mutual_nn_batched_ws_bytes(int src_rows, int tgt_rows) {
  return d3f::align_up(8 * (size_t)(src_rows > 0 ? src_rows : 1), 256) +
         d3f::align_up(8 * (size_t)(tgt_rows > 0 ? tgt_rows : 1), 256);
}

int d3f_mutual_nn_batched(const float* src_desc, int src_rows, const float* tgt_desc, int tgt_rows, const int32_t* seg,
                          int P, int max_src, int max_tgt, int C, int32_t* row_argmin, int32_t* col_argmin,
                          int32_t* mutual, void* ws, size_t ws_bytes, void* stream) {
  if (!src_desc || !tgt_desc || !seg || !row_argmin || !col_argmin || !ws || src_rows < 1 || tgt_rows < 1 || P < 1 ||
      P > 65535 || max_src < 1 || max_tgt < 1)
    return D3F_EINVAL;
  if (ws_bytes < d3f_mutual_nn_batched_ws_bytes(src_rows, tgt_rows)) return D3F_EWORKSPACE;
  const size_t half = d3f::align_up(8 * (size_t)src_rows, 256);
  hipStream_t s = (hipStream_t)stream;
  switch (C) {
    case 16: return run_batched<16>(src_desc, tgt_desc, seg, P, max_src, max_tgt, row_argmin, col_argmin, mutual, ws, half, s);
    case 32: return run_batched<32>(src_desc, tgt_desc, seg, P, max_src, max_tgt, row_argmin, col_argmin, mutual, ws, half, s);
    case 64: return run_batched<64>(src_desc, tgt_desc, seg, P, max_src, max_tgt, row_argmin, col_argmin, mutual, ws, half, s);
    case 128: return run_batched<128>(src_desc, tgt_desc, seg, P, max_src, max_tgt, row_argmin, col_argmin, mutual, ws, half, s);
    default: return D3F_EINVAL;
  }
}

int d3f_topk_scores(const float* scores, int rows, const int32_t* seg, int P, int k, int32_t* out, void* stream) {
  if (!scores || !seg || !out || rows < 1 || P < 1 || k < 1 || k > D3F_TOPK_MAX) return D3F_EINVAL;
  topk_kernel<<<P, kTopThreads, 0, (hipStream_t)stream>>>(scores, seg, k, out);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

}  // extern "C"
