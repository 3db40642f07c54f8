// Reverse neighbor tables: the transpose of a neighbor table idx [Nq, H] (reference datasets/dataloader.py:52-67 builds
// idx; the reference never needs its transpose -- autograd's index_add does the scatter) in CSR form,
//     rev_ent[rev_ptr[s] .. rev_ptr[s+1])  =  { q : idx[q, h] = s for some h },  ascending q,
// so that the KPConv grad-input becomes a gather (kpconv_dx_gather.hip).  Built once per pyramid table on the side
// stream, next to the radius searches.  Deterministic: per-row counts come from integer atomics (order-free), the
// slots inside a row are claimed in arbitrary order and then every row is sorted by rank (equal entries -- a table
// that lists a support twice in one row -- are all kept).
//   1 count   thread per table entry: atomicAdd(cnt[idx[e]], 1)
//   2 scan    one workgroup of 16 waves: exclusive scan of cnt -> rev_ptr, cnt reset to 0 (it becomes the cursor)
//   3 fill    thread per table entry: tmp[rev_ptr[s] + atomicAdd(cnt[s], 1)] = q
//   4 sort    one wave per row: rank of every entry among the row's entries -> rev_ent
#include "common.hpp"

namespace {

__global__ __launch_bounds__(256) void rev_count_kernel(const int32_t* __restrict__ idx, size_t n_entries, int Ns,
                                                        int32_t* __restrict__ cnt) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_entries) return;
  const int s = idx[e];
  if ((unsigned)s < (unsigned)Ns) atomicAdd(&cnt[s], 1);
}

// exclusive scan of cnt[0..Ns) into ptr[0..Ns]; cnt is cleared for its second life as the per-row fill cursor.
// One workgroup of 16 waves; wave w owns the contiguous range [w*span, (w+1)*span) and walks it 64 elements at a time
// (coalesced), scanning each 64-element tile with wave shuffles: pass 1 sums the ranges, the 16 range totals are
// scanned through LDS, pass 2 writes the prefix sums.
__device__ __forceinline__ int wave_inclusive_scan(int v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int u = __shfl_up(v, o, 64);
    if (lane >= o) v += u;
  }
  return v;
}

__global__ __launch_bounds__(1024) void rev_scan_kernel(int32_t* __restrict__ cnt, int Ns, int32_t* __restrict__ ptr) {
  __shared__ int tot[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int span = ((Ns + 15) / 16 + 63) / 64 * 64;  // multiple of 64 so tiles never straddle two waves
  const int b = wave * span, e = min(Ns, b + span);
  int s = 0;
  for (int i = b + lane; i < e; i += 64) s += cnt[i];
  s = d3f::wave_sum_i(s);
  if (lane == 0) tot[wave] = s;
  __syncthreads();
  int run = 0;
  for (int w = 0; w < wave; ++w) run += tot[w];
  for (int i0 = b; i0 < e; i0 += 64) {
    const int i = i0 + lane;
    const int c = i < e ? cnt[i] : 0;
    const int inc = wave_inclusive_scan(c, lane);
    if (i < e) {
      ptr[i] = run + inc - c;
      cnt[i] = 0;
    }
    run += __shfl(inc, 63, 64);
  }
  if (threadIdx.x == 1023) {
    int all = 0;
    for (int w = 0; w < 16; ++w) all += tot[w];
    ptr[Ns] = all;
  }
}

__global__ __launch_bounds__(256) void rev_fill_kernel(const int32_t* __restrict__ idx, size_t n_entries, int H, int Ns,
                                                       const int32_t* __restrict__ ptr, int32_t* __restrict__ cur,
                                                       int32_t* __restrict__ tmp) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_entries) return;
  const int s = idx[e];
  if ((unsigned)s >= (unsigned)Ns) return;
  const int pos = atomicAdd(&cur[s], 1);
  tmp[ptr[s] + pos] = (int)(e / (size_t)H);
}

// one wave per row; rows of <= 64 entries are ranked with shuffles, longer ones by re-reading the row
__global__ __launch_bounds__(256) void rev_sort_kernel(const int32_t* __restrict__ ptr, const int32_t* __restrict__ tmp,
                                                       int Ns, int32_t* __restrict__ ent) {
  const int lane = threadIdx.x & 63;
  const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= Ns) return;
  const int beg = ptr[s], len = ptr[s + 1] - beg;
  if (len <= 64) {
    const int v = lane < len ? tmp[beg + lane] : 0x7fffffff;
    int rank = 0;
    for (int j = 0; j < len; ++j) {
      const int u = __shfl(v, j, 64);
      rank += (u < v || (u == v && j < lane)) ? 1 : 0;  // ties (a table that lists a support twice) keep both entries
    }
    if (lane < len) ent[beg + rank] = v;
  } else {
    for (int i = lane; i < len; i += 64) {
      const int v = tmp[beg + i];
      int rank = 0;
      for (int j = 0; j < len; ++j) {
        const int u = tmp[beg + j];
        rank += (u < v || (u == v && j < i)) ? 1 : 0;
      }
      ent[beg + rank] = v;
    }
  }
}

// Search form -> exact form.  Row s of a search-form transpose lists every query point within the search radius of s,
// ranked; q really lists s iff key(q, s) = (d2 bits << 32 | s) <= last_key[q] (s survived q's truncation to the table
// width) and, when the row comes from a search with a larger radius (the 2r upsampling table standing in for the
// transpose of a pooling table), d2 < r2.  One wave per row evaluates that once, here, in the pyramid build (side
// stream: hidden under the previous pair's network step), and leaves the surviving entries compacted, in rank order,
// as float4 {q - s, q's index bits}: the grad-input kernel then reads its neighborhood as ONE coalesced kilobyte per
// row -- no position / key gathers (5 scattered 4..8-byte requests per entry before), no membership test, no
// compaction on the training stream.  Rows end with entries whose index is Nq.
__global__ __launch_bounds__(256) void rev_filter_kernel(const int32_t* __restrict__ ent, int W,
                                                         const uint64_t* __restrict__ last_key,
                                                         const float* __restrict__ q_pts, int Nq,
                                                         const float* __restrict__ s_pts, int Ns, float r2,
                                                         float4* __restrict__ out, int32_t* __restrict__ status) {
  const int lane = threadIdx.x & 63;
  const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= Ns) return;
  const float sx = s_pts[3 * (size_t)s], sy = s_pts[3 * (size_t)s + 1], sz = s_pts[3 * (size_t)s + 2];
  const float4 none = make_float4(0.f, 0.f, 0.f, __int_as_float(Nq));
  int kept = 0;
  for (int c0 = 0; c0 < W; c0 += 64) {
    int n = (c0 + lane < W) ? ent[(size_t)s * W + c0 + lane] : Nq;
    n = min(max(n, 0), Nq);
    const bool real = n < Nq;
    if (__ballot(real) == 0ull) break;   // ranked rows: shadow entries come last
    const int nn = real ? n : 0;
    const float qx = q_pts[3 * (size_t)nn], qy = q_pts[3 * (size_t)nn + 1], qz = q_pts[3 * (size_t)nn + 2];
    const float d2 = d3f::sqdist_exact(qx, qy, qz, sx, sy, sz);
    const uint64_t key = ((uint64_t)__float_as_uint(d2) << 32) | (unsigned)s;
    const bool member = real && key <= last_key[nn] && (r2 <= 0.0f || d2 < r2);
    // a full row of the wider search whose LAST entry is still within r: members may have been cut off
    if (r2 > 0.0f && status && c0 + lane == W - 1 && real && d2 < r2) atomicOr(status, D3F_ST_WIDE_OVERFLOW);
    const uint64_t m = __ballot(member);
    const int rank = __popcll(m & ((1ull << lane) - 1ull));
    if (member) out[(size_t)s * W + kept + rank] = make_float4(qx - sx, qy - sy, qz - sz, __int_as_float(n));
    kept += __popcll(m);
  }
  for (int i = kept + lane; i < W; i += 64) out[(size_t)s * W + i] = none;
}

}  // namespace

extern "C" {

int d3f_reverse_table_filter(const int32_t* rev_ent, int rev_width, const uint64_t* rev_last_key, const float* q_pts,
                             int Nq, const float* s_pts, int Ns, float rev_radius, float* rev_rel_out, int32_t* status,
                             void* stream) {
  if (!rev_ent || !rev_last_key || !q_pts || !s_pts || !rev_rel_out || rev_width < 1 || Nq < 0 || Ns < 0)
    return D3F_EINVAL;
  if ((double)Ns * rev_width * 16.0 >= 4294967295.0 * 4.0) return D3F_EINVAL;
  if (Ns == 0) return D3F_OK;
  const float r2 = rev_radius > 0.0f ? rev_radius * rev_radius : 0.0f;  // float32 product, like the search
  rev_filter_kernel<<<d3f::cdiv(Ns, 4), 256, 0, (hipStream_t)stream>>>(rev_ent, rev_width, rev_last_key, q_pts, Nq, s_pts,
                                                                       Ns, r2, (float4*)rev_rel_out, status);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

size_t d3f_reverse_table_ws_bytes(int Nq, int H, int Ns) {
  if (Nq < 0 || H < 1 || Ns < 0) return 0;
  return d3f::align_up(sizeof(int32_t) * (size_t)(Ns + 1), 256) + d3f::align_up(sizeof(int32_t) * (size_t)Nq * H + 4, 256);
}

int d3f_reverse_table_build(const int32_t* idx, int Nq, int H, int Ns, int32_t* rev_ptr, int32_t* rev_ent, void* ws,
                            size_t ws_bytes, void* stream) {
  if (!idx || !rev_ptr || !rev_ent || !ws || Nq < 0 || H < 1 || Ns < 0) return D3F_EINVAL;
  if (ws_bytes < d3f_reverse_table_ws_bytes(Nq, H, Ns)) return D3F_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  d3f::Carver carve(ws);
  int32_t* cnt = carve.take<int32_t>((size_t)Ns + 1);
  int32_t* tmp = carve.take<int32_t>((size_t)Nq * H + 1);
  const size_t n = (size_t)Nq * H;
  if (d3f::zero_async(cnt, sizeof(int32_t) * ((size_t)Ns + 1), st) != hipSuccess) return D3F_ELAUNCH;
  if (n) rev_count_kernel<<<d3f::cdiv((long long)n, 256), 256, 0, st>>>(idx, n, Ns, cnt);
  rev_scan_kernel<<<1, 1024, 0, st>>>(cnt, Ns, rev_ptr);
  if (n) {
    rev_fill_kernel<<<d3f::cdiv((long long)n, 256), 256, 0, st>>>(idx, n, H, Ns, rev_ptr, cnt, tmp);
    if (Ns) rev_sort_kernel<<<d3f::cdiv(Ns, 4), 256, 0, st>>>(rev_ptr, tmp, Ns, rev_ent);
  }
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

}  // extern "C"
