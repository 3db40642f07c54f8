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

}  // namespace

extern "C" {

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
