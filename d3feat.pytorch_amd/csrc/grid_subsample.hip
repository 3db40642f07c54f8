// Voxel-grid barycentre subsampling of a stacked batch of clouds.
//
// Replaces reference cpp_wrappers/cpp_subsampling (grid_subsampling.cpp:5-106 per cloud, :109-211 batch loop): the
// points-only branch D3Feat's collate takes, and the feature / label branches of the same call (d3f_grid_subsample_ex).
// Bit-exact contract:
//   origin  = floor(min * (1/dl)) * dl                 float32, per cloud        (grid_subsampling.cpp:25-27)
//   nX, nY  = (size_t)floor((max - origin)/dl) + 1                                (:30-31)
//   key     = iX + nX*iY + nX*nY*iZ,  i* = (size_t)floor((p - origin)/dl)         (:53-56)
//   bary    = (sequential float32 sum of member points IN INPUT ORDER) * (float)(1.0/count)   (:70,87)
//   row order (D3F_ORDER_REFERENCE) = iteration order of libstdc++'s std::unordered_map<size_t,...> filled in
//             point order (:48,59-60,85).
//   feature = (sequential float32 sum of member features in input order) / (float)count        (grid_subsampling.h:50,:89-95)
//   label   = per label column, the value with the most votes among the members; of several with the same count the
//             one std::max_element meets first in the iteration order of the cell's std::unordered_map<int,int> filled
//             in member order (grid_subsampling.h:51-56, .cpp:97-102) -- re-derived with the same list rule as the rows.
//
// Design (CDNA4): no global sort.  Cells live in an open-addressing hash table (atomicCAS on a 64-bit key =
// batch id << 56 | cell key).  Members of a cell are gathered with atomic cursors and re-ordered per cell by
// point index so the float sum is sequential in input order (float atomics would not be reproducible).  Cells
// are ranked by their first point through a bitmap popcount scan.  The unordered_map order is then re-derived
// analytically: with identity hash, a table of Bk buckets lists its nodes as
//     reverse( stable-group-by(bucket = key % Bk, groups ordered by first appearance) )
// of the insertion sequence, and a rehash re-inserts the current list in list order; the (size -> bucket count)
// schedule is probed from the host's own libstdc++ at first use.  One workgroup per cloud runs those phases.
#include <unordered_map>
#include <vector>

#include "common.hpp"

namespace {

constexpr uint64_t kEmpty = ~0ull;
constexpr uint64_t kKeyMask = (1ull << 56) - 1;
constexpr int kOrderThreads = 1024;
constexpr int kMaxPhases = 40;
constexpr int kLdsBuckets = 5120;   // covers libstdc++'s bucket counts up to 5087, i.e. clouds of up to 5087 cells

struct ElemGrid {
  float ox, oy, oz;
  uint64_t nx, nxny;
};

struct Schedule {
  int n;
  int size[kMaxPhases];    // phase j starts when the map already holds size[j] elements
  int bucket[kMaxPhases];  // bucket count during phase j
};

// Probe the local libstdc++ growth policy (grid_subsampling.cpp:48 uses the default-constructed map).
const Schedule& host_schedule(int n_needed) {
  static Schedule s = {0, {0}, {0}};
  static int probed = 0;
  if (n_needed <= probed) return s;
  int target = 1 << 20;
  while (target < n_needed) target <<= 1;
  std::unordered_map<size_t, int> m;
  size_t last = m.bucket_count();
  s.n = 0;
  for (int i = 0; i < target; ++i) {
    m.emplace((size_t)i, i);
    if (m.bucket_count() != last) {
      if (s.n < kMaxPhases) {
        s.size[s.n] = i;
        s.bucket[s.n] = (int)m.bucket_count();
        ++s.n;
      }
      last = m.bucket_count();
    }
  }
  probed = target;
  return s;
}

struct Layout {
  uint32_t M;
  uint64_t* tkey;      // [M]
  int32_t* tcount;     // [M + 64]   (+ allocator at [M])
  int32_t* tfirst;     // [M]
  int32_t* tstart;     // [M]
  int32_t* tfill;      // [M]
  float4* bary;        // [M]
  int32_t* slot_of;    // [N]
  int32_t* members;    // [N]
  uint64_t* bitmap;    // [N/64 + 2]
  int32_t* wprefix;    // [N/64 + 2 + B]
  ElemGrid* grid;      // [B]
  int32_t* ncell;      // [B] (zeroed with tcount)
  uint64_t* seq_key;   // [N]
  int32_t* seq_slot;   // [N]
  int32_t* curA;       // [N]
  int32_t* curB;       // [N]
  int32_t* tmpbk;      // [N]
  int32_t* memberT;    // [N]
  int32_t* bfirst;     // [3N + 1024 B]
  int32_t* bcnt;       // "
  int32_t* bcur;       // "
  int32_t* bbase;      // "
  int32_t* row_slot;   // [N]  table slot of every emitted row (feature / label pass)
  size_t bytes;
};

__host__ __device__ inline uint32_t table_size_for(int N) {
  uint32_t m = 64;
  while (m < 2u * (uint32_t)(N > 0 ? N : 1)) m <<= 1;
  return m;
}

Layout layout(void* ws, int N, int B) {
  Layout L;
  L.M = table_size_for(N);
  d3f::Carver c(ws);
  const size_t n = (size_t)(N > 0 ? N : 1), nw = n / 64 + 2, nb = 3 * n + 1024 * (size_t)B;
  L.tkey = c.take<uint64_t>(L.M);
  L.tcount = c.take<int32_t>(L.M + 64 + (size_t)B);
  L.ncell = L.tcount + L.M + 64;
  L.tfirst = c.take<int32_t>(L.M);
  L.tstart = c.take<int32_t>(L.M);
  L.tfill = c.take<int32_t>(L.M);
  L.bary = c.take<float4>(L.M);
  L.slot_of = c.take<int32_t>(n);
  L.members = c.take<int32_t>(n);
  L.bitmap = c.take<uint64_t>(nw);
  L.wprefix = c.take<int32_t>(nw + B);
  L.grid = c.take<ElemGrid>(B);
  L.seq_key = c.take<uint64_t>(n);
  L.seq_slot = c.take<int32_t>(n);
  L.curA = c.take<int32_t>(n);
  L.curB = c.take<int32_t>(n);
  L.tmpbk = c.take<int32_t>(n);
  L.memberT = c.take<int32_t>(n);
  L.bfirst = c.take<int32_t>(nb);
  L.bcnt = c.take<int32_t>(nb);
  L.bcur = c.take<int32_t>(nb);
  L.bbase = c.take<int32_t>(nb);
  L.row_slot = c.take<int32_t>(n);
  L.bytes = d3f::align_up(c.off, 256);
  return L;
}

// First launch of a call.  Workgroups 0..B-1: component-wise min/max of one cloud each, then its voxel frame.  The
// others re-initialise the call's state -- empty hash table, zeroed counters / bitmap / output rows -- which used to be
// four launches of their own (three fills + init) in front of every level.
__global__ __launch_bounds__(1024) void prep_kernel(const float* __restrict__ p, const int32_t* __restrict__ len, int B,
                                                    float dl, ElemGrid* __restrict__ grid, uint32_t M,
                                                    uint64_t* __restrict__ tkey, int32_t* __restrict__ tfirst,
                                                    int32_t* __restrict__ tcount, size_t n_count,
                                                    uint64_t* __restrict__ bitmap, size_t n_bitmap,
                                                    float* __restrict__ out_points, size_t n_out) {
  __shared__ float smin[3][16], smax[3][16];
  if ((int)blockIdx.x >= B) {
    const size_t stride = (size_t)(gridDim.x - B) * blockDim.x;
    const size_t t0 = (size_t)(blockIdx.x - B) * blockDim.x + threadIdx.x;
    for (size_t i = t0; i < M; i += stride) { tkey[i] = kEmpty; tfirst[i] = 0x7fffffff; }
    for (size_t i = t0; i < n_count; i += stride) tcount[i] = 0;
    for (size_t i = t0; i < n_bitmap; i += stride) bitmap[i] = 0ull;
    // rows past the emitted total stay zero: a capacity-shaped consumer never sees uninitialised (NaN) coordinates
    for (size_t i = t0; i < n_out; i += stride) out_points[i] = 0.0f;
    return;
  }
  const int b = blockIdx.x;
  const int start = d3f::batch_offset(len, b), n = len[b];
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float v = p[3 * (size_t)(start + i) + a];
      mn[a] = fminf(mn[a], v);
      mx[a] = fmaxf(mx[a], v);
    }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float lo = mn[a], hi = mx[a];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      lo = fminf(lo, __shfl_xor(lo, o, 64));
      hi = fmaxf(hi, __shfl_xor(hi, o, 64));
    }
    if (lane == 0) {
      smin[a][wave] = lo;
      smax[a][wave] = hi;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float lo[3], hi[3];
    for (int a = 0; a < 3; ++a) {
      lo[a] = smin[a][0];
      hi[a] = smax[a][0];
      for (int w = 1; w < (int)(blockDim.x >> 6); ++w) {
        lo[a] = fminf(lo[a], smin[a][w]);
        hi[a] = fmaxf(hi[a], smax[a][w]);
      }
    }
    ElemGrid g;
    const float inv = __fdiv_rn(1.0f, dl);
    g.ox = __fmul_rn(floorf(__fmul_rn(lo[0], inv)), dl);
    g.oy = __fmul_rn(floorf(__fmul_rn(lo[1], inv)), dl);
    g.oz = __fmul_rn(floorf(__fmul_rn(lo[2], inv)), dl);
    const uint64_t nx = (uint64_t)floorf(__fdiv_rn(__fsub_rn(hi[0], g.ox), dl)) + 1;
    const uint64_t ny = (uint64_t)floorf(__fdiv_rn(__fsub_rn(hi[1], g.oy), dl)) + 1;
    g.nx = nx;
    g.nxny = nx * ny;
    if (n <= 0) { g.ox = g.oy = g.oz = 0.0f; g.nx = 1; g.nxny = 1; }
    grid[b] = g;
  }
}

__device__ __forceinline__ uint32_t slot_hash(uint64_t k, uint32_t mask) {
  return (uint32_t)((k * 0x9E3779B97F4A7C15ull) >> 32) & mask;
}

__global__ void insert_kernel(const float* __restrict__ p, int N, const int32_t* __restrict__ len, int B, float dl,
                              const ElemGrid* __restrict__ grid, uint32_t mask, uint64_t* __restrict__ tkey,
                              int32_t* __restrict__ tcount, int32_t* __restrict__ tfirst,
                              int32_t* __restrict__ slot_of, int32_t* __restrict__ status) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N || i >= d3f::batch_offset(len, B)) return;  // N is a capacity; sum(len) rows are live
  int b, st;
  d3f::locate_batch(len, B, i, b, st);
  const ElemGrid g = grid[b];
  const uint64_t ix = (uint64_t)floorf(__fdiv_rn(__fsub_rn(p[3 * (size_t)i + 0], g.ox), dl));
  const uint64_t iy = (uint64_t)floorf(__fdiv_rn(__fsub_rn(p[3 * (size_t)i + 1], g.oy), dl));
  const uint64_t iz = (uint64_t)floorf(__fdiv_rn(__fsub_rn(p[3 * (size_t)i + 2], g.oz), dl));
  const uint64_t cell = ix + g.nx * iy + g.nxny * iz;
  if (cell > kKeyMask) atomicOr(status, D3F_ST_CELL_RANGE);
  const uint64_t k = ((uint64_t)b << 56) | (cell & kKeyMask);
  uint32_t s = slot_hash(k, mask);
  bool placed = false;
  for (uint32_t probe = 0; probe <= mask; ++probe) {
    const uint64_t old = atomicCAS((unsigned long long*)&tkey[s], (unsigned long long)kEmpty, (unsigned long long)k);
    if (old == kEmpty || old == k) { placed = true; break; }
    s = (s + 1) & mask;
  }
  if (!placed) { atomicOr(status, D3F_ST_TABLE_FULL); slot_of[i] = -1; return; }
  atomicAdd(&tcount[s], 1);
  atomicMin(&tfirst[s], i);
  slot_of[i] = (int32_t)s;
}

// member-list ranges: the total is one word, so the reservation is aggregated per WORKGROUP (wave scans, the four wave
// totals combined through LDS, ONE atomic by the workgroup) -- a per-cell atomic on a single word serialises (46 us for
// 12k cells), a per-wave one still queues 2048 of them at level 0 (50 us); the per-cloud cell counts likewise
__global__ __launch_bounds__(256) void alloc_kernel(uint32_t M, int B, const uint64_t* __restrict__ tkey,
                                                    int32_t* __restrict__ tcount, const int32_t* __restrict__ tfirst,
                                                    int32_t* __restrict__ tstart, int32_t* __restrict__ tfill,
                                                    uint64_t* __restrict__ bitmap, int32_t* __restrict__ ncell) {
  __shared__ int wtot[4], wbase[4], cells[4][D3F_MAX_BATCH];
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = s < M ? tcount[s] : 0;
  int incl = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wtot[wave] = incl;
  const int b = c > 0 ? (int)(tkey[s] >> 56) : -1;
  for (int k = 0; k < B; ++k) {
    const unsigned long long m = __ballot(b == k);
    if (lane == 0) cells[wave][k] = __popcll(m);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int total = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    const int base = total > 0 ? atomicAdd(&tcount[M], total) : 0;
    wbase[0] = base;
    wbase[1] = base + wtot[0];
    wbase[2] = base + wtot[0] + wtot[1];
    wbase[3] = base + wtot[0] + wtot[1] + wtot[2];
  }
  if ((int)threadIdx.x < B) {
    const int n = cells[0][threadIdx.x] + cells[1][threadIdx.x] + cells[2][threadIdx.x] + cells[3][threadIdx.x];
    if (n) atomicAdd(&ncell[threadIdx.x], n);
  }
  __syncthreads();
  if (c == 0) return;
  const int st = wbase[wave] + incl - c;
  tstart[s] = st;
  tfill[s] = st;
  const int f = tfirst[s];
  atomicOr((unsigned long long*)&bitmap[f >> 6], 1ull << (f & 63));
}

__global__ void scatter_kernel(int N, const int32_t* __restrict__ len, int B, const int32_t* __restrict__ slot_of,
                               int32_t* __restrict__ tfill, int32_t* __restrict__ members) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N || i >= d3f::batch_offset(len, B)) return;
  const int s = slot_of[i];
  if (s < 0) return;
  members[atomicAdd(&tfill[s], 1)] = i;
}

// one thread per occupied cell: order its members by point index, then the reference's sequential sum
__global__ void cell_sum_kernel(uint32_t M, const float* __restrict__ p, const int32_t* __restrict__ tcount,
                                const int32_t* __restrict__ tstart, int32_t* __restrict__ members,
                                float4* __restrict__ bary) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= M) return;
  const int c = tcount[s];
  if (c == 0) return;
  int32_t* m = members + tstart[s];
  // shell sort (gap sequence n/2, n/4, ... 1): members of a voxel are few, but stay O(n log^2 n) for outliers
  for (int gap = c >> 1; gap > 0; gap >>= 1) {
    for (int i = gap; i < c; ++i) {
      const int v = m[i];
      int j = i;
      for (; j >= gap && m[j - gap] > v; j -= gap) m[j] = m[j - gap];
      m[j] = v;
    }
  }
  float sx = 0.0f, sy = 0.0f, sz = 0.0f;
  for (int i = 0; i < c; ++i) {
    const size_t q = 3 * (size_t)m[i];
    sx = __fadd_rn(sx, p[q + 0]);
    sy = __fadd_rn(sy, p[q + 1]);
    sz = __fadd_rn(sz, p[q + 2]);
  }
  const float w = (float)(1.0 / (double)c);
  bary[s] = make_float4(__fmul_rn(sx, w), __fmul_rn(sy, w), __fmul_rn(sz, w), 0.0f);
}

// exclusive block scan of one int per thread (kOrderThreads threads); returns the prefix, total in *total
__device__ int block_exclusive_scan(int v, int* sh /*[16]*/, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  __syncthreads();
  if (lane == 63) sh[wave] = incl;
  __syncthreads();
  int wbase = 0, tot = 0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) {
    const int x = sh[w];
    if (w < wave) wbase += x;
    tot += x;
  }
  *total = tot;
  return wbase + incl - v;
}

__global__ __launch_bounds__(kOrderThreads) void order_kernel(
    const int32_t* __restrict__ len, int B, int max_p, int order, Schedule sched, const uint64_t* __restrict__ bitmap,
    int32_t* __restrict__ wprefix, const int32_t* __restrict__ slot_of, const uint64_t* __restrict__ tkey,
    const float4* __restrict__ bary, const int32_t* __restrict__ ncell, uint64_t* __restrict__ seq_key,
    int32_t* __restrict__ seq_slot, int32_t* __restrict__ curA, int32_t* __restrict__ curB,
    int32_t* __restrict__ tmpbk, int32_t* __restrict__ memberT, int32_t* __restrict__ bfirst,
    int32_t* __restrict__ bcnt, int32_t* __restrict__ bcur, int32_t* __restrict__ bbase,
    float* __restrict__ out_points, int32_t* __restrict__ out_len, int32_t* __restrict__ out_total, int out_cap,
    int32_t* __restrict__ row_slot, int32_t* __restrict__ status, int lds_cap) {
  __shared__ int sh[16];
  extern __shared__ int32_t lds_buckets[];   // 4 x kLdsBuckets: the bucket tables of the phases that fit (LDS atomics)
  const int b = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
  const int pstart = d3f::batch_offset(len, b), n = len[b], pend = pstart + n;
  const int Mc = ncell[b];
  const int limit = max_p < 1 ? 0x7fffffff : max_p;
  int obase = 0, total = 0;
  for (int k = 0; k < B; ++k) {
    const int e = min(ncell[k], limit);
    if (k < b) obase += e;
    total += e;
  }
  const int emit = min(Mc, limit);
  if (tid == 0) {
    // lengths handed downstream never exceed the buffer: on overflow the level is truncated AND flagged
    out_len[b] = max(0, min(emit, out_cap - obase));
    if (b == 0) {
      *out_total = total;
      if (total > out_cap) atomicOr(status, D3F_ST_CAPACITY);
    }
  }
  if (n <= 0 || Mc <= 0) return;

  // ---- rank cells by first point: popcount scan over this cloud's slice of the bitmap
  const int w0 = pstart >> 6, w1 = (pend + 63) >> 6, nw = w1 - w0;
  int32_t* wp = wprefix + w0 + b;  // +b: a boundary word is shared by two clouds
  auto masked_word = [&](int w) -> uint64_t {
    uint64_t v = bitmap[w];
    const int lo = w << 6;
    if (lo < pstart) v &= ~0ull << (pstart - lo);
    if (lo + 64 > pend) v &= (pend - lo) >= 64 ? ~0ull : ((1ull << (pend - lo)) - 1ull);
    return v;
  };
  {
    const int per = (nw + nthr - 1) / nthr;
    const int c0 = min(tid * per, nw), c1 = min(c0 + per, nw);
    int local = 0;
    for (int w = c0; w < c1; ++w) local += __popcll(masked_word(w0 + w));
    int tot;
    int run = block_exclusive_scan(local, sh, &tot);
    for (int w = c0; w < c1; ++w) {
      wp[w] = run;
      run += __popcll(masked_word(w0 + w));
    }
  }
  __syncthreads();
  uint64_t* skey = seq_key + pstart;
  int32_t* sslot = seq_slot + pstart;
  for (int i = pstart + tid; i < pend; i += nthr) {
    const int w = i >> 6;
    const uint64_t word = masked_word(w);
    if ((word >> (i & 63)) & 1ull) {
      const int r = wp[w - w0] + __popcll(word & ((1ull << (i & 63)) - 1ull));
      const int s = slot_of[i];
      skey[r] = tkey[s] & kKeyMask;
      sslot[r] = s;
    }
  }
  __syncthreads();

  int32_t* cur = curA + pstart;
  int32_t* nxt = curB + pstart;
  if (order == D3F_ORDER_REFERENCE) {
    int32_t* tb = tmpbk + pstart;
    int32_t* mt = memberT + pstart;
    const size_t boff = 3 * (size_t)pstart + 1024 * (size_t)b;  // bucket count <= ~2.3 x cells (libstdc++ prime table)
    for (int j = 0; j < sched.n; ++j) {
      const int e0 = sched.size[j];
      if (e0 >= Mc) break;
      const int e1 = (j + 1 < sched.n) ? sched.size[j + 1] : 0x7fffffff;
      const int nj = min(e1, Mc);
      const int Bk = sched.bucket[j];
      // a phase is ~6 barriers with bucket atomics in between: in LDS a phase costs a few us, through L2 ten times that
      const bool in_lds = lds_cap > 0 && Bk <= lds_cap;
      int32_t* bf = in_lds ? lds_buckets : bfirst + boff;
      int32_t* bc = in_lds ? lds_buckets + lds_cap : bcnt + boff;
      int32_t* bu = in_lds ? lds_buckets + 2 * lds_cap : bcur + boff;
      int32_t* bb = in_lds ? lds_buckets + 3 * lds_cap : bbase + boff;
      for (int k = tid; k < Bk; k += nthr) {
        bf[k] = 0x7fffffff;
        bc[k] = 0;
        bu[k] = 0;
      }
      __syncthreads();
      // insertion sequence of this phase: the current list (e0 nodes, list order), then new cells e0..nj-1
      for (int t = tid; t < nj; t += nthr) {
        const int el = t < e0 ? cur[t] : t;
        const int bk = (int)(skey[el] % (uint64_t)Bk);
        tb[t] = bk;
        atomicMin(&bf[bk], t);
        atomicAdd(&bc[bk], 1);
      }
      __syncthreads();
      {  // buckets ordered by first appearance: prefix of bucket sizes at their head positions
        const int per = (nj + nthr - 1) / nthr;
        const int c0 = min(tid * per, nj), c1 = min(c0 + per, nj);
        int local = 0;
        for (int t = c0; t < c1; ++t) {
          const int bk = tb[t];
          if (bf[bk] == t) local += bc[bk];
        }
        int tot;
        int run = block_exclusive_scan(local, sh, &tot);
        for (int t = c0; t < c1; ++t) {
          const int bk = tb[t];
          if (bf[bk] == t) {
            bb[bk] = run;
            run += bc[bk];
          }
        }
      }
      __syncthreads();
      for (int t = tid; t < nj; t += nthr) {
        const int bk = tb[t];
        mt[bb[bk] + atomicAdd(&bu[bk], 1)] = t;
      }
      __syncthreads();
      for (int t = tid; t < nj; t += nthr) {
        const int bk = tb[t];
        const int base = bb[bk], c = bc[bk];
        int rank = 0;
        for (int u = 0; u < c; ++u) rank += mt[base + u] < t;
        const int el = t < e0 ? cur[t] : t;
        nxt[nj - 1 - (base + rank)] = el;  // list = reverse(grouped sequence)
      }
      __syncthreads();
      int32_t* sw = cur; cur = nxt; nxt = sw;
    }
  } else {
    for (int r = tid; r < Mc; r += nthr) cur[r] = r;
    __syncthreads();
  }
  for (int pos = tid; pos < emit; pos += nthr) {
    if (obase + pos >= out_cap) continue;  // capacity overflow is reported through the status word
    const int slot = sslot[cur[pos]];
    const float4 v = bary[slot];
    float* o = out_points + 3 * (size_t)(obase + pos);
    o[0] = v.x; o[1] = v.y; o[2] = v.z;
    if (row_slot) row_slot[obase + pos] = slot;
  }
}

// thread per (emitted row, feature column): the reference's sequential sum over the cell's members, then / (float)count
__global__ void cell_feature_kernel(const int32_t* __restrict__ out_total, int out_cap, int fdim,
                                    const int32_t* __restrict__ row_slot, const int32_t* __restrict__ tcount,
                                    const int32_t* __restrict__ tstart, const int32_t* __restrict__ members,
                                    const float* __restrict__ features, float* __restrict__ out_features) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int rows = min(*out_total, out_cap);
  if (t >= (size_t)rows * fdim) return;
  const int row = (int)(t / fdim), ch = (int)(t % fdim);
  const int s = row_slot[row], c = tcount[s];
  const int32_t* m = members + tstart[s];      // sorted by point index (cell_sum_kernel)
  float acc = 0.0f;
  for (int i = 0; i < c; ++i) acc = __fadd_rn(acc, features[(size_t)m[i] * fdim + ch]);
  out_features[t] = __fdiv_rn(acc, (float)c);
}

// thread per emitted row: label votes.  Scratch (5 ints per member, at the cell's member range): distinct values in
// first-appearance order, their counts, two list buffers and the first-appearance index of every element's bucket.
__global__ void cell_label_kernel(const int32_t* __restrict__ out_total, int out_cap, int ldim, Schedule sched,
                                  const int32_t* __restrict__ row_slot, const int32_t* __restrict__ tcount,
                                  const int32_t* __restrict__ tstart, const int32_t* __restrict__ members,
                                  const int32_t* __restrict__ classes, int32_t* __restrict__ vals_,
                                  int32_t* __restrict__ cnts_, int32_t* __restrict__ listA, int32_t* __restrict__ listB,
                                  int32_t* __restrict__ fst_, int32_t* __restrict__ out_classes) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= min(*out_total, out_cap)) return;
  const int s = row_slot[row], c = tcount[s], base = tstart[s];
  const int32_t* m = members + base;
  int32_t* vals = vals_ + base;
  int32_t* cnts = cnts_ + base;
  int32_t* fst = fst_ + base;
  for (int ld = 0; ld < ldim; ++ld) {
    int D = 0;
    for (int i = 0; i < c; ++i) {                       // labels[ld][value] += 1 in member order (grid_subsampling.h:51-56)
      const int v = classes[(size_t)m[i] * ldim + ld];
      int j = 0;
      while (j < D && vals[j] != v) ++j;
      if (j == D) { vals[D] = v; cnts[D] = 0; ++D; }
      ++cnts[j];
    }
    // iteration order of the unordered_map<int,int>: same list rule as the rows (see order_kernel)
    int32_t* cur = listA + base;
    int32_t* nxt = listB + base;
    for (int j = 0; j < sched.n; ++j) {
      const int e0 = sched.size[j];
      if (e0 >= D) break;
      const int e1 = (j + 1 < sched.n) ? sched.size[j + 1] : 0x7fffffff;
      const int nj = min(e1, D);
      const uint64_t Bk = (uint64_t)sched.bucket[j];
      auto bucket_of = [&](int t) -> uint64_t {
        const int el = t < e0 ? cur[t] : t;
        return (uint64_t)(int64_t)vals[el] % Bk;        // std::hash<int> is the value converted to size_t
      };
      for (int t = 0; t < nj; ++t) {
        const uint64_t bk = bucket_of(t);
        int f = 0;
        while (bucket_of(f) != bk) ++f;
        fst[t] = f;
      }
      for (int t = 0; t < nj; ++t) {
        int g = 0;
        for (int u = 0; u < nj; ++u) g += (fst[u] < fst[t]) || (fst[u] == fst[t] && u < t);
        nxt[nj - 1 - g] = t < e0 ? cur[t] : t;
      }
      int32_t* sw = cur; cur = nxt; nxt = sw;
    }
    int best = cur[0];
    for (int pos = 1; pos < D; ++pos)                   // std::max_element keeps the FIRST of equal maxima (.cpp:100-101)
      if (cnts[cur[pos]] > cnts[best]) best = cur[pos];
    out_classes[(size_t)row * ldim + ld] = vals[best];
  }
}

}  // namespace

extern "C" {

size_t d3f_grid_subsample_ws_bytes(int N, int B) { return layout(nullptr, N, B < 1 ? 1 : B).bytes; }

int d3f_grid_subsample_ex(const float* points, int N, const int32_t* len, int B, float sampleDl, int max_p, int order,
                          const float* features, int fdim, const int32_t* classes, int ldim, float* out_points,
                          int out_cap, int32_t* out_len, int32_t* out_total, float* out_features, int32_t* out_classes,
                          void* ws, size_t ws_bytes, int32_t* status, void* stream_) {
  if (out_cap <= 0) out_cap = N;
  if (!points || !len || !out_points || !out_len || !out_total || !ws || !status || N < 1 || B < 1 ||
      B > D3F_MAX_BATCH || !(sampleDl > 0.0f) || (order != D3F_ORDER_REFERENCE && order != D3F_ORDER_FIRST_SEEN))
    return D3F_EINVAL;
  if ((features && (fdim < 1 || !out_features)) || (classes && (ldim < 1 || !out_classes))) return D3F_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  Layout L = layout(ws, N, B);
  if (ws_bytes < L.bytes) return D3F_EWORKSPACE;
  const Schedule& sched = host_schedule(N);
  const bool attrs = features || classes;
  if (features && d3f::zero_async(out_features, sizeof(float) * (size_t)fdim * out_cap, stream) != hipSuccess)
    return D3F_ELAUNCH;
  if (classes && d3f::zero_async(out_classes, sizeof(int32_t) * (size_t)ldim * out_cap, stream) != hipSuccess)
    return D3F_ELAUNCH;
  {
    int init_blocks = d3f::cdiv(L.M, 1024);
    if (init_blocks > 512) init_blocks = 512;
    prep_kernel<<<B + init_blocks, 1024, 0, stream>>>(points, len, B, sampleDl, L.grid, L.M, L.tkey, L.tfirst, L.tcount,
                                                      (size_t)L.M + 64 + (size_t)B, L.bitmap, (size_t)N / 64 + 2,
                                                      out_points, 3 * (size_t)out_cap);
  }
  insert_kernel<<<d3f::cdiv(N, 256), 256, 0, stream>>>(points, N, len, B, sampleDl, L.grid, L.M - 1, L.tkey, L.tcount,
                                                       L.tfirst, L.slot_of, status);
  alloc_kernel<<<d3f::cdiv(L.M, 256), 256, 0, stream>>>(L.M, B, L.tkey, L.tcount, L.tfirst, L.tstart, L.tfill, L.bitmap,
                                                        L.ncell);
  scatter_kernel<<<d3f::cdiv(N, 256), 256, 0, stream>>>(N, len, B, L.slot_of, L.tfill, L.members);
  cell_sum_kernel<<<d3f::cdiv(L.M, 256), 256, 0, stream>>>(L.M, points, L.tcount, L.tstart, L.members, L.bary);
  // 80 KB of dynamic LDS needs the opt-in, which is a per-DEVICE function attribute: asked for on every call (it is a
  // host-side table update, no stream operation; a process may drive several devices and threads).  Without it the
  // bucket tables stay in global memory.
  int lds_cap = 0;
  {
    const size_t want = 4 * sizeof(int32_t) * (size_t)kLdsBuckets;
    if (hipFuncSetAttribute((const void*)order_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want) == hipSuccess)
      lds_cap = kLdsBuckets;
    else
      (void)hipGetLastError();
  }
  order_kernel<<<B, kOrderThreads, 4 * sizeof(int32_t) * (size_t)lds_cap, stream>>>(
      len, B, max_p, order, sched, L.bitmap, L.wprefix, L.slot_of, L.tkey, L.bary, L.ncell, L.seq_key, L.seq_slot,
      L.curA, L.curB, L.tmpbk, L.memberT, L.bfirst, L.bcnt, L.bcur, L.bbase, out_points, out_len, out_total, out_cap,
      attrs ? L.row_slot : nullptr, status, lds_cap);
  // the order pass is done with its per-point scratch: the label pass reuses five of those arrays
  if (features)
    cell_feature_kernel<<<d3f::cdiv((long long)out_cap * fdim, 256), 256, 0, stream>>>(
        out_total, out_cap, fdim, L.row_slot, L.tcount, L.tstart, L.members, features, out_features);
  if (classes)
    cell_label_kernel<<<d3f::cdiv(out_cap, 64), 64, 0, stream>>>(out_total, out_cap, ldim, sched, L.row_slot, L.tcount,
                                                                L.tstart, L.members, classes, L.curA, L.curB, L.tmpbk,
                                                                L.memberT, L.seq_slot, out_classes);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

int d3f_grid_subsample(const float* points, int N, const int32_t* len, int B, float sampleDl, int max_p, int order,
                       float* out_points, int out_cap, int32_t* out_len, int32_t* out_total, void* ws, size_t ws_bytes,
                       int32_t* status, void* stream_) {
  return d3f_grid_subsample_ex(points, N, len, B, sampleDl, max_p, order, nullptr, 0, nullptr, 0, out_points, out_cap,
                               out_len, out_total, nullptr, nullptr, ws, ws_bytes, status, stream_);
}

}  // extern "C"
