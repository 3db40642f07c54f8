// Fixed-radius neighbor search on a hashed uniform cell list.
//
// Replaces reference cpp_wrappers/cpp_neighbors (neighbors.cpp:211-333, nanoflann kd-tree on one CPU thread).
// Same result contract: per query the supports of the SAME batch element with d2 < r2 (strict, float32,
// ((dx*dx)+(dy*dy))+(dz*dz) without FMA -- nanoflann.hpp:433-441,249-251), ascending in d2; ties in d2 are
// ordered by support index (the reference leaves them unspecified); row padded with Ns (neighbors.cpp:324).
//
// Design (CDNA4): supports are bucketed by a 64-bit cell key (batch, cx, cy, cz) hashed into a power-of-two
// table; the cell edge is radius*(1+1e-4) and cell coordinates are computed in f64 so a 27-cell scan provably
// covers every accepted support.  A point is accepted only while scanning ITS OWN cell key, so hash collisions
// neither lose nor duplicate candidates.  One 64-lane wave serves one query: lanes 0..26 look up the 27
// buckets, a wave prefix sum flattens their ranges so all 64 lanes test candidates, survivors are compacted
// with ballot/popcount into LDS as (d2 bits << 32 | index) and ranked by a wave-wide bitonic network
// (register shuffles for <= 64 candidates, LDS otherwise).
#include "common.hpp"

namespace {

constexpr double kCellSlack = 1.0 + 1e-4;
constexpr int kCand = 512;       // ranked candidates per query (overflow -> D3F_ST_CAND_OVERFLOW)
constexpr int kQueryWaves = 4;   // queries per workgroup

__host__ __device__ inline uint32_t table_size_for(int Ns) {
  uint32_t m = 64;
  while (m < 2u * (uint32_t)(Ns > 0 ? Ns : 1)) m <<= 1;
  return m;
}

struct GridLayout {
  uint32_t M;
  int32_t* cnt;      // [M + 64]  per-bucket population; cnt[M] is the global range allocator
  int32_t* start;    // [M]
  int32_t* end;      // [M]       fill cursor during the scatter == range end afterwards
  uint64_t* key_tmp; // [Ns]      cell key of support i (input order)
  float4* pts;       // [Ns]      supports in bucket order: x, y, z, bit-cast global index
  uint64_t* key;     // [Ns]      cell key per sorted support
  size_t bytes;
};

GridLayout grid_layout(void* ws, int Ns) {
  GridLayout g;
  g.M = table_size_for(Ns);
  d3f::Carver c(ws);
  const size_t n = (size_t)(Ns > 0 ? Ns : 1);
  g.cnt = c.take<int32_t>(g.M + 64);
  g.start = c.take<int32_t>(g.M);
  g.end = c.take<int32_t>(g.M);
  g.key_tmp = c.take<uint64_t>(n);
  g.pts = c.take<float4>(n);
  g.key = c.take<uint64_t>(n);
  g.bytes = d3f::align_up(c.off, 256);
  return g;
}

__device__ __forceinline__ int cell_coord(float v, double inv_cell) { return (int)floor((double)v * inv_cell); }

__device__ __forceinline__ uint64_t pack_key(int b, int cx, int cy, int cz) {
  return ((uint64_t)(uint32_t)b << 48) | ((uint64_t)(uint32_t)(cx + 32768) << 32) |
         ((uint64_t)(uint32_t)(cy + 32768) << 16) | (uint64_t)(uint32_t)(cz + 32768);
}

__device__ __forceinline__ uint32_t bucket_of(uint64_t key, uint32_t mask) {
  return (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 32) & mask;
}

__global__ void grid_count_kernel(const float* __restrict__ s, int Ns, const int32_t* __restrict__ s_len, int B,
                                  double inv_cell, uint32_t mask, int32_t* __restrict__ cnt,
                                  uint64_t* __restrict__ key_tmp, int32_t* __restrict__ status) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Ns || i >= d3f::batch_offset(s_len, B)) return;  // Ns is a row capacity; sum(s_len) rows are live
  int b, st;
  d3f::locate_batch(s_len, B, i, b, st);
  const int cx = cell_coord(s[3 * i + 0], inv_cell), cy = cell_coord(s[3 * i + 1], inv_cell),
            cz = cell_coord(s[3 * i + 2], inv_cell);
  if (cx < -32767 || cx > 32766 || cy < -32767 || cy > 32766 || cz < -32767 || cz > 32766)
    atomicOr(status, D3F_ST_CELL_RANGE);
  const uint64_t key = pack_key(b, cx, cy, cz);
  key_tmp[i] = key;
  atomicAdd(&cnt[bucket_of(key, mask)], 1);
}

__global__ __launch_bounds__(1024) void grid_alloc_kernel(uint32_t M, int32_t* __restrict__ cnt,
                                                          int32_t* __restrict__ start, int32_t* __restrict__ end) {
  // ranges need to be disjoint, not ordered; the running total is ONE word, so a whole workgroup of 16 waves reserves
  // its buckets together (wave scans, wave totals combined through LDS, one atomic) -- per-bucket atomics on that word
  // serialise, per-wave ones still queued 2048 deep at level 0 (23 us)
  __shared__ int wtot[16], wbase[16];
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = b < M ? cnt[b] : 0;
  int incl = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    int total = 0;
    for (int w = 0; w < 16; ++w) { wbase[w] = total; total += wtot[w]; }
    const int base = total > 0 ? atomicAdd(&cnt[M], total) : 0;
    for (int w = 0; w < 16; ++w) wbase[w] += base;
  }
  __syncthreads();
  if (b >= M) return;
  const int s = c ? wbase[wave] + incl - c : 0;
  start[b] = s;
  end[b] = s;
}

__global__ void grid_scatter_kernel(const float* __restrict__ s, int Ns, const int32_t* __restrict__ s_len, int B,
                                    uint32_t mask, const uint64_t* __restrict__ key_tmp, int32_t* __restrict__ end,
                                    float4* __restrict__ pts, uint64_t* __restrict__ key) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Ns || i >= d3f::batch_offset(s_len, B)) return;
  const uint64_t k = key_tmp[i];
  const int pos = atomicAdd(&end[bucket_of(k, mask)], 1);
  pts[pos] = make_float4(s[3 * i + 0], s[3 * i + 1], s[3 * i + 2], __int_as_float(i));
  key[pos] = k;
}

constexpr int kUpKeys = 32;   // transposed lists: coarse points ranked per fine point (one per lane of half a wave)

struct WaveScratch {
  uint64_t cand[kCand];
  uint64_t nkey[32];
  int pre[32];
  int st[32];
};

// x of lane (l ^ j) for the butterfly distances of a 64-lane wave WITHOUT the LDS crossbar: __shfl_xor compiles to
// ds_bpermute_b32 (an LDS-pipe instruction with its address arithmetic and ~100 cycles of dependent latency) -- the
// 64-key bitonic network of a query is 21 dependent exchanges of a 64-bit key = 42 of them.  Distances 1, 2, 8 are one
// DPP move (quad_perm / row_ror), 4 is two (row_shl / row_shr under complementary bank masks), 16 and 32 are gfx950's
// v_permlane16_swap / v_permlane32_swap (rows of the two operands exchanged; the lane's row bit picks the half).
__device__ __forceinline__ uint32_t lane_xor(uint32_t x, int j, int lane) {
  switch (j) {
    case 1: return (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0xB1, 0xF, 0xF, false);    // quad_perm [1,0,3,2]
    case 2: return (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x4E, 0xF, 0xF, false);    // quad_perm [2,3,0,1]
    case 4: {
      int r = __builtin_amdgcn_update_dpp((int)x, (int)x, 0x104, 0xF, 0x5, false);                  // banks 0, 2 <- lane + 4
      return (uint32_t)__builtin_amdgcn_update_dpp(r, (int)x, 0x114, 0xF, 0xA, false);              // banks 1, 3 <- lane - 4
    }
    case 8: return (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x128, 0xF, 0xF, false);   // row_ror:8
    case 16: {
      const auto p = __builtin_amdgcn_permlane16_swap(x, x, false, false);
      return (lane & 16) ? p[0] : p[1];
    }
    case 32: {
      const auto p = __builtin_amdgcn_permlane32_swap(x, x, false, false);
      return (lane & 32) ? p[0] : p[1];
    }
    default: return __shfl_xor(x, j, 64);
  }
}
__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int m) {
  const int lane = threadIdx.x & 63;
  const uint32_t lo = lane_xor((uint32_t)v, m, lane), hi = lane_xor((uint32_t)(v >> 32), m, lane);
  return ((uint64_t)hi << 32) | lo;
}

__global__ __launch_bounds__(kQueryWaves * 64) void radius_query_kernel(
    const float* __restrict__ q, int Nq, const int32_t* __restrict__ q_len, const int32_t* __restrict__ s_len, int B,
    int Ns, double inv_cell, float r2, uint32_t mask, const int32_t* __restrict__ start,
    const int32_t* __restrict__ end, const float4* __restrict__ pts, const uint64_t* __restrict__ key, int width,
    int32_t* __restrict__ out_idx, int32_t* __restrict__ out_counts, int32_t* __restrict__ max_count,
    int32_t* __restrict__ status, int32_t* __restrict__ out_wide, int wide_width, uint64_t* __restrict__ out_last_key,
    int max_count_group, float r2_prefix, float prune_r, int flag_empty, const int32_t* __restrict__ done_rows,
    int32_t* __restrict__ tr_counts, uint64_t* __restrict__ tr_keys) {
  __shared__ WaveScratch scratch[kQueryWaves];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int qi = blockIdx.x * kQueryWaves + wave;
  if (qi >= Nq) return;  // no workgroup barrier below: waves are independent
  // rows another producer has filled already (d3f_upsample_rows_from_pool: the transpose of the pooling search's lists)
  if (done_rows && done_rows[qi] > 0) return;
  WaveScratch& ws = scratch[wave];
  volatile uint64_t* cand = ws.cand;
  if (qi >= d3f::batch_offset(q_len, B)) {  // Nq is a row capacity: rows past sum(q_len) get an all-shadow row
    if (out_idx)
      for (int c = lane; c < width; c += 64) out_idx[(size_t)qi * width + c] = Ns;
    if (out_wide)
      for (int c = lane; c < wide_width; c += 64) out_wide[(size_t)qi * wide_width + c] = Ns;
    if (lane == 0 && out_counts) out_counts[qi] = 0;
    if (lane == 0 && out_last_key) out_last_key[qi] = ~0ull;
    return;
  }

  int b, qstart;
  d3f::locate_batch(q_len, B, qi, b, qstart);
  const float qx = q[3 * qi + 0], qy = q[3 * qi + 1], qz = q[3 * qi + 2];
  const int cx = cell_coord(qx, inv_cell), cy = cell_coord(qy, inv_cell), cz = cell_coord(qz, inv_cell);

  // lanes 0..26: one neighbor cell each.  A cell whose BOX is farther from the query than the search radius cannot hold
  // an accepted support (a point was put into the cell its coordinates fall in, in the same f64 arithmetic): it is not
  // scanned at all -- a sphere of one cell edge meets ~20.6 of the 27 cells on average, one of 0.75 edge (the prefix
  // form's reach) ~12.  The 1e-4 margin is a thousand times the f32 rounding of d2; results are unchanged bit for bit.
  int len = 0;
  if (lane < 27) {
    const int dx = lane % 3 - 1, dy = (lane / 3) % 3 - 1, dz = lane / 9 - 1;
    const double cell = 1.0 / inv_cell;
    auto gap = [&](float v, int c) -> double {
      const double lo = (double)c * cell, hi = lo + cell, x = (double)v;
      return x < lo ? lo - x : (x > hi ? x - hi : 0.0);
    };
    const double gx = gap(qx, cx + dx), gy = gap(qy, cy + dy), gz = gap(qz, cz + dz);
    const double reach = (double)prune_r * (1.0 + 1e-4);
    const uint64_t nk = pack_key(b, cx + dx, cy + dy, cz + dz);
    const uint32_t bk = bucket_of(nk, mask);
    const int st = start[bk];
    len = (gx * gx + gy * gy + gz * gz <= reach * reach) ? end[bk] - st : 0;
    ws.nkey[lane] = nk;
    ws.st[lane] = st;
  }
  // inclusive prefix over lanes 0..31 (27 cells) by DPP: shifts inside the rows of 16 (out-of-row sources read 0), then
  // row 0's total broadcast into row 1 -- five LDS-crossbar shuffles and their waits gone
  int incl = len;
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, true);    // row_shr:1
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, true);    // row_shr:2
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, true);    // row_shr:4
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, true);    // row_shr:8
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xA, 0xF, false);   // row_bcast:15 into rows 1 and 3
  if (lane < 32) ws.pre[lane] = incl - len;  // exclusive prefix; entries 27..31 == total
  const int P = __builtin_amdgcn_readlane(incl, 31);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // prefix mode (r2_prefix > 0, d3f_radius_query_prefix): only the entries within the PREFIX radius are ranked; of the
  // others the nearest is carried along as a running minimum of the same (d2, index) keys
  const bool prefix = r2_prefix > 0.0f;
  uint64_t nearest = ~0ull;
  int T = 0;
  for (int p0 = 0; p0 < P; p0 += 64) {
    const int p = p0 + lane;
    bool ok = false;
    uint64_t packed = 0;
    if (p < P) {
      int lo = 0;  // largest j with pre[j] <= p
#pragma unroll
      for (int step = 16; step > 0; step >>= 1) {
        const int j = lo + step;
        if (j < 27 && ws.pre[j] <= p) lo = j;
      }
      const int pos = ws.st[lo] + (p - ws.pre[lo]);
      const float4 sp = pts[pos];
      if (key[pos] == ws.nkey[lo]) {
        const float d2 = d3f::sqdist_exact(qx, qy, qz, sp.x, sp.y, sp.z);
        ok = d2 < r2;
        packed = ((uint64_t)__float_as_uint(d2) << 32) | (uint32_t)__float_as_int(sp.w);
        if (prefix) {
          if (ok && packed < nearest) nearest = packed;
          ok = d2 < r2_prefix;          // (r2_prefix <= r2: checked by the launcher)
        }
      }
    }
    const uint64_t m = __ballot(ok);
    if (ok) {
      const int slot = T + __popcll(m & ((1ull << lane) - 1ull));
      if (slot < kCand) cand[slot] = packed;
    }
    T += __popcll(m);
  }
  if (prefix) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint64_t other = shfl_xor_u64(nearest, o);
      nearest = other < nearest ? other : nearest;
    }
    if (T == 0 && nearest != ~0ull) {     // nothing within the prefix radius: the row is the one nearest entry
      if (lane == 0) cand[0] = nearest;
      T = 1;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  if (lane == 0) {
    if (out_counts) out_counts[qi] = T;
    // (the pre-check reads past the CU's L1, which another CU's atomic never refreshes: with a plain load thousands of
    // waves kept seeing the initial 0 and queued their atomics on the one word -- 71 us for 8000 queries, 12 us now)
    if (max_count) {
      int32_t* mc = max_count + (max_count_group > 0 ? b / max_count_group : 0);   // per group of clouds, or one word
      if (T > __hip_atomic_load(mc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(mc, T);
    }
    if (T > kCand) atomicOr(status, D3F_ST_CAND_OVERFLOW);
    // prefix form with a nearest bound: the caller vouched for a support within the bound of every query (the engine:
    // the voxel diagonal).  A row that found none is a broken promise -- e.g. a coarse level that overflowed its
    // capacity and dropped the query's voxel --, not an empty neighborhood: say so instead of handing out a shadow row
    if (flag_empty && T == 0) atomicOr(status, D3F_ST_NO_NEAREST);
    if (out_wide && T > wide_width) atomicOr(status, D3F_ST_WIDE_OVERFLOW);
  }
  const int Tc = T < kCand ? T : kCand;
  int32_t* row = out_idx ? out_idx + (size_t)qi * width : nullptr;
  if (tr_counts) {
    // the TRANSPOSE of this search on the side (d3f_radius_query_pool_transposed): every support s found within the radius
    // gets the key (d2 bits, this query) appended to its own list -- the same pair seen from s, with the same distance bits
    for (int i = lane; i < Tc; i += 64) {
      const uint64_t k = cand[i];
      const uint32_t sidx = (uint32_t)k;
      const int slot = atomicAdd(&tr_counts[sidx], 1);
      // (a list that outgrows its 32 slots is dropped by the ranking kernel and searched for like an empty one)
      if (slot < kUpKeys) tr_keys[(size_t)sidx * kUpKeys + slot] = (k & 0xffffffff00000000ull) | (uint32_t)qi;
    }
  }

  if (Tc <= 64) {
    // rank in registers: 64-key bitonic network over the wave
    uint64_t v = lane < Tc ? cand[lane] : ~0ull;
    // <= 16 (32) keys sit in the first 16 (32) lanes: the network's stages up to that width sort them (the other lane
    // groups hold ~0 only) -- 10 (15) compare-exchange steps instead of 21 for the short rows of the coarse levels and
    // of the prefix form
    const int kmax = Tc <= 16 ? 16 : (Tc <= 32 ? 32 : 64);
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
      if (k <= kmax) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
          const uint64_t o = shfl_xor_u64(v, j);
          const bool up = (lane & k) == 0, lower = (lane & j) == 0;
          const uint64_t mn = v < o ? v : o, mx = v < o ? o : v;
          v = (lower == up) ? mn : mx;
        }
      }
    }
    if (row) {
      if (lane < width) row[lane] = lane < Tc ? (int32_t)(uint32_t)v : Ns;
      for (int c = 64 + lane; c < width; c += 64) row[c] = Ns;
    }
    if (out_wide) {  // the whole ranked list (the reverse-table form of a same-cloud search, see d3feat_hip.h)
      int32_t* wrow = out_wide + (size_t)qi * wide_width;
      if (lane < wide_width) wrow[lane] = lane < Tc ? (int32_t)(uint32_t)v : Ns;
      for (int c = 64 + lane; c < wide_width; c += 64) wrow[c] = Ns;
    }
    if (out_last_key) {  // rank key of the last entry the capped row keeps; ~0 when the row keeps everything
      const uint64_t lk = __shfl((uint32_t)v, width <= 64 ? width - 1 : 63, 64) |
                          ((uint64_t)__shfl((uint32_t)(v >> 32), width <= 64 ? width - 1 : 63, 64) << 32);
      if (lane == 0) out_last_key[qi] = (Tc > width && width <= 64) ? lk : ~0ull;
    }
    return;
  }

  int n = 128;
  while (n < Tc) n <<= 1;
  for (int i = Tc + lane; i < n; i += 64) cand[i] = ~0ull;
  for (int k = 2; k <= n; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = lane; t < (n >> 1); t += 64) {
        const int a = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int c = a | j;
        const bool up = (a & k) == 0;
        const uint64_t va = cand[a], vc = cand[c];
        if ((va > vc) == up) {
          cand[a] = vc;
          cand[c] = va;
        }
      }
    }
  }
  if (row)
    for (int c = lane; c < width; c += 64) row[c] = c < Tc ? (int32_t)(uint32_t)cand[c] : Ns;
  if (out_wide)
    for (int c = lane; c < wide_width; c += 64)
      out_wide[(size_t)qi * wide_width + c] = c < Tc ? (int32_t)(uint32_t)cand[c] : Ns;
  if (out_last_key && lane == 0) out_last_key[qi] = Tc > width ? cand[width - 1] : ~0ull;
}

struct ZeroJobs {
  uint32_t* p[8];
  size_t words[8];
  int n;
};
__global__ void zero_many_kernel(ZeroJobs jobs) {
  const size_t stride = (size_t)gridDim.x * blockDim.x, t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int j = 0; j < jobs.n; ++j)
    for (size_t i = t0; i < jobs.words[j]; i += stride) jobs.p[j][i] = 0u;
}

// ---- upsampling rows as the transpose of the pooling search (d3f_radius_query_pool_transposed + d3f_upsample_rows_rank) ----

// 32 lanes per fine point: its keys ranked by a 32-key bitonic network (15 compare-exchange steps, all inside the half
// wave), the row written in 128-byte runs.  Rows without a key are left alone.
__global__ __launch_bounds__(256) void up_rank_kernel(int32_t* __restrict__ counts, const uint64_t* __restrict__ keys,
                                                      int Nf, int Nc, int width, int32_t* __restrict__ up) {
  const int lane = threadIdx.x & 63, l32 = lane & 31;
  const int f = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + (lane >> 5);
  int n = f < Nf ? counts[f] : 0;
  if (n > kUpKeys) {      // more coarse points around this fine point than a list holds (a volumetric cloud): no row from
    n = 0;                // here, the count goes back to 0 and the masked search that follows ranks the row in full
    if (l32 == 0) counts[f] = 0;
  }
  uint64_t v = l32 < n ? keys[(size_t)f * kUpKeys + l32] : ~0ull;
#pragma unroll
  for (int k = 2; k <= 32; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const uint64_t o = shfl_xor_u64(v, j);
      const bool up_dir = (l32 & k) == 0, lower = (l32 & j) == 0;
      const uint64_t mn = v < o ? v : o, mx = v < o ? o : v;
      v = (lower == up_dir) ? mn : mx;
    }
  }
  if (n > 0) {
    int32_t* row = up + (size_t)f * width;
    for (int c = l32; c < width; c += 32) row[c] = c < n ? (int32_t)(uint32_t)v : Nc;
  }
}

}  // namespace

extern "C" {

size_t d3f_radius_grid_ws_bytes(int Ns) { return grid_layout(nullptr, Ns).bytes; }

/* bytes at the START of a cell-list workspace that d3f_radius_grid_build clears (its bucket counters) */
size_t d3f_radius_grid_zero_bytes(int Ns) { return sizeof(int32_t) * ((size_t)table_size_for(Ns) + 64); }

/* clears up to 8 buffers in ONE launch (a pyramid build's five cell lists + its counters: one launch instead of nine) */
int d3f_zero_buffers(void* const* ptrs, const size_t* bytes, int n, void* stream_) {
  if (n < 0 || n > 8 || (n > 0 && (!ptrs || !bytes))) return D3F_EINVAL;
  ZeroJobs jobs;
  jobs.n = 0;
  size_t most = 0;
  for (int j = 0; j < n; ++j) {
    if (!ptrs[j] || bytes[j] == 0) continue;
    if (((uintptr_t)ptrs[j] & 3) || (bytes[j] & 3)) return D3F_EINVAL;
    jobs.p[jobs.n] = (uint32_t*)ptrs[j];
    jobs.words[jobs.n] = bytes[j] / 4;
    if (jobs.words[jobs.n] > most) most = jobs.words[jobs.n];
    ++jobs.n;
  }
  if (jobs.n == 0) return D3F_OK;
  size_t blocks = (most + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  zero_many_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream_>>>(jobs);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

static int radius_grid_build_impl(const float* supports, int Ns, const int32_t* s_len, int B, float radius, void* grid_ws,
                                  size_t grid_ws_bytes, int32_t* status, void* stream_, bool prezeroed);

int d3f_radius_grid_build(const float* supports, int Ns, const int32_t* s_len, int B, float radius, void* grid_ws,
                          size_t grid_ws_bytes, int32_t* status, void* stream_) {
  return radius_grid_build_impl(supports, Ns, s_len, B, radius, grid_ws, grid_ws_bytes, status, stream_, false);
}

/* the same when the caller has cleared the first d3f_radius_grid_zero_bytes(Ns) bytes of grid_ws already (d3f_zero_buffers) */
int d3f_radius_grid_build_prezeroed(const float* supports, int Ns, const int32_t* s_len, int B, float radius,
                                    void* grid_ws, size_t grid_ws_bytes, int32_t* status, void* stream_) {
  return radius_grid_build_impl(supports, Ns, s_len, B, radius, grid_ws, grid_ws_bytes, status, stream_, true);
}

static int radius_grid_build_impl(const float* supports, int Ns, const int32_t* s_len, int B, float radius, void* grid_ws,
                                  size_t grid_ws_bytes, int32_t* status, void* stream_, bool prezeroed) {
  if (!supports || !s_len || !grid_ws || !status || Ns < 0 || B < 1 || B > D3F_MAX_BATCH || !(radius > 0.0f))
    return D3F_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  GridLayout g = grid_layout(grid_ws, Ns);
  if (grid_ws_bytes < g.bytes) return D3F_EWORKSPACE;
  const double inv_cell = 1.0 / ((double)radius * kCellSlack);
  if (!prezeroed && d3f::zero_async(g.cnt, sizeof(int32_t) * (g.M + 64), stream) != hipSuccess) return D3F_ELAUNCH;
  if (Ns > 0) {
    grid_count_kernel<<<d3f::cdiv(Ns, 256), 256, 0, stream>>>(supports, Ns, s_len, B, inv_cell, g.M - 1, g.cnt,
                                                             g.key_tmp, status);
    D3F_LAUNCH_CHECK();
  }
  grid_alloc_kernel<<<d3f::cdiv(g.M, 1024), 1024, 0, stream>>>(g.M, g.cnt, g.start, g.end);
  D3F_LAUNCH_CHECK();
  if (Ns > 0) {
    grid_scatter_kernel<<<d3f::cdiv(Ns, 256), 256, 0, stream>>>(supports, Ns, s_len, B, g.M - 1, g.key_tmp, g.end, g.pts,
                                                                g.key);
    D3F_LAUNCH_CHECK();
  }
  return D3F_OK;
}

static int radius_query_launch(const void* grid_ws, const float* queries, int Nq, const int32_t* q_len, int Ns,
                               const int32_t* s_len, int B, float grid_radius, float radius, int width,
                               int32_t* out_idx, int32_t* out_counts, int32_t* max_count, int32_t* out_wide,
                               int wide_width, uint64_t* out_last_key, int max_count_group, int32_t* status,
                               void* stream_, float prefix_radius, float nearest_bound,
                               const int32_t* done_rows = nullptr, int32_t* tr_counts = nullptr,
                               uint64_t* tr_keys = nullptr);

int d3f_radius_query_ex(const void* grid_ws, const float* queries, int Nq, const int32_t* q_len, int Ns,
                        const int32_t* s_len, int B, float grid_radius, float radius, int width, int32_t* out_idx,
                        int32_t* out_counts, int32_t* max_count, int32_t* out_wide, int wide_width,
                        uint64_t* out_last_key, int max_count_group, int32_t* status, void* stream_) {
  return radius_query_launch(grid_ws, queries, Nq, q_len, Ns, s_len, B, grid_radius, radius, width, out_idx, out_counts,
                             max_count, out_wide, wide_width, out_last_key, max_count_group, status, stream_, 0.0f, 0.0f);
}

/* Prefix form of a search (the pyramid's upsampling tables inside the training engine): row q = the supports within
 * prefix_radius of q, ranked as in d3f_radius_query -- i.e. the leading part of the row a search with `radius` would
 * produce -- or, when there is none, the single nearest support within `radius`.  Column 0 is the nearest support
 * either way (what closest_pool reads, models/blocks.py:79-91) and the leading part is what the transpose of a pooling
 * table is read from (d3f_reverse_table_filter with rev_radius = prefix_radius); the entries between the two radii,
 * which nothing in the training step reads, are neither ranked nor stored. */
int d3f_radius_query_prefix(const void* grid_ws, const float* queries, int Nq, const int32_t* q_len, int Ns,
                            const int32_t* s_len, int B, float grid_radius, float radius, float prefix_radius,
                            float nearest_bound, int width, int32_t* out_idx, int32_t* status, void* stream_) {
  if (!(prefix_radius > 0.0f) || !(prefix_radius <= radius) || !out_idx) return D3F_EINVAL;
  if (nearest_bound != 0.0f && !(nearest_bound >= prefix_radius && nearest_bound <= radius)) return D3F_EINVAL;
  return radius_query_launch(grid_ws, queries, Nq, q_len, Ns, s_len, B, grid_radius, radius, width, out_idx, nullptr,
                             nullptr, nullptr, 0, nullptr, 0, status, stream_, prefix_radius, nearest_bound);
}

/* d3f_radius_query_prefix for the rows nobody has filled yet: rows with done_rows[q] > 0 are left untouched (see
 * d3f_upsample_rows_from_pool). */
int d3f_radius_query_prefix_missing(const void* grid_ws, const float* queries, int Nq, const int32_t* q_len, int Ns,
                                    const int32_t* s_len, int B, float grid_radius, float radius, float prefix_radius,
                                    float nearest_bound, int width, int32_t* out_idx, const int32_t* done_rows,
                                    int32_t* status, void* stream_) {
  if (!(prefix_radius > 0.0f) || !(prefix_radius <= radius) || !out_idx || !done_rows) return D3F_EINVAL;
  if (nearest_bound != 0.0f && !(nearest_bound >= prefix_radius && nearest_bound <= radius)) return D3F_EINVAL;
  return radius_query_launch(grid_ws, queries, Nq, q_len, Ns, s_len, B, grid_radius, radius, width, out_idx, nullptr,
                             nullptr, nullptr, 0, nullptr, 0, status, stream_, prefix_radius, nearest_bound, done_rows);
}

/* A pooling search (coarse queries over the fine cloud, reference datasets/dataloader.py:141-146) that leaves its TRANSPOSE
 * behind: besides the capped table, its max count(s) and the last kept keys (as d3f_radius_query_ex), every fine point f
 * found within the radius of coarse query c gets the key (d2 bits << 32 | c) appended to tr_keys[32 f ...], tr_counts[f]
 * counting them (cleared by the caller).  The pairs are exactly those of the upsampling search at the same radius seen
 * from the fine side, and d2 is the same bits either way round ((a - b)^2 == (b - a)^2, same summation order):
 * d3f_upsample_rows_rank turns the lists into the engine's upsampling rows.  A list that outgrows its 32 slots keeps
 * counting; the ranking drops it and the row is searched for like an empty one. */
int d3f_radius_query_pool_transposed(const void* grid_ws, const float* queries, int Nq, const int32_t* q_len, int Ns,
                                     const int32_t* s_len, int B, float grid_radius, float radius, int width,
                                     int32_t* out_idx, int32_t* max_count, uint64_t* out_last_key, int max_count_group,
                                     int32_t* tr_counts, uint64_t* tr_keys, int32_t* status, void* stream_) {
  if (!tr_counts || !tr_keys || !out_idx) return D3F_EINVAL;
  return radius_query_launch(grid_ws, queries, Nq, q_len, Ns, s_len, B, grid_radius, radius, width, out_idx, nullptr,
                             max_count, nullptr, 0, out_last_key, max_count_group, status, stream_, 0.0f, 0.0f, nullptr,
                             tr_counts, tr_keys);
}

/* The training engine's upsampling rows (prefix form: the coarse points within the POOLING radius of every fine point,
 * ranked by (d2, index): dataloader.py:147-152 restricted to what closest_pool, models/blocks.py:79-91, and the transposed
 * pooling table read) from the lists d3f_radius_query_pool_transposed left: rows of `up` [Nf, width] (shadow = Nc) with at
 * least one key are ranked and written; rows with counts[f] == 0 -- a fine point whose own voxel's barycentre lies
 * farther than the pooling radius, and the padding rows -- are for d3f_radius_query_prefix_missing; so are the rows
 * whose list outgrew its 32 slots (counts[f] is set back to 0). */
int d3f_upsample_rows_rank(int32_t* counts, const uint64_t* keys, int Nf, int Nc, int width, int32_t* up,
                           void* stream_) {
  if (!counts || !keys || !up || Nf < 0 || Nc < 0 || width < 1) return D3F_EINVAL;
  if (Nf > 0) {
    up_rank_kernel<<<d3f::cdiv(Nf, 8), 256, 0, (hipStream_t)stream_>>>(counts, keys, Nf, Nc, width, up);
    D3F_LAUNCH_CHECK();
  }
  return D3F_OK;
}

static int radius_query_launch(const void* grid_ws, const float* queries, int Nq, const int32_t* q_len, int Ns,
                               const int32_t* s_len, int B, float grid_radius, float radius, int width,
                               int32_t* out_idx, int32_t* out_counts, int32_t* max_count, int32_t* out_wide,
                               int wide_width, uint64_t* out_last_key, int max_count_group, int32_t* status,
                               void* stream_, float prefix_radius, float nearest_bound, const int32_t* done_rows,
                               int32_t* tr_counts, uint64_t* tr_keys) {
  if (!grid_ws || !queries || !q_len || !s_len || (!out_idx && !out_wide) || !status || Nq < 0 || Ns < 0 || B < 1 ||
      max_count_group < 0 ||
      B > D3F_MAX_BATCH || width < 1 || width > kCand || !(radius > 0.0f) || !(grid_radius >= radius) ||
      (out_wide && (wide_width < 1 || wide_width > kCand)))
    return D3F_EINVAL;
  if (Nq == 0) return D3F_OK;
  hipStream_t stream = (hipStream_t)stream_;
  GridLayout g = grid_layout(const_cast<void*>(grid_ws), Ns);
  const double inv_cell = 1.0 / ((double)grid_radius * kCellSlack);  // cells of the list the grid was built with
  // float32 product, like neighbors.cpp:226.  (Prefix form with a nearest bound: nothing beyond the bound is looked at.)
  const float rr = (prefix_radius > 0.0f && nearest_bound > 0.0f) ? nearest_bound : radius;
  const float r2 = rr * rr;
  radius_query_kernel<<<d3f::cdiv(Nq, kQueryWaves), kQueryWaves * 64, 0, stream>>>(
      queries, Nq, q_len, s_len, B, Ns, inv_cell, r2, g.M - 1, g.start, g.end, g.pts, g.key, width, out_idx,
      out_counts, max_count, status, out_wide, wide_width, out_last_key, max_count_group,
      prefix_radius > 0.0f ? prefix_radius * prefix_radius : 0.0f, nearest_bound > 0.0f ? nearest_bound : radius,
      (prefix_radius > 0.0f && nearest_bound > 0.0f) ? 1 : 0, done_rows, tr_counts, tr_keys);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

int d3f_radius_query(const void* grid_ws, const float* queries, int Nq, const int32_t* q_len, const float* supports,
                     int Ns, const int32_t* s_len, int B, float radius, int width, int32_t* out_idx,
                     int32_t* out_counts, int32_t* max_count, int32_t* status, void* stream_) {
  (void)supports;
  return d3f_radius_query_ex(grid_ws, queries, Nq, q_len, Ns, s_len, B, radius, radius, width, out_idx, out_counts,
                             max_count, nullptr, 0, nullptr, 0, status, stream_);
}

}  // extern "C"
