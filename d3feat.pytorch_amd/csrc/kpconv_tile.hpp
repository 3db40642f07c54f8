// Shared pieces of the fused KPConv kernels (forward, grad-weights): the MFMA aggregation of a wave's queries.
#pragma once
#include <type_traits>

#include "common.hpp"

namespace d3f {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2v __attribute__((vector_size(8)));
typedef unsigned u32x4v __attribute__((vector_size(16)));

template <int CV>
struct VecT;
template <>
struct VecT<1> { typedef float type; };
template <>
struct VecT<2> { typedef float2 type; };
template <>
struct VecT<4> { typedef float4 type; };

template <int CV>
__device__ __forceinline__ float vget(const typename VecT<CV>::type& v, int r);
template <>
__device__ __forceinline__ float vget<1>(const float& v, int) { return v; }
template <>
__device__ __forceinline__ float vget<2>(const float2& v, int r) { return r == 0 ? v.x : v.y; }
template <>
__device__ __forceinline__ float vget<4>(const float4& v, int r) {
  return r == 0 ? v.x : (r == 1 ? v.y : (r == 2 ? v.z : v.w));
}

// Raw buffer resources: a load whose byte offset falls outside [0, bytes) returns 0.  The neighbor tables use
// index == Ns for "no neighbor" (reference neighbors.cpp:324), and both the packed supports and the feature matrix
// have exactly Ns rows, so a shadow neighbor reads zeros (the reference concatenates a zero feature row,
// blocks.py:356) without any per-lane validity logic or select.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 buf_load_f4(__amdgpu_buffer_rsrc_t r, unsigned off) {
  const u32x4v v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
  return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
}
template <int CV>
__device__ __forceinline__ typename VecT<CV>::type buf_load_vec(__amdgpu_buffer_rsrc_t r, unsigned off);
template <>
__device__ __forceinline__ float buf_load_vec<1>(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
template <>
__device__ __forceinline__ float2 buf_load_vec<2>(__amdgpu_buffer_rsrc_t r, unsigned off) {
  const u32x2v v = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
  return make_float2(__uint_as_float(v[0]), __uint_as_float(v[1]));
}
template <>
__device__ __forceinline__ float4 buf_load_vec<4>(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return buf_load_f4(r, off);
}

// kernel point of a lane; lanes whose kernel-point index is >= K get a point at "infinity" so their influence
// clamps to exactly 0 without a select in the inner loop.
constexpr float kFarKernelPoint = 1e18f;

// influence of the kernel point on a neighbor at sp; (cx,cy,cz) = query + kernel point (blocks.py:283-336):
//   w = max(0, 1 - sqrt(|sp - q - kp|^2)/extent)
// The correctly rounded sqrtf and division expand to ~10 VALU instructions EACH on gfx950 and made the aggregation
// VALU-bound; the hardware v_sqrt_f32 (1 ulp) and a reciprocal multiply differ from the reference by <= 2 ulp of a
// weight in [0,1] (tests bound the effect at 2e-5 of the output range).
__device__ __forceinline__ float kp_influence(const float4& sp, float cx, float cy, float cz, float inv_extent) {
  const float dx = sp.x - cx, dy = sp.y - cy, dz = sp.z - cz;
  const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
  return fmaxf(0.0f, fmaf(-__builtin_amdgcn_sqrtf(d2), inv_extent, 1.0f));
}

// Aggregation of the FOUR queries a wave owns (q_first .. q_first+3) over one channel chunk of CC = 16*CV channels:
//   acc[r][i] = sum_h w[q, h, k = 4*lg + i] * x[idx[q,h], cbase + li*CV + r]      -> flush(slot, acc) per query
// lane (li = l & 15, lg = l >> 4) produces the A element w[q, h, k = li] of neighbor h = 4*g + lg itself and loads its
// own B elements as one CV-wide vector.  Measured on gfx950 (profiles/ablate_kpconv.py) the phase cost was the SUM of
// its L1 data-path, MFMA and VALU times, so the structure below minimises each and lets them overlap:
//   * per query ONE coalesced load fetches the index row (lane l <- idx[q, l]; H <= 64) and ONE gather fetches the
//     packed support of every neighbor (lane l <- spack[idx[q, l]]): the 16 lanes of an MFMA group no longer pull the
//     same 16 bytes through the texture path 16 times; a lane's neighbor position arrives by wave shuffle;
//   * the feature gathers (the only per-group loads left) of ALL groups of a query are issued back to back before
//     its first MFMA; overlap with the matrix core comes from the other waves of the SIMD (the register budget is
//     kept small for that: a second register set for cross-query prefetch cost one occupancy step and was slower);
//   * gathers are raw buffer loads: shadow / out-of-range indices read zeros (see make_rsrc).
// When `nn_lds` is non-null the per-query neighbor count nn = max(1, #{h : sum_c x[idx[q,h],c] > 0}) (the flag is
// spack[n].w) is also produced from the same support gather.
template <int CV, int NG>
struct QueryGather {
  float4 sp;                              // packed support of neighbor `lane` of the query
  typename VecT<CV>::type xv[NG];         // features of neighbor 4*g + lg, channels cbase + li*CV ..
};

template <int CV, int NSTEPS, typename Flush>
__device__ __forceinline__ void aggregate_wave_n(const int (&nall)[4], const float (&cqx)[4], const float (&cqy)[4],
                                                 const float (&cqz)[4], __amdgpu_buffer_rsrc_t rs_sp,
                                                 __amdgpu_buffer_rsrc_t rs_x, unsigned row_bytes, unsigned col_off,
                                                 float inv_extent, int lg, int lane, float* nn_lds, Flush&& flush) {
  constexpr int NG = 4 * NSTEPS;
  QueryGather<CV, NG> qg[1];  // (a second register set for cross-query prefetch cost an occupancy step: slower)
  auto gather = [&](QueryGather<CV, NG>& d, int nrow) {
    d.sp = buf_load_f4(rs_sp, (unsigned)nrow * 16u);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const unsigned n = (unsigned)__shfl(nrow, 4 * g + lg, 64);  // lanes >= H hold Ns
      d.xv[g] = buf_load_vec<CV>(rs_x, n * row_bytes + col_off);
    }
  };
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    gather(qg[0], nall[i]);
    const QueryGather<CV, NG>& c = qg[0];
    if (nn_lds) {
      const float f = wave_sum(c.sp.w);
      if (lane == 0) nn_lds[i] = fmaxf(f, 1.0f);
    }
    f32x4 acc[CV];
#pragma unroll
    for (int r = 0; r < CV; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      float4 sp;
      sp.x = __shfl(c.sp.x, 4 * g + lg, 64);
      sp.y = __shfl(c.sp.y, 4 * g + lg, 64);
      sp.z = __shfl(c.sp.z, 4 * g + lg, 64);
      const float w = kp_influence(sp, cqx[i], cqy[i], cqz[i], inv_extent);
#pragma unroll
      for (int r = 0; r < CV; ++r)
        acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, vget<CV>(c.xv[g], r), acc[r], 0, 0, 0);
    }
    flush(i, acc);
  }
}

// (query + kernel point, clamped index row) of the wave's four queries
template <int DUMMY = 0>
__device__ __forceinline__ void load_wave_queries(const float* __restrict__ q_pts, const int32_t* __restrict__ idx,
                                                  int q_first, int Nq, int H, int Ns, float kx, float ky, float kz,
                                                  int lane, int (&nall)[4], float (&cqx)[4], float (&cqy)[4],
                                                  float (&cqz)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = q_first + i;
    const bool live = q < Nq;
    const int n = (live && lane < H) ? idx[(size_t)q * H + lane] : Ns;
    nall[i] = (int)min((unsigned)n, (unsigned)Ns);  // negative / oversized entries behave like the shadow index
    const int qs = live ? q : 0;
    cqx[i] = q_pts[3 * (size_t)qs + 0] + kx;
    cqy[i] = q_pts[3 * (size_t)qs + 1] + ky;
    cqz[i] = q_pts[3 * (size_t)qs + 2] + kz;
  }
}

template <int CV, typename Flush>
__device__ __forceinline__ void aggregate_wave(const float* __restrict__ q_pts, const int32_t* __restrict__ idx,
                                               int q_first, int Nq, int H, int Ns, __amdgpu_buffer_rsrc_t rs_sp,
                                               __amdgpu_buffer_rsrc_t rs_x, int Cin, int cbase, float kx, float ky,
                                               float kz, float inv_extent, int lane, float* nn_lds, Flush&& flush) {
  const int li = lane & 15, lg = lane >> 4;
  const unsigned row_bytes = (unsigned)Cin * 4u;
  const unsigned col_off = (unsigned)(cbase + li * CV) * 4u;
  int nall[4];
  float cqx[4], cqy[4], cqz[4];  // query + kernel point, per query
  load_wave_queries(q_pts, idx, q_first, Nq, H, Ns, kx, ky, kz, lane, nall, cqx, cqy, cqz);
  // straight-line bodies per step count (H <= 64 -> 1..4 steps of 16 neighbors)
  const int nsteps = (H + 15) >> 4;
  if (nsteps == 3)
    aggregate_wave_n<CV, 3>(nall, cqx, cqy, cqz, rs_sp, rs_x, row_bytes, col_off, inv_extent, lg, lane, nn_lds, flush);
  else if (nsteps == 2)
    aggregate_wave_n<CV, 2>(nall, cqx, cqy, cqz, rs_sp, rs_x, row_bytes, col_off, inv_extent, lg, lane, nn_lds, flush);
  else if (nsteps == 4)
    aggregate_wave_n<CV, 4>(nall, cqx, cqy, cqz, rs_sp, rs_x, row_bytes, col_off, inv_extent, lg, lane, nn_lds, flush);
  else
    aggregate_wave_n<CV, 1>(nall, cqx, cqy, cqz, rs_sp, rs_x, row_bytes, col_off, inv_extent, lg, lane, nn_lds, flush);
}

// the same with the step count fixed at compile time (the launcher picks the instantiation from H): the registers of
// the widest body (16 gathers in flight per lane) are not reserved for tables that never need them
template <int CV, int NSTEPS, typename Flush>
__device__ __forceinline__ void aggregate_wave_steps(const float* __restrict__ q_pts, const int32_t* __restrict__ idx,
                                                     int q_first, int Nq, int H, int Ns, __amdgpu_buffer_rsrc_t rs_sp,
                                                     __amdgpu_buffer_rsrc_t rs_x, int Cin, int cbase, float kx, float ky,
                                                     float kz, float inv_extent, int lane, float* nn_lds, Flush&& flush) {
  const int li = lane & 15, lg = lane >> 4;
  int nall[4];
  float cqx[4], cqy[4], cqz[4];
  load_wave_queries(q_pts, idx, q_first, Nq, H, Ns, kx, ky, kz, lane, nall, cqx, cqy, cqz);
  aggregate_wave_n<CV, NSTEPS>(nall, cqx, cqy, cqz, rs_sp, rs_x, (unsigned)Cin * 4u, (unsigned)(cbase + li * CV) * 4u,
                               inv_extent, lg, lane, nn_lds, flush);
}

// store one query's D tile (rows k = 4*lg + i, column li -> channels li*CV + r) into a [16][CC] LDS row block.
// The tile always has 16 kernel-point rows (row 15 is exact zeros when K = 15), so the store needs no predicate.
template <int CV>
__device__ __forceinline__ void store_wf_tile(float* __restrict__ dst_q, int li, int lg, const f32x4 (&acc)[CV]) {
  constexpr int CC = 16 * CV;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float* dst = dst_q + (4 * lg + i) * CC + li * CV;
    if (CV == 1) dst[0] = acc[0][i];
    if (CV == 2) *(float2*)dst = make_float2(acc[0][i], acc[CV > 1 ? 1 : 0][i]);
    if (CV == 4)
      *(float4*)dst = make_float4(acc[0][i], acc[CV > 1 ? 1 : 0][i], acc[CV > 2 ? 2 : 0][i], acc[CV > 3 ? 3 : 0][i]);
  }
}

}  // namespace d3f
