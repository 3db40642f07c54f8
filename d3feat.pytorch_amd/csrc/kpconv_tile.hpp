// Shared pieces of the fused KPConv kernels (forward, grad-weights): the MFMA aggregation of one query.
#pragma once
#include <type_traits>

#include "common.hpp"

namespace d3f {

typedef float f32x4 __attribute__((ext_vector_type(4)));

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

// influence of kernel point (kx,ky,kz) on a neighbor at sp, seen from query (qx,qy,qz): blocks.py:283-336
__device__ __forceinline__ float kp_influence(const float4& sp, float qx, float qy, float qz, float kx, float ky,
                                              float kz, float extent) {
  const float dx = (sp.x - qx) - kx, dy = (sp.y - qy) - ky, dz = (sp.z - qz) - kz;
  const float d2 = dx * dx + dy * dy + dz * dz;
  return fmaxf(0.0f, 1.0f - sqrtf(d2) / extent);
}

// Aggregation of ONE query over one channel chunk of CC = 16*CV channels, by one wave:
//   acc[r][i] (+)= sum_h w[q, h, k = 4*lg + i] * x[idx[q,h], cbase + li*CV + r]
// lane (li = l & 15, lg = l >> 4) produces the A element w[q, h0 + lg, k = li] itself and loads its own B elements
// as one CV-wide vector.  U neighbor groups (4 neighbors each) are fetched per step, branch-free (shadow lanes read
// row 0 and are masked) so all loads of a step are in flight before the first MFMA needs them.
// cnt accumulates spack[n].w (the "neighbor has a positive feature sum" flag) on the li == 0 lanes.
template <int CV>
__device__ __forceinline__ void aggregate_query(const int32_t* __restrict__ row, int H, int Ns,
                                                const float4* __restrict__ spack, const float* __restrict__ x,
                                                int Cin, int cbase, float qx, float qy, float qz, float kx, float ky,
                                                float kz, bool klive, float extent, int li, int lg, f32x4 (&acc)[CV],
                                                float& cnt) {
  typedef typename VecT<CV>::type xvec;
  auto step = [&](int h0, auto ucount) {
    constexpr int U = decltype(ucount)::value;
    int n[U];
    bool valid[U];
    float4 sp[U];
    xvec xv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int h = h0 + 4 * u + lg;
      n[u] = h < H ? row[h] : Ns;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      valid[u] = (unsigned)n[u] < (unsigned)Ns;
      const int nc = valid[u] ? n[u] : 0;
      sp[u] = spack[nc];
      xv[u] = *(const xvec*)(x + (size_t)nc * Cin + cbase + li * CV);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float w = (valid[u] && klive) ? kp_influence(sp[u], qx, qy, qz, kx, ky, kz, extent) : 0.0f;
      cnt += (valid[u] && li == 0) ? sp[u].w : 0.0f;
#pragma unroll
      for (int r = 0; r < CV; ++r)
        acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, valid[u] ? vget<CV>(xv[u], r) : 0.0f, acc[r], 0, 0, 0);
    }
  };
  int h0 = 0;
  for (; h0 + 16 <= H; h0 += 16) step(h0, std::integral_constant<int, 4>());
  const int rest = (H - h0 + 3) >> 2;  // remaining groups of 4 neighbors: 0..4 (13..15 left -> 4), fetched in one step
  if (rest >= 4) step(h0, std::integral_constant<int, 4>());
  else if (rest == 3) step(h0, std::integral_constant<int, 3>());
  else if (rest == 2) step(h0, std::integral_constant<int, 2>());
  else if (rest == 1) step(h0, std::integral_constant<int, 1>());
}

// store the D tile of aggregate_query (rows k = 4*lg + i, column li -> channels li*CV + r) into a [K][CC] LDS row
template <int CV>
__device__ __forceinline__ void store_wf_tile(float* __restrict__ dst_q, int K, int li, int lg, const f32x4 (&acc)[CV]) {
  constexpr int CC = 16 * CV;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = 4 * lg + i;
    if (k < K) {
      float* dst = dst_q + k * CC + li * CV;
      if (CV == 1) dst[0] = acc[0][i];
      if (CV == 2) *(float2*)dst = make_float2(acc[0][i], acc[CV > 1 ? 1 : 0][i]);
      if (CV == 4)
        *(float4*)dst = make_float4(acc[0][i], acc[CV > 1 ? 1 : 0][i], acc[CV > 2 ? 2 : 0][i], acc[CV > 3 ? 3 : 0][i]);
    }
  }
}

}  // namespace d3f
