// Influence / aggregation modes of KPConv (reference models/blocks.py:327-352), shared by the general-path kernels
// (kpconv.hip) and the deformable ones (kpconv_deform.hip).
#pragma once
#include "common.hpp"

namespace d3f {

// Influence of kernel point `kk` (one per lane of a 16-lane group) on a neighbor at squared distance d2, for the
// modes of blocks.py:327-352.  mode bits 0-1: 0 'linear' max(0, 1 - d/extent), 1 'constant' 1, 2 'gaussian'
// exp(-d2 / gauss_denom) with gauss_denom = 2 (0.3 extent)^2 + 1e-9 (blocks.py:66-73,341-342); bit 2 ('closest'
// aggregation, :348-350): only the kernel point nearest to the neighbor keeps its weight (first index on ties).
__device__ __forceinline__ float influence_weight(float d2, bool klive, int kk, float extent, float gauss_denom,
                                                  int mode) {
  float w;
  switch (mode & 3) {
    case 1: w = 1.0f; break;
    case 2: w = expf(-__fdiv_rn(d2, gauss_denom)); break;
    default: w = fmaxf(0.0f, 1.0f - __fdiv_rn(__fsqrt_rn(d2), extent)); break;
  }
  if (!klive) w = 0.0f;
  if (mode & 4) {
    float bd = klive ? d2 : __builtin_huge_valf();
    int bk = kk;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      const float od = __shfl_xor(bd, o, 64);
      const int ok = __shfl_xor(bk, o, 64);
      if (od < bd || (od == bd && ok < bk)) { bd = od; bk = ok; }
    }
    if (kk != bk) w = 0.0f;
  }
  return w;
}

static inline float gauss_denominator(float extent) {
  const double sigma = 0.3 * (double)extent;
  return (float)(2.0 * sigma * sigma + 1e-9);
}

// mode = influence (0 'linear', 1 'constant', 2 'gaussian') | 4 when aggregation_mode == 'closest'.
static inline bool kpconv_mode_ok(int mode) { return mode >= 0 && mode < 8 && (mode & 3) != 3; }

}  // namespace d3f
