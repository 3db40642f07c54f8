// Dense mutual-nearest-neighbor descriptor matching on the f32 matrix cores.
//
// Replaces reference geometric_registration/common.py:5-21 (build_correspondence):
//   distance = sqrt(2 - 2 * S @ T.T); source_idx = argmin(axis=1); target_idx = argmin(axis=0); keep mutual pairs.
// At N = 19.3k the reference's distance matrix is 1.5 GB; it is never formed here.  One workgroup owns 64 source
// rows (16 per wave, A fragments resident in registers for the whole sweep) and streams target tiles of 64
// descriptors; each 16x16 product tile comes out of v_mfma_f32_16x16x4_f32 (exact f32 FMA chains) and is folded
// into a running per-row (value, index) minimum.  The column arg-min is the same kernel with S and T swapped, so
// no atomics and no cross-workgroup reduction are needed.  argmin ties -> lowest index (np.argmin); a negative
// 2-2s (NaN after the reference's sqrt) wins like NaN does in np.argmin (first NaN).
#include "common.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kRowsPerWG = 64;
constexpr int kColsPerTile = 64;

// key ordering: NaN-equivalents (v < 0) first, then v ascending, then index ascending
__device__ __forceinline__ bool better(float v, int j, float bv, int bj) {
  return v < bv || (v == bv && j < bj);
}

template <int C>
__global__ __launch_bounds__(256) void row_argmin_kernel(const float* __restrict__ S, int Ns,
                                                         const float* __restrict__ T, int Nt,
                                                         int32_t* __restrict__ out_arg) {
  static_assert(C % 16 == 0, "descriptor width must be a multiple of 16");
  constexpr int KS = C / 16;  // float4 chunks per lane: reduction index c = 16*u + 4*(lane>>4) + t
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int row0 = blockIdx.x * kRowsPerWG + wave * 16;
  if (row0 >= Ns) return;
  // A fragments: A[i = li][kk = lk] for step (u,t) is S[row0+li][16u + 4lk + t]
  float4 afrag[KS];
  {
    const int r = min(row0 + li, Ns - 1);
#pragma unroll
    for (int u = 0; u < KS; ++u) afrag[u] = *(const float4*)(S + (size_t)r * C + 16 * u + 4 * lk);
  }
  float bv[4];
  int bj[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { bv[r] = INFINITY; bj[r] = 0x7fffffff; }

  for (int col0 = 0; col0 < Nt; col0 += kColsPerTile) {
    f32x4 acc[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      acc[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int cj = min(col0 + 16 * nb + li, Nt - 1);
      float4 bfrag[KS];
#pragma unroll
      for (int u = 0; u < KS; ++u) bfrag[u] = *(const float4*)(T + (size_t)cj * C + 16 * u + 4 * lk);
#pragma unroll
      for (int u = 0; u < KS; ++u) {
        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[u].x, bfrag[u].x, acc[nb], 0, 0, 0);
        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[u].y, bfrag[u].y, acc[nb], 0, 0, 0);
        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[u].z, bfrag[u].z, acc[nb], 0, 0, 0);
        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[u].w, bfrag[u].w, acc[nb], 0, 0, 0);
      }
    }
    // D layout: col = li (target col0 + 16nb + li), row = 4*lk + r
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      const int j = col0 + 16 * nb + li;
      if (j < Nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = 2.0f - 2.0f * acc[nb][r];
          v = v < 0.0f ? -INFINITY : v;
          if (better(v, j, bv[r], bj[r])) { bv[r] = v; bj[r] = j; }
        }
      }
    }
  }
  // combine the 16 lanes that share a row (same lk, different li)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv[r], o, 64);
      const int oj = __shfl_xor(bj[r], o, 64);
      if (better(ov, oj, bv[r], bj[r])) { bv[r] = ov; bj[r] = oj; }
    }
    const int row = row0 + 4 * lk + r;
    if (li == 0 && row < Ns) out_arg[row] = bj[r];
  }
}

__global__ void mutual_kernel(const int32_t* __restrict__ row_arg, const int32_t* __restrict__ col_arg, int Ns,
                              int32_t* __restrict__ mutual) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < Ns) mutual[i] = col_arg[row_arg[i]] == i ? 1 : 0;
}

template <int C>
int run(const float* S, int Ns, const float* T, int Nt, int32_t* ra, int32_t* ca, int32_t* mu, hipStream_t stream) {
  row_argmin_kernel<C><<<d3f::cdiv(Ns, kRowsPerWG), 256, 0, stream>>>(S, Ns, T, Nt, ra);
  row_argmin_kernel<C><<<d3f::cdiv(Nt, kRowsPerWG), 256, 0, stream>>>(T, Nt, S, Ns, ca);
  if (mu) mutual_kernel<<<d3f::cdiv(Ns, 256), 256, 0, stream>>>(ra, ca, Ns, mu);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

}  // namespace

extern "C" {

int d3f_mutual_nn(const float* src_desc, int Ns, const float* tgt_desc, int Nt, int C, int32_t* row_argmin,
                  int32_t* col_argmin, int32_t* mutual, void* stream) {
  if (!src_desc || !tgt_desc || !row_argmin || !col_argmin || Ns < 1 || Nt < 1) return D3F_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  switch (C) {
    case 16: return run<16>(src_desc, Ns, tgt_desc, Nt, row_argmin, col_argmin, mutual, s);
    case 32: return run<32>(src_desc, Ns, tgt_desc, Nt, row_argmin, col_argmin, mutual, s);
    case 64: return run<64>(src_desc, Ns, tgt_desc, Nt, row_argmin, col_argmin, mutual, s);
    case 128: return run<128>(src_desc, Ns, tgt_desc, Nt, row_argmin, col_argmin, mutual, s);
    default: return D3F_EINVAL;
  }
}

}  // extern "C"
