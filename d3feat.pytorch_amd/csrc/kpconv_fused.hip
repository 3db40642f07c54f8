// Fused KPConv kernels for gfx950: neighbor gather + influence weights + K-way aggregation + (K*Cin)->Cout
// contraction + neighbor-count normalisation in ONE launch, both matrix products on v_mfma_f32_16x16x4_f32.
//
// Reference semantics: models/blocks.py:277-380 (see kpconv.hip).  Mapping to the hardware:
//
//   workgroup = 4 waves = one tile of 16 queries.  Input channels are walked in super-chunks of CC = 16*CV (<= 64).
//
//   phase A (aggregation, per wave: 4 of the 16 queries, one after the other)
//     MFMA  D[k, c] += A[k, h] * B[h, c]   with  A = influence weights w[q, h, k]  (16 kernel-point rows x 4 neighbors)
//                                                  B = gathered features x[idx[q,h], c] (4 neighbors x 16 channels)
//     lane l = (k = l & 15, hh = l >> 4) computes ITS OWN A element (one sqrt per lane, the kernel point lives in
//     3 VGPRs) and loads ITS OWN B elements as one CV-wide vector, so the weights never touch LDS.  Channel c of
//     MFMA r in column j is  cbase + j*CV + r : a column permutation that makes the x gather a 16*CV-float
//     contiguous row segment per 16 lanes (64..256 B).
//     Supports are read from a packed float4 {x, y, z, [sum_c x > 0]} array (one 16-B load instead of four).
//   the 16 x (K*CC) tile of weighted features goes to LDS (row stride K*CC + 4 floats)
//   phase B (contraction): out[16 x Cout] += wf[16 x K*CC] @ W[k*Cin + c, :]; A fragments are 16-B LDS reads
//     (4 MFMA k-steps each), B fragments stream from L2; waves split the Cout blocks (and the reduction range when
//     Cout < 64, combined through LDS float atomics).  Accumulators stay in registers across channel chunks.
//   epilogue: divide by nn (count of neighbors with positive feature sum) and store.
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
__device__ __forceinline__ float vget<4>(const float4& v, int r) { return r == 0 ? v.x : (r == 1 ? v.y : (r == 2 ? v.z : v.w)); }

// spack[n] = {s.x, s.y, s.z, (sum_c x[n,c] > 0) ? 1 : 0}; one wave per 64/CPW supports
__global__ __launch_bounds__(256) void pack_supports_kernel(const float* __restrict__ s_pts,
                                                            const float* __restrict__ x, int Ns, int Cin,
                                                            float4* __restrict__ spack) {
  // 16 lanes cooperate on one support row
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = t >> 4, l = t & 15;
  float s = 0.0f;
  if (n < Ns)
    for (int c = l; c < Cin; c += 16) s += x[(size_t)n * Cin + c];
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (n < Ns && l == 0)
    spack[n] = make_float4(s_pts[3 * (size_t)n], s_pts[3 * (size_t)n + 1], s_pts[3 * (size_t)n + 2],
                           s > 0.0f ? 1.0f : 0.0f);
}

template <int CV, int NBW, int WK>
__global__ __launch_bounds__(256) void kpconv_fwd_fused_kernel(
    const float* __restrict__ q_pts, const float4* __restrict__ spack, const int32_t* __restrict__ idx,
    const float* __restrict__ x, const float* __restrict__ kp, const float* __restrict__ W, int Nq, int Ns, int H,
    int Cin, int Cout, int K, float extent, float* __restrict__ out, float* __restrict__ nn_out) {
  typedef typename VecT<CV>::type xvec;
  constexpr int CC = 16 * CV;
  constexpr int WN = 4 / WK;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int RS = K * CC + 4;
  float* wf = lds;             // [16][RS]
  float* nn_l = lds + 16 * RS; // [16]
  float* red = nn_l + 16;      // [16][Cout] when WK > 1

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int q0 = blockIdx.x * 16;
  const bool klive = li < K;
  const float kx = klive ? kp[3 * li + 0] : 0.0f, ky = klive ? kp[3 * li + 1] : 0.0f,
              kz = klive ? kp[3 * li + 2] : 0.0f;
  const int wn = (WK == 1) ? wave : (WK == 2 ? (wave & 1) : 0);
  const int wk = (WK == 1) ? 0 : (WK == 2 ? (wave >> 1) : wave);

  f32x4 acc2[NBW];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) acc2[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (WK > 1)
    for (int t = threadIdx.x; t < 16 * Cout; t += 256) red[t] = 0.0f;

  const int nchunks = Cin / CC;
  for (int ch = 0; ch < nchunks; ++ch) {
    const int cbase = ch * CC;
    // ------------------------------------------------------------------ phase A
#pragma unroll 1
    for (int i = 0; i < 4; ++i) {
      const int ql = wave * 4 + i;
      const int q = q0 + ql;
      f32x4 acc[CV];
#pragma unroll
      for (int r = 0; r < CV; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
      float cnt = 0.0f;
      if (q < Nq) {
        const float qx = q_pts[3 * (size_t)q + 0], qy = q_pts[3 * (size_t)q + 1], qz = q_pts[3 * (size_t)q + 2];
        const int32_t* row = idx + (size_t)q * H;
#pragma unroll 4
        for (int h0 = 0; h0 < H; h0 += 4) {
          const int h = h0 + lg;
          const int n = h < H ? row[h] : Ns;
          const bool valid = (unsigned)n < (unsigned)Ns;
          float w = 0.0f;
          xvec xv;
          if (valid) {
            const float4 sp = spack[n];
            const float dx = (sp.x - qx) - kx, dy = (sp.y - qy) - ky, dz = (sp.z - qz) - kz;
            const float d2 = dx * dx + dy * dy + dz * dz;
            w = klive ? fmaxf(0.0f, 1.0f - sqrtf(d2) / extent) : 0.0f;
            xv = *(const xvec*)(x + (size_t)n * Cin + cbase + li * CV);
            cnt += (li == 0) ? sp.w : 0.0f;
          } else {
            xv = xvec();
          }
#pragma unroll
          for (int r = 0; r < CV; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, vget<CV>(xv, r), acc[r], 0, 0, 0);
        }
      }
      // D layout: row k = 4*lg + r2, column j = li  ->  channel cbase + li*CV + r
#pragma unroll
      for (int r2 = 0; r2 < 4; ++r2) {
        const int k = 4 * lg + r2;
        if (k < K) {
          float* dst = wf + ql * RS + k * CC + li * CV;
          if (CV == 1) dst[0] = acc[0][r2];
          if (CV == 2) *(float2*)dst = make_float2(acc[0][r2], acc[CV > 1 ? 1 : 0][r2]);
          if (CV == 4)
            *(float4*)dst = make_float4(acc[0][r2], acc[CV > 1 ? 1 : 0][r2], acc[CV > 2 ? 2 : 0][r2],
                                        acc[CV > 3 ? 3 : 0][r2]);
        }
      }
      if (ch == 0) {
        cnt += __shfl_xor(cnt, 16, 64);
        cnt += __shfl_xor(cnt, 32, 64);
        if (lane == 0) {
          const float v = fmaxf(cnt, 1.0f);
          nn_l[ql] = v;
          if (q < Nq) nn_out[q] = v;
        }
      }
    }
    __syncthreads();
    // ------------------------------------------------------------------ phase B
    const int steps = (K * CC) >> 4;
    for (int s = wk; s < steps; s += WK) {
      const int kc0 = s << 4;
      const float4 a = *(const float4*)(wf + li * RS + kc0 + 4 * lg);
      const int k = kc0 / CC, c0 = kc0 % CC;
      const float* wrow = W + (size_t)(k * Cin + cbase + c0 + 4 * lg) * Cout;
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) {
        const int col = (wn + nb * WN) * 16 + li;
        const float b0 = wrow[col], b1 = wrow[Cout + col], b2 = wrow[2 * Cout + col], b3 = wrow[3 * Cout + col];
        acc2[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b0, acc2[nb], 0, 0, 0);
        acc2[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b1, acc2[nb], 0, 0, 0);
        acc2[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b2, acc2[nb], 0, 0, 0);
        acc2[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b3, acc2[nb], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // ------------------------------------------------------------------ epilogue
  if (WK == 1) {
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
      const int col = (wn + nb * WN) * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rowl = 4 * lg + r;
        if (q0 + rowl < Nq) out[(size_t)(q0 + rowl) * Cout + col] = acc2[nb][r] / nn_l[rowl];
      }
    }
  } else {
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
      const int col = (wn + nb * WN) * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(&red[(4 * lg + r) * Cout + col], acc2[nb][r]);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 16 * Cout; t += 256) {
      const int rowl = t / Cout;
      if (q0 + rowl < Nq) out[(size_t)q0 * Cout + t] = red[t] / nn_l[rowl];
    }
  }
}

bool kpconv_fused_supported(int Cin, int Cout, int K) {
  const bool cin_ok = (Cin == 16 || Cin == 32 || (Cin % 64 == 0 && Cin <= 512));
  const bool cout_ok = (Cout == 16 || Cout == 32 || Cout == 64 || Cout == 128 || Cout == 256 || Cout == 512);
  return cin_ok && cout_ok && K >= 1 && K <= 16;
}

size_t kpconv_fused_ws_bytes(int Ns) { return align_up(sizeof(float4) * (size_t)(Ns > 0 ? Ns : 1), 256); }

template <int CV>
static int launch_fused_cv(const float* q_pts, const float4* spack, const int32_t* idx, const float* x,
                           const float* kp, const float* W, int Nq, int Ns, int H, int Cin, int Cout, int K,
                           float extent, float* out, float* nn_out, hipStream_t stream) {
  const int grid = cdiv(Nq, 16);
  const int CC = 16 * CV;
  const size_t lds_base = sizeof(float) * (size_t)(16 * (K * CC + 4) + 16);
#define D3F_LAUNCH(NBW, WK)                                                                                     \
  {                                                                                                             \
    const size_t lds = lds_base + ((WK) > 1 ? sizeof(float) * 16 * (size_t)Cout : 0);                           \
    kpconv_fwd_fused_kernel<CV, NBW, WK><<<grid, 256, lds, stream>>>(q_pts, spack, idx, x, kp, W, Nq, Ns, H, Cin, \
                                                                      Cout, K, extent, out, nn_out);            \
  }
  switch (Cout) {
    case 16: D3F_LAUNCH(1, 4) break;
    case 32: D3F_LAUNCH(1, 2) break;
    case 64: D3F_LAUNCH(1, 1) break;
    case 128: D3F_LAUNCH(2, 1) break;
    case 256: D3F_LAUNCH(4, 1) break;
    case 512: D3F_LAUNCH(8, 1) break;
    default: return D3F_EINVAL;
  }
#undef D3F_LAUNCH
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

int kpconv_forward_fused(const float* q_pts, int Nq, const float* s_pts, int Ns, const int32_t* idx, int H,
                         const float* x, int Cin, const float* kp, int K, const float* W, int Cout, float extent,
                         float* out, float* nn_out, void* ws, hipStream_t stream) {
  float4* spack = (float4*)ws;
  if (Ns > 0) {
    pack_supports_kernel<<<cdiv((long long)Ns * 16, 256), 256, 0, stream>>>(s_pts, x, Ns, Cin, spack);
    D3F_LAUNCH_CHECK();
  }
  if (Cin == 16) return launch_fused_cv<1>(q_pts, spack, idx, x, kp, W, Nq, Ns, H, Cin, Cout, K, extent, out, nn_out, stream);
  if (Cin == 32) return launch_fused_cv<2>(q_pts, spack, idx, x, kp, W, Nq, Ns, H, Cin, Cout, K, extent, out, nn_out, stream);
  return launch_fused_cv<4>(q_pts, spack, idx, x, kp, W, Nq, Ns, H, Cin, Cout, K, extent, out, nn_out, stream);
}

}  // namespace d3f
