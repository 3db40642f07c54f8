// Neighbor max-pool and nearest ("closest") pool, forward + backward.
//
// Replaces reference models/blocks.py:94-110 (max_pool: cat zero shadow row, gather [n2,H,d], torch.max over H)
// and :79-91 (closest_pool: gather column 0).  The reference materialises the gathered [n2,H,d] tensor; here a
// thread owns one (query, 4 channels) group (one channel when C % 4 != 0); consecutive lanes own consecutive channel
// groups, so every neighbor row is read as coalesced 16-B loads and the neighbor index is (nearly) wave-uniform.
// Element indices are 32-bit whenever the matrices have fewer than 2^31 elements: a 64-bit division by a run-time C per
// thread costs more than the loads of these kernels (closest_pool forward at 38k x 128: 47 -> 15 us).
#include "common.hpp"

namespace {

// argmax convention: first maximal neighbor in row order, like torch.max(dim=1) on CPU/ROCm.
// `clear` (optional, Ns*C floats): the scatter target of the backward pass, zeroed here on the side so that the
// backward needs no fill launch of its own.
__global__ void max_pool_fwd_kernel(const float* __restrict__ x, int Ns, int C, const int32_t* __restrict__ idx,
                                    int Nq, int H, float* __restrict__ out, int32_t* __restrict__ argmax,
                                    float* __restrict__ clear, const int32_t* __restrict__ width, d3f::RowGroups rg) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (clear)
    for (size_t i = t; i < (size_t)Ns * C; i += (size_t)gridDim.x * blockDim.x) clear[i] = 0.0f;
  if (t >= (size_t)Nq * C) return;
  const int n = (int)(t / C), c = (int)(t % C);
  if (rg.len) {  // stacked pairs: every group of clouds has the width of the table its own batch would have had
    const int Hg = width ? min(H, max(1, width[d3f::group_of_row(rg, n)])) : H;
    const int32_t* grow = idx + (size_t)n * H;
    float gbest = -INFINITY;
    int garg = Ns;
    for (int h = 0; h < Hg; ++h) {
      const int m = grow[h];
      const bool real = m >= 0 && m < Ns;
      const float v = real ? x[(size_t)m * C + c] : 0.0f;
      if (v > gbest || h == 0) { gbest = v; garg = real ? m : Ns; }
    }
    out[(size_t)n * C + c] = gbest;
    if (argmax) argmax[(size_t)n * C + c] = garg;
    return;
  }
  const int32_t* row = idx + (size_t)n * H;
  float best = -INFINITY;
  int arg = Ns;
  // the reference's table has min(limit, max_count) columns (dataloader.py:64-66): with a wider, static-shape table the
  // device-resident max_count says how many leading columns it would have kept
  const int Hw = width ? min(H, max(1, __builtin_amdgcn_readfirstlane(*width))) : H;  // wave-uniform loop bound
  for (int h = 0; h < Hw; ++h) {
    const int m = row[h];
    const bool real = m >= 0 && m < Ns;
    const float v = real ? x[(size_t)m * C + c] : 0.0f;  // shadow row is zeros (blocks.py:103)
    if (v > best || h == 0) {
      best = v;
      arg = real ? m : Ns;
    }
  }
  out[(size_t)n * C + c] = best;
  if (argmax) argmax[(size_t)n * C + c] = arg;
}

// 4 channels per thread (C % 4 == 0), 32-bit element indices
__global__ void max_pool_fwd_v4_kernel(const float* __restrict__ x, int Ns, int C4, const int32_t* __restrict__ idx,
                                       int Nq, int H, float* __restrict__ out, int32_t* __restrict__ argmax,
                                       float* __restrict__ clear, const int32_t* __restrict__ width,
                                       d3f::RowGroups rg) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (clear) {
    const uint32_t n4 = (uint32_t)Ns * (uint32_t)C4;
    for (uint32_t i = t; i < n4; i += gridDim.x * blockDim.x) ((float4*)clear)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (t >= (uint32_t)Nq * (uint32_t)C4) return;
  const uint32_t n = t / (uint32_t)C4, c4 = t - n * (uint32_t)C4;
  const int32_t* row = idx + (size_t)n * H;
  int Hw = H;
  if (width) Hw = min(H, max(1, rg.len ? width[d3f::group_of_row(rg, (int)n)] : *width));
  float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  int a0 = Ns, a1 = Ns, a2 = Ns, a3 = Ns;
  const float4* x4 = (const float4*)x;
  for (int h = 0; h < Hw; ++h) {
    const int m = row[h];
    const bool real = m >= 0 && m < Ns;
    const float4 v = real ? x4[(size_t)m * C4 + c4] : make_float4(0.f, 0.f, 0.f, 0.f);  // shadow row is zeros (blocks.py:103)
    const int mm = real ? m : Ns;
    if (v.x > best.x || h == 0) { best.x = v.x; a0 = mm; }
    if (v.y > best.y || h == 0) { best.y = v.y; a1 = mm; }
    if (v.z > best.z || h == 0) { best.z = v.z; a2 = mm; }
    if (v.w > best.w || h == 0) { best.w = v.w; a3 = mm; }
  }
  ((float4*)out)[t] = best;
  if (argmax) ((int4*)argmax)[t] = make_int4(a0, a1, a2, a3);
}

__global__ void max_pool_bwd_kernel(const float* __restrict__ go, const int32_t* __restrict__ argmax, int Nq, int C,
                                    int Ns, float* __restrict__ gx) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)Nq * C) return;
  const int m = argmax[t];
  if (m >= 0 && m < Ns) atomicAdd(&gx[(size_t)m * C + (t % C)], go[t]);
}

__global__ void max_pool_bwd32_kernel(const float* __restrict__ go, const int32_t* __restrict__ argmax, uint32_t total,
                                      uint32_t C, int Ns, float* __restrict__ gx) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int m = argmax[t];
  if (m >= 0 && m < Ns) atomicAdd(&gx[(uint32_t)m * C + t % C], go[t]);
}

// out has Cs extra columns per row filled from `skip` [Nq, Cs]: the decoder's upsample + concatenation
// (architectures.py:311-313 after blocks.py:712) written by ONE launch (Cs = 0: plain closest_pool)
__global__ void closest_pool_fwd_kernel(const float* __restrict__ x, int Ns, int C, const int32_t* __restrict__ idx,
                                        int Nq, int H, const float* __restrict__ skip, int Cs,
                                        float* __restrict__ out, float* __restrict__ clear) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (clear)
    for (size_t i = t; i < (size_t)Ns * C; i += (size_t)gridDim.x * blockDim.x) clear[i] = 0.0f;
  const int W = C + Cs;
  if (t >= (size_t)Nq * W) return;
  const int n = (int)(t / W), c = (int)(t % W);
  if (c < C) {
    const int m = idx[(size_t)n * H];
    out[t] = (m >= 0 && m < Ns) ? x[(size_t)m * C + c] : 0.0f;
  } else {
    out[t] = skip[(size_t)n * Cs + (c - C)];
  }
}

// 16 bytes per thread (C % 4 == 0 and Cs % 4 == 0), 32-bit element indices
__global__ void closest_pool_fwd_v4_kernel(const float* __restrict__ x, int Ns, int C4, const int32_t* __restrict__ idx,
                                           int Nq, int H, const float* __restrict__ skip, int Cs4,
                                           float* __restrict__ out, float* __restrict__ clear) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (clear) {
    const uint32_t n4 = (uint32_t)Ns * (uint32_t)C4;
    for (uint32_t i = t; i < n4; i += gridDim.x * blockDim.x) ((float4*)clear)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const uint32_t W4 = (uint32_t)(C4 + Cs4);
  if (t >= (uint32_t)Nq * W4) return;
  const uint32_t n = t / W4, c = t - n * W4;
  float4 v;
  if (c < (uint32_t)C4) {
    const int m = idx[(size_t)n * H];
    v = (m >= 0 && m < Ns) ? ((const float4*)x)[(size_t)m * C4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
  } else {
    v = ((const float4*)skip)[(size_t)n * Cs4 + (c - C4)];
  }
  ((float4*)out)[t] = v;
}

// go has row stride ld >= C (the gradient of a concatenation arrives as a column slice: no contiguous copy needed)
__global__ void closest_pool_bwd_kernel(const float* __restrict__ go, int ld, const int32_t* __restrict__ idx, int Nq,
                                        int H, int C, int Ns, float* __restrict__ gx) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)Nq * C) return;
  const int n = (int)(t / C), c = (int)(t % C);
  const int m = idx[(size_t)n * H];
  if (m >= 0 && m < Ns) atomicAdd(&gx[(size_t)m * C + c], go[(size_t)n * ld + c]);
}

__global__ void closest_pool_bwd32_kernel(const float* __restrict__ go, uint32_t ld, const int32_t* __restrict__ idx,
                                          uint32_t total, uint32_t H, uint32_t C, int Ns, float* __restrict__ gx) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const uint32_t n = t / C, c = t - n * C;
  const int m = idx[(size_t)n * H];
  // (a zero contributes nothing, bit for bit -- and the descriptor head's incoming gradient is zero in ~70 % of its rows:
  // the losses read the correspondences only)
  const float v = go[(size_t)n * ld + c];
  if (m >= 0 && m < Ns && v != 0.0f) atomicAdd(&gx[(uint32_t)m * C + c], v);
}

}  // namespace

extern "C" {

int d3f_max_pool_forward(const float* x, int Ns, int C, const int32_t* idx, int Nq, int H, float* out,
                         int32_t* argmax_out, float* grad_x_clear, const int32_t* width_dev, const int32_t* q_len,
                         int B, int group, void* stream) {
  if (!x || !idx || !out || Ns < 0 || C < 1 || Nq < 0 || H < 1) return D3F_EINVAL;
  if (q_len && (B < 1 || B > D3F_MAX_BATCH || group < 1)) return D3F_EINVAL;
  const d3f::RowGroups rg = {q_len, B, group};
  if (Nq == 0) {
    if (grad_x_clear && d3f::zero_async(grad_x_clear, sizeof(float) * (size_t)Ns * C, (hipStream_t)stream) != hipSuccess)
      return D3F_ELAUNCH;
    return D3F_OK;
  }
  const bool small = (long long)Nq * C < (1ll << 31) && (long long)Ns * C < (1ll << 31);
  const bool aligned = ((uintptr_t)x | (uintptr_t)out | (uintptr_t)argmax_out | (uintptr_t)grad_x_clear) % 16 == 0;
  if (C % 4 == 0 && small && aligned)
    max_pool_fwd_v4_kernel<<<d3f::cdiv((long long)Nq * (C / 4), 256), 256, 0, (hipStream_t)stream>>>(
        x, Ns, C / 4, idx, Nq, H, out, argmax_out, grad_x_clear, width_dev, rg);
  else
    max_pool_fwd_kernel<<<d3f::cdiv((long long)Nq * C, 256), 256, 0, (hipStream_t)stream>>>(
        x, Ns, C, idx, Nq, H, out, argmax_out, grad_x_clear, width_dev, rg);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

int d3f_max_pool_backward(const float* grad_out, const int32_t* argmax, int Nq, int C, int Ns, float* grad_x,
                          int grad_x_precleared, void* stream) {
  if (!grad_out || !argmax || !grad_x || Nq < 0 || C < 1 || Ns < 0) return D3F_EINVAL;
  if (!grad_x_precleared &&
      d3f::zero_async(grad_x, sizeof(float) * (size_t)Ns * C, (hipStream_t)stream) != hipSuccess)
    return D3F_ELAUNCH;
  if (Nq == 0) return D3F_OK;
  if ((long long)Nq * C < (1ll << 31) && (long long)Ns * C < (1ll << 31))
    max_pool_bwd32_kernel<<<d3f::cdiv((long long)Nq * C, 256), 256, 0, (hipStream_t)stream>>>(
        grad_out, argmax, (uint32_t)Nq * (uint32_t)C, (uint32_t)C, Ns, grad_x);
  else
    max_pool_bwd_kernel<<<d3f::cdiv((long long)Nq * C, 256), 256, 0, (hipStream_t)stream>>>(grad_out, argmax, Nq, C,
                                                                                            Ns, grad_x);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

int d3f_closest_pool_forward(const float* x, int Ns, int C, const int32_t* idx, int Nq, int H, const float* skip,
                             int Cs, float* out, float* grad_x_clear, void* stream) {
  if (!x || !idx || !out || Ns < 0 || C < 1 || Nq < 0 || H < 1 || Cs < 0 || (Cs > 0 && !skip)) return D3F_EINVAL;
  if (Nq == 0) {
    if (grad_x_clear && d3f::zero_async(grad_x_clear, sizeof(float) * (size_t)Ns * C, (hipStream_t)stream) != hipSuccess)
      return D3F_ELAUNCH;
    return D3F_OK;
  }
  const bool small = (long long)Nq * (C + Cs) < (1ll << 31) && (long long)Ns * C < (1ll << 31);
  const bool aligned = ((uintptr_t)x | (uintptr_t)out | (uintptr_t)skip | (uintptr_t)grad_x_clear) % 16 == 0;
  if (C % 4 == 0 && Cs % 4 == 0 && small && aligned)
    closest_pool_fwd_v4_kernel<<<d3f::cdiv((long long)Nq * ((C + Cs) / 4), 256), 256, 0, (hipStream_t)stream>>>(
        x, Ns, C / 4, idx, Nq, H, skip, Cs / 4, out, grad_x_clear);
  else
    closest_pool_fwd_kernel<<<d3f::cdiv((long long)Nq * (C + Cs), 256), 256, 0, (hipStream_t)stream>>>(
        x, Ns, C, idx, Nq, H, skip, Cs, out, grad_x_clear);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

int d3f_closest_pool_backward(const float* grad_out, int ld, const int32_t* idx, int Nq, int H, int C, int Ns,
                              float* grad_x, int grad_x_precleared, void* stream) {
  if (!grad_out || !idx || !grad_x || Nq < 0 || C < 1 || Ns < 0 || H < 1 || ld < C) return D3F_EINVAL;
  if (!grad_x_precleared &&
      d3f::zero_async(grad_x, sizeof(float) * (size_t)Ns * C, (hipStream_t)stream) != hipSuccess)
    return D3F_ELAUNCH;
  if (Nq == 0) return D3F_OK;
  if ((long long)Nq * ld < (1ll << 31) && (long long)Ns * C < (1ll << 31))
    closest_pool_bwd32_kernel<<<d3f::cdiv((long long)Nq * C, 256), 256, 0, (hipStream_t)stream>>>(
        grad_out, (uint32_t)ld, idx, (uint32_t)Nq * (uint32_t)C, (uint32_t)H, (uint32_t)C, Ns, grad_x);
  else
    closest_pool_bwd_kernel<<<d3f::cdiv((long long)Nq * C, 256), 256, 0, (hipStream_t)stream>>>(grad_out, ld, idx, Nq,
                                                                                                H, C, Ns, grad_x);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

}  // extern "C"
