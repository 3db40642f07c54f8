// Guarded SGD step on the flat parameter / gradient / momentum buffers.
//
// Replaces the reference's optimizer step and its host-side NaN/Inf guard (trainer.py:104-111: loop over parameters,
// `torch.isfinite(p.grad).all()` -> one device->host sync each, then torch.optim.SGD(momentum 0.98, weight decay
// 1e-6), training_3DMatch.py:62-76).  Two launches over the flat buffers and no host sync:
//   1. nonfinite_kernel : state[0] |= any(!isfinite(g))  (or the pair's device status word is set: a pyramid that
//                         overflowed a capacity must not reach the parameters -- state[2] |= flags, ++state[3])
//   2. sgd_kernel       : if (!state[0]) { buf = momentum*buf + (g + wd*p); p -= lr*buf; } else ++state[1]
// The operation order is torch.optim.SGD's (d = g + wd*p; buf = buf*momentum + d; p = p + (-lr)*buf), unfused.
// HBM-bound: 4 B/param read in (1), 12 B read + 8 B written in (2).
// Several pairs in flight on one GPU (train.PairLanes) leave one gradient buffer each: both kernels take up to four of
// them and the update uses their SUM (4 B/param more per extra lane and kernel, no pass of its own for the addition).
#include "common.hpp"

namespace {

constexpr int kMaxLanes = 4;
struct Lanes {
  const float* g[kMaxLanes];
  int n;
};

__device__ __forceinline__ bool nonfinite(float v) { return (__float_as_uint(v) & 0x7f800000u) == 0x7f800000u; }

__global__ __launch_bounds__(256) void nonfinite_kernel(Lanes lanes, size_t n, int* __restrict__ state,
                                                        const int32_t* __restrict__ pair_status) {
  if (pair_status && blockIdx.x == 0 && threadIdx.x == 0) {
    const int f = *pair_status;
    if (f) {
      atomicOr(state, 1);
      atomicOr(state + 2, f);
      atomicAdd(state + 3, 1);
    }
  }
  const size_t n4 = n / 4;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  // exponent bits all ones <=> Inf/NaN: OR the words of four independent 16-byte loads per round, test once
  unsigned acc = 0;
  bool bad = false;
  auto fold = [](uint4 v) {
    const unsigned e = 0x7f800000u;
    return (unsigned)(((v.x & e) == e) | ((v.y & e) == e) | ((v.z & e) == e) | ((v.w & e) == e));
  };
  for (int l = 0; l < lanes.n; ++l) {  // every lane on its own: Inf + (-Inf) must not hide in the sum
    const float* g = lanes.g[l];
    const uint4* g4 = (const uint4*)g;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
      const uint4 a = g4[i], b = g4[i + stride], c = g4[i + 2 * stride], d = g4[i + 3 * stride];
      acc |= fold(a) | fold(b) | fold(c) | fold(d);
    }
    for (; i < n4; i += stride) acc |= fold(g4[i]);
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) bad |= nonfinite(g[n4 * 4 + threadIdx.x]);
  }
  bad |= acc != 0;
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(state, 1);
}

__device__ __forceinline__ void sgd1(float g, float& p, float& b, float lr, float mom, float wd, float gs) {
  g = g * gs;
  const float d = g + wd * p;
  b = b * mom + d;
  p = p + (-lr) * b;
}

__global__ __launch_bounds__(256) void sgd_kernel(Lanes lanes, float* __restrict__ p,
                                                  float* __restrict__ buf, size_t n, float lr, float mom, float wd,
                                                  const float* __restrict__ hyper, int* __restrict__ state) {
  const bool skip = __builtin_nontemporal_load(state) != 0;
  if (skip) {
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(state + 1, 1);
    return;
  }
  float gs = 1.0f;
  if (hyper) {  // device-resident {lr, momentum, weight_decay, grad_scale}: changeable under a captured graph
    lr = hyper[0];
    mom = hyper[1];
    wd = hyper[2];
    gs = hyper[3];  // 1/world_size after a SUM all-reduce: the mean is taken here instead of by a pass of its own
  }
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 gv = ((const float4*)lanes.g[0])[i];
    for (int l = 1; l < lanes.n; ++l) {  // fixed order: lane 0 + lane 1 + ...
      const float4 o = ((const float4*)lanes.g[l])[i];
      gv.x += o.x;
      gv.y += o.y;
      gv.z += o.z;
      gv.w += o.w;
    }
    float4 pv = ((float4*)p)[i], bv = ((float4*)buf)[i];
    sgd1(gv.x, pv.x, bv.x, lr, mom, wd, gs);
    sgd1(gv.y, pv.y, bv.y, lr, mom, wd, gs);
    sgd1(gv.z, pv.z, bv.z, lr, mom, wd, gs);
    sgd1(gv.w, pv.w, bv.w, lr, mom, wd, gs);
    ((float4*)p)[i] = pv;
    ((float4*)buf)[i] = bv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const size_t i = n4 * 4 + threadIdx.x;
    float gv = lanes.g[0][i];
    for (int l = 1; l < lanes.n; ++l) gv += lanes.g[l][i];
    sgd1(gv, p[i], buf[i], lr, mom, wd, gs);
  }
}

__global__ void poison_kernel(float* __restrict__ g, const int32_t* __restrict__ pair_status, int* __restrict__ state) {
  if (threadIdx.x != 0) return;
  const int f = *pair_status;
  if (f) {
    g[0] = __uint_as_float(0x7fc00000u);
    atomicOr(state + 2, f);
    atomicAdd(state + 3, 1);
  }
}

}  // namespace

extern "C" {

/* The _lanes form takes n_grads (1..4) gradient buffers (a HOST array of device pointers) and steps on their sum -- one
 * buffer per pair in flight on this GPU; a non-finite value in ANY of them skips the update.
 * grad, params, momentum_buf: [n] fp32, 16-byte aligned.  state: int32[4] on the device = {scratch flag, number of
 * skipped steps so far, OR of the pair-status flags that caused a skip, number of such skips}; state[0] is reset here,
 * the others only ever grow.  pair_status (optional, device int32[1]): the device status word of the pair this
 * gradient came from; non-zero skips the update like a non-finite gradient does.  hyper_device: NULL, or float[4] on
 * the device = {lr, momentum, weight_decay, grad_scale} read at execution time instead of the scalar arguments
 * (grad_scale multiplies the gradient first: 1/world_size turns an all-reduced SUM into the mean). */
int d3f_sgd_guarded_step_lanes(const float* const* grads, int n_grads, float* params, float* momentum_buf, size_t n,
                               float lr, float momentum, float weight_decay, const float* hyper_device, int32_t* state,
                               const int32_t* pair_status, void* stream) {
  if (!grads || n_grads < 1 || n_grads > kMaxLanes || !params || !momentum_buf || !state) return D3F_EINVAL;
  Lanes lanes = {};
  lanes.n = n_grads;
  uintptr_t bits = (uintptr_t)params | (uintptr_t)momentum_buf;
  for (int l = 0; l < n_grads; ++l) {
    if (!grads[l]) return D3F_EINVAL;
    lanes.g[l] = grads[l];
    bits |= (uintptr_t)grads[l];
  }
  if ((bits & 15) != 0) return D3F_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (d3f::zero_async(state, sizeof(int32_t), st) != hipSuccess) return D3F_ELAUNCH;
  if (n == 0) return D3F_OK;
  const int blocks = (int)std::min<size_t>(2048, (size_t)d3f::cdiv((long long)(n / 4 + 1), 256));
  nonfinite_kernel<<<blocks, 256, 0, st>>>(lanes, n, state, pair_status);
  D3F_LAUNCH_CHECK();
  sgd_kernel<<<blocks, 256, 0, st>>>(lanes, params, momentum_buf, n, lr, momentum, weight_decay, hyper_device, state);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

int d3f_sgd_guarded_step(const float* grad, float* params, float* momentum_buf, size_t n, float lr, float momentum,
                         float weight_decay, const float* hyper_device, int32_t* state, const int32_t* pair_status,
                         void* stream) {
  return d3f_sgd_guarded_step_lanes(&grad, 1, params, momentum_buf, n, lr, momentum, weight_decay, hyper_device, state,
                                    pair_status, stream);
}

/* Data-parallel form of the same gate: BEFORE the gradient exchange, a rank whose pair raised a status flag turns
 * grad[0] into NaN, so that the guard evaluated on the REDUCED gradient skips the step on every rank alike.
 * state as above (state[2] |= flags, ++state[3] on this rank). */
int d3f_poison_gradient_if_status(float* grad, const int32_t* pair_status, int32_t* state, void* stream) {
  if (!grad || !pair_status || !state) return D3F_EINVAL;
  poison_kernel<<<1, 64, 0, (hipStream_t)stream>>>(grad, pair_status, state);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

}  // extern "C"
