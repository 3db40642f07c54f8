// Batch normalisation over the stacked points of a feature matrix x [N, C], forward and backward.
//
// Replaces the use_bn=True branch of the reference's BatchNormBlock (models/blocks.py:454-455,465-471): the features
// are viewed as one sample of C channels and N positions and pushed through nn.BatchNorm1d(C, momentum), i.e.
//   training: mean[c] = 1/N sum_n x[n,c],  var[c] = 1/N sum_n (x[n,c]-mean[c])^2         (biased, used to normalise)
//             y[n,c]  = (x[n,c]-mean[c]) / sqrt(var[c]+eps) * gamma[c] + beta[c]
//             running_mean = (1-m) running_mean + m mean;  running_var = (1-m) running_var + m var N/(N-1)
//   eval    : y[n,c]  = (x[n,c]-running_mean[c]) / sqrt(running_var[c]+eps) * gamma[c] + beta[c]
// with an optional LeakyReLU fused behind it (the UnaryBlock's, blocks.py:497,510-512).
//
// HBM-bound: x is read twice for the statistics (two-pass variance: sum-of-squares cancels badly for activations with
// a large mean) and once more to normalise; 12 + 4 bytes per element forward, 16 + 4 backward.  All reductions are
// fixed-order (per-chunk partials in a workspace, combined in chunk order), so results do not depend on scheduling.
//   launch 1  partial column sums per row chunk                          -> part[chunk, C]
//   launch 2  mean from the partials; partial sums of squared deviations -> part2[chunk, C]
//   launch 3  var from the partials; normalise (+ LeakyReLU); block 0 stores mean / invstd and updates the running stats
// backward (training statistics):
//   launch 1  partial sums of g and g*xhat per chunk (g = gradient behind the activation)
//   launch 2  dgamma, dbeta from the partials; dx = gamma*invstd * (g - dbeta/N - xhat*dgamma/N)
// N may be a capacity: with n_live != NULL only the first min(N, *n_live) rows are live (rows past that are written as
// zeros forward / zero gradient backward), so the op can sit inside a captured graph over capacity-shaped buffers.
#include "common.hpp"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxChunks = 256;

struct Geometry {
  int cols;        // columns handled by one workgroup (<= 256, power-of-two divisor layout below)
  int lanes;       // row lanes per workgroup = kThreads / cols
  int col_tiles;   // gridDim.y
  int chunks;      // gridDim.x
  int chunk_rows;
};

Geometry geometry(int N, int C) {
  Geometry g;
  g.cols = 1;
  while (g.cols < C && g.cols < kThreads) g.cols <<= 1;
  g.lanes = kThreads / g.cols;
  g.col_tiles = d3f::cdiv(C, g.cols);
  int rows = d3f::cdiv(N, kMaxChunks);
  rows = d3f::cdiv(rows, g.lanes) * g.lanes;
  if (rows < 4 * g.lanes) rows = 4 * g.lanes;
  g.chunk_rows = rows;
  g.chunks = d3f::cdiv(N > 0 ? N : 1, rows);
  return g;
}

__device__ __forceinline__ int live_rows(int N, const int32_t* n_live) {
  return n_live ? max(0, min(N, *n_live)) : N;
}

// sum over this chunk's rows of f(row, col) (a pair of values) for the thread's column; the row lanes are combined
// through LDS in lane order
template <typename F>
__device__ __forceinline__ float2 chunk_column_sum(int n, int C, int cols, int lanes, int chunk_rows, float2* sh, F f,
                                                   int& col_out) {
  const int cl = threadIdx.x % cols, lane = threadIdx.x / cols;
  const int col = blockIdx.y * cols + cl;
  const int r0 = blockIdx.x * chunk_rows, r1 = min(n, r0 + chunk_rows);
  float2 acc = make_float2(0.0f, 0.0f);
  if (col < C)
    for (int r = r0 + lane; r < r1; r += lanes) {
      const float2 v = f(r, col);
      acc.x += v.x;
      acc.y += v.y;
    }
  sh[lane * cols + cl] = acc;
  __syncthreads();
  float2 tot = make_float2(0.0f, 0.0f);
  if (lane == 0)
    for (int l = 0; l < lanes; ++l) {
      tot.x += sh[l * cols + cl].x;
      tot.y += sh[l * cols + cl].y;
    }
  col_out = (lane == 0 && col < C) ? col : -1;
  return tot;
}

__device__ __forceinline__ float combine(const float* __restrict__ part, int chunks, int C, int col) {
  float s = 0.0f;
  for (int k = 0; k < chunks; ++k) s += part[(size_t)k * C + col];
  return s;
}

__global__ __launch_bounds__(kThreads) void bn_sum_kernel(const float* __restrict__ x, int N, int C,
                                                          const int32_t* __restrict__ n_live, Geometry g,
                                                          float* __restrict__ part) {
  __shared__ float2 sh[kThreads];
  const int n = live_rows(N, n_live);
  int col;
  const float2 s = chunk_column_sum(n, C, g.cols, g.lanes, g.chunk_rows, sh,
                                    [&](int r, int c) { return make_float2(x[(size_t)r * C + c], 0.0f); }, col);
  if (col >= 0) part[(size_t)blockIdx.x * C + col] = s.x;
}

__global__ __launch_bounds__(kThreads) void bn_dev_kernel(const float* __restrict__ x, int N, int C,
                                                          const int32_t* __restrict__ n_live, Geometry g,
                                                          const float* __restrict__ part, float* __restrict__ part2) {
  __shared__ float2 sh[kThreads];
  __shared__ float mean_sh[kThreads];
  const int n = live_rows(N, n_live);
  const int cl = threadIdx.x % g.cols, c0 = blockIdx.y * g.cols + cl;
  if (threadIdx.x < g.cols) mean_sh[cl] = c0 < C ? combine(part, g.chunks, C, c0) / (float)max(n, 1) : 0.0f;
  __syncthreads();
  const float m = mean_sh[cl];
  int col;
  const float2 s = chunk_column_sum(n, C, g.cols, g.lanes, g.chunk_rows, sh, [&](int r, int c) {
    const float d = x[(size_t)r * C + c] - m;
    return make_float2(d * d, 0.0f);
  }, col);
  if (col >= 0) part2[(size_t)blockIdx.x * C + col] = s.x;
}

__global__ __launch_bounds__(kThreads) void bn_apply_kernel(
    const float* __restrict__ x, int N, int C, const int32_t* __restrict__ n_live, Geometry g,
    const float* __restrict__ part, const float* __restrict__ part2, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ running_mean, float* __restrict__ running_var, float momentum,
    float eps, int training, float slope, float* __restrict__ y, float* __restrict__ save_mean,
    float* __restrict__ save_invstd) {
  __shared__ float mean_sh[kThreads], scale_sh[kThreads], shift_sh[kThreads];
  const int n = live_rows(N, n_live);
  const int cl = threadIdx.x % g.cols, lane = threadIdx.x / g.cols, c0 = blockIdx.y * g.cols + cl;
  if (threadIdx.x < g.cols && c0 < C) {
    float m, v;
    if (training) {
      m = combine(part, g.chunks, C, c0) / (float)max(n, 1);
      v = combine(part2, g.chunks, C, c0) / (float)max(n, 1);
    } else {
      m = running_mean[c0];
      v = running_var[c0];
    }
    const float inv = 1.0f / sqrtf(v + eps);
    const float ga = gamma ? gamma[c0] : 1.0f, be = beta ? beta[c0] : 0.0f;
    mean_sh[cl] = m;
    scale_sh[cl] = inv * ga;
    shift_sh[cl] = be;
    if (blockIdx.x == 0) {
      if (save_mean) save_mean[c0] = m;
      if (save_invstd) save_invstd[c0] = inv;
      if (training && running_mean && running_var && n > 0) {   // unbiased variance for the running estimate
        const float unb = n > 1 ? v * ((float)n / (float)(n - 1)) : v;
        running_mean[c0] = (1.0f - momentum) * running_mean[c0] + momentum * m;
        running_var[c0] = (1.0f - momentum) * running_var[c0] + momentum * unb;
      }
    }
  }
  __syncthreads();
  if (c0 >= C) return;
  const float m = mean_sh[cl], sc = scale_sh[cl], sf = shift_sh[cl];
  const int r0 = blockIdx.x * g.chunk_rows, r1 = min(N, r0 + g.chunk_rows);
  for (int r = r0 + lane; r < r1; r += g.lanes) {
    float v = 0.0f;
    if (r < n) {
      v = (x[(size_t)r * C + c0] - m) * sc + sf;
      v = v > 0.0f ? v : v * slope;
    }
    y[(size_t)r * C + c0] = v;
  }
}

// g = go * act'(bn(x)); partial sums of g and of g * xhat
__global__ __launch_bounds__(kThreads) void bn_bwd_sum_kernel(
    const float* __restrict__ x, int N, int C, const int32_t* __restrict__ n_live, Geometry g,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ save_mean,
    const float* __restrict__ save_invstd, float slope, const float* __restrict__ go, float* __restrict__ part,
    float* __restrict__ part2) {
  __shared__ float2 sh[kThreads];
  const int n = live_rows(N, n_live);
  const int cl = threadIdx.x % g.cols, c0 = blockIdx.y * g.cols + cl;
  const bool ok = c0 < C;
  const float m = ok ? save_mean[c0] : 0.0f, inv = ok ? save_invstd[c0] : 0.0f;
  const float ga = ok && gamma ? gamma[c0] : 1.0f, be = ok && beta ? beta[c0] : 0.0f;
  int col;
  const float2 s = chunk_column_sum(n, C, g.cols, g.lanes, g.chunk_rows, sh, [&](int r, int c) {
    const float xh = (x[(size_t)r * C + c] - m) * inv;
    const float gg = go[(size_t)r * C + c] * ((xh * ga + be) > 0.0f ? 1.0f : slope);
    return make_float2(gg, gg * xh);
  }, col);
  if (col >= 0) {
    part[(size_t)blockIdx.x * C + col] = s.x;
    part2[(size_t)blockIdx.x * C + col] = s.y;
  }
}

__global__ __launch_bounds__(kThreads) void bn_bwd_apply_kernel(
    const float* __restrict__ x, int N, int C, const int32_t* __restrict__ n_live, Geometry g,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ save_mean,
    const float* __restrict__ save_invstd, float slope, int training, const float* __restrict__ go,
    const float* __restrict__ part, const float* __restrict__ part2, float* __restrict__ grad_x,
    float* __restrict__ grad_gamma, float* __restrict__ grad_beta) {
  __shared__ float db_sh[kThreads], dg_sh[kThreads];
  const int n = live_rows(N, n_live);
  const int cl = threadIdx.x % g.cols, lane = threadIdx.x / g.cols, c0 = blockIdx.y * g.cols + cl;
  if (threadIdx.x < g.cols && c0 < C) {
    const float db = combine(part, g.chunks, C, c0), dg = combine(part2, g.chunks, C, c0);
    db_sh[cl] = db;
    dg_sh[cl] = dg;
    if (blockIdx.x == 0) {
      if (grad_beta) grad_beta[c0] = db;
      if (grad_gamma) grad_gamma[c0] = dg;
    }
  }
  __syncthreads();
  if (c0 >= C || !grad_x) return;
  const float m = save_mean[c0], inv = save_invstd[c0];
  const float ga = gamma ? gamma[c0] : 1.0f, be = beta ? beta[c0] : 0.0f;
  const float inv_n = 1.0f / (float)max(n, 1);
  const float mb = training ? db_sh[cl] * inv_n : 0.0f, mg = training ? dg_sh[cl] * inv_n : 0.0f;
  const int r0 = blockIdx.x * g.chunk_rows, r1 = min(N, r0 + g.chunk_rows);
  for (int r = r0 + lane; r < r1; r += g.lanes) {
    float v = 0.0f;
    if (r < n) {
      const float xh = (x[(size_t)r * C + c0] - m) * inv;
      const float gg = go[(size_t)r * C + c0] * ((xh * ga + be) > 0.0f ? 1.0f : slope);
      v = ga * inv * (gg - mb - xh * mg);
    }
    grad_x[(size_t)r * C + c0] = v;
  }
}

}  // namespace

extern "C" {

size_t d3f_batchnorm_ws_bytes(int N, int C) {
  if (N < 0 || C < 1) return 0;
  const Geometry g = geometry(N, C);
  return 2 * d3f::align_up(sizeof(float) * (size_t)g.chunks * C, 256);
}

int d3f_batchnorm_forward(const float* x, int N, int C, const int32_t* n_live, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, float momentum, float eps, int training,
                          float slope, float* y, float* save_mean, float* save_invstd, void* ws, size_t ws_bytes,
                          void* stream) {
  if (!x || !y || N < 0 || C < 1 || !(eps > 0.0f)) return D3F_EINVAL;
  if (!training && (!running_mean || !running_var)) return D3F_EINVAL;
  if (N == 0) return D3F_OK;
  if (!ws || ws_bytes < d3f_batchnorm_ws_bytes(N, C)) return D3F_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const Geometry g = geometry(N, C);
  d3f::Carver carve(ws);
  float* part = carve.take<float>((size_t)g.chunks * C);
  float* part2 = carve.take<float>((size_t)g.chunks * C);
  const dim3 grid(g.chunks, g.col_tiles);
  if (training) {
    bn_sum_kernel<<<grid, kThreads, 0, st>>>(x, N, C, n_live, g, part);
    bn_dev_kernel<<<grid, kThreads, 0, st>>>(x, N, C, n_live, g, part, part2);
  }
  bn_apply_kernel<<<grid, kThreads, 0, st>>>(x, N, C, n_live, g, part, part2, gamma, beta, running_mean, running_var,
                                             momentum, eps, training ? 1 : 0, slope, y, save_mean, save_invstd);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

int d3f_batchnorm_backward(const float* x, int N, int C, const int32_t* n_live, const float* gamma, const float* beta,
                           const float* save_mean, const float* save_invstd, float slope, int training,
                           const float* grad_y, float* grad_x, float* grad_gamma, float* grad_beta, void* ws,
                           size_t ws_bytes, void* stream) {
  if (!x || !grad_y || !save_mean || !save_invstd || N < 0 || C < 1) return D3F_EINVAL;
  if (N == 0) return D3F_OK;
  if (!ws || ws_bytes < d3f_batchnorm_ws_bytes(N, C)) return D3F_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const Geometry g = geometry(N, C);
  d3f::Carver carve(ws);
  float* part = carve.take<float>((size_t)g.chunks * C);
  float* part2 = carve.take<float>((size_t)g.chunks * C);
  const dim3 grid(g.chunks, g.col_tiles);
  bn_bwd_sum_kernel<<<grid, kThreads, 0, st>>>(x, N, C, n_live, g, gamma, beta, save_mean, save_invstd, slope, grad_y,
                                               part, part2);
  bn_bwd_apply_kernel<<<grid, kThreads, 0, st>>>(x, N, C, n_live, g, gamma, beta, save_mean, save_invstd, slope,
                                                 training ? 1 : 0, grad_y, part, part2, grad_x, grad_gamma, grad_beta);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

}  // extern "C"
