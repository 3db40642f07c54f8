// Library identity + device check.
#include <string.h>

#include "common.hpp"

namespace {

// One launch moves the inputs of a training step into its static buffers: up to 24 jobs, job j by the workgroups with
// blockIdx.y == j.  kind 0: dst words = src words (device-to-device copy of a 4-byte multiple); kind 1: dst bytes =
// (src doubles > threshold) -- the "negative" mask of the circle loss, dist_keypts > safe_radius (reference
// utils/loss.py:119), computed while the keypoint distances are being copied anyway.
struct CopyJobs {
  const void* src[24];
  void* dst[24];
  size_t n[24];       // words (kind 0) / elements (kind 1)
  int kind[24];
  double thr;
};
__global__ __launch_bounds__(256) void copy_many_kernel(CopyJobs jobs) {
  const int j = blockIdx.y;
  const size_t stride = (size_t)gridDim.x * blockDim.x, t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n = jobs.n[j];
  if (jobs.kind[j] == 0) {
    const uint32_t* s = (const uint32_t*)jobs.src[j];
    uint32_t* d = (uint32_t*)jobs.dst[j];
    for (size_t i = t0; i < n; i += stride) d[i] = s[i];
  } else {
    const double* s = (const double*)jobs.src[j];
    uint8_t* d = (uint8_t*)jobs.dst[j];
    for (size_t i = t0; i < n; i += stride) d[i] = s[i] > jobs.thr ? 1 : 0;
  }
}

// One launch for the permuted copies W'[k][o][c] = W[k][c][o] of several KPConv weight tensors [K, Cin, Cout] (what the
// transposed-aggregation grad-input contracts with: reference autograd of torch.matmul(weighted_features, self.weights),
// models/blocks.py:369-374, seen from the supports).  Job j by the workgroups with blockIdx.y == j; a workgroup moves
// 32 x 32 tiles through LDS (both sides coalesced), tile t of a job = (k, c-tile, o-tile).
struct PermuteJobs {
  const float* src[16];
  float* dst[16];
  int K[16], Cin[16], Cout[16];
};
__global__ __launch_bounds__(256) void permute_weights_kernel(PermuteJobs jobs) {
  __shared__ float tile[32][33];
  const int j = blockIdx.y;
  const int K = jobs.K[j], Cin = jobs.Cin[j], Cout = jobs.Cout[j];
  const int tc = Cin / 32, to = Cout / 32, ntiles = K * tc * to;
  const float* src = jobs.src[j];
  float* dst = jobs.dst[j];
  const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;     // 32 x 8 threads
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int k = t / (tc * to), r = t - k * tc * to, ct = r / to, ot = r - ct * to;
    const float* s = src + ((size_t)k * Cin + ct * 32) * Cout + ot * 32;
    float* d = dst + ((size_t)k * Cout + ot * 32) * Cin + ct * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) tile[ly + 8 * i][lx] = s[(size_t)(ly + 8 * i) * Cout + lx];     // tile[c][o]
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) d[(size_t)(ly + 8 * i) * Cin + lx] = tile[lx][ly + 8 * i];        // dst[o][c]
    __syncthreads();
  }
}

}  // namespace

namespace d3f {
static d3f_tunables g_tunables = {};   // all zero = every built-in choice
const d3f_tunables& tunables() { return g_tunables; }
}  // namespace d3f

extern "C" {

void d3f_get_tunables(d3f_tunables* out) {
  if (out) *out = d3f::g_tunables;
}
int d3f_set_tunables(const d3f_tunables* in) {
  if (!in || in->atb_task_us < 0 || in->atb_form < 0 || in->atb_form > 2 || in->atb_first_form_wgs < 0 ||
      in->match_wgs < 0 || in->atb_pipe < 0 || in->atb_pipe > 16 || in->xw_rows < 0 || in->xw_rows > 4 || in->xw_split < 0 ||
      in->xw_split > 64 || in->rowgemm_wide < 0 || in->rowgemm_wide > 2 ||
      in->rowgemm_rt < 0 || in->rowgemm_rt > 1)
    return D3F_EINVAL;
  d3f::g_tunables = *in;
  return D3F_OK;
}

/* Replaces the ~25 separate copy launches with which a stacked training step's inputs (2Q clouds, their features, Q
 * correspondence tables and keypoint-distance matrices: the dataset item of reference datasets/ThreeDMatch.py:135-149,
 * Q times) reach the step's static buffers, and the two launches of the loss's neighbor mask.
 * kinds[j] = 0: copy bytes[j] (a multiple of 4) from srcs[j] to dsts[j]; 1: dsts[j][i] (uint8) = srcs[j][i] (float64) >
 * threshold for i < bytes[j] / 8.  n <= 24. */
int d3f_copy_buffers(const void* const* srcs, void* const* dsts, const size_t* bytes, const int* kinds, int n,
                     double threshold, void* stream) {
  if (n < 0 || n > 24 || (n > 0 && (!srcs || !dsts || !bytes || !kinds))) return D3F_EINVAL;
  CopyJobs jobs;
  jobs.thr = threshold;
  int m = 0;
  size_t most = 0;
  for (int j = 0; j < n; ++j) {
    if (bytes[j] == 0) continue;
    if (!srcs[j] || !dsts[j] || (kinds[j] != 0 && kinds[j] != 1)) return D3F_EINVAL;
    if (kinds[j] == 0 && (((uintptr_t)srcs[j] | (uintptr_t)dsts[j] | bytes[j]) & 3)) return D3F_EINVAL;
    if (kinds[j] == 1 && (((uintptr_t)srcs[j] | bytes[j]) & 7)) return D3F_EINVAL;
    jobs.src[m] = srcs[j];
    jobs.dst[m] = dsts[j];
    jobs.n[m] = kinds[j] == 0 ? bytes[j] / 4 : bytes[j] / 8;
    jobs.kind[m] = kinds[j];
    if (jobs.n[m] > most) most = jobs.n[m];
    ++m;
  }
  if (m == 0) return D3F_OK;
  size_t blocks = (most + 1023) / 1024;     // ~4 elements per thread of the largest job
  if (blocks > 256) blocks = 256;
  if (blocks < 1) blocks = 1;
  copy_many_kernel<<<dim3((unsigned)blocks, (unsigned)m), 256, 0, (hipStream_t)stream>>>(jobs);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

int d3f_permute_kpconv_weights(const float* const* srcs, float* const* dsts, const int* K, const int* Cin,
                               const int* Cout, int n, void* stream) {
  if (n < 0 || n > 16 || (n > 0 && (!srcs || !dsts || !K || !Cin || !Cout))) return D3F_EINVAL;
  if (n == 0) return D3F_OK;
  PermuteJobs jobs;
  long long most = 0;
  for (int j = 0; j < n; ++j) {
    if (!srcs[j] || !dsts[j] || K[j] < 1 || Cin[j] < 32 || Cout[j] < 32 || Cin[j] % 32 || Cout[j] % 32) return D3F_EINVAL;
    jobs.src[j] = srcs[j];
    jobs.dst[j] = dsts[j];
    jobs.K[j] = K[j];
    jobs.Cin[j] = Cin[j];
    jobs.Cout[j] = Cout[j];
    const long long tiles = (long long)K[j] * (Cin[j] / 32) * (Cout[j] / 32);
    if (tiles > most) most = tiles;
  }
  for (int j = n; j < 16; ++j) {
    jobs.src[j] = nullptr;
    jobs.dst[j] = nullptr;
    jobs.K[j] = jobs.Cin[j] = jobs.Cout[j] = 0;
  }
  const unsigned blocks = (unsigned)(most > 512 ? 512 : most);
  permute_weights_kernel<<<dim3(blocks, (unsigned)n), 256, 0, (hipStream_t)stream>>>(jobs);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

const char* d3f_version(void) { return "d3feat-hip 0.1 (gfx950)"; }

// returns 1 for gfx950, 0 for another architecture, negative HIP error code (negated) if the query itself failed
int d3f_device_arch_ok(void) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return -(int)e;
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) return -(int)e;
  return strstr(prop.gcnArchName, "gfx950") != nullptr ? 1 : 0;
}

int d3f_device_arch_name(char* out, int n) {
  int dev = 0;
  if (!out || n < 1) return D3F_EINVAL;
  out[0] = 0;
  if (hipGetDevice(&dev) != hipSuccess) return D3F_ELAUNCH;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return D3F_ELAUNCH;
  strncpy(out, prop.gcnArchName, (size_t)n - 1);
  out[n - 1] = 0;
  return D3F_OK;
}

}  // extern "C"
