// Shared device helpers for libd3feat_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "d3feat_hip.h"

#define D3F_WAVE 64
#define D3F_MAX_BATCH 64  // clouds per stacked batch the kernels index by linear scan

#define D3F_LAUNCH_CHECK()                          \
  do {                                              \
    if (hipGetLastError() != hipSuccess) return D3F_ELAUNCH; \
  } while (0)

namespace d3f {

// Zero-fill as an ordinary kernel launch.  hipMemsetAsync nodes were observed NOT to re-execute under hipGraph replay
// of this library's launches (second replay saw dirty allocators), so every re-initialisation is a kernel.
__global__ static void zero_fill_kernel(uint32_t* __restrict__ p, size_t n_words) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
static inline hipError_t zero_async(void* p, size_t bytes, hipStream_t stream) {
  const size_t words = (bytes + 3) / 4;  // all buffers of this library are 4-byte multiples
  if (words == 0) return hipSuccess;
  size_t blocks = (words + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  zero_fill_kernel<<<(unsigned)blocks, 256, 0, stream>>>((uint32_t*)p, words);
  return hipGetLastError();
}

// the library's tunables (misc.hip; d3f_set_tunables): plain loads of a process-wide struct, no environment reads
const d3f_tunables& tunables();

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Carves aligned sub-buffers out of one caller-provided workspace.
struct Carver {
  char* base;
  size_t off;
  explicit Carver(void* p) : base((char*)p), off(0) {}
  template <typename T>
  T* take(size_t n) {
    off = align_up(off, 256);
    T* r = (T*)(base + off);
    off += n * sizeof(T);
    return r;
  }
};

// (batch element, first row of that element) of stacked row `i`, given per-element lengths.
__device__ __forceinline__ void locate_batch(const int32_t* __restrict__ len, int B, int i, int& b, int& start) {
  int s = 0, k = 0;
  for (; k < B - 1; ++k) {
    const int l = len[k];
    if (i < s + l) break;
    s += l;
  }
  b = k;
  start = s;
}

__device__ __forceinline__ int batch_offset(const int32_t* __restrict__ len, int b) {
  int s = 0;
  for (int k = 0; k < b; ++k) s += len[k];
  return s;
}

// Rows of a stacked batch partitioned into groups of `group` consecutive clouds (e.g. the 2 fragments of a pair when 8
// pairs are stacked for inference): per-group scalars (feature maximum, table width) are indexed by group_of_row.
// len == nullptr: one group.
struct RowGroups {
  const int32_t* len;
  int B;
  int group;
};
__device__ __forceinline__ int group_of_row(const RowGroups& rg, int row) {
  if (!rg.len || rg.group <= 0) return 0;
  int b, st;
  locate_batch(rg.len, rg.B, row, b, st);
  return b / rg.group;
}

// squared distance with the reference's float32 evaluation order and NO fused multiply-add:
// d2 = dx*dx; d2 += dy*dy; d2 += dz*dz   (nanoflann.hpp:433-441).  NOTE: on AMD the __f*_rn intrinsics are plain
// operators, so this is only exact because the library is compiled with -ffp-contract=off (see _native.build).
__device__ __forceinline__ float sqdist_exact(float ax, float ay, float az, float bx, float by, float bz) {
  const float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
  float d2 = __fmul_rn(dx, dx);
  d2 = __fadd_rn(d2, __fmul_rn(dy, dy));
  d2 = __fadd_rn(d2, __fmul_rn(dz, dz));
  return d2;
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & (D3F_WAVE - 1); }

// Measurement aid (d3f_debug_set_phase_clock): per-phase shader cycles of a kernel's waves.  `out` (device uint64, null
// in normal operation): out[0] = number of waves recorded (atomic ticket, taken once per wave at its end), out[1] =
// record capacity; record r (8 words) starts at out[8 + 8 r] and holds the cycles the wave spent before each lap(i).
// The laps live in scalar registers until the wave's last instruction: no memory operation of the clock sits inside
// the measured phases (an atomic per lap would queue behind the kernel's own loads and be waited for by its vmcnt).
struct PhaseClock {
  unsigned long long* out;
  unsigned long long t, acc[6];
  __device__ __forceinline__ void start(unsigned long long* o) {
    out = o;
    if (out) {
#pragma unroll
      for (int i = 0; i < 6; ++i) acc[i] = 0;
      t = __builtin_readcyclecounter();
    }
  }
  __device__ __forceinline__ void lap(int i) {
    if (out) {
      const unsigned long long n = __builtin_readcyclecounter();
#pragma unroll
      for (int j = 0; j < 6; ++j)
        if (j == i) acc[j] += n - t;
      t = n;
    }
  }
  __device__ __forceinline__ void done() {
    if (out && (threadIdx.x & 63) == 0) {
      const unsigned long long r = atomicAdd(out, 1ull);
      if (r < out[1]) {
#pragma unroll
        for (int j = 0; j < 6; ++j) out[8 + 8 * r + j] = acc[j];
      }
    }
  }
};
unsigned long long* phase_clock_ptr();   // kpconv_fused.hip

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace d3f
