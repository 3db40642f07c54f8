// Cycles of ONE compute phase of the GEMM tile kernel (csrc/gemm.hip) in isolation: 32 (FM=1) or 64 (FM=2)
// v_mfma_f32_16x16x4_f32 fed by ds_read_b64 fragment reads from a resident LDS tile -- no global loads, no staging.
// Variants: with/without the LDS reads, with/without the MFMAs, 1 or 2 workgroups per CU.
// hipcc --offload-arch=gfx950 -O3 mfma_lds_phase.hip -o mfma_lds_phase
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int LD = 68;
template <int FM, int MODE>  // MODE bit0: LDS reads, bit1: MFMA
__global__ __launch_bounds__(256) void k(int iters, float* out, long long* cyc) {
  __shared__ __attribute__((aligned(16))) float lds[(32 * FM + 64) * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;
  for (int i = tid; i < (32 * FM + 64) * LD; i += 256) lds[i] = (float)(i % 7) * 0.25f;
  __syncthreads();
  const float* As = lds;
  const float* Bs = lds + 32 * FM * LD;
  f32x4 acc[FM][2];
  for (int a = 0; a < FM; ++a) for (int b = 0; b < 2; ++b) acc[a][b] = (f32x4){0, 0, 0, 0};
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    float a[2][2][FM][2], b[2][2][2][2];
    auto read = [&](int ps, int slot) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int kk = 2 * ps + h;
#pragma unroll
        for (int f = 0; f < FM; ++f) {
          float2 v = (MODE & 1) ? *(const float2*)(As + (wr * 16 * FM + f * 16 + li) * LD + kk * 8 + 2 * lg) : make_float2(1.f + it, 2.f);
          a[slot][h][f][0] = v.x; a[slot][h][f][1] = v.y;
        }
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          float2 v = (MODE & 1) ? *(const float2*)(Bs + (wc * 32 + f * 16 + li) * LD + kk * 8 + 2 * lg) : make_float2(3.f, 4.f + it);
          b[slot][h][f][0] = v.x; b[slot][h][f][1] = v.y;
        }
      }
    };
    read(0, 0);
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int cur = ps & 1;
      if (ps + 1 < 4) read(ps + 1, cur ^ 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int fa = 0; fa < FM; ++fa)
#pragma unroll
            for (int fb = 0; fb < 2; ++fb) {
              if (MODE & 2) acc[fa][fb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][h][fa][s], b[cur][h][fb][s], acc[fa][fb], 0, 0, 0);
              else acc[fa][fb][0] += a[cur][h][fa][s] * b[cur][h][fb][s];
            }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = clock64();
  float s = 0;
  for (int a = 0; a < FM; ++a) for (int b = 0; b < 2; ++b) s += acc[a][b][0] + acc[a][b][3];
  if (s == 12345.f) out[0] = s;
  if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int FM, int MODE>
void run(int wgs, const char* what) {
  float* d; long long* c; hipMalloc(&d, 4); hipMalloc(&c, 8);
  const int iters = 2000;
  k<FM, MODE><<<wgs, 256>>>(iters, d, c);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<FM, MODE><<<wgs, 256>>>(iters, d, c);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  printf("FM=%d %-22s wgs %4d: %8.1f ns/phase (wall), %7.1f s_memtime ticks/phase; MFMA-only bound %d cycles\n", FM, what, wgs,
         ms * 1e6 / iters, (double)h / iters, 32 * 16 * FM * 2);
}
int main() {
  for (int wgs : {256, 512}) {
    run<1, 3>(wgs, "lds reads + mfma"); run<1, 2>(wgs, "mfma only"); run<1, 1>(wgs, "lds reads only");
    run<2, 3>(wgs, "lds reads + mfma"); run<2, 2>(wgs, "mfma only"); run<2, 1>(wgs, "lds reads only");
  }
  return 0;
}
