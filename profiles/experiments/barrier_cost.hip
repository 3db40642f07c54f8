// How long does one __syncthreads() take in a 256-thread workgroup, by LDS footprint and grid size?
// hipcc --offload-arch=gfx950 -O3 profiles/experiments/barrier_cost.hip -o /tmp/barrier_cost && /tmp/barrier_cost
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDS_FLOATS>
__global__ __launch_bounds__(256) void k(int iters, float* out) {
  __shared__ float lds[LDS_FLOATS];
  float acc = threadIdx.x;
  for (int t = 0; t < iters; ++t) {
    lds[(threadIdx.x * 33 + t) % LDS_FLOATS] = acc;
    __syncthreads();
    acc += lds[(threadIdx.x * 7 + t) % LDS_FLOATS];
    __syncthreads();
  }
  if (acc == 12345.678f) out[0] = acc;
}
template <int L>
void run(int wgs, int iters, float* d) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<L><<<wgs, 256>>>(iters, d);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 10; ++r) k<L><<<wgs, 256>>>(iters, d);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("lds %6d B  wgs %5d iters %4d : %.2f us/launch, %.1f ns per barrier pair\n", L * 4, wgs, iters, ms * 100, ms * 1e5 / iters);
}
int main() {
  float* d; hipMalloc(&d, 4);
  // workgroups resident per CU by LDS footprint: time of 2048 workgroups / time of 256
  run<1024>(256, 1024, d);
  run<1024>(2048, 1024, d);
  run<4096>(2048, 1024, d);   // 16 KB
  run<5120>(2048, 1024, d);   // 20 KB
  run<6144>(2048, 1024, d);   // 24 KB
  run<8192>(2048, 1024, d);   // 32 KB
  run<9216>(2048, 1024, d);   // 36 KB
  run<10240>(2048, 1024, d);  // 40 KB
  run<12288>(2048, 1024, d);  // 48 KB
  run<14336>(2048, 1024, d);  // 56 KB
  run<16384>(2048, 1024, d);  // 64 KB
  int v = 0;
  hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, 0);
  printf("MaxSharedMemoryPerMultiprocessor %d\n", v);
  hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, 0);
  printf("MaxSharedMemoryPerBlock %d\n", v);
  int nb = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k<14336>, 256, 0);
  printf("occupancy API: 56 KB kernel -> %d blocks/CU\n", nb);
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k<9216>, 256, 0);
  printf("occupancy API: 36 KB kernel -> %d blocks/CU\n", nb);
  return 0;
}
