// Does the L2 float-atomic rate depend on the width of the contiguous segment a wave touches per row?
// hipcc --offload-arch=gfx950 -O3 atomic_width.hip -o atomic_width && ./atomic_width
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

template <int W>  // W = floats per row segment covered by consecutive lanes (16, 32, 64)
__global__ void scatter_kernel(const int* __restrict__ rows, int n_seg, int C, float* __restrict__ dst) {
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  constexpr int SEG = 64 / W;  // row segments per wave instruction
  for (int it = 0; it < 8; ++it) {
    const long s = (wave * 8 + it) * SEG + lane / W;
    if (s < n_seg) {
      const int r = rows[s];
      atomicAdd(&dst[(long)r * C + (lane % W)], 1.0f);
    }
  }
}

int main() {
  const int N = 38000, C = 64;
  const long total_floats = 51L * 1000 * 1000;  // like the level-0 KPConv grad-input scatter
  float* dst;
  hipMalloc(&dst, sizeof(float) * (size_t)N * C);
  hipMemset(dst, 0, sizeof(float) * (size_t)N * C);
  for (int W : {16, 32, 64}) {
    const long n_seg = total_floats / W;
    std::vector<int> h(n_seg);
    srand(1);
    for (long i = 0; i < n_seg; ++i) h[i] = rand() % N;
    int* rows;
    hipMalloc(&rows, sizeof(int) * n_seg);
    hipMemcpy(rows, h.data(), sizeof(int) * n_seg, hipMemcpyHostToDevice);
    const int SEG = 64 / W;
    const long waves = (n_seg + 8L * SEG - 1) / (8L * SEG);
    const int blocks = (int)((waves * 64 + 255) / 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (W == 16) scatter_kernel<16><<<blocks, 256>>>(rows, (int)n_seg, C, dst);
      if (W == 32) scatter_kernel<32><<<blocks, 256>>>(rows, (int)n_seg, C, dst);
      if (W == 64) scatter_kernel<64><<<blocks, 256>>>(rows, (int)n_seg, C, dst);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("segment %3d floats (%3d B): %ld segments, %.1f us, %.1f G floats/s, %.2f G segments/s\n", W, 4 * W, n_seg,
           ms * 1e3, total_floats / (ms * 1e-3) / 1e9, n_seg / (ms * 1e-3) / 1e9);
    hipFree(rows);
  }
  return 0;
}
