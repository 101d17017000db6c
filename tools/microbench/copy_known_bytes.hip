// A float4 copy of a known size: the absolute reference for the TCC read / write request counters (tools/experiments/traffic_r4.sh).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/copy_known_bytes.hip -o tools/microbench/copy_known_bytes.bin
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void copy4(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
int main() {
  const size_t bytes = 1ull << 30;   // 1 GiB read + 1 GiB written: far beyond the 256 MB Infinity Cache
  float4 *a, *b;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes);
  hipMemset(a, 1, bytes);
  hipDeviceSynchronize();
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(copy4, dim3(256 * 8), dim3(256), 0, 0, a, b, bytes / 16);
  hipDeviceSynchronize();
  printf("copied %zu bytes x 3\n", bytes);
  return 0;
}
