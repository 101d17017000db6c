// Pure f32-MFMA issue-rate microbenchmark: what v_mfma_f32_32x32x2_f32 sustains on this chip with nothing else going
// on (no LDS, no memory), to put the conv kernels' numbers next to the spec peak of 157.3 TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512, 2) void mfma_loop(float* out, int iters, float a, float b) {
  f32x16 acc[8];
  for (int t = 0; t < 8; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
  }
  float s = 0.f;
  for (int t = 0; t < 8; ++t)
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 8 * 512 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int blocks_per_cu = 1; blocks_per_cu <= 1; ++blocks_per_cu)
    for (int ms_target = 0; ms_target < 3; ++ms_target) {
      const int iters = 20000 << ms_target;   // ~5, 10, 20 ms
      const int grid = 256 * blocks_per_cu;
      hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(512), 0, 0, d, 100, 1.0f, 0.5f);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(512), 0, 0, d, iters, 1.0f, 0.5f);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      const double flops = (double)grid * 8 /*waves*/ * iters * 8 * (32.0 * 32 * 2 * 2);
      printf("grid %d x 512 threads, %d iters: %.3f ms  %.1f TFLOP/s  (implied clock %.2f GHz at 256 FLOP/clk/CU)\n", grid,
             iters, ms, flops / ms / 1e9, flops / ms / 1e9 / (256.0 * 256 / 1e3) );
    }
  return 0;
}
