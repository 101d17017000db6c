// What the f32 MFMA pipe sustains when the wave also does what a conv kernel must do between MFMAs: LDS operand reads
// and a few VALU ops per MFMA (mode 1, 2) -- a ceiling for kernels of that shape, next to the pure-MFMA 156.8 TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_mix.hip -o /tmp/mfma_mix && /tmp/mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(512, 2) void mix(float* out, int iters, long long* cyc) {
  const long long c0 = clock64();
  __shared__ float lds[16384];
  for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = (float)(i & 7) * 0.125f;
  __syncthreads();
  f32x16 acc[8];
  for (int t = 0; t < 8; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int lane = threadIdx.x & 63;
  int off = lane;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      float a = 1.0f, b = 0.5f;
      if (MODE >= 1) {                       // two LDS operand reads per MFMA (conflict-free)
        a = lds[(off + t * 64) & 16383];
        b = lds[(off + t * 64 + 4096) & 16383];
      }
      if (MODE >= 2) {                       // + four dependent VALU ops per MFMA (the transform's share)
        b = b - a;
        b = b + 0.25f;
        b = b * 0.5f + a;
        b = b - 0.125f;
      }
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    }
    off = (off + 512) & 16383;
  }
  float s = 0.f;
  for (int t = 0; t < 8; ++t)
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = clock64() - c0;   // shader-clock cycles of one wave's lifetime
}

template <int MODE>
void run(float* d, const char* what) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 40000;
  long long* dc;
  hipMalloc(&dc, 8);
  hipLaunchKernelGGL(mix<MODE>, dim3(256), dim3(512), 0, 0, d, 100, dc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(mix<MODE>, dim3(256), dim3(512), 0, 0, d, iters, dc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = 256.0 * 8 * iters * 8 * (32.0 * 32 * 2 * 2);
  long long hc = 0;
  hipMemcpy(&hc, dc, 8, hipMemcpyDeviceToHost);
  printf("%-46s %.3f ms  %.1f TFLOP/s   shader clock %.0f MHz, %.1f cycles per MFMA per SIMD\n", what, ms, flops / ms / 1e9,
         hc / (ms * 1e3), (double)hc / (iters * 8.0 * 2));
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 512 * sizeof(float));
  run<0>(d, "MFMA only");
  run<1>(d, "MFMA + 2 LDS reads per MFMA");
  run<2>(d, "MFMA + 2 LDS reads + 4 VALU per MFMA");
  return 0;
}
