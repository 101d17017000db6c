// Same question as mfma_mix.hip for the 16x16x4 f32 MFMA: does the narrower instruction sustain a higher rate (clock)
// under the LDS-read + VALU mix of a conv kernel?  Each 32x32x2 is replaced by two 16x16x4 (same FLOPs, same operand
// dwords per FLOP when tiles are register-reused the same way; here: no reuse, 2 LDS reads per MFMA as in mfma_mix).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_mix16.hip -o /tmp/mfma_mix16 && /tmp/mfma_mix16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512, 2) void mix(float* out, int iters) {
  __shared__ float lds[16384];
  for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = (float)(i & 7) * 0.125f;
  __syncthreads();
  f32x4 acc[16];
  for (int t = 0; t < 16; ++t)
    for (int r = 0; r < 4; ++r) acc[t][r] = 0.f;
  const int lane = threadIdx.x & 63;
  int off = lane;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      float a = 1.0f, b = 0.5f;
      if (MODE >= 1) {
        a = lds[(off + t * 64) & 16383];
        b = lds[(off + t * 64 + 4096) & 16383];
      }
      if (MODE >= 2) {                       // two VALU per 16x16x4 = four per 32x32x2-equivalent
        b = b - a;
        b = b * 0.5f + a;
      }
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
    }
    off = (off + 512) & 16383;
  }
  float s = 0.f;
  for (int t = 0; t < 16; ++t)
    for (int r = 0; r < 4; ++r) s += acc[t][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(float* d, const char* what) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 40000;
  hipLaunchKernelGGL(mix<MODE>, dim3(256), dim3(512), 0, 0, d, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(mix<MODE>, dim3(256), dim3(512), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = 256.0 * 8 * iters * 16 * (16.0 * 16 * 4 * 2);
  printf("%-52s %.3f ms  %.1f TFLOP/s\n", what, ms, flops / ms / 1e9);
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 512 * sizeof(float));
  run<0>(d, "16x16x4 MFMA only");
  run<1>(d, "16x16x4 MFMA + 2 LDS reads per MFMA");
  run<2>(d, "16x16x4 MFMA + 2 LDS reads + 2 VALU per MFMA");
  return 0;
}
