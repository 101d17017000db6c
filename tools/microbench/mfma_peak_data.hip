// Does the f32 matrix pipe hold its clock on REAL operands?  mfma_peak.hip feeds every MFMA the constants (1.0, 0.5) and sustains
// 2.39 GHz; conv_wino44_kernel runs at 1.82-1.86 GHz (profiles/r04/wino44_clock.json).  Here the same MFMA-only loop (no LDS, no
// memory, two waves per SIMD, eight independent accumulators per wave) is fed
//   mode 0: the constants;  mode 1: one random operand pair per lane, the same for every MFMA;
//   mode 2: sixteen random operand pairs per lane, a different one for every MFMA (what a convolution's operands look like)
// and the shader clock is read inside the kernel (s_memtime against the 100 MHz s_memrealtime, averaged over the workgroups).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_peak_data.hip -o tools/microbench/mfma_peak_data.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512, 2) void mfma_loop(const float* __restrict__ ops, float* out, long long* clk, int iters, int mode) {
  f32x16 acc[8];
  for (int t = 0; t < 8; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float a[16], b[16];
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  for (int k = 0; k < 16; ++k) {
    const float ra = ops[(gid * 32 + 2 * k) & 0xfffff], rb = ops[(gid * 32 + 2 * k + 1) & 0xfffff];
    a[k] = mode == 0 ? 1.0f : mode == 1 ? ops[(gid * 32) & 0xfffff] : ra;
    b[k] = mode == 0 ? 0.5f : mode == 1 ? ops[(gid * 32 + 1) & 0xfffff] : rb;
  }
  const long long w0 = wall_clock64(), c0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[8 * h + t], b[8 * h + t], acc[t], 0, 0, 0);
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int t = 0; t < 8; ++t)
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[gid] = s;
  if (threadIdx.x == 0) {
    atomicAdd((unsigned long long*)&clk[0], (unsigned long long)(c1 - c0));
    atomicAdd((unsigned long long*)&clk[1], (unsigned long long)(w1 - w0));
  }
}

int main(int argc, char** argv) {
  const int only = argc > 1 ? atoi(argv[1]) : -1;
  float *d, *ops;
  long long* clk;
  hipMalloc(&d, 256 * 512 * sizeof(float));
  hipMalloc(&ops, (1 << 20) * sizeof(float));
  hipMalloc(&clk, 16);
  std::vector<float> h(1 << 20);
  srand(7);
  for (auto& v : h) v = (float)rand() / RAND_MAX * 2.0f - 1.0f;      // uniform in [-1, 1): sums stay finite for millions of MFMAs
  hipMemcpy(ops, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const char* names[3] = {"constants (1.0, 0.5)", "one random pair per lane", "sixteen random pairs per lane, rotating"};
  for (int mode = 0; mode < 3; ++mode) {
    if (only >= 0 && mode != only) continue;
    for (int rep = 0; rep < 2; ++rep) {
      const int iters = 40000, grid = 256;
      hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(512), 0, 0, ops, d, clk, 200, mode);
      hipMemset(clk, 0, 16);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(512), 0, 0, ops, d, clk, iters, mode);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      long long c[2];
      hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
      const double flops = (double)grid * 8 /*waves*/ * iters * 16 * (32.0 * 32 * 2 * 2);
      printf("mode %d %-40s %8.3f ms  %6.1f TFLOP/s  shader clock %4.0f MHz (s_memtime / s_memrealtime)\n", mode, names[mode], ms,
             flops / ms / 1e9, 100.0 * (double)c[0] / (double)c[1]);
    }
  }
  return 0;
}
