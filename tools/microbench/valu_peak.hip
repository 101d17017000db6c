// What the f32 vector ALU sustains on independent accumulator chains: v_fmac_f32 and v_pk_fma_f32, 1 / 2 / 4 waves per SIMD,
// with and without wave-uniform 16-byte LDS reads feeding one operand (the shape of conv_n8.h).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_peak.hip -o /tmp/valu_peak && /tmp/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void k(float* out, int iters, long long* cyc) {
  const long long c0 = clock64();
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += THREADS) lds[i] = (float)(i & 7) * 0.125f;
  __syncthreads();
  float acc[64];
#pragma unroll
  for (int r = 0; r < 64; ++r) acc[r] = 0.f;
  float x[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) x[r] = (float)(threadIdx.x + r) * 1e-3f;
  int off = 0;
  for (int i = 0; i < iters; ++i) {
    float w[8] = {0.5f, 0.25f, 0.125f, 1.0f, 2.0f, 0.75f, 1.5f, 0.3f};
    if (MODE & 2) {   // weights from LDS: two wave-uniform b128 reads per 64 FMAs
      const float4 a = *reinterpret_cast<const float4*>(lds + off);
      const float4 b = *reinterpret_cast<const float4*>(lds + off + 4);
      w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
      off = (off + 8) & 4095;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(w[j]));
    }
    if (MODE & 1) {
#pragma unroll
      for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f32x2 a = {acc[p * 8 + 2 * j], acc[p * 8 + 2 * j + 1]};
          const f32x2 ww = {w[2 * j], w[2 * j + 1]};
          const f32x2 xx = {x[p], x[p]};
          asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a) : "v"(xx), "v"(ww));
          acc[p * 8 + 2 * j] = a[0];
          acc[p * 8 + 2 * j + 1] = a[1];
        }
    } else {
#pragma unroll
      for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int j = 0; j < 8; ++j) asm("v_fmac_f32 %0, %1, %2" : "+v"(acc[p * 8 + j]) : "v"(x[p]), "v"(w[j]));
    }
  }
  float s = 0.f;
  for (int r = 0; r < 64; ++r) s += acc[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = clock64() - c0;
}

template <int MODE, int THREADS>
void run(float* d, const char* what) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 200000;
  long long* dc;
  hipMalloc(&dc, 8);
  hipLaunchKernelGGL((k<MODE, THREADS>), dim3(256), dim3(THREADS), 0, 0, d, 100, dc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, THREADS>), dim3(256), dim3(THREADS), 0, 0, d, iters, dc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = 256.0 * THREADS * iters * 64 * 2;
  long long hc = 0;
  hipMemcpy(&hc, dc, 8, hipMemcpyDeviceToHost);
  printf("%-40s %d waves/SIMD %8.3f ms  %6.1f TFLOP/s   shader clock %.0f MHz, %.2f cycles per 64 lane-FMAs per SIMD\n", what, THREADS / 256, ms,
         flops / ms / 1e9, hc / (ms * 1e3), (double)hc / ((double)iters * 64.0 * (THREADS / 256)));
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 1024 * sizeof(float));
  run<0, 256>(d, "v_fmac_f32");
  run<0, 512>(d, "v_fmac_f32");
  run<0, 1024>(d, "v_fmac_f32");
  run<1, 256>(d, "v_pk_fma_f32");
  run<1, 512>(d, "v_pk_fma_f32");
  run<1, 1024>(d, "v_pk_fma_f32");
  run<2, 512>(d, "v_fmac_f32 + LDS weights");
  run<2, 1024>(d, "v_fmac_f32 + LDS weights");
  run<3, 512>(d, "v_pk_fma_f32 + LDS weights");
  run<3, 1024>(d, "v_pk_fma_f32 + LDS weights");
  return 0;
}
