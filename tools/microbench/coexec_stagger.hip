// Do the vector ALU and the matrix pipe of a SIMD execute together on MI355X -- and does it take a STAGGER of the two waves
// that share a SIMD to make them?  (MI355X_MICROARCH.md, "Two waves per SIMD", item 9: on a gather + MFMA
// kernel, delaying waves 4-7 by half a block cut 4-8 % and SQ_VALU_MFMA_COEXEC_CYCLES rose.)
//
// 512-thread workgroups (two waves per SIMD: waves w and w + 4 share one), one per CU.  A block = NV independent v_fma_f32
// (four chains) + NM v_mfma_f32_32x32x2_f32 (two accumulators) + s_barrier -- the shape of conv_wino44_kernel's super-step
// (VALU transform + MFMAs between two barriers).  Modes:
//   0 lockstep      every wave: VALU then MFMA                 (both partners in the same phase at the same time)
//   1 stagger hi    waves 4-7: MFMA then VALU, waves 0-3 VALU then MFMA   (the guide's recipe: split by wave >= 4)
//   2 stagger odd   odd waves: MFMA then VALU                  (the guide's counter-example: split by parity -- NOT SIMD partners)
//   3 VALU only     4 MFMA only                                (the two components alone)
// If the pipes co-execute, mode 1 approaches max(VALU, MFMA) per block where mode 0 pays VALU + MFMA.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/coexec_stagger.hip -o tools/microbench/coexec_stagger.bin
//   tools/microbench/coexec_stagger.bin            (all modes)  |  coexec_stagger.bin MODE  (one mode, for rocprofv3 --pmc)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int NM>
__global__ __launch_bounds__(512, 2) void blocks(float* out, int iters, int mode, float a, float b) {
  const int wave = threadIdx.x >> 6;
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  float v0 = a, v1 = b, v2 = a + b, v3 = a - b;
  const bool mfma_first = (mode == 1 && wave >= 4) || (mode == 2 && (wave & 1));
  const bool do_v = mode != 4, do_m = mode != 3;
  auto valu = [&]() {
#pragma unroll
    for (int k = 0; k < NV / 4; ++k) {
      asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5"
                   : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(a), "v"(b));
    }
  };
  auto mfma = [&]() {
#pragma unroll
    for (int k = 0; k < NM / 2; ++k) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
    }
  };
  for (int i = 0; i < iters; ++i) {
    if (mfma_first) {
      if (do_m) mfma();
      __builtin_amdgcn_sched_barrier(0);
      if (do_v) valu();
    } else {
      if (do_v) valu();
      __builtin_amdgcn_sched_barrier(0);
      if (do_m) mfma();
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  }
  float s = v0 + v1 + v2 + v3;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, int NM>
void run(float* d, int mode_only) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const char* names[5] = {"lockstep", "stagger waves 4-7", "stagger odd waves", "VALU only", "MFMA only"};
  const int iters = 20000;
  float t[5] = {0, 0, 0, 0, 0};
  for (int mode = 0; mode < 5; ++mode) {
    if (mode_only >= 0 && mode != mode_only) continue;
    hipLaunchKernelGGL((blocks<NV, NM>), dim3(256), dim3(512), 0, 0, d, 200, mode, 1.0f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((blocks<NV, NM>), dim3(256), dim3(512), 0, 0, d, iters, mode, 1.0f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&t[mode], e0, e1);
    printf("NV %3d NM %2d  %-18s %8.3f ms  %7.1f ns per block\n", NV, NM, names[mode], t[mode], t[mode] * 1e6 / iters);
  }
  if (mode_only < 0)
    printf("   VALU + MFMA alone = %.3f ms; lockstep / that = %.2f; stagger 4-7 / lockstep = %.3f; stagger odd / lockstep = %.3f\n",
           t[3] + t[4], t[0] / (t[3] + t[4]), t[1] / t[0], t[2] / t[0]);
}

int main(int argc, char** argv) {
  float* d;
  hipMalloc(&d, 256 * 512 * sizeof(float));
  const int mode_only = argc > 1 ? atoi(argv[1]) : -1;
  run<128, 8>(d, mode_only);     // 128 v_fma ~ 512 issue cycles per wave, 8 MFMAs = 512 matrix-pipe cycles per wave
  if (mode_only < 0) {
    run<64, 8>(d, -1);
    run<32, 8>(d, -1);           // the conv_wino44 proportion: ~20 VALU per 9 MFMAs
  }
  return 0;
}
