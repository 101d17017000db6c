// What can the L2 -> LDS feed (global_load_lds_dwordx4 in a ring of three stages, counted vmcnt) sustain per CU and chip-wide,
// alone and beside LDS-fed f32 MFMAs?  Sizes the operand-stream side of the convolution kernels:
//   * a Winograd F(4x4,3x3) tile (64 couts x 32 tiles) needs ~20 B/clk/CU at the full MFMA rate, F(2x2,3x3) ~11, the
//     64x64 1x1 tiles 16;
//   * "shared" bytes are read by every workgroup of an XCD group at the same stage index (weights: L2 hits after the first
//     reader), "private" bytes by one workgroup only (its input patch: MALL / HBM).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/ldsdma_feed.hip -o tools/microbench/ldsdma_feed.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("%s failed: %s\n", #x, hipGetErrorString(e_));                        \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

// WAVES waves; per stage every wave issues SH dwordx4 LDS-DMA instructions from the group-shared stream and PR from the
// workgroup-private stream (1 KB each), then MF MFMAs whose A operand is a dword read from the landed stage.
template <int WAVES, int SH, int PR, int MF>
__global__ __launch_bounds__(WAVES * 64) void feed_kernel(const float* __restrict__ shared, long long shared_group_floats,
                                                          const float* __restrict__ priv, long long priv_wg_floats,
                                                          int nstages, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int STAGE = WAVES * (SH + PR) * 256;   // floats
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float* sh = shared + (long long)(blockIdx.x & 7) * shared_group_floats + (wave * SH) * 256 + lane * 4;
  const float* pr = priv + (long long)blockIdx.x * priv_wg_floats + (wave * PR) * 256 + lane * 4;
  auto issue = [&](int s, int buf) {
    float* L = smem + buf * STAGE + wave * (SH + PR) * 256;
#pragma unroll
    for (int k = 0; k < SH; ++k)
      __builtin_amdgcn_global_load_lds((gptr_t)(sh + (long long)s * WAVES * SH * 256 + k * 256), (lptr_t)(L + k * 256), 16, 0, 0);
#pragma unroll
    for (int k = 0; k < PR; ++k)
      __builtin_amdgcn_global_load_lds((gptr_t)(pr + (long long)s * WAVES * PR * 256 + k * 256), (lptr_t)(L + (SH + k) * 256), 16, 0,
                                       0);
  };
  constexpr int NACC = MF >= 8 ? 8 : (MF > 0 ? MF : 1);
  f32x16 acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
  const float bval = 1.0f + lane * 1e-3f;

  issue(0, 0);
  if (nstages > 1) issue(1, 1);
  int cur = 0;
  for (int it = 0; it < nstages; ++it) {
    if (it + 1 < nstages)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SH + PR) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const int nb = (cur + 1 == 3) ? 0 : cur + 1, fb = (nb + 1 == 3) ? 0 : nb + 1;
    if (it + 2 < nstages) issue(it + 2, fb);
    if constexpr (MF > 0) {
      // operands: one ds_read_b32 per MFMA, four in flight ahead of their use, exact lgkmcnt (see conv_kernel.h)
      const unsigned base = lds0 + (unsigned)(cur * STAGE + lane) * 4u;
      float a[2][4];
      auto load4 = [&](float (&o)[4], int g) {
        const unsigned ad = base + (unsigned)((g * 4) % ((SH + PR) * WAVES * 4)) * 256u;
        asm volatile("ds_read_b32 %0, %1" : "=v"(o[0]) : "v"(ad));
        asm volatile("ds_read_b32 %0, %1 offset:256" : "=v"(o[1]) : "v"(ad));
        asm volatile("ds_read_b32 %0, %1 offset:512" : "=v"(o[2]) : "v"(ad));
        asm volatile("ds_read_b32 %0, %1 offset:768" : "=v"(o[3]) : "v"(ad));
      };
      load4(a[0], 0);
      constexpr int NG = (MF + 3) / 4;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        float(&o)[4] = a[g & 1];
        if (g + 1 < NG) {
          load4(a[(g + 1) & 1], g + 1);
          asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]));
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]));
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (g * 4 + k < MF) acc[(g * 4 + k) % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(o[k], bval, acc[(g * 4 + k) % NACC], 0, 0, 0);
      }
    }
    cur = nb;
  }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  if (MF == 0) s = smem[tid];
  out[(long long)blockIdx.x * WAVES * 64 + tid] = s;
}

template <int WAVES, int SH, int PR, int MF>
void run(const char* what, const float* shared, const float* priv, float* out, int grid, int nstages, size_t lds_min) {
  constexpr int STAGE_B = WAVES * (SH + PR) * 1024;
  size_t lds = 3 * (size_t)STAGE_B;
  if (lds < lds_min) lds = lds_min;
  auto fn = feed_kernel<WAVES, SH, PR, MF>;
  CK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const long long sgf = (long long)nstages * WAVES * SH * 256, pwf = (long long)nstages * WAVES * PR * 256;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(WAVES * 64), lds, 0, shared, sgf, priv, pwf, nstages, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep && ms < best) best = ms;
  }
  const double bytes = (double)grid * nstages * STAGE_B;
  const double flops = (double)grid * nstages * WAVES * MF * 4096.0;
  const double rounds = grid / 256.0;
  printf("%-34s waves %d  stage %2d KB (%d shared + %d private per wave)  lds %3zu KB  grid %4d  %4d stages: %8.1f us  "
         "%6.2f TB/s into LDS  %5.1f GB/s per CU  %6.1f TFLOP/s  (stage time %.2f us per workgroup-slot)\n",
         what, WAVES, STAGE_B / 1024, SH, PR, lds / 1024, grid, nstages, best * 1e3, bytes / best / 1e9, bytes / best / 1e6 / 256.0,
         flops / best / 1e9, best * 1e3 / nstages / (rounds < 1 ? 1 : rounds));
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
}

int main() {
  const int NST = 128;
  const size_t shared_bytes = (size_t)8 * NST * 8 * 4 * 1024;         // 8 groups x stages x waves x SH KB (largest config)
  const size_t priv_bytes = (size_t)1024 * NST * 8 * 2 * 1024;        // up to 1024 workgroups x stages x waves x PR KB
  float *sh, *pr, *out;
  CK(hipMalloc(&sh, shared_bytes));
  CK(hipMalloc(&pr, priv_bytes));
  CK(hipMalloc(&out, (size_t)1024 * 512 * 4));
  CK(hipMemset(sh, 0, shared_bytes));
  CK(hipMemset(pr, 0, priv_bytes));
  const size_t ONE = 100 * 1024;    // > 80 KB of LDS: one workgroup per CU, like the Winograd kernels
  for (int grid : {256, 512}) {
    printf("---- grid %d\n", grid);
    run<8, 4, 1, 0>("feed only, wino4-like stage", sh, pr, out, grid, NST, ONE);
    run<8, 4, 1, 32>("wino4-like: 32 MFMA/wave/stage", sh, pr, out, grid, NST, ONE);
    run<8, 2, 1, 0>("feed only, F(4x4) KC=2 stage", sh, pr, out, grid, NST, ONE);
    run<8, 2, 1, 9>("F(4x4) KC=2: 9 MFMA/wave/stage", sh, pr, out, grid, NST, ONE);
    run<8, 4, 2, 0>("feed only, F(4x4) KC=4 stage", sh, pr, out, grid, NST, ONE);
    run<8, 4, 2, 18>("F(4x4) KC=4: 18 MFMA/wave/stage", sh, pr, out, grid, NST, ONE);
    run<8, 4, 0, 0>("feed only, shared only", sh, pr, out, grid, NST, ONE);
    run<8, 0, 2, 0>("feed only, private only", sh, pr, out, grid, NST, ONE);
  }
  printf("---- 4-wave workgroups (1x1 class), two or more per CU\n");
  for (int grid : {256, 512, 1024}) {
    run<4, 2, 2, 0>("feed only, 64x64xK32 stage", sh, pr, out, grid, NST, 0);
    run<4, 2, 2, 16>("64x64xK32: 16 MFMA/wave/stage", sh, pr, out, grid, NST, 0);
    run<4, 4, 0, 16>("same, all bytes shared", sh, pr, out, grid, NST, 0);
  }
  return 0;
}
