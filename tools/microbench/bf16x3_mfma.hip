// Split-bf16 products for the F(4x4,3x3) GEMMs: a measured answer (review round 4, item 9; docs/EXPERIMENTS.md R5-4).
//
// The Winograd k-step of csrc/conv_wino44.h is 36 independent [64 couts x K] x [K x 32 tiles] GEMMs on v_mfma_f32_32x32x2_f32
// (64 cycles per instruction and SIMD for 2 k: 64 FLOP/clk/SIMD).  v_mfma_f32_32x32x16_bf16 does 16 k in 32 cycles -- 16x the
// rate -- but takes bf16 operands.  An f32 value splits EXACTLY into three bf16 planes x = x0 + x1 + x2 (8 + 8 + 8 significant
// bits); the product x*y is then the sum of nine plane products, of which the six largest (x0y0, x0y1, x1y0, x0y2, x2y0, x1y1)
// keep a relative error of ~3 * 2^-24, the three largest (x0y0, x0y1, x1y0) of ~2^-16.  Six bf16 MFMAs per 16 channels replace
// eight f32 MFMAs: 192 against 512 matrix-pipe cycles.
//
// Part 1 (rate): a workgroup of eight waves per CU runs the k-step's matrix work for 16 input channels per trip -- nine
// accumulator tiles per wave, every operand read from LDS as the conv kernel does (f32 form: two ds_read_b32 per MFMA; bf16
// form: three A planes + three B planes per tile as ds_read_b128, 1.5x the operand bytes) -- optionally with V vector-ALU
// operations per MFMA beside it (the input transform + the plane split would live there).  Reported: time per trip, the
// f32-equivalent TFLOP/s, the shader clock (clock64 against wall time).
// Part 2 (error): one [32 x K] x [K x 32] product (K = 1024, one Winograd position of a 1024-channel layer) computed on the
// matrix cores in the four forms, against an f64 reference on the host: relative RMS and maximum error.
//
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/bf16x3_mfma.hip -o /tmp/bf16x3 && /tmp/bf16x3
//   (counters: rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -- /tmp/bf16x3 rate)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));                \
      return 1;                                                                     \
    }                                                                               \
  } while (0)

// ------------------------------------------------------------------------------------------------ part 1: rate
// FORM 0: f32 MFMAs (72 per wave and trip); FORM 1: bf16 x 3 planes, six products (54 per wave and trip); FORM 2: three products (27)
template <int FORM, int NVALU>
__global__ __launch_bounds__(512, 1) void kstep(float* out, int trips, long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) float lds[];     // 128 KB of operand image (dynamic: above the 64 KB static limit)
  for (int i = threadIdx.x; i < 32768; i += 512) lds[i] = (float)((i * 7) & 15) * 0.0625f - 0.4f;
  __syncthreads();
  const long long c0 = clock64();
  f32x16 acc[9];
  for (int t = 0; t < 9; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int off = (lane + wave * 320) & 8191;
  float side = 0.25f * lane;
  for (int i = 0; i < trips; ++i) {
    if (FORM == 0) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float a = lds[(off + (t * 8 + k) * 64) & 32767];
          const float b = lds[(off + (t * 8 + k) * 64 + 16384) & 32767];
#pragma unroll
          for (int v = 0; v < NVALU; ++v) side = side * 0.999f + a;      // dependent vector-ALU work beside the matrix work
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
        }
    } else {
      constexpr int NP = FORM == 1 ? 6 : 3;
      // plane products in the order (a-plane, b-plane): 00 01 10 | 02 20 11
      constexpr int pa[6] = {0, 0, 1, 0, 2, 1}, pb[6] = {0, 1, 0, 2, 0, 1};
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        bf16x8 A[3], B[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          if (FORM == 2 && p == 2) break;
          const float4 qa = *reinterpret_cast<const float4*>(&lds[((off + (t * 6 + p) * 64) * 4) & 32767]);
          const float4 qb = *reinterpret_cast<const float4*>(&lds[((off + (t * 6 + 3 + p) * 64) * 4) & 32767]);
          memcpy(&A[p], &qa, 16);
          memcpy(&B[p], &qb, 16);
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#pragma unroll
          for (int v = 0; v < NVALU; ++v) side = side * 0.999f + 0.5f;
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[pa[p]], B[pb[p]], acc[t], 0, 0, 0);
        }
      }
    }
    off = (off + 576) & 8191;
  }
  float s = side;
  for (int t = 0; t < 9; ++t)
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = clock64() - c0;
}

template <int FORM, int NVALU>
int run_rate(float* d, long long* dc, const char* what) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int trips = 4000;
  CK(hipFuncSetAttribute((const void*)kstep<FORM, NVALU>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  hipLaunchKernelGGL((kstep<FORM, NVALU>), dim3(256), dim3(512), 131072, 0, d, 50, dc);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((kstep<FORM, NVALU>), dim3(256), dim3(512), 131072, 0, d, trips, dc);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  long long hc = 0;
  CK(hipMemcpy(&hc, dc, 8, hipMemcpyDeviceToHost));
  // f32-equivalent work of a trip: 9 tiles x 16 channels x 32 x 32 x 2 FLOP per wave, 8 waves, 256 workgroups
  const double flops = 256.0 * 8 * trips * 9.0 * 16 * 32 * 32 * 2;
  printf("%-58s %8.3f us/trip  %7.1f TFLOP/s f32-equivalent  clock %4.0f MHz  %6.0f cycles/trip\n", what, ms * 1e3 / trips,
         flops / ms / 1e9, hc / (ms * 1e3), (double)hc / trips);
  return 0;
}

// ------------------------------------------------------------------------------------------------ part 2: error
// one wave: C[32][32] = A[32][K] * B[K][32]; form 0: f32 MFMA; 1: six plane products; 2: three; 3: two planes, four products
__device__ __forceinline__ void split3(float x, __bf16& p0, __bf16& p1, __bf16& p2) {
  p0 = (__bf16)x;
  const float r1 = x - (float)p0;
  p1 = (__bf16)r1;
  const float r2 = r1 - (float)p1;
  p2 = (__bf16)r2;
}

template <int FORM>
__global__ __launch_bounds__(64) void gemm32(const float* __restrict__ A, const float* __restrict__ B, int K, float* __restrict__ C) {
  const int lane = threadIdx.x, l31 = lane & 31, half = lane >> 5;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (FORM == 0) {
    for (int k0 = 0; k0 < K; k0 += 2) {
      const float a = A[l31 * K + k0 + half];
      const float b = B[(k0 + half) * 32 + l31];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
  } else {
    for (int k0 = 0; k0 < K; k0 += 16) {
      bf16x8 a[3], b[3];
      for (int i = 0; i < 8; ++i) {
        const int k = k0 + half * 8 + i;
        __bf16 p0, p1, p2;
        split3(A[l31 * K + k], p0, p1, p2);
        a[0][i] = p0; a[1][i] = p1; a[2][i] = p2;
        split3(B[k * 32 + l31], p0, p1, p2);
        b[0][i] = p0; b[1][i] = p1; b[2][i] = p2;
      }
      // smallest products first, so that they are not absorbed one by one by a large running sum
      if (FORM == 1) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
      }
      if (FORM == 3) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
    }
  }
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + l31] = acc[r];
}

int run_error() {
  const int K = 1024;
  std::mt19937 rng(7);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> A(32 * K), B(K * 32);
  // A: transformed weights (zero mean); B: transformed activations of a ReLU network (mostly non-negative, wide dynamic range)
  for (auto& v : A) v = nd(rng) * 0.05f;
  for (auto& v : B) v = std::fabs(nd(rng)) * std::exp(nd(rng));
  std::vector<double> ref(32 * 32, 0.0);
  for (int i = 0; i < 32; ++i)
    for (int k = 0; k < K; ++k)
      for (int j = 0; j < 32; ++j) ref[i * 32 + j] += (double)A[i * K + k] * (double)B[k * 32 + j];
  float *dA, *dB, *dC;
  CK(hipMalloc(&dA, A.size() * 4));
  CK(hipMalloc(&dB, B.size() * 4));
  CK(hipMalloc(&dC, 32 * 32 * 4));
  CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  const char* names[4] = {"f32 MFMA (v_mfma_f32_32x32x2_f32)", "bf16 x 3 planes, 6 products", "bf16 x 3 planes, 3 products (00 01 10)",
                          "bf16 x 2 planes, 4 products"};
  double rms_ref = 0;
  for (double v : ref) rms_ref += v * v;
  rms_ref = std::sqrt(rms_ref / ref.size());
  printf("\nerror of one [32 x %d] x [%d x 32] product against f64 (output RMS %.4g):\n", K, K, rms_ref);
  for (int form = 0; form < 4; ++form) {
    if (form == 0) hipLaunchKernelGGL(gemm32<0>, dim3(1), dim3(64), 0, 0, dA, dB, K, dC);
    if (form == 1) hipLaunchKernelGGL(gemm32<1>, dim3(1), dim3(64), 0, 0, dA, dB, K, dC);
    if (form == 2) hipLaunchKernelGGL(gemm32<2>, dim3(1), dim3(64), 0, 0, dA, dB, K, dC);
    if (form == 3) hipLaunchKernelGGL(gemm32<3>, dim3(1), dim3(64), 0, 0, dA, dB, K, dC);
    CK(hipDeviceSynchronize());
    std::vector<float> C(32 * 32);
    CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
    double se = 0, mx = 0;
    for (int i = 0; i < 32 * 32; ++i) {
      const double e = (double)C[i] - ref[i];
      se += e * e;
      mx = std::max(mx, std::fabs(e));
    }
    printf("  %-44s relative RMS error %.3e   max |error| / output RMS %.3e\n", names[form], std::sqrt(se / (32 * 32)) / rms_ref, mx / rms_ref);
  }
  return 0;
}

int main(int argc, char** argv) {
  const bool only_rate = argc > 1 && !strcmp(argv[1], "rate");
  float* d;
  long long* dc;
  CK(hipMalloc(&d, 256 * 512 * sizeof(float)));
  CK(hipMalloc(&dc, 8));
  printf("k-step matrix work of 16 input channels per trip, 8 waves per CU, operands from LDS:\n");
  if (run_rate<0, 0>(d, dc, "f32 MFMA x 72, 2 ds_read_b32 each")) return 1;
  if (run_rate<0, 1>(d, dc, "f32 MFMA x 72 + 1 VALU op per MFMA")) return 1;
  if (run_rate<1, 0>(d, dc, "bf16 x 3 planes, 6 products: MFMA x 54, 6 ds_read_b128 per tile")) return 1;
  if (run_rate<1, 4>(d, dc, "bf16 x 3 planes, 6 products + 4 VALU ops per MFMA")) return 1;
  if (run_rate<1, 8>(d, dc, "bf16 x 3 planes, 6 products + 8 VALU ops per MFMA")) return 1;
  if (run_rate<2, 0>(d, dc, "bf16 x 2 planes, 3 products: MFMA x 27")) return 1;
  if (only_rate) return 0;
  return run_error();
}
