// Which bits of a hipExtStreamCreateWithCUMask mask select which XCD / CU on this chip, and what does a kernel that needs
// half the chip cost on a half-chip stream?   hipcc --offload-arch=gfx950 -O2 tools/microbench/cu_mask_probe.hip -o /tmp/cu_mask_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>
#define CK(x)                                                               \
  do {                                                                      \
    hipError_t e_ = (x);                                                    \
    if (e_ != hipSuccess) {                                                 \
      printf("%s failed: %s\n", #x, hipGetErrorString(e_));                 \
      exit(1);                                                              \
    }                                                                       \
  } while (0)

__global__ void where_kernel(unsigned* out, long long spin) {
  unsigned xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  const long long t0 = clock64();
  while (clock64() - t0 < spin) {}
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = xcc;
    out[2 * blockIdx.x + 1] = hwid;
  }
}

static void probe(const char* what, const std::vector<uint32_t>& mask, unsigned* d_out, int grid) {
  hipStream_t st;
  CK(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
  CK(hipMemsetAsync(d_out, 0xff, grid * 8, st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(where_kernel, dim3(grid), dim3(256), 0, st, d_out, 2000ll);
  CK(hipEventRecord(e0, st));
  hipLaunchKernelGGL(where_kernel, dim3(grid), dim3(256), 0, st, d_out, 200000ll);   // ~100 us of spinning per workgroup
  CK(hipEventRecord(e1, st));
  CK(hipStreamSynchronize(st));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned> h(grid * 2);
  CK(hipMemcpy(h.data(), d_out, grid * 8, hipMemcpyDeviceToHost));
  std::map<unsigned, int> per_xcc;
  std::map<unsigned long long, int> per_cu;
  for (int i = 0; i < grid; ++i) {
    const unsigned xcc = h[2 * i] & 0xf, hw = h[2 * i + 1];
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    per_xcc[xcc]++;
    per_cu[((unsigned long long)xcc << 16) | (se << 8) | (sh << 4) | cu]++;
  }
  printf("%-34s grid %4d: %7.1f us, distinct CUs %3zu, workgroups per XCD:", what, grid, ms * 1e3, per_cu.size());
  for (auto& kv : per_xcc) printf(" %u:%d", kv.first, kv.second);
  printf("\n");
  CK(hipStreamDestroy(st));
}

int main() {
  unsigned* d_out;
  CK(hipMalloc(&d_out, 4096 * 8));
  auto mask_of = [](auto pred) {
    std::vector<uint32_t> m(8, 0);
    for (int i = 0; i < 256; ++i)
      if (pred(i)) m[i / 32] |= 1u << (i % 32);
    return m;
  };
  for (int grid : {128, 256, 512}) {
    probe("all 256 bits", mask_of([](int) { return true; }), d_out, grid);
    probe("bits 0..127", mask_of([](int i) { return i < 128; }), d_out, grid);
    probe("bits 128..255", mask_of([](int i) { return i >= 128; }), d_out, grid);
    probe("even bits", mask_of([](int i) { return (i & 1) == 0; }), d_out, grid);
    probe("bits with (i % 8) < 4", mask_of([](int i) { return (i % 8) < 4; }), d_out, grid);
    probe("bits with (i / 32) even", mask_of([](int i) { return ((i / 32) & 1) == 0; }), d_out, grid);
    probe("bits 0..31", mask_of([](int i) { return i < 32; }), d_out, grid);
    probe("bits i % 8 == 0", mask_of([](int i) { return i % 8 == 0; }), d_out, grid);
  }
  return 0;
}
