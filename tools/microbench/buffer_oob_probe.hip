// Does the bounds check of a raw buffer load (stride 0, offen) on gfx950 see the SCALAR offset?  The conv kernels stage their
// operands with `buffer_load_dword(x4) ... offen lds`: per-lane byte offset + a scalar offset advanced per stage, and rely on
// out-of-range lanes receiving zeros (padding; the channels past Cin).  Prints what a lane gets for
//   (voffset, soffset) = (in, in), (out, 0), (0, out), (in, in but the sum out).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/buffer_oob_probe.hip -o tools/microbench/buffer_oob_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void probe(const float* in, int nbytes, float* out) {
  __shared__ __attribute__((aligned(16))) float sm[64 * 8];
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, nbytes, 0x00020000);
  const unsigned lane = threadIdx.x;
  for (int i = 0; i < 8; ++i) sm[i * 64 + lane] = -1.0f;
  __syncthreads();
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)(sm + 0 * 64), 4, lane * 4, 256, 0, 0);              // in range
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)(sm + 1 * 64), 4, 0x80000000u + lane * 4, 0, 0, 0);  // voffset out
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)(sm + 2 * 64), 4, lane * 4, nbytes, 0, 0);           // soffset out
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)(sm + 3 * 64), 4, lane * 4, nbytes - 128, 0, 0);     // sum straddles the end
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)(sm + 4 * 64), 4, 0x80000000u + lane * 4, nbytes - 128, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = 0; i < 5; ++i) out[i * 64 + lane] = sm[i * 64 + lane];
}
int main() {
  const int n = 4096;                      // floats in the buffer proper; the allocation is twice that (nothing faults)
  float *d, *o, h[2 * n], ho[5 * 64];
  for (int i = 0; i < 2 * n; ++i) h[i] = 1000.0f + i;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, n * 4, o);
  hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
  const char* what[5] = {"voffset in, soffset in          ", "voffset OUT, soffset 0          ", "voffset in, soffset = size      ",
                         "sum straddles the end (lane 32+)", "voffset OUT + soffset           "};
  for (int i = 0; i < 5; ++i)
    printf("%s: lane 0 -> %8.1f  lane 31 -> %8.1f  lane 32 -> %8.1f  lane 63 -> %8.1f\n", what[i], ho[i * 64], ho[i * 64 + 31],
           ho[i * 64 + 32], ho[i * 64 + 63]);
  printf("(buffer = floats 1000 .. %d; values >= %d lie past num_records; 0.0 = zeroed by the bounds check)\n", 1000 + n - 1, 1000 + n);
  return 0;
}
