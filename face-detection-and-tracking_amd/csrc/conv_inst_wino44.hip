// Instantiation of the Winograd F(4x4,3x3) kernel.
#include "conv_wino44.h"

namespace fdt {
void conv_fill_wino44(void* row) {
  KernelEntry* r = (KernelEntry*)row;
  r[TILE_WINO44_32x64] = wino44_entry();
  r[TILE_WINO44B_32x64] = wino44b_entry();
}
void conv_fill_wino44_d2(void* row) {
  KernelEntry* r = (KernelEntry*)row;
  r[TILE_WINO44_32x64] = wino44d2_entry();
}
}  // namespace fdt

#ifdef FDT_W44_STAMPS
extern "C" int fdt_debug_w44_times(long long* out) {
  FDT_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(fdt::g_w44_time), 64));
  long long z[8] = {0};
  FDT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(fdt::g_w44_time), z, 64));
  return FDT_OK;
}
#endif
