// Persistent-tile 1x1 / stride-1 kernels (conv_1x1p.h).
#include "conv_1x1p.h"

namespace fdt {
void conv_fill_1x1_p(void* r16, void* r32) {
  KernelEntry* a = (KernelEntry*)r16;
  KernelEntry* b = (KernelEntry*)r32;
  a[TILE_P_128x64] = entry_p<P_K16_N64>();
  a[TILE_P_128x128] = entry_p<P_K16_N128>();
  b[TILE_P_128x64] = entry_p<P_K32_N64>();
}
int conv_1x1p_resident(ConvTile t) { return t == TILE_P_128x128 ? P_K16_N128::RESIDENT : P_K16_N64::RESIDENT; }
}  // namespace fdt
