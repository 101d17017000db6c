// 7x7 stem convolutions on raw uint8 frames (conv_stem_u8.h).
#include "conv_stem_u8.h"

namespace fdt {
void conv_fill_stem_u8(void* r2, void* r4) {
  KernelEntry* a = (KernelEntry*)r2;
  KernelEntry* b = (KernelEntry*)r4;
  a[TILE_128x64W] = entry_stem<STEM_S2_N64>();
  b[TILE_128x32W] = entry_stem<STEM_S4_N32>();
}
}  // namespace fdt
