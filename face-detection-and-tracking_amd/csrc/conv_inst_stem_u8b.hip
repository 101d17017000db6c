// Raw-uint8 7x7 stems on the bf16 matrix pipe (conv_stem_u8b.h): one exact bf16 plane of pixels x three planes of weights.
#include "conv_stem_u8b.h"

namespace fdt {
void conv_fill_stem_u8b(void* row_s2, void* row_s4) {
  KernelEntry* r2 = (KernelEntry*)row_s2;
  KernelEntry* r4 = (KernelEntry*)row_s4;
  r2[TILE_128x64W] = entry_stem_u8b<StemU8B_S2>();
  r4[TILE_128x32W] = entry_stem_u8b<StemU8B_S4>();
}
}  // namespace fdt
