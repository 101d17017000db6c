// Persistent-tile split-bf16 1x1 / stride-1 kernels (conv_1x1p_b3.h).
#include "conv_1x1p_b3.h"

namespace fdt {
void conv_fill_1x1_pb3(void* row) {
  KernelEntry* r = (KernelEntry*)row;
  r[TILE_P_128x64] = entry_pb3<PB3_N64>();
  r[TILE_P_128x128] = entry_pb3<PB3_N128>();
}
int conv_1x1pb3_resident(ConvTile t) { return t == TILE_P_128x128 ? PB3_N128::RESIDENT : PB3_N64::RESIDENT; }
}  // namespace fdt
