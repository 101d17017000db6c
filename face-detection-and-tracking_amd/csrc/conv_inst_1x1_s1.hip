// Instantiations of conv_kernel<> for one convolution class (compiled in parallel with the others).
#include "conv_kernel.h"

namespace fdt {
void conv_fill_1x1_s1(void* row) {
  KernelEntry* r = (KernelEntry*)row;
  fill_row<G_1x1_S1>(r);
  r[TILE_128x128R4] = entry<G_1x1_S1, T_128x128R4>();
  r[TILE_128x64R4] = entry<G_1x1_S1, T_128x64R4>();
  r[TILE_64x64R4] = entry<G_1x1_S1, T_64x64R4>();
  r[TILE_64x128R4] = entry<G_1x1_S1, T_64x128R4>();
}
}  // namespace fdt
