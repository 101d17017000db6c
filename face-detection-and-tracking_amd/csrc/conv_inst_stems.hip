// Instantiations of conv_kernel<> for one convolution class (compiled in parallel with the others).
#include "conv_kernel.h"

namespace fdt {
// stems / FaceBox RDCL: only the tiles their Cout needs
void conv_fill_stems(void* r72, void* r74, void* r52, void* r72p1) {
  KernelEntry* a = (KernelEntry*)r72;
  KernelEntry* b = (KernelEntry*)r74;
  KernelEntry* c = (KernelEntry*)r52;
  a[TILE_128x64] = entry<G_7x7_S2, T_128x64>();
  a[TILE_64x64] = entry<G_7x7_S2, T_64x64>();
  a[TILE_128x64W] = entry<G_7x7_S2, T_128x64W>();
  b[TILE_128x32] = entry<G_7x7_S4, T_128x32>();
  c[TILE_128x64] = entry<G_5x5_S2, T_128x64>();
  c[TILE_64x64] = entry<G_5x5_S2, T_64x64>();
  KernelEntry* d = (KernelEntry*)r72p1;
  d[TILE_128x32] = entry<G_7x7_S2_P1, T_128x32>();
  d[TILE_128x64] = entry<G_7x7_S2_P1, T_128x64>();
}
}  // namespace fdt
