// 1x1 stride-1 kernels with 32 / 64 input channels per LDS stage.
#include "conv_kernel.h"

namespace fdt {
void conv_fill_1x1_s1_deep(void* r32, void* r64) {
  KernelEntry* a = (KernelEntry*)r32;
  KernelEntry* b = (KernelEntry*)r64;
  a[TILE_128x128] = entry<G_1x1_S1_K32, T_128x128>();
  a[TILE_128x64] = entry<G_1x1_S1_K32, T_128x64>();
  a[TILE_64x64] = entry<G_1x1_S1_K32, T_64x64>();
  a[TILE_64x128] = entry<G_1x1_S1_K32, T_64x128>();
  a[TILE_128x64W] = entry<G_1x1_S1_K32, T_128x64W>();
  a[TILE_64x64R3] = entry<G_1x1_S1_K32, T_64x64R3>();
  a[TILE_128x64R3] = entry<G_1x1_S1_K32, T_128x64R3>();
  a[TILE_128x128R3] = entry<G_1x1_S1_K32, T_128x128R3>();
  a[TILE_128x128R4] = entry<G_1x1_S1_K32, T_128x128R4>();
  a[TILE_128x64R4] = entry<G_1x1_S1_K32, T_128x64R4>();
  a[TILE_64x64R4] = entry<G_1x1_S1_K32, T_64x64R4>();
  a[TILE_64x128R4] = entry<G_1x1_S1_K32, T_64x128R4>();
  b[TILE_128x64] = entry<G_1x1_S1_K64, T_128x64>();
  b[TILE_64x64] = entry<G_1x1_S1_K64, T_64x64>();
  b[TILE_64x128] = entry<G_1x1_S1_K64, T_64x128>();
  b[TILE_64x64R3] = entry<G_1x1_S1_K64, T_64x64R3>();
}
}  // namespace fdt
