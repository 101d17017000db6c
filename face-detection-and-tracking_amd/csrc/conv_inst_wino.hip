// Instantiations of the Winograd F(2x2,3x3) kernel.
#include "conv_wino.h"

namespace fdt {
void conv_fill_wino(void* row) {
  KernelEntry* r = (KernelEntry*)row;
  r[TILE_WINO_64x64] = wino_entry<W_64x64>();
  r[TILE_WINO_64x64R3] = wino_entry<W_64x64R3>();
  r[TILE_WINO_128x32] = wino_entry<W_128x32>();
  r[TILE_WINO_128x32R3] = wino_entry<W_128x32R3>();
  r[TILE_WINO_32x128] = wino_entry<W_32x128>();
  // TILE_WINO_32x128R3 would need 215 KB of LDS (> 160 KB per CU): not instantiated
  r[TILE_WINO_64x64W] = wino_entry<W_64x64W>();
  r[TILE_WINO8_64x64] = wino2_entry<W_64x64>();
  r[TILE_WINO8_64x64R3] = wino2_entry<W_64x64R3>();
  r[TILE_WINO8_128x32R3] = wino2_entry<W_128x32R3>();
  r[TILE_WINO8_64x64W] = wino2_entry<W_64x64W>();
  r[TILE_WINO4_64x64R3] = wino4_entry<W_64x64R3>();
  r[TILE_WINO4_64x64W] = wino4_entry<W_64x64W>();
}
void conv_fill_wino_d2(void* row) {
  KernelEntry* r = (KernelEntry*)row;
  r[TILE_WINO8_64x64] = wino2_entry<WD2_64x64>();
  r[TILE_WINO8_64x64R3] = wino2_entry<WD2_64x64R3>();
  r[TILE_WINO8_128x32R3] = wino2_entry<WD2_128x32R3>();
  r[TILE_WINO8_64x64W] = wino2_entry<WD2_64x64W>();
  r[TILE_WINO4_64x64R3] = wino4_entry<WD2_64x64R3>();
  r[TILE_WINO4_64x64W] = wino4_entry<WD2_64x64W>();
}
}  // namespace fdt
