// Split-bf16 1x1 / stride-1 kernels (conv_b3.h): three bf16 planes per operand on v_mfma_f32_32x32x16_bf16.
#include "conv_b3.h"
#include "conv_stem_b3.h"

namespace fdt {
void conv_fill_1x1_b3(void* row) {
  KernelEntry* r = (KernelEntry*)row;
  r[TILE_128x128R3] = entry_b3<TB3_128x128>();
  r[TILE_128x64R3] = entry_b3<TB3_128x64>();
  r[TILE_128x128WR3] = entry_b3<TB3_128x128W>();
  r[TILE_128x64WR3] = entry_b3<TB3_128x64W>();
  r[TILE_128x128W] = entry_b3<TB3_128x128W4>();
  r[TILE_128x64W] = entry_b3<TB3_128x64W4>();
  r[TILE_R2_128x128] = entry_b3<TB3_R2_128>();
  r[TILE_R2_128x64] = entry_b3<TB3_R2_64>();
  r[TILE_R1_128x128] = entry_b3<TB3_R1_128>();
  r[TILE_R1_128x64] = entry_b3<TB3_R1_64>();
}
void conv_fill_1x1_s2_b3(void* row) {
  KernelEntry* r = (KernelEntry*)row;
  r[TILE_128x128W] = entry_b3<TB3_128x128W4, 2>();
  r[TILE_128x64W] = entry_b3<TB3_128x64W4, 2>();
}
void conv_fill_3x3_s2_b3(void* row) {
  KernelEntry* r = (KernelEntry*)row;
  r[TILE_128x128W] = entry_b3<TB3_128x128W4, 2, 3>();
  r[TILE_128x64W] = entry_b3<TB3_128x64W4, 2, 3>();
}
void conv_fill_stem_b3(void* row) {
  KernelEntry* r = (KernelEntry*)row;
  r[TILE_128x32W] = entry_stem_s4_b3();
}
}  // namespace fdt
