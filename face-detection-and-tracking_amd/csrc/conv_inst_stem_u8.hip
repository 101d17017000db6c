// 7x7 stem convolutions on raw uint8 frames (conv_stem_u8.h) and the stride-4 stem of FaceBoxes in both input forms (conv_stem_s4.h).
#include "conv_stem_u8.h"
#include "conv_stem_s4.h"

namespace fdt {
void conv_fill_stem_u8(void* r2) {
  KernelEntry* a = (KernelEntry*)r2;
  a[TILE_128x64W] = entry_stem<STEM_S2_N64>();
}
void conv_fill_stem_s4(void* rf, void* ru) {
  ((KernelEntry*)rf)[TILE_128x32W] = entry_stem_s4<false>();
  ((KernelEntry*)ru)[TILE_128x32W] = entry_stem_s4<true>();
}
}  // namespace fdt
