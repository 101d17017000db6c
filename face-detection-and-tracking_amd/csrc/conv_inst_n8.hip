// Instantiation of the packed-f32 VALU kernel for narrow 3x3 heads (conv_n8.h).
#include "conv_n8.h"

namespace fdt {
void conv_fill_n8(void* row) { ((KernelEntry*)row)[TILE_N8_32x64] = n8_entry(); }
}  // namespace fdt
