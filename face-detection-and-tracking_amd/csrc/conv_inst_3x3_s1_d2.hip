// Instantiations of conv_kernel<> for one convolution class (compiled in parallel with the others).
#include "conv_kernel.h"

namespace fdt {
void conv_fill_3x3_s1_d2(void* row) { fill_row<G_3x3_S1_D2>((KernelEntry*)row); }
}  // namespace fdt
