// Instantiations of conv_kernel<> for one convolution class (compiled in parallel with the others).
#include "conv_kernel.h"

namespace fdt {
void conv_fill_1x1_s1(void* row) { fill_row<G_1x1_S1>((KernelEntry*)row); }
}  // namespace fdt
