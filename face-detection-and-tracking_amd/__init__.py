"""MI355X-native (gfx950) per-frame face-detection-and-tracking hot path.

Host-side mirror of the reference's Python interface for the path (same module names,
argument meaning and error behaviour: `pyramid.build_sfd`, `layers.Detect`,
`layers.PriorBoxLayer`, `layers.box_utils.decode/nms`, `utils.calc_performance.calculate_iou`,
`iouTracke_cal`) over hand-written HIP kernels reached through the C ABI of
`csrc/libfdt_hip.so` (include/fdt.h).  PyTorch appears only as a container for weights and
returned tensors.  There is no CPU fallback: without the built library every op raises.
"""
from . import _lib  # noqa: F401
from ._lib import FdtError, device_count, device_name  # noqa: F401
