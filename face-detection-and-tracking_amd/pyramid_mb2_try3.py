"""PyramidBox on the MobileNetV2 "try3" backbone with the reference's module interface (reference
pyramid_mb2_try3.py:137-366).  The forward pass (:218-340) runs as HIP kernels behind
`fdt_model_forward`; depthwise 3x3 stages use a dedicated HBM-bound kernel, not the matrix cores."""
from . import _lib
from ._net import DetectorNet
from .layers import Detect, PriorBoxLayer


class SFD_mobile(DetectorNet):
    _arch = _lib.ARCH_TRY3
    _n_sources = 5
    # pyramid_mb2_try3.py:144
    _default_priorbox = staticmethod(lambda size: PriorBoxLayer(size, size, stride=[4, 8, 16, 32, 64],
                                                                box=(16, 32, 64, 128, 256)))
    # pyramid_mb2_try3.py:216
    _default_detect = staticmethod(lambda nc: Detect(nc, 0, 750, 0.2, 0.35))

    def __init__(self, phase='test', num_classes=2, size=640, device=0):
        super().__init__(phase, num_classes, size, device)


def build_sfd_mobile(phase, size=640, num_classes=2):
    if phase != "test" and phase != "train":
        print("Error: Phase not recognized")
        return
    if size != 640:
        print("Error: Sorry only 640 is supported currently!")
        return
    return SFD_mobile(phase, num_classes, size)
