"""PyramidBox on the Mobilenetv1/Mobilenetv2-block backbone "try2" with the reference's module interface
(reference pyramid_mobile_try2.py:141-398).  Res50-sized pyramid (6 sources, 512-channel SSH context modules,
max-in-out heads) on a backbone of inverted-residual blocks with 3x3 / 5x5 / 7x7 and dilated depthwise
convolutions and grouped 1x1 lateral layers; the forward pass runs as HIP kernels behind `fdt_model_forward`
(depthwise stages on an HBM-bound kernel, every dense / grouped conv on the matrix cores)."""
from . import _lib
from ._net import DetectorNet
from .layers import Detect, PriorBoxLayer


class SFD_mobile(DetectorNet):
    _arch = _lib.ARCH_TRY2
    _n_sources = 6
    # pyramid_mobile_try2.py: PriorBoxLayer(size, size, stride=[4, 8, 16, 32, 64, 128])
    _default_priorbox = staticmethod(lambda size: PriorBoxLayer(size, size, stride=[4, 8, 16, 32, 64, 128]))
    # pyramid_mobile_try2.py: Detect(num_classes, 0, 750, 0.3, 0.5)
    _default_detect = staticmethod(lambda nc: Detect(nc, 0, 750, 0.3, 0.5))

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
