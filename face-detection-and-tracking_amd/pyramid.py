"""PyramidBox-Res50 detector with the reference's module interface (reference pyramid.py:106-374):
`build_sfd(phase, size, num_classes)` -> object with `load_state_dict`, `.priorbox`, `.firstTime`,
`.detect`, `.cuda()`, `.eval()`, `net(x) -> Tensor[B,2,750,5]`.  The forward pass
(pyramid.py:218-351) runs as HIP kernels on the MI355X behind `fdt_model_forward`."""
from . import _lib
from ._net import DetectorNet
from .layers import Detect, PriorBoxLayer


class SFD(DetectorNet):
    _arch = _lib.ARCH_RES50
    _n_sources = 6
    # pyramid.py:113: PriorBoxLayer(size, size, stride=[4, 8, 16, 32, 64, 128])
    _default_priorbox = staticmethod(lambda size: PriorBoxLayer(size, size, stride=[4, 8, 16, 32, 64, 128]))
    # pyramid.py:198: Detect(num_classes, 0, 750, 0.3, 0.5)
    _default_detect = staticmethod(lambda nc: Detect(nc, 0, 750, 0.3, 0.5))

    def __init__(self, block=None, num_blocks=(3, 4, 6, 3), phase='test', num_classes=2, size=640, device=0):
        super().__init__(phase, num_classes, size, device)


def build_sfd(phase, size=640, num_classes=2):
    if phase != "test" and phase != "train":
        print("Error: Phase not recognized")
        return
    if size != 640:
        print("Error: Sorry only 640 is supported currently!")
        return
    return SFD(None, [3, 4, 6, 3], phase, num_classes, size)
