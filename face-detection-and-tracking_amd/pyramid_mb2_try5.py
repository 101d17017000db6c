"""PyramidBox on the MobileNetV2 "try5" backbone with the reference's module interface (reference
pyramid_mb2_try5.py:137-370).  Same graph as try3 except the LFPN smooth layers (:184-191): smooth_c2/3/4
are an InvertedResidual followed by the 3x3 conv, and smooth_c6 is `Conv2d(160, 160, kernel_size=1,
padding=1)` -- a 1x1 conv that GROWS the map by one zero-padded pixel per side, so the last detection
source is (h+2)x(w+2) and the prior count changes accordingly (priors follow the source sizes, :270-283)."""
from . import _lib
from ._net import DetectorNet
from .layers import Detect, PriorBoxLayer


def _half(n):
    return (n - 1) // 2 + 1


class SFD_mobile(DetectorNet):
    _arch = _lib.ARCH_TRY5
    _n_sources = 5
    # pyramid_mb2_try5.py:144
    _default_priorbox = staticmethod(lambda size: PriorBoxLayer(size, size, stride=[4, 8, 16, 32, 64],
                                                                box=(16, 32, 64, 128, 256)))
    # pyramid_mb2_try5.py:219
    _default_detect = staticmethod(lambda nc: Detect(nc, 0, 750, 0.2, 0.35))

    def __init__(self, phase='test', num_classes=2, size=640, device=0):
        super().__init__(phase, num_classes, size, device)

    @staticmethod
    def source_sizes(H, W):
        """(h, w) of the five detection sources for an HxW input."""
        h, w = _half(_half(H)), _half(_half(W))
        out = []
        for lvl in range(5):
            out.append((h + 2, w + 2) if lvl == 4 else (h, w))      # smooth_c6: kernel 1, padding 1
            h, w = _half(h), _half(w)
        return out

    def _num_priors_for(self, x, fmt, B, H, W):
        return sum(h * w for h, w in self.source_sizes(H, W))


def build_sfd_mobile(phase, size=640, num_classes=2):
    if phase != "test" and phase != "train":
        print("Error: Phase not recognized")
        return
    if size != 640:
        print("Error: Sorry only 640 is supported currently!")
        return
    return SFD_mobile(phase, num_classes, size)
