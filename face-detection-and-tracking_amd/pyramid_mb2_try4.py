"""PyramidBox on the MobileNetV2 "try4" backbone with the reference's module interface (reference
pyramid_mb2_try4.py:137-370).  try5's graph with two more changes: the stem `conv_bn` uses a 7x7 kernel
while its padding stays 1 (:16, so a 1024-wide input gives a 510-wide stem), and smooth_c5 too is a
`Conv2d(320, 320, kernel_size=1, padding=1)` (:190) -- the c5 source and the map that feeds the top-down
path are (h+2)x(w+2); ContextTexture crops the upsampled map back (pyramid_mb2_try4.py:61-69)."""
from . import _lib
from ._net import DetectorNet
from .layers import Detect, PriorBoxLayer


def _half(n):
    return (n - 1) // 2 + 1


class SFD_mobile(DetectorNet):
    _arch = _lib.ARCH_TRY4
    _n_sources = 5
    # pyramid_mb2_try4.py:144
    _default_priorbox = staticmethod(lambda size: PriorBoxLayer(size, size, stride=[4, 8, 16, 32, 64],
                                                                box=(16, 32, 64, 128, 256)))
    # pyramid_mb2_try4.py:219
    _default_detect = staticmethod(lambda nc: Detect(nc, 0, 750, 0.2, 0.35))

    def __init__(self, phase='test', num_classes=2, size=640, device=0):
        super().__init__(phase, num_classes, size, device)

    @staticmethod
    def source_sizes(H, W):
        """(h, w) of the five detection sources for an HxW input."""
        h, w = _half((H - 5) // 2 + 1), _half((W - 5) // 2 + 1)     # 7x7/s2/p1 stem, then features.2 (s2)
        out = []
        for lvl in range(5):
            out.append((h + 2, w + 2) if lvl >= 3 else (h, w))      # smooth_c5 / smooth_c6: kernel 1, padding 1
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
