"""FaceBoxes detector with the reference's module interface (reference FACEBOX/networks.py:60-116):
`FaceBox()` -> object with `load_state_dict`, `.cuda()`, `.eval()`, `net(x) -> (loc_preds, conf_preds)`
(raw, un-softmaxed conf, like the reference).  RDCL (7x7/s4 + CReLU, 5x5/s2 + CReLU), three Inception
blocks, conv3_x/conv4_x and the 21/1/1-anchor multibox heads all run as HIP kernels."""
import ctypes as C

import numpy as np
import torch

from .. import _lib
from .._net import DetectorNet


class FaceBox(DetectorNet):
    input_size = 1024
    _arch = _lib.ARCH_FACEBOX
    _n_sources = 3
    _default_priorbox = staticmethod(lambda size: None)
    _default_detect = staticmethod(lambda nc: None)

    def __init__(self, device=0):
        super().__init__('test', 2, 1024, device)

    def _sync_attributes(self, H, W):
        if (H, W) != (1024, 1024):
            raise ValueError("FaceBox anchors are defined for 1024x1024 inputs (reference "
                             "FACEBOX/encoderl.py:21-27, My_test_facebox.py:13), got %dx%d" % (H, W))

    def __call__(self, x):
        """x: [B,3,1024,1024] f32 (BGR/255) or uint8 [B,1024,1024,3] BGR.  Returns (loc [B,21824,4],
        conf [B,21824,2]) as torch tensors, conf NOT softmaxed (reference networks.py:114-116)."""
        if not self._loaded:
            raise RuntimeError("weights not loaded: call load_state_dict first")
        x, fmt, B, H, W = self._prepare(x)
        self._sync_attributes(H, W)
        loc = np.empty((B, 21824, 4), np.float32)
        conf = np.empty((B, 21824, 2), np.float32)
        _lib.check(_lib.lib().fdt_model_forward_raw(self._h, _lib.ptr(x), fmt, B, H, W, _lib.ptr(loc),
                                                    _lib.ptr(conf)))
        return torch.from_numpy(loc), torch.from_numpy(conf)

    forward = __call__

    def detect_frames(self, frames, conf_thresh=0.35, nms_thresh=0.5):
        """Whole `detect(im)` of reference FACEBOX/My_test_facebox.py:12-36 after the resize, on the GPU.
        Returns a list of (boxes [k,4] in [0,1], probs [k]) per image."""
        x, fmt, B, H, W = self._prepare(frames)
        boxes = np.empty((B, 21824, 4), np.float32)
        probs = np.empty((B, 21824), np.float32)
        counts = np.zeros(B, np.int32)
        if fmt == _lib.FRAME_U8_HWC_BGR and (H, W) != (1024, 1024):
            # line :13 of the reference's detect(): im = cv2.resize(im, (1024, 1024)) -- here on the GPU
            _lib.check(_lib.lib().fdt_model_detect_facebox_resized(self._h, _lib.ptr(x), 0, B, H, W, float(conf_thresh),
                                                                   float(nms_thresh), _lib.ptr(boxes), _lib.ptr(probs),
                                                                   _lib.ptr(counts), None))
            return [(boxes[b, :counts[b]].copy(), probs[b, :counts[b]].copy()) for b in range(B)]
        self._sync_attributes(H, W)
        _lib.check(_lib.lib().fdt_model_detect_facebox(self._h, _lib.ptr(x), fmt, B, H, W, float(conf_thresh),
                                                       float(nms_thresh), _lib.ptr(boxes), _lib.ptr(probs),
                                                       _lib.ptr(counts)))
        return [(boxes[b, :counts[b]].copy(), probs[b, :counts[b]].copy()) for b in range(B)]
