"""`DataEncoder` with the inference half of the reference's interface (reference
FACEBOX/encoderl.py:10-48 anchors, :217-266 nms_np, :308-325 decode_np).  Anchors, decode and the
greedy NMS run on the GPU (`fdt_facebox_anchors`, `fdt_facebox_decode`, `fdt_nms`)."""
import ctypes as C

import numpy as np
import torch

from .. import _lib


def _np(x, dtype=np.float32):
    if isinstance(x, torch.Tensor):
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(x, dtype=dtype)


class DataEncoder:
    def __init__(self):
        a = np.empty((21824, 4), np.float32)
        _lib.check(_lib.lib().fdt_facebox_anchors(_lib.ptr(a)))
        self.default_boxes = torch.from_numpy(a)
        self.default_boxes_np = a

    def decode_np(self, loc, conf, conf_thres=0.35):
        """loc [21824,4], conf [21824,2] (softmaxed) -> (boxes[keep] in [0,1], probs[keep])."""
        loc, conf = _np(loc), _np(conf)
        P = loc.shape[0]
        boxes = np.empty((P, 4), np.float32)
        probs = np.empty(P, np.float32)
        n = C.c_int(0)
        _lib.check(_lib.lib().fdt_facebox_decode(_lib.ptr(loc), _lib.ptr(conf), _lib.ptr(self.default_boxes_np),
                                                 P, float(conf_thres), 0.5, _lib.ptr(boxes), _lib.ptr(probs),
                                                 C.byref(n)))
        return boxes[:n.value].copy(), probs[:n.value].copy()
