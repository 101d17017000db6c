"""`Detect` with the reference's constructor and call signature (reference
layers/functions/detection.py:9-84): decode -> threshold -> per-image greedy NMS -> top-k pack,
all on the GPU behind `fdt_detect`."""
import numpy as np
import torch

from ... import _lib
from ...data import face as cfg


def _np(x, dtype=np.float32):
    if isinstance(x, torch.Tensor):
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(x, dtype=dtype)


class Detect:
    def __init__(self, num_classes, bkg_label, top_k, conf_thresh, nms_thresh):
        self.num_classes = num_classes
        self.background_label = bkg_label
        self.top_k = top_k
        self.nms_thresh = nms_thresh
        if nms_thresh <= 0:
            raise ValueError('nms_threshold must be non negative.')
        self.conf_thresh = conf_thresh
        self.variance = cfg['variance']
        self.nms_top_k = 5000

    def __call__(self, loc_data, conf_data, prior_data):
        loc = _np(loc_data)
        pri = _np(prior_data)
        num = loc.shape[0]
        P = pri.shape[0]
        conf = _np(conf_data).reshape(num, P, self.num_classes)
        loc = loc.reshape(num, P, 4)
        out = np.empty((num, self.num_classes, self.top_k, 5), dtype=np.float32)
        counts = np.zeros((num, self.num_classes), dtype=np.int32)
        _lib.check(_lib.lib().fdt_detect(_lib.ptr(loc), _lib.ptr(conf), _lib.ptr(pri), num, P,
                                         self.num_classes, self.top_k, float(self.conf_thresh),
                                         float(self.nms_thresh), self.nms_top_k,
                                         float(self.variance[0]), float(self.variance[1]),
                                         _lib.ptr(out), _lib.ptr(counts)))
        self.last_counts = counts
        return torch.from_numpy(out)
