"""`decode` and `nms` with the reference's signatures (reference layers/box_utils.py:238-258,
275-340), executed by the HIP kernels behind `fdt_decode` / `fdt_nms`."""
import ctypes as C

import numpy as np
import torch

from .. import _lib


def _np(x, dtype=np.float32):
    if isinstance(x, torch.Tensor):
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(x, dtype=dtype)


def decode(loc, priors, variances):
    loc, priors = _np(loc), _np(priors)
    out = np.empty_like(loc)
    _lib.check(_lib.lib().fdt_decode(_lib.ptr(loc), _lib.ptr(priors), loc.shape[0],
                                     float(variances[0]), float(variances[1]), _lib.ptr(out)))
    return torch.from_numpy(out)


def nms(boxes, scores, overlap=0.5, top_k=200):
    """Returns (keep LongTensor[n] zero padded, count) like the reference."""
    boxes, scores = _np(boxes).reshape(-1, 4), _np(scores).reshape(-1)
    n = scores.shape[0]
    keep = np.zeros(n, dtype=np.int64)
    count = C.c_int(0)
    if boxes.size == 0:
        return torch.from_numpy(keep), 0
    _lib.check(_lib.lib().fdt_nms(_lib.ptr(boxes), _lib.ptr(scores), n, float(overlap), int(top_k),
                                  _lib.ptr(keep), C.byref(count)))
    return torch.from_numpy(keep), count.value
