"""`calculate_iou` / `calc_pr` with the reference's signatures (reference
utils/calc_performance.py:54-92).  The pairwise IoU matrix is computed on the GPU behind
`fdt_pairwise_iou` in the input's precision (f64 in the tracker, f32 if f32 comes in), like
numpy does for the reference."""
import numpy as np

from .. import _lib


def calculate_iou(box_a, box_b):
    a = np.asarray(box_a)
    b = np.asarray(box_b)
    dt = np.result_type(a.dtype, b.dtype)
    if dt != np.float32:
        dt = np.float64
    a = np.ascontiguousarray(a, dtype=dt).reshape(-1, 4)
    b = np.ascontiguousarray(b, dtype=dt).reshape(-1, 4)
    out = np.empty((a.shape[0], b.shape[0]), dtype=dt)
    _lib.check(_lib.lib().fdt_pairwise_iou(_lib.ptr(a), a.shape[0], _lib.ptr(b), b.shape[0],
                                           _lib.F64 if dt == np.float64 else _lib.F32,
                                           _lib.ptr(out)))
    return out


def calc_pr(predict, truth, iou_thresh=0.5):
    truth = np.hstack((truth[:, :2], truth[:, 2:] + truth[:, :2]))
    iou = calculate_iou(truth, predict[:, :4])
    truth_num, _ = iou.shape
    tf = (np.max(iou, 0) > iou_thresh).astype(np.int32)
    return np.vstack((tf, predict[:, 4])), truth_num
