"""`calculate_iou` / `calculate_distance` / `calc_pr` with the reference's signatures (reference
utils/calc_performance.py:34-92).  The pairwise matrices are computed on the GPU behind `fdt_pairwise_iou` /
`fdt_pairwise_distance` in the input's precision (f64 in the tracker, f32 if f32 comes in), like
numpy does for the reference."""
import numpy as np

from .. import _lib


def _pairwise(fn_name, box_a, box_b):
    a = np.asarray(box_a)
    b = np.asarray(box_b)
    dt = np.result_type(a.dtype, b.dtype)
    if dt != np.float32:
        dt = np.float64
    a = np.ascontiguousarray(a, dtype=dt).reshape(-1, 4)
    b = np.ascontiguousarray(b, dtype=dt).reshape(-1, 4)
    out = np.empty((a.shape[0], b.shape[0]), dtype=dt)
    _lib.check(getattr(_lib.lib(), fn_name)(_lib.ptr(a), a.shape[0], _lib.ptr(b), b.shape[0],
                                            _lib.F64 if dt == np.float64 else _lib.F32, _lib.ptr(out)))
    return out


def calculate_iou(box_a, box_b):
    """[A,4] x [B,4] (x1,y1,x2,y2) -> IoU [A,B]; 0/0 -> NaN, no epsilon (reference :54-74)."""
    return _pairwise("fdt_pairwise_iou", box_a, box_b)


def calculate_distance(box_a, box_b):
    """[A,4] x [B,4] -> the tracker's `use_iou = False` measure [A,B] (reference :34-51)."""
    return _pairwise("fdt_pairwise_distance", box_a, box_b)


def calc_pr(predict, truth, iou_thresh=0.5):
    """predict [P,5] (x1,y1,x2,y2,score), truth [T,4] (x,y,w,h) -> ([[hit flags], [scores]], T); reference :77-92:
    a prediction is a hit when its best IoU over all truth boxes exceeds `iou_thresh` (strictly)."""
    predict = np.asarray(predict)
    truth = np.asarray(truth)
    corners = np.concatenate((truth[:, :2], truth[:, :2] + truth[:, 2:]), axis=1)
    overlaps = calculate_iou(corners, predict[:, :4])                    # [T, P]
    hits = (overlaps.max(axis=0) > iou_thresh).astype(np.int32)
    return np.vstack((hits, predict[:, 4])), overlaps.shape[0]
