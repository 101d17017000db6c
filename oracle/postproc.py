"""CPU oracle (test infrastructure only, see oracle/__init__.py): SSD post-processing
and the IoU tracker of the reference, restated with numpy.

All arithmetic is IEEE f32/f64 basic operations in the reference's operand order,
so it is bit-identical to the reference's torch-CPU / numpy path; the one
transcendental (`exp` in `decode`) goes through `torch.exp` on CPU, the same
routine the reference calls (reference layers/box_utils.py:255).
"""
import math

import numpy as np
import torch

F32 = np.float32


# --------------------------------------------------------------------------- priors
class PriorBoxLayer:
    """Restates reference layers/functions/prior_box.py:9-44."""

    def __init__(self, width, height, stride=(4, 8, 16, 32, 64, 128),
                 box=(16, 32, 64, 128, 256, 512), scale=(1, 1, 1, 1, 1, 1),
                 aspect_ratios=([], [], [], [], [], [])):
        self.width, self.height = width, height
        self.stride, self.box = stride, box
        self.scales, self.aspect_ratios = scale, aspect_ratios

    def __call__(self, prior_idx, f_width, f_height):
        # prior_box.py:31-41 -- Python floats are f64; the f32 cast happens once in
        # torch.Tensor(mean) (prior_box.py:43).
        rows = []
        st, bx = self.stride[prior_idx], self.box[prior_idx]
        j = np.arange(f_width, dtype=np.float64)
        i = np.arange(f_height, dtype=np.float64)
        cx = (j + 0.5) * st / self.width
        cy = (i + 0.5) * st / self.height
        per_cell = []
        for scale in range(self.scales[prior_idx]):
            box_scale = (2 ** (1 / 3)) ** scale
            side_x = bx * box_scale / self.width
            side_y = bx * box_scale / self.height
            per_cell.append((side_x, side_y))
            for ar in self.aspect_ratios[prior_idx]:
                per_cell.append((side_x / math.sqrt(ar), side_y * math.sqrt(ar)))
        out = np.empty((f_height, f_width, len(per_cell), 4), dtype=np.float64)
        out[..., 0] = cx[None, :, None]
        out[..., 1] = cy[:, None, None]
        for k, (sx, sy) in enumerate(per_cell):
            out[:, :, k, 2] = sx
            out[:, :, k, 3] = sy
        return out.reshape(-1, 4).astype(F32)


def feature_sizes(height, width, arch="res50"):
    """Spatial size of each detection source for an HxW input.

    conv k7/s2/p3, maxpool k3/s2/p1 and every stride-2 3x3/p1 conv map n -> (n-1)//2+1
    (reference pyramid.py:229-236; pyramid_mb2_try3.py:229-238).
    """
    def half(n):
        return (n - 1) // 2 + 1
    h, w = half(half(height)), half(half(width))       # stride 4
    n = 6 if arch == "res50" else 5
    out = []
    for _ in range(n):
        out.append((h, w))
        h, w = half(h), half(w)
    return out


def build_priors(priorbox, height, width, arch="res50", sizes=None):
    """cat of priorbox(idx, f_w, f_h) over the sources (reference pyramid.py:275-283).  `sizes` = the (h, w)
    of the sources when they do not follow the stride pyramid (try4 / try5: 1x1 convs with padding 1)."""
    if sizes is None:
        sizes = feature_sizes(height, width, arch)
    return np.concatenate([priorbox(idx, fw, fh) for idx, (fh, fw) in enumerate(sizes)], 0)


# --------------------------------------------------------------------------- decode
def decode(loc, priors, variances=(0.1, 0.2)):
    """Restates reference layers/box_utils.py:238-258 (f32, reference operand order)."""
    loc = np.asarray(loc, dtype=F32)
    priors = np.asarray(priors, dtype=F32)
    v0, v1 = F32(variances[0]), F32(variances[1])
    cxcy = priors[:, :2] + (loc[:, :2] * v0) * priors[:, 2:]
    e = torch.exp(torch.from_numpy(np.ascontiguousarray(loc[:, 2:] * v1))).numpy()
    wh = priors[:, 2:] * e
    boxes = np.concatenate([cxcy, wh], 1).astype(F32)
    boxes[:, :2] -= boxes[:, 2:] / F32(2)
    boxes[:, 2:] += boxes[:, :2]
    return boxes


def softmax2(conf):
    """nn.Softmax(dim=-1) on [...,2] logits (reference pyramid.py:197,332), torch-CPU."""
    return torch.softmax(torch.from_numpy(np.ascontiguousarray(conf, dtype=F32)), -1).numpy()


# --------------------------------------------------------------------------- nms
def nms(boxes, scores, overlap=0.5, top_k=200):
    """Restates reference layers/box_utils.py:275-340.

    Returns (keep[int64, n] zero-padded, count).  Ascending *stable* sort, walk from
    the top: among equal scores the highest index is taken first.
    """
    boxes = np.asarray(boxes, dtype=F32)
    scores = np.asarray(scores, dtype=F32)
    n = scores.shape[0]
    keep = np.zeros(n, dtype=np.int64)
    if boxes.size == 0:
        return keep, 0
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    area = (x2 - x1) * (y2 - y1)                      # :295
    idx = np.argsort(scores, kind="stable")           # :296
    idx = idx[-top_k:]                                # :298
    thr = F32(overlap)
    count = 0
    while idx.size > 0:
        i = idx[-1]                                   # :309
        keep[count] = i
        count += 1
        if idx.size == 1:
            break
        idx = idx[:-1]
        xx1 = np.maximum(x1[idx], x1[i])              # :321-324
        yy1 = np.maximum(y1[idx], y1[i])
        xx2 = np.minimum(x2[idx], x2[i])
        yy2 = np.minimum(y2[idx], y2[i])
        w = np.maximum(xx2 - xx1, F32(0))             # :327-331
        h = np.maximum(yy2 - yy1, F32(0))
        inter = w * h
        union = (area[idx] - inter) + area[i]         # :336
        with np.errstate(divide="ignore", invalid="ignore"):
            iou = inter / union
        idx = idx[iou < thr]                          # :339 (NaN is dropped, == thr is dropped)
    return keep, count


# --------------------------------------------------------------------------- Detect
class Detect:
    """Restates reference layers/functions/detection.py:9-84."""

    def __init__(self, num_classes, bkg_label, top_k, conf_thresh, nms_thresh):
        self.num_classes = num_classes
        self.background_label = bkg_label
        self.top_k = top_k
        self.nms_thresh = nms_thresh
        if nms_thresh <= 0:
            raise ValueError('nms_threshold must be non negative.')   # :28-29
        self.conf_thresh = conf_thresh
        self.variance = [0.1, 0.2]                                    # data/config.py:18
        self.nms_top_k = 5000

    def __call__(self, loc_data, conf_data, prior_data):
        loc_data = np.asarray(loc_data, dtype=F32)
        prior_data = np.asarray(prior_data, dtype=F32)
        num = loc_data.shape[0]
        num_priors = prior_data.shape[0]
        conf = np.asarray(conf_data, dtype=F32).reshape(num, num_priors, self.num_classes)
        loc_data = loc_data.reshape(num, num_priors, 4)
        output = np.zeros((num, self.num_classes, self.top_k, 5), dtype=F32)
        counts = np.zeros((num, self.num_classes), dtype=np.int64)
        for i in range(num):
            decoded = decode(loc_data[i], prior_data, self.variance)   # :55
            for cl in range(1, self.num_classes):
                sc = conf[i, :, cl]
                mask = sc > F32(self.conf_thresh)                      # :64 strict
                sel = np.nonzero(mask)[0]
                if sel.size == 1:
                    continue       # :66-72: 0-dim after squeeze() -> `continue`, nothing emitted
                scores = sc[sel]
                boxes = decoded[sel]
                ids, count = nms(boxes, scores, self.nms_thresh,
                                 min(boxes.shape[0], self.nms_top_k))  # :79
                count = min(count, self.top_k)                         # :80
                ids = ids[:count]
                output[i, cl, :count, 0] = scores[ids]
                output[i, cl, :count, 1:] = boxes[ids]
                counts[i, cl] = count
        self.last_counts = counts
        return output


# --------------------------------------------------------------------------- IoU
def intersect(box_a, box_b):
    """Restates reference utils/calc_performance.py:4-31."""
    a_hi = box_a[:, None, 2:]
    a_lo = box_a[:, None, :2]
    b_hi = box_b[None, :, 2:]
    b_lo = box_b[None, :, :2]
    d = np.minimum(a_hi, b_hi) - np.maximum(a_lo, b_lo)
    d = np.maximum(d, 0)
    return d[:, :, 0] * d[:, :, 1]


def calculate_iou(box_a, box_b):
    """Restates reference utils/calc_performance.py:54-74 ([A,4]x[B,4] -> [A,B], no eps)."""
    box_a = np.asarray(box_a)
    box_b = np.asarray(box_b)
    inter = intersect(box_a, box_b)
    area_a = ((box_a[:, 2] - box_a[:, 0]) * (box_a[:, 3] - box_a[:, 1]))[:, None]
    area_b = ((box_b[:, 2] - box_b[:, 0]) * (box_b[:, 3] - box_b[:, 1]))[None, :]
    union = area_a + area_b - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        return inter / union


def calculate_distance(box_a, box_b):
    """Restates reference utils/calc_performance.py:34-51 ([A,4]x[B,4] -> [A,B]): fourth root of
    (mean extent difference)^2 + (centre difference)^2, numpy broadcasting in the reference's operand order."""
    box_a = np.asarray(box_a)
    box_b = np.asarray(box_b)
    a_hi, a_lo = box_a[:, None, 2:], box_a[:, None, :2]          # the reference calls these *_x1y1 / *_x2y2 (:37-40)
    b_hi, b_lo = box_b[None, :, 2:], box_b[None, :, :2]
    a_dxdy = a_hi - a_lo
    b_dxdy = b_hi - b_lo
    ca_xy = (a_hi + a_lo) / 2
    cb_xy = (b_hi + b_lo) / 2
    delt_xy = cb_xy - ca_xy
    delt_dxdy = a_dxdy - b_dxdy
    delt_z = (delt_dxdy[:, :, 0] + delt_dxdy[:, :, 1]) / 2
    dis = delt_z * delt_z + delt_xy[:, :, 0] * delt_xy[:, :, 0] + delt_xy[:, :, 1] * delt_xy[:, :, 1]
    return dis ** 0.25


def calc_pr(predict, truth, iou_thresh=0.5):
    """Restates reference utils/calc_performance.py:77-92."""
    truth = np.hstack((truth[:, :2], truth[:, 2:] + truth[:, :2]))
    iou = calculate_iou(truth, predict[:, :4])
    truth_num, _ = iou.shape
    tf = (np.max(iou, 0) > iou_thresh).astype(np.int32)
    return np.vstack((tf, predict[:, 4])), truth_num


# --------------------------------------------------------------------------- host unpack
def unpack_detections(detections, width, height, score_thresh=0.4, shrink=1):
    """Restates reference iouTracke_cal.py:53-84 on a [1,2,top_k,5] f32 array."""
    det = np.asarray(detections, dtype=F32)
    scale = np.array([width, height, width, height], dtype=F32)
    boxes, scores = [], []
    for i in range(det.shape[1]):
        j = 0
        while det[0, i, j, 0] >= F32(score_thresh):
            scores.append(det[0, i, j, 0])
            boxes.append(det[0, i, j, 1:] * scale)
            j += 1
            if j >= det.shape[2]:
                break
    if len(boxes) == 0:
        return np.array([[0, 0, 0, 0, 0.4]])                   # :73-74 (f64 dummy)
    b = np.array(boxes, dtype=F32) / shrink
    return np.column_stack((b[:, 0], b[:, 1], b[:, 2], b[:, 3],
                            np.array(scores, dtype=F32)))


# --------------------------------------------------------------------------- tracker
class IouTracker:
    """Restates the inline tracker of reference iouTracke_cal.py:113-156,174-177."""

    def __init__(self, sigma_iou=0.4, sigma_h=0.6, t_min=5, use_iou=True, sigma_dis=8):
        self.sigma_iou, self.sigma_h, self.t_min = sigma_iou, sigma_h, t_min
        self.use_iou, self.sigma_dis = use_iou, sigma_dis        # :23,:26 (module-level configuration)
        self.frame_num = 0
        self.tracks_active = []
        self.tracks_finished = []

    def step(self, det0):
        self.frame_num += 1                                      # :118
        dets = np.asarray(det0).tolist()                         # :127
        updated = []
        for track in self.tracks_active:                         # :129
            if len(dets) > 0:                                    # :130 (no else: track dropped)
                if self.use_iou:
                    iou = calculate_iou(np.array(dets)[:, :4], np.array([track['bboxes'][-1]]))
                    best = int(iou.argmax())                     # :133 (NaN wins argmax)
                    matched = iou[best] > self.sigma_iou         # :134
                else:
                    dis = calculate_distance(np.array(dets)[:, :4], np.array([track['bboxes'][-1]]))
                    best = int(dis.argmin())                     # :137 (NaN wins argmin too)
                    matched = dis[best] < self.sigma_dis         # :138
                if matched:
                    track['bboxes'].append(dets[best][:4])
                    track['max_score'] = max(track['max_score'], dets[best][4])
                    updated.append(track)
                    del dets[best]
                else:
                    if track['max_score'] > self.sigma_h and len(track['bboxes']) > self.t_min:
                        self.tracks_finished.append(track)       # :146-148
        new = [{'bboxes': [d[:4]], 'max_score': d[4], 'start_frame': self.frame_num}
               for d in dets]                                    # :150-154
        self.tracks_active = updated + new

    def finish(self):
        self.tracks_finished += [t for t in self.tracks_active
                                 if t['max_score'] > self.sigma_h
                                 and len(t['bboxes']) >= self.t_min]   # :174-175
        return self.tracks_finished


# --------------------------------------------------------------------------- PR metric
def gen_tp_fp(tf_conf):
    """Restates reference draw_curve/draw_pr_roc.py:5-19 (the explicit loop)."""
    _, M = tf_conf.shape
    true_pos, false_pos = np.zeros(M), np.zeros(M)
    for i in range(1, M + 1):
        true_pos[i - 1] = np.count_nonzero(tf_conf[0, :i])
        false_pos[i - 1] = i - true_pos[i - 1]
    return true_pos, false_pos


def ap_against_reference(pred_list, ref_list, iou_thresh=0.5):
    """"mAP vs CPU ref" (SURVEY.md 8(d)): treat the CPU-reference detections of every frame as ground
    truth, match the candidate detections with calc_pr semantics (calc_performance.py:77-92), sort by
    score (My_test.py:169) and integrate the PR curve of draw_pr_roc.py:28-31.
    pred_list / ref_list: per frame arrays [n,5] (x1,y1,x2,y2,score)."""
    tf_conf = np.zeros((2, 0))
    truth_num = 0
    for pred, ref in zip(pred_list, ref_list):
        truth = np.column_stack((ref[:, 0], ref[:, 1], ref[:, 2] - ref[:, 0], ref[:, 3] - ref[:, 1]))
        with np.errstate(all="ignore"):
            tf, tn = calc_pr(np.asarray(pred, dtype=np.float64), truth.astype(np.float64), iou_thresh)
        tf_conf = np.hstack((tf_conf, tf))
        truth_num += tn
    tf_conf = tf_conf[:, np.argsort(tf_conf[1, :], kind="stable")[::-1]]
    tp, fp = gen_tp_fp(tf_conf)
    recall = tp / max(truth_num, 1)
    precision = tp / np.maximum(tp + fp, 1)
    r = np.concatenate([[0.0], recall])
    return float(np.sum((r[1:] - r[:-1]) * precision)), int(truth_num), int(tf_conf.shape[1])
