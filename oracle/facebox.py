"""CPU oracle (test infrastructure only, see oracle/__init__.py): FaceBoxes forward + anchors +
decode_np/nms_np, restating reference FACEBOX/networks.py:43-57,87-116,
FACEBOX/multibox_layer.py:28-50 and FACEBOX/encoderl.py:12-48,217-266,308-325."""
import itertools

import numpy as np
import torch
import torch.nn.functional as F

F32 = np.float32


def _t(sd):
    return {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))) for k, v in sd.items()}


def _cbr(sd, n, x, stride=1, padding=0):
    # conv_bn_relu: networks.py:11-16
    x = F.conv2d(x, sd[n + ".0.weight"], sd[n + ".0.bias"], stride, padding)
    x = F.batch_norm(x, sd[n + ".1.running_mean"], sd[n + ".1.running_var"], sd[n + ".1.weight"],
                     sd[n + ".1.bias"], False, 0.0, 1e-5)
    return F.relu(x)


def _inception(sd, n, x):
    x1 = _cbr(sd, n + ".conv1", x)
    x2 = _cbr(sd, n + ".conv2", F.max_pool2d(x, kernel_size=3, stride=1, padding=1))
    x3 = _cbr(sd, n + ".conv4", _cbr(sd, n + ".conv3", x), 1, 1)
    x4 = _cbr(sd, n + ".conv7", _cbr(sd, n + ".conv6", _cbr(sd, n + ".conv5", x), 1, 1), 1, 1)
    return torch.cat([x1, x2, x3, x4], 1)


def _bn(sd, n, x):
    return F.batch_norm(x, sd[n + ".running_mean"], sd[n + ".running_var"], sd[n + ".weight"], sd[n + ".bias"],
                        False, 0.0, 1e-5)


@torch.no_grad()
def forward(sd, x, want=()):
    """x [B,3,1024,1024] f32 (BGR/255).  Returns raw loc [B,21824,4], conf [B,21824,2] (+ `want`)."""
    sd = _t(sd)
    x = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.asarray(x, dtype=F32))
    t = {}
    x = _bn(sd, "bn1", F.conv2d(x, sd["conv1.weight"], sd["conv1.bias"], 4, 3))         # :89-90
    x = F.relu(torch.cat([x, -x], 1))                                                  # :92
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    t["crelu_pool1"] = x
    x = _bn(sd, "bn2", F.conv2d(x, sd["conv2.weight"], sd["conv2.bias"], 2, 2))
    x = F.relu(torch.cat([x, -x], 1))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    t["crelu_pool2"] = x
    for n in ("inception1", "inception2", "inception3"):
        x = _inception(sd, n, x)
        t[n] = x
    hs = [x]
    x = _cbr(sd, "conv3_2", _cbr(sd, "conv3_1", x), 2, 1)
    hs.append(x)
    x = _cbr(sd, "conv4_2", _cbr(sd, "conv4_1", x), 2, 1)
    hs.append(x)
    t.update(hs0=hs[0], hs1=hs[1], hs2=hs[2])
    locs, confs = [], []
    for i, h in enumerate(hs):                                                         # multibox_layer.py:34-48
        yl = F.conv2d(h, sd["multilbox.loc_layers.%d.weight" % i], sd["multilbox.loc_layers.%d.bias" % i], 1, 1)
        locs.append(yl.permute(0, 2, 3, 1).contiguous().view(yl.size(0), -1, 4))
        yc = F.conv2d(h, sd["multilbox.conf_layers.%d.weight" % i], sd["multilbox.conf_layers.%d.bias" % i], 1, 1)
        confs.append(yc.permute(0, 2, 3, 1).contiguous().view(yc.size(0), -1, 2))
    out = {"loc": torch.cat(locs, 1).numpy(), "conf": torch.cat(confs, 1).numpy()}
    for k in want:
        out[k] = t[k].numpy()
    return out


def anchors():
    """DataEncoder.__init__: encoderl.py:21-47 (Python floats, one rounding to f32)."""
    scale = 1024.
    steps = [s / scale for s in (32, 64, 128)]
    sizes = [s / scale for s in (32, 256, 512)]
    aspect_ratios = ((1, 2, 4), (1,), (1,))
    feature_map_sizes = (32, 16, 8)
    density = [[-3, -1, 1, 3], [-1, 1], [0]]
    boxes = []
    for i in range(3):
        fm = feature_map_sizes[i]
        for h, w in itertools.product(range(fm), repeat=2):
            cx = (w + 0.5) * steps[i]
            cy = (h + 0.5) * steps[i]
            s = sizes[i]
            for j, ar in enumerate(aspect_ratios[i]):
                if i == 0:
                    for dx, dy in itertools.product(density[j], repeat=2):
                        boxes.append((cx + dx / 8. * s * ar, cy + dy / 8. * s * ar, s * ar, s * ar))
                else:
                    boxes.append((cx, cy, s * ar, s * ar))
    return np.array(boxes, dtype=np.float64).astype(F32)


def nms_np(bboxes, scores, threshold=0.5):
    """encoderl.py:217-266 (mode "Union").  Tie order: the reference's argsort()[::-1] is an unstable
    quicksort (unpinned); here ties go to the higher index first, like the PyramidBox path."""
    x1, y1, x2, y2 = bboxes.transpose()
    areas = (x2 - x1) * (y2 - y1)
    order = np.argsort(scores, kind="stable")[::-1]
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(i)
        xx1 = np.maximum(x1[i], x1[order[1:]])
        yy1 = np.maximum(y1[i], y1[order[1:]])
        xx2 = np.minimum(x2[i], x2[order[1:]])
        yy2 = np.minimum(y2[i], y2[order[1:]])
        w = np.maximum(F32(0.0), xx2 - xx1)
        h = np.maximum(F32(0.0), yy2 - yy1)
        inter = w * h
        with np.errstate(all="ignore"):
            ovr = inter / (areas[i] + areas[order[1:]] - inter)
        order = order[1:][ovr < F32(threshold)]
    return keep


def decode_np(loc, conf, default_boxes, conf_thres=0.35, nms_thres=0.5):
    """encoderl.py:308-325.  conf is softmaxed.  numpy f32 arithmetic incl. np.exp like the reference."""
    loc = np.asarray(loc, F32)
    score = np.asarray(conf, F32)[:, 1]
    ids = np.where(score > F32(conf_thres))[0]
    cxcy = loc[ids, :2] * F32(0.1) * default_boxes[ids, 2:] + default_boxes[ids, :2]
    wh = np.exp(loc[ids, 2:] * F32(0.2)) * default_boxes[ids, 2:]
    boxes = np.hstack([cxcy - wh / 2, cxcy + wh / 2])
    keep = nms_np(boxes, score[ids], nms_thres)
    return boxes[keep], score[ids][keep]


def detect(sd, frame_u8_1024, conf_thres=0.35):
    """My_test_facebox.py:12-36 after the resize."""
    x = torch.from_numpy(np.ascontiguousarray(frame_u8_1024.transpose(2, 0, 1))).float().div(255)[None]
    o = forward(sd, x)
    conf = torch.softmax(torch.from_numpy(o["conf"][0]), dim=1).numpy()
    return decode_np(o["loc"][0], conf, anchors(), conf_thres)
