"""Evaluation driver mirroring reference My_test.py (config 1 plumbing): per image `detect_face(x)`
re-creates the priorbox for the image size, resets `firstTime`, installs `Detect(2,0,750,threshold,0.35)`
(reference My_test.py:31-36), runs the detector on the GPU and walks the rows while
`score >= threshold` (:47-56); `evaluate()` accumulates `calc_pr` columns, sorts them by score and
appends `[0, truth_num]` (:163-171) -- the array the PR/ROC script consumes.

Dataset iteration (`Data_collector`, cv2.imread) is outside the path: `evaluate` takes any iterable of
(image uint8 BGR HWC, target [n,4] x,y,w,h)."""
import numpy as np

from .layers import Detect, PriorBoxLayer
from .utils.calc_performance import calc_pr

net = None
net_name = 'repo'
threshold = 0.0          # reference argparse default (My_test.py:80): every zero-padded row passes


def detect_face(x):
    height, width, _ = x.shape
    if net_name in ('repo', 'repo_my', 'try1', 'try2'):
        net.priorbox = PriorBoxLayer(width, height)
    else:
        net.priorbox = PriorBoxLayer(width, height, stride=[4, 8, 16, 32, 64], box=(16, 32, 64, 128, 256))
    net.firstTime = True
    net.detect = Detect(2, 0, 750, threshold, 0.35)
    detections = net(np.ascontiguousarray(x, dtype=np.uint8)).numpy()
    scale = np.array([width, height, width, height], dtype=np.float32)
    boxes, scores = [], []
    for i in range(detections.shape[1]):
        j = 0
        while detections[0, i, j, 0] >= np.float32(threshold):
            scores.append(detections[0, i, j, 0])
            boxes.append(detections[0, i, j, 1:] * scale)
            j += 1
            if j >= detections.shape[2]:
                break
    if len(boxes) == 0:
        return np.array([[0, 0, 0, 0, 0.4]])
    b = np.array(boxes, dtype=np.float32)
    return np.column_stack((b[:, 0], b[:, 1], b[:, 2], b[:, 3], np.array(scores, dtype=np.float64)))


def evaluate(samples, iou_thresh=0.5):
    """Returns the [2, M+1] array reference My_test.py:169-171 saves as data_of_<net>.npy."""
    tf_conf = np.zeros((2, 0))
    truth_num = 0
    for image, target in samples:
        predict = detect_face(image)
        tf_conf_, truth_num_ = calc_pr(predict, np.asarray(target, dtype=np.float64), iou_thresh=iou_thresh)
        tf_conf = np.hstack((tf_conf, tf_conf_))
        truth_num += truth_num_
    tf_conf = tf_conf[:, np.argsort(tf_conf[1, :])[::-1]]
    return np.hstack((tf_conf, [[0], [truth_num]]))
