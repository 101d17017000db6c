"""PR / ROC data of the reference's only accuracy metric (reference draw_curve/draw_pr_roc.py:5-34),
without the matplotlib plotting: `gen_tp_fp` with the reference's signature plus `pr_roc` /
`average_precision` helpers used by the parity report ("mAP vs CPU ref", BASELINE.json)."""
import numpy as np


def gen_tp_fp(tf_conf):
    """tf_conf [2,M] (row 0: matched flag, row 1: score, already sorted by score descending) ->
    (true_pos[M], false_pos[M]) cumulative counts.  reference draw_pr_roc.py:5-19 (vectorised)."""
    _, M = tf_conf.shape
    true_pos = np.cumsum(tf_conf[0, :] != 0).astype(np.float64)
    false_pos = np.arange(1, M + 1, dtype=np.float64) - true_pos
    return true_pos, false_pos


def pr_roc(data):
    """data = the array My_test.py saves: [2, M+1] whose last column is [0, truth_num]
    (reference My_test.py:169-171).  Returns recall, precision, false_pos (draw_pr_roc.py:28-34)."""
    truth_num = data[1, -1]
    tp, fp = gen_tp_fp(data[:, :-1])
    with np.errstate(divide="ignore", invalid="ignore"):
        recall = tp / truth_num
        precision = tp / (tp + fp)
    return recall, precision, fp


def average_precision(data):
    """Area under the PR curve (step integration over recall).  The reference only plots the curve;
    this scalar is the build's summary of it."""
    recall, precision, _ = pr_roc(data)
    if recall.size == 0:
        return 0.0
    r = np.concatenate([[0.0], recall])
    return float(np.sum((r[1:] - r[:-1]) * precision))
