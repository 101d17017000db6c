"""GPU test of the iouTracke_cal-compatible entry point: detect_face unpack, device-resident vs
host-stepped association, track-ID/boxes bit-exact vs the oracle tracker given identical boxes, and
the .npy track file the reference's display script reads."""
import importlib
import os

import numpy as np
import pytest

from oracle import postproc as opp

pytestmark = pytest.mark.gpu


def M(name):
    return importlib.import_module("face-detection-and-tracking_amd." + name)


@pytest.fixture(scope="module")
def entry(synth):
    cal = M("iouTracke_cal")
    sd = synth.make_state_dict("res50", seed=0, conf_shift=-3.0)    # plenty of scores >= 0.4
    cal.load_net(sd, 160, 128, which='repo')
    cal.net.detect = M("layers").Detect(2, 0, 750, 0.3, 0.5)
    yield cal
    cal.net.close()


def frames_sequence(synth):
    base = synth.make_frames(3, 128, 160, seed=77)
    return [base[i // 7] for i in range(21)]        # each frame shown 7 times: tracks persist


def test_detect_face_matches_oracle_unpack(entry, synth):
    fr = frames_sequence(synth)[0]
    det = entry.detect_face(fr, 1)
    y = entry.net(fr).numpy()
    exp = opp.unpack_detections(y, 160, 128, 0.4)
    assert det.dtype == exp.dtype and np.array_equal(det, exp) and det.shape[0] > 3
    with pytest.raises(NotImplementedError):
        entry.detect_face(fr, 2)


def test_track_device_resident_equals_host_and_oracle(entry, synth, tmp_path):
    frames = frames_sequence(synth)
    t_dev = entry.track(frames, device_resident=True)
    t_host = entry.track(frames, device_resident=False)
    ref = opp.IouTracker(entry.sigma_iou, entry.sigma_h, entry.t_min)
    for f in frames:
        with np.errstate(all="ignore"):
            ref.step(entry.detect_face(f, 1))
    t_ref = [{"bboxes": [list(map(float, b)) for b in t["bboxes"]], "max_score": float(t["max_score"]),
              "start_frame": t["start_frame"]} for t in ref.finish()]
    assert len(t_ref) > 3
    assert t_dev == t_ref and t_host == t_ref          # track-ID assignment bit-exact given identical boxes
    # on-disk schema of reference iouTracke_cal.py:177 / iouTracke_display.py:29
    p = str(tmp_path / "video.npy")
    entry.save_tracks(t_dev, p)
    back = np.load(p, allow_pickle=True).tolist()
    assert back == t_dev and set(back[0]) == {"bboxes", "max_score", "start_frame"}


def test_my_test_evaluation_harness(synth):
    """Config-1 plumbing (reference My_test.py): per-image priorbox reset, threshold-0 row walk (all
    2x750 zero-padded rows pass), calc_pr accumulation; PR data == oracle's on the same detections."""
    mt = M("My_test")
    pr = M("draw_curve.draw_pr_roc")
    net = M("pyramid").build_sfd('test', 640, 2)
    net.load_state_dict(synth.make_state_dict("res50", seed=0, conf_shift=-3.0))
    mt.net, mt.net_name, mt.threshold = net, 'repo', 0.0
    frames = synth.make_frames(2, 96, 128, seed=3)
    det = mt.detect_face(frames[0])
    assert det.shape == (1500, 5)                      # 2 x 750 rows at threshold 0
    mt.threshold = 0.5
    samples = []
    for f in frames:
        d = mt.detect_face(f)
        truth = np.column_stack((d[:3, 0], d[:3, 1], d[:3, 2] - d[:3, 0], d[:3, 3] - d[:3, 1]))
        samples.append((f, truth))
    data = mt.evaluate(samples, 0.5)
    assert data.shape[0] == 2 and data[1, -1] == 6 and data[0, -1] == 0
    rec, prec, fp = pr.pr_roc(data)
    tp_o, fp_o = opp.gen_tp_fp(data[:, :-1])
    assert np.array_equal(fp, fp_o) and np.array_equal(rec, tp_o / 6)
    assert rec[-1] == 1.0 and 0 < pr.average_precision(data) <= 1.0
    net.close()


def test_device_side_resize_ingest(entry, synth):
    """8(f)-1: cv2.resize(frame,(W,H)) + mean-subtract on the GPU.  Bit-exact vs the oracle's restatement of
    OpenCV's 8-bit INTER_LINEAR path (parity with cv2 itself is unpinned: cv2 is not available)."""
    from oracle import ingest
    rng = np.random.default_rng(12)
    src = rng.integers(0, 256, (270, 480, 3), dtype=np.uint8)          # 1080p / 4
    small = ingest.resize_linear_u8(src, 160, 128)
    assert small.shape == (128, 160, 3)
    y_ref = entry.net(small).numpy()                   # resize on the host (oracle), ingest on the GPU
    stem_ref = entry.net.get_tensor("input")
    y = entry.net.forward_resized(src, (160, 128)).numpy()
    assert np.array_equal(entry.net.get_tensor("input"), stem_ref)    # the resized+mean-subtracted tensor: bit-exact
    assert np.array_equal(y, y_ref)
    # identity size goes through the same kernel and must be the identity
    y_id = entry.net.forward_resized(small, (160, 128)).numpy()
    assert np.array_equal(y_id, y_ref)
