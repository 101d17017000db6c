"""The kernel plans bench.py TIMES, pinned against the oracle at their own size (review round 4, weak A1).

A committed plan under face-detection-and-tracking_amd/tuned/ applies only when the forward's batch equals the plan's
(csrc/model.hip: the hints carry their B), so a batch-1 call on a handle holding `res50_1024x1024_b4.plan` runs the analytic
plan instead -- which is what every full-size parity test did until now.  Here the timed path itself is driven: the C-ABI
pipeline (`fdt_pipeline_step_frame` for the grouped plans, `fdt_pipeline_step` for the batch-1 ones), with the committed plan,
over distinct full-size frames -- the reference fixture's frame in every batch position plus seeded frames -- and every frame's
Detect record is compared with `oracle.pyramidbox.detect_frame` (reference pyramid.py:218-351 + layers/functions/detection.py:
34-84: same count, IoU within 1e-3, scores within 1e-4), the fixture frame also with what the REFERENCE produced
(tests/golden/nets.npz), and the tracks with the oracle tracker (reference iouTracke_cal.py:126-156,174-177) -- bit-equal on
the GPU's own detections, structurally equal (same tracks, boxes within tolerance) on the oracle's detections.
`test_every_committed_plan_is_pinned` fails when a plan file exists that no parity test imports."""
import importlib
import os
import re

import numpy as np
import pytest
import torch

from conftest import load_npz
from oracle import postproc as opp
from oracle import pyramidbox as opb

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TUNED = os.path.join(ROOT, "face-detection-and-tracking_amd", "tuned")
BOX_IOU_TOL = 1e-3
SCORE_ATOL = 1e-4


def M(name):
    return importlib.import_module("face-detection-and-tracking_amd." + name)


# (plan file, fixture key of tests/golden/nets.npz, frames per launch chain, slots)
TIMED = [
    ("res50_1024x1024_b4.plan", "res50_1024x1024", 4, 4),    # the headline (bench.py default at 1024^2)
    ("res50_1024x1024_b1.plan", "res50_1024x1024", 1, 8),    # its `ungrouped` leg = BASELINE's batch-1 wording
    ("res50_640x480_b8.plan", "res50_480x640", 8, 2),        # C4 default (the tracker's frame size)
    ("res50_640x480_b1.plan", "res50_480x640", 1, 8),        # C4 --group 1, iouTracke_cal.track()
    ("res50_1024x1024_b2.plan", "res50_1024x1024", 2, 4),
    ("res50_1024x1024_b8.plan", "res50_1024x1024", 8, 2),
    ("res50_640x480_b2.plan", "res50_480x640", 2, 4),
    ("res50_640x480_b4.plan", "res50_480x640", 4, 4),
]
# plans pinned against a reference fixture elsewhere: (file, test that imports it)
PINNED_ELSEWHERE = {
    "res50_1920x1080_b1.plan": "tests/test_gpu_model.py::test_res50_native_1080p_vs_reference_fixture",
    "try3_1024x1024_b8.plan": "tests/test_gpu_model.py::test_try3_1024_batch8_vs_reference_fixture",
    "try3_1024x1024_b1.plan": "tests/test_gpu_timed_plans.py::test_try3_batch1_plan_vs_reference_fixture",
    "facebox_1024x1024_b16.plan": "tests/test_gpu_facebox.py::test_config5_as_baseline_words_it_4k_sources_and_batch_16_in_one_call",
}


def test_every_committed_plan_is_pinned():
    """CPU-checkable bookkeeping: every file under tuned/ is named by a parity test that imports it."""
    files = sorted(f for f in os.listdir(TUNED) if f.endswith(".plan"))
    named = {t[0] for t in TIMED} | set(PINNED_ELSEWHERE)
    assert [f for f in files if f not in named] == []
    for f, where in PINNED_ELSEWHERE.items():
        if f not in files:
            continue
        path, fn = where.split("::")
        src = open(os.path.join(ROOT, path)).read()
        assert "def %s(" % fn in src, where
        stem = f[:-5]
        arch, w, h, b = re.match(r"([a-z0-9]+)_(\d+)x(\d+)_b(\d+)", stem).groups()
        body = src[src.index("def %s(" % fn):]
        body = body[:body.index("\ndef ", 1)] if "\ndef " in body[1:] else body
        assert "tuned_plan_text" in body or stem in body, "%s does not import %s" % (where, f)


def test_every_row_of_every_committed_plan_names_an_instantiated_kernel():
    """CPU-checkable: a plan row `layer kind tile split map [combine]` must name a (kernel class, tile) the library instantiates --
    a stale row would be ignored silently (model.hip: a hint that does not fit the layer falls back to the analytic choice) and the
    timed handle would no longer run the committed plan -- and, within one file, a layer appears once."""
    import ctypes
    L = importlib.import_module("face-detection-and-tracking_amd._lib").lib()
    L.fdt_debug_conv_class.restype = ctypes.c_int
    L.fdt_debug_conv_class.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    files = sorted(f for f in os.listdir(TUNED) if f.endswith(".plan"))
    assert files
    for f in files:
        seen = set()
        rows = [l.split() for l in open(os.path.join(TUNED, f)).read().splitlines() if l.strip()]
        assert rows[0][0] == "shape" and len(rows[0]) == 4, (f, rows[0])
        for r in rows[1:]:
            assert len(r) in (5, 6), (f, r)
            name, kind, tile, split, mp = r[0], int(r[1]), int(r[2]), int(r[3]), int(r[4])
            assert name not in seen, (f, name)
            seen.add(name)
            base = ctypes.c_int(-1)
            assert L.fdt_debug_conv_class(kind, tile, ctypes.byref(base)) == 1, "%s: %s names class %d tile %d, not instantiated" % (f, name, kind, tile)
            assert split >= 1 and 0 <= mp < 8, (f, r)
            if "downsample" in name and "layer1" not in name:
                assert base.value == 1, (f, r)          # the stride-2 1x1 of pyramid.py:87-91
        assert len(seen) >= 20, (f, len(seen))


@pytest.fixture(scope="module")
def oracle_threads():
    n = torch.get_num_threads()
    torch.set_num_threads(min(16, os.cpu_count() or 16))     # the oracle's best count on the GPU boxes (bench.py sweep)
    yield
    torch.set_num_threads(n)


_ORACLE = {}


def oracle_record(sd, frame, key):
    """oracle Detect record [1,2,750,5] of a frame (default Res50 Detect: 0.3 / 0.5), cached per (size, seed)."""
    if key not in _ORACLE:
        _ORACLE[key] = opb.detect_frame(sd, frame, "res50")
    return _ORACLE[key]


def frames_for(synth, H, W, fixture_seed, G, n_groups):
    """n_groups x G frames.  The fixture's frame sits at a different batch entry in every group (entry g * (G - 1) // (n_groups
    - 1): first ... last), for G = 1 it is frame 0; the other entries cycle through eight seeded frames, so the oracle runs
    on nine distinct frames whatever G is."""
    fix = synth.make_frames(1, H, W, seed=fixture_seed)[0]
    pool = [synth.make_frames(1, H, W, seed=4000 + s)[0] for s in range(8)]
    out, keys, nxt = [], [], 0
    for g in range(n_groups):
        at = (g * (G - 1)) // max(1, n_groups - 1) if G > 1 else (0 if g == 0 else -1)
        for j in range(G):
            if j == at:
                out.append(fix); keys.append((H, W, "fixture"))
            else:
                out.append(pool[nxt % 8]); keys.append((H, W, 4000 + nxt % 8)); nxt += 1
    return np.stack(out), keys


def compare_record(got, exp, what):
    """got / exp: [2,750,5] Detect records of one frame."""
    assert not got[0].any(), what                                  # class-0 plane is zeros (detection.py:48)
    n_exp = int((exp[1, :, 0] > 0).sum())
    n_got = int((got[1, :, 0] > 0).sum())
    assert n_got == n_exp and n_exp > 0, (what, n_got, n_exp)
    iou = opp.calculate_iou(exp[1, :n_exp, 1:].astype(np.float64), got[1, :n_exp, 1:].astype(np.float64))
    j = iou.argmax(1)
    assert len(set(j.tolist())) == n_exp, what
    d_iou = float((1 - iou[np.arange(n_exp), j]).max())
    d_sc = float(np.abs(got[1, j, 0] - exp[1, :n_exp, 0]).max())
    assert d_iou <= BOX_IOU_TOL and d_sc <= SCORE_ATOL, (what, d_iou, d_sc)
    assert np.abs(got[1, :n_exp, 0] - exp[1, :n_exp, 0]).max() <= SCORE_ATOL, what     # rows move only among near-equal scores
    return d_iou, d_sc


def tracks_key(tracks):
    return [(t["start_frame"], float(t["max_score"]), [list(map(float, b)) for b in t["bboxes"]]) for t in tracks]


@pytest.mark.gpu
@pytest.mark.parametrize("plan_file,fix_key,G,NF", TIMED, ids=[t[0][:-5] for t in TIMED])
def test_timed_plan_through_the_cabi_pipeline_vs_oracle(res50_sd, synth, oracle_threads, plan_file, fix_key, G, NF):
    d, meta = load_npz("nets")
    m = meta[fix_key]
    H, W = m["H"], m["W"]
    assert (m["conf_t"], m["nms_t"]) == (0.3, 0.5)                  # the reference's default Detect (pyramid.py:198)
    plan_text = open(os.path.join(TUNED, plan_file)).read()
    assert plan_text.split()[:4] == ["shape", str(G), str(H), str(W)], plan_text[:40]
    n_groups = {1: 8, 2: 4, 4: 4, 8: 2}[G]                          # 8 / 8 / 16 / 16 frames; every group keeps its slot
    assert n_groups <= NF
    frames, keys = frames_for(synth, H, W, m["frame_seed"], G, n_groups)
    N = len(frames)
    dev = torch.device("cuda", 0)
    fd = torch.from_numpy(frames).to(dev)
    torch.cuda.synchronize()

    net = M("pyramid").build_sfd('test', 640, 2)
    net.load_state_dict(res50_sd)
    net.priorbox = M("layers").PriorBoxLayer(W, H)
    T_MIN = 3                                                       # short sequences: tracks of >= 3 frames count (both sides)
    pipe = M("pipeline").CabiPipeline(net, H, W, 0, inflight=NF, batch=G, plan_text=plan_text, log_frames=64, t_min=T_MIN)
    pipe.prime(fd[0:G])
    if G > 1:
        for i in range(N - 1):                                      # the last group is launched partly filled by finish()
            pipe.step_frame(i, fd[i:i + 1])
        pipe.flush()
        n_seen = N - 1
    else:
        for i in range(N):
            pipe.step(i, fd[i:i + 1])
        n_seen = N
    # every frame's record, as the tracker stream consumed it (slot k holds group g = k while N <= NF * G)
    recs = []
    if G > 1:
        for g in range(n_groups):
            r = pipe.record_of_slot(g % NF)
            recs.extend(r[j] for j in range(G))
    else:
        assert N == NF
        recs = [pipe.record_of_slot(i)[0] for i in range(N)]
    # the handle really runs the imported plan at this batch
    exported = net.export_plan()
    assert exported.split()[:4] == ["shape", str(G), str(H), str(W)]
    plan_rows = [ln.split() for ln in plan_text.strip().splitlines()[1:]]
    ran = {ln.split()[0]: ln.split() for ln in exported.strip().splitlines()[1:]}
    # (kernel class, tile, split-K, workgroup map) of every layer as committed; the stem may run as its raw-uint8 class
    # (csrc/conv_stem_u8.h: same MFMA sequence, the frame's mean subtraction inside the staging)
    differ = [r[0] for r in plan_rows if ran[r[0]][:len(r)] != r]
    assert set(differ) <= {"conv1"}, "the forward did not run the committed plan: %s" % differ
    got_tracks = pipe.finish()
    pipe.close()
    net.close()

    worst = (0.0, 0.0)
    ref_gpu = opp.IouTracker(0.4, 0.6, T_MIN)
    ref_cpu = opp.IouTracker(0.4, 0.6, T_MIN)
    for i in range(n_seen):
        exp = oracle_record(res50_sd, frames[i], keys[i])[0]
        di, ds = compare_record(recs[i], exp, "%s frame %d (%s)" % (plan_file, i, keys[i][2]))
        worst = (max(worst[0], di), max(worst[1], ds))
        if keys[i][2] == "fixture":                                 # ... and against what the reference itself produced
            ref_out = d[fix_key + "_out"]
            ref_rec = np.zeros((2, 750, 5), np.float32)
            ref_rec[1, :ref_out.shape[0]] = ref_out
            compare_record(recs[i], ref_rec, "%s frame %d vs reference fixture" % (plan_file, i))
        with np.errstate(all="ignore"):
            ref_gpu.step(opp.unpack_detections(recs[i][None], W, H, 0.4))
            ref_cpu.step(opp.unpack_detections(exp[None], W, H, 0.4))
    want = ref_gpu.finish()
    assert tracks_key(got_tracks) == tracks_key(want) and len(want) >= 1        # bit-equal given identical boxes
    cpu_tracks = ref_cpu.finish()
    assert len(cpu_tracks) == len(want)
    for a, b in zip(want, cpu_tracks):                              # same tracks on the oracle's own detections
        assert a["start_frame"] == b["start_frame"] and len(a["bboxes"]) == len(b["bboxes"])
        assert abs(a["max_score"] - b["max_score"]) <= SCORE_ATOL
        assert np.abs(np.array(a["bboxes"]) - np.array(b["bboxes"])).max() <= 1e-3 * max(H, W)
    print("%s: %d frames, max IoU deficit %.2e, max score diff %.2e, %d tracks" % (plan_file, n_seen, worst[0], worst[1], len(want)))


@pytest.mark.gpu
def test_try3_batch1_plan_vs_reference_fixture(try3_sd, synth):
    """tuned/try3_1024x1024_b1.plan (bench.py --arch try3 --batch 1) against the reference's own output for the seeded frame."""
    d, meta = load_npz("nets")
    key = "try3_1024x1024"
    m = meta[key]
    net = M("pyramid_mb2_try3").build_sfd_mobile('test', 640, 2)
    net.load_state_dict(try3_sd)
    net.priorbox = M("layers").PriorBoxLayer(1024, 1024, stride=[4, 8, 16, 32, 64], box=(16, 32, 64, 128, 256))
    net.detect = M("layers").Detect(2, 0, 750, m["conf_t"], m["nms_t"])
    plan = net.tuned_plan_text(1024, 1024, 1)
    assert plan is not None
    net.import_plan(plan)
    frame = synth.make_frames(1, 1024, 1024, seed=m["frame_seed"])[0]
    y = net(frame).numpy()
    assert net.export_plan().split()[:4] == ["shape", "1", "1024", "1024"]
    exp = np.zeros((2, 750, 5), np.float32)
    exp[1, :d[key + "_out"].shape[0]] = d[key + "_out"]
    compare_record(y[0], exp, "try3_1024x1024_b1.plan vs reference fixture")
    assert np.array_equal(net(frame).numpy(), y)                    # graph replay: same bits
    net.close()
