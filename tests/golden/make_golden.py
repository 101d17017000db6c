#!/usr/bin/env python3
"""Generate the committed golden fixtures by running the REFERENCE ITSELF.

Run once in the build container (needs /root/reference, which never travels):

    python tests/golden/make_golden.py [--only NAME]

It imports the unmodified reference through the harness shims in `_refshim.py`,
feeds it seeded inputs / the seeded synthetic state-dict from
`face-detection-and-tracking_amd/synth.py`, and stores inputs + the reference's outputs
(data only) under tests/golden/.  tests/test_oracle_golden.py then pins `oracle/`
against these files, and the GPU parity tests compare the HIP path with the oracle
and with these files.

The inline tracker (reference iouTracke_cal.py:126-156,174-175) is not a function, so
its source lines are read from the reference file at generation time and exec'd
here -- the reference's own statements run, nothing is copied into the repo.
"""
import argparse
import hashlib
import importlib
import json
import os
import sys
import textwrap

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import _refshim  # noqa: E402

_refshim.install()
import torch  # noqa: E402

synth = importlib.import_module("face-detection-and-tracking_amd.synth")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, os.path.getsize(path), "bytes")


# ----------------------------------------------------------------------------- priors
def gen_priors():
    from layers import PriorBoxLayer
    out = {}
    meta = {}
    cases = [("res50", 640, 640), ("res50", 1024, 1024), ("res50", 640, 480), ("res50", 200, 136),
             ("res50", 1920, 1080), ("try3", 1024, 1024), ("try3", 640, 480), ("try3", 200, 136)]
    from oracle.postproc import feature_sizes
    for arch, W, H in cases:
        if arch == "res50":
            pb = PriorBoxLayer(W, H)
        else:
            pb = PriorBoxLayer(W, H, stride=[4, 8, 16, 32, 64], box=(16, 32, 64, 128, 256))
        pri = torch.cat([pb(i, fw, fh) for i, (fh, fw) in enumerate(feature_sizes(H, W, arch))], 0).numpy()
        key = "%s_%dx%d" % (arch, W, H)
        meta[key] = {"shape": list(pri.shape), "sha256": sha(pri)}
        if pri.shape[0] < 5000:
            out[key] = pri
        else:
            out[key + "_head"] = pri[:64]
            out[key + "_tail"] = pri[-64:]
    # a config with scales>1 and aspect ratios (generic PriorBoxLayer signature)
    pb = PriorBoxLayer(320, 240, stride=(8, 16), box=(32, 64), scale=(2, 1),
                       aspect_ratios=([2.0], [0.5, 3.0]))
    out["generic_320x240"] = torch.cat([pb(0, 40, 30), pb(1, 20, 15)], 0).numpy()
    out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    save("priors", **out)


# ----------------------------------------------------------------------------- detect ops
def _rand_case(rng, n, n_clusters, score_lo=0.0, score_hi=1.0, size=0.08):
    """loc/conf/priors such that decoded boxes form overlapping clusters."""
    centers = rng.uniform(0.1, 0.9, (n_clusters, 2))
    which = rng.integers(0, n_clusters, n)
    priors = np.empty((n, 4), np.float32)
    priors[:, :2] = centers[which] + rng.normal(0, size / 4, (n, 2))
    priors[:, 2:] = rng.uniform(size * 0.7, size * 1.4, (n, 2))
    loc = rng.normal(0, 1.0, (n, 4)).astype(np.float32)
    s = rng.uniform(score_lo, score_hi, n).astype(np.float32)
    conf = np.stack([1 - s, s], 1).astype(np.float32)
    return loc, conf, priors


def gen_detect():
    from layers import Detect
    from layers.box_utils import decode, nms
    rng = np.random.default_rng(42)
    out = {}
    cases = {}

    def run(name, loc, conf, priors, top_k=750, conf_t=0.3, nms_t=0.5):
        det = Detect(2, 0, top_k, conf_t, nms_t)
        y = det(torch.from_numpy(loc[None]), torch.from_numpy(conf[None]), torch.from_numpy(priors))
        out[name + "_loc"] = loc
        out[name + "_conf"] = conf
        out[name + "_priors"] = priors
        out[name + "_out"] = y.numpy()
        cases[name] = {"top_k": top_k, "conf_t": conf_t, "nms_t": nms_t,
                       "n_out": int((y[0, 1, :, 0] > 0).sum())}

    run("rand300", *_rand_case(rng, 300, 12))
    run("rand2000_t035", *_rand_case(rng, 2000, 40), conf_t=0.2, nms_t=0.35)
    # zero / one / two candidates (quirk: exactly one candidate emits nothing)
    loc, conf, pri = _rand_case(rng, 50, 5, 0.0, 0.29)
    run("cand0", loc, conf, pri)
    c1 = conf.copy(); c1[17] = [0.1, 0.9]
    run("cand1", loc, c1, pri)
    c2 = c1.copy(); c2[3] = [0.2, 0.8]
    run("cand2", loc, c2, pri)
    # exact score ties (stable ascending sort => highest index first)
    loc, conf, pri = _rand_case(rng, 400, 10)
    conf[:, 1] = np.round(conf[:, 1] * 8) / 8
    conf[:, 0] = 1 - conf[:, 1]
    run("ties", loc, conf.astype(np.float32), pri)
    # IoU exactly == threshold is suppressed (keep only IoU < thr): two boxes, IoU = 0.5
    pri = np.array([[0.5, 0.5, 0.5, 0.5], [0.5, 0.375, 0.5, 0.25], [0.2, 0.2, 0.1, 0.1]], np.float32)
    loc = np.zeros((3, 4), np.float32)
    conf = np.array([[0.1, 0.9], [0.2, 0.8], [0.3, 0.7]], np.float32)
    run("iou_eq_thr", loc, conf, pri, nms_t=0.5)
    # zero-area boxes: IoU 0/0 = NaN is not < thr => suppressed
    pri = np.array([[0.5, 0.5, 0.0, 0.0], [0.5, 0.5, 0.0, 0.0], [0.3, 0.3, 0.1, 0.1]], np.float32)
    run("nan_iou", np.zeros((3, 4), np.float32), conf, pri)
    # more than nms_top_k=5000 candidates (only the 5000 best enter NMS)
    run("over5000", *_rand_case(rng, 6000, 300, 0.31, 1.0, size=0.03))
    # more than top_k=750 survivors: a grid of disjoint boxes
    g = 32
    yy, xx = np.meshgrid(np.arange(g), np.arange(g), indexing="ij")
    pri = np.stack([(xx.ravel() + 0.5) / g, (yy.ravel() + 0.5) / g,
                    np.full(g * g, 0.5 / g), np.full(g * g, 0.5 / g)], 1).astype(np.float32)
    loc = np.zeros((g * g, 4), np.float32)
    s = rng.uniform(0.35, 1.0, g * g).astype(np.float32)
    run("over750", loc, np.stack([1 - s, s], 1).astype(np.float32), pri)
    # batch of 2 images
    l1, c1_, p1 = _rand_case(rng, 500, 20)
    l2, c2_, _ = _rand_case(rng, 500, 20)
    det = Detect(2, 0, 750, 0.3, 0.5)
    y = det(torch.from_numpy(np.stack([l1, l2])), torch.from_numpy(np.stack([c1_, c2_])), torch.from_numpy(p1))
    out["batch2_loc"], out["batch2_conf"], out["batch2_priors"] = np.stack([l1, l2]), np.stack([c1_, c2_]), p1
    out["batch2_out"] = y.numpy()
    cases["batch2"] = {"top_k": 750, "conf_t": 0.3, "nms_t": 0.5}

    # standalone decode and nms
    loc, conf, pri = _rand_case(rng, 1000, 25)
    boxes = decode(torch.from_numpy(loc), torch.from_numpy(pri), [0.1, 0.2])
    out["decode_loc"], out["decode_priors"], out["decode_out"] = loc, pri, boxes.numpy()
    for nm, thr, tk in (("nms_a", 0.5, 200), ("nms_b", 0.35, 1000), ("nms_c", 0.3, 50)):
        keep, count = nms(boxes, torch.from_numpy(conf[:, 1].copy()), thr, tk)
        out[nm + "_keep"] = keep.numpy()
        cases[nm] = {"overlap": thr, "top_k": tk, "count": int(count)}
    out["nms_boxes"], out["nms_scores"] = boxes.numpy(), conf[:, 1].copy()
    try:
        Detect(2, 0, 750, 0.3, 0.0)
        cases["nms_thresh_zero_raises"] = False
    except ValueError:
        cases["nms_thresh_zero_raises"] = True
    out["meta_json"] = np.frombuffer(json.dumps(cases).encode(), dtype=np.uint8)
    save("detect_ops", **out)


# ----------------------------------------------------------------------------- iou / pr
def gen_iou():
    from utils.calc_performance import calculate_iou, calc_pr
    rng = np.random.default_rng(5)
    out = {}

    def boxes(n, dtype, scale=640.0):
        xy = rng.uniform(0, scale * 0.8, (n, 2))
        wh = rng.uniform(4, scale * 0.3, (n, 2))
        return np.concatenate([xy, xy + wh], 1).astype(dtype)
    for dt, nm in ((np.float64, "f64"), (np.float32, "f32")):
        a, b = boxes(37, dt), boxes(53, dt)
        a[5] = b[7]                        # identical box -> IoU 1
        a[6] = [10, 10, 10, 10]            # zero area
        b[9] = [10, 10, 10, 10]            # zero area vs zero area -> 0/0 = NaN
        b[10] = [0, 0, 0, 0]
        with np.errstate(all="ignore"):
            out["iou_%s_a" % nm], out["iou_%s_b" % nm] = a, b
            out["iou_%s_out" % nm] = calculate_iou(a, b)
    a = boxes(700, np.float64)
    b = boxes(900, np.float64)
    out["iou_big_a"], out["iou_big_b"] = a, b
    out["iou_big_sha"] = np.frombuffer(sha(calculate_iou(a, b)).encode(), dtype=np.uint8)
    pred = np.concatenate([boxes(40, np.float64), rng.uniform(0, 1, (40, 1))], 1)
    truth = boxes(12, np.float64)
    truth[:, 2:] -= truth[:, :2]           # x,y,w,h
    pred[:6, :4] = np.hstack((truth[:6, :2], truth[:6, 2:] + truth[:6, :2])) + rng.normal(0, 2, (6, 4))
    tf, tn = calc_pr(pred, truth, 0.5)
    out["pr_pred"], out["pr_truth"], out["pr_out"], out["pr_truth_num"] = pred, truth, tf, np.array(tn)
    save("iou", **out)


# ----------------------------------------------------------------------------- unpack + tracker
def _tracker_sources():
    src = open(os.path.join(_refshim.REFERENCE, "iouTracke_cal.py")).read().splitlines()
    body = textwrap.dedent("\n".join(src[126:155]))       # :127-155  dets=... tracks_active=...
    final = textwrap.dedent("\n".join(src[173:175]))      # :174-175
    assert body.lstrip().startswith("dets = det0.tolist()"), body[:80]
    assert "tracks_active = updated_tracks + new_tracks" in body
    assert final.lstrip().startswith("tracks_finished +="), final[:80]
    return compile(body, "ref_tracker_body", "exec"), compile(final, "ref_tracker_final", "exec")


def make_track_sequence(rng, n_frames, n_faces, w=640, h=480, p_drop=0.1, jitter=3.0, birth=0.05,
                        empty_frames=()):
    """Synthetic per-frame detections [n,5] f32 (x1,y1,x2,y2,score): random-walk faces."""
    faces = []

    def spawn():
        cx, cy = rng.uniform(60, w - 60), rng.uniform(60, h - 60)
        s = rng.uniform(30, 90)
        return [cx, cy, s, rng.uniform(0.45, 0.99)]
    for _ in range(n_faces):
        faces.append(spawn())
    frames = []
    for f in range(n_frames):
        dets = []
        for fc in faces:
            fc[0] += rng.normal(0, jitter); fc[1] += rng.normal(0, jitter)
            fc[2] *= np.exp(rng.normal(0, 0.02))
            if rng.uniform() < p_drop:
                continue
            sc = float(np.clip(fc[3] + rng.normal(0, 0.05), 0.4, 1.0))
            dets.append([fc[0] - fc[2] / 2, fc[1] - fc[2] / 2, fc[0] + fc[2] / 2, fc[1] + fc[2] / 2, sc])
        if rng.uniform() < birth:
            faces.append(spawn())
        if rng.uniform() < birth / 2 and len(faces) > 1:
            faces.pop(int(rng.integers(0, len(faces))))
        rng.shuffle(dets)
        if f in empty_frames or len(dets) == 0:
            frames.append(np.array([[0, 0, 0, 0, 0.4]]))        # the reference's dummy row
        else:
            frames.append(np.array(dets, dtype=np.float32))
    return frames


def gen_tracker():
    import iouTracke_cal as ref_cal
    from utils.calc_performance import calculate_iou, calculate_distance
    body, final = _tracker_sources()
    rng = np.random.default_rng(99)
    seqs = {
        "walk": make_track_sequence(rng, 60, 4),
        "crowd": make_track_sequence(rng, 40, 12, p_drop=0.2, jitter=6.0, birth=0.2),
        "gaps": make_track_sequence(rng, 50, 3, p_drop=0.3, empty_frames=(10, 11, 30)),
        # fewer dets than tracks: `dets` runs empty mid-loop and later tracks are dropped
        "exhaust": make_track_sequence(rng, 80, 6, p_drop=0.3, jitter=2.0),
    }
    result = {}
    for name, frames in seqs.items():
        ns = dict(np=np, calculate_iou=calculate_iou, calculate_distance=calculate_distance,
                  use_iou=True, sigma_iou=0.4, sigma_dis=8, sigma_h=0.6, t_min=5,
                  tracks_active=[], tracks_finished=[], frame_num=0)
        for det0 in frames:
            ns["frame_num"] += 1                           # :118
            ns["det0"] = det0
            with np.errstate(all="ignore"):
                exec(body, ns)
        exec(final, ns)
        result[name] = {
            "frames": [f.tolist() for f in frames],
            "frame_dtypes": [str(f.dtype) for f in frames],
            "tracks": [{"bboxes": [list(map(float, b)) for b in t["bboxes"]],
                        "max_score": float(t["max_score"]), "start_frame": int(t["start_frame"])}
                       for t in ns["tracks_finished"]],
        }
        print("tracker", name, "frames", len(frames), "tracks", len(result[name]["tracks"]))

    # unpack: the reference's own detect_face() with a stub net returning a crafted tensor
    unpack = {}
    rngu = np.random.default_rng(3)
    for nm, nrow in (("some", 9), ("none", 0), ("full", 750)):
        y = np.zeros((1, 2, 750, 5), np.float32)
        sc = np.sort(rngu.uniform(0.4, 1.0, nrow).astype(np.float32))[::-1]
        y[0, 1, :nrow, 0] = sc
        xy = rngu.uniform(0, 0.7, (nrow, 2))
        y[0, 1, :nrow, 1:3] = xy
        y[0, 1, :nrow, 3:5] = xy + rngu.uniform(0.02, 0.3, (nrow, 2))
        if nm == "some":
            y[0, 1, nrow:nrow + 3, 0] = [0.39, 0.3, 0.2]   # below the 0.4 walk threshold
            y[0, 1, nrow:nrow + 3, 1:] = 0.5
        ref_cal.net = lambda x, _y=y: torch.from_numpy(_y)
        img = np.zeros((480, 640, 3), np.uint8)
        det = ref_cal.detect_face(img, 1)
        unpack[nm] = {"y_rows": y[0, 1, :max(nrow + 3, 1)].tolist(), "nrow": nrow,
                      "det": np.asarray(det).tolist(), "dtype": str(np.asarray(det).dtype)}
    with open(os.path.join(HERE, "tracker.json"), "w") as f:
        json.dump({"sequences": result, "unpack": unpack}, f)
    print("wrote tracker.json", os.path.getsize(os.path.join(HERE, "tracker.json")))


# ----------------------------------------------------------------------------- nets
def _ref_net(arch, sd):
    if arch == "res50":
        from pyramid import build_sfd
        net = build_sfd("test", 640, 2)
    else:
        mod = {"try3": "pyramid_mb2_try3", "try4": "pyramid_mb2_try4", "try5": "pyramid_mb2_try5",
               "try1": "pyramid_mobile_try1", "try2": "pyramid_mobile_try2"}[arch]
        net = importlib.import_module(mod).build_sfd_mobile("test", 640, 2)
    ref_keys = list(net.state_dict().keys())
    assert ref_keys == list(sd.keys()), "synthetic schema != reference state_dict keys"
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    net.eval()
    return net


def gen_nets():
    from layers import PriorBoxLayer, Detect
    from oracle import pyramidbox as opb
    out, meta = {}, {}
    for arch in ("res50", "try3"):
        sd = synth.make_state_dict(arch, seed=0)
        net = _ref_net(arch, sd)
        # capture pre-Detect tensors of the reference through its own detect hook
        for (H, W, seed, conf_t, nms_t) in ((64, 64, 7, 0.02, 0.35), (136, 200, 8, 0.02, 0.35),
                                            (480, 640, 9, None, None), (1024, 1024, 1234, None, None)):
            if arch == "try3" and H == 1024:
                pass
            frame = synth.make_frames(1, H, W, seed=seed)[0]
            x = torch.from_numpy(opb.preprocess(frame))
            if arch == "res50":
                net.priorbox = PriorBoxLayer(W, H)
                dflt = (0.3, 0.5)
            else:
                net.priorbox = PriorBoxLayer(W, H, stride=[4, 8, 16, 32, 64], box=(16, 32, 64, 128, 256))
                dflt = (0.2, 0.35)
            ct, nt = (conf_t, nms_t) if conf_t is not None else dflt
            net.firstTime = True
            real = Detect(2, 0, 750, ct, nt)
            cap = {}

            def spy(loc, conf, priors, _real=real, _cap=cap):
                _cap["loc"], _cap["conf"] = loc.numpy().copy(), conf.numpy().copy()
                return _real(loc, conf, priors)
            net.detect = spy
            with torch.no_grad():
                y = net(x).numpy()
            key = "%s_%dx%d" % (arch, H, W)
            n_out = int((y[0, 1, :, 0] > 0).sum())
            n_cand = int((cap["conf"][0, :, 1] > np.float32(ct)).sum())
            meta[key] = {"H": H, "W": W, "frame_seed": seed, "conf_t": ct, "nms_t": nt, "n_out": n_out,
                         "n_cand": n_cand, "P": int(cap["loc"].shape[1]),
                         "loc_sha": sha(cap["loc"]), "conf_sha": sha(cap["conf"])}
            out[key + "_out"] = y[0, 1, :max(n_out, 1)]
            if cap["loc"].shape[1] <= 3000:
                out[key + "_loc"], out[key + "_conf"] = cap["loc"], cap["conf"]
            else:
                sel = np.linspace(0, cap["loc"].shape[1] - 1, 1024).astype(np.int64)
                out[key + "_sel"] = sel
                out[key + "_loc_s"], out[key + "_conf_s"] = cap["loc"][0, sel], cap["conf"][0, sel]
            print(key, meta[key])
    out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    save("nets", **out)


def gen_nets45():
    """try4 / try5: the reference modules pyramid_mb2_try4.py / pyramid_mb2_try5.py run on
    seeded weights and frames; stored: the pre-Detect loc/conf (whole for small inputs, a strided sample at
    640x480), the Detect rows, the source sizes and three intermediate maps at the smallest size."""
    from layers import PriorBoxLayer, Detect
    from oracle import pyramidbox as opb
    out, meta = {}, {}
    for arch in ("try4", "try5"):
        sd = synth.make_state_dict(arch, seed=0)
        net = _ref_net(arch, sd)
        for (H, W, seed, ct, nt) in ((64, 64, 17, 0.02, 0.35), (136, 200, 18, 0.02, 0.35), (480, 640, 19, 0.2, 0.35)):
            frame = synth.make_frames(1, H, W, seed=seed)[0]
            x = torch.from_numpy(opb.preprocess(frame))
            net.priorbox = PriorBoxLayer(W, H, stride=[4, 8, 16, 32, 64], box=(16, 32, 64, 128, 256))
            net.firstTime = True
            real = Detect(2, 0, 750, ct, nt)
            cap = {}

            def spy(loc, conf, priors, _real=real, _cap=cap):
                _cap["loc"], _cap["conf"], _cap["priors"] = loc.numpy().copy(), conf.numpy().copy(), priors.numpy().copy()
                return _real(loc, conf, priors)
            net.detect = spy
            with torch.no_grad():
                y = net(x).numpy()
            key = "%s_%dx%d" % (arch, H, W)
            n_out = int((y[0, 1, :, 0] > 0).sum())
            meta[key] = {"H": H, "W": W, "frame_seed": seed, "conf_t": ct, "nms_t": nt, "n_out": n_out,
                         "n_cand": int((cap["conf"][0, :, 1] > np.float32(ct)).sum()), "P": int(cap["loc"].shape[1]),
                         "loc_sha": sha(cap["loc"]), "conf_sha": sha(cap["conf"]), "priors_sha": sha(cap["priors"])}
            out[key + "_out"] = y[0, 1, :max(n_out, 1)]
            if cap["loc"].shape[1] <= 3000:
                out[key + "_loc"], out[key + "_conf"], out[key + "_priors"] = cap["loc"], cap["conf"], cap["priors"]
            else:
                sel = np.linspace(0, cap["loc"].shape[1] - 1, 1024).astype(np.int64)
                out[key + "_sel"] = sel
                out[key + "_loc_s"], out[key + "_conf_s"] = cap["loc"][0, sel], cap["conf"][0, sel]
                out[key + "_priors_s"] = cap["priors"][sel]
            print(key, meta[key])
    out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    save("nets45", **out)


def gen_nets12():
    """try1 / try2: reference pyramid_mobile_try1.py / _try2.py on seeded weights and frames."""
    from layers import PriorBoxLayer, Detect
    from oracle import pyramidbox as opb
    out, meta = {}, {}
    for arch in ("try1", "try2"):
        sd = synth.make_state_dict(arch, seed=0)
        net = _ref_net(arch, sd)
        nms_d = 0.3 if arch == "try1" else 0.5
        for (H, W, seed, ct, nt) in ((64, 64, 37, 0.02, 0.35), (136, 200, 38, 0.02, 0.35), (480, 640, 39, 0.3, nms_d)):
            frame = synth.make_frames(1, H, W, seed=seed)[0]
            x = torch.from_numpy(opb.preprocess(frame))
            net.priorbox = PriorBoxLayer(W, H)
            net.firstTime = True
            real = Detect(2, 0, 750, ct, nt)
            cap = {}

            def spy(loc, conf, priors, _real=real, _cap=cap):
                _cap["loc"], _cap["conf"], _cap["priors"] = loc.numpy().copy(), conf.numpy().copy(), priors.numpy().copy()
                return _real(loc, conf, priors)
            net.detect = spy
            with torch.no_grad():
                y = net(x).numpy()
            key = "%s_%dx%d" % (arch, H, W)
            n_out = int((y[0, 1, :, 0] > 0).sum())
            meta[key] = {"H": H, "W": W, "frame_seed": seed, "conf_t": ct, "nms_t": nt, "n_out": n_out,
                         "n_cand": int((cap["conf"][0, :, 1] > np.float32(ct)).sum()), "P": int(cap["loc"].shape[1]),
                         "loc_sha": sha(cap["loc"]), "conf_sha": sha(cap["conf"]), "priors_sha": sha(cap["priors"])}
            out[key + "_out"] = y[0, 1, :max(n_out, 1)]
            if cap["loc"].shape[1] <= 3000:
                out[key + "_loc"], out[key + "_conf"] = cap["loc"], cap["conf"]
            else:
                sel = np.linspace(0, cap["loc"].shape[1] - 1, 1024).astype(np.int64)
                out[key + "_sel"] = sel
                out[key + "_loc_s"], out[key + "_conf_s"] = cap["loc"][0, sel], cap["conf"][0, sel]
            print(key, meta[key], "conf range", float(cap["conf"][0, :, 1].min()), float(cap["conf"][0, :, 1].max()))
    out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    save("nets12", **out)


# ----------------------------------------------------------------------------- FaceBoxes (real weights)
def gen_facebox():
    """Reference FaceBox + DataEncoder with the weights that ship in the reference tree
    (FACEBOX/faceboxes.pt, loaded with weights_only=True).  The weights are saved as plain arrays
    (tests/golden/faceboxes_weights.npz) so the GPU box can rebuild the same model; inputs are two of
    the reference's sample JPEGs (PIL-decoded, resized to 1024x1024, RGB->BGR; stored post-resize since
    cv2.resize is unavailable) and one seeded noise frame."""
    import torch.nn.functional as F
    from PIL import Image
    from FACEBOX.networks import FaceBox
    from FACEBOX.encoderl import DataEncoder
    sd = torch.load(os.path.join(_refshim.REFERENCE, "FACEBOX", "faceboxes.pt"), map_location="cpu",
                    weights_only=True)
    net = FaceBox()
    net.load_state_dict(sd)
    net.eval()
    np.savez_compressed(os.path.join(HERE, "faceboxes_weights.npz"), **{k: v.numpy() for k, v in sd.items()})
    enc = DataEncoder()
    out = {"anchors_sha": np.frombuffer(sha(enc.default_boxes_np).encode(), dtype=np.uint8),
           "anchors_head": enc.default_boxes_np[:64], "anchors_tail": enc.default_boxes_np[-400:]}
    imgdir = os.path.join(_refshim.REFERENCE, "image_and_anno", "test_image", "try1")
    names = sorted(os.listdir(imgdir))
    frames = []
    for nm in (names[0], names[7]):
        im = Image.open(os.path.join(imgdir, nm)).convert("RGB").resize((1024, 1024), Image.BILINEAR)
        frames.append(np.ascontiguousarray(np.asarray(im)[:, :, ::-1]))
    frames.append(synth.make_frames(1, 1024, 1024, seed=55)[0])
    meta = {}
    for i, fr in enumerate(frames):
        x = torch.from_numpy(fr.transpose((2, 0, 1)).copy()).float().div(255)          # My_test_facebox.py:14-15
        with torch.no_grad():
            loc, conf = net(x[None])
        loc = loc.squeeze(0)
        confs = F.softmax(conf.squeeze(0), dim=1)
        boxes, probs = enc.decode_np(loc, confs)
        sel = np.linspace(0, 21823, 2048).astype(np.int64)
        key = "img%d" % i
        if i < 2:
            out[key + "_frame"] = fr
        out[key + "_sel"] = sel
        out[key + "_loc_s"], out[key + "_conf_s"] = loc.numpy()[sel], conf.squeeze(0).numpy()[sel]
        out[key + "_boxes"], out[key + "_probs"] = boxes, probs
        meta[key] = {"n": int(len(probs)), "n_cand": int((confs[:, 1] > 0.35).sum()),
                     "top": float(probs.max()) if len(probs) else 0.0, "noise_seed": 55 if i == 2 else None}
        print("facebox", key, meta[key])
    # decode_np / nms_np on crafted predictions: ~600 candidates on clustered anchors, distinct scores
    rng = np.random.default_rng(8)
    loc = rng.normal(0, 1.0, (21824, 4)).astype(np.float32)
    sc = np.zeros(21824, np.float32)
    hot = rng.choice(21824, 600, replace=False)
    hot[:300] = (rng.integers(0, 40, 300) * 21 * 7 + rng.integers(0, 21, 300)) % 21824   # crowded cells
    sc[hot] = rng.permutation(np.linspace(0.36, 0.999, hot.size)).astype(np.float32)
    conf = np.stack([1 - sc, sc], 1).astype(np.float32)
    boxes, probs = enc.decode_np(torch.from_numpy(loc), torch.from_numpy(conf))
    out["dec_loc"], out["dec_conf"], out["dec_boxes"], out["dec_probs"] = loc, conf, boxes, probs
    meta["dec"] = {"n": int(len(probs)), "n_cand": int((sc > 0.35).sum())}
    print("facebox dec", meta["dec"])
    out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    save("facebox", **out)


# ----------------------------------------------------------------------------- round 2: configs VERDICT r01 found untested
def gen_nets_r2():
    """C1 (Res50 640x640, My_test.py's size) and C3 (try3 1024x1024 BATCH 8: eight seeded frames through ONE call of the
    reference module, pyramid_mb2_try3.py:218-340) -> tests/golden/nets_r2.npz.  Stored per image: the Detect rows, a
    strided sample of the pre-Detect loc / conf, and their sha256."""
    _gen_net_cases("nets_r2", (("res50", 640, 640, [640]), ("try3", 1024, 1024, [2000 + i for i in range(8)])))


def gen_nets_r3():
    """C4' (Res50 at the NATIVE 1080 x 1920 frame size of BASELINE config 4's source video: one forward of the reference
    module, pyramid.py:218-351) -> tests/golden/nets_r3.npz, same layout as nets_r2."""
    _gen_net_cases("nets_r3", (("res50", 1080, 1920, [1080]),))


def _gen_net_cases(fname, cases):
    from layers import PriorBoxLayer, Detect
    from oracle import pyramidbox as opb
    out, meta = {}, {}
    for arch, H, W, seeds in cases:
        sd = synth.make_state_dict(arch, seed=0)
        net = _ref_net(arch, sd)
        frames = np.stack([synth.make_frames(1, H, W, seed=sv)[0] for sv in seeds])
        x = torch.from_numpy(np.stack([opb.preprocess(f)[0] for f in frames]))
        if arch == "res50":
            net.priorbox = PriorBoxLayer(W, H)
            ct, nt = 0.3, 0.5
        else:
            net.priorbox = PriorBoxLayer(W, H, stride=[4, 8, 16, 32, 64], box=(16, 32, 64, 128, 256))
            ct, nt = 0.2, 0.35
        net.firstTime = True
        real = Detect(2, 0, 750, ct, nt)
        cap = {}

        def spy(loc, conf, priors, _real=real, _cap=cap):
            _cap["loc"], _cap["conf"] = loc.numpy().copy(), conf.numpy().copy()
            return _real(loc, conf, priors)
        net.detect = spy
        with torch.no_grad():
            y = net(x).numpy()
        key = "%s_%dx%d_b%d" % (arch, H, W, len(seeds))
        P = int(cap["loc"].shape[1])
        sel = np.linspace(0, P - 1, 512).astype(np.int64)
        out[key + "_sel"] = sel
        n_outs = []
        for b in range(len(seeds)):
            n_out = int((y[b, 1, :, 0] > 0).sum())
            n_outs.append(n_out)
            out["%s_out%d" % (key, b)] = y[b, 1, :max(n_out, 1)]
            out["%s_loc_s%d" % (key, b)] = cap["loc"][b, sel]
            out["%s_conf_s%d" % (key, b)] = cap["conf"][b, sel]
        assert not y[:, 0].any()
        meta[key] = {"H": H, "W": W, "frame_seeds": seeds, "conf_t": ct, "nms_t": nt, "n_out": n_outs, "P": P,
                     "n_cand": [int((cap["conf"][b, :, 1] > np.float32(ct)).sum()) for b in range(len(seeds))],
                     "loc_sha": sha(cap["loc"]), "conf_sha": sha(cap["conf"])}
        print(key, meta[key])
    out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    save(fname, **out)


def gen_facebox_r2():
    """FaceBoxes on SIX more of the reference's sample JPEGs (image_and_anno/test_image/try1), chosen for multi-face
    outputs, through the reference's own FaceBox + DataEncoder.decode_np with the in-tree weights
    (FACEBOX/My_test_facebox.py:12-36 after the resize).  Frames are stored post-resize (PIL bilinear; cv2 is absent)."""
    import torch.nn.functional as F
    from PIL import Image
    from FACEBOX.networks import FaceBox
    from FACEBOX.encoderl import DataEncoder
    sd = torch.load(os.path.join(_refshim.REFERENCE, "FACEBOX", "faceboxes.pt"), map_location="cpu", weights_only=True)
    net = FaceBox()
    net.load_state_dict(sd)
    net.eval()
    enc = DataEncoder()
    imgdir = os.path.join(_refshim.REFERENCE, "image_and_anno", "test_image", "try1")
    names = sorted(os.listdir(imgdir))
    res = []
    for nm in names:
        im = Image.open(os.path.join(imgdir, nm)).convert("RGB").resize((1024, 1024), Image.BILINEAR)
        fr = np.ascontiguousarray(np.asarray(im)[:, :, ::-1])
        x = torch.from_numpy(fr.transpose((2, 0, 1)).copy()).float().div(255)
        with torch.no_grad():
            loc, conf = net(x[None])
        confs = F.softmax(conf.squeeze(0), dim=1)
        boxes, probs = enc.decode_np(loc.squeeze(0), confs)
        res.append((len(probs), nm, fr, boxes, probs, int((confs[:, 1] > 0.35).sum())))
        print("facebox_r2", nm, len(probs), "faces,", res[-1][5], "candidates")
    res.sort(key=lambda r: (-r[0], r[1]))
    out, meta = {}, {}
    for i, (n, nm, fr, boxes, probs, ncand) in enumerate(res[:6]):
        key = "img%d" % i
        out[key + "_frame"], out[key + "_boxes"], out[key + "_probs"] = fr, boxes, probs
        meta[key] = {"n": int(n), "n_cand": ncand, "file": nm, "top": float(probs.max()),
                     "min_gap": float(np.min(np.abs(np.diff(np.sort(probs))))) if n > 1 else None}
    print("facebox_r2 kept", {k: (v["n"], v["file"]) for k, v in meta.items()})
    out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    save("facebox_r2", **out)


# ----------------------------------------------------------------------------- use_iou = False
def gen_distance():
    """calculate_distance (reference utils/calc_performance.py:34-51) on seeded boxes, and the inline tracker run with
    `use_iou = False` (iouTracke_cal.py:136-138) on sequences of the tracker fixture's generator."""
    from utils.calc_performance import calculate_iou, calculate_distance
    rng = np.random.default_rng(11)
    out = {}

    def boxes(n, dtype, scale=640.0):
        xy = rng.uniform(0, scale * 0.8, (n, 2))
        wh = rng.uniform(4, scale * 0.3, (n, 2))
        return np.concatenate([xy, xy + wh], 1).astype(dtype)
    for dt, nm in ((np.float64, "f64"), (np.float32, "f32")):
        a, b = boxes(41, dt), boxes(67, dt)
        a[5] = b[7]                        # identical box -> distance 0
        a[6] = [10, 10, 10, 10]
        b[9] = [10, 10, 10, 10]            # both degenerate at the same point -> 0
        b[10] = [0, 0, 0, 0]
        out["dis_%s_a" % nm], out["dis_%s_b" % nm] = a, b
        out["dis_%s_out" % nm] = calculate_distance(a, b)
    save("distance", **out)

    body, final = _tracker_sources()
    rngt = np.random.default_rng(123)
    seqs = {
        "walk": make_track_sequence(rngt, 60, 4),
        "crowd": make_track_sequence(rngt, 40, 12, p_drop=0.2, jitter=6.0, birth=0.2),
        "gaps": make_track_sequence(rngt, 50, 3, p_drop=0.3, empty_frames=(10, 11, 30)),
        "fast": make_track_sequence(rngt, 60, 5, p_drop=0.1, jitter=5.0),   # steps around sigma_dis = 8
    }
    result = {}
    for name, frames in seqs.items():
        ns = dict(np=np, calculate_iou=calculate_iou, calculate_distance=calculate_distance,
                  use_iou=False, sigma_iou=0.4, sigma_dis=8, sigma_h=0.6, t_min=5,
                  tracks_active=[], tracks_finished=[], frame_num=0)
        for det0 in frames:
            ns["frame_num"] += 1
            ns["det0"] = det0
            exec(body, ns)
        exec(final, ns)
        result[name] = {
            "frames": [f.tolist() for f in frames],
            "frame_dtypes": [str(f.dtype) for f in frames],
            "tracks": [{"bboxes": [list(map(float, b)) for b in t["bboxes"]],
                        "max_score": float(t["max_score"]), "start_frame": int(t["start_frame"])}
                       for t in ns["tracks_finished"]],
        }
        print("tracker(use_iou=False)", name, "frames", len(frames), "tracks", len(result[name]["tracks"]))
    with open(os.path.join(HERE, "tracker_dis.json"), "w") as f:
        json.dump({"sequences": result}, f)
    print("wrote tracker_dis.json", os.path.getsize(os.path.join(HERE, "tracker_dis.json")))


def gen_tp_fp_fixture():
    """gen_tp_fp (reference draw_curve/draw_pr_roc.py:5-19).  The module runs its plotting script on import, so -- as for
    the inline tracker -- the function's own lines are read from the reference file at generation time and exec'd."""
    src = open(os.path.join(_refshim.REFERENCE, "draw_curve", "draw_pr_roc.py")).read().splitlines()
    body = "\n".join(src[4:19])
    assert body.startswith("def gen_tp_fp(tf_conf):") and "return true_pos, false_pos" in body
    ns = {"np": np}
    exec(compile(body, "ref_gen_tp_fp", "exec"), ns)
    rng = np.random.default_rng(77)
    out = {}
    for nm, M, p in (("a", 257, 0.6), ("b", 40, 0.1), ("c", 1, 1.0), ("d", 1500, 0.35)):
        flags = (rng.uniform(size=M) < p).astype(np.float64)
        flags[rng.integers(0, M)] *= 2.0                      # any non-zero counts (np.count_nonzero)
        conf = np.sort(rng.uniform(size=M))[::-1]
        tf = np.stack([flags, conf])
        tp, fp = ns["gen_tp_fp"](tf)
        out["tf_" + nm], out["tp_" + nm], out["fp_" + nm] = tf, tp, fp
    tp, fp = ns["gen_tp_fp"](np.zeros((2, 0)))
    out["tf_empty"], out["tp_empty"], out["fp_empty"] = np.zeros((2, 0)), tp, fp
    save("tp_fp", **out)


def gen_tracker_r5():
    """Round 5: sequences that force the device tracker's EXACT association form (csrc/tracker.hip falls back to it per frame),
    expected tracks from the reference's own inline tracker lines (iouTracke_cal.py:127-155,174-175) as in gen_tracker():
      stack      -- every face is reported 8..10 times per frame (near-identical boxes, as an un-suppressed detector would):
                    each track has more than six detections above sigma_iou, frames of plain walking in between;
      zero_mid   -- (0,0,0,0) rows among real detections on consecutive frames: a zero-box track meets a zero-box detection,
                    0/0 = NaN IoU (utils/calc_performance.py:54-74 has no epsilon), numpy's arg-max returns the NaN;
      neg_sigma  -- the loop with sigma_iou = -0.5 (every remaining detection matches, also the zero-IoU ones)."""
    from utils.calc_performance import calculate_iou, calculate_distance
    body, final = _tracker_sources()
    rng = np.random.default_rng(505)

    def stack_seq(n_frames, n_faces, w=640, h=480):
        faces = [[rng.uniform(80, w - 80), rng.uniform(80, h - 80), rng.uniform(40, 90), rng.uniform(0.5, 0.99)]
                 for _ in range(n_faces)]
        frames = []
        for f in range(n_frames):
            dets = []
            dup = 1 if f % 5 == 4 else int(rng.integers(8, 11))      # every fifth frame: one detection per face
            for fc in faces:
                fc[0] += rng.normal(0, 2.0); fc[1] += rng.normal(0, 2.0)
                for _ in range(dup):
                    jx, jy, js = rng.normal(0, 0.6, 3)
                    sz = fc[2] * (1 + 0.01 * js)
                    sc = float(np.clip(fc[3] + rng.normal(0, 0.03), 0.4, 1.0))
                    dets.append([fc[0] + jx - sz / 2, fc[1] + jy - sz / 2, fc[0] + jx + sz / 2, fc[1] + jy + sz / 2, sc])
            rng.shuffle(dets)
            frames.append(np.array(dets, dtype=np.float32))
        return frames

    def zero_mid_seq():
        frames = make_track_sequence(rng, 40, 5, p_drop=0.1)
        for f in (6, 7, 8, 20, 21):                                   # a zero box among the real ones, consecutive frames
            fr = frames[f]
            if fr.shape[0] and fr[0, 2] > 0:
                at = int(rng.integers(0, fr.shape[0] + 1))
                frames[f] = np.insert(fr, at, np.array([0, 0, 0, 0, 0.55], fr.dtype), axis=0)
        frames[30] = np.array([[0, 0, 0, 0, 0.4]])                    # the dummy row (:73-74) twice in a row as well
        frames[31] = np.array([[0, 0, 0, 0, 0.4]])
        return frames

    seqs = {"stack": (stack_seq(30, 4), 0.4), "zero_mid": (zero_mid_seq(), 0.4),
            "neg_sigma": (make_track_sequence(rng, 30, 5, p_drop=0.2), -0.5)}
    result = {}
    for name, (frames, sigma) in seqs.items():
        ns = dict(np=np, calculate_iou=calculate_iou, calculate_distance=calculate_distance,
                  use_iou=True, sigma_iou=sigma, sigma_dis=8, sigma_h=0.6, t_min=5,
                  tracks_active=[], tracks_finished=[], frame_num=0)
        for det0 in frames:
            ns["frame_num"] += 1
            ns["det0"] = det0
            with np.errstate(all="ignore"):
                exec(body, ns)
        exec(final, ns)
        result[name] = {
            "sigma_iou": sigma,
            "frames": [f.tolist() for f in frames],
            "frame_dtypes": [str(f.dtype) for f in frames],
            "tracks": [{"bboxes": [list(map(float, b)) for b in t["bboxes"]],
                        "max_score": float(t["max_score"]), "start_frame": int(t["start_frame"])}
                       for t in ns["tracks_finished"]],
        }
        print("tracker_r5", name, "frames", len(frames), "dets/frame", [len(f) for f in frames][:8], "tracks", len(result[name]["tracks"]))
    with open(os.path.join(HERE, "tracker_r5.json"), "w") as f:
        json.dump({"sequences": result}, f)
    print("wrote tracker_r5.json", os.path.getsize(os.path.join(HERE, "tracker_r5.json")))


GENS = {"tracker_r5": gen_tracker_r5, "facebox": gen_facebox, "priors": gen_priors, "detect": gen_detect, "iou": gen_iou, "tracker": gen_tracker, "nets": gen_nets,
        "nets45": gen_nets45, "nets12": gen_nets12, "nets_r2": gen_nets_r2, "nets_r3": gen_nets_r3,
        "facebox_r2": gen_facebox_r2, "tp_fp": gen_tp_fp_fixture, "distance": gen_distance}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    for k, fn in GENS.items():
        if a.only in (None, k):
            fn()
