#!/usr/bin/env python3
"""What does the sequential association cost on the detections bench.py itself produces?  

    python tools/experiments/tracker_on_bench_records.py [--size 1024 | --height 480 --width 640] [--frames 64] [--dump F.npy]

1. Runs the bench's own workload (Res50, seeded synthetic weights, `--unique-frames 8` synthetic frames cycling, committed plan)
   through the product pipeline on ONE rank and keeps every frame's Detect record [2, 750, 5] (what an all-gather would carry).
2. Replays those records through fdt_tracker_step_dev_multi with G = 1, 2, 4, 8 frames per launch -- the launch an N-rank
   frame-parallel step runs on EVERY rank -- and reports us per launch and per frame (HIP events on the tracker's stream),
   detections per frame (score >= 0.4) and live tracks.
3. The same with post-NMS-like records (tools/tracker_bench.py's generator: no two detections of a frame overlap).
With a library built with -DFDT_TRK_TIMING the per-phase device clocks are printed too.
"""
import argparse
import ctypes
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("face-detection-and-tracking_amd")
synth = importlib.import_module("face-detection-and-tracking_amd.synth")
layers = importlib.import_module("face-detection-and-tracking_amd.layers")
pipeline = importlib.import_module("face-detection-and-tracking_amd.pipeline")
trk = importlib.import_module("face-detection-and-tracking_amd.tracker")
lib = pkg._lib

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--height", type=int, default=0)
ap.add_argument("--width", type=int, default=0)
ap.add_argument("--frames", type=int, default=64)
ap.add_argument("--dump", default="")
ap.add_argument("--load", default="", help="use records from this .npy instead of running the detector")
args = ap.parse_args()
H, W = args.height or args.size, args.width or args.size
dev = torch.device("cuda", 0)
TOP_K = 750
REC = 2 * TOP_K * 5
F = args.frames

if args.load:
    recs = np.load(args.load)
    if hasattr(recs, "files"):
        recs = recs["recs"]
    F = recs.shape[0]
else:
    sd = synth.make_state_dict("res50", seed=0)
    net = importlib.import_module("face-detection-and-tracking_amd.pyramid").SFD(device=0)
    net.priorbox = layers.PriorBoxLayer(W, H)
    net.load_state_dict(sd)
    net.cuda(); net.eval()
    net.enable_graph(True)
    net._sync_attributes(H, W)
    plan = os.path.join(ROOT, "face-detection-and-tracking_amd", "tuned", "res50_%dx%d_b1.plan" % (W, H))
    plan_text = open(plan).read() if os.path.exists(plan) else None
    pipe = pipeline.DetectTrackPipeline(net, H, W, dev, inflight=1, plan_text=plan_text)
    frames_d = torch.from_numpy(synth.make_frames(8, H, W, seed=1234)).to(dev)      # bench.py: --unique-frames 8, seed 1234 + rank
    pipe.prime(frames_d[0:1])
    recs = np.zeros((F, REC), np.float32)
    for i in range(F):
        pipe.step(i, frames_d[i % 8:i % 8 + 1])
        recs[i] = pipe.record_of_slot(0).reshape(-1)
    pipe.finish()
    pipe.close()
    net.close()
    if args.dump:
        np.save(args.dump, recs)


def post_nms_like(N, persist, F, seed=0):
    rng = np.random.default_rng(seed)
    G_ = 40
    cells = rng.permutation(G_ * G_)[:N]
    out = np.zeros((F, 2, TOP_K, 5), np.float32)
    for f in range(F):
        move = rng.uniform(size=N) >= persist
        free = np.setdiff1d(np.arange(G_ * G_), cells)
        cells[move] = rng.permutation(free)[:int(move.sum())]
        xy = np.stack([(cells % G_) / G_, (cells // G_) / G_], 1) * 0.9 + 0.02 + rng.uniform(0, 0.001, (N, 2))
        out[f, 1, :N, 0] = np.sort(rng.uniform(0.41, 1.0, N))[::-1]
        out[f, 1, :N, 1:3] = xy
        out[f, 1, :N, 3:5] = xy + 0.018
    return out.reshape(F, REC)


L = lib.lib()
st = torch.cuda.Stream(device=dev)
sp = ctypes.c_void_p(st.cuda_stream)


def phases(frames):
    buf = (ctypes.c_longlong * 8)()
    try:
        L.fdt_debug_trk_times(buf)
    except AttributeError:
        return ""
    v = list(buf)
    return "   phases us/frame: unpack %.1f scan %.1f sort %.1f greedy %.1f spawn %.1f tail %.1f; exact-form frames %d of %d" % (
        v[0] / 100.0 / frames, v[5] / 100.0 / frames, v[1] / 100.0 / frames, v[2] / 100.0 / frames, v[3] / 100.0 / frames,
        v[4] / 100.0 / frames, v[6], frames)


def replay(recs, label):
    F = recs.shape[0]
    n_det = [(int((r.reshape(2, TOP_K, 5)[1, :, 0] >= 0.4).sum())) for r in recs]
    print("%s: %d frames, detections/frame (score >= 0.4) min %d median %d max %d" % (
        label, F, min(n_det), int(np.median(n_det)), max(n_det)))
    rd = torch.from_numpy(recs).to(dev)
    ref_tracks = None
    for G in (1, 2, 4, 8):
        t = trk.IouTracker(0.4, 0.6, 5, max_dets=2 * TOP_K, log_frames=256)
        warm = 8                                     # frames: the tracker reaches its steady population within a cycle
        for f0 in range(0, warm, G):
            t.step_dev_multi(ctypes.c_void_p(rd.data_ptr() + 4 * f0 * REC), G, REC, 2, TOP_K, W, H, 0.4, sp)
        torch.cuda.synchronize()
        phases(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        launches = 0
        for f0 in range(warm, F - G + 1, G):
            t.step_dev_multi(ctypes.c_void_p(rd.data_ptr() + 4 * f0 * REC), G, REC, 2, TOP_K, W, H, 0.4, sp)
            launches += 1
        e1.record(st)
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3
        ph = phases(launches * G)
        tracks = t.finish()
        key = [(x["start_frame"], x["max_score"], x["bboxes"]) for x in tracks]
        if G == 1:
            ref_tracks = key
        same = "" if (F - warm) % 8 else ("  tracks == G=1: %s" % (key == ref_tracks))
        print("  G=%d: %7.1f us per launch, %6.1f us per frame (%d launches, %d finished tracks)%s%s" % (
            G, us / launches, us / launches / G, launches, len(tracks), same, ph))
        t.close()


replay(recs, "bench records %dx%d" % (W, H))
for N, persist in ((50, 0.9), (300, 0.9), (500, 0.9), (750, 0.9)):
    replay(post_nms_like(N, persist, F), "post-NMS-like N=%d persist=%.1f" % (N, persist))
