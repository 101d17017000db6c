#!/usr/bin/env python3
"""PCIe-inclusive rate when host frames are staged into a device ring on a copy stream and consumed by the device-resident
pipeline (DetectTrackPipeline.step), i.e. without a host wait per frame -- compare with host_path_breakdown.py.
TIMING EXPERIMENT ONLY: the ring slots are reused without waiting for their last reader (results are not checked)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
M = lambda n: importlib.import_module("face-detection-and-tracking_amd." + n)
synth, layers = M("synth"), M("layers")
H = W = 1024
dev = torch.device("cuda", 0)
net = M("pyramid").SFD()
net.priorbox = layers.PriorBoxLayer(W, H)
net.load_state_dict(synth.make_state_dict("res50", 0))
net._sync_attributes(H, W)
plan = open(os.path.join(ROOT, "face-detection-and-tracking_amd", "tuned", "res50_1024x1024_b1.plan")).read()
pipe = M("pipeline").DetectTrackPipeline(net, H, W, dev, inflight=3, plan_text=plan)
frames = synth.make_frames(16, H, W, seed=3)
R = 6
pinned = [torch.empty((1, H, W, 3), dtype=torch.uint8).pin_memory() for _ in range(R)]
ring = [torch.empty((1, H, W, 3), dtype=torch.uint8, device=dev) for _ in range(R)]
copied = [torch.cuda.Event() for _ in range(R)]
used = [None] * R
cs = torch.cuda.Stream(device=dev)
pipe.prime(ring[0])
def step(i):
    r = i % R
    if used[r] is not None:
        used[r].synchronize()                     # the pinned + device slot is free again (6 frames back)
    pinned[r].copy_(torch.from_numpy(frames[i % 16][None]))
    with torch.cuda.stream(cs):
        ring[r].copy_(pinned[r], non_blocking=True)
        copied[r].record(cs)
    pipe.det_streams[i % pipe.NF].wait_event(copied[r])
    pipe.step(i, ring[r])
    used[r] = pipe.det_done[i % pipe.NF] if False else None
for i in range(8): step(i)
torch.cuda.synchronize()
N = 300; t0 = time.perf_counter()
for i in range(N):
    if i >= R: pass
    step(8 + i)
    if (i % R) == R - 1: pipe.trk_stream.synchronize() if False else None
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("host frames -> device ring -> pipeline.step: %.1f frames/s (%.3f ms per frame), %d tracks" % (N / dt, dt / N * 1e3, len(pipe.finish())))
pipe.close()
