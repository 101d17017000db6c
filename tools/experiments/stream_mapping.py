#!/usr/bin/env python3
"""Does the stream -> hardware-queue mapping (GPU_MAX_HW_QUEUES = 4, assigned as streams are created) matter for the
device-resident pipeline?  K dummy streams are created (and touched) before the pipeline's own streams.
    python tools/experiments/stream_mapping.py K"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
M = lambda n: importlib.import_module("face-detection-and-tracking_amd." + n)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 0
synth, layers = M("synth"), M("layers")
H = W = 1024
dev = torch.device("cuda", 0)
dummies = [torch.cuda.Stream(device=dev) for _ in range(K)]
for s in dummies:
    with torch.cuda.stream(s):
        torch.zeros(1, device=dev)
net = M("pyramid").SFD()
net.priorbox = layers.PriorBoxLayer(W, H)
net.load_state_dict(synth.make_state_dict("res50", 0))
net._sync_attributes(H, W)
plan = open(os.path.join(ROOT, "face-detection-and-tracking_amd", "tuned", "res50_1024x1024_b1.plan")).read()
pipe = M("pipeline").DetectTrackPipeline(net, H, W, dev, inflight=3, plan_text=plan)
fr = torch.from_numpy(synth.make_frames(16, H, W, seed=3)).to(dev)
pipe.prime(fr[0:1])
for i in range(8): pipe.step(i, fr[i % 16:i % 16 + 1])
torch.cuda.synchronize()
N = 96; t0 = time.perf_counter()
for i in range(N): pipe.step(8 + i, fr[i % 16:i % 16 + 1])
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("K = %d dummy streams first: %.1f frames/s" % (K, N / dt))
pipe.finish(); pipe.close()
