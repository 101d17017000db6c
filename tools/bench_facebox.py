#!/usr/bin/env python3
"""Config 5 of BASELINE.json: FaceBoxes, 1024x1024 frames, batch 16 per step on one GPU, /255 + forward +
softmax + decode_np + nms_np all on device (reference FACEBOX/My_test_facebox.py:12-36 after the resize).
Real weights (tests/golden/faceboxes_weights.npz = the reference's FACEBOX/faceboxes.pt as arrays).
    python tools/bench_facebox.py [--batch 16] [--steps 50]"""
import argparse, ctypes, importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16); ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--warmup", type=int, default=5); ap.add_argument("--cpu-frames", type=int, default=4)
a = ap.parse_args()
lib = importlib.import_module("face-detection-and-tracking_amd._lib")
synth = importlib.import_module("face-detection-and-tracking_amd.synth")
FaceBox = importlib.import_module("face-detection-and-tracking_amd.FACEBOX.networks").FaceBox
z = np.load(os.path.join(ROOT, "tests", "golden", "faceboxes_weights.npz"))
sd = {k: z[k] for k in z.files}
net = FaceBox(); net.load_state_dict(sd)
g = np.load(os.path.join(ROOT, "tests", "golden", "facebox.npz"))
real = [g["img0_frame"], g["img1_frame"]]
frames_h = np.stack([real[i % 2] for i in range(a.batch)])
dev = torch.device("cuda", 0)
frames_d = torch.from_numpy(frames_h).to(dev)
counts = torch.zeros(a.batch, dtype=torch.int32, device=dev)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); sp = ctypes.c_void_p(st.cuda_stream)
L = lib.lib()
net.detect_frames(frames_h)          # plan
net.autotune(3)
def step():
    lib.check(L.fdt_model_detect_facebox_dev(net._h, ctypes.c_void_p(frames_d.data_ptr()), lib.FRAME_U8_HWC_BGR,
                                             a.batch, 1024, 1024, 0.35, 0.5, ctypes.c_void_p(counts.data_ptr()), sp))
for _ in range(a.warmup): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps): step()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
res = net.detect_frames(frames_h)
from oracle import facebox as ofb
times = []
for i in range(a.cpu_frames):
    t1 = time.perf_counter(); rb, rp = ofb.detect(sd, frames_h[i % 2]); times.append(time.perf_counter() - t1)
    gb, gp = res[i % 2]
    assert len(gp) == len(rp) and np.abs(gp - rp).max() < 1e-4 and np.abs(gb - rb).max() < 1e-4
per = float(np.mean(times[1:]))
print(json.dumps({"metric": "frames/sec (FaceBoxes detect) at 1024x1024", "value": round(a.batch * a.steps / dt, 2),
                  "unit": "frames/s", "n_gpus": 1, "steps": a.steps, "ms_per_step": round(dt / a.steps * 1e3, 3),
                  "dtype": "f32", "data": "2 reference sample images (post-resize) tiled to the batch",
                  "config": {"workload": "FaceBoxes 1024x1024 batch=%d, decode_np+nms_np on device" % a.batch,
                             "weights": "reference FACEBOX/faceboxes.pt", "faces_per_image": [int(c) for c in counts.cpu()[:2]]},
                  "cpu_baseline": {"value": round(1 / per, 3), "unit": "frames/s", "cores": torch.get_num_threads(),
                                   "kind": "port", "sample": "%d frames, oracle/facebox.py" % (len(times) - 1)},
                  "parity": "boxes/probs of the CPU-sample frames within 1e-4 of the oracle"}))
