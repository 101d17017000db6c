"""Which library call leaves a sticky HIP error behind?  hipGetLastError() after every step of a try3 forward sequence."""
import ctypes, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401  (load order, see _lib.py)
M = lambda n: importlib.import_module("face-detection-and-tracking_amd." + n)
lib = M("_lib"); L = lib.lib()
hip = ctypes.CDLL("libamdhip64.so")
def chk(tag):
    e = hip.hipGetLastError()
    print("%-40s lastError=%d" % (tag, e), flush=True)
synth = M("synth")
sd = synth.make_state_dict("try3", seed=0)
H, W, B = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (512, 512, 2)))
frames = synth.make_frames(B, H, W, seed=31)
chk("start")
net = M("pyramid_mb2_try3").build_sfd_mobile('test', 640, 2); chk("create")
net.load_state_dict(sd); chk("load_state_dict")
net.priorbox = M("layers").PriorBoxLayer(W, H, stride=[4, 8, 16, 32, 64], box=(16, 32, 64, 128, 256))
net.detect = M("layers").Detect(2, 0, 750, 0.02, 0.35)
try:
    y = net(frames).numpy(); chk("forward 1")
    y = net(frames).numpy(); chk("forward 2 (graph)")
    net.profile(True); chk("profile on")
    net(frames); chk("forward 3 (profiled)")
    p = net.profile_read(); chk("profile_read")
    net.profile(False)
    net.get_tensor("stem"); chk("get_tensor stem")
    net.get_tensor("features.1.conv.3"); chk("get_tensor features.1.conv.3")
except Exception as e:
    print("EXC", e); chk("after exception")
net.close(); chk("close")
