#!/usr/bin/env python3
"""Time the fused expand + depthwise kernel on the InvertedResidual shapes of try3 (batch 8 by default).
    python tools/expand_dw_bench.py [B]"""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = importlib.import_module("face-detection-and-tracking_amd._lib").lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
# (block, Cin, H, W, hidden, stride): pyramid_mb2_try3.py:150-168 at a 1024x1024 input
shapes = [("features.2", 16, 512, 512, 96, 2), ("features.3", 24, 256, 256, 144, 1), ("features.4", 24, 256, 256, 144, 2),
          ("features.5/6", 32, 128, 128, 192, 1), ("features.7", 32, 128, 128, 192, 2), ("features.8-10", 64, 64, 64, 384, 1),
          ("features.11", 64, 64, 64, 384, 1), ("features.12/13", 96, 64, 64, 576, 1), ("features.14", 96, 64, 64, 576, 2),
          ("features.15/16", 160, 32, 32, 960, 1)]
for name, cin, h, w, hid, s in shapes:
    ms = C.c_float(0)
    rc = lib.fdt_debug_expand_dw_bench(B, cin, h, w, hid, s, 10, C.byref(ms))
    ho, wo = (h - 1) // s + 1, (w - 1) // s + 1
    by = 4.0 * B * (cin * h * w + hid * ho * wo)
    print("%-16s cin %3d %4dx%-4d hid %3d s%d: %s" % (name, cin, h, w, hid, s,
          "%.1f us  %.2f TB/s (in + out only)" % (ms.value * 1e3, by / ms.value / 1e9) if rc == 0 else "n/a (rc %d)" % rc))
