#!/usr/bin/env python3
"""expand_dw_kernel on the three try3 blocks that use it at 1024^2, batch 8 (fdt_debug_expand_dw_bench: zero-filled buffers)."""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
lib = importlib.import_module("face-detection-and-tracking_amd._lib")
L = lib.lib()
for name, cin, hw, hid, stride in (("features.2", 16, 512, 96, 2), ("features.3", 24, 256, 144, 1), ("features.4", 24, 256, 144, 2)):
    ms = C.c_float(0)
    rc = L.fdt_debug_expand_dw_bench(8, cin, hw, hw, hid, stride, 20, C.byref(ms))
    print("%s  %d -> %d @%d^2 stride %d batch 8: rc %d  %.1f us" % (name, cin, hid, hw, stride, rc, ms.value * 1e3), flush=True)
