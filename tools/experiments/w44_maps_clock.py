#!/usr/bin/env python3
"""F(4x4) kernel at the production launch size (four frames per launch): workgroup -> tile map vs launch time and the shader clock
held in the main loop (stamps build: tools/experiments/w44_stamps.sh).  Does keeping an XCD on few channel tiles (its share of
the transformed weights L2-resident) change the power the L2 -> LDS feed draws?"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import conv_bench as cb
L = cb.lib.lib()
buf = (ctypes.c_longlong * 8)()
MAPS = ["rows", "xcd-spatial", "xcd-channel", "xcd-region"]
for kind, cin, h, w, cout, split, B in ((14, 256, 256, 256, 256, 1, 4), (14, 512, 128, 128, 512, 1, 4), (14, 1024, 64, 64, 1024, 2, 4), (14, 256, 256, 256, 256, 1, 1)):
    for m in range(4):
        os.environ["FDT_CONV_MAP"] = str(m)
        cb.bench(kind, 32, split, cin, h, w, cout, iters=2, B=B)
        L.fdt_debug_w44_times(buf)
        ms = cb.bench(kind, 32, split, cin, h, w, cout, iters=10, B=B)
        L.fdt_debug_w44_times(buf)
        v = list(buf)
        n = max(v[4], 1)
        print("%4d -> %4d @%3d^2 /%d batch %d  map %-12s %7.1f us per launch  main loop %6.1f us per workgroup  shader clock %4.0f MHz  %4.0f cycles per k-step"
              % (cin, cout, h, split, B, MAPS[m], ms * 1e3, v[1] / 100.0 / n, 100.0 * v[5] / max(v[1], 1), v[5] / n / (cin / 2.0 / split)), flush=True)
