#!/usr/bin/env python3
"""F(4x4) k-step at one and four frames per launch: which half of the LDS-DMA feed stalls it -- the patch (HBM / MALL) or the
transformed weights (L2)?  Stamps build x FDT_W44_EXP 32 (no patch DMA) / 64 (no weight DMA); results wrong by construction."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import conv_bench as cb
L = cb.lib.lib()
buf = (ctypes.c_longlong * 8)()
for kind, cin, h, w, cout, split, B in ((14, 256, 256, 256, 256, 1, 1), (14, 256, 256, 256, 256, 1, 4), (14, 512, 128, 128, 512, 1, 4)):
    cb.bench(kind, 32, split, cin, h, w, cout, iters=2, B=B)
    L.fdt_debug_w44_times(buf)
    ms = cb.bench(kind, 32, split, cin, h, w, cout, iters=10, B=B)
    L.fdt_debug_w44_times(buf)
    v = list(buf)
    n = max(v[4], 1)
    print("%s  %4d -> %4d @%3d^2 /%d batch %d  %7.1f us per launch  shader clock %4.0f MHz  %4.0f cycles per k-step"
          % (os.environ.get("W44_LABEL", ""), cin, cout, h, split, B, ms * 1e3, 100.0 * v[5] / max(v[1], 1), v[5] / n / (cin / 2.0 / split)), flush=True)
