#!/usr/bin/env python3
"""Run ONE conv configuration a few times (for rocprofv3 --pmc / --kernel-trace).
    python tools/one_conv.py KIND TILE SPLIT CIN H W COUT [RES] [ITERS] [BATCH]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cb = importlib.import_module("tools.conv_bench") if False else None
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conv_bench as cb
a = list(map(int, sys.argv[1:]))
kind, tile, split, cin, h, w, cout = a[:7]
res = a[7] if len(a) > 7 else 0
iters = a[8] if len(a) > 8 else 5
B = a[9] if len(a) > 9 else 1
ms = cb.bench(kind, tile, split, cin, h, w, cout, res, 0, iters, B)
k, s = cb.GEOM[kind]
gf = 2.0 * B * ((h - 1) // s + 1) * ((w - 1) // s + 1) * cout * cin * k * k / 1e9
print("%s %s /%d cin %d %dx%d cout %d batch %d: %.1f us %.1f TF/s" % (cb.KIND[kind], cb.TILE[tile], split, cin, h, w, cout, B, ms * 1e3, gf / ms))
