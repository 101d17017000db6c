#!/usr/bin/env python3
"""The 1x1 / stride-1 shapes of the Res50 backbone at 1024^2 (pyramid.py:97-103): best one-tile-per-workgroup variant of the
direct classes (kinds 0 / 10 / 11, every tile and split) against the persistent-tile class (kinds 16 / 17, conv_1x1p.h).
    python tools/experiments/bench_1x1_r4.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from conv_bench import bench, KIND, TILE, KC

SHAPES = [  # name, cin, h, w, cout, res
    ("layer1.0.conv1", 64, 256, 256, 64, 0), ("layer1.x.conv1", 256, 256, 256, 64, 0), ("layer1.x.conv3", 64, 256, 256, 256, 1),
    ("layer1.0.downsample", 64, 256, 256, 256, 0),
    ("layer2.0.conv1", 256, 256, 256, 128, 0), ("layer2.x.conv1", 512, 128, 128, 128, 0), ("layer2.x.conv3", 128, 128, 128, 512, 1),
    ("layer3.0.conv1", 512, 128, 128, 256, 0), ("layer3.x.conv1", 1024, 64, 64, 256, 0), ("layer3.x.conv3", 256, 64, 64, 1024, 1),
    ("layer4.0.conv1", 1024, 64, 64, 512, 0), ("layer4.x.conv1", 2048, 32, 32, 512, 0), ("layer4.x.conv3", 512, 32, 32, 2048, 1),
    ("latlayer_fc", 2048, 32, 32, 2048, 0), ("ct_py.up_conv c3", 512, 128, 128, 256, 0),
]
ITERS = int(os.environ.get("ITERS", "30"))
for name, cin, h, w, cout, res in SHAPES:
    gf = 2.0 * h * w * cout * cin / 1e9
    best = None
    for kind in (0, 10, 11):
        nst = (cin + KC[kind] - 1) // KC[kind]
        for t in range(len(TILE)):
            for sp in (1, 2, 4, 8, 16):
                if sp > 1 and (sp > nst // 2 or nst < 8):
                    break
                ms = bench(kind, t, sp, cin, h, w, cout, res, 0, iters=ITERS)
                if ms and (best is None or ms < best[0]):
                    best = (ms, kind, t, sp)
    line = "%-22s %5.2f GF  best direct %-7s %-9s /%-2d %6.1f us |" % (name, gf, KIND[best[1]], TILE[best[2]], best[3], best[0] * 1e3)
    for kind, t in ((16, 34), (16, 35), (17, 34)):
        ms = bench(kind, t, 1, cin, h, w, cout, res, 0, iters=ITERS)
        line += "  %s %s %s" % (KIND[kind], TILE[t], ("%6.1f us" % (ms * 1e3)) if ms else "   n/a")
    print(line, flush=True)
