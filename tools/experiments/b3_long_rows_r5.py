#!/usr/bin/env python3
"""R5-13: split-bf16 1x1 class, tile shapes 4x32 (tiles 5 / 6) against 2x64 (37 / 38) and 1x128 (39 / 40) pixels, batch 4."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import conv_bench as cb
SHAPES = [("layer1.x.conv3", 64, 256, 256, 256, 1), ("layer1.x.conv1", 256, 256, 256, 64, 0), ("layer1.0.downsample", 64, 256, 256, 256, 0),
          ("layer2.0.conv1", 256, 256, 256, 128, 0), ("layer2.x.conv3", 128, 128, 128, 512, 1), ("layer2.x.conv1", 512, 128, 128, 128, 0),
          ("layer3.x.conv3", 256, 64, 64, 1024, 1), ("layer3.x.conv1", 1024, 64, 64, 256, 0), ("conv3_ct_py.main", 512, 128, 128, 512, 0)]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for name, cin, h, w, cout, res in SHAPES:
    gf = 2.0 * B * h * w * cout * cin / 1e9
    row = []
    for t in (5, 37, 39, 6, 38, 40):
        ms = cb.bench(21, t, 1, cin, h, w, cout, res, 0, 30, B)
        row.append("%s %6.1f us" % (cb.TILE[t], ms * 1e3) if ms else "%s -" % cb.TILE[t])
    print("%-20s cin %4d %3dx%-3d cout %4d res %d | %s" % (name, cin, h, w, cout, res, " | ".join(row)), flush=True)
