#!/usr/bin/env python3
"""R5-13: the split-bf16 1x1 class one tile per workgroup (21, tiles 5 / 6) against persistent (26, tiles 35 / 34) and the persistent
f32 class (16) on the layer1 / layer2 shapes of the backbone, batch 4."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import conv_bench as cb
SHAPES = [("layer1.x.conv3", 64, 256, 256, 256, 1), ("layer1.x.conv1", 256, 256, 256, 64, 0), ("layer1.0.downsample", 64, 256, 256, 256, 0),
          ("layer2.0.conv1", 256, 256, 256, 128, 0), ("layer2.x.conv3", 128, 128, 128, 512, 1), ("layer2.x.conv1", 512, 128, 128, 128, 0),
          ("layer3.x.conv3", 256, 64, 64, 1024, 1), ("layer3.x.conv1", 1024, 64, 64, 256, 0), ("layer4.x.conv3", 512, 32, 32, 2048, 1),
          ("conv3_ct_py.main", 512, 128, 128, 512, 0)]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for name, cin, h, w, cout, res in SHAPES:
    gf = 2.0 * B * h * w * cout * cin / 1e9
    row = []
    for kind, tiles in ((21, (5, 6)), (26, (35, 34)), (16, (35, 34))):
        best = None
        for t in tiles:
            ms = cb.bench(kind, t, 1, cin, h, w, cout, res, 0, 30, B)
            if ms and (best is None or ms < best[0]):
                best = (ms, t)
        row.append("%s %s %6.1f us %6.1f" % (cb.KIND[kind], cb.TILE[best[1]], best[0] * 1e3, gf / best[0]) if best else "%s -" % cb.KIND[kind])
    print("%-20s cin %4d %3dx%-3d cout %4d res %d | %s" % (name, cin, h, w, cout, res, " | ".join(row)), flush=True)
