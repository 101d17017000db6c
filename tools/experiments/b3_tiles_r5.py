#!/usr/bin/env python3
"""R5-11: the split-bf16 1x1 class, wave layouts 2 x 2 (tiles 11, 12) against 4 x 1 (tiles 5, 6) on the backbone's shapes, batch 4."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import conv_bench as cb
SHAPES = [("layer2.x.conv1", 512, 128, 128, 128, 0), ("layer3.x.conv1", 1024, 64, 64, 256, 0), ("layer3.x.conv3", 256, 64, 64, 1024, 1),
          ("layer4.x.conv1", 2048, 32, 32, 512, 0), ("layer4.x.conv3", 512, 32, 32, 2048, 1), ("latlayer_fc", 2048, 32, 32, 2048, 0),
          ("conv3_ct_py.main", 512, 128, 128, 512, 0), ("conv4_ct_py.main", 1024, 64, 64, 1024, 0), ("layer2.0.conv1", 256, 256, 256, 128, 0),
          ("layer1.x.conv1", 256, 256, 256, 64, 0), ("layer2.x.conv3", 128, 128, 128, 512, 1), ("layer1.x.conv3", 64, 256, 256, 256, 1)]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for name, cin, h, w, cout, res in SHAPES:
    gf = 2.0 * B * h * w * cout * cin / 1e9
    row = []
    for t in (11, 5, 25, 12, 6, 26):
        best = None
        for sp in (1, 2, 4):
            ms = cb.bench(21, t, sp, cin, h, w, cout, res, 0, 30, B)
            if ms and (best is None or ms < best[0]):
                best = (ms, sp)
        row.append("%s/%d %6.1f us %6.1f" % (cb.TILE[t], best[1], best[0] * 1e3, gf / best[0]) if best else "%s -" % cb.TILE[t])
    print("%-18s cin %4d %3dx%-3d cout %4d res %d | %s" % (name, cin, h, w, cout, res, " | ".join(row)), flush=True)
print("stride 2 (downsample branch): f32 class 1 best tile against class 23")
for name, cin, h, w, cout in [("layer2.0.downsample", 256, 256, 256, 512), ("layer3.0.downsample", 512, 128, 128, 1024), ("layer4.0.downsample", 1024, 64, 64, 2048)]:
    gf = 2.0 * B * (h // 2) * (w // 2) * cout * cin / 1e9
    row = []
    for kind, tiles in ((1, (0, 1, 7, 8)), (23, (5, 6, 25, 26))):
        best = None
        for t in tiles:
            for sp in (1, 2, 4):
                ms = cb.bench(kind, t, sp, cin, h, w, cout, 0, 0, 30, B)
                if ms and (best is None or ms < best[0]):
                    best = (ms, sp, t)
        row.append("%s %s/%d %6.1f us %6.1f" % (cb.KIND[kind], cb.TILE[best[2]], best[1], best[0] * 1e3, gf / best[0]) if best else "%s -" % cb.KIND[kind])
    print("%-20s cin %4d %3dx%-3d cout %4d | %s" % (name, cin, h, w, cout, " | ".join(row)), flush=True)
