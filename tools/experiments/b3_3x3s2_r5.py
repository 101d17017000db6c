#!/usr/bin/env python3
"""R5-14: 3x3 / stride 2 -- the f32 direct class (4) against the split-bf16 class (27) on the three backbone layers, batch 4 and 1."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import conv_bench as cb
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for name, cin, h, w, cout in [("layer2.0.conv2", 128, 256, 256, 128), ("layer3.0.conv2", 256, 128, 128, 256), ("layer4.0.conv2", 512, 64, 64, 512), ("facebox conv3_2", 128, 32, 32, 256)]:
    gf = 2.0 * B * (h // 2) * (w // 2) * cout * cin * 9 / 1e9
    row = []
    for kind, tiles in ((4, range(0, 14)), (27, (5, 6))):
        best = None
        for t in tiles:
            for sp in (1, 2, 4, 8):
                ms = cb.bench(kind, t, sp, cin, h, w, cout, 0, 0, 30, B)
                if ms and (best is None or ms < best[0]):
                    best = (ms, sp, t)
        row.append("%s %s/%d %6.1f us %6.1f" % (cb.KIND[kind], cb.TILE[best[2]], best[1], best[0] * 1e3, gf / best[0]) if best else "%s -" % cb.KIND[kind])
    print("%-18s cin %4d %3dx%-3d cout %4d | %s" % (name, cin, h, w, cout, " | ".join(row)), flush=True)
