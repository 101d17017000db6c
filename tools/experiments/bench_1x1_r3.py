#!/usr/bin/env python3
"""The direct-kernel layer classes of the Res50 graph at 1024x1024 (best variant each): used to compare the staging forms."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import conv_bench as cb

SHAPES = [(0, 1024, 64, 64, 256, 0), (0, 256, 64, 64, 1024, 1), (0, 512, 128, 128, 128, 0), (0, 128, 128, 128, 512, 1),
          (0, 64, 256, 256, 256, 1), (0, 256, 256, 256, 64, 0), (0, 2048, 32, 32, 512, 0), (0, 512, 32, 32, 2048, 1),
          (0, 256, 256, 256, 256, 0), (4, 128, 256, 256, 128, 0), (4, 256, 128, 128, 256, 0), (5, 3, 1024, 1024, 64, 0)]
for kind, cin, h, w, cout, res in SHAPES:
    best = None
    kinds = (0, 10, 11) if kind == 0 else (kind,)
    for k in kinds:
        for t in range(len(cb.TILE)):
            for sp in (1, 2, 4, 8):
                if sp > 1 and sp > cin // (2 * cb.KC[k]):
                    break
                ms = cb.bench(k, t, sp, cin, h, w, cout, res)
                if ms and (best is None or ms < best[0]):
                    best = (ms, cb.KIND[k], cb.TILE[t], sp)
    kk, s = cb.GEOM[kind]
    gf = 2.0 * ((h - 1) // s + 1) * ((w - 1) // s + 1) * cout * cin * kk * kk / 1e9
    print("%-6s %4d -> %4d @ %4dx%-4d res %d: %7.1f us  %6.1f TF/s  %s %s /%d" % (cb.KIND[kind], cin, cout, h, w, res, best[0] * 1e3,
                                                                                  gf / best[0], best[1], best[2], best[3]))
