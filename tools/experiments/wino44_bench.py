#!/usr/bin/env python3
"""Winograd F(4x4,3x3) (conv_wino44.h, kind 14 / tile 32) against the best F(2x2,3x3) variant on the 3x3/s1 shapes of the
Res50 graph at 1024x1024: microseconds per launch (fdt_debug_conv_bench, random data, split-K swept).
    python tools/experiments/wino44_bench.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import conv_bench as cb

SHAPES = [(256, 256, 256, 256), (512, 128, 128, 512), (1024, 64, 64, 1024), (512, 128, 128, 256), (256, 256, 256, 128),
          (1024, 64, 64, 256), (128, 128, 128, 128), (256, 64, 64, 256), (64, 256, 256, 64), (512, 32, 32, 512)]
for cin, h, w, cout in SHAPES:
    gf = 2.0 * h * w * cout * cin * 9 / 1e9
    best2 = None
    for t in (29, 30, 22, 24):
        for sp in (1, 2, 4, 8, 16):
            if sp > cin // 16:
                break
            ms = cb.bench(8, t, sp, cin, h, w, cout)
            if ms and (best2 is None or ms < best2[0]):
                best2 = (ms, cb.TILE[t], sp)
    best4 = None
    per_tile = {}
    for t4 in (32, 33):
        for sp in (1, 2, 3, 4, 6, 8, 16):
            if sp > cin // 8:
                break
            ms = cb.bench(14, t4, sp, cin, h, w, cout)
            if ms and (t4 not in per_tile or ms < per_tile[t4][0]):
                per_tile[t4] = (ms, sp)
    best4 = min(per_tile.values())
    print("%4d -> %4d @ %3dx%-3d %6.2f GFLOP | F(2x2) %-10s /%-2d %7.1f us %6.1f alg TF/s | F(4x4) /%-2d %7.1f us %6.1f alg TF/s "
          "(%5.1f executed) | x%.2f | 8 waves %6.1f us /%d, 12 waves %6.1f us /%d" % (cin, cout, h, w, gf, best2[1], best2[2], best2[0] * 1e3, gf / best2[0], best4[1],
                                        best4[0] * 1e3, gf / best4[0], gf / 4 / best4[0], best2[0] / best4[0],
                                        per_tile[32][0] * 1e3, per_tile[32][1], per_tile[33][0] * 1e3, per_tile[33][1]))
