#!/usr/bin/env python3
"""Dilation-2 Winograd F(4x4,3x3) (kind 15 / tile 32) against the best F(2x2,3x3) dilated variant on the SSH context shapes."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import conv_bench as cb

for cin, h, w, cout in [(256, 256, 256, 128), (128, 256, 256, 128), (256, 128, 128, 128), (128, 128, 128, 128), (256, 64, 64, 128)]:
    gf = 2.0 * h * w * cout * cin * 9 / 1e9
    best2 = min((cb.bench(9, t, sp, cin, h, w, cout) or 1e9, cb.TILE[t], sp) for t in (29, 30, 22, 24) for sp in (1, 2, 4, 8) if sp <= cin // 16)
    best4 = min((cb.bench(15, 32, sp, cin, h, w, cout) or 1e9, sp) for sp in (1, 2, 4, 8, 16) if sp <= cin // 8)
    print("%4d -> %4d @ %3dx%-3d d2 %6.2f GFLOP | F(2x2) %-10s /%-2d %7.1f us | F(4x4) /%-2d %7.1f us (%5.1f executed TF/s) | x%.2f"
          % (cin, cout, h, w, gf, best2[1], best2[2], best2[0] * 1e3, best4[1], best4[0] * 1e3, gf / 4 / best4[0], best2[0] / best4[0]))
