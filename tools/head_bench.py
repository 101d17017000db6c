#!/usr/bin/env python3
"""Narrow 3x3 heads (Cout = 8): the packed-f32 VALU kernel (kind 13) against the best MFMA variant per shape.
    python tools/head_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conv_bench as cb

if __name__ == "__main__":
    # (B, Cin, H, W): try3 b8 levels 0-2, Res50 b1 levels 0-2
    for (B, cin, h, w) in ((8, 256, 256, 256), (8, 256, 128, 128), (8, 256, 64, 64), (1, 512, 256, 256),
                           (1, 512, 128, 128), (1, 512, 64, 64), (2, 512, 256, 256)):
        gf = 2.0 * B * h * w * 8 * cin * 9 / 1e9
        best = None
        for kind in (2, 8):
            for t in range(len(cb.TILE)):
                for sp in (1, 2, 4, 8, 16):
                    ms = cb.bench(kind, t, sp, cin, h, w, 8, B=B, iters=10)
                    if ms and (best is None or ms < best[0]):
                        best = (ms, cb.KIND[kind], cb.TILE[t], sp)
        line = "B%d cin %d %dx%d (%.2f GF): MFMA best %s %s /%d %.1f us (%.1f TF/s);" % (
            B, cin, h, w, gf, best[1], best[2], best[3], best[0] * 1e3, gf / best[0])
        for sp in (1, 2, 4, 8, 16, 32):
            ms = cb.bench(13, 31, sp, cin, h, w, 8, B=B, iters=10)
            if ms:
                line += "  n8/%d %.1f us (%.1f TF/s)" % (sp, ms * 1e3, gf / ms)
        print(line, flush=True)
