#!/usr/bin/env python3
"""Sweep (tile, split-K) for conv shapes with the library's tuning hook.
    python tools/conv_bench.py                 # the Res50 @1024 layer classes
    python tools/conv_bench.py K CIN H W COUT [res]   # one shape, all tiles/splits (K = kind index)
"""
import ctypes as C
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = importlib.import_module("face-detection-and-tracking_amd._lib")
KIND = ["1x1s1", "1x1s2", "3x3s1", "3x3d2", "3x3s2", "7x7s2", "7x7s4", "5x5s2", "3x3wino", "3x3d2wino", "1x1k32", "1x1k64", "7x7s2p1", "3x3n8", "3x3wino44", "3x3d2wino44", "1x1p16", "1x1p32", "7x7s2u8", "7x7s4u8", "7x7s4k168", "1x1b3", "7x7s4b3", "1x1s2b3", "7x7s2u8b", "7x7s4u8b", "1x1pb3", "3x3s2b3"]
TILE = ["128x128", "128x64", "128x32", "64x64", "64x128", "128x128W", "128x64W", "128x128R3", "128x64R3",
        "64x64R3", "64x128R3", "128x128WR3", "128x64WR3", "128x32R3", "w64x64", "w64x64R3", "w128x32", "w128x32R3",
        "w32x128", "w32x128R3", "w64x64W", "w8_64x64", "w8_64x64R3", "w8_128x32R3", "w8_64x64W", "128x128R4", "128x64R4", "64x64R4", "64x128R4", "w4_64x64R3", "w4_64x64W", "n8_32x64", "w44_32x64", "w44b_32x64", "p128x64", "p128x128", "128x32W", "r2_128x128", "r2_128x64", "r1_128x128", "r1_128x64"]
GEOM = {0: (1, 1), 1: (1, 2), 2: (3, 1), 3: (3, 1), 4: (3, 2), 5: (7, 2), 6: (7, 4), 7: (5, 2), 8: (3, 1), 9: (3, 1), 10: (1, 1), 11: (1, 1), 12: (7, 2), 13: (3, 1), 14: (3, 1), 15: (3, 1), 16: (1, 1), 17: (1, 1), 18: (7, 2), 19: (7, 4), 20: (7, 4), 21: (1, 1), 22: (7, 4), 23: (1, 2), 24: (7, 2), 25: (7, 4), 26: (1, 1), 27: (3, 2)}
KC = {0: 16, 1: 16, 2: 4, 3: 4, 4: 4, 5: 2, 6: 2, 7: 2, 8: 8, 9: 8, 10: 32, 11: 64, 12: 2, 13: 1, 14: 2, 15: 2, 16: 16, 17: 32, 18: 4, 19: 3, 20: 3, 21: 16, 22: 3, 23: 16, 24: 3, 25: 3, 26: 16, 27: 16}


def bench(kind, tile, split, cin, h, w, cout, res=0, up=0, iters=20, B=1):
    L = lib.lib()
    L.fdt_debug_conv_bench.restype = C.c_int
    ms = C.c_float(0)
    rc = L.fdt_debug_conv_bench(kind, tile, split, B, cin, h, w, cout, res, up, 1, iters, C.byref(ms))
    return ms.value if rc == 0 else None


def sweep(kind, cin, h, w, cout, res=0, up=0, B=1):
    k, s = GEOM[kind]
    ho, wo = (h - 1) // s + 1, (w - 1) // s + 1
    gf = 2.0 * B * ho * wo * cout * cin * k * k / 1e9
    nst = (cin + KC[kind] - 1) // KC[kind]
    rows = []
    for t in range(len(TILE)):
        for sp in (1, 2, 4, 8, 16, 32, 64):
            if sp > 1 and sp > nst // 2:
                break
            ms = bench(kind, t, sp, cin, h, w, cout, res, up, B=B)
            if ms:
                rows.append((ms, t, sp))
    rows.sort()
    print("%s cin %d %dx%d cout %d res %d: %.2f GFLOP" % (KIND[kind], cin, h, w, cout, res, gf))
    for ms, t, sp in rows[:6]:
        print("   %-8s /%-2d %8.1f us %7.1f TF/s" % (TILE[t], sp, ms * 1e3, gf / ms))
    return rows


if __name__ == "__main__":
    if len(sys.argv) > 5:
        a = list(map(int, sys.argv[1:]))
        sweep(a[0], a[1], a[2], a[3], a[4], a[5] if len(a) > 5 else 0)
    else:
        shapes = [(8, 256, 256, 256, 256, 0), (8, 512, 128, 128, 512, 0), (8, 1024, 64, 64, 1024, 0),
                  (8, 64, 256, 256, 64, 0), (8, 128, 128, 128, 128, 0), (8, 256, 64, 64, 256, 0),
                  (8, 512, 32, 32, 512, 0), (8, 2048, 32, 32, 256, 0), (2, 256, 256, 256, 256, 0), (0, 64, 256, 256, 256, 1), (0, 256, 256, 256, 64, 0),
                  (2, 64, 256, 256, 64, 0), (0, 512, 128, 128, 128, 0), (0, 128, 128, 128, 512, 1),
                  (2, 128, 128, 128, 128, 0), (0, 1024, 64, 64, 256, 0), (0, 256, 64, 64, 1024, 1),
                  (2, 256, 64, 64, 256, 0), (0, 2048, 32, 32, 512, 0), (0, 512, 32, 32, 2048, 1),
                  (2, 512, 32, 32, 512, 0), (0, 2048, 32, 32, 2048, 0), (2, 2048, 32, 32, 256, 0),
                  (2, 512, 256, 256, 8, 0), (2, 1024, 64, 64, 1024, 0), (3, 256, 256, 256, 128, 0)]
        for sh in shapes:
            sweep(*sh)
