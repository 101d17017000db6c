#!/usr/bin/env python3
"""Where does a workgroup of conv_wino44_kernel spend its time?  (library built with make EXTRA=-DFDT_W44_STAMPS, loaded through
FDT_LIB: tools/experiments/w44_stamps.sh)  Per workgroup averages of the constant-rate (100 MHz) clock between four stamps."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import conv_bench as cb
L = cb.lib.lib()
buf = (ctypes.c_longlong * 8)()
SHAPES = [(14, 256, 256, 256, 256, 1), (14, 512, 128, 128, 512, 1), (14, 1024, 64, 64, 1024, 2), (14, 256, 256, 256, 128, 1), (15, 256, 256, 256, 128, 1),
          (14, 128, 128, 128, 128, 1)]
for kind, cin, h, w, cout, split in SHAPES[:int(os.environ.get('W44_SHAPES', len(SHAPES)))]:
    cb.bench(kind, 32, split, cin, h, w, cout, iters=3)
    L.fdt_debug_w44_times(buf)                      # drop the warm-up launches
    ms = cb.bench(kind, 32, split, cin, h, w, cout, iters=20)
    L.fdt_debug_w44_times(buf)
    v = list(buf)
    n = max(v[4], 1)
    us = [x / 100.0 / n for x in v[:4]]
    mhz = 100.0 * v[5] / max(v[1], 1)
    print("%s %4d -> %4d @ %3dx%-3d /%d: %6.1f us per launch; per workgroup (%d per launch): prologue %5.1f  main loop %6.1f  epilogue round 0 %5.1f  round 1 %5.1f us;"
          " shader clock in the main loop %4.0f MHz (s_memtime / s_memrealtime), %.0f cycles per k-step"
          % (cb.KIND[kind], cin, cout, h, w, split, ms * 1e3, n // 22, us[0], us[1], us[2], us[3], mhz, v[5] / n / (cin / 2.0 / split)), flush=True)
