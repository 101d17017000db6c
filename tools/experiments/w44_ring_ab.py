#!/usr/bin/env python3
"""conv_wino44_kernel: four vs six patch ring slots (libraries built from the two versions of conv_wino44.h), same process order
alternated by the calling script."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import conv_bench as cb
out = []
for kind, cin, h, w, cout, split, B in ((14, 256, 256, 256, 256, 1, 1), (14, 256, 256, 256, 256, 1, 4), (14, 512, 128, 128, 512, 1, 4), (14, 1024, 64, 64, 1024, 2, 4),
                                        (15, 256, 256, 256, 128, 1, 4), (14, 2048, 32, 32, 256, 8, 4), (14, 128, 128, 128, 128, 1, 4)):
    ms = min(cb.bench(kind, 32, split, cin, h, w, cout, iters=20, B=B) for _ in range(2))
    out.append("%6.1f" % (ms * 1e3))
print(os.environ.get("W44_LABEL", ""), " ".join(out), flush=True)
