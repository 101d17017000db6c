#!/usr/bin/env python3
"""Plan variants that differ only in the workgroup map (5th column) of chosen layers:
    python tools/experiments/plan_map_variants.py COMMITTED.plan OUTDIR
V1: the Winograd 3x3 layers on the 256^2 / 128^2 maps -> CONV_MAP_XCD_REGION (3); V2: V1 + the 1x1 layers on those maps."""
import os, sys
src, outdir = sys.argv[1:3]
big3 = ("smooth_c3", "conv2_SSH.", "smooth_c4", "conv3_SSH.", "layer1.0.conv2", "layer1.1.conv2", "layer1.2.conv2", "layer2.1.conv2",
        "layer2.2.conv2", "layer2.3.conv2")
big1 = ("layer1.", "layer2.", "conv3_ct_py.", "conv4_ct_py.up", "layer3.0.conv1")
lines = open(src).read().splitlines()
def variant(rule):
    out = []
    for ln in lines:
        p = ln.split()
        if len(p) >= 5 and p[0] != "shape" and rule(p):
            p[4] = "3"
            ln = " ".join(p)
        out.append(ln)
    return "\n".join(out) + "\n"
is3 = lambda p: p[1] in ("8", "9", "14", "15") and p[0].startswith(big3)
is1 = lambda p: p[1] in ("0", "10", "11") and p[0].startswith(big1) and "conv2" not in p[0]
open(os.path.join(outdir, "v1.plan"), "w").write(variant(is3))
open(os.path.join(outdir, "v2.plan"), "w").write(variant(lambda p: is3(p) or is1(p)))
