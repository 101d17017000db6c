#!/usr/bin/env python3
"""Take from a freshly autotuned plan only the entries of the given kernel classes (by kind index) and put them into the
committed plan; everything else keeps its committed entry (the tuner's picks wobble from run to run on near-ties).
    python tools/experiments/merge_plan_kinds.py COMMITTED.plan NEW.plan OUT.plan 16 17"""
import sys
old, new, out = sys.argv[1:4]
kinds = set(sys.argv[4:])
n = {}
for ln in open(new):
    p = ln.split()
    if len(p) >= 4 and p[0] != "shape":
        n[p[0]] = ln
res, changed = [], []
for ln in open(old):
    p = ln.split()
    if len(p) >= 4 and p[0] != "shape" and p[0] in n and n[p[0]].split()[1] in kinds:
        res.append(n[p[0]])
        changed.append((p[0], ln.strip(), n[p[0]].strip()))
    else:
        res.append(ln)
open(out, "w").writelines(res)
for c in changed:
    print("%-28s %-24s -> %s" % (c[0], c[1].split(None, 1)[1], c[2].split(None, 1)[1]))
print("%d entries changed" % len(changed))
