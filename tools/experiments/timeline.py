#!/usr/bin/env python3
"""GPU timeline statistics from a rocprofv3 --kernel-trace CSV: wall time, union of busy intervals, time with >=2 kernels
in flight, per-kernel-class totals.   python tools/timeline.py kernel_trace.csv [t0_frac t1_frac]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
lo, hi = ev[0][0], max(e[1] for e in ev)
f0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
f1 = float(sys.argv[3]) if len(sys.argv) > 3 else 0.6
w0, w1 = lo + (hi - lo) * f0, lo + (hi - lo) * f1
pts = []
for s, e, n in ev:
    s, e = max(s, w0), min(e, w1)
    if e > s:
        pts.append((s, 1)); pts.append((e, -1))
pts.sort()
busy = multi = 0
depth, last = 0, w0
hist = collections.Counter()
for t, d in pts:
    if depth >= 1: busy += t - last
    if depth >= 2: multi += t - last
    hist[min(depth, 4)] += t - last
    depth += d; last = t
hist[0] += w1 - last
wall = w1 - w0
print("window %.1f ms: busy (>=1 kernel) %.1f%%, >=2 kernels %.1f%%" % (wall / 1e6, 100 * busy / wall, 100 * multi / wall))
print("time by #kernels in flight:", {k: "%.1f%%" % (100 * v / wall) for k, v in sorted(hist.items())})
# per-bucket view (20 ms): busy %, >=2 %
B = 20e6
nb = int((hi - lo) / B) + 1
busy_b = [0.0] * nb; multi_b = [0.0] * nb
pts = []
for s, e, n in ev:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
depth, last = 0, lo
for t, d in pts:
    a = last
    while a < t:
        b = min(t, lo + (int((a - lo) / B) + 1) * B)
        k = int((a - lo) / B)
        if depth >= 1: busy_b[k] += b - a
        if depth >= 2: multi_b[k] += b - a
        a = b
    depth += d; last = t
print("bucket(20ms): busy% / >=2%")
print(" ".join("%d/%d" % (100 * x / B, 100 * y / B) for x, y in zip(busy_b, multi_b)))
