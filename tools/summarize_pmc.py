#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel name (count, sum, mean) -> small CSV.
    python tools/summarize_pmc.py IN.csv OUT.csv"""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: [0, 0.0])
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        k = (r["Kernel_Name"], r["Counter_Name"])
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Kernel_Name", "Counter_Name", "Dispatches", "Sum", "Mean"])
    for (kn, cn), (n, s) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([kn, cn, n, "%.6g" % s, "%.6g" % (s / n)])
