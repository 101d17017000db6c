#!/usr/bin/env python3
"""Cross-check of bench.py's roofline figures against a rocprofv3 --kernel-trace --stats run of the same command:
sums the conv kernels (all conv_kernel / conv_wino* instantiations) and the split-K reduce passes that finish them.
    python tools/rocprof_conv_summary.py KERNEL_STATS.csv FRAMES [BENCH_LINE.json]"""
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
forwards = float(sys.argv[2])
once = [r for r in rows if "head_finalize_all_kernel" in r["Name"]]     # launched exactly once per forward
if once:
    forwards = float(sum(int(r["Calls"]) for r in once))                # (the argument is only the fallback)
line = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1]) if len(sys.argv) > 3 else None
# frames per forward: the cross-frame group (or --batch) of the bench line; every forward of the profiled command runs it
# (the command is run with --ungrouped-steps 0)
G = (line["config"].get("frames_grouped_per_launch") or line["config"].get("batch_per_gpu") or 1) if line else 1
frames = forwards * G
conv = [r for r in rows if any(k in r["Name"] for k in ("conv_kernel", "conv_wino", "conv1x1p_kernel", "conv_stem_u8_kernel", "conv_stem_s4_kernel", "conv_n8_kernel", "conv_b3_kernel", "conv_stem_s4_b3_kernel", "conv_stem_u8b_kernel", "conv1x1p_b3_kernel"))]
red = [r for r in rows if "splitk_reduce" in r["Name"]]
tot = lambda rs: sum(float(r["TotalDurationNs"]) for r in rs)
calls = lambda rs: sum(int(r["Calls"]) for r in rs)
print("forwards in the run                    : %g  (x %d frames per forward = %g frames)" % (forwards, G, frames))
print("conv kernel launches per forward       : %.1f  (%d template instantiations)" % (calls(conv) / forwards, len(conv)))
print("conv kernels, ms per frame             : %.3f  (avg %.2f us per launch)" % (tot(conv) / frames / 1e6,
                                                                              tot(conv) / calls(conv) / 1e3))
print("split-K / upsample-add reduce, ms/frame: %.3f  (%.1f launches per forward)" % (tot(red) / frames / 1e6,
                                                                                   calls(red) / forwards))
print("conv + reduce, ms per frame            : %.3f" % ((tot(conv) + tot(red)) / frames / 1e6))
if line:
    d = line
    r = d["roofline"]
    cs = r["conv_stack"]
    print("bench.py (HIP events, same ops)        : %.3f ms per frame over %d launches; %.2f TFLOP/s executed, %.2f algorithmic" %
          (cs["ms_per_frame"], cs.get("launches_per_forward", cs.get("launches_per_frame")), cs["achieved_executed"], cs["achieved_algorithmic"]))
    ms = (tot(conv) + tot(red)) / frames / 1e6
    print("from the rocprof durations             : %.2f TFLOP/s executed (%.3f GFLOP/frame), %.2f algorithmic (%.3f GFLOP/frame)" %
          (cs["executed_gflop_per_frame"] / ms, cs["executed_gflop_per_frame"], cs["algorithmic_gflop_per_frame"] / ms,
           cs["algorithmic_gflop_per_frame"]))
    dom = max(conv, key=lambda x: float(x["TotalDurationNs"]))
    print("dominant kernel by rocprof time        : %s" % dom["Name"])
    print("   calls/forward %.1f, average %.2f us (bench.py: %s, %d launches/forward, average %.2f us)" %
          (int(dom["Calls"]) / forwards, float(dom["TotalDurationNs"]) / int(dom["Calls"]) / 1e3, r["kernel"],
           r.get("launches_per_forward", r.get("launches_per_frame")), r["avg_launch_us"]))
