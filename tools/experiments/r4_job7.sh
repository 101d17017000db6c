#!/bin/bash
set -e
O=gpurun_out/r4j7; mkdir -p $O
T=face-detection-and-tracking_amd/tuned
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
# baseline line with the committed plan (new measurement fields)
timeout -k 10 600 python bench.py --steps 128 --warmup 16 --cpu-frames 0 --host-frames 0 > $O/bench_before.json 2> $O/bench_before.err
cp $T/res50_1024x1024_b1.plan $O/res50_1024x1024_b1.committed.plan
timeout -k 10 900 python bench.py --autotune 2 --save-plan 1 --tune-iters 8 --steps 32 --warmup 8 --cpu-frames 0 --host-frames 0 > $O/bench_autotune.json 2> $O/bench_autotune.err
cp $T/res50_1024x1024_b1.plan $O/res50_1024x1024_b1.autotuned.plan
python tools/experiments/merge_plan_kinds.py $O/res50_1024x1024_b1.committed.plan $O/res50_1024x1024_b1.autotuned.plan $O/res50_1024x1024_b1.merged.plan 16 17 | tee $O/merge.txt
cp $O/res50_1024x1024_b1.merged.plan $T/res50_1024x1024_b1.plan
timeout -k 10 600 python bench.py --steps 128 --warmup 16 --cpu-frames 0 --host-frames 0 > $O/bench_after.json 2> $O/bench_after.err
python - <<'PY'
import json
for n in ("before","autotune","after"):
    l=json.loads(open("gpurun_out/r4j7/bench_%s.json"%n).read().strip().splitlines()[-1])
    r=l["roofline"]
    print(n, l["value"], "dominant", r["frac"], "backbone", r["backbone"]["frac"], r["backbone"]["ms_per_frame"], "sum-of-events", r["backbone"]["frac_sum_of_per_launch_events"], r["backbone"]["ms_per_frame_sum_of_per_launch_events"], "conv_stack", r["conv_stack"]["ms_per_frame"], r["conv_stack"]["all_ops_contiguous_ms_per_frame"], l["parity"])
PY
