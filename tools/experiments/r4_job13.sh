#!/bin/bash
set -e
O=gpurun_out/r4j13; mkdir -p $O
timeout -k 10 600 python bench.py --steps 64 --warmup 8 > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
timeout -k 10 600 python bench.py --steps 20 --warmup 3 --cpu-frames 0 --host-frames 0 > $O/bench_driver_like.json 2> $O/bench_driver_like.err
FDT_BENCH_PIPELINE=torch timeout -k 10 600 python bench.py --steps 64 --warmup 8 --cpu-frames 0 --host-frames 0 > $O/bench_torch_pipeline.json 2> $O/bench_torch_pipeline.err
timeout -k 10 600 python bench.py --height 480 --width 640 --steps 256 --warmup 32 --cpu-frames 0 --host-frames 0 > $O/bench_640x480.json 2> $O/bench_640x480.err
FDT_BENCH_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 2 --steps 20 --warmup 4 > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err
python - <<'PY'
import json
for n in ("default","driver_like","torch_pipeline","640x480","2rank_gloo"):
    l=json.loads(open("gpurun_out/r4j13/bench_%s.json"%n).read().strip().splitlines()[-1])
    r=l["roofline"]
    print(n, l["value"], l["ms_per_step"], l["config"].get("timed_loop","")[:20], "| dominant", r["frac"], "backbone", r["backbone"]["frac"] if r.get("backbone") else None, "|", l["parity"], "| host", (l.get("host_path") or {}).get("value"), "| cpu", (l.get("cpu_baseline") or {}).get("value"))
PY
