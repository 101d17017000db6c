#!/bin/bash
set -e
O=gpurun_out/r4j8; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_model.py -m gpu -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
for g in 1 4; do
  timeout -k 10 600 python bench.py --height 480 --width 640 --group $g --steps 256 --warmup 32 --cpu-frames 0 --host-frames 0 > $O/bench_640x480_g$g.json 2> $O/bench_640x480_g$g.err
done
timeout -k 10 600 python bench.py --height 480 --width 640 --steps 20 --warmup 3 --cpu-frames 0 --host-frames 0 > $O/bench_640x480_driver_like.json 2> $O/bench_640x480_driver_like.err
timeout -k 10 600 python bench.py --height 480 --width 640 --source 1080x1920 --steps 256 --warmup 32 --cpu-frames 0 --host-frames 0 > $O/bench_640x480_from1080p.json 2> $O/bench_640x480_from1080p.err
python - <<'PY'
import json
for n in ("g1","g4","driver_like","from1080p"):
    l=json.loads(open("gpurun_out/r4j8/bench_640x480_%s.json"%n).read().strip().splitlines()[-1])
    r=l["roofline"]
    print(n, l["value"], l["ms_per_step"], "group", l["config"]["frames_grouped_per_launch"], "dominant", r["kernel"], r["frac"], "timed", r["timed_step"]["frac_executed"], l["parity"])
PY
