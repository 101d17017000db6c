#!/bin/bash
# host-frame engines A/B: fdt_pipeline_step_host vs the ticket interface stepped from Python
set -e
mkdir -p gpurun_out/r4j20
python -m pytest tests/test_gpu_entry.py tests/test_gpu_cabi_pipeline.py -x -q -m gpu > gpurun_out/r4j20/tests.log 2>&1 || { tail -30 gpurun_out/r4j20/tests.log; exit 1; }
tail -3 gpurun_out/r4j20/tests.log
for nf in 3 4 6; do
  FDT_HOST_PIPE_NF=$nf python bench.py --steps 200 --warmup 20 --cpu-frames 1 --cpu-threads 64 --host-frames 256 > gpurun_out/r4j20/bench_nf$nf.json 2> gpurun_out/r4j20/bench_nf$nf.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4j20/bench_nf$nf.json").read().strip().splitlines()[-1])
print("nf=$nf value", d["value"], "host", json.dumps(d["config"].get("host_path") or d.get("host_path"))[:600])
PY
done
python bench.py --height 480 --width 640 --source 1080x1920 --group 1 --steps 200 --warmup 20 --cpu-frames 1 --cpu-threads 64 --host-frames 256 > gpurun_out/r4j20/bench_c4.json 2> gpurun_out/r4j20/bench_c4.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r4j20/bench_c4.json").read().strip().splitlines()[-1])
print("c4 g1 value", d["value"], "host", json.dumps(d["config"].get("host_path") or d.get("host_path"))[:600])
PY
