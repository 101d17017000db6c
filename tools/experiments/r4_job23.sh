#!/bin/bash
set -e
O=gpurun_out/r4j23; mkdir -p $O
FDT_LIB=$PWD/tools/experiments/w44_libs/libfdt_hip_stamps.so python tools/experiments/w44_stamps.py 2>&1 | grep -v amdgpu.ids | tee $O/w44_stamps.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python -c "import json;d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]);print('default', d['value'], d['config']['frames_grouped_per_launch'], d['roofline']['frac'], d['roofline']['backbone']['frac'], d['parity'], d['host_path']['engines'], d['ungrouped'])"
FDT_BENCH_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --steps 64 --warmup 8 > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err
tail -c 1200 $O/bench_2rank_gloo.json
