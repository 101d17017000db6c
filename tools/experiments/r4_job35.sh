#!/bin/bash
set -e -o pipefail
OUT=gpurun_out/profiles_r04; mkdir -p $OUT /tmp/raw; export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu > $OUT/../r4_final_gpu_tests.log 2>&1 || { tail -30 $OUT/../r4_final_gpu_tests.log; exit 1; }
tail -2 $OUT/../r4_final_gpu_tests.log
timeout -k 10 300 python bench.py --steps 64 --warmup 8 --arch try3 > $OUT/bench_line_try3_1024.json
timeout -k 10 300 python bench.py --steps 32 --warmup 4 --arch try3 --batch 8 > $OUT/bench_line_try3_1024_b8.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/raw -o kttry3 --output-format csv -- python bench.py --steps 32 --warmup 4 --arch try3 --batch 8 --cpu-frames 0 --host-frames 0 > $OUT/bench_try3_b8_under_rocprof.log 2>&1
cp /tmp/raw/kttry3_kernel_stats.csv $OUT/rocprofv3_kernel_stats_bench_try3_1024_b8.csv
python -c "
import json
for f in ('bench_line_try3_1024.json','bench_line_try3_1024_b8.json'):
    d=json.loads(open('$OUT/'+f).read().strip().splitlines()[-1]); print(f, d['value'], d['roofline'].get('frac'), (d['roofline'].get('hbm_side') or {}).get('frac'), d['parity'])
"
