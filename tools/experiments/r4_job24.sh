#!/bin/bash
# the one-slot rocprof pass of tools/refresh_profiles.sh alone
set -e -o pipefail
OUT=gpurun_out/profiles_r04; mkdir -p $OUT /tmp/raw; export TMPDIR=/tmp
B="python bench.py --steps 48 --warmup 8 --cpu-frames 0 --host-frames 0 --ungrouped-steps 0"
timeout -k 10 400 python bench.py --steps 64 --warmup 8 > $OUT/bench_line_res50_1024.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/raw -o kt1 --output-format csv -- $B --inflight 1 > $OUT/bench_inflight1_under_rocprof.log 2>&1
cp /tmp/raw/kt1_kernel_stats.csv $OUT/rocprofv3_kernel_stats_bench_res50_1024_inflight1.csv
python tools/rocprof_conv_summary.py $OUT/rocprofv3_kernel_stats_bench_res50_1024_inflight1.csv 60 $OUT/bench_line_res50_1024.json | tee $OUT/rocprof_vs_bench_inflight1.txt
