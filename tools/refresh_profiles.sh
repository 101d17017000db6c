#!/bin/bash
# Regenerates profiles/<round>/ on the GPU box: rocprofv3 kernel stats of the default bench command, two
# separate PMC passes (FETCH_SIZE, WRITE_SIZE) summarised per kernel, the per-layer HIP-event table and the
# bench lines of the other configurations.  Run through gpurun from the repo root:
#   gpurun --timeout 1100 -- 'bash tools/refresh_profiles.sh r01'
# Outputs land in gpurun_out/profiles_<round>/ (copy them into profiles/<round>/ afterwards).
set -e -o pipefail
R=${1:-r01}
OUT=gpurun_out/profiles_$R
mkdir -p $OUT /tmp/raw
export TMPDIR=/tmp
B="python bench.py --steps 48 --warmup 8 --cpu-frames 0"
timeout -k 10 300 python bench.py --steps 64 --warmup 8 > $OUT/bench_line_res50_1024.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/raw -o kt --output-format csv -- $B > $OUT/bench_under_rocprof.log 2>&1
cp /tmp/raw/kt_kernel_stats.csv $OUT/rocprofv3_kernel_stats_bench_res50_1024.csv
# 48 timed + 8 warm-up + 5 profiled frames in that command
python tools/rocprof_conv_summary.py $OUT/rocprofv3_kernel_stats_bench_res50_1024.csv 61 $OUT/bench_line_res50_1024.json > $OUT/rocprof_vs_bench.txt
P="python bench.py --steps 8 --warmup 2 --cpu-frames 0 --inflight 1 --profile-frames 1"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $C -d /tmp/raw -o pmc_$C --output-format csv -- $P > /tmp/raw/pmc_$C.log 2>&1
  python tools/summarize_pmc.py /tmp/raw/pmc_${C}_counter_collection.csv $OUT/pmc_${C}_by_kernel.csv
done
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/raw -o calib --output-format csv -- python tools/one_conv.py 0 0 1 256 256 256 128 > /tmp/raw/calib.log 2>&1
python tools/summarize_pmc.py /tmp/raw/calib_counter_collection.csv $OUT/pmc_FETCH_SIZE_calibration_1x1_67MB.csv
python tools/traffic_json.py $OUT/pmc_FETCH_SIZE_by_kernel.csv $OUT/pmc_WRITE_SIZE_by_kernel.csv 105 $OUT/conv_hbm_traffic.json $OUT/pmc_FETCH_SIZE_calibration_1x1_67MB.csv
timeout -k 10 300 python tools/profile_layers.py > $OUT/per_layer_hip_events_res50_1024.txt
timeout -k 10 300 python bench.py --steps 64 --warmup 8 --height 480 --width 640 --cpu-frames 2 > $OUT/bench_line_res50_640x480.json
timeout -k 10 300 python bench.py --steps 128 --warmup 16 --source 1080x1920 --height 480 --width 640 --cpu-frames 3 > $OUT/bench_line_res50_640x480_from_1080p.json
timeout -k 10 300 python bench.py --steps 32 --warmup 4 --height 1080 --width 1920 --cpu-frames 1 > $OUT/bench_line_res50_1920x1080.json
timeout -k 10 300 python bench.py --steps 64 --warmup 8 --arch try3 --cpu-frames 4 > $OUT/bench_line_try3_1024.json
timeout -k 10 300 python bench.py --steps 32 --warmup 4 --arch try3 --batch 8 --cpu-frames 3 > $OUT/bench_line_try3_1024_b8.json
timeout -k 10 300 python bench.py --arch facebox --batch 16 --steps 50 --warmup 5 > $OUT/bench_line_facebox_b16.json
tail -c 600 $OUT/bench_line_res50_1024.json
