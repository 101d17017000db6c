#!/bin/bash
# Regenerates profiles/<round>/ on the GPU box: rocprofv3 kernel stats of the default bench command, two separate PMC
# passes (FETCH_SIZE, WRITE_SIZE) summarised per kernel, MFMA-busy PMC of the dominant Winograd kernel and of the three
# 1x1 instantiations that take the most time, the per-layer HIP-event table and the bench lines of the other
# configurations.  Run through gpurun from the repo root:
#   gpurun --timeout 1200 -- 'bash tools/refresh_profiles.sh r05 a'      (rocprofv3 stats, traffic and PMC passes)
#   gpurun --timeout 1200 -- 'bash tools/refresh_profiles.sh r05 b'      (per-layer tables and the bench lines of every configuration)
# Outputs land in gpurun_out/profiles_<round>/ (copy them into profiles/<round>/ afterwards).
set -e -o pipefail
R=${1:-r05}
PART=${2:-all}
OUT=gpurun_out/profiles_$R
mkdir -p $OUT /tmp/raw
export TMPDIR=/tmp
PLAN=face-detection-and-tracking_amd/tuned/res50_1024x1024_b4.plan     # the default bench runs four frames per launch chain (--group auto)
# the single-kernel PMC passes below measure the (kernel class, tile, split-K) the COMMITTED plan runs for these layers
# (read from the plan, so a re-tune cannot leave the PMC files describing a kernel that is no longer used)
plan_of() { awk -v L="$1" '$1==L {print $2, $3, $4; f=1} END {if(!f) exit 1}' $PLAN || { echo "refresh_profiles: layer $1 not in $PLAN" >&2; exit 1; }; }
K_WINO=$(plan_of conv2_SSH.conv1)
K_A=$(plan_of layer3.1.conv1)
K_B=$(plan_of layer1.0.conv3)
K_C=$(plan_of layer2.1.conv3)
case "$K_WINO" in "14 32 "*|"14 33 "*) ;; *) echo "refresh_profiles: conv2_SSH.conv1 is no longer a Winograd F(4x4,3x3) kernel ($K_WINO)" >&2; exit 1;; esac
K_WD2=$(plan_of conv2_SSH.conv2)       # the dilated SSH context conv: quarter-split F(2x2,3x3)
B="python bench.py --steps 48 --warmup 8 --cpu-frames 0 --host-frames 0 --ungrouped-steps 0 --latency-frames 0"
if [ "$PART" != "b" ]; then
timeout -k 10 400 python bench.py --steps 64 --warmup 8 > $OUT/bench_line_res50_1024.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/raw -o kt --output-format csv -- $B > $OUT/bench_under_rocprof.log 2>&1
cp /tmp/raw/kt_kernel_stats.csv $OUT/rocprofv3_kernel_stats_bench_res50_1024.csv
# every forward of that command is a four-frame forward (priming, 56 / 4 timed + warm-up groups, the sequential parity re-run, the
# per-op profiled and the segment-timed forwards); their number is read from the head_finalize_all_kernel calls
python tools/rocprof_conv_summary.py $OUT/rocprofv3_kernel_stats_bench_res50_1024.csv 60 $OUT/bench_line_res50_1024.json > $OUT/rocprof_vs_bench.txt
# The default command keeps eight slots in flight: two F(4x4) launches of different slots share the CUs and each one's duration in
# the trace above is longer than alone (their sum per frame exceeds the step time).  The SAME command with one slot (--inflight 1:
# launches back to back on one stream) is what bench.py's per-launch figures (serial profile pass) are comparable with:
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/raw -o kt1 --output-format csv -- $B --inflight 1 > $OUT/bench_inflight1_under_rocprof.log 2>&1
cp /tmp/raw/kt1_kernel_stats.csv $OUT/rocprofv3_kernel_stats_bench_res50_1024_inflight1.csv
python tools/rocprof_conv_summary.py $OUT/rocprofv3_kernel_stats_bench_res50_1024_inflight1.csv 60 $OUT/bench_line_res50_1024.json > $OUT/rocprof_vs_bench_inflight1.txt
# HBM traffic by request size class, per dispatch (tools/experiments/traffic_r4.sh has the calibration passes)
P="python bench.py --steps 32 --warmup 8 --cpu-frames 0 --host-frames 0 --inflight 1 --profile-frames 1 --graph 0 --ungrouped-steps 0 --latency-frames 0"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d /tmp/raw -o fwd_rd --output-format csv -- $P > /tmp/raw/fwd_rd.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d /tmp/raw -o fwd_wr --output-format csv -- $P > /tmp/raw/fwd_wr.log 2>&1
python tools/dump_ops.py --batch 4 > $OUT/ops_1024_b4.json 2> /tmp/raw/dump_ops.err
python tools/traffic_by_class.py /tmp/raw/fwd_rd_counter_collection.csv /tmp/raw/fwd_wr_counter_collection.csv $OUT/ops_1024_b4.json $OUT/conv_hbm_traffic_b4.json - 4 > $OUT/conv_hbm_traffic_b4_summary.txt
# the same two passes for the one-frame-per-launch form (--group 1, the batch-1 plan; calibration passes: tools/experiments/traffic_r4.sh)
P1="$P --group 1"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d /tmp/raw -o fwd1_rd --output-format csv -- $P1 > /tmp/raw/fwd1_rd.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d /tmp/raw -o fwd1_wr --output-format csv -- $P1 > /tmp/raw/fwd1_wr.log 2>&1
python tools/dump_ops.py > $OUT/ops_1024.json 2> /tmp/raw/dump_ops1.err
CAL="-"; [ -d profiles/$R/traffic ] && CAL=profiles/$R/traffic
python tools/traffic_by_class.py /tmp/raw/fwd1_rd_counter_collection.csv /tmp/raw/fwd1_wr_counter_collection.csv $OUT/ops_1024.json $OUT/conv_hbm_traffic.json $CAL 1 > $OUT/conv_hbm_traffic_summary.txt
# matrix-pipe occupancy of the production kernels (one kernel per process; SQ counters + GRBM in one pass)
#        name                       kind tile split cin  h   w  cout res
pmc_one() {
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
    -d /tmp/raw -o mfma_$1 --output-format csv -- python tools/one_conv.py $2 $3 $4 $5 $6 $7 $8 $9 8 > $OUT/pmc_mfma_$1_times.txt 2>&1 || \
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
    -d /tmp/raw -o mfma_$1 --output-format csv -- python tools/one_conv.py $2 $3 $4 $5 $6 $7 $8 $9 8 > $OUT/pmc_mfma_$1_times.txt 2>&1
  python tools/summarize_pmc.py /tmp/raw/mfma_$1_counter_collection.csv $OUT/pmc_mfma_$1.csv
}
pmc_one wino44_256to256_256x256  $K_WINO 256 256 256 256 0    # conv2_SSH.conv1 / smooth_c3: Winograd F(4x4,3x3)
pmc_one wino4d2_256to128_256x256 $K_WD2 256 256 256 128 0     # conv2_SSH.conv2: dilated, quarter-split F(2x2,3x3)
pmc_one 1x1_1024to256_64x64      $K_A 1024 64 64 256 0        # layer3.x.conv1
pmc_one 1x1_64to256_256x256_res  $K_B 64 256 256 256 1        # layer1.x.conv3 (+ residual)
pmc_one 1x1_128to512_128x128_res $K_C 128 128 128 512 1       # layer2.x.conv3 (+ residual)
echo "conv2_SSH.conv1 $K_WINO | conv2_SSH.conv2 $K_WD2 | layer3.1.conv1 $K_A | layer1.0.conv3 $K_B | layer2.1.conv3 $K_C" > $OUT/pmc_mfma_kernels.txt
# the vector-ALU kernel of the 8-channel heads (conv_n8.h): VALU / LDS activity instead of matrix-pipe occupancy
K_HEAD=$(plan_of face_loc.0)
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
  -d /tmp/raw -o valu_head --output-format csv -- python tools/one_conv.py $K_HEAD 512 256 256 8 0 8 > $OUT/pmc_valu_head_512to8_256x256_times.txt 2>&1 && \
  python tools/summarize_pmc.py /tmp/raw/valu_head_counter_collection.csv $OUT/pmc_valu_head_512to8_256x256.csv || echo "head PMC pass failed (counters unavailable)" > $OUT/pmc_valu_head_512to8_256x256.csv
echo "face_loc.0 $K_HEAD" >> $OUT/pmc_mfma_kernels.txt
fi
if [ "$PART" != "a" ]; then
timeout -k 10 300 python tools/profile_layers.py > $OUT/per_layer_hip_events_res50_1024.txt
timeout -k 10 300 python tools/profile_layers.py --batch 4 > $OUT/per_layer_hip_events_res50_1024_b4.txt
timeout -k 10 300 python bench.py --steps 256 --warmup 32 --group 1 --cpu-frames 0 --host-frames 0 > $OUT/bench_line_res50_1024_group1.json
timeout -k 10 300 python bench.py --steps 256 --warmup 32 --height 480 --width 640 > $OUT/bench_line_res50_640x480.json
timeout -k 10 300 python bench.py --steps 256 --warmup 32 --height 480 --width 640 --group 1 --cpu-frames 0 --host-frames 0 > $OUT/bench_line_res50_640x480_group1.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/raw -o kt480 --output-format csv -- python bench.py --steps 64 --warmup 8 --height 480 --width 640 --cpu-frames 0 --host-frames 0 --ungrouped-steps 0 --latency-frames 0 > $OUT/bench_640x480_under_rocprof.log 2>&1
cp /tmp/raw/kt480_kernel_stats.csv $OUT/rocprofv3_kernel_stats_bench_res50_640x480.csv
timeout -k 10 200 python tools/profile_layers.py --height 480 --width 640 > $OUT/per_layer_hip_events_res50_640x480.txt
timeout -k 10 300 python bench.py --steps 128 --warmup 16 --source 1080x1920 --height 480 --width 640 --host-frames 256 > $OUT/bench_line_res50_640x480_from_1080p.json
timeout -k 10 300 python bench.py --steps 32 --warmup 4 --height 1080 --width 1920 --cpu-frames 4 --host-frames 32 > $OUT/bench_line_res50_1920x1080.json
timeout -k 10 300 python bench.py --steps 64 --warmup 8 --arch try3 > $OUT/bench_line_try3_1024.json
timeout -k 10 300 python bench.py --steps 32 --warmup 4 --arch try3 --batch 8 > $OUT/bench_line_try3_1024_b8.json
timeout -k 10 300 python bench.py --arch facebox --batch 16 --steps 100 --warmup 8 > $OUT/bench_line_facebox_4k_b16.json
timeout -k 10 400 python bench.py --steps 32 --warmup 6 --batch 2 --cpu-frames 0 > $OUT/bench_line_res50_1024_b2.json
timeout -k 10 300 python bench.py --steps 48 --warmup 8 --height 480 --width 640 --batch 4 --cpu-frames 0 > $OUT/bench_line_res50_640x480_b4.json
timeout -k 10 300 python bench.py --steps 256 --warmup 32 --height 480 --width 640 --group 4 --cpu-frames 0 --host-frames 0 --ungrouped-steps 0 > $OUT/bench_line_res50_640x480_group4.json
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line_res50_1024_driver_style.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/raw -o ktfb --output-format csv -- python bench.py --arch facebox --batch 16 --steps 50 --warmup 5 --cpu-frames 0 > $OUT/bench_facebox_under_rocprof.log 2>&1
cp /tmp/raw/ktfb_kernel_stats.csv $OUT/rocprofv3_kernel_stats_bench_facebox_4k_b16.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/raw -o kttry3 --output-format csv -- python bench.py --steps 32 --warmup 4 --arch try3 --batch 8 --cpu-frames 0 --host-frames 0 --latency-frames 0 > $OUT/bench_try3_b8_under_rocprof.log 2>&1
cp /tmp/raw/kttry3_kernel_stats.csv $OUT/rocprofv3_kernel_stats_bench_try3_1024_b8.csv
fi
ls $OUT | wc -l
