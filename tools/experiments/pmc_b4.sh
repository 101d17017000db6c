#!/bin/bash
# matrix-pipe occupancy of the production kernels at the launch size the default bench runs (four frames per launch): the
# batch-4 plan's (class, tile, split) of the dominant F(4x4) layer and of the three 1x1 layers round 3's review named
set -e -o pipefail
OUT=gpurun_out/pmc_b4; mkdir -p $OUT /tmp/raw; export TMPDIR=/tmp
PLAN=face-detection-and-tracking_amd/tuned/res50_1024x1024_b4.plan
plan_of() { awk -v L="$1" '$1==L {print $2, $3, $4; f=1} END {if(!f) exit 1}' $PLAN; }
pmc_one() {
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
    -d /tmp/raw -o b4_$1 --output-format csv -- python tools/one_conv.py $2 $3 $4 $5 $6 $7 $8 $9 8 4 > $OUT/pmc_b4_$1_times.txt 2>&1
  python tools/summarize_pmc.py /tmp/raw/b4_$1_counter_collection.csv $OUT/pmc_b4_$1.csv
  grep -h "TF/s" $OUT/pmc_b4_$1_times.txt
  python - <<PY
import csv
r={x['Counter_Name']:float(x['Mean']) for x in csv.DictReader(open('$OUT/pmc_b4_$1.csv')) if float(x['Dispatches'])>1}
print('   $1: matrix pipe busy %.1f %% of the CU-busy cycles; CU busy %.0f %% of the GPU-active cycles' % (100*r['SQ_VALU_MFMA_BUSY_CYCLES']/(4*r['SQ_BUSY_CU_CYCLES']), 100*r['SQ_BUSY_CU_CYCLES']/256/(r['GRBM_GUI_ACTIVE']/8)))
PY
}
pmc_one wino44_256to256_256x256  $(plan_of conv2_SSH.conv1) 256 256 256 256 0
pmc_one 1x1_1024to256_64x64      $(plan_of layer3.1.conv1) 1024 64 64 256 0
pmc_one 1x1_64to256_256x256_res  $(plan_of layer1.0.conv3) 64 256 256 256 1
pmc_one 1x1_128to512_128x128_res $(plan_of layer2.1.conv3) 128 128 128 512 1
pmc_one 1x1_256to64_256x256      $(plan_of layer1.1.conv1) 256 256 256 64 0
