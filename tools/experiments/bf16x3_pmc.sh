#!/bin/bash
# split-bf16 products for the F(4x4) GEMMs -- rate, clock, error (tools/microbench/bf16x3_mfma.hip) and the
# co-execution counter of the forms with vector-ALU work beside the MFMAs (docs/EXPERIMENTS.md R5-4)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r5bf16; mkdir -p $O /tmp/raw
timeout -k 10 120 $R/tools/microbench/bf16x3_mfma.bin > $O/times_and_error.txt 2>&1
cat $O/times_and_error.txt
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d /tmp/raw -o bf --output-format csv -- $R/tools/microbench/bf16x3_mfma.bin rate > $O/pmc_times.txt 2>&1 \
  && python $R/tools/summarize_pmc.py /tmp/raw/bf_counter_collection.csv $O/pmc.csv
cat $O/pmc.csv | sed 's/(float\*, int, long long\*)//' | awk -F, '{print $1, $2, $3, $5}' | sort
