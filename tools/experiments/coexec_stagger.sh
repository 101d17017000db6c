#!/bin/bash
# VALU / MFMA co-execution with and without a half-block stagger of the SIMD partners (tools/microbench/coexec_stagger.hip)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r4coexec; mkdir -p $O /tmp/raw
$R/tools/microbench/coexec_stagger.bin > $O/times.txt 2>&1
cat $O/times.txt
cd /tmp
for m in 0 1 2; do
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d /tmp/raw -o cx_$m --output-format csv -- $R/tools/microbench/coexec_stagger.bin $m > $O/pmc_mode${m}_times.txt 2>&1 \
    && python $R/tools/summarize_pmc.py /tmp/raw/cx_${m}_counter_collection.csv $O/pmc_mode$m.csv
  echo "== mode $m"; grep -v Kernel_Name $O/pmc_mode$m.csv | sed 's/.*float)",//' | sort | awk -F, '{print $1, $2, $4}' | paste - - - - - - - -
done
