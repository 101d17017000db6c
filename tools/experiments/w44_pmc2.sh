#!/bin/bash
# second PMC look at the F(4x4) kernel (8-wave form, 12-wave form) and the F(2x2) kernel on 256 -> 256 @ 256x256: FIFO-full and co-execution counters
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3h
mkdir -p $O /tmp/raw
pass() {
  local name=$1; shift
  local ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc "${ctr[@]}" -d /tmp/raw -o $name --output-format csv -- python $R/tools/one_conv.py $1 $2 $3 256 256 256 256 0 6 > $O/${name}_times.txt 2>&1 \
    && python $R/tools/summarize_pmc.py /tmp/raw/${name}_counter_collection.csv $O/pmc_$name.csv || echo "pass $name failed" >> $O/failed.txt
}
for K in "w44 14 32 1" "w44b 14 33 1" "w22 8 30 1"; do
  set -- $K
  pass ${1}_fifo SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE -- $2 $3 $4
  pass ${1}_lvl SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -- $2 $3 $4
done
ls $O; cat $O/failed.txt 2>/dev/null
