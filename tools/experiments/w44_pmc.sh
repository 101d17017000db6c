#!/bin/bash
# PMC of the Winograd F(4x4,3x3) kernel beside the F(2x2,3x3) kernel on 256 -> 256 @ 256x256 (run on the GPU box).
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3e
mkdir -p $O /tmp/raw
pass() {  # name, counters..., then -- kind tile split
  local name=$1; shift
  local ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc "${ctr[@]}" -d /tmp/raw -o $name --output-format csv -- python $R/tools/one_conv.py $1 $2 $3 256 256 256 256 0 6 > $O/${name}_times.txt 2>&1 \
    && python $R/tools/summarize_pmc.py /tmp/raw/${name}_counter_collection.csv $O/pmc_$name.csv || echo "pass $name failed" >> $O/failed.txt
}
for K in "w44 14 32 1" "w22 8 30 1"; do
  set -- $K
  pass ${1}_mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -- $2 $3 $4
  pass ${1}_inst SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -- $2 $3 $4
  pass ${1}_wait SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL -- $2 $3 $4
done
ls $O
