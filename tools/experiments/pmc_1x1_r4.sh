#!/bin/bash
# PMC of the 1x1 kernels (direct 128x64W vs persistent p128x64) on two backbone shapes: where do the waves spend their time?
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4pmc
mkdir -p $O /tmp/raw
rocprofv3 -L > $O/counters_list.txt 2>&1 || true
pass() {  # name, counters..., then -- kind tile split cin h w cout res
  local name=$1; shift
  local ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc "${ctr[@]}" -d /tmp/raw -o $name --output-format csv -- python $R/tools/one_conv.py "$@" 6 > $O/${name}_times.txt 2>&1 \
    && python $R/tools/summarize_pmc.py /tmp/raw/${name}_counter_collection.csv $O/pmc_$name.csv || echo "pass $name failed" >> $O/failed.txt
}
for K in "l2c3_direct 0 6 1 128 128 128 512 1" "l2c3_pers 16 34 1 128 128 128 512 1" "l3c1_direct 10 9 1 1024 64 64 256 0" "l1c3_direct 0 6 1 64 256 256 256 1" "l1c3_pers 16 34 1 64 256 256 256 1"; do
  set -- $K
  n=$1; shift
  pass ${n}_mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -- "$@"
  pass ${n}_wait SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_WAVES -- "$@"
  pass ${n}_inst SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -- "$@"
done
ls $O
