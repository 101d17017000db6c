#!/bin/bash
# PMC of the split-bf16 1x1 kernel on three backbone shapes at batch 4: where do the waves spend their time? (R5-11)
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5pmc
mkdir -p $O /tmp/raw
pass() {  # name, counters..., then -- kind tile split cin h w cout res
  local name=$1; shift
  local ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc "${ctr[@]}" -d /tmp/raw -o $name --output-format csv -- python $R/tools/one_conv.py "$@" 6 4 > $O/${name}_times.txt 2>&1 \
    && python $R/tools/summarize_pmc.py /tmp/raw/${name}_counter_collection.csv $O/pmc_$name.csv || echo "pass $name failed" >> $O/failed.txt
}
for K in "l3c1 21 12 1 1024 64 64 256 0" "l3c3 21 12 1 256 64 64 1024 1" "fc 21 11 1 2048 32 32 2048 0" "l2c1 21 11 1 512 128 128 128 0"; do
  set -- $K
  n=$1; shift
  pass ${n}_mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -- "$@"
  pass ${n}_wait SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_WAVES -- "$@"
  pass ${n}_inst SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -- "$@"
done
for f in $O/pmc_*.csv; do echo "== $f"; grep conv_b3 $f | sed 's/fdt::(anonymous namespace):://; s/(fdt::ConvArgs)//' ; done > $O/summary.txt
cat $O/*_mfma_times.txt | grep "TF/s" >> $O/summary.txt
