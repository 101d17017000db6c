#!/bin/bash
# LDS bank conflicts of the strided direct convs (stride-4 FaceBoxes conv1, stride-2 stems / 3x3): is the patch's column stride what bounds them?
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r4strided; mkdir -p $O /tmp/raw
cd /tmp
#         name   kind tile split cin h w cout
for K in "fb_conv1 6 2 1 3 1024 1024 24" "fb_conv2 7 1 1 48 128 128 64" "stem 5 6 1 3 1024 1024 64" "l2_3x3s2 4 3 1 128 256 256 128" "l3_3x3s2 4 6 4 256 128 128 256"; do
  set -- $K; n=$1; shift
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d /tmp/raw -o $n --output-format csv -- python $R/tools/one_conv.py "$@" 0 6 > $O/${n}_times.txt 2>&1 \
    && python $R/tools/summarize_pmc.py /tmp/raw/${n}_counter_collection.csv $O/pmc_$n.csv
  echo "== $n: $(tail -1 $O/${n}_times.txt)"
  grep "conv_kernel" $O/pmc_$n.csv | sed 's/.*ConvArgs)",//' | awk -F, '{printf "   %s mean %s\n", $1, $4}'
done
