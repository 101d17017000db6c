#!/bin/bash
# A/B: conv_wino44_kernel's input transform with v_pk_fma_f32 / v_pk_add_f32 (production) against the
# same transform with two scalar v_fma_f32 / v_add_f32 each (make EXTRA=-DFDT_W44_NOPK), everything else identical.
# Libraries are built on the build host (hipcc cross-compiles):
#   D=face-detection-and-tracking_amd/csrc; touch $D/conv_wino44.h; make -C $D -j8 EXTRA=-DFDT_W44_NOPK; cp $D/libfdt_hip.so tools/experiments/w44_libs/libfdt_hip_nopk.so
#   touch $D/conv_wino44.h; make -C $D -j8; cp $D/libfdt_hip.so tools/experiments/w44_libs/libfdt_hip_pk.so
# then on the GPU box:  bash tools/experiments/w44_pk_ab.sh
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r4pk; mkdir -p $O /tmp/raw
cd $R
for rep in 1 2 3; do
  for v in pk nopk; do
    for S in "14 32 1 256 256 256 256" "14 32 1 512 128 128 512" "15 32 1 256 256 256 128"; do
      echo -n "rep $rep $v: " >> $O/times.txt
      FDT_LIB=$R/tools/experiments/w44_libs/libfdt_hip_$v.so python tools/one_conv.py $S 0 20 2>/dev/null | tail -1 >> $O/times.txt
    done
  done
done
cat $O/times.txt
cd /tmp
for v in pk nopk; do
  FDT_LIB=$R/tools/experiments/w44_libs/libfdt_hip_$v.so timeout -k 10 150 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d /tmp/raw -o w44_$v --output-format csv -- python $R/tools/one_conv.py 14 32 1 256 256 256 256 0 6 > $O/pmc_${v}_times.txt 2>&1 \
    && python $R/tools/summarize_pmc.py /tmp/raw/w44_${v}_counter_collection.csv $O/pmc_w44_$v.csv
  grep wino44 $O/pmc_w44_$v.csv | sed 's/.*ConvArgs)",//'
done
