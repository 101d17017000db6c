#!/bin/bash
# conv_wino44_kernel (Winograd F(4x4,3x3)): which part of a k-step is the bottleneck?  Builds the library once per
# FDT_W44_EXP value HERE (cross-compile), then on the GPU box:  bash tools/experiments/w44_variants.sh run
#   bit mask: 1 no input transform, 2 no workgroup barrier, 4 no LDS-DMA in the loop, 8 no MFMA, 16 no operand reads
set -e
EXPS=${EXPS:-"1 2 4 8 16 5 7 21 23 13"}
D=face-detection-and-tracking_amd/csrc
if [ "$1" = "run" ]; then
  for e in 0 $EXPS; do
    echo "== FDT_W44_EXP=$e"
    FDT_LIB=$PWD/tools/experiments/w44_libs/libfdt_hip_exp$e.so python tools/one_conv.py 14 32 1 256 256 256 256 0 10
  done
  exit 0
fi
mkdir -p tools/experiments/w44_libs
for e in $EXPS 0; do
  touch $D/conv_wino44.h
  make -C $D -j8 EXTRA=-DFDT_W44_EXP=$e > /dev/null
  cp $D/libfdt_hip.so tools/experiments/w44_libs/libfdt_hip_exp$e.so
done
