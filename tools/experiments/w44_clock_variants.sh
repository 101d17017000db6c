#!/bin/bash
# Which part of conv_wino44_kernel's k-step pulls the clock down?  Stamps build (shader clock inside the main loop) x FDT_W44_EXP
# phase deletions (bit mask: 1 no input transform, 4 no LDS-DMA in the loop, 8 no MFMA, 16 no operand reads; results are wrong
# for every value but 0).  Build host: for e in 1 4 16 8 21 5: make EXTRA="-DFDT_W44_STAMPS -DFDT_W44_EXP=$e" -> w44_libs/.
O=gpurun_out/w44_clock; mkdir -p $O
for e in 0 1 4 16 5 21 8; do
  lib=$PWD/tools/experiments/w44_libs/libfdt_hip_stamps_exp$e.so
  [ $e = 0 ] && lib=$PWD/tools/experiments/w44_libs/libfdt_hip_stamps.so
  echo "== FDT_W44_EXP=$e" | tee -a $O/out.txt
  W44_SHAPES=1 FDT_LIB=$lib timeout -k 10 120 python tools/experiments/w44_stamps.py 2>&1 | grep -v amdgpu.ids | tee -a $O/out.txt
done
