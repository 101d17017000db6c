#!/bin/bash
O=gpurun_out/r4j6; mkdir -p $O
for R in 4 3 2 1; do
  echo "== FDT_P1_RESIDENT=$R" >> $O/resident.txt
  for S in "64 256 256 256 1" "128 128 128 512 1" "256 256 256 64 0" "256 64 64 1024 1" "512 128 128 128 0"; do
    FDT_P1_RESIDENT=$R python tools/one_conv.py 16 34 1 $S 30 >> $O/resident.txt 2>&1
    FDT_P1_RESIDENT=$R python tools/one_conv.py 17 34 1 $S 30 >> $O/resident.txt 2>&1
  done
done
grep -v amdgpu.ids $O/resident.txt
