#!/bin/bash
# Ceiling of an in-kernel split-K combine: the reduce passes simply not launched (FDT_SKIP_OPS=@reduce, csrc/conv.hip hook).
# the hooks this script sets are compiled in only with -DFDT_EXPERIMENTS (the product library ignores them):
(cd face-detection-and-tracking_amd/csrc && touch model.hip conv.hip && make -s -j8 EXTRA=-DFDT_EXPERIMENTS > /dev/null)
for SZ in "--height 480 --width 640" "" "--arch facebox"; do
  for SK in "" "@reduce"; do
    FDT_SKIP_OPS="$SK" python bench.py --steps 96 --warmup 12 --cpu-frames 0 --host-frames 0 --profile-frames 1 $SZ 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s %-10s %9.1f frames/s  %.3f ms/step' % ('$SZ', '$SK', d['value'], d['ms_per_step']))"
  done
done
