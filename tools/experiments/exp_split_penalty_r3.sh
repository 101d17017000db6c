#!/bin/bash
# round 3 (eight frames in flight, lazy reduce passes): does a plan that splits K less than the isolated-latency optimum do better?
for SZ in "" "--height 480 --width 640"; do
  for P in 0 0.05 0.1 0.2; do
    FDT_TUNE_SPLIT_PENALTY=$P python bench.py --autotune 2 --steps 192 --warmup 24 --cpu-frames 0 --host-frames 0 --profile-frames 1 $SZ 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('penalty $P', '$SZ', d['value'], d['ms_per_step'])"
  done
done
