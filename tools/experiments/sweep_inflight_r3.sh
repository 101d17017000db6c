#!/bin/bash
# frames in flight with the round-3 plans (F(4x4) layers, grouped head finalize): 640x480 and 1024x1024, batch 1
for NF in 4 5 6 8 10 12 16; do
  for SZ in "--height 480 --width 640" ""; do
    python bench.py --steps 192 --warmup 24 --cpu-frames 0 --host-frames 0 --profile-frames 1 --inflight $NF $SZ 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$NF', d['metric'][-9:], d['value'], d['ms_per_step'])"
  done
done
