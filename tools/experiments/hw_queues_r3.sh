#!/bin/bash
# hardware queues x frames in flight (round 3 kernels): does every stream want a queue of its own?
for Q in 2 4 8; do for NF in 8 12; do for SZ in "" "--height 480 --width 640"; do
GPU_MAX_HW_QUEUES=$Q python bench.py --steps 192 --warmup 24 --cpu-frames 0 --host-frames 0 --profile-frames 1 --inflight $NF $SZ 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queues $Q inflight $NF', '$SZ', d['value'], d['ms_per_step'])"
done; done; done
