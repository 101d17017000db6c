#!/bin/bash
# frames (batches) in flight for the other bench configurations
run() { python bench.py --cpu-frames 0 --host-frames 0 --profile-frames 1 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '->', d['value'], d['ms_per_step'])"; }
for NF in 3 4 6 8; do
  run --steps 64 --warmup 8 --height 480 --width 640 --batch 4 --inflight $NF
  run --steps 48 --warmup 8 --batch 2 --inflight $NF
  run --steps 48 --warmup 8 --arch try3 --batch 8 --inflight $NF
  run --steps 128 --warmup 16 --arch try3 --inflight $NF
  run --steps 48 --warmup 8 --height 1080 --width 1920 --inflight $NF
done
