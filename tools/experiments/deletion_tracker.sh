#!/bin/bash
# what does the device tracker (one track_step launch per frame on its own stream) cost the multi-stream step?
export FDT_EXPERIMENTS=1   # pipeline.py honours its experiment hooks only with this set
for R in 1 2; do for T in 0 1; do for SZ in "" "--height 480 --width 640"; do
FDT_EXP_NO_TRACK=$T python bench.py --steps 192 --warmup 24 --cpu-frames 0 --host-frames 0 --profile-frames 1 $SZ 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep $R no_track=$T', '$SZ', d['value'], d['ms_per_step'])"
done; done; done
