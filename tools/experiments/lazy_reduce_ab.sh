#!/bin/bash
# A/B: reduce passes grouped in front of their first consumer (default) against one reduce pass per split layer (FDT_LAZY_REDUCE=0)
for R in 1 2 3; do for L in 0 1; do for SZ in "" "--height 480 --width 640"; do
FDT_LAZY_REDUCE=$L python bench.py --steps 192 --warmup 24 --cpu-frames 0 --host-frames 0 --profile-frames 1 $SZ 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep $R lazy=$L', '$SZ', d['value'], d['ms_per_step'])"
done; done; done
