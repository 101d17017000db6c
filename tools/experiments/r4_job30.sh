#!/bin/bash
# 640x480: re-tune the batch-4 / batch-8 plans with the round-4 kernel classes, then group 4 vs 8
set -e
O=gpurun_out/r4j30; mkdir -p $O
T=face-detection-and-tracking_amd/tuned
cp $T/res50_640x480_b4.plan $O/res50_640x480_b4.plan.before
for b in 4 8; do
  timeout -k 10 500 python bench.py --height 480 --width 640 --autotune 2 --save-plan 1 --tune-iters 6 --steps 32 --warmup 4 --batch $b --cpu-frames 0 --host-frames 0 --profile-frames 1 > $O/tune_b$b.json 2> $O/tune_b$b.err
  cp $T/res50_640x480_b$b.plan $O/
  python -c "import json;d=json.loads(open('$O/tune_b$b.json').read().strip().splitlines()[-1]);print('batch $b', d['value'], d['roofline'].get('frac'), d['roofline']['backbone']['frac'])"
done
for cfg in "4 8" "8 4" "8 8"; do set -- $cfg
python bench.py --height 480 --width 640 --group $1 --inflight $2 --steps 512 --warmup 32 --cpu-frames 0 --host-frames 0 --profile-frames 1 --ungrouped-steps 0 > $O/c4_g$1_nf$2.json 2> $O/c4_g$1_nf$2.err
python -c "import json;d=json.loads(open('$O/c4_g$1_nf$2.json').read().strip().splitlines()[-1]);print('640x480 group $1 slots $2', d['value'], d['roofline']['frac'], d['roofline']['backbone']['frac'], d['parity'])"
done
