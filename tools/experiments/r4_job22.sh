#!/bin/bash
# grouped hand-over at 1024^2: slots x group sweep (b8 plan tuned first)
set -e
O=gpurun_out/r4j22; mkdir -p $O
T=face-detection-and-tracking_amd/tuned
timeout -k 10 500 python bench.py --autotune 2 --save-plan 1 --tune-iters 5 --steps 16 --warmup 2 --batch 8 --inflight 4 --cpu-frames 0 --host-frames 0 --profile-frames 1 > $O/tune_b8.json 2> $O/tune_b8.err
cp $T/res50_1024x1024_b8.plan $O/
python -c "import json;d=json.loads(open('$O/tune_b8.json').read().strip().splitlines()[-1]);print('batch 8', d['value'], d['roofline'].get('frac'), d['roofline']['backbone']['frac'])"
for cfg in "4 2" "4 3" "4 4" "4 8" "8 2" "8 4" "2 8"; do set -- $cfg
  python bench.py --group $1 --inflight $2 --steps 384 --warmup 32 --cpu-frames 0 --host-frames 0 --profile-frames 1 > $O/bench_g$1_nf$2.json 2> $O/bench_g$1_nf$2.err
  python -c "import json;d=json.loads(open('$O/bench_g$1_nf$2.json').read().strip().splitlines()[-1]);print('group $1 slots $2', d['value'], d['roofline'].get('frac'), d['roofline']['backbone']['frac'], d['parity']['tracks_equal'] if d.get('parity') else None)"
done
