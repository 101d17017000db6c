#!/bin/bash
# (a) sclk / power while the default bench runs; (b) grouped hand-over at 1024^2: G = 2 and 4 (plans re-tuned first)
set -e
O=gpurun_out/r4j21; mkdir -p $O
T=face-detection-and-tracking_amd/tuned
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.5; done ) > $O/clocks.txt 2>&1 &
POLL=$!
python bench.py --steps 400 --warmup 20 --cpu-frames 0 --host-frames 0 --profile-frames 1 > $O/bench_g1.json 2> $O/bench_g1.err
kill $POLL || true
sort $O/clocks.txt | uniq -c | sort -rn | head -8
for b in 2 4; do
  timeout -k 10 500 python bench.py --autotune 2 --save-plan 1 --tune-iters 5 --steps 24 --warmup 4 --batch $b --cpu-frames 0 --host-frames 0 --profile-frames 1 > $O/tune_b$b.json 2> $O/tune_b$b.err
  cp $T/res50_1024x1024_b$b.plan $O/
  python -c "import json;d=json.loads(open('$O/tune_b$b.json').read().strip().splitlines()[-1]);print('batch $b', d['value'], d['roofline'].get('frac'), d['roofline']['backbone']['frac'])"
done
for g in 1 2 4; do
  python bench.py --group $g --steps 256 --warmup 16 --cpu-frames 1 --cpu-threads 64 --host-frames 0 > $O/bench_group$g.json 2> $O/bench_group$g.err
  python -c "import json;d=json.loads(open('$O/bench_group$g.json').read().strip().splitlines()[-1]);print('group $g', d['value'], d['roofline'].get('frac'), d['roofline']['backbone']['frac'], d.get('parity'))"
done
