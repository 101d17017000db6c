#!/bin/bash
set -e
O=gpurun_out/r4j34; mkdir -p $O
for nf in 2 3 4 6 8 12; do
  python bench.py --arch try3 --batch 8 --inflight $nf --steps 64 --warmup 6 --cpu-frames 0 --host-frames 0 --profile-frames 1 > $O/t3_nf$nf.json 2> $O/t3_nf$nf.err
  python -c "import json;d=json.loads(open('$O/t3_nf$nf.json').read().strip().splitlines()[-1]);print('try3 b8 inflight $nf', d['value'], d['ms_per_step'])"
done
