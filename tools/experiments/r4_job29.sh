#!/bin/bash
set -e
O=gpurun_out/r4j29; mkdir -p $O
for g in 4 8; do
python bench.py --height 480 --width 640 --group $g --steps 512 --warmup 32 --cpu-frames 0 --host-frames 0 --profile-frames 1 --ungrouped-steps 0 > $O/c4_g$g.json 2> $O/c4_g$g.err
python -c "import json;d=json.loads(open('$O/c4_g$g.json').read().strip().splitlines()[-1]);print('640x480 group $g', d['value'], d['roofline']['frac'], d['roofline']['backbone']['frac'], d['parity'])"
done
for nf in 2 4 6 8; do
python bench.py --arch facebox --batch 16 --inflight $nf --steps 200 --warmup 8 --cpu-frames 0 > $O/fb_nf$nf.json 2> $O/fb_nf$nf.err
python -c "import json;d=json.loads(open('$O/fb_nf$nf.json').read().strip().splitlines()[-1]);print('facebox inflight $nf', d['value'], d['ms_per_step'])"
done
