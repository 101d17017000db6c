#!/bin/bash
set -e
O=gpurun_out/r4j37; mkdir -p $O
run() { n=$1; shift; python bench.py "$@" --cpu-frames 0 --host-frames 0 --profile-frames 1 --ungrouped-steps 0 > $O/$n.json 2> $O/$n.err; python -c "import json;d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]);print('$n', d['value'])"; }
run c4 --height 480 --width 640 --steps 512 --warmup 32
run c4b --height 480 --width 640 --steps 512 --warmup 32
run g1 --group 1 --steps 256 --warmup 16
run p1080 --height 1080 --width 1920 --steps 48 --warmup 6
run try3b8 --arch try3 --batch 8 --steps 64 --warmup 6
python bench.py --arch facebox --batch 16 --steps 200 --warmup 8 --cpu-frames 0 --host-frames 0 > $O/fb.json 2> $O/fb.err; python -c "import json;d=json.loads(open('$O/fb.json').read().strip().splitlines()[-1]);print('facebox', d['value'])"
