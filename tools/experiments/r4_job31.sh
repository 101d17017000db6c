#!/bin/bash
# hardware queues: does the step gain from more than the default four? (streams of the slots + the tracker stream share them)
set -e
O=gpurun_out/r4j31; mkdir -p $O
for q in 4 8 12; do
  GPU_MAX_HW_QUEUES=$q python bench.py --steps 256 --warmup 16 --cpu-frames 0 --host-frames 0 --profile-frames 1 --ungrouped-steps 0 > $O/q$q.json 2> $O/q$q.err
  python -c "import json;d=json.loads(open('$O/q$q.json').read().strip().splitlines()[-1]);print('GPU_MAX_HW_QUEUES=$q', d['value'], d['parity'])"
done
for q in 4 8; do
  GPU_MAX_HW_QUEUES=$q python bench.py --height 480 --width 640 --steps 512 --warmup 32 --cpu-frames 0 --host-frames 0 --profile-frames 1 --ungrouped-steps 0 > $O/c4_q$q.json 2> $O/c4_q$q.err
  python -c "import json;d=json.loads(open('$O/c4_q$q.json').read().strip().splitlines()[-1]);print('640x480 GPU_MAX_HW_QUEUES=$q', d['value'])"
  GPU_MAX_HW_QUEUES=$q python bench.py --arch facebox --batch 16 --steps 200 --warmup 8 --cpu-frames 0 > $O/fb_q$q.json 2> $O/fb_q$q.err
  python -c "import json;d=json.loads(open('$O/fb_q$q.json').read().strip().splitlines()[-1]);print('facebox GPU_MAX_HW_QUEUES=$q', d['value'])"
  GPU_MAX_HW_QUEUES=$q python bench.py --arch try3 --batch 8 --steps 48 --warmup 4 --cpu-frames 0 --host-frames 0 > $O/t3_q$q.json 2> $O/t3_q$q.err
  python -c "import json;d=json.loads(open('$O/t3_q$q.json').read().strip().splitlines()[-1]);print('try3 b8 GPU_MAX_HW_QUEUES=$q', d['value'])"
done
