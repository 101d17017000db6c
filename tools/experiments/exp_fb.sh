for nf in 4 6 8; do timeout -k 10 300 python bench.py --arch facebox --steps 60 --warmup 6 --cpu-frames 0 --inflight $nf 2>gpurun_out/fb_nf$nf.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('inflight $nf:', d['value'], d['ms_per_step'], d['roofline']['timed_step'])"; done; tail -3 gpurun_out/fb_nf3.err
