#!/bin/bash
set -e
O=gpurun_out/r4j28; mkdir -p $O
python -m pytest tests/test_gpu_entry.py tests/test_gpu_facebox.py tests/test_gpu_pipeline.py -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for rep in 1 2; do
python bench.py --arch facebox --batch 16 --steps 100 --warmup 8 --cpu-frames 0 > $O/fb_$rep.json 2> $O/fb_$rep.err
python -c "import json;d=json.loads(open('$O/fb_$rep.json').read().strip().splitlines()[-1]);r=d['roofline'];print('facebox', d['value'], d['ms_per_step'], [(o['op'],o['ms'],o['GBps']) for o in r['by_op'][:4]], r['timed_step'])"
done
python bench.py --steps 128 --warmup 16 --source 1080x1920 --height 480 --width 640 --cpu-frames 0 --host-frames 128 > $O/c4.json 2> $O/c4.err
python -c "import json;d=json.loads(open('$O/c4.json').read().strip().splitlines()[-1]);print('c4 from 1080p', d['value'], d['ms_per_step'], d['host_path']['engines'])"
