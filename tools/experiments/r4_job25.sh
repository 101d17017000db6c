#!/bin/bash
set -e
O=gpurun_out/r4j25; mkdir -p $O
for r in 1 2 3; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --host-frames 0 > $O/driver_$r.json 2> $O/driver_$r.err
  python -c "import json;d=json.loads(open('$O/driver_$r.json').read().strip().splitlines()[-1]);print('driver-style', d['value'], d['ms_per_step'], d['parity'], d['ungrouped'])"
done
python bench.py --gpus 1 --steps 20 --warmup 5 --group 1 --cpu-frames 0 --host-frames 0 > $O/driver_g1.json 2> $O/driver_g1.err
python -c "import json;d=json.loads(open('$O/driver_g1.json').read().strip().splitlines()[-1]);print('driver-style group 1', d['value'], d['ms_per_step'])"
python bench.py --gpus 1 --steps 22 --warmup 5 --cpu-frames 0 --host-frames 0 > $O/driver_22.json 2> $O/driver_22.err
python -c "import json;d=json.loads(open('$O/driver_22.json').read().strip().splitlines()[-1]);print('22 steps', d['value'], d['ms_per_step'], d['parity'])"
