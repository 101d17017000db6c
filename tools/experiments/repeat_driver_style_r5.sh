#!/bin/bash
# Run-to-run spread of the driver's command on one box (5 repeats), and one long run (1500 timed steps): is the 20-step figure a steady-state figure?
mkdir -p gpurun_out/r5z
for i in 1 2 3 4 5; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --host-frames 0 --latency-frames 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver-style run $i', d['value'], d['ms_per_step'], 'ungrouped', d['ungrouped']['value'])" >> gpurun_out/r5z/repeat.txt
done
python bench.py --steps 1500 --warmup 50 --cpu-frames 0 --host-frames 0 --latency-frames 0 --ungrouped-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('1500 steps', d['value'], d['ms_per_step'], d['parity'].get('tracks_equal'))" >> gpurun_out/r5z/repeat.txt
cat gpurun_out/r5z/repeat.txt
