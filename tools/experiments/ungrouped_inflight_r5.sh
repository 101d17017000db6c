mkdir -p gpurun_out/r5q
for nf in 6 8 12 16; do
  python bench.py --steps 256 --warmup 32 --group 1 --inflight $nf --cpu-frames 0 --host-frames 0 --ungrouped-steps 0 --latency-frames 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('1024 group 1 inflight $nf', d['value'], d['ms_per_step'])" >> gpurun_out/r5q/ungrouped_inflight.txt
done
for nf in 8 12 16; do
  python bench.py --height 480 --width 640 --steps 512 --warmup 32 --group 1 --inflight $nf --cpu-frames 0 --host-frames 0 --ungrouped-steps 0 --latency-frames 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('640x480 group 1 inflight $nf', d['value'], d['ms_per_step'])" >> gpurun_out/r5q/ungrouped_inflight.txt
done
cat gpurun_out/r5q/ungrouped_inflight.txt
