mkdir -p gpurun_out/r5p
for cfg in "8 8" "8 4" "4 8" "4 4" "4 6" "2 8"; do set -- $cfg
  python bench.py --steps 128 --warmup 16 --group $1 --inflight $2 --cpu-frames 0 --host-frames 0 --ungrouped-steps 0 --latency-frames 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('group $1 inflight $2', d['value'], d['ms_per_step'])" >> gpurun_out/r5p/sweep_c2.txt
done
for cfg in "8 8" "8 4" "16 4" "4 8"; do set -- $cfg
  python bench.py --height 480 --width 640 --steps 256 --warmup 32 --group $1 --inflight $2 --cpu-frames 0 --host-frames 0 --ungrouped-steps 0 --latency-frames 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('640x480 group $1 inflight $2', d['value'], d['ms_per_step'])" >> gpurun_out/r5p/sweep_c2.txt
done
cat gpurun_out/r5p/sweep_c2.txt
