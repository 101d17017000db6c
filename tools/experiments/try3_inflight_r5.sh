mkdir -p gpurun_out/r5aa
for nf in 2 3 4 6 8; do
  python bench.py --arch try3 --batch 8 --steps 48 --warmup 6 --inflight $nf --cpu-frames 0 --host-frames 0 --latency-frames 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('try3 batch 8 inflight $nf', d['value'], d['ms_per_step'])" >> gpurun_out/r5aa/try3_inflight.txt
done
cat gpurun_out/r5aa/try3_inflight.txt
