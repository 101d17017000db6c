for Q in 3 4 5 6 9 12 16; do for SZ in "" "--height 480 --width 640"; do
GPU_MAX_HW_QUEUES=$Q python bench.py --steps 192 --warmup 24 --cpu-frames 0 --host-frames 0 --profile-frames 1 $SZ 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queues $Q', '$SZ', d['value'], d['ms_per_step'])"
done; done
