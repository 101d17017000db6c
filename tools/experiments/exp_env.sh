run() { env "$@" timeout -k 10 300 python bench.py --steps 64 --warmup 8 --cpu-frames 0 --profile-frames 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])"; }
run A=1
run GPU_MAX_HW_QUEUES=8
run GPU_MAX_HW_QUEUES=2
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run A=1
