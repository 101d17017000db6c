for R in 1 2 3; do for NF in 4 8 12; do
python bench.py --steps 256 --warmup 32 --cpu-frames 0 --host-frames 0 --profile-frames 1 --inflight $NF 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$R', '$NF', d['value'], d['ms_per_step'])"
done; done
