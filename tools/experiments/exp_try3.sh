for cfg in "1 3" "1 4" "1 6" "8 3" "8 4"; do set -- $cfg
timeout -k 10 300 python bench.py --arch try3 --batch $1 --steps 48 --warmup 6 --cpu-frames 0 --profile-frames 1 --inflight $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('try3 batch $1 inflight $2:', d['value'], d['ms_per_step'], d['parity'])"; done
