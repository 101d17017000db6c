for cfg in "2 2" "2 3" "4 2"; do set -- $cfg
  timeout -k 10 500 python bench.py --steps 32 --warmup 6 --cpu-frames 0 --profile-frames 1 --batch $1 --inflight $2 --autotune 2 > gpurun_out/exp_b1024_$1_$2.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/exp_b1024_$1_$2.json"))
print("batch $1 inflight $2:", d["value"], d["ms_per_step"], d["parity"], d["roofline"]["timed_step"]["frac_executed"])
PY
done
