for lim in 28000 49152 0; do
  FDT_TUNE_MAX_LDS=$lim timeout -k 10 400 python bench.py --steps 64 --warmup 8 --cpu-frames 0 --autotune 2 > gpurun_out/exp_lds_$lim.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/exp_lds_$lim.json"))
print("lds limit $lim:", d["value"], d["ms_per_step"], "serial conv ms", d["roofline"]["conv_stack"]["ms_per_frame"])
PY
done
