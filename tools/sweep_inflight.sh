for nf in 2 4 6; do timeout -k 10 200 python bench.py --steps 96 --warmup 12 --height 480 --width 640 --cpu-frames 0 --profile-frames 1 --inflight $nf > gpurun_out/s480_nf$nf.json 2> gpurun_out/s480_nf$nf.err; done
for nf in 2 4 5; do timeout -k 10 200 python bench.py --steps 48 --warmup 8 --cpu-frames 0 --profile-frames 1 --inflight $nf > gpurun_out/s1024_nf$nf.json 2> gpurun_out/s1024_nf$nf.err; done
for b in 2 4; do timeout -k 10 400 python bench.py --steps 48 --warmup 8 --height 480 --width 640 --cpu-frames 0 --profile-frames 1 --batch $b --inflight 2 > gpurun_out/s480_b$b.json 2> gpurun_out/s480_b$b.err; done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/s*.json")):
    try:
        d=json.load(open(f)); print(f, d["value"], d["ms_per_step"], d["parity"])
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-500:])
PY
