# 640x480 (the tracker's real frame size, config 4): 8 frames in flight per GPU as batch x inflight
for cfg in "8 1" "4 2" "2 4" "8 2" "4 3"; do set -- $cfg; timeout -k 10 400 python bench.py --steps 48 --warmup 8 --height 480 --width 640 --cpu-frames 0 --profile-frames 1 --batch $1 --inflight $2 --save-plan 1 > gpurun_out/s480_b$1_nf$2.json 2> gpurun_out/s480_b$1_nf$2.err; done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/s480_b*_nf*.json")):
    try:
        d=json.load(open(f)); print(f, d["value"], d["ms_per_step"], d["parity"], d["roofline"]["timed_step"])
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-500:])
PY
cp face-detection-and-tracking_amd/tuned/res50_640x480_b*.plan gpurun_out/ 
