#!/bin/bash
# Re-autotune the committed plans (run on the GPU box); the new plan files are copied to gpurun_out/plans_new/.
mkdir -p gpurun_out/plans_new
T=face-detection-and-tracking_amd/tuned
run() { # name, bench args...
  local n=$1; shift
  timeout -k 10 400 python bench.py --autotune 2 --save-plan 1 --cpu-frames 0 --host-frames 0 "$@" > gpurun_out/plans_new/$n.json 2> gpurun_out/plans_new/$n.err && head -c 120 gpurun_out/plans_new/$n.json && echo
}
run res50_1024x1024_b1 --steps 64 --warmup 8 &&
run res50_640x480_b1 --steps 64 --warmup 8 --height 480 --width 640 &&
run res50_640x480_b4 --steps 32 --warmup 4 --height 480 --width 640 --batch 4 &&
run res50_640x480_b2 --steps 32 --warmup 4 --height 480 --width 640 --batch 2 &&
run res50_1920x1080_b1 --steps 16 --warmup 4 --height 1080 --width 1920 &&
run res50_1024x1024_b2 --steps 24 --warmup 4 --batch 2 &&
run try3_1024x1024_b8 --steps 24 --warmup 4 --arch try3 --batch 8 &&
run try3_1024x1024_b1 --steps 48 --warmup 8 --arch try3
cp $T/*.plan gpurun_out/plans_new/
timeout -k 10 300 python tools/profile_layers.py > gpurun_out/plans_new/per_layer_1024.txt 2>&1
