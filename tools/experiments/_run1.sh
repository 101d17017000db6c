mkdir -p gpurun_out/r3g/plans
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r3g/tests.log 2>&1; echo "rc $?" >> gpurun_out/r3g/tests.log; tail -3 gpurun_out/r3g/tests.log
timeout -k 10 400 python bench.py --steps 64 --warmup 8 --autotune 2 --save-plan 1 --cpu-frames 0 --host-frames 0 > gpurun_out/r3g/at1024.json 2> gpurun_out/r3g/at1024.err; head -c 110 gpurun_out/r3g/at1024.json; echo
timeout -k 10 400 python bench.py --steps 64 --warmup 8 --height 480 --width 640 --autotune 2 --save-plan 1 --cpu-frames 0 --host-frames 0 > gpurun_out/r3g/at480.json 2> gpurun_out/r3g/at480.err; head -c 110 gpurun_out/r3g/at480.json; echo
cp face-detection-and-tracking_amd/tuned/res50_1024x1024_b1.plan face-detection-and-tracking_amd/tuned/res50_640x480_b1.plan gpurun_out/r3g/plans/
