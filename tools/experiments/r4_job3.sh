#!/bin/bash
# round-4 GPU job 3: tracker parity tests + tracker cost on the bench's own records (candidate form, second cut)
set -e
O=gpurun_out/r4j3; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_entry.py tests/test_gpu_pipeline.py tests/test_gpu_postproc.py -m gpu -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
cd face-detection-and-tracking_amd/csrc && cp libfdt_hip.so /tmp/libfdt_hip.keep && touch tracker.hip && make -s EXTRA=-DFDT_TRK_TIMING tracker.o libfdt_hip.so > /dev/null && cd ../..
timeout -k 10 300 python tools/experiments/tracker_on_bench_records.py --size 1024 --load tools/experiments/data/bench_records_1024.npz > $O/tracker_1024_phases.txt 2>&1
cat $O/tracker_1024_phases.txt
timeout -k 10 300 python tools/experiments/tracker_on_bench_records.py --height 480 --width 640 --load tools/experiments/data/bench_records_640x480.npz > $O/tracker_640x480_phases.txt 2>&1
head -6 $O/tracker_640x480_phases.txt
cp /tmp/libfdt_hip.keep face-detection-and-tracking_amd/csrc/libfdt_hip.so
