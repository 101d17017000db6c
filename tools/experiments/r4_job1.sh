#!/bin/bash
# round-4 GPU job 1: (a) two ranks on one GPU through the round-3/4 N>1 bench path (gloo host exchange), (b) tracker cost on the
# bench's own records, product build and -DFDT_TRK_TIMING build
set -e
O=gpurun_out/r4j1; mkdir -p $O
FDT_BENCH_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 2 --steps 20 --warmup 4 > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err
tail -c 1500 $O/bench_2rank_gloo.json
timeout -k 10 300 python tools/experiments/tracker_on_bench_records.py --size 1024 --frames 72 --dump $O/records_1024.npy > $O/tracker_1024.txt 2>&1
cat $O/tracker_1024.txt
timeout -k 10 300 python tools/experiments/tracker_on_bench_records.py --height 480 --width 640 --frames 72 --dump $O/records_640x480.npy > $O/tracker_640x480.txt 2>&1
cat $O/tracker_640x480.txt
# per-phase clocks: tracker.o rebuilt with the timing hooks into a scratch copy of the library
cd face-detection-and-tracking_amd/csrc && cp libfdt_hip.so /tmp/libfdt_hip.keep && touch tracker.hip && make -s EXTRA=-DFDT_TRK_TIMING tracker.o libfdt_hip.so > /dev/null && cd ../..
timeout -k 10 300 python tools/experiments/tracker_on_bench_records.py --size 1024 --load $O/records_1024.npy > $O/tracker_1024_phases.txt 2>&1
cat $O/tracker_1024_phases.txt
cp /tmp/libfdt_hip.keep face-detection-and-tracking_amd/csrc/libfdt_hip.so
