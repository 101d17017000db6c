#!/bin/bash
set -e
O=gpurun_out/r4j4; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_conv.py -m gpu -x -q -k "persistent or every_tile_variant_matches_torch" > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
timeout -k 10 900 python tools/experiments/bench_1x1_r4.py > $O/bench_1x1.txt 2>&1
cat $O/bench_1x1.txt
