#!/bin/bash
# Does a pipeline with three frames in flight prefer LESS split-K than the isolated-latency autotuner picks?
# FDT_TUNE_SPLIT_PENALTY=p ranks candidates by ms * (1 + p * log2(split)).
cd "$(dirname "$0")/../.."
for p in 0 0.05 0.1 0.2; do
  echo "== split penalty $p"
  FDT_TUNE_SPLIT_PENALTY=$p python bench.py --cpu-frames 0 --autotune 2 2>/dev/null | tail -1 | cut -c1-130
done
