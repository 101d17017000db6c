#!/bin/bash
# Throughput of the committed plans with the split-K combine forced inside the conv kernels (FDT_FORCE_COMBINE=1, optionally
# only up to a split factor) against the reduce passes (=0): the autotuner ranks by isolated latency, the pipeline by throughput.
# the hooks this script sets are compiled in only with -DFDT_EXPERIMENTS (the product library ignores them):
(cd face-detection-and-tracking_amd/csrc && touch model.hip conv.hip && make -s -j8 EXTRA=-DFDT_EXPERIMENTS > /dev/null)
for SZ in "" "--height 480 --width 640"; do
  for F in "0 4096" "1 4096" "1 2" "1 4" "1 8"; do
    set -- $F
    FDT_FORCE_COMBINE=$1 FDT_FORCE_COMBINE_MAXS=$2 python bench.py --steps 96 --warmup 12 --cpu-frames 0 --host-frames 0 --profile-frames 1 $SZ 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-26s force=%s maxS=%-5s %8.1f frames/s  %.3f ms/step  %s' % ('$SZ', '$1', '$2', d['value'], d['ms_per_step'], d.get('parity',{}).get('tracks_equal')))"
  done
done
