#!/bin/bash
# Detector streams confined to CU partitions (FDT_CU_PARTS, fdt_stream_create_partition): do frames in flight overlap better
# when their kernels CANNOT compete for the same CUs?  Committed plans (tuned for 256 CUs); with and without graph replay.
export FDT_EXPERIMENTS=1   # pipeline.py honours its experiment hooks only with this set
for SZ in "" "--height 480 --width 640"; do
  for P in 1 2 4; do
    for G in 1 0; do
      for NF in 4 8; do
        FDT_GRAPH=$G FDT_CU_PARTS=$P python bench.py --steps 96 --warmup 12 --cpu-frames 0 --host-frames 0 --profile-frames 1 --inflight $NF $SZ 2>/dev/null |
          python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-26s parts=%s graph=%s inflight=%s %8.1f frames/s  %.3f ms/step  %s' % ('$SZ', '$P', '$G', '$NF', d['value'], d['ms_per_step'], d.get('parity',{}).get('tracks_equal')))"
      done
    done
  done
done
