#!/bin/bash
# Several autotune runs of one configuration (the tuner ranks by the minimum of a few timed launches: its picks differ from run
# to run), each candidate plan then measured twice through the committed-plan path; the candidates and their rates land in
# gpurun_out/plans_cand/.   bash tools/experiments/best_plan.sh <plan name> <bench size args...>
NAME=$1; shift
T=face-detection-and-tracking_amd/tuned
OUT=gpurun_out/plans_cand
mkdir -p $OUT
cp $T/$NAME.plan $OUT/${NAME}_committed.plan
rate() { python bench.py --steps 256 --warmup 24 --cpu-frames 0 --host-frames 0 --profile-frames 1 "$@" 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])"; }
i=0
for P in 0 0.05 0.1 0 0.05 0.1; do
  i=$((i+1))
  FDT_TUNE_SPLIT_PENALTY=$P python bench.py --autotune 2 --save-plan 1 --tune-iters 8 --steps 32 --warmup 8 --cpu-frames 0 --host-frames 0 --profile-frames 1 "$@" > /dev/null 2>&1
  cp $T/$NAME.plan $OUT/${NAME}_cand${i}_p$P.plan
done
for f in $OUT/${NAME}_*.plan; do
  cp $f $T/$NAME.plan
  echo "$(basename $f) $(rate "$@") $(rate "$@")"
done | tee $OUT/${NAME}_rates.txt
cp $OUT/${NAME}_committed.plan $T/$NAME.plan
