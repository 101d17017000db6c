#!/bin/bash
# tracker stream (and some detector streams) at high priority = another pool of hardware queues: less queue sharing?
export FDT_EXPERIMENTS=1   # pipeline.py honours its experiment hooks only with this set
for R in 1 2; do for CFG in "0 0" "-1 0" "-1 2" "0 2" "-1 4"; do set -- $CFG; for SZ in "" "--height 480 --width 640"; do
FDT_TRK_PRIO=$1 FDT_DET_PRIO_ALT=$2 python bench.py --steps 192 --warmup 24 --cpu-frames 0 --host-frames 0 --profile-frames 1 $SZ 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep $R trk_prio=$1 det_alt=$2', '$SZ', d['value'], d['ms_per_step'], d.get('parity',{}).get('tracks_equal'))"
done; done; done
