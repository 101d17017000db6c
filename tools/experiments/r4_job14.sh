#!/bin/bash
set -e
O=gpurun_out/r4j14; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -3 $O/tests.log
for rep in 1 2; do for f in 1 0; do
  echo -n "rep $rep res50 1024 fuse=$f: " >> $O/ab.txt
  FDT_FUSE_INGEST=$f python bench.py --steps 256 --warmup 24 --cpu-frames 0 --host-frames 0 --profile-frames 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], 'backbone', r['backbone']['ms_per_frame'], r['backbone']['frac'], 'all', r['conv_stack']['all_ops_contiguous_ms_per_frame'], d['parity'])" >> $O/ab.txt
  echo -n "rep $rep facebox fuse=$f: " >> $O/ab.txt
  FDT_FUSE_INGEST=$f python bench.py --arch facebox --batch 16 --steps 100 --warmup 8 --cpu-frames 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], 'fwd', r['forward']['ms_per_batch'], r['forward']['launches'], [ (o['op'], o['ms']) for o in r['by_op'][:4]])" >> $O/ab.txt
done; done
cat $O/ab.txt
