#!/bin/bash
set -e
O=gpurun_out/r4j18; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py tests/test_gpu_facebox.py -m gpu -x -q -k "whole_inverted or fused_inverted or try3 or config5 or expand" > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -3 $O/tests.log
for rep in 1 2; do for f in auto 1; do
  echo -n "rep $rep try3 b8 FDT_FUSE_IR=$f: " >> $O/ab.txt
  if [ $f = auto ]; then unset FDT_FUSE_IR; else export FDT_FUSE_IR=$f; fi
  python bench.py --arch try3 --batch 8 --steps 32 --warmup 4 --cpu-frames 0 --host-frames 0 --autotune 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], 'conv_stack', r['conv_stack']['ms_per_frame'], r['conv_stack']['launches_per_frame'], 'hbm_side', r['hbm_side']['frac'], r['hbm_side']['ms_per_frame'], d['parity'])" >> $O/ab.txt
done; done
unset FDT_FUSE_IR
cat $O/ab.txt
