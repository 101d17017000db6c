#!/bin/bash
set -e
export TMPDIR=/tmp
R=$PWD
O=gpurun_out/r4j12; mkdir -p $O /tmp/raw
T=face-detection-and-tracking_amd/tuned
timeout -k 10 900 python -m pytest tests/test_gpu_cabi_pipeline.py tests/test_gpu_conv.py -m gpu -x -q -k "cabi or plain_c or xcd_aware or persistent" > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -3 $O/tests.log
# workgroup maps on the F(4x4) calibration layer: time and HBM read bytes
for m in 0 1 3; do
  echo "== map $m" >> $O/maps_w44.txt
  FDT_CONV_MAP=$m python tools/one_conv.py 14 32 1 256 256 256 256 0 20 2>/dev/null | tail -1 >> $O/maps_w44.txt
  FDT_CONV_MAP=$m python tools/one_conv.py 14 32 1 256 256 256 64 0 20 2>/dev/null | tail -1 >> $O/maps_w44.txt
  (cd /tmp && FDT_CONV_MAP=$m timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d /tmp/raw -o m$m --output-format csv -- python $R/tools/one_conv.py 14 32 1 256 256 256 256 0 4 > /dev/null 2>&1 && python $R/tools/summarize_pmc.py /tmp/raw/m${m}_counter_collection.csv $R/$O/map${m}_rd.csv)
  grep wino44 $O/map${m}_rd.csv | sed 's/.*ConvArgs)",//' >> $O/maps_w44.txt
done
cat $O/maps_w44.txt
cp $T/res50_1024x1024_b1.plan $O/v0.plan
python tools/experiments/plan_map_variants.py $O/v0.plan $O
for rep in 1 2; do for v in v0 v1 v2; do
  cp $O/$v.plan $T/res50_1024x1024_b1.plan
  echo -n "rep $rep $v: " >> $O/variants.txt
  python bench.py --steps 256 --warmup 24 --cpu-frames 0 --host-frames 0 --profile-frames 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['backbone']['ms_per_frame'], d['roofline']['conv_stack']['all_ops_contiguous_ms_per_frame'])" >> $O/variants.txt
done; done
cp $O/v0.plan $T/res50_1024x1024_b1.plan
cat $O/variants.txt
