# the hooks this script sets are compiled in only with -DFDT_EXPERIMENTS (the product library ignores them):
(cd face-detection-and-tracking_amd/csrc && touch model.hip conv.hip && make -s -j8 EXTRA=-DFDT_EXPERIMENTS > /dev/null)
for R in 1 2; do for S in "" "@detect"; do for SZ in "" "--height 480 --width 640"; do
FDT_SKIP_OPS="$S" python bench.py --steps 192 --warmup 24 --cpu-frames 0 --host-frames 0 --profile-frames 1 $SZ 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep $R skip=[$S]', '$SZ', d['value'], d['ms_per_step'])"
done; done; done
