#!/bin/bash
# FaceBoxes stem: conv_stem_s4.h (K = 168, three workgroups per CU) vs the generic class-6 kernel
set -e
O=gpurun_out/r4j27; mkdir -p $O
python -m pytest tests/test_gpu_conv.py -x -q -k "facebox_stem or every_tile_variant" > $O/tests_conv.log 2>&1 || { tail -30 $O/tests_conv.log; exit 1; }
tail -2 $O/tests_conv.log
python -m pytest tests/test_gpu_facebox.py -x -q > $O/tests_fb.log 2>&1 || { tail -40 $O/tests_fb.log; exit 1; }
tail -2 $O/tests_fb.log
for rep in 1 2; do
python bench.py --arch facebox --batch 16 --steps 100 --warmup 8 --cpu-frames 0 > $O/fb_new_$rep.json 2> $O/fb_new_$rep.err
python -c "import json;d=json.loads(open('$O/fb_new_$rep.json').read().strip().splitlines()[-1]);r=d['roofline'];print('new', d['value'], d['ms_per_step'], [(o['op'],o['ms']) for o in r['by_op'][:4]], r['mfma_side'])"
done
FDT_FUSE_INGEST=2 python bench.py --arch facebox --batch 16 --steps 100 --warmup 8 --cpu-frames 0 > $O/fb_new_u8.json 2> $O/fb_new_u8.err
python -c "import json;d=json.loads(open('$O/fb_new_u8.json').read().strip().splitlines()[-1]);r=d['roofline'];print('new u8', d['value'], d['ms_per_step'], [(o['op'],o['ms']) for o in r['by_op'][:4]])"
