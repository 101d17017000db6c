#!/bin/bash
set -e
O=gpurun_out/r4j36; mkdir -p $O
python -m pytest tests/test_gpu_conv.py -x -q -k "winograd_f4x4 or every_tile or channels_past or combine or split_k" > $O/tests_conv.log 2>&1 || { tail -30 $O/tests_conv.log; exit 1; }
tail -2 $O/tests_conv.log
python - <<'PY' | tee $O/times.txt
import sys
sys.path.insert(0, "tools")
import conv_bench as cb
for kind, cin, h, w, cout, split, B in ((14, 256, 256, 256, 256, 1, 1), (14, 256, 256, 256, 256, 1, 4), (14, 512, 128, 128, 512, 1, 4), (14, 1024, 64, 64, 1024, 2, 4), (15, 256, 256, 256, 128, 1, 4), (14, 128, 128, 128, 128, 1, 4), (14, 64, 256, 256, 64, 1, 4)):
    ms = cb.bench(kind, 32, split, cin, h, w, cout, iters=20, B=B)
    print("%s %4d -> %4d @%3d^2 /%d batch %d: %7.1f us" % (cb.KIND[kind], cin, cout, h, split, B, ms * 1e3), flush=True)
PY
python -m pytest tests/test_gpu_model.py tests/test_gpu_pipeline.py -x -q > $O/tests_model.log 2>&1 || { tail -30 $O/tests_model.log; exit 1; }
tail -2 $O/tests_model.log
for r in 1 2; do
python bench.py --steps 256 --warmup 16 --cpu-frames 0 --host-frames 0 --profile-frames 2 --ungrouped-steps 0 > $O/bench_$r.json 2> $O/bench_$r.err
python -c "import json;d=json.loads(open('$O/bench_$r.json').read().strip().splitlines()[-1]);print('bench', d['value'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline']['backbone']['frac'], d['parity'])"
done
