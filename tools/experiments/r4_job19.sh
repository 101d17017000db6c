#!/bin/bash
set -e
O=gpurun_out/r4j19; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -30 $O/smoke.log; exit 1; }
tail -2 $O/smoke.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -3 $O/tests.log
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python -c "
import json; l=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); r=l['roofline']
print(l['value'], l['ms_per_step'], r['frac'], r['backbone']['frac'], r['traffic'], r['traffic_over_algorithmic'], l['parity']['tracks_equal'], l['cpu_baseline']['value'], l['host_path']['value'])"
