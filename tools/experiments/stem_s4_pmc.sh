#!/bin/bash
# FaceBoxes conv1 alone at batch 16: the generic class-6 kernel's tiles vs conv_stem_s4.h, then counters of the latter
set -e
O=gpurun_out/stem_s4; mkdir -p $O /tmp/raw; export TMPDIR=/tmp
python - <<'PY' | tee $O/times.txt
import sys, os
sys.path.insert(0, "tools")
import conv_bench as cb
for kind, tiles in ((6, range(37)), (20, (36,))):
    for t in tiles:
        ms = cb.bench(kind, t, 1, 3, 1024, 1024, 24, iters=20, B=16)
        if ms:
            print("%-10s %-10s %7.1f us  %5.1f algorithmic TFLOP/s" % (cb.KIND[kind], cb.TILE[t], ms * 1e3, 2.0 * 16 * 256 * 256 * 24 * 147 / ms / 1e9))
PY
cat > /tmp/one16.py <<'PY'
import sys
sys.path.insert(0, "tools")
import conv_bench as cb
print(cb.bench(20, 36, 1, 3, 1024, 1024, 24, iters=6, B=16))
PY
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_LEVEL_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set -d /tmp/raw -o s4_$n --output-format csv -- python /tmp/one16.py > $O/pmc_$n.log 2>&1 || true
  python tools/summarize_pmc.py /tmp/raw/s4_${n}_counter_collection.csv $O/pmc_$n.csv || true
  grep stem_s4 $O/pmc_$n.csv | sed 's/^.*ConvArgs)",//' || true
done
