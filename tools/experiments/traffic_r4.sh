#!/bin/bash
# HBM traffic of the conv kernels as EVIDENCE: the TCC's memory-side read requests by SIZE CLASS
# (TCC_EA0_RDREQ_32B / _64B / _128B: bytes = 32 a + 64 b + 128 c, no correction factor to guess) next to FETCH_SIZE, calibrated
# on a 1 GiB float4 copy and on one known-byte single-read layer per STAGING SHAPE of the plan; then per launch of one forward
# of the committed plan (eager, one frame in flight, and once more with graph replay + eight frames in flight).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r4traffic; mkdir -p $O /tmp/raw
cd /tmp
RD="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
WR="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
pass() { # name counters -- command...
  local name=$1; shift
  local ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc "${ctr[@]}" -d /tmp/raw -o $name --output-format csv -- "$@" > $O/${name}.log 2>&1 \
    && cp /tmp/raw/${name}_counter_collection.csv $O/${name}_dispatches.csv && python $R/tools/summarize_pmc.py /tmp/raw/${name}_counter_collection.csv $O/${name}.csv \
    || echo "pass $name failed" >> $O/failed.txt
}
# --- calibration: absolute reference + one single-read layer per staging shape (Cout = one channel tile: the input is staged once)
pass cal_copy_rd $RD -- $R/tools/microbench/copy_known_bytes.bin
pass cal_copy_fetch FETCH_SIZE -- $R/tools/microbench/copy_known_bytes.bin
pass cal_copy_wr $WR -- $R/tools/microbench/copy_known_bytes.bin
pass cal_copy_write WRITE_SIZE -- $R/tools/microbench/copy_known_bytes.bin
#               kind tile split cin h w cout
for K in "t8x16_64Brows 0 0 1 256 256 256 128" "t4x32_128Brows 0 6 1 256 256 256 64" "persistent_p16 16 34 1 256 256 256 64" "wino44_16Bpieces 14 32 1 256 256 256 64" "wino22_w4 8 30 1 256 256 256 64" "k32_t4x32 10 6 1 256 256 256 64"; do
  set -- $K; n=$1; shift
  pass cal_${n}_rd $RD -- python $R/tools/one_conv.py "$@" 0 4
  pass cal_${n}_fetch FETCH_SIZE -- python $R/tools/one_conv.py "$@" 0 4
  pass cal_${n}_wr $WR -- python $R/tools/one_conv.py "$@" 0 4
done
# --- the forward of the committed plan
cd $R
python tools/dump_ops.py > $O/ops_1024.json 2> $O/dump_ops.err
cd /tmp
P="python $R/bench.py --steps 8 --warmup 2 --cpu-frames 0 --host-frames 0 --inflight 1 --profile-frames 1 --graph 0"
pass fwd_rd $RD -- $P
pass fwd_wr $WR -- $P
pass fwd_fetch FETCH_SIZE -- $P
PG="python $R/bench.py --steps 16 --warmup 8 --cpu-frames 0 --host-frames 0 --profile-frames 1"
pass fwdgraph_rd $RD -- $PG
pass fwdgraph_wr $WR -- $PG
ls $O; cat $O/failed.txt 2>/dev/null
