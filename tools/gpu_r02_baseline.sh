#!/bin/bash
# round-2 baseline of the round-1 kernels: the four non-headline BASELINE workloads at FULL size on one GPU
# (kernel trace + two PMC passes each) and the balance of contiguous block partitions of C2
set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for W in ${WORKLOADS:-c2pp c3 c4 c5}; do
  rm -rf $R/gpurun_out/base_${W}_*
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/base_${W}_trace -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --workload $W > $R/gpurun_out/base_${W}_trace.log 2>&1; rc=$?; echo "[$W trace] rc=$rc $(tail -1 $R/gpurun_out/base_${W}_trace.log | cut -c1-160)"; [ $rc -ge 124 ] && exit $rc
  timeout -k 10 500 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/base_${W}_sq -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --workload $W > $R/gpurun_out/base_${W}_sq.log 2>&1; rc=$?; echo "[$W sq] rc=$rc"; [ $rc -ge 124 ] && exit $rc
  timeout -k 10 500 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/base_${W}_f64 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --workload $W > $R/gpurun_out/base_${W}_f64.log 2>&1; rc=$?; echo "[$W f64] rc=$rc"; [ $rc -ge 124 ] && exit $rc
done
cd $R
for W in c2 c2pp; do
  timeout -k 10 300 python3 tools/shard_balance.py $W 8 > gpurun_out/balance_$W.json 2> gpurun_out/balance_$W.err; echo "[balance $W] rc=$? $(cat gpurun_out/balance_$W.json | cut -c1-600)"
done
