#!/bin/bash
# end-of-round evidence: full GPU tests, default bench, kernel trace, traffic and SQ counters of the final build
set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "[pytest] rc=$rc $(tail -1 gpurun_out/pytest_gpu.log)"; [ $rc -ge 124 ] && exit $rc
timeout -k 10 600 python bench.py > gpurun_out/bench_default.log 2>&1; rc=$?; echo "[bench] rc=$rc"; [ $rc -ge 124 ] && exit $rc
timeout -k 10 600 python bench.py --mode faithful --no-cpu > gpurun_out/bench_faithful.log 2>&1; rc=$?; echo "[bench faithful] rc=$rc"; [ $rc -ge 124 ] && exit $rc
cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/final_*
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final_trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu > $R/gpurun_out/final_trace.log 2>&1; echo "[trace] rc=$?"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/final_$C -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu > $R/gpurun_out/final_$C.log 2>&1; rc=$?; echo "[$C] rc=$rc"; [ $rc -ge 124 ] && exit $rc
done
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/final_sq -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu > $R/gpurun_out/final_sq.log 2>&1; rc=$?; echo "[sq] rc=$rc"; [ $rc -ge 124 ] && exit $rc
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/final_f64 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu > $R/gpurun_out/final_f64.log 2>&1; echo "[f64] rc=$?"
