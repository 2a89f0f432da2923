#!/bin/bash
# HBM traffic of one full-size C2 launch: separate --pmc passes (FETCH_SIZE / WRITE_SIZE cannot share a pass)
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for MODE in fast faithful; do
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/traffic_${MODE}_$C -- python3 $R/bench.py --steps 2 --warmup 1 --mode $MODE --no-cpu > $R/gpurun_out/traffic_${MODE}_$C.log 2>&1
  rc=$?; echo "[$MODE $C] rc=$rc"; [ $rc -ge 124 ] && exit $rc
done
done
cd $R && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final_fast -- python3 $R/bench.py --steps 3 --warmup 1 --mode fast --no-cpu > $R/gpurun_out/prof_final_fast.log 2>&1; echo "[trace fast] rc=$?"
