#!/bin/bash
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/traffic_fast_*
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/traffic_fast_$C -- python3 $R/bench.py --steps 2 --warmup 1 --mode fast --no-cpu > $R/gpurun_out/traffic_fast_$C.log 2>&1
  rc=$?; echo "[fast $C] rc=$rc"; [ $rc -ge 124 ] && exit $rc
done
cd $R; MODE=fast bash tools/gpu_workloads.sh
