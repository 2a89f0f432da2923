#!/bin/bash
mkdir -p gpurun_out
PYTEST_ARGS="-k 'full_size or fortran or truth'" 
timeout -k 10 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -k "full_size or fortran or truth or edge" > gpurun_out/pytest_gpu.log 2>&1; rc=$?
echo "[pytest] rc=$rc"; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest_gpu.log | tail -12; [ $rc -ge 124 ] && exit $rc
MODE=fast bash tools/gpu_workloads.sh
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --mode fast --layout sample --no-cpu | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('layout=sample', round(d['value']), d['roofline']['kernel_ms'])"
