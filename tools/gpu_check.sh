#!/bin/bash
# quick GPU session: full parity suite, then the five bench workloads (fast flavour)
mkdir -p gpurun_out
timeout -k 10 ${PYTEST_TO:-1000} python -m pytest tests -m gpu -q --no-header -p no:cacheprovider ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1; rc=$?
echo "[pytest] rc=$rc"; grep -E "^(FAILED|ERROR)|passed|failed|Error" gpurun_out/pytest_gpu.log | tail -12; [ $rc -ne 0 ] && tail -40 gpurun_out/pytest_gpu.log && exit $rc
MODE=fast bash tools/gpu_workloads.sh
