#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 ${PYTEST_TO:-900} python -m pytest tests -m gpu -q --no-header -p no:cacheprovider ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1
echo "rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest_gpu.log | tail -40
