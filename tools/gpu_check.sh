#!/bin/bash
# quick GPU session: full parity suite, then the default bench line
mkdir -p gpurun_out
timeout -k 10 ${PYTEST_TO:-1000} python -m pytest tests -m gpu -q --no-header -p no:cacheprovider ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1; rc=$?
echo "[pytest] rc=$rc"; grep -E "^(FAILED|ERROR)|passed|failed|Error" gpurun_out/pytest_gpu.log | tail -12; [ $rc -ne 0 ] && tail -60 gpurun_out/pytest_gpu.log
[ $rc -ge 124 ] && exit $rc
timeout -k 10 600 python bench.py > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err; rc2=$?; echo "[bench] rc=$rc2"; tail -c 3000 gpurun_out/bench_default.log; tail -5 gpurun_out/bench_default.err
exit $rc
