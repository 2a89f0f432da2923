#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python tools/gpu_explore.py > gpurun_out/explore2.log 2>&1; rc=$?; echo "[explore] rc=$rc"; [ $rc -ge 124 ] && exit $rc
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --mode fast --no-cpu > gpurun_out/bench_fast.log 2>&1; rc=$?; echo "[bench fast] rc=$rc"; tail -1 gpurun_out/bench_fast.log | cut -c1-400; [ $rc -ge 124 ] && exit $rc
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --mode faithful --no-cpu > gpurun_out/bench_faithful.log 2>&1; rc=$?; echo "[bench faithful] rc=$rc"; tail -1 gpurun_out/bench_faithful.log | cut -c1-400
