#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python tools/gpu_explore.py ${DECKS:-c2_neuman74_fullpen neuman74_partpen c3_moench c4_malama_partpen hantush_lay1 hantush_lay3 malama_fullpen} > gpurun_out/explore3.log 2>&1; rc=$?; echo "[explore] rc=$rc"; [ $rc -ge 124 ] && exit $rc
grep -E "^ +fast|Error|error" gpurun_out/explore3.log | cut -c1-250 | head -40
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --mode fast --no-cpu > gpurun_out/bench_fast.log 2>&1; rc=$?; echo "[bench fast] rc=$rc"; tail -1 gpurun_out/bench_fast.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
