#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python tools/gpu_explore.py $DECKS > gpurun_out/explore4.log 2>&1; rc=$?; echo "[explore] rc=$rc"; [ $rc -ge 124 ] && exit $rc
grep -E "^ +fast|rror" gpurun_out/explore4.log | cut -c1-250 | head -40
MODE=fast bash tools/gpu_workloads.sh
