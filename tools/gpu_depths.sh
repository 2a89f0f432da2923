#!/bin/bash
# contour-style calls (21 depths, walked two at a time): every family, running areas in registers on / off
set -o pipefail
mkdir -p gpurun_out
for D in ${DECKS:-c2_neuman74_fullpen neuman74_partpen c3_moench c4_malama_partpen}; do
  for V in 0 1; do
    echo "UCF_NZC2=$V $(UCF_NZC2=$V timeout -k 10 200 python3 tools/bench_depths.py $D ${NZ:-21} 2>&1 | tail -1)"
  done
done
