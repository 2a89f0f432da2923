#!/bin/bash
# fast vs faithful on random parameter sets, per family group: NSETS (default 150), SEED, groups in FGROUPS ("0 2,12 16 ...")
set -o pipefail
mkdir -p gpurun_out
for G in ${FGROUPS:-0 2,12 16}; do
  UCF_FUZZ_MODELS=$G timeout -k 10 500 python3 tools/fuzz_flavours.py ${NSETS:-150} ${SEED:-5} > gpurun_out/fuzz_$G.log 2>&1; rc=$?
  echo "[fuzz $G] rc=$rc"; tail -4 gpurun_out/fuzz_$G.log; [ $rc -ge 124 ] && exit $rc
done
exit 0
