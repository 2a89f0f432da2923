#!/bin/bash
# the 1/8 shard of C2 (bench.py --nt 128: what one rank of a strong-scaling run at N = 8 computes) with 1 / 2 / 4 / 8 parts per work item
mkdir -p gpurun_out
for ns in ${NSPLITS:-1 2 4 8}; do
  for nt in ${NTS:-128 256}; do
    UCF_NSPLIT=$ns timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu --nt $nt > gpurun_out/sh_${ns}_$nt.log 2> gpurun_out/sh_${ns}_$nt.err; rc=$?
    echo "[nsplit $ns nt $nt] rc=$rc $(tail -1 gpurun_out/sh_${ns}_$nt.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), 'ms/step', [(k['name'].split('::')[1][:20], round(k['ms'],3)) for k in d['roofline']['kernels']])" 2>&1 | tail -1)"
    [ $rc -ge 124 ] && exit $rc
  done
done
exit 0
