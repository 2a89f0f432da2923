#!/bin/bash
# times experimental library variants (tools/ubench/libucf_*.so) on the bench workloads; results of the
# hot-only variants are not valid drawdowns -- timing only
mkdir -p gpurun_out
cp unconfined_amd/libucf.so /tmp/libucf_orig.so
for v in ${VARIANTS:-w2 w3 w4}; do
  cp tools/ubench/libucf_$v.so unconfined_amd/libucf.so
  for w in ${WORKLOADS:-c2 c2pp}; do
    timeout -k 10 300 python bench.py --steps 2 --warmup 1 --mode fast --no-cpu --workload $w > gpurun_out/var_${v}_$w.log 2>&1; rc=$?
    echo "[$v $w] rc=$rc $(tail -1 gpurun_out/var_${v}_$w.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), 'pts/s', round(d['roofline']['kernel_ms'],1),'ms')" 2>&1 | tail -1)"
    [ $rc -ge 124 ] && exit $rc
  done
done
cp /tmp/libucf_orig.so unconfined_amd/libucf.so
