#!/bin/bash
# times experimental library variants (tools/ubench/libucf_<tag>.so, built by build_variant.sh) on the bench workloads.
# The product library is never touched: the variant is loaded through UCF_LIB_PATH (unconfined_amd/lib.py).
mkdir -p gpurun_out
for v in ${VARIANTS:-base}; do
  lib=$PWD/tools/ubench/libucf_$v.so
  [ $v = base ] && lib=$PWD/unconfined_amd/libucf.so
  for w in ${WORKLOADS:-c2 c2pp}; do
    UCF_LIB_PATH=$lib timeout -k 10 300 python bench.py --steps 3 --warmup 1 --mode fast --no-cpu --workload $w ${EXTRA:-} > gpurun_out/var_${v}_$w.log 2> gpurun_out/var_${v}_$w.err; rc=$?
    echo "[$v $w] rc=$rc $(tail -1 gpurun_out/var_${v}_$w.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']), 'pt/s', round(d['ms_per_step'],2), 'ms/step', [(k['name'].split('::')[1][:22], round(k['ms'],2)) for k in r['kernels'] if k['ms'] > 0.5])" 2>&1 | tail -1)"
    [ $rc -ge 124 ] && exit $rc
  done
done
exit 0
