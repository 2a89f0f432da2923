#!/bin/bash
# a variant library (tools/ubench/libucf_<tag>.so, loaded through UCF_LIB_PATH) against the product on one box:  VAR=<tag> WORKLOADS="c2pp c4" bash tools/gpu_variant.sh
mkdir -p gpurun_out
one() { local tag=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 300 python bench.py --no-cpu --warmup 2 "$@" > gpurun_out/v_$tag.log 2> gpurun_out/v_$tag.err; local rc=$?
  echo "[$tag] rc=$rc $(tail -1 gpurun_out/v_$tag.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['ms_per_step'],3), 'ms/step;', [(k['name'].split('::')[1][:16], round(k['ms'],3)) for k in r['kernels'] if k['ms'] > 0.05])" 2>&1 | tail -1)"
  [ $rc -ge 124 ] && exit $rc; return 0; }
V=$PWD/tools/ubench/libucf_${VAR}.so
for w in ${WORKLOADS:-c2pp c3 c4}; do
  one ${w}_prod X=1 -- --steps 3 --workload $w
  one ${w}_$VAR UCF_LIB_PATH=$V -- --steps 3 --workload $w
done
exit 0
