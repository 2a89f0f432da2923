#!/bin/bash
# the library of the commit before (tools/ubench/libucf_prev.so, loaded through UCF_LIB_PATH) against the product, alternating on one box
mkdir -p gpurun_out
one() { # tag, env..., -- bench args
  local tag=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 300 python bench.py --no-cpu --warmup 2 "$@" > gpurun_out/ab_$tag.log 2> gpurun_out/ab_$tag.err; local rc=$?
  echo "[$tag] rc=$rc $(tail -1 gpurun_out/ab_$tag.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['ms_per_step'],3), 'ms/step;', [(k['name'].split('::')[1][:16], round(k['ms'],3)) for k in r['kernels'] if k['ms'] > 0.05])" 2>&1 | tail -1)"
  [ $rc -ge 124 ] && exit $rc
}
PREV=$PWD/tools/ubench/libucf_prev.so
for rep in a b; do
  one c2_prev_$rep UCF_LIB_PATH=$PREV -- --steps 10
  one c2_new_$rep X=1 -- --steps 10
  one sh_prev_$rep UCF_LIB_PATH=$PREV -- --steps 20 --nt 128
  one sh_new_$rep X=1 -- --steps 20 --nt 128
done
for w in ${WORKLOADS:-c2pp c5 c3 c4}; do
  one ${w}_prev UCF_LIB_PATH=$PREV -- --steps 3 --workload $w
  one ${w}_new X=1 -- --steps 3 --workload $w
done
exit 0
