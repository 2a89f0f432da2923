#!/bin/bash
# persistent grid + work counter (default) against one workgroup per four work units (UCF_PERSIST=0: the static scheme of the
# builds before), alternating on one box: full C2, its 1/8 shard (bench.py --nt 128), C2pp / C5 / C3
mkdir -p gpurun_out
one() { # tag, env..., -- bench args
  local tag=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 300 python bench.py --no-cpu --warmup 2 "$@" > gpurun_out/ps_$tag.log 2> gpurun_out/ps_$tag.err; local rc=$?
  echo "[$tag] rc=$rc $(tail -1 gpurun_out/ps_$tag.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['ms_per_step'],3), 'ms/step;', [(k['name'].split('::')[1][:16], round(k['ms'],3)) for k in r['kernels'] if k['ms'] > 0.05])" 2>&1 | tail -1)"
  [ $rc -ge 124 ] && exit $rc
}
for rep in a b; do for ps in 0 1; do
  one c2_p${ps}_$rep UCF_PERSIST=$ps -- --steps 10
  one sh_p${ps}_$rep UCF_PERSIST=$ps -- --steps 20 --nt 128
done; done
one c2_p0_t0 UCF_PERSIST=0 UCF_TAIL_LSPLIT=0 -- --steps 10
one sh_p0_t0 UCF_PERSIST=0 UCF_TAIL_LSPLIT=0 -- --steps 20 --nt 128
for w in ${WORKLOADS:-c2pp c5 c3}; do for ps in 0 1; do one ${w}_p${ps} UCF_PERSIST=$ps -- --steps 3 --workload $w; done; done
exit 0
