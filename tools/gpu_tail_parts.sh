#!/bin/bash
# finer parts for the last items of a launch (UCF_TAIL_LSPLIT / UCF_TAIL_ITEMS, launch_transform_): the full C2 sweep and its
# 1/8 shard (bench.py --nt 128) over the settings; then the parity suite with EVERY item of every launch in 8 parts
mkdir -p gpurun_out
one() { # tag, env..., -- bench args
  local tag=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 300 python bench.py --no-cpu --warmup 2 "$@" > gpurun_out/tp_$tag.log 2> gpurun_out/tp_$tag.err; local rc=$?
  echo "[$tag] rc=$rc $(tail -1 gpurun_out/tp_$tag.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['ms_per_step'],3), 'ms/step;', [(k['name'].split('::')[1][:16], round(k['ms'],3)) for k in r['kernels'] if k['ms'] > 0.05])" 2>&1 | tail -1)"
  [ $rc -ge 124 ] && exit $rc
}
for W in ${WORKLOADS:-c2}; do
  one ${W}_full_t0 UCF_TAIL_LSPLIT=0 -- --steps 10 --workload $W
  for lt in 1 2 3; do for ni in 5120 10240; do
    one ${W}_full_t${lt}_$ni UCF_TAIL_LSPLIT=$lt UCF_TAIL_ITEMS=$ni -- --steps 10 --workload $W
  done; done
  one ${W}_full_t3_2560 UCF_TAIL_LSPLIT=3 UCF_TAIL_ITEMS=2560 -- --steps 10 --workload $W
done
one sh_t0 UCF_TAIL_LSPLIT=0 -- --steps 20 --nt 128
for ns in 1 2; do for lt in 2 3; do for ni in 5120 10240; do
  one sh_n${ns}_t${lt}_$ni UCF_NSPLIT=$ns UCF_TAIL_LSPLIT=$lt UCF_TAIL_ITEMS=$ni -- --steps 20 --nt 128
done; done; done
[ -n "$SKIP_TESTS" ] && exit 0
UCF_TAIL_LSPLIT=3 UCF_TAIL_ITEMS=100000000 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stages.py -m gpu -q --no-header -p no:cacheprovider -x > gpurun_out/pytest_tail8.log 2>&1; rc=$?
echo "[pytest, every item in 8 parts] rc=$rc $(tail -1 gpurun_out/pytest_tail8.log)"; [ $rc -ne 0 ] && tail -40 gpurun_out/pytest_tail8.log
exit $rc
