#!/bin/bash
# quick look at a change of the fast kernels: bench lines of the workloads (per-kernel ms), then the cut / parity / stage tests
mkdir -p gpurun_out
one() { local tag=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 300 python bench.py --no-cpu --warmup 2 "$@" > gpurun_out/q_$tag.log 2> gpurun_out/q_$tag.err; local rc=$?
  echo "[$tag] rc=$rc $(tail -1 gpurun_out/q_$tag.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']), 'pt/s', round(d['ms_per_step'],3), 'ms/step;', [(k['name'].split('::')[1][:16], round(k['ms'],3)) for k in r['kernels'] if k['ms'] > 0.05])" 2>&1 | tail -1)"
  [ $rc -ge 124 ] && exit $rc; return 0; }
for rep in a b; do one c2_$rep X=1 -- --steps 10; done
for w in ${WORKLOADS:-c2pp c3 c4 c5 mnm hstorage c1}; do one $w X=1 -- --steps 3 --workload $w; done
one sh X=1 -- --steps 20 --nt 128
[ -n "$SKIP_TESTS" ] && exit 0
timeout -k 10 1000 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1; rc=$?
echo "[pytest] rc=$rc $(tail -1 gpurun_out/pytest_gpu.log)"; [ $rc -ne 0 ] && tail -40 gpurun_out/pytest_gpu.log
exit $rc
