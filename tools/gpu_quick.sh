#!/bin/bash
# quick look at a build: (optionally) the GPU tests, then one bench line per workload (fast flavour, no CPU leg)
mkdir -p gpurun_out
if [ -z "$SKIP_TESTS" ]; then
  timeout -k 10 ${PYTEST_TO:-1000} python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1; rc=$?
  echo "[pytest] rc=$rc $(tail -1 gpurun_out/pytest_gpu.log)"; [ $rc -ne 0 ] && tail -50 gpurun_out/pytest_gpu.log; [ $rc -ge 124 ] && exit $rc
fi
for w in ${WORKLOADS:-c2 c2pp c3 c4 c5}; do
  timeout -k 10 600 python bench.py --steps 3 --warmup 1 --no-cpu --workload $w ${EXTRA:-} > gpurun_out/q_$w.log 2> gpurun_out/q_$w.err; rc=$?
  echo "[$w] rc=$rc $(tail -1 gpurun_out/q_$w.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']), 'pt/s', round(d['ms_per_step'],2), 'ms/step; kernels:', [(k['name'].split('::')[1][:28], round(k['ms'],2), k['launches_per_step']) for k in r['kernels']])" 2>&1 | tail -1)"
  [ $rc -ge 124 ] && exit $rc
done
exit 0
