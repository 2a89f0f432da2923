#!/bin/bash
# one GPU-box session: parity tests, bench (both modes), rocprof kernel trace.  Stops after a timeout.
set -o pipefail
mkdir -p gpurun_out
run() { # name, timeout, cmd...
  local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > gpurun_out/$name.log 2>&1
  local rc=$?
  echo "[$name] rc=$rc"
  if [ $rc -ge 124 ]; then echo "[$name] timed out/killed - stopping"; exit $rc; fi
  return 0
}
run pytest_gpu ${PYTEST_TO:-900} python -m pytest tests -m gpu -q --no-header -p no:cacheprovider ${PYTEST_ARGS:-}
tail -15 gpurun_out/pytest_gpu.log
run bench_fast 600 python bench.py --steps 3 --warmup 1 --mode fast
tail -2 gpurun_out/bench_fast.log
run bench_faithful 600 python bench.py --steps 3 --warmup 1 --mode faithful --no-cpu
tail -1 gpurun_out/bench_faithful.log
cd /tmp && export TMPDIR=/tmp
run_prof() { timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --mode $1 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof_$1.log 2>&1; echo "[prof_$1] rc=$?"; }
run_prof fast
find $GRAFT_REPO_ROOT/gpurun_out/prof_fast -name "*stats*" | head
