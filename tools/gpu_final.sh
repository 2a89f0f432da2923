#!/bin/bash
# The evidence set of a build, one pass (tools/collect_profiles.py <tag> copies the summaries into profiles/):
#   for every BASELINE workload at FULL size on one GPU -- C2 1024x256 (headline), C2pp, C3 2048x512, C4 4096x1024,
#   C5 1024x256 -- the five driver-reproducible commands
#     rocprofv3 --kernel-trace --stats            -- python3 bench.py --steps 3 --warmup 1 --no-cpu --workload W
#     rocprofv3 --pmc <SQ issue counters>         -- python3 bench.py --steps 1 --warmup 1 --no-cpu --workload W
#     rocprofv3 --pmc <fp64 instruction counters> -- (same)
#     rocprofv3 --pmc FETCH_SIZE                  -- (same)       (separate passes: the TCC counters do not fit one)
#     rocprofv3 --pmc WRITE_SIZE                  -- (same)
#   then the GPU tests, the default bench line (with the CPU baseline) and the faithful-flavour line.
# WORKLOADS="c2 c4" restricts the first part; SKIP_TESTS=1 skips the second (but for the bench lines of BENCH_WORKLOADS).
# A call on the box is limited to 20 minutes, so a pass is two calls (the counters of a workload and its bench line in the same one):
#   WORKLOADS="c2 c2pp c3 c1" BENCH_WORKLOADS="c2pp c3 c1" bash tools/gpu_final.sh
#   WORKLOADS="c4 c5 hstorage mnm" BENCH_WORKLOADS="c4 c5 hstorage mnm" SKIP_TESTS=1 bash tools/gpu_final.sh
# and tools/collect_profiles.py once more in the build container over the merged gpurun_out/.
set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
run() { # name, rocprof args..., then bench args after --
  local name=$1; shift
  rm -rf $R/gpurun_out/$name
  timeout -k 10 500 rocprofv3 "$@" > $R/gpurun_out/$name.log 2>&1; local rc=$?
  echo "[$name] rc=$rc"; [ $rc -ge 124 ] && exit $rc; return 0
}
for W in ${WORKLOADS:-c2 c2pp c3 c4 c5 c1 hstorage mnm}; do
  B="python3 $R/bench.py --warmup 1 --no-cpu --workload $W"
  run fin_${W}_trace --kernel-trace --stats --output-format csv -d $R/gpurun_out/fin_${W}_trace -- $B --steps 3
  run fin_${W}_sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/fin_${W}_sq -- $B --steps 1
  run fin_${W}_f64 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/fin_${W}_f64 -- $B --steps 1
  run fin_${W}_FETCH_SIZE --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/fin_${W}_FETCH_SIZE -- $B --steps 1
  run fin_${W}_WRITE_SIZE --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/fin_${W}_WRITE_SIZE -- $B --steps 1
done
cd $R
# the counters of THIS build become profiles/pmc_r03.json on the box, so that the bench lines below carry their roofline
python3 tools/collect_profiles.py ${TAG:-r03} > gpurun_out/collect.log 2>&1; cp profiles/pmc_r03.json gpurun_out/pmc_r03.json
if [ -n "$SKIP_TESTS" ]; then      # second call of a pass cut in two (a call is limited to 20 minutes): the bench lines of its workloads only
  for W in ${BENCH_WORKLOADS:-}; do
    timeout -k 10 600 python bench.py --no-cpu --workload $W --steps 3 > gpurun_out/bench_$W.log 2> gpurun_out/bench_$W.err; echo "[bench $W] rc=$?"
  done
  exit 0
fi
timeout -k 10 1000 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "[pytest] rc=$rc $(tail -1 gpurun_out/pytest_gpu.log)"; [ $rc -ge 124 ] && exit $rc
timeout -k 10 600 python bench.py > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err; rc=$?; echo "[bench] rc=$rc"; [ $rc -ge 124 ] && exit $rc
timeout -k 10 600 python bench.py --mode faithful --no-cpu > gpurun_out/bench_faithful.log 2> gpurun_out/bench_faithful.err; rc=$?; echo "[bench faithful] rc=$rc"
for W in ${BENCH_WORKLOADS:-c2pp c3 c4 c5 c1 hstorage mnm}; do
  timeout -k 10 600 python bench.py --no-cpu --workload $W --steps 3 > gpurun_out/bench_$W.log 2> gpurun_out/bench_$W.err; echo "[bench $W] rc=$?"
done
# the per-rank work of a strong-scaling run of C2 at N = 8 (128 of the 1024 time rows) on this one GPU
timeout -k 10 300 python bench.py --no-cpu --nt 128 --steps 20 --warmup 2 > gpurun_out/bench_nt128.log 2> gpurun_out/bench_nt128.err; echo "[bench nt128] rc=$?"
# what each rank of a strong-scaling run at N = 2, 4, 8 REALLY computes: contiguous blocks of the sweep's own time rows, one after the other
for G in 2 4 8; do timeout -k 10 300 python tools/shard_balance.py c2 $G > gpurun_out/shard_balance_$G.log 2> gpurun_out/shard_balance_$G.err; echo "[shard_balance $G] rc=$? $(tail -c 400 gpurun_out/shard_balance_$G.log)"; done
# rehearsal of the N > 1 path on this one-GPU box: bench.py starts its own two ranks (both on cuda:0, gloo), strong and weak
for S in strong weak; do
  UCF_BENCH_ONE_DEVICE=1 UCF_BENCH_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --steps 3 --scaling $S > gpurun_out/bench_gpus2_$S.log 2> gpurun_out/bench_gpus2_$S.err; echo "[bench --gpus 2 $S] rc=$?"
done
