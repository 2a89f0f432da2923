#!/bin/bash
# kernel-trace profile of one bench workload: WORKLOAD=c2 MODE=fast bash tools/gpu_prof.sh
mkdir -p gpurun_out
W=${WORKLOAD:-c2}; M=${MODE:-fast}
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_$W
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$W -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --mode $M --no-cpu --workload $W ${EXTRA:-} > $GRAFT_REPO_ROOT/gpurun_out/prof_$W.log 2>&1; rc=$?
echo "[prof $W] rc=$rc"; [ $rc -ge 124 ] && exit $rc
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_$W -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>4s} avg_ms {float(r['AverageNs'])/1e6:9.3f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} {r['Percentage']:>6s}%")
PY
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_$W.log | cut -c1-300
