#!/bin/bash
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/multi_prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/multi_prof -- python3 $R/tools/bench_multi.py ${NPL:-256} ${NPTS:-240} > $R/gpurun_out/multi_prof.log 2>&1; echo rc=$?
tail -1 $R/gpurun_out/multi_prof.log
f=$(find $R/gpurun_out/multi_prof -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>4s} avg_ms {float(r['AverageNs'])/1e6:9.3f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} {r['Percentage']:>6s}%")
PY
