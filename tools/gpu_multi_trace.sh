#!/bin/bash
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/multi_trace
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/multi_trace -- python3 $R/tools/bench_multi.py ${NPL:-16} ${NPTS:-240} > $R/gpurun_out/multi_trace.log 2>&1; echo rc=$?
tail -1 $R/gpurun_out/multi_trace.log
f=$(find $R/gpurun_out/multi_trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows=[r for r in rows if 'ucf_' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
last=rows[-120:]
for r in last[:60]:
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:10.1f} {(int(r['End_Timestamp'])-t0)/1e3:10.1f} us  q{r.get('Queue_Id','?')} s{r.get('Stream_Id','?')} {r['Kernel_Name'][:60]}")
PY
