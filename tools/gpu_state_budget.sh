#!/bin/bash
# launches per step against the integration-state budget (UCF_STATE_BYTES; default 8 GiB): C3 and C4
set -o pipefail
mkdir -p gpurun_out
for W in c3 c4; do
  for B in 8 24 64; do
    UCF_STATE_BYTES=$((B << 30)) timeout -k 10 400 python3 bench.py --workload $W --steps 2 --warmup 1 --no-cpu --no-other-workloads > gpurun_out/sb_${W}_$B.log 2> gpurun_out/sb_${W}_$B.err; rc=$?
    python3 - $W $B $rc <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/sb_%s_%s.log" % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%s budget %s GiB rc=%s: %.0f pt/s %.2f ms/step, %s launches x %.2f ms" % (sys.argv[1], sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"], r["kernel_launches_per_step"], r["kernel_ms"]))
except Exception as e:
    print(sys.argv[1:], "no line", e)
PY
    [ $rc -ge 124 ] && exit $rc
  done
done
exit 0
