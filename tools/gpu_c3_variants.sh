#!/bin/bash
# C3 (two depths, unfolded water-table kernel): running areas in registers (NZC = 2) x 3 / 4 waves per SIMD
set -o pipefail
mkdir -p gpurun_out
for V in "0 3" "1 3" "1 4" "0 4"; do
  set -- $V
  UCF_NZC2=$1 UCF_UNFOLD_WAVES_RT=$2 timeout -k 10 300 python3 bench.py --workload ${W:-c3} --steps 2 --warmup 1 --no-cpu --no-other-workloads > gpurun_out/c3v_$1_$2.log 2> gpurun_out/c3v_$1_$2.err; rc=$?
  python3 - "$1" "$2" $rc <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/c3v_%s_%s.log" % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
    k = [(x["name"][-34:], round(x["ms"], 2)) for x in d["roofline"]["kernels"] if "integrate" in x["name"]] if d.get("roofline") else d.get("kernels")
    print("NZC2=%s waves=%s rc=%s: %.0f pt/s %.2f ms/step %s" % (sys.argv[1], sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"], k))
except Exception as e:
    print("NZC2=%s waves=%s rc=%s: no line (%s)" % (sys.argv[1], sys.argv[2], sys.argv[3], e))
PY
  [ $rc -ge 124 ] && exit $rc
done
exit 0
