#!/bin/bash
# One SQ counter pass + one timed bench line per (library variant, workload): VALU / SALU instructions per (wave, abscissa)
# of the integrate kernel, VALU-busy, issue-stall share, kernel ms.   VARIANTS="base kv4" WORKLOADS="c2 c2pp" bash tools/gpu_pmc_quick.sh
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
for v in ${VARIANTS:-base}; do
  lib=$R/tools/ubench/libucf_$v.so
  [ $v = base ] && lib=$R/unconfined_amd/libucf.so
  export UCF_LIB_PATH=$lib
  for w in ${WORKLOADS:-c2}; do
    cd /tmp; export TMPDIR=/tmp
    rm -rf $R/gpurun_out/pq_${v}_$w
    timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
      --output-format csv -d $R/gpurun_out/pq_${v}_$w -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --workload $w ${EXTRA:-} > $R/gpurun_out/pq_${v}_$w.log 2>&1; rc=$?
    [ $rc -ge 124 ] && { echo "[$v $w] pmc pass rc=$rc"; exit $rc; }
    cd $R
    timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu --workload $w ${EXTRA:-} > gpurun_out/pt_${v}_$w.log 2> gpurun_out/pt_${v}_$w.err; rc=$?
    [ $rc -ge 124 ] && { echo "[$v $w] bench rc=$rc"; exit $rc; }
    python3 - $v $w <<'PY'
import csv, glob, json, os, sys, collections
v, w = sys.argv[1:3]
nabs = {"c2": 543, "c2pp": 543, "c3": 543, "c4": 703, "c5": 543}[w]
fs = sorted(glob.glob(f"gpurun_out/pq_{v}_{w}/*/*counter_collection.csv"), key=os.path.getmtime)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[-1])):
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
line = None
for ln in open(f"gpurun_out/pt_{v}_{w}.log"):
    if ln.startswith("{"): line = json.loads(ln)
ms = {k["name"]: (k["ms"], k["launches_per_step"]) for k in line["roofline"]["kernels"]} if line else {}
for k, cs in acc.items():
    if "integrate" not in k and "dehoog_tiles" not in k and "finish" not in k: continue
    c = {n: sum(x) / len(x) for n, x in cs.items()}
    nm = k.replace("void ", "").split("(")[0]
    t = next((m for n, m in ms.items() if n == nm), (None, None))
    per = c["SQ_WAVES"] * (nabs if "integrate" in k else 1)
    print(f"[{v} {w}] {nm[-52:]:52s} ms {t[0] and round(t[0], 2)} x{t[1]} VALU/w/abs {c['SQ_INSTS_VALU'] / per:8.1f} SALU {c['SQ_INSTS_SALU'] / per:7.1f} "
          f"valu_busy {min(1.0, c['SQ_ACTIVE_INST_VALU'] * 4 / (c['GRBM_GUI_ACTIVE'] / 8 * 1024)):.3f} wait_inst/wave_cyc {c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']:.3f} waves {c['SQ_WAVES']:.0f}")
if line: print(f"[{v} {w}] {round(line['value'])} pt/s {line['ms_per_step']:.2f} ms/step")
PY
  done
done
exit 0
