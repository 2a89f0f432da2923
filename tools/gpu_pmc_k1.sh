#!/bin/bash
# PMC passes on the bench workload (no trace domains): instruction mix and issue statistics per kernel
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; W=${WORKLOAD:-c2}
rm -rf $R/gpurun_out/pmck_*
pass() { name=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmck_$name -- python3 $R/bench.py --steps 1 --warmup 1 --mode fast --no-cpu --workload $W > $R/gpurun_out/pmck_$name.log 2>&1; rc=$?; echo "[$name] rc=$rc"; [ $rc -ge 124 ] && exit $rc; return 0; }
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE && \
pass b SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 GRBM_GUI_ACTIVE && \
pass c SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE
cd $R && python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmck_*')):
    for f in glob.glob(d + '/*/*counter_collection.csv'):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0][-40:]
            agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        for k, v in agg.items():
            if 'integrate' in k or 'point_kernel' in k or 'dehoog' in k:
                print(d.split('/')[-1], k, {a: f'{b:.4g}' for a, b in v.items()})
PY
