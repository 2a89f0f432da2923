#!/bin/bash
# PMC passes for the point kernel (no trace domains together with --pmc)
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
MODE=${1:-fast}; WL=${2:-c2}; NT=${3:-256}
pass() { name=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_${MODE}_${WL}_$name -- python3 $R/bench.py --steps 1 --warmup 1 --mode $MODE --no-cpu --workload $WL --nt $NT > $R/gpurun_out/pmc_${MODE}_${WL}_$name.log 2>&1; echo "[$name] rc=$?"; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD && \
pass sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_IFETCH
