#!/bin/bash
# PMC passes for the point kernel on a reduced sweep (no trace domains together with --pmc)
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
MODE=${1:-fast}
rocprofv3 -L > $R/gpurun_out/counters_list.txt 2>&1
pass() { name=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_${MODE}_$name -- python3 $R/bench.py --steps 1 --warmup 1 --mode $MODE --no-cpu --nt 256 > $R/gpurun_out/pmc_${MODE}_$name.log 2>&1; echo "[$name] rc=$?"; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD && \
pass sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_IFETCH && \
pass tcc1 FETCH_SIZE && pass tcc2 WRITE_SIZE && pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
find $R/gpurun_out -name "*counter_collection.csv" | head
