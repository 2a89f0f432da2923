#!/bin/bash
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmcf_*
pass() { name=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmcf_$name -- python3 $R/bench.py --steps 2 --warmup 1 --mode fast --no-cpu > $R/gpurun_out/pmcf_$name.log 2>&1; rc=$?; echo "[$name] rc=$rc"; [ $rc -ge 124 ] && exit $rc; return 0; }
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE && \
pass b SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
