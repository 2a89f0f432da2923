#!/bin/bash
# One kernel instantiation of the fast flavour compiled on its own (seconds): resource usage + ISA listing.
#   tools/probe_kernel.sh 'integrate_kernel<2,1,6,false,true,false>' [out.s] [-D...]
set -e
k=$1; out=${2:-/tmp/probe.s}; shift; shift || true
here=$(cd "$(dirname "$0")" && pwd)
src=$here/../unconfined_amd/csrc
cat > /tmp/ucf_probe.hip <<EOS
#define UCF_FAST 1
#define UCF_NS ucf_fast
#define UCF_TU 1
#define UCF_PROBE $k
#include "$src/ucf_device.h"
EOS
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -ffp-contract=fast "$@" --cuda-device-only -S /tmp/ucf_probe.hip -o $out \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|VGPRs Spill|SGPRs:|SGPRs Spill|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: [^ ]* *//' | tr '\n' ' '
echo
