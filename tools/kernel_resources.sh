#!/bin/bash
# register / scratch / occupancy table of the kernels of one flavour: tools/kernel_resources.sh fast|faithful [extra flags]
cd "$(dirname "$0")/../unconfined_amd/csrc"
fl=${1:-fast}; shift
c=off; [ "$fl" = fast ] && c=fast
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=$c "$@" -c ucf_kernels_$fl.hip -o /tmp/kres_$fl.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|SGPRs:|VGPRs:|Spill|ScratchSize|Occupancy" | sed 's/.*remark: [^ ]* *//; s/\[-Rpass.*//' \
 | awk '/Function Name/ {if (line) print line; line=$3; next} {gsub(/^ +/,""); line=line " | " $0} END {print line}' \
 | sed -E 's/_ZN[0-9]+ucf_(fast|faithful)[0-9]+([a-z_]+kernel)(ILi([0-9])ELi([0-9])EE|ILi([0-9])EE)?[^ ]*/\2<\4\5\6>/; s/TotalSGPRs: /S/; s/VGPRs: /V/; s/ScratchSize \[bytes\/lane\]: /scr/; s/Occupancy \[waves\/SIMD\]: /occ/; s/SGPRs Spill: /Ssp/; s/VV/V/; s/VGPRs Spill: /Vsp/'
