#!/bin/bash
# A/B build of the fast flavour with extra -D switches:  tools/ubench/build_variant.sh <tag> [-DUCF_FOLD_WAVES=6 ...]
# -> tools/ubench/libucf_<tag>.so (the other objects come from the regular build); time it with run_variants.sh
set -e
tag=$1; shift
here=$(cd "$(dirname "$0")" && pwd)
src=$here/../../unconfined_amd/csrc
objs=""
# LAYOUTS="1": only that translation unit is rebuilt with the switches (enough to time a grid workload), the others come from the regular build
for l in 0 1 2 3; do
  case " ${LAYOUTS:-0 1 2 3} " in
    *" $l "*) hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -ffp-contract=fast "$@" -c $src/ucf_kernels_fast_l$l.hip -o /tmp/ucf_var_${tag}_l$l.o &
              objs="$objs /tmp/ucf_var_${tag}_l$l.o" ;;
    *) objs="$objs $src/build/kernels_fast_l$l.o" ;;
  esac
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $here/libucf_$tag.so $src/build/kernels_faithful.o $objs $src/build/peak.o $src/build/api.o
echo "built $here/libucf_$tag.so"
