#!/bin/bash
# Experimental build of the same ABI: recompile the listed translation units with extra flags, link with the
# product's other objects (build/*.o from __graft_entry__.build()) into exp/<name>.so (RCED_LIB=exp/<name>.so).
# Usage: tools/mkexp.sh <name> "<unit1 unit2 ...>" [hipcc flags...]     e.g. tools/mkexp.sh occ2 "train_api" -DRCED_TM_OCC=2
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; units=$2; shift 2
mkdir -p $R/exp/obj_$name
objs=""
for u in rced_api kernels_fused audio_api train_api train_mfma_v2; do
  if [[ " $units " == *" $u "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c -o $R/exp/obj_$name/$u.o $R/fullycnnspeechenhancement_amd/csrc/$u.hip &
    objs="$objs $R/exp/obj_$name/$u.o"
  else
    objs="$objs $R/build/$u.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $R/exp/$name.so $objs
echo "built exp/$name.so"
