#!/bin/bash
# build_variant.sh NAME SOURCE.hip [-DFLAG ...]: a second build of the library with ONE source recompiled under extra flags, for A/B runs of
# two builds on one box (DD_HOTPATH_LIB=driving-dirty_amd/csrc/build/libdd_NAME.so).  The other objects come from the regular build.
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../driving-dirty_amd/csrc"
stem=${src%.hip}
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c "$src" -o "build/var_${name}_${stem}.o"
objs=$(ls build/*.o | grep -v "/var_" | grep -v "/${stem}.o" | grep -v diag)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "build/libdd_${name}.so" $objs "build/var_${name}_${stem}.o"
echo "built driving-dirty_amd/csrc/build/libdd_${name}.so"
