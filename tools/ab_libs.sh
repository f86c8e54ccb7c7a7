#!/bin/bash
# ab_libs.sh "CMD" NAME [NAME ...]: run CMD with the regular library and with each variant build (tools/build_variant.sh), twice, on one box.
cmd=$1; shift
for rep in 1 2; do
  echo "== regular"; timeout -k 10 300 bash -c "$cmd" 2>&1 | grep -v amdgpu.ids
  for n in "$@"; do
    echo "== $n"; DD_HOTPATH_LIB=driving-dirty_amd/csrc/build/libdd_$n.so timeout -k 10 300 bash -c "$cmd" 2>&1 | grep -v amdgpu.ids
  done
done
