#!/bin/bash
# usage: ab.sh tag  -- alternates libdd_base.so / libdd_nt.so, two runs each
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for v in base nt; do
    cp driving-dirty_amd/csrc/libdd_$v.so driving-dirty_amd/csrc/libdd_hotpath.so
    tools/gpu_steps.sh "200|ab_${v}_$i|python bench.py --no-cpu-baseline --steps 20 --warmup 5" > /dev/null || exit 1
    echo "$v $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ab_${v}_$i.log)"
  done
done
