#!/bin/bash
# usage: ab_env.sh VAR=VALUE  -- alternates the default and the given environment setting, two bench runs each (same box)
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  tools/gpu_steps.sh "200|abe_base_$i|python bench.py --no-cpu-baseline --steps 20 --warmup 5" > /dev/null || exit 1
  echo "base $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/abe_base_$i.log)"
  tools/gpu_steps.sh "200|abe_alt_$i|$1 python bench.py --no-cpu-baseline --steps 20 --warmup 5" > /dev/null || exit 1
  echo "$1 $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/abe_alt_$i.log)"
done
