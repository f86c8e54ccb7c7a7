#!/bin/bash
# usage: ab_multi.sh "VAR=a" "VAR=b" ...  -- bench.py once per setting, two rounds (same box); "" = default
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for v in "$@"; do
    tag=$(echo "$v" | tr -c 'A-Za-z0-9' '_')
    tools/gpu_steps.sh "200|abm_${tag}_$i|$v python bench.py --no-cpu-baseline --steps 20 --warmup 5" > /dev/null || exit 1
    echo "[$v] $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/abm_${tag}_$i.log)"
  done
done
