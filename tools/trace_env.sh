#!/bin/bash
# kernel trace of the headline step under two environments on one box: tools/trace_env.sh "ENV_A" "ENV_B"
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
n=0
for e in "$@"; do
  n=$((n+1))
  env $e timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tre_$n -o b -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-others > gpurun_out/tre_$n.log 2>&1 || exit 1
done
echo done
