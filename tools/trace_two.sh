#!/bin/bash
# kernel trace of the headline step with two builds of the library on one box: tools/trace_two.sh OLD.so ["ENV for the old build"]
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tr_head -o b -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-others > gpurun_out/tr_head.log 2>&1 || exit 1
env DD_HOTPATH_LIB=$1 $2 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tr_old -o b -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-others > gpurun_out/tr_old.log 2>&1 || exit 1
echo done
