#!/bin/bash
# kernel trace of the headline step with two builds of the library on one box: tools/trace_two.sh OLD.so  (DD_HOTPATH_LIB override)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for t in head old; do
  if [ $t = old ]; then export DD_HOTPATH_LIB=$1; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tr_$t -o b -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-others --adam-overlap on > gpurun_out/tr_$t.log 2>&1 || exit 1
done
echo done
