#!/bin/bash
set -o pipefail
out=gpurun_out/r5b
mkdir -p $out
for mode in "DD_ADAM_LATE=0 on" "DD_ADAM_LATE=1 on" "DD_ADAM_LATE=0 off" "DD_ADAM_LATE=1 off"; do
  set -- $mode
  for rep in 1 2; do
    env $1 timeout -k 10 300 python bench.py --no-others --no-cpu-baseline --steps 20 --warmup 5 --fuse-linear-wgrad $2 > $out/s.json 2> $out/s.err || { tail -20 $out/s.err; exit 1; }
    python - <<PY
import json
d=json.loads(open("$out/s.json").read().strip().splitlines()[-1])
print("$1 fuse $2 rep $rep", d["ms_per_step"], "ms", d["config"]["final_loss"], d["roofline"].get("launch_ms"))
PY
  done
done
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export DD_ADAM_LATE=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_late -o b -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-others > $out/trace_late.log 2>&1 || { tail -20 $out/trace_late.log; exit 1; }
python tools/trace_timeline.py $out/trace_late 30 > $out/timeline_late.txt || exit 1
grep -v 'conv_strip\|pool4\|linear_\|mlp_tail\|bce_\|stitch\|pack' $out/timeline_late.txt
find $out/trace_late -name '*kernel_trace.csv' -delete
