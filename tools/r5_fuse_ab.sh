#!/bin/bash
set -o pipefail
out=gpurun_out/r5c
mkdir -p $out
python -m pytest tests/test_gpu_round5.py -x -q > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -1 $out/pytest.log
timeout -k 10 300 python tools/bench_rankb.py 2>&1 | grep -v amdgpu.ids | tee $out/alone.log || exit 1
for cfg in 2 5; do
  for f in off on off on; do
    timeout -k 10 300 python bench.py --config $cfg --no-others --no-cpu-baseline --steps 20 --warmup 5 --fuse-linear-wgrad $f > $out/s.json 2> $out/s.err || { tail -20 $out/s.err; exit 1; }
    python - <<PY
import json
d=json.loads(open("$out/s.json").read().strip().splitlines()[-1])
print("config $cfg fuse $f", d["ms_per_step"], "ms", d["config"]["final_loss"], d["roofline"].get("launch_ms"))
PY
  done
done
timeout -k 10 300 python tools/ab_fuse.py 2>&1 | grep -v amdgpu.ids | tee $out/ae.log || exit 1
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o b -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-others > $out/trace.log 2>&1 || { tail -20 $out/trace.log; exit 1; }
python tools/trace_timeline.py $out/trace 30 > $out/timeline.txt || exit 1
grep -v 'conv_strip\|pool4\|linear_\|mlp_tail\|bce_\|stitch\|pack' $out/timeline.txt
find $out/trace -name '*kernel_trace.csv' -delete
