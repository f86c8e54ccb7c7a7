#!/bin/bash
# Round 5: the rank-B optimizer pass -- kernel tests, variants alone, and the headline step with / without it.
set -o pipefail
out=gpurun_out/r5a
mkdir -p $out
rm -f $out/alone.log
python -m pytest tests/test_gpu_round5.py -x -q > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
for v in 0 5 2; do
  echo "== variant $v" >> $out/alone.log
  DD_RANKB_VARIANT=$v timeout -k 10 300 python tools/bench_rankb.py >> $out/alone.log 2>&1 || exit 1
done
echo "== variant 0, 2 blocks per CU" >> $out/alone.log
timeout -k 10 300 python tools/bench_rankb.py >> $out/alone.log 2>&1 || exit 1
grep -v amdgpu.ids $out/alone.log
for v in off 0 5 2; do
  if [ $v = off ]; then f=off; else f=on; fi
  for rep in 1 2; do
    DD_RANKB_VARIANT=$v timeout -k 10 300 python bench.py --no-others --no-cpu-baseline --steps 20 --warmup 5 --fuse-linear-wgrad $f > $out/step_$v.$rep.json 2> $out/step_$v.$rep.err || { tail -20 $out/step_$v.$rep.err; exit 1; }
    python - <<PY
import json
d=json.loads(open("$out/step_$v.$rep.json").read().strip().splitlines()[-1])
print("variant $v rep $rep", d["ms_per_step"], "ms", d["config"]["final_loss"], d["roofline"].get("launch_ms"))
PY
  done
done
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for f in on off; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$f -o b -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-others --fuse-linear-wgrad $f > $out/trace_$f.log 2>&1 || { tail -20 $out/trace_$f.log; exit 1; }
  python tools/trace_timeline.py $out/trace_$f 30 > $out/timeline_$f.txt || exit 1
  cat $out/timeline_$f.txt
  find $out/trace_$f -name '*kernel_trace.csv' -delete
done
