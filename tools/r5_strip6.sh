#!/bin/bash
set -o pipefail
out=gpurun_out/r5f; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_round5.py -x -q -k "strip6 or spatial_mapping" > $out/pytest.log 2>&1 || { tail -40 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "spatial or bbox or joint or heads or merging" > $out/pytest2.log 2>&1 || { tail -40 $out/pytest2.log; exit 1; }
tail -2 $out/pytest2.log
for s in 0 1 0 1; do
  DD_STRIP6=$s timeout -k 10 300 python bench.py --config 3 --no-others --no-cpu-baseline --steps 10 --warmup 3 > $out/s.json 2> $out/s.err || { tail -20 $out/s.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$out/s.json").read().strip().splitlines()[-1])
print("config 3 DD_STRIP6=$s", d["ms_per_step"], "ms", d["config"]["final_loss"])
PY
done
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o b -- python3 bench.py --config 3 --steps 4 --warmup 2 --no-cpu-baseline --no-others > $out/trace.log 2>&1 || { tail -20 $out/trace.log; exit 1; }
grep -i 'strip6\|view_to\|gconv_fwd_kernel<1\|gconv_wgrad_kernel<1' $out/trace/b_kernel_stats.csv | cut -c1-160
find $out/trace -name '*kernel_trace.csv' -delete
