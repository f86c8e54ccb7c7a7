#!/bin/bash
# usage: ab_multi.sh "VAR=a" "VAR=b VAR2=c" ...  -- bench.py (headline config only) once per setting, three rounds on the same box;
# "" = default.  Prints ms/step and the in-step launch times of the two c2 kernels.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2; do
  for v in "$@"; do
    tag=$(echo "$v" | tr -c 'A-Za-z0-9' '_')
    log=gpurun_out/abm_${tag}_$i.log
    env $v timeout -k 10 200 python bench.py --no-cpu-baseline --no-others --steps 20 --warmup 5 > $log 2>&1
    rc=$?
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$v] killed (rc $rc): stopping"; exit $rc; fi
    python3 - "$v" $log <<'PY'
import json, sys
v, log = sys.argv[1], sys.argv[2]
line = [l for l in open(log) if l.startswith("{")]
if not line:
    print(f"[{v}] no result (see {log})")
else:
    r = json.loads(line[-1])
    k = r["roofline"]["kernels"]
    print(f"[{v or 'default'}] {r['ms_per_step']:.3f} ms/step  c2_fwd {k['c2_forward']['launch_ms']:.3f}  c2_dgrad_w1 {k.get('c2_dgrad_w1', {}).get('launch_ms', float('nan')):.3f}  loss {r['config']['final_loss']}")
PY
  done
done
