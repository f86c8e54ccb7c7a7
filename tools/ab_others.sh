#!/bin/bash
# usage: ab_others.sh "ENV_A" "ENV_B" ... -- full bench.py (headline + others) per environment, one box; prints every step time.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2; do
  for v in "$@"; do
    tag=$(echo "$v" | tr -c 'A-Za-z0-9' '_')
    log=gpurun_out/abo_${tag}_$i.log
    env $v timeout -k 10 400 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $log 2>&1
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
    print(f"[{v}] headline {r['ms_per_step']:.3f}  " + "  ".join(f"{k.split('_')[0]}_{k.split('_')[-1]} {d['ms_per_step']:.3f}" for k, d in r.get("others", {}).items()))
PY
  done
done
