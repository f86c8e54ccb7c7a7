#!/bin/bash
# usage: ab_env_args.sh "ENV=.. | ARGS" ...  -- bench.py (headline config) per setting, two rounds on one box.  Left of '|': environment, right: arguments.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2; do
  for v in "$@"; do
    e="${v%%|*}"; a="${v#*|}"
    tag=$(echo "$v" | tr -c 'A-Za-z0-9' '_')
    log=gpurun_out/abe_${tag}_$i.log
    env $e timeout -k 10 200 python bench.py --no-cpu-baseline --no-others --steps 20 --warmup 5 $a > $log 2>&1
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
    print(f"[{v}] {r['ms_per_step']:.3f} ms/step  " + "  ".join(f"{n} {d['launch_ms']:.3f}" for n, d in k.items()))
PY
  done
done
