#!/bin/bash
# usage: ab_cfg.sh "[ENV=.. |] ARGS A" ...  -- bench.py per setting (any config), two rounds on one box; prints ms/step.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2; do
  for v in "$@"; do
    tag=$(echo "$v" | tr -c 'A-Za-z0-9' '_')
    log=gpurun_out/abc_${tag}_$i.log
    e=""; a="$v"
    case "$v" in *"|"*) e="${v%%|*}"; a="${v#*|}";; esac
    env $e timeout -k 10 300 python bench.py --no-cpu-baseline --no-others $a > $log 2>&1
    rc=$?
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$v] killed (rc $rc): stopping"; exit $rc; fi
    python3 - "$v" $log <<'PY'
import json, sys
v, log = sys.argv[1], sys.argv[2]
line = [l for l in open(log) if l.startswith("{")]
print(f"[{v}] " + (f"{json.loads(line[-1])['ms_per_step']:.3f} ms/step" if line else f"no result (see {log})"))
PY
  done
done
