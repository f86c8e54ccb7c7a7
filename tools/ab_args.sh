#!/bin/bash
# usage: ab_args.sh "ARGS A" "ARGS B" ... -- bench.py (headline config) once per argument set, two rounds on the same box ("" = default).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2; do
  for v in "$@"; do
    tag=$(echo "$v" | tr -c 'A-Za-z0-9' '_')
    log=gpurun_out/aba_${tag}_$i.log
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-others --steps 20 --warmup 5 $v > $log 2>&1
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
    print(f"[{v or 'default'}] {r['ms_per_step']:.3f} ms/step  " + "  ".join(f"{n} {d['launch_ms']:.3f}" for n, d in k.items()))
PY
  done
done
