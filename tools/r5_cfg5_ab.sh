#!/bin/bash
set -o pipefail
out=gpurun_out/r5e; mkdir -p $out
run() {
  env $1 timeout -k 10 300 python bench.py --config 5 --no-others --no-cpu-baseline --steps 20 --warmup 5 $2 > $out/s.json 2> $out/s.err || { tail -20 $out/s.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$out/s.json").read().strip().splitlines()[-1])
k=d["roofline"]["kernels"]
print("$1 $2 :", d["ms_per_step"], "ms", " ".join(f"{n}={v['launch_ms']}" for n,v in k.items()))
PY
}
for rep in 1 2; do
run "X=0" "--adam-overlap on --adam-blocks-per-cu 1"
run "X=0" "--adam-overlap on --adam-blocks-per-cu 2"
run "X=0" "--adam-overlap on --adam-blocks-per-cu 4"
run "X=0" "--adam-overlap off --adam-blocks-per-cu 1"
run "X=0" "--adam-overlap off --adam-blocks-per-cu 2"
run "X=0" "--adam-overlap off --adam-blocks-per-cu 4"
done
