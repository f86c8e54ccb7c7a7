#!/bin/bash
# ab_c3.sh "ENV=VAL ..." [REPS]: config-3 (CFG=4: config-4) step time with and without an environment setting, alternating, on one box
for rep in $(seq 1 ${2:-2}); do
  for e in "" "$1"; do
    ms=$(env $e python bench.py --config ${CFG:-3} --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --no-others 2>/dev/null | python -c "import sys, json; print(json.loads(sys.stdin.readline())['ms_per_step'])")
    echo "[${e:-default}] $ms ms"
  done
done
