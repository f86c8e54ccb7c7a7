#!/bin/bash
# Run GPU steps one after another inside ONE gpurun call; every step has its own timeout, and a step that
# is killed by its timeout (rc 124/137) ends the session (no further GPU step after a hang).
# usage: tools/gpu_steps.sh "<secs>|<logname>|<command>" ...
mkdir -p gpurun_out
cd /root/repo 2>/dev/null || true
export TMPDIR=/tmp
for spec in "$@"; do
  secs="${spec%%|*}"; rest="${spec#*|}"; name="${rest%%|*}"; cmd="${rest#*|}"
  echo "=== [$name] $cmd" | tee -a gpurun_out/session.log
  timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "=== [$name] rc=$rc" | tee -a gpurun_out/session.log
  tail -4 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step timed out: stopping the session" | tee -a gpurun_out/session.log; exit $rc; fi
done
exit 0
