#!/bin/bash
# Run GPU steps one after another on the GPU box; a step that was killed or timed out (rc 124 / 137 / 143) ends the sequence
# (no further GPU step after a hang), an ordinary failure (assertion, rc 1) does not.
# usage: tools/gpu_seq.sh "name1::cmd1" "name2::cmd2" ...   (logs: gpurun_out/<name>.log)
mkdir -p gpurun_out
for step in "$@"; do
  name="${step%%::*}"; cmd="${step#*::}"
  echo "=== $name: $cmd"
  bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "rc=$rc" >> "gpurun_out/$name.log"
  echo "=== $name rc=$rc"; tail -3 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 143 ]; then echo "step $name was killed: stopping"; exit $rc; fi
done
exit 0
