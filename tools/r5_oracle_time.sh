#!/bin/bash
out=gpurun_out/r5d; mkdir -p $out
T="tests/test_gpu_round3.py::test_joint_training_step_three_way tests/test_gpu_round2.py::test_full_size_autoencoder_step_three_way"
echo "== default"; python -m pytest $T -x -q --durations=3 2>&1 | grep -E 'passed|failed|s call' 
echo "== MIOPEN_FIND_MODE=2"; MIOPEN_FIND_MODE=2 python -m pytest $T -x -q --durations=3 2>&1 | grep -E 'passed|failed|s call'
echo "== second run default (warm MIOpen cache)"; python -m pytest $T -x -q --durations=3 2>&1 | grep -E 'passed|failed|s call'
ls ~/.cache/miopen 2>/dev/null | head; du -sh ~/.cache/miopen ~/.config/miopen 2>/dev/null
