#!/bin/bash
# SQ counter breakdown of the c2 Winograd kernels (register-row form; DD_WINO2_RING=1 in the environment: the LDS-ring form).
# usage: tools/sq_wino.sh TAG     (rocprofv3 --pmc passes with --kernel-trace only; program directly after `--`)
set -u
export TMPDIR=/tmp
T=${1:-new}
O=gpurun_out/sq_$T
mkdir -p $O
P="rocprofv3 --kernel-trace --output-format csv"
timeout -k 10 300 $P --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/a -o p -- python3 tools/bench_one.py wino2_fwd,wino2_dgrad_w1,wino2_dgrad > $O/a.log 2>&1 || exit 1
timeout -k 10 300 $P --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $O/b -o p -- python3 tools/bench_one.py wino2_fwd,wino2_dgrad_w1,wino2_dgrad > $O/b.log 2>&1 || exit 1
python3 tools/pmc_sq.py gpurun_out/sq_$T.json $O/a $O/b > $O/sum.log 2>&1
tail -n 3 $O/sum.log
