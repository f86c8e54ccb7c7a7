#!/bin/bash
# One gpurun call: refresh every profile that DESIGN.md / bench.py cite, at HEAD.  usage: tools/profile_round.sh r02
# (separate rocprofv3 passes: kernel trace + stats; SQ counters; FETCH_SIZE; WRITE_SIZE -- never --pmc together with other traces
# than --kernel-trace; the program after `--` is python3 itself)
set -u
R=${1:-r02}
export TMPDIR=/tmp
O=gpurun_out/prof_$R
mkdir -p $O profiles
P="rocprofv3 --kernel-trace --output-format csv"
run() { name=$1; shift; timeout -k 10 400 "$@" > $O/$name.log 2>&1; echo "[$name] rc=$?"; }
run bench_stats rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -o b -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-others
run bbox_stats rocprofv3 --kernel-trace --stats --output-format csv -d $O/bbox_stats -o b -- python3 tools/bench_models.py --which bbox --steps 4 --warmup 2
run mfma $P --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/mfma -o p -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-others
run up_sq $P --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $O/up_sq -o p -- python3 tools/bench_gconv.py --batch 32 --only up
run fetch $P --pmc FETCH_SIZE -d $O/fetch -o p -- python3 tools/bench_one.py wino2_fwd,wino2_dgrad_w1
run write $P --pmc WRITE_SIZE -d $O/write -o p -- python3 tools/bench_one.py wino2_fwd,wino2_dgrad_w1
cp $O/bench_stats/b_kernel_stats.csv profiles/${R}_bench_kernel_stats.csv 2>/dev/null
cp $O/bbox_stats/b_kernel_stats.csv profiles/${R}_bbox_bs32_kernel_stats.csv 2>/dev/null
python3 tools/pmc_mfma.py $O/mfma gpurun_out/${R}_mfma_util.json > $O/mfma_sum.log 2>&1
python3 tools/pmc_sq.py gpurun_out/${R}_upconv_sq_counters.json $O/up_sq > $O/up_sq_sum.log 2>&1
python3 tools/pmc_traffic.py $O/fetch $O/write gpurun_out/${R}_c2_fwd_traffic.json > $O/traffic_fwd.log 2>&1
python3 tools/pmc_traffic.py $O/fetch $O/write gpurun_out/${R}_c2_dgrad_w1_traffic.json dgrad_w1 > $O/traffic_dg.log 2>&1
cp profiles/${R}_bench_kernel_stats.csv profiles/${R}_bbox_bs32_kernel_stats.csv gpurun_out/ 2>/dev/null
for f in $O/*_sum.log $O/traffic_*.log; do tail -n 2 $f; done
