#!/bin/bash
# One gpurun call: refresh every profile that DESIGN.md / bench.py cite, at HEAD.  usage: tools/profile_round.sh r03
# (separate rocprofv3 passes: kernel trace + stats; SQ counters; FETCH_SIZE; WRITE_SIZE -- never --pmc together with other traces
# than --kernel-trace; the program after `--` is python3 itself)
set -u
R=${1:-r05}
export TMPDIR=/tmp
O=gpurun_out/prof_$R
mkdir -p $O profiles
P="rocprofv3 --kernel-trace --output-format csv"
run() { name=$1; shift; timeout -k 10 400 "$@" > $O/$name.log 2>&1; rc=$?; echo "[$name] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping"; exit $rc; fi; }
run bench_stats rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -o b -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-others
run bbox_stats rocprofv3 --kernel-trace --stats --output-format csv -d $O/bbox_stats -o b -- python3 bench.py --config 3 --steps 4 --warmup 2
run joint_stats rocprofv3 --kernel-trace --stats --output-format csv -d $O/joint_stats -o b -- python3 bench.py --config 4 --steps 4 --warmup 2
run bf16_stats rocprofv3 --kernel-trace --stats --output-format csv -d $O/bf16_stats -o b -- python3 bench.py --config 5 --steps 4 --warmup 2
run ae_stats rocprofv3 --kernel-trace --stats --output-format csv -d $O/ae_stats -o b -- python3 tools/profile_ae.py 32
run mfma $P --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/mfma -o p -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-others
run up_sq $P --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $O/up_sq -o p -- python3 tools/bench_gconv.py --batch 32 --only up
run up_wait $P --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $O/up_wait -o p -- python3 tools/bench_gconv.py --batch 32 --only up
run fetch $P --pmc FETCH_SIZE -d $O/fetch -o p -- python3 tools/bench_one.py wino2_fwd,wino2_dgrad_w1,wino2_wgrad
run write $P --pmc WRITE_SIZE -d $O/write -o p -- python3 tools/bench_one.py wino2_fwd,wino2_dgrad_w1,wino2_wgrad
run up_fetch $P --pmc FETCH_SIZE -d $O/up_fetch -o p -- python3 tools/bench_gconv.py --batch 32 --only up
run up_write $P --pmc WRITE_SIZE -d $O/up_write -o p -- python3 tools/bench_gconv.py --batch 32 --only up
for n in bench bbox joint bf16 ae; do cp $O/${n}_stats/b_kernel_stats.csv profiles/${R}_${n}_kernel_stats.csv 2>/dev/null; done
mv profiles/${R}_ae_kernel_stats.csv profiles/${R}_ae_bs32_kernel_stats.csv 2>/dev/null
mv profiles/${R}_bbox_kernel_stats.csv profiles/${R}_bbox_bs32_kernel_stats.csv 2>/dev/null
python3 tools/pmc_mfma.py $O/mfma gpurun_out/${R}_mfma_util.json > $O/mfma_sum.log 2>&1
python3 tools/pmc_sq.py gpurun_out/${R}_upconv_sq_counters.json $O/up_sq $O/up_wait > $O/up_sq_sum.log 2>&1
python3 tools/pmc_traffic.py $O/fetch $O/write gpurun_out/${R}_c2_fwd_traffic.json > $O/traffic_fwd.log 2>&1
python3 tools/pmc_traffic.py $O/fetch $O/write gpurun_out/${R}_c2_dgrad_w1_traffic.json dgrad_w1 > $O/traffic_dg.log 2>&1
# the dilated up-convs at bs 32: x [32,256,256,96] = 805,306,368 B, y / g [32,298,298,64] = 727,449,600 B (up_conv_1);
# x [32,298,298,64], y / g [32,340,340,32] = 473,497,600 B (up_conv_2).  Forward: read x, write y.  Data gradient: read g and the
# ReLU source x, write dx.  Weight gradient: read x and g.
python3 tools/pmc_traffic_any.py $O/up_fetch $O/up_write gpurun_out/${R}_upconv_traffic.json \
  'dconv_tfwd_kernel<7, 7, 7, 0=1532755968:up_conv_1 forward' 'dconv_gfwd_kernel<7, 7, 3=2338062336:up_conv_1 data gradient' \
  'dconv_wgrad_kernel<7, 7, 96, 64=1532755968:up_conv_1 weight gradient' \
  'dconv_tfwd_kernel<7, 7, 9, 1=1200947200:up_conv_2 forward' 'dconv_mwin_kernel<7, 7, 2=1928396800:up_conv_2 data gradient' \
  'dconv_wgrad_kernel<7, 7, 64, 32=1200947200:up_conv_2 weight gradient' > $O/traffic_up.log 2>&1
cp profiles/${R}_*_kernel_stats.csv gpurun_out/ 2>/dev/null
for f in $O/*_sum.log $O/traffic_*.log; do tail -n 2 $f; done
# the split-product experiment (csrc/dconv_split.hip): the same config-3 step with DD_DCONV_SPLIT=1
export DD_DCONV_SPLIT=1
run bbox_split_stats rocprofv3 --kernel-trace --stats --output-format csv -d $O/bbox_split_stats -o b -- python3 bench.py --config 3 --steps 4 --warmup 2
unset DD_DCONV_SPLIT
cp $O/bbox_split_stats/b_kernel_stats.csv profiles/${R}_bbox_split_products_kernel_stats.csv 2>/dev/null
cp profiles/${R}_bbox_split_products_kernel_stats.csv gpurun_out/ 2>/dev/null
