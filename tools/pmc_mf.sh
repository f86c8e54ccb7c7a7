#!/bin/bash
# HBM traffic of the multi-row data-gradient kernels (csrc/dconv_m.hip) against the kernels they replace: FETCH_SIZE / WRITE_SIZE passes.
set -u
export TMPDIR=/tmp
O=gpurun_out/mf_pmc
mkdir -p $O
P="rocprofv3 --kernel-trace --output-format csv"
for v in on off; do
  if [ $v = off ]; then export DD_DCONV_MFWD_OFF=1; fi
  timeout -k 10 300 $P --pmc FETCH_SIZE -d $O/fetch_$v -o p -- python3 tools/ab_mfwd.py --iters 3 --no-check > $O/fetch_$v.log 2>&1; echo fetch_$v rc=$?
  timeout -k 10 300 $P --pmc WRITE_SIZE -d $O/write_$v -o p -- python3 tools/ab_mfwd.py --iters 3 --no-check > $O/write_$v.log 2>&1; echo write_$v rc=$?
done
# algorithmic bytes: input + output + mask, bs 32: up2 dgrad g 340^2 x 32ch, dx and mask 298^2 x 64ch; up3 dgrad g 382^2 x 16, dx and mask 340^2 x 32
python3 tools/pmc_traffic_any.py $O/fetch_on $O/write_on $O/traffic_on.json '_kernel<7, 7, 2, 4, 3=1928524800:up2_dgrad' '_kernel<7, 7, 1, 4, 3=1245839360:up3_dgrad' > $O/traffic_on.log 2>&1
python3 tools/pmc_traffic_any.py $O/fetch_off $O/write_off $O/traffic_off.json 'dconv_fwd_kernel<7, 7, 2>=1928524800:up2_dgrad' 'dconv_fwd_kernel<7, 7, 1>=1245839360:up3_dgrad' > $O/traffic_off.log 2>&1
cat $O/traffic_on.log $O/traffic_off.log
