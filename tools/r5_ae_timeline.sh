# kernel timeline of the autoencoder step (bs 32) with the early rank-B pass leaving SPARE compute units free: one trace per value
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r5ae; mkdir -p $out
for s in ${SPARES:-0 8}; do
  export SPARE=$s
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace$s -o b -- python3 tools/profile_ae.py 32 > $out/trace$s.log 2>&1 || { tail -5 $out/trace$s.log; exit 1; }
  grep "AE bs" $out/trace$s.log
  python tools/trace_timeline.py $out/trace$s 15 > $out/timeline_spare$s.txt
  find $out/trace$s -name '*kernel_trace.csv' -delete
done
