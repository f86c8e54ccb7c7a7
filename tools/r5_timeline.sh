# kernel timeline of one step of `bench.py --config N` (default 3): every kernel >= 15 us with its stream, and the gaps on the main stream
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
cfg=${1:-3}; shift
out=gpurun_out/r5tl; mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $out/trace -o b -- python3 bench.py --config $cfg --steps 4 --warmup 2 --no-cpu-baseline --no-others "$@" > $out/trace.log 2>&1 || { tail -5 $out/trace.log; exit 1; }
python tools/trace_timeline.py $out/trace 0 > $out/timeline_cfg$cfg.txt
python - $out/timeline_cfg$cfg.txt <<'PY'
import sys
rows=[l.split() for l in open(sys.argv[1]) if l[0]==' ']
ev=[(float(r[0]),float(r[2]),r[5],r[-1]) for r in rows]
q1=[e for e in ev if e[2]=="q1"]
busy=sum(e[1]-e[0] for e in q1)
print("kernels on q1:",len(q1),"busy ms %.3f"%busy, "span %.3f"%(q1[-1][1]-q1[0][0]))
gaps=[(q1[i+1][0]-q1[i][1], q1[i][3], q1[i+1][3]) for i in range(len(q1)-1)]
print("sum of gaps %.3f ms"%sum(g[0] for g in gaps if g[0]>0))
for g in sorted(gaps,reverse=True)[:15]: print("  %.1f us  after %s before %s"%(g[0]*1e3,g[1][:40],g[2][:40]))
PY
find $out/trace -name '*kernel_trace.csv' -delete
