cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r5m; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace -o b -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-others > $out/trace.log 2>&1
python tools/trace_timeline.py $out/trace 0 > $out/timeline_full.txt
python - <<'PY'
import re
rows=[l.split() for l in open("gpurun_out/r5m/timeline_full.txt") if l[0]==' ']
ev=[(float(r[0]),float(r[2]),r[5],r[-1]) for r in rows]
q1=[e for e in ev if e[2]=="q1"]
busy=sum(e[1]-e[0] for e in q1)
print("kernels on q1:",len(q1),"busy ms %.3f"%busy, "span %.3f"%(q1[-1][1]-q1[0][0]))
gaps=[(q1[i+1][0]-q1[i][1], q1[i][3], q1[i+1][3]) for i in range(len(q1)-1)]
print("sum of gaps %.3f ms"%sum(g[0] for g in gaps if g[0]>0))
for g in sorted(gaps,reverse=True)[:12]: print("  %.1f us  after %s before %s"%(g[0]*1e3,g[1][:30],g[2][:30]))
PY
find $out/trace -name '*kernel_trace.csv' -delete
