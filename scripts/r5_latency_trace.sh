#!/bin/bash
# kernel trace of the single-pair loop (configs[1]): per-kernel totals and the gaps between launches
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
python scripts/bench_latency.py
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/q_lat -o q -- python scripts/bench_latency.py > gpurun_out/q_lat.log 2>&1
tail -1 gpurun_out/q_lat.log
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/q_lat/q_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the last 20 calls: split by k_quantize / first kernel of a call? use the k_select as end marker
ends=[i for i,r in enumerate(rows) if "k_select" in r["Kernel_Name"]]
calls=[]
for a,b in zip(ends[-11:-1], ends[-10:]):
    seg=rows[a+1:b+1]
    t0=int(seg[0]["Start_Timestamp"]); t1=int(seg[-1]["End_Timestamp"])
    busy=sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in seg)
    calls.append((len(seg),(t1-t0)/1e3,busy/1e3))
print("launches, span us, busy us:", calls[-3:])
import collections, re
seg=rows[ends[-2]+1:ends[-1]+1]
agg=collections.OrderedDict()
for r in seg:
    n=re.sub(r"\(anonymous namespace\)::","",r["Kernel_Name"]); n=re.sub(r"\(.*","",n).replace("void ","")[:50]
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    c,t=agg.get(n,(0,0.0)); agg[n]=(c+1,t+d)
for n,(c,t) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:28]: print(f"{n:52s} {c:3d} {t:8.1f} us")
# gaps
gaps=[(int(seg[i+1]["Start_Timestamp"])-int(seg[i]["End_Timestamp"]))/1e3 for i in range(len(seg)-1)]
big=sorted([(g,re.sub(r"\(.*","",seg[i]["Kernel_Name"])[-40:],re.sub(r"\(.*","",seg[i+1]["Kernel_Name"])[-40:]) for i,g in enumerate(gaps)], reverse=True)[:8]
print("sum of gaps us", round(sum(g for g in gaps if g>0),1)); 
for g in big: print(g)
PY
