#!/usr/bin/env python3
"""Per-CALL kernel durations of a rocprofv3 --kernel-trace run (the --stats summary only gives averages, which hide a kernel
that takes 7 us on 99 calls and 260 us on the hundredth):

    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 bench.py --workload c1 --steps 300 ...
    python tools/trace_calls.py <dir> [kernel substring, default k_step_small] [slow threshold in us, default 100]

prints p50 / p99 / max and the five longest calls of every kernel with more than 50 calls, and the kernels that follow the
slow k_step_small calls (the small-system candidate rebuild, profiles/r05_experiments.md section 8)."""
import csv,glob,collections,sys
d0=sys.argv[1]
rows=[]
for f in glob.glob(d0+"/**/*kernel_trace.csv", recursive=True):
    rows+=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
for r in rows:
    r["Kernel_Name"]=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","")
d=collections.defaultdict(list)
for r in rows:
    d[r["Kernel_Name"].split("(")[0][-40:]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in d.items():
    if len(v)>50:
        v2=sorted(v); print(k, len(v), "p50 %.1f p99 %.1f max %.1f top5 %s"%(v2[len(v2)//2], v2[int(len(v2)*0.99)], v2[-1], [round(x,1) for x in v2[-5:]]))
pat=sys.argv[2] if len(sys.argv)>2 else "k_step_small"
thr=float(sys.argv[3])*1e3 if len(sys.argv)>3 else 100000
idx=[i for i,r in enumerate(rows) if pat in r["Kernel_Name"] and (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))>thr]
print("slow %s calls:"%pat, len(idx))
for i in idx[1:3]:
    t0=int(rows[i]["Start_Timestamp"])
    for r in rows[i:i+16]:
        print("   %8.1f us  +%6.1f  %s"%((int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3,r["Kernel_Name"].split("(")[0][-44:]))
    print()
