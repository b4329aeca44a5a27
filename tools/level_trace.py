#!/usr/bin/env python3
"""Per evaluation of the last steady-state steps: every kernel of the step with its duration and the gap in front of it, from a
rocprofv3 --kernel-trace CSV (one line per evaluation; lower phases of several level launches show each level).
usage: tools/level_trace.py <dir with *_kernel_trace.csv>"""
import csv,glob,re,sys
rows=[]
for f in glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),(re.findall(r"(\w+_kernel)",r["Kernel_Name"]) or ["?"])[0]))
rows.sort()
# steady state: last 6 evaluations
idx=[i for i,r in enumerate(rows) if r[2]=="expm64_kernel"]
for s,e in zip(idx[-4:-1],idx[-3:]):
    prev=None
    out=[]
    for st,en,n in rows[s:e]:
        gap=(st-prev)/1e3 if prev else 0
        out.append(f"{n.replace('_kernel','')}:{(en-st)/1e3:.1f}(+{gap:.1f})")
        prev=en
    print(" ".join(out), " total", (rows[e-1][1]-rows[s][0])/1e3)
