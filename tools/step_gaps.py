#!/usr/bin/env python3
"""Where one evaluation's wall time goes, from a rocprofv3 --kernel-trace CSV: per kernel of the step its duration and the gap
in front of it (end of the previous kernel -> start of this one); the gap in front of the step's first kernel is the host's
turnaround (result read, next parameters, launch).  usage: tools/step_gaps.py <dir with *_kernel_trace.csv> [first kernel substring]"""
import csv
import re
import glob
import sys
import numpy as np

d = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "expm"
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), (re.findall(r"(\w+_kernel|__amd_\w+)", r["Kernel_Name"]) or [r["Kernel_Name"][:40]])[0]))
rows.sort()
steps, cur = [], []
for st, en, name in rows:
    if first in name and cur:
        steps.append(cur)
        cur = []
    cur.append((st, en, name))
steps = steps[len(steps) // 2:]  # steady state: the second half
sig = {}
for s in steps:
    key = tuple(n for _, _, n in s)
    sig.setdefault(key, []).append(s)
key, group = max(sig.items(), key=lambda kv: len(kv[1]))
print(f"{len(group)} steps of the form: {' -> '.join(key)}")
tot = np.array([g[-1][1] - g[0][0] for g in group]) / 1e3
for i, name in enumerate(key):
    dur = np.array([g[i][1] - g[i][0] for g in group]) / 1e3
    if i:
        gap = np.array([g[i][0] - g[i - 1][1] for g in group]) / 1e3
        print(f"   gap {np.median(gap):6.2f} us")
    print(f"{name:28s} {np.median(dur):7.2f} us")
print(f"first start -> last end: {np.median(tot):.2f} us")
per, turn = [], []
for a, b in zip(steps[:-1], steps[1:]):
    if tuple(n for _, _, n in a) == key and tuple(n for _, _, n in b) == key:
        per.append((b[0][0] - a[0][0]) / 1e3)
        turn.append((b[0][0] - a[-1][1]) / 1e3)
if per:
    print(f"period {np.median(per):.2f} us; last kernel's end -> next step's first kernel: {np.median(turn):.2f} us")
