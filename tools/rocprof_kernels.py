#!/usr/bin/env python3
"""Per-kernel launch statistics out of a rocprofv3 results database (rocprofv3 --kernel-trace --stats -d DIR ...).
usage: tools/rocprof_kernels.py DIR [csv-out]"""
import glob
import sqlite3
import sys

dbs = glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True)
if not dbs:
    sys.exit("no *_results.db under " + sys.argv[1])
rows = []
for db in dbs:
    cur = sqlite3.connect(db).cursor()
    rows += cur.execute("select name, end - start, vgpr_count, lds_size, grid_x * grid_y * grid_z / (workgroup_x * workgroup_y * workgroup_z) from kernels").fetchall()
by = {}
for name, dur, vg, lds, wgs in rows:
    by.setdefault(name, []).append((dur, vg, lds, wgs))
lines = ["kernel,calls,avg_us,min_us,max_us,stddev_us,total_ms,vgprs,lds_bytes,workgroups_last"]
for name, v in sorted(by.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
    d = [x[0] / 1e3 for x in v]
    avg = sum(d) / len(d)
    sd = (sum((x - avg) ** 2 for x in d) / len(d)) ** 0.5
    short = name.replace("hyhip::(anonymous namespace)::", "").replace("void ", "")
    lines.append(f"\"{short[:110]}\",{len(d)},{avg:.2f},{min(d):.2f},{max(d):.2f},{sd:.2f},{sum(d) / 1e3:.3f},{v[-1][1]},{v[-1][2]},{v[-1][3]}")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
