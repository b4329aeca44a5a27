#!/usr/bin/env python3
"""Summary of a row-split class-table launch (HYPHY_HIP_REP_TIMELINE=file with the team walk on: one record per workgroup):
when the workgroups started and ended, how many shared a CU, where their cycles went, by walk length."""
import sys
import numpy as np
a = np.loadtxt(sys.argv[1], comments="#")
t0 = a[:, 1].min()
start, end = (a[:, 1] - t0) / 100.0, (a[:, 2] - t0) / 100.0   # us
nodes = a[:, 3].astype(int)
hw = a[:, 4].astype(np.int64)
xcc, cu, se = (hw >> 32) & 7, (hw >> 8) & 15, (hw >> 13) & 7
cu_id = xcc * 1000 + se * 16 + cu
print(f"workgroups {len(a)}, launch span {end.max():.1f} us; start: median {np.median(start):.2f} p90 {np.percentile(start, 90):.2f} max {start.max():.2f} us; "
      f"started later than 2 us: {int((start > 2.0).sum())}")
print(f"distinct (xcc, se, cu) seen {len(np.unique(cu_id))}; workgroups per CU: median {np.median(np.unique(cu_id, return_counts=True)[1]):.0f} max {np.unique(cu_id, return_counts=True)[1].max()}")
names = ["index maps", "gathers", "exchange", "product", "publish"]
for n in sorted(set(nodes)):
    m = nodes == n
    dur = end[m] - start[m]
    print(f"walks of {n} nodes: {int(m.sum())}; duration median {np.median(dur):.2f} max {dur.max():.2f} us; end median {np.median(end[m]):.2f} max {end[m].max():.2f} us")
    tot = a[m, 5:10].sum()
    for k, nm in enumerate(names):
        c = a[m, 5 + k]
        print(f"    {nm:10s} {100 * c.sum() / tot:5.1f} %   per walk median {np.median(c):8.0f} cycles" + (f"   per node {np.median(c) / max(1, n):8.0f}" if k in (1, 2, 3) else ""))
