#!/usr/bin/env python3
"""Summary of a class-table kernel timeline (HYPHY_HIP_REP_TIMELINE=file): where the waves' cycles go."""
import sys
import numpy as np
a = np.loadtxt(sys.argv[1], comments="#")
t0 = a[:, 1].min()
start, end = (a[:, 1] - t0) / 100.0, (a[:, 2] - t0) / 100.0   # us
print(f"waves {len(a)}, launch span {end.max():.1f} us; wave start median {np.median(start):.1f} max {start.max():.1f}; wave end median {np.median(end):.1f}")
print(f"items {int(a[:, 3].sum())} (per wave median {np.median(a[:, 3]):.0f}, max {a[:, 3].max():.0f}); failed polls {int(a[:, 4].sum())}")
names = ["tickets", "waiting", "gathers", "product", "publish"]
tot = a[:, 5:10].sum()
for k, n in enumerate(names):
    c = a[:, 5 + k]
    print(f"  {n:8s} {100 * c.sum() / tot:5.1f} %   per item {c.sum() / max(1, a[:, 3].sum()):9.0f} cycles   per wave median {np.median(c):9.0f}")
