"""Summarise a HYPHY_HIP_TIMELINE dump: where do the waves of the pruning kernel spend their cycles?"""
import sys, collections
import numpy as np
rows = [list(map(int, l.split())) for l in open(sys.argv[1]) if not l.startswith("#")]
a = np.array(rows)
LEAF, LAST = 4, 2
for wg in sorted(set(a[:, 0]))[:3]:
    for w in range(4):
        r = a[(a[:, 0] == wg) & (a[:, 1] == w)]
        if len(r) == 0: continue
        t0 = r[:, 4]; tc = r[:, 5]; tb = r[:, 6]; tf = r[:, 7]
        total = (np.where(tf > 0, tf, tc)[-1] - t0[0])
        comp = tc - t0
        leaf = (r[:, 3] & LEAF) > 0
        last = (r[:, 3] & LAST) > 0
        gap = t0[1:] - np.where(last[:-1], tf[:-1], tc[:-1])
        print(f"wg {wg} wave {w}: total {total} cyc | internal entries {(~leaf).sum()} mean {comp[~leaf].mean():.0f} "
              f"(min {comp[~leaf].min()} max {comp[~leaf].max()}) | leaf entries {leaf.sum()} mean {comp[leaf].mean():.0f} | "
              f"finalise: wait-barrier {np.mean(tb[last]-tc[last]):.0f} rest {np.mean(tf[last]-tb[last]):.0f} x{last.sum()} | "
              f"between entries mean {gap.mean():.0f}")
r = a[(a[:, 0] == 0) & (a[:, 1] == 0)]
print("first 40 entries of wg0 wave0: flags, compute cycles, barrier wait, finalise rest")
for x in r[:40]:
    print(x[2], x[3], x[5] - x[4], (x[6] - x[5]) if x[6] else 0, (x[7] - x[6]) if x[7] else 0)
