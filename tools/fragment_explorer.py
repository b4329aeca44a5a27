"""Offline replica of build_schedule's level peeling (api.hip) to inspect how a tree is cut into
subtree fragments for a given max_frag: prints the fragment sizes per level and a crude cost model
(sum over levels of the largest fragment = critical path in nodes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hyphy_amd import data


def levels(flat_parents, L, max_frag):
    I = len(flat_parents) - L
    par = [int(flat_parents[L + i]) for i in range(I)]  # internal index of the parent, -1 for the root
    children = [[] for _ in range(I)]
    for n in range(I):
        if par[n] >= 0:
            children[par[n]].append(n)
    done = [False] * I
    out = []
    while True:
        size = [0] * I
        for n in range(I):
            if done[n]:
                continue
            size[n] = 1 + sum(size[c] for c in children[n])
        root = I - 1
        if size[root] <= max_frag:
            frags = [[n for n in range(I) if not done[n]]]
        else:
            frag_root = [-1] * I
            for n in range(I - 1, -1, -1):
                if done[n]:
                    continue
                p = par[n]
                if p >= 0 and not done[p] and frag_root[p] >= 0:
                    frag_root[n] = frag_root[p]
                elif size[n] <= max_frag:
                    frag_root[n] = n
            d = {}
            for n in range(I):
                if not done[n] and frag_root[n] >= 0:
                    d.setdefault(frag_root[n], []).append(n)
            frags = list(d.values())
        for f in frags:
            for n in f:
                done[n] = True
        out.append([len(f) for f in frags])
        if done[root]:
            break
    return out


if __name__ == "__main__":
    taxa, seed = int(sys.argv[1]), int(sys.argv[2])
    syn = data.evolve(taxa, 30, 3, seed=seed)
    fp = np.asarray(syn.flat.flat_parents)
    for mf in [int(x) for x in sys.argv[3:]]:
        lv = levels(fp, syn.flat.L, mf)
        print(mf, lv, "critical path (nodes):", sum(max(l) for l in lv))
