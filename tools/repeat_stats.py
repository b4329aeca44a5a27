#!/usr/bin/env python3
"""Subtree-repeat statistics of a bench workload: per internal node the number of distinct leaf sub-patterns below it
(U_n), and what a class-compressed lower phase + per-pattern trunk would execute for a threshold theta.

The reference skips a node at a site whose subtree leaves equal the previous site's (tcc masks, src/core/tree.cpp:2801-2858);
the device form is per-node class tables.  Host-only; no device needed.

    python tools/repeat_stats.py mg94_64x10k [theta ...]
"""
import sys
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hyphy_amd import data  # noqa: E402
import bench  # noqa: E402


def classes(leaf_codes, parents, L, I):
    """class id per (internal node, pattern) bottom-up; returns (cls [I][S], U [I])."""
    S = leaf_codes.shape[1]
    children = [[] for _ in range(I)]
    for n in range(L + I - 1):
        children[parents[n]].append(n)
    cls = np.zeros((I, S), dtype=np.int64)
    U = np.zeros(I, dtype=np.int64)
    for i in range(I):
        key = np.zeros(S, dtype=np.int64)
        rows = []
        for c in children[i]:
            rows.append(leaf_codes[c] if c < L else cls[c - L])
        arr = np.stack(rows, axis=1)
        _, inv = np.unique(arr, axis=0, return_inverse=True)
        cls[i] = inv.reshape(-1)
        U[i] = inv.max() + 1
    return cls, U, children


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "mg94_64x10k"
    thetas = [float(x) for x in sys.argv[2:]] or [0.0, 0.3, 0.5, 0.7, 0.9, 1.0]
    wl = bench.WORKLOADS[name]
    syn = data.evolve(wl["taxa"], wl["sites"], wl["unit"], seed=wl["seed"], p_change=wl.get("p_change", 0.04))
    D = 61 if wl["unit"] == 3 else 4
    pd = data.from_states(syn.states, D, compress_patterns=(D > 4))
    flat = syn.flat
    L, I = flat.L, flat.I
    S = pd.S
    parents = np.asarray(flat.flat_parents)
    cls, U, children = classes(np.asarray(pd.leaf_codes), parents, L, I)
    nleaves = np.zeros(I, dtype=int)
    height = np.zeros(I, dtype=int)
    for i in range(I):
        for c in children[i]:
            nleaves[i] += 1 if c < L else nleaves[c - L]
            height[i] = max(height[i], 1 if c < L else height[c - L] + 1)
    print(f"{name}: S = {S}, L = {L}, I = {I}")
    print("node leaves height U  U/S")
    for i in range(I):
        print(f"{i:4d} {nleaves[i]:4d} {height[i]:3d} {U[i]:7d} {U[i] / S:.3f}")
    total = (I - 1) * S
    for th in thetas:
        comp = [i for i in range(I - 1) if U[i] <= th * S and U[i] <= 32767]
        cset = set(comp)
        roots = [i for i in comp if parents[L + i] not in cset]
        lower = int(sum(U[i] for i in comp))
        trunk_nodes = [i for i in range(I) if i not in cset]
        trunk = (len(trunk_nodes) - 1) * S
        depth = max([height[i] for i in roots], default=0)
        tabs = int(sum(U[i] for i in comp)) * 512
        print(f"theta {th:.2f}: compressed nodes {len(comp)} (roots {len(roots)}, depth {depth}), lower edges {lower}, trunk nodes {len(trunk_nodes)} "
              f"trunk edges {trunk}, executed/total = {(lower + trunk) / total:.3f}, tables {tabs / 1e6:.1f} MB, "
              f"gen-leaf gathers/site {len(roots)}")


if __name__ == "__main__":
    main()
