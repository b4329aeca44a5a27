"""Host-side decisions of hyphy_hip_create that need no device (C-ABI: hyphy_hip_plan_reroot, hyphy_hip_plan_pattern_order):
the node steady-state passes may be re-rooted at, and the device-side pattern order."""
import collections

import numpy as np

from hyphy_amd import hip, tree


def _eccentricities(flat):
    fp, L = np.asarray(flat.flat_parents), flat.L
    N = len(fp)
    adj = [[] for _ in range(N)]
    for c, p in enumerate(fp):
        if p >= 0:
            adj[c].append(L + p)
            adj[L + p].append(c)
    ecc = {}
    for r in range(L, N):
        dist = {r: 0}
        q = collections.deque([r])
        while q:
            u = q.popleft()
            for v in adj[u]:
                if v not in dist:
                    dist[v] = dist[u] + 1
                    q.append(v)
        ecc[r - L] = max(dist[leaf] for leaf in range(L))
    return ecc


def test_reroot_plan_finds_the_centre_of_the_tree():
    rng = np.random.default_rng(11)
    n_rerooted = 0
    for trial in range(120):
        n = int(rng.integers(5, 90))
        root = (tree.caterpillar_tree(max(5, n)) if trial % 5 == 0
                else tree.random_tree(n, rng, trifurcating_root=bool(rng.integers(0, 2))))
        flat = tree.flatten(root)
        ecc = _eccentricities(flat)
        best = min(ecc.values())
        I = flat.I
        centres = sorted(k for k, e in ecc.items() if e == best)
        assert len(centres) in (1, 2)            # the centre of a tree: one node or two adjacent ones
        cands = [hip.plan_reroot(flat.flat_parents, flat.L, c) for c in (0, 1)]
        cands = [c for c in cands if len(c)]
        offered = sorted(int(c[-1]) for c in cands)
        def depth(k):                                # edges between internal node k and the given root
            d = 0
            while k != I - 1:
                k, d = int(flat.flat_parents[flat.L + k]), d + 1
            return d
        # every centre node other than the given root is offered (if it is within the 32 twin images), nothing else
        # (trees with fewer than four internal nodes are left alone)
        assert offered == [k for k in centres if k != I - 1 and depth(k) <= 32 and I >= 4], (trial, offered, centres)
        for path in cands:
            assert int(path[0]) == I - 1 and len(path) - 1 <= 32
            for a, b in zip(path[:-1], path[1:]):  # each step goes from a node to one of its children
                assert int(flat.flat_parents[flat.L + int(b)]) == int(a)
        n_rerooted += bool(cands) and ecc[I - 1] > best
    assert n_rerooted > 20                         # most random trees are not hung from their centre


def test_pattern_order_is_a_permutation_grouped_by_majority_state():
    rng = np.random.default_rng(3)
    L, S, D = 12, 500, 61
    base = rng.integers(0, D, size=S)
    codes = np.where(rng.random((L, S)) < 0.25, rng.integers(0, D, size=(L, S)), base[None, :]).astype(np.int64)
    codes[3, 17] = -2                              # an ambiguity code sorts like any other value
    order = hip.plan_pattern_order(D, codes)
    assert sorted(order.tolist()) == list(range(S))
    major = np.array([np.bincount(codes[:, s][codes[:, s] >= 0], minlength=D).argmax() for s in range(S)])
    assert np.all(np.diff(major[order]) >= 0)      # primary key: the pattern's most frequent state
    # inside a group: lexicographic by leaf
    for m in np.unique(major):
        cols = [tuple(codes[:, s]) for s in order if major[s] == m]
        assert cols == sorted(cols)
    # short alignments keep the caller's order
    assert hip.plan_pattern_order(D, codes[:, :20]).tolist() == list(range(20))
