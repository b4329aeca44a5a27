"""Host-side decisions of hyphy_hip_create that need no device (C-ABI: hyphy_hip_plan_reroot, hyphy_hip_plan_pattern_order,
hyphy_hip_plan_schedule): the node steady-state passes may be re-rooted at, the device-side pattern order, and the join
table of chain schedules as the kernels decode it."""
import collections

import numpy as np
import pytest

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


def _flat_caterpillar(L):
    """Post-order parents of a caterpillar with L leaves and a trifurcating root (L - 2 internal nodes)."""
    I = L - 2
    fp = np.empty(L + I, dtype=np.int64)
    fp[0] = fp[1] = 0                      # the cherry at the far end
    for k in range(1, I):
        fp[L + k - 1] = k                  # internal node k - 1 hangs below internal node k
        fp[k + 1] = k                      # ... together with leaf k + 1
    fp[L - 1] = I - 1                      # the root's third child
    fp[L + I - 1] = -1
    return fp


def test_chain_schedule_join_table_decodes_for_every_kernel_and_cut():
    """prune.hip decodes jn[n].x as parent | image slot << 16 (negative: root) and jn[n].y & 0xff as the arrivals a node waits
    for: every record of every cut must decode to the topology the schedule was built on — random trees, both chain kernels,
    re-rooted and not."""
    rng = np.random.default_rng(5)
    seen_chain = seen_rr = 0
    for trial in range(60):
        n = int(rng.integers(6, 200))
        flat = tree.flatten(tree.caterpillar_tree(n) if trial % 6 == 0 else tree.random_tree(n, rng, trifurcating_root=bool(rng.integers(0, 2))))
        for kernel in (1, 2):
            for m in (1, 3, 8):
                for rr in (False, True):
                    info = hip.plan_schedule(flat.flat_parents, flat.L, kernel=kernel, chain_m=m, ntiles=int(rng.integers(1, 700)), reroot=rr)
                    assert info["decode_errors"] == 0, (trial, kernel, m, rr, info)
                    if info["chain"]:
                        seen_chain += 1
                        assert info["max_slot"] < 32768 and info["max_need"] <= 255
                    seen_rr += info["rerooted"]
    assert seen_chain > 300 and seen_rr > 20


def test_trees_beyond_the_packed_join_table_get_no_chain_schedule():
    """r02 ADVICE: parent | slot << 16 in one signed int goes negative (= "root") once an image slot reaches 2^15, i.e. for
    trees of roughly 10 000 taxa and more (twin slots of re-rooted schedules first), and arrivals are an 8-bit field.  Such trees
    must fall back to a cut that needs no join table instead of silently truncating the pass."""
    ok = hip.plan_schedule(_flat_caterpillar(9000), 9000, kernel=1, chain_m=12, ntiles=4)
    assert ok["chain"] == 1 and ok["decode_errors"] == 0 and ok["max_slot"] < 32768
    big = hip.plan_schedule(_flat_caterpillar(12000), 12000, kernel=1, chain_m=12, ntiles=4)
    assert big["chain"] == 0 and big["programs"] >= 1
    big_rr = hip.plan_schedule(_flat_caterpillar(11000), 11000, kernel=2, chain_m=5, ntiles=4, reroot=True)
    assert big_rr["chain"] == 0
    # a star of 300 cherries below the root: 300 internal children of one node
    L, I = 600, 301
    fp = np.empty(L + I, dtype=np.int64)
    for k in range(300):
        fp[2 * k] = fp[2 * k + 1] = k
        fp[L + k] = 300
    fp[L + 300] = -1
    star = hip.plan_schedule(fp, L, kernel=1, chain_m=1, ntiles=16)
    assert star["chain"] == 0
    fp2 = fp[: 2 * 200 + 201].copy()       # 200 cherries: fits
    L2 = 400
    fp2 = np.empty(L2 + 201, dtype=np.int64)
    for k in range(200):
        fp2[2 * k] = fp2[2 * k + 1] = k
        fp2[L2 + k] = 200
    fp2[L2 + 200] = -1
    star2 = hip.plan_schedule(fp2, L2, kernel=1, chain_m=1, ntiles=16)
    assert star2["chain"] == 1 and star2["max_need"] == 200 and star2["decode_errors"] == 0


def test_subtree_repeat_classes_and_compressed_set():
    """Subtree repeats (repeats.hip; the reference's tcc masks, src/core/tree.cpp:2801-2858): the classes hyphy_hip_create counts per
    internal node are the distinct leaf sub-patterns below it, the compressed set is closed downwards and never contains the root, and
    the reference's run-length form (a site whose subtree leaves equal the PREVIOUS site's is skipped) can never skip more than the
    class form does."""
    from hyphy_amd import data
    rng = np.random.default_rng(5)
    for trial in range(6):
        taxa = int(rng.integers(6, 40))
        syn = data.evolve(taxa, 400, 3, seed=100 + trial, p_change=0.06)
        pd = data.from_states(syn.states, 61, compress_patterns=True)
        flat = syn.flat
        L, I = flat.L, flat.I
        codes = np.asarray(pd.leaf_codes)
        if trial % 2:   # ambiguity codes are characters like any other
            mask = rng.random(codes.shape) < 0.04
            codes = codes.copy()
            codes[mask] = -1 - rng.integers(0, 3, size=int(mask.sum()))
        S = codes.shape[1]
        theta = [0.2, 0.35, 0.6][trial % 3]
        classes, comp, work = hip.plan_repeats(flat.flat_parents, L, codes, theta)
        # independent count: leaves below every internal node, distinct columns of the leaf table restricted to them
        parents = np.asarray(flat.flat_parents)
        below = [set() for _ in range(I)]
        for n in range(L + I - 1):
            below[parents[n]] |= ({n} if n < L else below[n - L])
        for n in range(I):
            rows = sorted(below[n])
            want = len({tuple(codes[rows, s]) for s in range(S)})
            assert classes[n] == want, (trial, n, classes[n], want)
        assert not comp[I - 1]
        for n in range(I - 1):
            kids_ok = all(comp[c - L] for c in range(L, L + I - 1) if parents[c] == n)
            assert comp[n] == (classes[n] <= theta * S and classes[n] <= 32000 and kids_ok), (trial, n)
        assert work == sum(int(classes[n]) if comp[n] else S for n in range(I - 1))
        # the reference's masks: node n is skipped at site position s > 0 when its leaves equal those of position s - 1; at best
        # (sites sorted so that equal sub-patterns are neighbours) that leaves one evaluation per class
        for n in range(I - 1):
            rows = sorted(below[n])
            runs = 1 + sum(1 for s in range(1, S) if tuple(codes[rows, s]) != tuple(codes[rows, s - 1]))
            assert runs >= classes[n]


def test_nucgen_source_covers_every_node_and_compiles_for_gfx950():
    """hyphy_hip_plan_nucgen (no device; hiprtc cross-compiles): the straight-line kernel the library generates for a 4-state
    partition's steady-state full pass names every internal node once, reads every leaf's code once, takes the lookup path for
    plain leaves and the guarded general path for leaves with ambiguity codes, and compiles for gfx950."""
    import re
    rng = np.random.default_rng(5)
    for trial in range(6):
        n = int(rng.integers(4, 70))
        root = tree.random_tree(n, rng, trifurcating_root=bool(trial % 2))
        flat = tree.flatten(root)
        amb = np.zeros(flat.L, dtype=np.int64)
        if trial >= 3:
            amb[rng.integers(0, flat.L, size=2)] = 1
        small = bool(trial % 3 == 0)   # (the small-shard form: matrices in LDS, exponentials and combine inside the launch)
        src, ok = hip.plan_nucgen(flat.flat_parents, flat.L, amb, compile_it=True, small=small)
        assert src, "the generator must cover a plain full pass"
        assert ok, "hiprtc must compile the generated source for gfx950"
        finals = re.findall(r"double n(\d+)_0 = ", src)
        assert sorted(int(x) for x in finals) == list(range(flat.I))              # every internal node finalised exactly once
        codes = re.findall(r"const int k(\d+) = codes\[", src)
        assert sorted(int(x) for x in codes) == list(range(flat.L))               # every leaf code fetched once, up front
        assert src.count("leaf_general_(L_") == int(amb.sum())                    # guarded general path only where codes can be < 0
        assert len(re.findall(r"const double \*P\d+ = (?:L_|PT) \+ ", src)) == flat.I - 1   # one matrix per internal edge
        assert ("hyhip::expm4_one(" in src) == small and ("hyhip::combine_partials(" in src) == small
        assert "a.partials[" not in src                                           # lazy steady state: nothing persisted, nothing re-read
    # a tree beyond the LDS leaf table (L > 256): not covered, the interpreter stays
    big = tree.flatten(tree.caterpillar_tree(300))
    src, ok = hip.plan_nucgen(big.flat_parents, big.L, None, compile_it=False)
    assert src == "" and not ok


def _random_trunk(rng, n_internal):
    """A random tree over generalised leaves in the layout of a trunk view: leaves 0 .. L - 1, internal nodes L + i with children
    before parents, the root last, every internal node with at least two children."""
    kids = []                       # per internal node: children as ("leaf",) placeholders or internal indices
    roots = []                      # internal nodes without a parent yet
    for i in range(n_internal):
        k = []
        n_int = int(rng.integers(0, min(3, len(roots)) + 1)) if i + 1 < n_internal else len(roots)
        for _ in range(n_int):
            k.append(roots.pop(int(rng.integers(len(roots)))))
        n_leaf = max(0, 2 - len(k)) + int(rng.integers(0, 3))
        k += [None] * n_leaf
        kids.append(k)
        roots.append(i)
    L = sum(c is None for k in kids for c in k)
    parents = np.full(L + n_internal - 1, -1, dtype=np.int64)
    leaf = 0
    children = [[] for _ in range(n_internal)]
    for i, k in enumerate(kids):
        for c in k:
            if c is None:
                parents[leaf] = L + i
                children[i].append(leaf)
                leaf += 1
            else:
                parents[L + c] = L + i
                children[i].append(L + c)
    return L, parents, children


@pytest.mark.parametrize("seed", range(12))
def test_trunk_walk_program_computes_the_root_conditionals(seed):
    """hyphy_hip_plan_trunk_walk (repeats.hip: plan_trunk_walk, what trunk_walk_kernel interprets): random trunks, the program run by
    a small stack machine in numpy — inputs multiplied in, the edge product, a push in front of every second internal child's chain
    and its pop behind that child's product — against the plain recursion; one chain, and two chains whose products meet at the root."""
    from hyphy_amd import hip
    rng = np.random.default_rng(100 + seed)
    n_int = int(rng.integers(1, 14))
    L, parents, children = _random_trunk(rng, n_int)
    D = 3
    P = rng.uniform(0.1, 1.0, (n_int, D, D))          # one matrix per internal node's branch
    E_leaf = rng.uniform(0.1, 1.0, (L, D))            # what a generalised leaf contributes to its parent

    def cond(i):
        v = np.ones(D)
        for c in children[i]:
            v = v * (E_leaf[c] if c < L else P[c - L] @ cond(c - L))
        return v

    want = cond(n_int - 1)
    plan = hip.plan_trunk_walk(parents, L)
    assert plan["depth"] <= max(1, n_int)

    def run(first, end):
        acc, stack, seen = np.ones(D), [], 0
        for k in range(first, end):
            node, n_in, in0, flags = plan["nodes"][k]
            if flags & 1:
                stack.append(acc)
                acc = np.ones(D)
            for j in range(n_in):
                acc = acc * E_leaf[plan["inputs"][in0 + j]]
            if node < 0:
                assert k == end - 1 and not stack
                return acc
            seen += 1
            acc = P[node] @ acc
            if flags & 2:
                acc = acc * stack.pop()
        raise AssertionError("a chain must end at the root")

    one = run(*plan["one"])
    assert np.allclose(one, want, rtol=1e-12)
    if plan["two"] is not None:
        a, b = run(*plan["two"][0]), run(*plan["two"][1])
        assert np.allclose(a * b, want, rtol=1e-12)
        n0, n1 = (e - f - 1 for f, e in plan["two"])          # real nodes per chain
        assert n0 + n1 == n_int - 1 and min(n0, n1) >= 2 and 4 * min(n0, n1) >= max(n0, n1)
    # every internal node but the root is walked exactly once per form; the stack never goes deeper than reported
    first, end = plan["one"]
    walked = sorted(n for n, _, _, _ in plan["nodes"][first:end] if n >= 0)
    assert walked == list(range(n_int - 1))
    depth = d = 0
    for _, _, _, flags in plan["nodes"][first:end]:
        d += 1 if flags & 1 else 0
        depth = max(depth, d)
        d -= 1 if flags & 2 else 0
    assert depth <= plan["depth"]


def test_trunk_walk_program_of_a_balanced_trunk_nests_its_waiting_products():
    """A complete binary trunk of 15 internal nodes: the walk keeps the heavier (here: first) child's chain running and parks it
    while the sibling subtree is walked — three levels deep at the bottom; the two-chain form splits the root's two subtrees 7 + 7."""
    from hyphy_amd import hip
    depth_levels = 4
    n_int = 2 ** depth_levels - 1
    L = 2 ** depth_levels
    # internal nodes in post-order: build recursively
    parents = {}
    counter = {"leaf": 0, "int": 0}

    def build(level):
        if level == depth_levels:
            c = counter["leaf"]
            counter["leaf"] += 1
            return c
        a, b = build(level + 1), build(level + 1)
        me = L + counter["int"]
        counter["int"] += 1
        parents[a] = me
        parents[b] = me
        return me

    root = build(0)
    assert root == L + n_int - 1
    par = np.array([parents[c] for c in range(L + n_int - 1)], dtype=np.int64)
    plan = hip.plan_trunk_walk(par, L)
    assert plan["depth"] == 3 and plan["two"] is not None
    assert [e - f - 1 for f, e in plan["two"]] == [7, 7]
    first, end = plan["one"]
    assert sum(1 for _, _, _, fl in plan["nodes"][first:end] if fl & 1) == sum(1 for _, _, _, fl in plan["nodes"][first:end] if fl & 2) == 7
