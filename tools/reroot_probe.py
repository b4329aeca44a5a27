"""What would re-rooting the COMPUTATION at the node that minimises the tree height buy?  Times the pruning launch for a
workload's tree as given and for the same unrooted tree rooted at its best internal node (same data, same model; the
likelihood of a reversible model is the same, so the result is checked too).  Usage (GPU box): python tools/reroot_probe.py [workload]"""
import collections, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from hyphy_amd import data, hip, tree as htree

wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "mg94_32x5k"]
syn = data.evolve(wl["taxa"], wl["sites"], wl["unit"], seed=wl["seed"], p_change=wl.get("p_change", 0.04))


def neighbours(root):
    adj = collections.defaultdict(list)
    stack = [root]
    while stack:
        n = stack.pop()
        for c in n.children:
            adj[id(n)].append(c)
            adj[id(c)].append(n)
            stack.append(c)
    return adj


def height_from(node, adj):
    best, stack = 0, [(node, None, 0)]
    while stack:
        n, par, d = stack.pop()
        best = max(best, d)
        for m in adj[id(n)]:
            if m is not par:
                stack.append((m, n, d + 1))
    return best


def rerooted(root):
    adj = neighbours(root)
    inner, stack = [], [root]
    while stack:
        n = stack.pop()
        if n.children:
            inner.append(n)
            stack.extend(n.children)
    best = min(inner, key=lambda n: height_from(n, adj))

    def build(n, par):
        m = htree.Node(n.name)
        for k in adj[id(n)]:
            if k is not par:
                c = build(k, n)
                c.parent = m
                m.children.append(c)
        return m
    sys.setrecursionlimit(10000)
    return build(best, None), height_from(root, adj), height_from(best, adj)


def time_tree(root, states_by_name, label):
    flat = htree.flatten(root)
    states = np.stack([states_by_name[nm] for nm in flat.leaf_names])
    pd = data.from_states(states, 61)
    T, pi = bench.templates_for(3)
    B = flat.n_branches
    nodes = np.arange(B, dtype=np.int64)
    co = np.empty((B, 2))
    co[:, 0] = 0.05
    co[:, 1] = 0.05 * 0.3
    with hip.HipPartition(61, flat.flat_parents, flat.L, pd.leaf_codes, None, pd.pattern_freq) as part:
        part.set_q_templates(T)
        step = part.prepare_built_step(nodes, nodes, pi, co)
        for _ in range(300):
            ll = step()
        for k in range(300):
            co[:, 1] = 0.05 * (0.3 + 0.001 * k)
            ll = step()
        pt = part.prune_timings(256)
        print(f"{label:10s} I = {flat.I}  prune {1e3 * float(np.median(pt)):7.1f} us  logL {ll!r}  {part.schedule_info()}")


by_name = {nm: syn.states[k] for k, nm in enumerate(syn.flat.leaf_names)}
new_root, h0, h1 = rerooted(syn.tree)
print(f"tree height (edges to the deepest leaf): as given {h0}, rooted at its best internal node {h1}")
time_tree(syn.tree, by_name, "as given")
time_tree(new_root, by_name, "re-rooted")
