"""Randomised GPU stress test against the CPU oracle: random trees / sizes / kernels / fragment cuts and random
sequences of full passes, partial updates, pinned evaluations, branch-cache line searches and downloads.
Usage (GPU box): python tests/stress_codon.py [n_cases] [seed0]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
from hyphy_amd import data, models, tree
from oracle import oracle
PF = np.array([[0.3, 0.2, 0.25, 0.25], [0.2, 0.3, 0.3, 0.2], [0.25, 0.25, 0.2, 0.3]])
REV = dict(AC=0.5, AT=0.4, CG=0.4, CT=1.2, GT=0.4)
RTOL = 1e-10
t0 = time.time()
n_checks = 0
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    taxa = int(rng.integers(4, 70))
    codons = int(rng.integers(20, 700))
    kernel = str(int(rng.integers(0, 3)))   # 0: workgroup per tile, 1: wave per tile, 2: row-split workgroups on chain schedules
    frag = int(rng.choice([2, 3, 5, 8, 13, 1000]))
    slots, persist = str(int(rng.choice([2, 3]))), str(rng.choice(["lazy", "always"]))
    chain_m = str(int(rng.choice([0, 1, 2, 3, 5])))   # wave-per-tile kernel: chain schedule with sources of <= m nodes (0: level-peeled fragments)
    kernel, frag, slots, persist, chain_m = (os.environ.get("STRESS_" + k, v) for k, v in (("KERNEL", kernel), ("FRAGMENT", frag), ("SLOTS", slots), ("CACHE", persist), ("CHAIN_M", chain_m)))
    os.environ["HYPHY_HIP_WAVE_VARIANT"] = os.environ.get("STRESS_WAVE_VARIANT", str(int(rng.choice([0, 0, 1, 2]))))  # instantiation of the wave kernel
    if int(chain_m) > 0:
        os.environ["HYPHY_HIP_CHAIN_M"] = str(chain_m)
    else:
        os.environ.pop("HYPHY_HIP_CHAIN_M", None)
    os.environ["HYPHY_HIP_KERNEL"] = str(kernel)
    os.environ["HYPHY_HIP_FRAGMENT"] = str(frag)
    os.environ["HYPHY_HIP_SLOTS"] = slots
    os.environ["HYPHY_HIP_CACHE"] = persist
    tiles = os.environ.get("STRESS_TILES", str(int(rng.choice([1, 1, 1, 2, 4]))))
    os.environ["HYPHY_HIP_TILES"] = tiles
    from hyphy_amd import hip
    root = tree.random_tree(taxa, rng, trifurcating_root=bool(rng.integers(0, 2)))
    syn = data.evolve(taxa, codons, 3, seed=seed0 + case, tree=root, p_change=float(rng.uniform(0.02, 0.3)))
    seqs = syn.seqs
    if rng.random() < 0.4:
        seqs = data.inject_missing(seqs, 3, 0.05, seed0 + case)
    pd = data.compress(seqs, 3)
    flat = syn.flat
    B, L, I = flat.n_branches, flat.L, flat.I
    tb = rng.uniform(0.005, 0.8, B)
    Q = models.mg94rev_Q_batch(tb, 0.5, REV, PF)
    pi = models.f3x4_codon_freqs(PF)
    nodes = np.arange(B, dtype=np.int64)
    none = np.zeros(0, dtype=np.int64)
    op = oracle.OraclePartition(61, flat.flat_parents, L, pd.leaf_codes, pd.ambig, pd.pattern_freq)
    op.set_P(nodes, oracle.expm(Q, True))
    with hip.HipPartition(61, flat.flat_parents, L, pd.leaf_codes, pd.ambig, pd.pattern_freq) as part:
        def check(tag, got, ref):
            global n_checks
            n_checks += 1
            if not (abs(got - ref) <= RTOL * abs(ref) or (got == ref)):
                raise SystemExit(f"MISMATCH case {case} ({taxa} taxa, {codons} codons, kernel {kernel}, frag {frag}, chain_m {chain_m}, slots {slots}, cache {persist}, T {tiles}) {tag}: {got!r} vs {ref!r}")
        check("first", part.evaluate(nodes, nodes, Q, pi), op.compute_block(nodes, pi))
        for step in range(int(rng.integers(4, 10))):
            what = rng.choice(["full", "partial", "branch_cache", "pinned", "download"], p=[0.3, 0.3, 0.15, 0.15, 0.1])
            if what == "full":
                Q = models.mg94rev_Q_batch(tb * rng.uniform(0.8, 1.2), 0.5, REV, PF)
                op.set_P(nodes, oracle.expm(Q, True))
                check("full", part.evaluate(nodes, nodes, Q, pi), op.compute_block(nodes, pi))
            elif what == "partial":
                ch = np.unique(rng.integers(0, B, size=int(rng.integers(1, 4)))).astype(np.int64)
                Q[ch] = Q[ch] * rng.uniform(0.3, 3.0)
                upd = np.unique(np.concatenate([flat.path_update_nodes(int(n)) for n in ch])).astype(np.int64)
                op.set_P(ch, oracle.expm(Q[ch], True))
                check("partial", part.evaluate(upd, ch, Q[ch], pi), op.compute_block(upd, pi))
            elif what == "branch_cache":
                node = int(rng.integers(0, B))
                part.branch_cache_build(node)
                for f in (0.5, 2.0):
                    Q2 = Q[node] * f
                    got = part.branch_cache_evaluate(node, Q2)
                    op.set_P([node], oracle.expm(Q2, True)[None])
                    upd = flat.path_update_nodes(node)
                    check("branch_cache", got, op.compute_block(upd, pi))
                    Q[node] = Q2
                # the host tree now holds the last matrix; an ordinary pass over the path re-synchronises
                upd = flat.path_update_nodes(node)
                check("after_cache", part.evaluate(upd, none, np.zeros((0, 61, 61)), pi), op.compute_block(upd, pi))
            elif what == "pinned":
                code = int(rng.integers(0, L + I))
                states = rng.integers(0, 61, size=part.S)
                upd = set()
                if int(flat.flat_parents[code]) >= 0:
                    upd.update(int(x) for x in flat.path_update_nodes(code))
                if code >= L:
                    upd.update(int(c) for c in flat.children_of(code - L))
                upd = np.array(sorted(upd), dtype=np.int64)
                part.set_pinned_states(code, states)
                op.set_branch(code, states)
                got, ref = part.evaluate(upd, none, np.zeros((0, 61, 61)), pi), op.compute_block(upd, pi)
                part.set_pinned_states(None)
                op.set_branch(None)
                if np.isfinite(ref):
                    check("pinned", got, ref)
                check("unpinned", part.evaluate(upd, none, np.zeros((0, 61, 61)), pi), op.compute_block(upd, pi))
            else:
                cache, _ = part.download_partials()
                for n in range(op.I):
                    x, y = cache[n], op.cache[0][n]
                    sx, sy = x.sum(1, keepdims=True), y.sum(1, keepdims=True)
                    ok = (sy[:, 0] > 0)
                    if not np.allclose(x[ok] / sx[ok], y[ok] / sy[ok], rtol=1e-8, atol=1e-300):
                        raise SystemExit(f"MISMATCH case {case} download node {n}")
                n_checks += 1
    print(f"case {case}: {taxa} taxa x {codons} codons ({pd.S} patterns), kernel {kernel}, fragment {frag}, chain_m {chain_m}, slots {slots}, {persist}, T {tiles}: ok", flush=True)
print(f"{n_cases} cases, {n_checks} checks passed in {time.time() - t0:.0f} s")
