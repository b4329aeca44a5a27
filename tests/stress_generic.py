"""Randomised GPU stress test, second part: arbitrary state counts (4-state kernel, 1-4 MFMA row blocks with and
without padding), rate classes (batched pruning + mixing on the device), caterpillar and random trees, partial
updates per class.  Usage (GPU box): python tests/stress_generic.py [n_cases] [seed0]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
from hyphy_amd import data, tree
from oracle import oracle
RTOL = 1e-10
LOG_SCALER = 64 * np.log(2.0)


def random_q(D, t, rng_q):
    pi = rng_q.dirichlet(np.full(D, 5.0))
    R = rng_q.uniform(0.1, 2.0, (D, D))
    R = (R + R.T) * (rng_q.random((D, D)) < 0.6)  # sparse-ish symmetric exchangeabilities
    R = np.maximum(R, R.T)
    Q = R * pi[None, :]
    np.fill_diagonal(Q, 0.0)
    Q += np.diag(np.full(D - 1, 0.05), 1) * pi[None, :] + np.diag(np.full(D - 1, 0.05), -1) * pi[None, :]  # connected
    np.fill_diagonal(Q, 0.0)
    np.fill_diagonal(Q, -Q.sum(1))
    Q /= -(pi * np.diag(Q)).sum()
    return Q * t, pi


t0 = time.time()
n_checks = 0
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    D = int(rng.choice([4, 4, 20, 20, 30, 48, 61, 64]))
    taxa = int(rng.integers(3, 90))
    sites = int(rng.integers(5, 3000 if D == 4 else 500))
    n_cat = int(rng.choice([1, 1, 2, 3, 4]))
    kernel = os.environ.get("STRESS_KERNEL", str(int(rng.integers(0, 3))))
    frag = os.environ.get("STRESS_FRAGMENT", str(int(rng.choice([2, 4, 7, 11, 1000]))))
    persist = os.environ.get("STRESS_CACHE", str(rng.choice(["lazy", "always"])))
    tiles = os.environ.get("STRESS_TILES", str(int(rng.choice([1, 1, 2, 3, 4]))))  # patterns tiles per workgroup (workgroup kernel)
    shards = os.environ.get("STRESS_SHARDS", str(int(rng.choice([0, 0, 2, 3]))))  # pattern shards (here: all on one device)
    chain_m = os.environ.get("STRESS_CHAIN_M", str(int(rng.choice([0, 1, 2, 4]))))  # wave kernel: chain schedule (0: level-peeled fragments)
    os.environ["HYPHY_HIP_WAVE_VARIANT"] = os.environ.get("STRESS_WAVE_VARIANT", str(int(rng.choice([0, 0, 1, 2]))))  # instantiation of the wave kernel
    if int(chain_m) > 0:
        os.environ["HYPHY_HIP_CHAIN_M"] = chain_m
    else:
        os.environ.pop("HYPHY_HIP_CHAIN_M", None)
    os.environ["HYPHY_HIP_KERNEL"], os.environ["HYPHY_HIP_FRAGMENT"], os.environ["HYPHY_HIP_CACHE"] = kernel, frag, persist
    os.environ["HYPHY_HIP_TILES"], os.environ["HYPHY_HIP_FORCE_SHARDS"] = tiles, shards
    from hyphy_amd import hip
    root = tree.caterpillar_tree(taxa) if rng.random() < 0.25 else tree.random_tree(taxa, rng, trifurcating_root=bool(rng.integers(0, 2)))
    flat = tree.flatten(root)
    L, I, B = flat.L, flat.I, flat.n_branches
    # leaf states: a few "ancestral" columns with noise, so that patterns repeat and likelihoods stay sane
    base = rng.integers(0, D, size=sites)
    states = np.where(rng.random((L, sites)) < rng.uniform(0.05, 0.5), rng.integers(0, D, size=(L, sites)), base[None, :])
    pd = data.from_states(states, D, compress_patterns=bool(rng.integers(0, 2)))
    ambig = np.zeros((0, D))
    codes = pd.leaf_codes.copy()
    if rng.random() < 0.4:  # ambiguity vectors (resolution sets), referenced as -(index + 1)
        n_amb = int(rng.integers(1, 4))
        ambig = (rng.random((n_amb, D)) < 0.5).astype(np.float64)
        ambig[:, 0] = 1.0
        hit = rng.random(codes.shape) < 0.03
        codes[hit] = -(rng.integers(0, n_amb, size=int(hit.sum())) + 1)
    tb = rng.uniform(0.01, 0.6, B)
    rates = rng.uniform(0.2, 3.0, n_cat)
    weights = rng.dirichlet(np.full(n_cat, 3.0))
    Q1, pi = random_q(D, 1.0, np.random.default_rng(1000 + seed0 + case))
    # template model (device-side Q construction): Q_b = sum_k coeff[b][k] T_k; T_0 + T_1 = offdiag(Q1)
    templated = shards == "0" and (D > 4 or n_cat == 1) and rng.random() < 0.45  # (class batch from templates: MFMA path only)
    offd = Q1 - np.diag(np.diag(Q1))
    split = rng.random((D, D)) < 0.5
    T = np.stack([offd * split, offd * ~split])
    k1 = rng.uniform(0.3, 2.0)  # "omega"

    def q_from(coeff):  # [n][2] -> [n][D][D]
        Qn = coeff[:, 0, None, None] * T[0][None] + coeff[:, 1, None, None] * T[1][None]
        idx = np.arange(D)
        Qn[:, idx, idx] = -Qn.sum(2)
        return Qn

    co = np.stack([np.stack([tb * r, tb * r * k1], axis=1) for r in rates])  # [C][B][2]
    Q = np.stack([q_from(co[c]) for c in range(n_cat)])  # [C][B][D][D]
    nodes = np.arange(B, dtype=np.int64)
    none = np.zeros(0, dtype=np.int64)
    op = oracle.OraclePartition(D, flat.flat_parents, L, codes, ambig, pd.pattern_freq, n_cat)

    def oracle_value(upd):
        lik, sc = [], []
        for c in range(n_cat):
            a, b = op.site_block(upd, pi, cat=c)
            lik.append(a)
            sc.append(b)
        if n_cat == 1:
            ok = pd.pattern_freq > 0
            return float(np.sum(pd.pattern_freq[ok] * (np.log(lik[0][ok]) - sc[0][ok] * LOG_SCALER)))
        return oracle.mix_categories(weights, np.stack(lik), np.stack(sc), pd.pattern_freq)[0]

    def check(tag, got, ref):
        global n_checks
        n_checks += 1
        if not (abs(got - ref) <= RTOL * abs(ref) or got == ref):
            raise SystemExit(f"MISMATCH case {case} (D {D}, {taxa} taxa, {pd.S} patterns, {n_cat} classes, kernel {kernel}, frag {frag}, chain_m {chain_m}, {persist}, T {tiles}, shards {shards}, templated {templated}) {tag}: {got!r} vs {ref!r}")

    for c in range(n_cat):
        op.set_P(nodes, oracle.expm(Q[c], D > 4), cat=c)
    with hip.HipPartition(D, flat.flat_parents, L, codes, ambig, pd.pattern_freq, n_cat) as part:
        if templated:
            part.set_q_templates(T)

        def device_value(upd, qn):
            if templated:
                cc = np.ascontiguousarray(co[:, qn].reshape(-1, 2))  # class-major coefficient rows of the changed branches
                if n_cat == 1:
                    return part.prepare_built_step(upd, qn, pi, cc)()
                return part.prepare_built_categories_step(upd, qn, weights, pi, cc)()
            q = np.ascontiguousarray(Q[:, qn])
            if n_cat == 1:
                return part.evaluate(upd, qn, q[0] if len(qn) else np.zeros((0, D, D)), pi)
            return part.evaluate_categories(upd, qn, q if len(qn) else np.zeros((n_cat, 0, D, D)), weights, pi)
        check("first", device_value(nodes, nodes), oracle_value(nodes))
        for step in range(int(rng.integers(3, 8))):
            what = rng.choice(["full", "partial", "none"], p=[0.4, 0.5, 0.1])
            if what == "full":
                co = co * rng.uniform(0.8, 1.25)
                Q = np.stack([q_from(co[c]) for c in range(n_cat)])
                for c in range(n_cat):
                    op.set_P(nodes, oracle.expm(Q[c], D > 4), cat=c)
                check("full", device_value(nodes, nodes), oracle_value(nodes))
            elif what == "partial":
                ch = np.unique(rng.integers(0, B, size=int(rng.integers(1, 4)))).astype(np.int64)
                co[:, ch] = co[:, ch] * rng.uniform(0.3, 3.0)
                Q = np.stack([q_from(co[c]) for c in range(n_cat)])
                upd = np.unique(np.concatenate([flat.path_update_nodes(int(n)) for n in ch])).astype(np.int64)
                for c in range(n_cat):
                    op.set_P(ch, oracle.expm(Q[c][ch], D > 4), cat=c)
                check("partial", device_value(upd, ch), oracle_value(upd))
            elif not templated:
                check("nothing dirty", device_value(none, none), oracle_value(none))
        if templated and D > 4 and n_cat == 1 and rng.random() < 0.6:
            # per-site batched fits on a few patterns: every pattern under its own multipliers, against one
            # oracle likelihood function per pattern with explicitly exponentiated matrices
            G = int(rng.integers(1, 4))
            bgroup = rng.integers(0, G, size=B)
            bcoef = co[0] * rng.uniform(0.5, 2.0)
            n_sets = int(rng.integers(1, 4))
            smult = np.exp(rng.uniform(np.log(0.02), np.log(30.0), (n_sets, pd.S, G, 2)))
            got = part.site_fits_evaluate(bgroup, bcoef, smult, pi)
            for st, s_ in zip(rng.integers(0, n_sets, size=5), rng.integers(0, pd.S, size=5)):
                Qs = q_from(smult[st, s_][bgroup] * bcoef)
                o1 = oracle.OraclePartition(D, flat.flat_parents, L, codes[:, s_:s_ + 1], ambig, np.ones(1, dtype=np.int64))
                o1.set_P(nodes, oracle.expm(Qs, True))
                ref = o1.site_log_likelihoods(nodes, pi)[0]
                n_checks += 1
                if not (abs(got[st, s_] - ref) <= 1e-9 * max(1.0, abs(ref)) or got[st, s_] == ref):
                    raise SystemExit(f"MISMATCH case {case} site fits (D {D}, {taxa} taxa, set {st}, pattern {s_}): {got[st, s_]!r} vs {ref!r}")
    print(f"case {case}: D {D}, {taxa} taxa, {pd.S} patterns, {n_cat} classes, kernel {kernel}, fragment {frag}, chain_m {chain_m}, {persist}, T {tiles}, shards {shards}, templated {templated}: ok", flush=True)
print(f"{n_cases} cases, {n_checks} checks passed in {time.time() - t0:.0f} s")
