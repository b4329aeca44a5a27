"""GPU parity tests proper: the HIP path (through the C-ABI) against the committed golden
vectors of the real reference and against the CPU oracle on the same seeded inputs.
Tolerance: north_star asks 1e-6 relative on logL; FP64 end to end lets us hold 1e-10."""
import os
import numpy as np
import pytest

from tests import common

pytestmark = pytest.mark.gpu

RTOL = 1e-10
CASES = ["codon_small", "codon_ambig", "codon_deep", "codon_wide", "nuc_small", "nuc_ambig", "nuc_deep", "nuc_wide",
         "ref_smallcodon", "ref_fluHA"]  # (the last two: real data of the reference's own tests SimpleOptimizations/SmallCodon.bf, IntermediateNuc.bf)


def _hip():
    from hyphy_amd import hip
    return hip


def _mk(fx, C=1, **kw):
    hip = _hip()
    return hip.HipPartition(int(fx["D"]), fx["flat_parents"], int(fx["L"]), fx["leaf_codes"], fx["ambig"],
                            fx["pattern_freq"], C, **kw)


def test_device_present_and_library_loaded():
    hip = _hip()
    assert hip.device_count() >= 1
    assert b"gfx950" in hip.load().hyphy_hip_version()


def test_expm_batch_matches_reference_goldens():
    hip = _hip()
    z = common.load("expm")
    for k in z:
        if not k.startswith("Q_"):
            continue
        P = hip.expm_batch(z[k])
        Pref = z["P_" + k[2:]]
        assert np.max(np.abs(P - Pref)) < 5e-14, (k, np.max(np.abs(P - Pref)))
        assert np.max(np.abs(P.sum(1) - 1)) < 1e-14


def test_expm_batch_matches_oracle_on_benchmark_matrices():
    from hyphy_amd import models
    from oracle import oracle
    hip = _hip()
    rng = np.random.default_rng(3)
    ts = rng.uniform(0.001, 2.0, size=40)
    pf = np.array([[0.3, 0.2, 0.25, 0.25], [0.2, 0.3, 0.3, 0.2], [0.25, 0.25, 0.2, 0.3]])
    Q = models.mg94rev_Q_batch(ts, 0.7, dict(AC=0.5, AT=0.4, CG=0.4, CT=1.2, GT=0.4), pf)
    P = hip.expm_batch(Q)
    Po = oracle.expm(Q, True)
    assert np.max(np.abs(P - Po)) < 5e-14


@pytest.mark.parametrize("name", CASES)
def test_logl_and_sites_match_reference(name):
    fx = common.load(name)
    Q = common.fixture_Q(fx)
    nodes = common.all_nodes(fx)
    with _mk(fx) as part:
        ll, lik, sc = part.evaluate(nodes, nodes, Q, fx["root_freqs"], per_site=True)
    ref = float(fx["logl"])
    assert abs(ll - ref) <= RTOL * abs(ref), (ll, ref)
    site = (np.log(lik) - sc * 64 * np.log(2.0))[fx["site_to_pattern"]]
    assert np.max(np.abs(site - fx["site_logl"]) / np.abs(fx["site_logl"])) < RTOL
    if name.endswith("deep"):
        assert sc.max() > 0  # the rescaling branch was really exercised


@pytest.mark.parametrize("kernel,chain_m", [("1", None), ("2", "2"), ("2", "5")])
@pytest.mark.parametrize("name", ["codon_small", "codon_ambig", "codon_deep", "codon_wide"])
def test_wave_kernels_match_reference(name, kernel, chain_m, monkeypatch):
    """The goldens are small shards (the library would pick the workgroup-per-tile kernel): force the
    wave-per-tile kernel (register hand-over, chained fragments) and the row-split workgroups on chain schedules
    (kernel 2: joins through global memory, every product split over four waves) through the same checks, including the
    downloaded conditionals of every internal node."""
    from oracle import oracle
    monkeypatch.setenv("HYPHY_HIP_KERNEL", kernel)
    if chain_m:
        monkeypatch.setenv("HYPHY_HIP_CHAIN_M", chain_m)
        monkeypatch.setenv("HYPHY_HIP_POISON", "1")
    fx = common.load(name)
    nodes = common.all_nodes(fx)
    Q = common.fixture_Q(fx)
    with _mk(fx) as part:
        assert part.prune_kernel_name() == ("prune_wave_kernel" if kernel == "1" else "prune_mfma_kernel")
        ll, sl, sc = part.evaluate(nodes, nodes, Q, fx["root_freqs"], per_site=True)
        cache, counts = part.download_partials()
    ref = float(fx["logl"])
    assert abs(ll - ref) <= RTOL * abs(ref)
    site = np.log(sl) - sc * 64 * np.log(2.0)
    op = oracle.OraclePartition(int(fx["D"]), fx["flat_parents"], int(fx["L"]), fx["leaf_codes"], fx["ambig"], fx["pattern_freq"])
    op.set_P(nodes, oracle.expm(Q, True))
    osl = op.site_log_likelihoods(nodes, fx["root_freqs"])
    assert np.allclose(site, osl, rtol=1e-10, atol=1e-9)
    oc = op.cache[0]   # conditionals node by node (normalised: the exponent bookkeeping differs by design)
    for n in range(op.I):
        a, b = cache[n], oc[n]
        assert np.allclose(a / a.sum(1, keepdims=True), b / b.sum(1, keepdims=True), rtol=1e-9, atol=1e-300), (name, n)


@pytest.mark.parametrize("name", ["codon_small", "codon_deep", "nuc_ambig"])
def test_partials_match_oracle(name):
    """download_partials returns the reference layout; conditionals agree with the CPU
    restatement node by node once the power-of-2^64 exponents are applied."""
    from oracle import oracle
    fx = common.load(name)
    Q = common.fixture_Q(fx)
    nodes = common.all_nodes(fx)
    op = oracle.OraclePartition(int(fx["D"]), fx["flat_parents"], int(fx["L"]), fx["leaf_codes"], fx["ambig"],
                                fx["pattern_freq"])
    P = oracle.expm(Q, str(fx["kind"]) == "codon")
    op.set_P(nodes, P)
    op.compute_block(nodes, fx["root_freqs"])
    with _mk(fx) as part:
        part.evaluate(nodes, nodes, P, fx["root_freqs"], q_is_probability=True)
        cache, counts = part.download_partials()
    # oracle: stored = true * sticky factor  ->  compare log-magnitudes and normalised vectors
    oc = op.cache[0]
    for n in range(op.I):
        a, b = cache[n], oc[n]
        na, nb = a.sum(1, keepdims=True), b.sum(1, keepdims=True)
        assert np.allclose(a / na, b / nb, rtol=1e-9, atol=1e-300), (name, n)


@pytest.mark.parametrize("materialize", [False, True])
def test_device_q_templates_match_host_q(materialize, monkeypatch):
    """hyphy_hip_set_q_templates / build_q / evaluate_device (device-resident Q, fused into the expm
    kernel or materialised) agree with host-built rate matrices through hyphy_hip_evaluate."""
    import torch
    import bench
    from hyphy_amd import models
    if materialize:
        monkeypatch.setenv("HYPHY_HIP_MATERIALIZE_Q", "1")
    fx = common.load("codon_wide")
    nodes = common.all_nodes(fx)
    T, pi = bench.templates_for(3)
    tb = np.asarray(fx["t"], dtype=np.float64)
    omega = 0.37
    Q = models.mg94rev_Q_batch(tb, omega, bench.REV, bench.POS_FREQS)
    d_out = torch.zeros(2, dtype=torch.float64, device="cuda")
    with _mk(fx) as part:
        ref = part.evaluate(nodes, nodes, Q, pi)
    with _mk(fx) as part:
        part.set_q_templates(T)
        co = np.stack([tb, tb * omega], axis=1).copy()
        step = part.prepare_device_step(nodes, nodes, pi, d_out.data_ptr(), co)
        step()
        part.synchronize()
        got = float(d_out[0].item())
        # second call with a different omega re-uses the cached schedule / slots
        co[:, 1] = tb * 0.5
        step()
        part.synchronize()
        got2 = float(d_out[0].item())
        # the same device scalar through the host-mapped result record (hyphy_hip_fetch_device_scalar: what a multi-rank
        # host uses behind its all-reduce), with further work queued in front of it and interleaved synchronous calls
        fetch = part.prepare_fetch(d_out.data_ptr())
        co[:, 1] = tb * omega
        step()
        got3 = fetch()
        co[:, 1] = tb * 0.5
        stream = torch.cuda.Stream()
        part.set_stream(stream.cuda_stream)
        with torch.cuda.stream(stream):
            step()
            d_out[0] += 1.0      # (the caller's own work between the evaluation and the read-back, same stream)
            got4 = fetch() - 1.0
        part.set_stream(-1)      # (HYPHY_HIP_OWN_STREAM)
    with _mk(fx) as part:
        ref2 = part.evaluate(nodes, nodes, models.mg94rev_Q_batch(tb, 0.5, bench.REV, bench.POS_FREQS), pi)
    assert abs(got - ref) <= RTOL * abs(ref)
    assert abs(got2 - ref2) <= RTOL * abs(ref2)
    assert got3 == got and abs(got4 - got2) <= 1e-12 * abs(got2)


def test_q_is_probability_path():
    from oracle import oracle
    fx = common.load("codon_small")
    nodes = common.all_nodes(fx)
    P = oracle.expm(common.fixture_Q(fx), True)
    with _mk(fx) as part:
        ll = part.evaluate(nodes, nodes, P, fx["root_freqs"], q_is_probability=True)
    assert abs(ll - float(fx["logl"])) <= RTOL * abs(float(fx["logl"]))


@pytest.mark.parametrize("seed,taxa,trif", [(1, 64, True), (2, 96, True), (3, 33, False)])
def test_random_trees_match_oracle(seed, taxa, trif, monkeypatch):
    """Random topologies at the benchmark's shape (many pending subtrees -> exercises the LDS slot
    cache, slot exhaustion and the register hand-off), every tile-count variant of the kernel."""
    from hyphy_amd import data, models, tree
    from oracle import oracle
    rng = np.random.default_rng(seed)
    root = tree.random_tree(taxa, rng, trifurcating_root=trif)
    syn = data.evolve(taxa, 150, 3, seed=seed, tree=root)
    pd = data.from_states(syn.states, 61)
    flat = syn.flat
    pf = np.array([[0.3, 0.2, 0.25, 0.25], [0.2, 0.3, 0.3, 0.2], [0.25, 0.25, 0.2, 0.3]])
    Q = models.mg94rev_Q_batch(rng.uniform(0.01, 0.3, flat.n_branches), 0.5,
                               dict(AC=0.5, AT=0.4, CG=0.4, CT=1.2, GT=0.4), pf)
    pi = models.f3x4_codon_freqs(pf)
    nodes = np.arange(flat.n_branches, dtype=np.int64)
    op = oracle.OraclePartition(61, flat.flat_parents, flat.L, pd.leaf_codes, None, pd.pattern_freq)
    op.set_P(nodes, oracle.expm(Q, True))
    ref = op.compute_block(nodes, pi)
    hip = _hip()
    for tiles in ("1", "2", "3", "4"):
        monkeypatch.setenv("HYPHY_HIP_TILES", tiles)
        with hip.HipPartition(61, flat.flat_parents, flat.L, pd.leaf_codes, None, pd.pattern_freq) as part:
            ll = part.evaluate(nodes, nodes, Q, pi)
        assert abs(ll - ref) <= RTOL * abs(ref), (tiles, ll, ref)


@pytest.mark.parametrize("D", [2, 5, 16, 20, 33, 48, 64])
def test_generic_state_counts_match_oracle(D, monkeypatch):
    """Every row-block count of the MFMA kernel (NW = 1..4) and the expm variants, on random reversible
    D-state models (D = 20: amino acids), with ambiguity vectors, against the CPU oracle."""
    from hyphy_amd import models, tree
    from oracle import oracle
    rng = np.random.default_rng(100 + D)
    root = tree.random_tree(14, rng)
    flat = tree.flatten(root)
    L, B, S = flat.L, flat.n_branches, 70
    pi = rng.dirichlet(np.ones(D) * 4)
    Sx = rng.uniform(0.1, 2.0, size=(D, D))
    Sx = (Sx + Sx.T) / 2
    Q = np.empty((B, D, D))
    for b, t in enumerate(rng.uniform(0.02, 0.6, B)):
        q = Sx * pi[None, :] * t
        np.fill_diagonal(q, 0.0)
        Q[b] = models.finish_rate_matrix(q)
    codes = rng.integers(0, D, size=(L, S)).astype(np.int64)
    ambig = (rng.random((3, D)) < 0.5).astype(np.float64)
    ambig[:, 0] = 1.0
    ambig[2, :] = 1.0
    codes[rng.random((L, S)) < 0.08] = -int(rng.integers(1, 4))
    freq = rng.integers(1, 5, size=S).astype(np.int64)
    nodes = np.arange(B, dtype=np.int64)
    op = oracle.OraclePartition(D, flat.flat_parents, L, codes, ambig, freq)
    op.set_P(nodes, oracle.expm(Q, False))
    ref = op.compute_block(nodes, pi)
    hip = _hip()
    for tiles in ("1", "2", "4"):
        monkeypatch.setenv("HYPHY_HIP_TILES", tiles)
        with hip.HipPartition(D, flat.flat_parents, L, codes, ambig, freq) as part:
            ll, lik, sc = part.evaluate(nodes, nodes, Q, pi, per_site=True)
            cache, counts = part.download_partials()
        assert abs(ll - ref) <= RTOL * abs(ref), (D, tiles, ll, ref)
        assert cache.shape == (flat.I, S, D) and np.isfinite(cache).all()


def test_multifurcating_tree_matches_oracle():
    from hyphy_amd import data, models, tree
    from oracle import oracle
    newick = "((a,b,c,d,e)X,(f,(g,h,i)Y,j)Z,k,(l,m)W,n)"
    root = tree.parse_newick(newick)
    syn = data.evolve(14, 120, 3, seed=9, tree=root)
    pd = data.from_states(syn.states, 61)
    flat = syn.flat
    pf = np.array([[0.3, 0.2, 0.25, 0.25], [0.2, 0.3, 0.3, 0.2], [0.25, 0.25, 0.2, 0.3]])
    Q = models.mg94rev_Q_batch(np.linspace(0.02, 0.4, flat.n_branches), 0.4,
                               dict(AC=0.5, AT=0.4, CG=0.4, CT=1.2, GT=0.4), pf)
    pi = models.f3x4_codon_freqs(pf)
    nodes = np.arange(flat.n_branches, dtype=np.int64)
    op = oracle.OraclePartition(61, flat.flat_parents, flat.L, pd.leaf_codes, None, pd.pattern_freq)
    op.set_P(nodes, oracle.expm(Q, True))
    ref = op.compute_block(nodes, pi)
    with _hip().HipPartition(61, flat.flat_parents, flat.L, pd.leaf_codes, None, pd.pattern_freq) as part:
        ll = part.evaluate(nodes, nodes, Q, pi)
    assert abs(ll - ref) <= RTOL * abs(ref)


@pytest.mark.parametrize("name", ["codon_deep", "nuc_deep", "codon_small", "codon_wide"])
def test_partial_update_equals_full(name):
    """DetermineNodesForUpdate-style dirty lists: change one branch at a time."""
    from hyphy_amd import tree
    from oracle import oracle
    fx = common.load(name)
    L = int(fx["L"])
    flat = tree.flat_from_parents(fx["flat_parents"], L)
    Q = common.fixture_Q(fx)
    nodes = common.all_nodes(fx)
    sparse = str(fx["kind"]) == "codon"
    op = oracle.OraclePartition(int(fx["D"]), fx["flat_parents"], L, fx["leaf_codes"], fx["ambig"], fx["pattern_freq"])
    op.set_P(nodes, oracle.expm(Q, sparse))
    op.compute_block(nodes, fx["root_freqs"])
    rng = np.random.default_rng(1)
    with _mk(fx) as part:
        part.evaluate(nodes, nodes, Q, fx["root_freqs"])
        for node in rng.choice(len(nodes), size=6, replace=False):
            Q[node] = Q[node] * rng.uniform(0.3, 3.0)
            un = flat.path_update_nodes(int(node))
            ll = part.evaluate(un, [node], Q[node][None], fx["root_freqs"])
            op.set_P([node], oracle.expm(Q[node], sparse)[None])
            ref = op.compute_block(un, fx["root_freqs"])
            assert abs(ll - ref) <= RTOL * abs(ref), (name, node, ll, ref)


@pytest.mark.parametrize("name,kernel,shards", [("codon_wide", "0", None), ("codon_wide", "1", None), ("codon_deep", "0", None),
                                                ("codon_deep", "1", None), ("codon_ambig", "1", None),
                                                ("codon_small", "0", "3"), ("codon_ambig", "0", None)])
def test_branch_cache_equals_full_evaluation(name, kernel, shards, monkeypatch):
    """hyphy_hip_branch_cache_build / _evaluate (device ComputeBranchCache + ComputeLLWithBranchCache,
    tree_evaluator.cpp:4286, tree.cpp:3383): for a sample of branches (leaf branches, internal branches,
    branches at the root, the deepest ones) the cached one-contraction log-L equals the full evaluation —
    at the build-time branch length (the reference's own checksum, likefunc.cpp:11177-11250) and at changed
    lengths — and the oracle's full evaluation with that matrix substituted."""
    from oracle import oracle
    monkeypatch.setenv("HYPHY_HIP_KERNEL", kernel)  # workgroup-per-tile / wave-per-tile pruning kernel
    if shards:
        monkeypatch.setenv("HYPHY_HIP_FORCE_SHARDS", shards)
    fx = common.load(name)
    nodes = common.all_nodes(fx)
    Q = common.fixture_Q(fx)
    pi = fx["root_freqs"]
    L, I = int(fx["L"]), len(fx["flat_parents"]) - int(fx["L"])
    B = L + I - 1
    parents = np.asarray(fx["flat_parents"])
    depth = np.zeros(L + I, dtype=int)  # the root (last internal node) has depth 0; parents have larger indices
    for n in list(range(L + I - 2, L - 1, -1)) + list(range(L)):
        depth[n] = depth[L + parents[n]] + 1
    rng = np.random.default_rng(7)
    picks = {0, L - 1, L, B - 1, int(np.argmax(depth[:L])), int(np.argmax(depth[L:B]) + L)}
    picks |= set(int(x) for x in rng.integers(0, B, size=4))
    D = int(fx["D"])
    op = oracle.OraclePartition(D, fx["flat_parents"], L, fx["leaf_codes"], fx["ambig"], fx["pattern_freq"])
    with _mk(fx) as part:
        full = part.evaluate(nodes, nodes, Q, pi)
        for node in sorted(picks):
            part.evaluate(nodes, nodes, Q, pi)  # (an ordinary evaluation drops the cache and restores every matrix)
            part.branch_cache_build(node)
            same = part.branch_cache_evaluate(node, Q[node])
            assert abs(same - full) <= 1e-10 * abs(full), (name, node, same, full)
            for f in (0.25, 3.0):
                Q2 = Q.copy()
                Q2[node] = Q[node] * f
                cached = part.branch_cache_evaluate(node, Q2[node])
                op.set_P(nodes, oracle.expm(Q2, True))
                ref = op.compute_block(nodes, pi)
                assert abs(cached - ref) <= RTOL * abs(ref), (name, node, f, cached, ref)


def test_branch_cache_per_class_and_per_site():
    """One cache per rate class; per-pattern outputs of the cached evaluation equal those of a full one."""
    fx = common.load("codon_cat3")
    nodes = common.all_nodes(fx)
    pi = fx["root_freqs"]
    vals = [float(v) for v in fx["cat_values"]]
    Qs = [common.fixture_Q(fx, v) for v in vals]
    node = int(fx["L"]) + 2
    with _mk(fx, C=len(vals)) as part:
        for c, Q in enumerate(Qs):
            part.evaluate(nodes, nodes, Q, pi, cat=c)
        for c in range(len(vals)):
            part.branch_cache_build(node, cat=c)
        for c, Q in enumerate(Qs):
            ll, sl, sc = part.branch_cache_evaluate(node, Q[node] * 1.7, cat=c, per_site=True)
            Q2 = Q.copy()
            Q2[node] = Q[node] * 1.7
            with _mk(fx) as p2:
                ref, rl, rc = p2.evaluate(nodes, nodes, Q2, pi, per_site=True)
            assert abs(ll - ref) <= RTOL * abs(ref)
            assert np.allclose(np.log(sl) - sc * 64 * np.log(2.0), np.log(rl) - rc * 64 * np.log(2.0), rtol=1e-10, atol=1e-9)


def test_branch_cache_requires_build_and_is_dropped_by_evaluate():
    fx = common.load("codon_small")
    nodes = common.all_nodes(fx)
    Q = common.fixture_Q(fx)
    with _mk(fx) as part:
        part.evaluate(nodes, nodes, Q, fx["root_freqs"])
        with pytest.raises(RuntimeError):
            part.branch_cache_evaluate(3, Q[3])
        part.branch_cache_build(3)
        part.branch_cache_evaluate(3, Q[3])
        part.evaluate(nodes, nodes, Q, fx["root_freqs"])
        with pytest.raises(RuntimeError):
            part.branch_cache_evaluate(3, Q[3])


@pytest.mark.parametrize("kernel", ["0", "1"])
def test_lazy_persistence_is_transparent(kernel, monkeypatch):
    """Default cache policy: a full pass that follows a full pass does not store its conditionals.  A partial
    update, a download or a branch-cache build after such a pass must still see current data (the library
    re-runs a persisting pass first), and HYPHY_HIP_CACHE=always gives the same numbers."""
    from oracle import oracle
    monkeypatch.setenv("HYPHY_HIP_KERNEL", kernel)
    fx = common.load("codon_wide")
    nodes = common.all_nodes(fx)
    pi = fx["root_freqs"]
    L = int(fx["L"])
    Q1, Q2 = common.fixture_Q(fx), common.fixture_Q(fx, 1.3)
    op = oracle.OraclePartition(int(fx["D"]), fx["flat_parents"], L, fx["leaf_codes"], fx["ambig"], fx["pattern_freq"])

    def ref(Q):
        op.set_P(nodes, oracle.expm(Q, True))
        return op.compute_block(nodes, pi)
    from hyphy_amd import tree
    flat = tree.flat_from_parents(fx["flat_parents"], L)
    changed = np.array([3, L + 5], dtype=np.int64)
    upd = np.unique(np.concatenate([flat.path_update_nodes(int(n)) for n in changed])).astype(np.int64)
    Q3 = Q2.copy()
    Q3[changed] = Q1[changed] * 0.5
    r2, r3 = ref(Q2), ref(Q3)
    results = {}
    for policy in ("lazy", "always"):
        monkeypatch.setenv("HYPHY_HIP_CACHE", policy)
        with _mk(fx) as part:
            part.evaluate(nodes, nodes, Q1, pi)            # first pass: stored
            a = part.evaluate(nodes, nodes, Q2, pi)        # full after full: lazy -> not stored
            cache, _ = part.download_partials()            # must reflect Q2
            b = part.evaluate(nodes, nodes, Q2, pi)        # lazy again
            c = part.evaluate(upd, changed, Q3[changed], pi)   # partial update on stale copies -> promoted
            part.branch_cache_build(L + 5)
            d = part.branch_cache_evaluate(L + 5, Q3[L + 5])
        results[policy] = (a, b, c, d, cache)
        assert abs(a - r2) <= RTOL * abs(r2) and abs(b - r2) <= RTOL * abs(r2)
        assert abs(c - r3) <= RTOL * abs(r3) and abs(d - r3) <= RTOL * abs(r3)
        op.set_P(nodes, oracle.expm(Q2, True))
        op.compute_block(nodes, pi)
        for n in range(op.I):
            x, y = cache[n], op.cache[0][n]
            assert np.allclose(x / x.sum(1, keepdims=True), y / y.sum(1, keepdims=True), rtol=1e-9, atol=1e-300), (policy, n)
    # (not bit for bit: the steady-state passes of the lazy policy may run a re-rooted schedule, a different order of the same products)
    for x, y in zip(results["lazy"][:4], results["always"][:4]):
        assert abs(x - y) <= 1e-13 * abs(y)


def test_lazy_persistence_nucleotide_path():
    """Same policy in the 4-state kernel: children still in registers are not stored in a sweep; a partial
    update and a download afterwards see current data."""
    from hyphy_amd import tree
    from oracle import oracle
    fx = common.load("nuc_wide")
    nodes = common.all_nodes(fx)
    pi = fx["root_freqs"]
    L = int(fx["L"])
    flat = tree.flat_from_parents(fx["flat_parents"], L)
    Q1, Q2 = common.fixture_Q(fx), common.fixture_Q(fx, 1.4)
    op = oracle.OraclePartition(4, fx["flat_parents"], L, fx["leaf_codes"], fx["ambig"], fx["pattern_freq"])
    op.set_P(nodes, oracle.expm(Q2, False))
    r2 = op.compute_block(nodes, pi)
    changed = np.array([5, L + 7], dtype=np.int64)
    upd = np.unique(np.concatenate([flat.path_update_nodes(int(n)) for n in changed])).astype(np.int64)
    Q3 = Q2.copy()
    Q3[changed] = Q1[changed] * 0.6
    with _mk(fx) as part:
        part.evaluate(nodes, nodes, Q1, pi)
        a = part.evaluate(nodes, nodes, Q2, pi)          # full after full: lazy
        cache, _ = part.download_partials()
        b = part.evaluate(nodes, nodes, Q2, pi)
        c = part.evaluate(upd, changed, Q3[changed], pi)  # promoted
    assert abs(a - r2) <= RTOL * abs(r2) and abs(b - r2) <= RTOL * abs(r2)
    for n in range(op.I):
        x, y = cache[n], op.cache[0][n]
        assert np.allclose(x / x.sum(1, keepdims=True), y / y.sum(1, keepdims=True), rtol=1e-9, atol=1e-300), n
    op.set_P(nodes, oracle.expm(Q3, False))
    r3 = op.compute_block(nodes, pi)
    assert abs(c - r3) <= RTOL * abs(r3)


def _pin_update_list(flat, code):
    """Dirty list of an evaluation with node ``code`` pinned: DetermineNodesForUpdate with addOne = the node
    (tree.cpp:3268-3297) lists the node, its ancestors and the direct children of every touched internal node —
    including the pinned node's own children, so that the node itself is recomputed."""
    L = flat.L
    out = set()
    if int(flat.flat_parents[code]) >= 0:
        out.update(int(x) for x in flat.path_update_nodes(int(code)))
    if code >= L:
        out.update(int(c) for c in flat.children_of(code - L))
    return np.array(sorted(out), dtype=np.int64)


@pytest.mark.parametrize("kernel", ["0", "1"])
def test_pinned_states_reproduce_reference_marginal_support(kernel, monkeypatch):
    """hyphy_hip_set_pinned_states (ComputeBlock's branchIndex / branchValues): the device's pinned per-pattern
    likelihoods reproduce the support matrix of the REAL reference's ReconstructAncestors (lf, MARGINAL) —
    partial updates along the path of the pinned node, exactly the call sequence of
    RecoverAncestralSequencesMarginal (likefunc2.cpp:932-1040)."""
    from hyphy_amd import tree
    monkeypatch.setenv("HYPHY_HIP_KERNEL", kernel)
    fx = common.load("codon_small_marginal")
    L = int(fx["L"])
    nodes = common.all_nodes(fx)
    Q = common.fixture_Q(fx)
    pi = fx["root_freqs"]
    flat = tree.flat_from_parents(fx["flat_parents"], L)
    I = flat.I
    none = np.zeros(0, dtype=np.int64)
    with _mk(fx) as part:
        S = part.S
        _, base, bsc = part.evaluate(nodes, nodes, Q, pi, per_site=True)
        ours = np.zeros((I, S, 61))
        for i in range(I):
            un = _pin_update_list(flat, L + i)
            for k in range(60):
                part.set_pinned_states(L + i, np.full(S, k))
                _, lk, sc = part.evaluate(un, none, np.zeros((0, 61, 61)), pi, per_site=True)
                ours[i, :, k] = lk / base * np.exp(-(sc - bsc) * 64 * np.log(2.0))
            part.set_pinned_states(None)
            part.evaluate(un, none, np.zeros((0, 61, 61)), pi)   # restore the path (forced recompute, as the reference does)
        ll = part.evaluate(nodes[:0], none, np.zeros((0, 61, 61)), pi)
    ours[:, :, 60] = 1.0 - ours[:, :, :60].sum(2)
    ref = fx["support"].reshape(I, S, 61)
    used = set()
    for i in range(I):
        match = [r for r in range(I) if r not in used and np.allclose(ours[i], ref[r], rtol=1e-9, atol=1e-12)]
        assert match, (kernel, i)
        used.add(match[0])


@pytest.mark.parametrize("name", ["codon_ambig", "nuc_ambig", "codon_deep"])
def test_pinned_states_match_oracle(name):
    """Pinned leaves and pinned internal nodes (4-state and MFMA paths, ambiguity codes, deep trees with
    rescaling) against the CPU restatement."""
    from hyphy_amd import tree
    from oracle import oracle
    fx = common.load(name)
    L, D = int(fx["L"]), int(fx["D"])
    nodes = common.all_nodes(fx)
    Q = common.fixture_Q(fx)
    pi = fx["root_freqs"]
    flat = tree.flat_from_parents(fx["flat_parents"], L)
    op = oracle.OraclePartition(D, fx["flat_parents"], L, fx["leaf_codes"], fx["ambig"], fx["pattern_freq"])
    op.set_P(nodes, oracle.expm(Q, D > 4))
    rng = np.random.default_rng(3)
    with _mk(fx) as part:
        S = part.S
        part.evaluate(nodes, nodes, Q, pi)
        op.site_block(nodes, pi)   # (per-site mode throughout: the oracle's site corrections are cumulative state)
        for code in (1, L - 1, L + 0, L + flat.I // 2, L + flat.I - 2):
            states = rng.integers(0, D, size=S)
            un = _pin_update_list(flat, int(code))
            part.set_pinned_states(code, states)
            op.set_branch(code, states)
            _, lk, sc = part.evaluate(un, np.zeros(0, dtype=np.int64), np.zeros((0, D, D)), pi, per_site=True)
            rl, rs = op.site_block(un, pi)
            part.set_pinned_states(None)
            op.set_branch(None)
            ok = rl > 0
            assert np.array_equal(lk > 0, ok), (name, code)
            assert np.allclose(np.log(lk[ok]) - sc[ok] * 64 * np.log(2.0), np.log(rl[ok]) - rs[ok] * 64 * np.log(2.0),
                               rtol=1e-10, atol=1e-9), (name, code)
            part.evaluate(un, np.zeros(0, dtype=np.int64), np.zeros((0, D, D)), pi)
            op.site_block(un, pi)


@pytest.mark.parametrize("kernel,chain_m", [(None, None), ("1", "2"), ("2", "2")])
def test_categories_match_reference(kernel, chain_m, monkeypatch):
    """Three rate classes batched into one launch (a grid row per class) and mixed on the device: the library's own kernel
    choice, and the two chain kernels forced (per-class arrival counters, deposits and exponents: wave per tile / row-split
    workgroups), first pass and steady state."""
    if kernel:
        monkeypatch.setenv("HYPHY_HIP_KERNEL", kernel)
        monkeypatch.setenv("HYPHY_HIP_CHAIN_M", chain_m)
        monkeypatch.setenv("HYPHY_HIP_POISON", "1")
    fx = common.load("codon_cat3")
    C = len(fx["cat_weights"])
    nodes = common.all_nodes(fx)
    Q = np.stack([common.fixture_Q(fx, float(v)) for v in fx["cat_values"]])
    ref = float(fx["logl"])
    with _mk(fx, C) as part:
        for rep in range(3):
            ll, lik, sc = part.evaluate_categories(nodes, nodes, Q, fx["cat_weights"], fx["root_freqs"], per_site=True)
            assert abs(ll - ref) <= RTOL * abs(ref), (kernel, rep, ll, ref, part.schedule_info())
            site = (np.log(lik) - sc * 64 * np.log(2.0))[fx["site_to_pattern"]]
            assert np.max(np.abs(site - fx["site_logl"]) / np.abs(fx["site_logl"])) < RTOL, (kernel, rep)


def test_categories_built_on_device_match_host_q():
    """Device-side Q for all classes (class-major coefficients) + batched pruning + device mixing."""
    import bench
    from hyphy_amd import models
    fx = common.load("codon_cat3")
    C = len(fx["cat_weights"])
    nodes = common.all_nodes(fx)
    tb = np.asarray(fx["t"], dtype=np.float64)
    omega = float(fx["omega"])
    T = np.zeros((2, 61, 61))
    rv = dict(zip(common.REV_KEYS, (float(x) for x in fx["rev"])), AG=1.0)
    for (i, j, name, ns, pf) in models.mg94rev_template(fx["pos_freqs"]):
        T[1 if ns else 0, i, j] = rv[name] * pf
    co = np.concatenate([np.stack([tb * v, tb * v * omega], axis=1) for v in fx["cat_values"]]).copy()
    with _mk(fx, C) as part:
        part.set_q_templates(T)
        step = part.prepare_built_categories_step(nodes, nodes, fx["cat_weights"], fx["root_freqs"], co)
        ll = step()
        ll_again = step()
    ref = float(fx["logl"])
    assert abs(ll - ref) <= RTOL * abs(ref)
    assert ll_again == ll


def test_categories_built_sites_full_and_partial_passes_with_template_switches():
    """What the adapter's category hook does (r06, INTEGRATION.md "rate classes"): the classes' templates side by side (C * K of
    them, class c's coefficients at its own columns), hyphy_hip_evaluate_categories_built_sites for full passes AND for one-branch
    updates (an optimiser's line search), with one-class calls in between that switch the template set to K and back — mixed
    per-pattern values and exponents against the dense-matrix entry point at every step."""
    from hyphy_amd import models, tree
    fx = common.load("codon_cat3")
    C = len(fx["cat_weights"])
    L = int(fx["L"])
    flat = tree.flat_from_parents(fx["flat_parents"], L)
    nodes = common.all_nodes(fx)
    B = len(nodes)
    tb = np.asarray(fx["t"], dtype=np.float64).copy()
    omega = float(fx["omega"])
    rv = dict(zip(common.REV_KEYS, (float(x) for x in fx["rev"])), AG=1.0)
    T1 = np.zeros((1, 61, 61))            # K = 1 per class: Q_b^(c) = t_b * (class value) * (syn + omega * nonsyn)
    for (i, j, name, ns, pf) in models.mg94rev_template(fx["pos_freqs"]):
        T1[0, i, j] = rv[name] * pf * (omega if ns else 1.0)
    vals = [float(v) for v in fx["cat_values"]]
    Tst = np.concatenate([T1 * v for v in vals])            # [C][61][61]: class c's template carries its rate

    def dense(tbv):
        Q = np.zeros((C, B, 61, 61))
        for c in range(C):
            for b in range(B):
                M = Tst[c] * tbv[b]
                np.fill_diagonal(M, 0.0)
                np.fill_diagonal(M, -M.sum(1))
                Q[c, b] = M
        return Q

    def rows(tbv, which):
        co = np.zeros((C * len(which), C))
        for c in range(C):
            for k, b in enumerate(which):
                co[c * len(which) + k, c] = tbv[b]
        return co

    with _mk(fx, C) as a, _mk(fx, C) as b:
        a.set_q_templates(Tst)
        ll, lik, sc = a.evaluate_categories_built_sites(nodes, nodes, fx["cat_weights"], fx["root_freqs"], rows(tb, nodes))
        ll0, lik0, sc0 = b.evaluate_categories(nodes, nodes, dense(tb), fx["cat_weights"], fx["root_freqs"], per_site=True)
        ref = float(fx["logl"])
        assert abs(ll - ref) <= RTOL * abs(ref) and abs(ll0 - ref) <= RTOL * abs(ref)
        rng = np.random.default_rng(3)
        for it, node in enumerate(rng.choice(B, size=6, replace=False)):
            tb[node] *= rng.uniform(0.4, 2.5)
            un = flat.path_update_nodes(int(node))
            if it % 2 == 1:   # a one-class call with that class's own K = 1 template set in between (then back to the stacked set)
                a.set_q_templates(Tst[1:2])
                step = a.prepare_built_step(nodes, nodes, fx["root_freqs"], np.ascontiguousarray(tb[:, None]), cat=1)
                step()
                a.set_q_templates(Tst)
                ll, lik, sc = a.evaluate_categories_built_sites(nodes, nodes, fx["cat_weights"], fx["root_freqs"], rows(tb, nodes))
                ll0, lik0, sc0 = b.evaluate_categories(nodes, nodes, dense(tb), fx["cat_weights"], fx["root_freqs"], per_site=True)
            else:
                ll, lik, sc = a.evaluate_categories_built_sites(un, [node], fx["cat_weights"], fx["root_freqs"], rows(tb, [node]))
                ll0, lik0, sc0 = b.evaluate_categories(un, [node], dense(tb)[:, [node]], fx["cat_weights"], fx["root_freqs"], per_site=True)
            assert abs(ll - ll0) <= RTOL * abs(ll0), (it, ll, ll0)
            va, vb = np.log(lik) - sc * 64 * np.log(2.0), np.log(lik0) - sc0 * 64 * np.log(2.0)
            assert np.max(np.abs(va - vb) / np.abs(vb)) < RTOL, it


@pytest.mark.parametrize("dense_batch", [False, True])
def test_batched_and_one_class_evaluations_interleaved_with_branch_caches(dense_batch):
    """The call pattern of the adapter under Optimize with rate classes (r06): full passes of all classes in ONE batched evaluation
    (template rows or dense matrices), one-branch updates batched or one class at a time, a branch cache per class built behind a
    one-class pass, line searches through the caches — every step's per-pattern mixture against a partition that only ever
    evaluates one class at a time.  (Found with this sequence: a batched evaluation writes its C * n_q slot numbers across the regions
    of the device slot table that the classes use one at a time; a one-class call whose own list had not changed kept the batch's
    numbers — exponentials landed beyond the last class's images.)"""
    from hyphy_amd import hip, models, tree
    fx = common.load("codon_cat3")
    C = len(fx["cat_weights"])
    L = int(fx["L"])
    flat = tree.flat_from_parents(fx["flat_parents"], L)
    nodes = common.all_nodes(fx)
    B = len(nodes)
    tb = np.asarray(fx["t"], dtype=np.float64).copy()
    omega = float(fx["omega"])
    rv = dict(zip(common.REV_KEYS, (float(x) for x in fx["rev"])), AG=1.0)
    T1 = np.zeros((1, 61, 61))
    for (i, j, name, ns, pf) in models.mg94rev_template(fx["pos_freqs"]):
        T1[0, i, j] = rv[name] * pf * (omega if ns else 1.0)
    Tst = np.concatenate([T1 * float(v) for v in fx["cat_values"]])
    w, pi = np.asarray(fx["cat_weights"]), fx["root_freqs"]

    def dense_c(tbv, c, which):
        Q = np.zeros((len(which), 61, 61))
        for k, b in enumerate(which):
            M = Tst[c] * tbv[b]
            np.fill_diagonal(M, 0.0)
            np.fill_diagonal(M, -M.sum(1))
            Q[k] = M
        return Q

    def rows(tbv, which):
        co = np.zeros((C * len(which), C))
        for c in range(C):
            for k, b in enumerate(which):
                co[c * len(which) + k, c] = tbv[b]
        return co

    def mix(per):
        m = np.min([p[1] for p in per], axis=0)
        out = sum(w[c] * lik * np.exp2(-64.0 * (sc - m)) for c, (lik, sc) in enumerate(per))
        return np.log(out) - m * 64 * np.log(2.0)

    def site(lik, sc):
        return np.log(lik) - sc * 64 * np.log(2.0)

    with _mk(fx, C) as part, _mk(fx, C) as refp:
        for c in range(C):   # first evaluations one class at a time, like the adapter's
            part.evaluate(nodes, nodes, dense_c(tb, c, nodes), pi, cat=c)
        part.set_q_templates(Tst)
        rng = np.random.default_rng(5)
        cached_node = 0
        for it in range(32):
            kind = it % 4
            if kind == 0:      # batched full pass
                if dense_batch:
                    _, lik, sc = part.evaluate_categories(nodes, nodes, np.stack([dense_c(tb, c, nodes) for c in range(C)]), w, pi, per_site=True)
                else:
                    _, lik, sc = part.evaluate_categories_built_sites(nodes, nodes, w, pi, rows(tb, nodes))
                got = site(lik, sc)
            elif kind == 1:    # one-branch update: batched / one class at a time in turn
                node = int(rng.integers(0, B))
                tb[node] *= rng.uniform(0.5, 2.0)
                un = flat.path_update_nodes(node)
                if it % 8 == 1:
                    _, lik, sc = part.evaluate_categories_built_sites(un, [node], w, pi, rows(tb, [node]))
                    got = site(lik, sc)
                else:
                    got = mix([part.evaluate(un, [node], dense_c(tb, c, [node]), pi, cat=c, per_site=True)[1:] for c in range(C)])
            elif kind == 2:    # one class at a time + a branch cache per class (the policy's "build the cache after this pass")
                node = int(rng.integers(0, B))
                tb[node] *= rng.uniform(0.5, 2.0)
                un = flat.path_update_nodes(node)
                per = []
                for c in range(C):
                    per.append(part.evaluate(un, [node], dense_c(tb, c, [node]), pi, cat=c, per_site=True)[1:])
                    part.branch_cache_build(node, cat=c)
                got = mix(per)
                cached_node = node
            else:              # a line-search step through the caches, then the host's ordinary re-evaluation of that branch
                tb[cached_node] *= rng.uniform(0.7, 1.4)
                got = mix([part.branch_cache_evaluate(cached_node, dense_c(tb, c, [cached_node])[0], cat=c, per_site=True)[1:] for c in range(C)])
                un = flat.path_update_nodes(cached_node)
                for c in range(C):
                    part.evaluate(un, [cached_node], dense_c(tb, c, [cached_node]), pi, cat=c)
            want = mix([refp.evaluate(nodes, nodes, dense_c(tb, c, nodes), pi, cat=c, per_site=True)[1:] for c in range(C)])
            assert np.max(np.abs(got - want) / np.abs(want)) < RTOL, (it, kind)


def test_sharded_partition_equals_single(monkeypatch):
    """device_count > 1 semantics (pattern shards + Neumaier combine) exercised on one GPU by
    mapping every shard to device 0."""
    fx = common.load("nuc_small")
    Q = common.fixture_Q(fx)
    nodes = common.all_nodes(fx)
    monkeypatch.setenv("HYPHY_HIP_FORCE_SHARDS", "3")
    with _mk(fx) as part:
        ll, lik, sc = part.evaluate(nodes, nodes, Q, fx["root_freqs"], per_site=True)
    monkeypatch.delenv("HYPHY_HIP_FORCE_SHARDS")
    ref = float(fx["logl"])
    assert abs(ll - ref) <= RTOL * abs(ref)
    site = (np.log(lik) - sc * 64 * np.log(2.0))[fx["site_to_pattern"]]
    assert np.max(np.abs(site - fx["site_logl"]) / np.abs(fx["site_logl"])) < RTOL


def test_zero_likelihood_site_gives_minus_infinity():
    """tree_evaluator.cpp:4094-4112: a pattern with likelihood 0 -> -INFINITY."""
    fx = common.load("nuc_small")
    nodes = common.all_nodes(fx)
    P = np.tile(np.eye(4), (len(nodes), 1, 1))  # identity transitions: any variable site is impossible
    with _mk(fx) as part:
        ll = part.evaluate(nodes, nodes, P, fx["root_freqs"], q_is_probability=True)
    assert ll == -np.inf


def test_error_paths():
    hip = _hip()
    fx = common.load("nuc_small")
    with pytest.raises(hip.HipUnsupported):
        hip.HipPartition(100, fx["flat_parents"], int(fx["L"]), fx["leaf_codes"], None, fx["pattern_freq"])
    with _mk(fx) as part:
        with pytest.raises(hip.HipError):   # first evaluation must supply every matrix
            part.evaluate(common.all_nodes(fx), [0], common.fixture_Q(fx)[:1], fx["root_freqs"])
    Q = np.full((1, 4, 4), np.nan)
    with pytest.raises(hip.HipError):
        hip.expm_batch(Q)


@pytest.mark.parametrize("script,n_cases,seed", [("stress_codon.py", 30, 1), ("stress_generic.py", 50, 1)])
def test_randomised_stress_with_poisoned_allocations(script, n_cases, seed):
    """Random trees / sizes / kernels / fragment cuts and random sequences of full passes, partial updates, pinned
    evaluations, branch-cache line searches and downloads against the oracle, in ONE process (recycled device memory)
    and with every fresh device allocation filled with 0xff bytes (HYPHY_HIP_POISON): anything read before it is
    written changes the result.  (Found: the fragment-root publish flag shared a bit with the hand-off consumer flag.)"""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, HYPHY_HIP_POISON="1")
    r = subprocess.run([sys.executable, os.path.join(here, script), str(n_cases), str(seed)], env=env,
                       capture_output=True, text=True, timeout=900)
    tail = "\n".join((r.stdout + r.stderr).strip().splitlines()[-5:])
    assert r.returncode == 0 and "checks passed" in tail, tail


# ---- per-site batched fits --------------------------------------------------------------------------
def _site_fit_case(D, taxa, sites, K, G, n_sets, seed, caterpillar=False, balanced=False, mult_hi=5.0, ambiguity=True):
    from hyphy_amd import data, tree
    rng = np.random.default_rng(seed)
    if balanced:
        counter = [0, 0]

        def bal(n):
            if n == 1:
                counter[0] += 1
                return tree.Node(name=f"T{counter[0]}")
            counter[1] += 1
            nd = tree.Node(name=f"N{counter[1]}", children=[bal(n // 2), bal(n - n // 2)])
            for c in nd.children:
                c.parent = nd
            return nd
        root = bal(taxa)
    else:
        root = tree.caterpillar_tree(taxa) if caterpillar else tree.random_tree(taxa, rng)
    flat = tree.flatten(root)
    L, B = flat.L, flat.n_branches
    base = rng.integers(0, D, size=sites)
    states = np.where(rng.random((L, sites)) < 0.3, rng.integers(0, D, size=(L, sites)), base[None, :])
    codes = states.astype(np.int64)
    ambig = np.zeros((0, D))
    if ambiguity:
        ambig = (rng.random((2, D)) < 0.5).astype(np.float64)
        ambig[:, 0] = 1.0
        ambig[1, :] = 1.0  # fully missing
        hit = rng.random(codes.shape) < 0.05
        codes[hit] = -(rng.integers(0, 2, size=int(hit.sum())) + 1)
    pi = rng.dirichlet(np.full(D, 5.0))
    T = np.zeros((K, D, D))
    for k in range(K):
        R = rng.uniform(0.1, 2.0, (D, D)) * (rng.random((D, D)) < 0.4)
        R = np.maximum(R, R.T) + np.diag(np.full(D - 1, 0.05), 1) + np.diag(np.full(D - 1, 0.05), -1)
        T[k] = R * pi[None, :]
        np.fill_diagonal(T[k], 0.0)
    T /= max(float((T.sum(0).sum(1) * pi).sum()), 1e-9)  # ~ one expected substitution per unit coefficient
    bgroup = rng.integers(0, G, size=B)
    bcoef = rng.uniform(0.01, 0.5, (B, K))
    bcoef[rng.integers(0, B)] = 0.0  # a zero-length branch
    smult = np.exp(rng.uniform(np.log(0.01), np.log(mult_hi), (n_sets, sites, G, K)))
    smult[0, :3] = 0.0          # sites whose every rate is zero
    smult[-1, 3:6, :, 0] = 0.0  # alpha = 0
    return flat, codes, ambig, pi, T, bgroup, bcoef, smult


def _site_fit_reference(D, flat, codes, ambig, pi, T, bgroup, bcoef, smult):
    """The reference's way (FEL.bf:609+): one single-site likelihood function per site — exponentiate every branch's
    own rate matrix (oracle restatement of _Matrix::Exponentiate), then prune that one pattern."""
    from oracle import oracle
    n_sets, S, G, K = smult.shape
    B = flat.n_branches
    nodes = np.arange(B, dtype=np.int64)
    out = np.zeros((n_sets, S))
    idx = np.arange(D)
    for st in range(n_sets):
        for s in range(S):
            x = smult[st, s][bgroup] * bcoef  # [B][K]
            Q = np.einsum("bk,kij->bij", x, T)
            Q[:, idx, idx] = -Q.sum(2)
            op = oracle.OraclePartition(D, flat.flat_parents, flat.L, codes[:, s:s + 1], ambig, np.ones(1, dtype=np.int64))
            op.set_P(nodes, oracle.expm(Q, True))
            out[st, s] = op.site_log_likelihoods(nodes, pi)[0]
    return out


@pytest.mark.parametrize("name,kw", [
    ("codon_random", dict(D=61, taxa=14, sites=40, K=2, G=2, n_sets=3, seed=1)),
    ("codon_caterpillar", dict(D=61, taxa=24, sites=20, K=2, G=2, n_sets=2, seed=2, caterpillar=True)),
    ("codon_balanced_spills", dict(D=61, taxa=32, sites=18, K=2, G=3, n_sets=2, seed=3, balanced=True)),
    ("protein_one_template", dict(D=20, taxa=10, sites=50, K=1, G=1, n_sets=2, seed=4)),
    ("states_48_three_templates", dict(D=48, taxa=9, sites=17, K=3, G=2, n_sets=1, seed=5)),
    ("codon_fast_sites", dict(D=61, taxa=8, sites=16, K=2, G=2, n_sets=2, seed=6, mult_hi=800.0)),
])
def test_site_fits_match_per_site_reference(name, kw):
    """hyphy_hip_site_fits_evaluate (exp(Q) applied by uniformisation, matrices never formed) against one oracle
    likelihood function per site with explicitly exponentiated matrices.  Tolerance 1e-9 relative on the site log-L."""
    hip = _hip()
    D = kw["D"]
    flat, codes, ambig, pi, T, bgroup, bcoef, smult = _site_fit_case(**kw)
    ref = _site_fit_reference(D, flat, codes, ambig, pi, T, bgroup, bcoef, smult)
    with hip.HipPartition(D, flat.flat_parents, flat.L, codes, ambig, np.ones(codes.shape[1], dtype=np.int64)) as part:
        part.set_q_templates(T)
        got = part.site_fits_evaluate(bgroup, bcoef, smult, pi)
        again = part.site_fits_evaluate(bgroup, bcoef, smult[0], pi)
    assert got.shape == ref.shape
    fin = np.isfinite(ref)
    assert np.array_equal(np.isfinite(got), fin), name
    assert np.max(np.abs(got[fin] - ref[fin]) / np.maximum(1.0, np.abs(ref[fin]))) < 1e-9, name
    assert np.array_equal(again, got[0])


def test_site_fits_with_unit_multipliers_equal_the_ordinary_evaluation():
    """All multipliers 1: every site sees the same matrices, so the result must agree with the per-site output of
    hyphy_hip_evaluate on the same partition (device against device, different kernels and algorithms)."""
    hip = _hip()
    flat, codes, ambig, pi, T, bgroup, bcoef, smult = _site_fit_case(D=61, taxa=20, sites=200, K=2, G=2, n_sets=1, seed=11)
    B = flat.n_branches
    nodes = np.arange(B, dtype=np.int64)
    Q = np.einsum("bk,kij->bij", bcoef, T)
    idx = np.arange(61)
    Q[:, idx, idx] = -Q.sum(2)
    with hip.HipPartition(61, flat.flat_parents, flat.L, codes, ambig, np.ones(200, dtype=np.int64)) as part:
        part.set_q_templates(T)
        got = part.site_fits_evaluate(bgroup, bcoef, np.ones((200, 2, 2)), pi)
        ll, lik, sc = part.evaluate(nodes, nodes, Q, pi, per_site=True)
    ref = np.log(lik) - sc * 64 * np.log(2.0)
    assert np.max(np.abs(got - ref) / np.abs(ref)) < 1e-10
    assert abs(got.sum() - ll) < 1e-9 * abs(ll)


def test_site_fits_error_paths():
    hip = _hip()
    fx = common.load("nuc_small")
    with _mk(fx) as part:
        part.set_q_templates(np.ones((1, 4, 4)))
        with pytest.raises(hip.HipUnsupported):
            part.site_fits_evaluate(np.zeros(part.B, dtype=np.int64), np.ones((part.B, 1)), np.ones((part.S, 1, 1)), fx["root_freqs"])
    flat, codes, ambig, pi, T, bgroup, bcoef, smult = _site_fit_case(D=20, taxa=5, sites=8, K=1, G=1, n_sets=1, seed=3)
    with hip.HipPartition(20, flat.flat_parents, flat.L, codes, ambig, np.ones(8, dtype=np.int64)) as part:
        with pytest.raises(hip.HipError):  # templates not set
            part.site_fits_evaluate(bgroup, bcoef, smult, pi)
        part.set_q_templates(T)
        with pytest.raises(hip.HipError):  # negative multiplier
            part.site_fits_evaluate(bgroup, bcoef, -smult, pi)
        with pytest.raises(hip.HipUnsupported):  # total rate beyond the entry point's limit (series length ~ rate)
            part.site_fits_evaluate(bgroup, bcoef, smult * 1e7, pi)


def test_fel_driver_on_the_device():
    """hyphy_amd/fel.py over the HIP entry point: alternative contains the null, optima at least as good as every
    grid point, and the fitted log-likelihoods reproduce through an independent evaluation."""
    hip = _hip()
    from hyphy_amd import fel
    flat, codes, ambig, pi, T, bgroup, bcoef, smult = _site_fit_case(D=61, taxa=10, sites=48, K=2, G=2, n_sets=1, seed=21,
                                                                     ambiguity=False)
    tested = bgroup == 0
    tested[0], tested[1] = True, False
    with hip.HipPartition(61, flat.flat_parents, flat.L, codes, ambig, np.ones(48, dtype=np.int64)) as part:
        part.set_q_templates(T)
        res = fel.fel(part, tested, bcoef[:, 0], bcoef[:, 1], pi, max_iter=200)
        group = np.where(tested, 0, 1)
        theta = np.stack([res.alpha, res.beta, res.beta_nuisance], axis=1)
        again = part.site_fits_evaluate(group, bcoef, fel._multipliers(theta, np.array([[0, 1], [0, 2]])), pi)
        grid = part.site_fits_evaluate(group, bcoef, fel._multipliers(
            np.broadcast_to(np.array([(a, b, b) for a, b in fel.START_GRID])[:, None, :], (12, 48, 3)), np.array([[0, 1], [0, 2]])), pi)
    assert np.allclose(again, res.logl_alt, rtol=0, atol=1e-9)
    assert (res.logl_alt >= grid.max(0) - 1e-9).all()
    assert (res.logl_alt >= res.logl_null - 1e-7).all()
    assert ((res.p_value >= 0) & (res.p_value <= 1)).all()


def test_site_fits_on_a_sharded_partition(monkeypatch):
    """device_count > 1 semantics (pattern shards) for the per-site entry point, all shards mapped to device 0."""
    hip = _hip()
    flat, codes, ambig, pi, T, bgroup, bcoef, smult = _site_fit_case(D=61, taxa=12, sites=75, K=2, G=2, n_sets=2, seed=31)
    freq = np.ones(75, dtype=np.int64)
    with hip.HipPartition(61, flat.flat_parents, flat.L, codes, ambig, freq) as part:
        part.set_q_templates(T)
        one = part.site_fits_evaluate(bgroup, bcoef, smult, pi)
    monkeypatch.setenv("HYPHY_HIP_FORCE_SHARDS", "3")
    with hip.HipPartition(61, flat.flat_parents, flat.L, codes, ambig, freq) as part:
        part.set_q_templates(T)
        three = part.site_fits_evaluate(bgroup, bcoef, smult, pi)
    monkeypatch.delenv("HYPHY_HIP_FORCE_SHARDS")
    assert np.array_equal(one, three)


@pytest.mark.parametrize("name,kw,n_mix", [
    ("codon_two_classes", dict(D=61, taxa=12, sites=36, K=2, G=2, n_sets=2, seed=41), 2),
    ("codon_three_classes_spills", dict(D=61, taxa=24, sites=20, K=2, G=1, n_sets=1, seed=42, balanced=True), 3),
    ("protein_two_classes", dict(D=20, taxa=9, sites=33, K=1, G=2, n_sets=1, seed=43), 2),
])
def test_site_fits_branch_site_mixture_matches_explicit_form_reference(name, kw, n_mix):
    """hyphy_hip_site_fits_evaluate_mixture against the reference's explicit-form route restated with the oracle: per site,
    P_b = sum_m w_m Exp(Q_b^(m)) (tree.cpp:3047-3090) and a one-pattern pruning pass.  1e-9 relative on the site log-L."""
    hip = _hip()
    from oracle import oracle
    D = kw["D"]
    flat, codes, ambig, pi, T, bgroup, bcoef, smult = _site_fit_case(**kw)
    rng = np.random.default_rng(kw["seed"] + 100)
    n_sets, S, G, K = smult.shape
    sm = np.exp(rng.uniform(np.log(0.02), np.log(8.0), (n_sets, S, n_mix, G, K)))
    sm[0, :2] = 0.0                       # every rate zero in every component
    sm[-1, 2:5, 0] = 0.0                  # one component without substitutions
    sw = rng.dirichlet(np.full(n_mix, 2.0), size=(n_sets, S))
    sw[0, 5] = np.eye(n_mix)[0]           # degenerate mixture
    B = flat.n_branches
    nodes = np.arange(B, dtype=np.int64)
    idx = np.arange(D)
    ref = np.zeros((n_sets, S))
    for st in range(n_sets):
        for s in range(S):
            P = np.zeros((B, D, D))
            for m in range(n_mix):
                Q = np.einsum("bk,kij->bij", sm[st, s, m][bgroup] * bcoef, T)
                Q[:, idx, idx] = -Q.sum(2)
                P += sw[st, s, m] * oracle.expm(Q, True)
            op = oracle.OraclePartition(D, flat.flat_parents, flat.L, codes[:, s:s + 1], ambig, np.ones(1, dtype=np.int64))
            op.set_P(nodes, P)
            ref[st, s] = op.site_log_likelihoods(nodes, pi)[0]
    with hip.HipPartition(D, flat.flat_parents, flat.L, codes, ambig, np.ones(S, dtype=np.int64)) as part:
        part.set_q_templates(T)
        got = part.site_fits_evaluate_mixture(bgroup, bcoef, sm, sw, pi)
        plain = part.site_fits_evaluate(bgroup, bcoef, sm[:, :, 0], pi)
        degenerate = part.site_fits_evaluate_mixture(bgroup, bcoef, sm, np.broadcast_to(np.eye(n_mix)[0], sw.shape).copy(), pi)
    fin = np.isfinite(ref)
    assert np.array_equal(np.isfinite(got), fin), name
    assert np.max(np.abs(got[fin] - ref[fin]) / np.maximum(1.0, np.abs(ref[fin]))) < 1e-9, name
    ok = np.isfinite(plain)
    assert np.array_equal(np.isfinite(degenerate), ok)
    assert np.max(np.abs(degenerate[ok] - plain[ok]) / np.maximum(1.0, np.abs(plain[ok]))) < 1e-12


def test_meme_driver_on_the_device():
    """hyphy_amd/fel.py::meme over hyphy_hip_site_fits_evaluate_mixture: nesting, parameter ranges, and the reported
    optimum reproduces through an independent evaluation."""
    hip = _hip()
    from hyphy_amd import fel
    flat, codes, ambig, pi, T, bgroup, bcoef, smult = _site_fit_case(D=61, taxa=8, sites=32, K=2, G=2, n_sets=1, seed=51,
                                                                     ambiguity=False)
    tested = bgroup == 0
    tested[0], tested[1] = True, False
    with hip.HipPartition(61, flat.flat_parents, flat.L, codes, ambig, np.ones(32, dtype=np.int64)) as part:
        part.set_q_templates(T)
        res = fel.meme(part, tested, bcoef[:, 0], bcoef[:, 1], pi, max_iter=150)
        group = np.where(tested, 0, 1)
        sm = np.empty((32, 2, 2, 2))
        sm[..., 0] = res.alpha[:, None, None]
        sm[:, 0, 0, 1], sm[:, 1, 0, 1] = res.beta_minus, res.beta_plus
        sm[:, :, 1, 1] = res.beta_nuisance[:, None]
        sw = np.stack([res.weight_minus, 1 - res.weight_minus], axis=1)
        again = part.site_fits_evaluate_mixture(group, bcoef, sm, sw, pi)
    assert np.allclose(again, res.logl_alt, rtol=0, atol=1e-8)
    assert (res.beta_minus <= res.alpha + 1e-12).all()
    assert (res.logl_alt >= res.logl_null - 1e-7).all()
    assert ((res.p_value >= 0) & (res.p_value <= 1)).all()


@pytest.mark.parametrize("name", ["codon_mix2", "codon_mix3"])
def test_explicit_form_mixture_on_the_device_matches_reference(name):
    """hyphy_hip_evaluate_mixture: every branch's transition matrix is sum_m w_m exp(Q_bm), exponentiated and mixed on
    the device (the reference's explicit-form models, tree.cpp:3047-3090) — log L and per-site log L of the reference's
    own explicit-form likelihood function, then a partial update (one branch's components change)."""
    from hyphy_amd import models, tree
    from oracle import oracle
    fx = common.load(name)
    rev = dict(zip(common.REV_KEYS, (float(x) for x in fx["rev"])))
    t = np.asarray(fx["t"], dtype=np.float64)
    nodes = common.all_nodes(fx)
    Qc = np.stack([models.mg94rev_Q_batch(t, float(om), rev, fx["pos_freqs"]) for om in fx["omegas"]], axis=1)   # [B, M, D, D]
    W = np.tile(np.asarray(fx["weights"], dtype=np.float64), (len(nodes), 1))
    with _mk(fx) as part:
        ll, lik, sc = part.evaluate_mixture(nodes, nodes, Qc, W, fx["root_freqs"], per_site=True)
        ref = float(fx["logl"])
        assert abs(ll - ref) <= RTOL * abs(ref), (ll, ref)
        site = (np.log(lik) - sc * 64 * np.log(2.0))[fx["site_to_pattern"]]
        assert np.max(np.abs(site - fx["site_logl"]) / np.abs(fx["site_logl"])) < RTOL
        # partial update: branch 3 gets other components and weights; oracle with explicitly mixed matrices
        L = int(fx["L"])
        flat = tree.flat_from_parents(fx["flat_parents"], L)
        node = 3
        Q2 = Qc[node] * np.array([0.5, 2.0, 1.5][:Qc.shape[1]])[:, None, None]
        w2 = np.asarray(fx["weights"], dtype=np.float64)[::-1].copy()
        upd = flat.path_update_nodes(node)
        got = part.evaluate_mixture(upd, np.array([node]), Q2[None], w2[None], fx["root_freqs"])
        P = np.einsum("bm,bmij->bij", W, np.stack([oracle.expm(Qc[:, m], True) for m in range(Qc.shape[1])], axis=1))
        P[node] = np.einsum("m,mij->ij", w2, oracle.expm(Q2, True))
        op = oracle.OraclePartition(int(fx["D"]), fx["flat_parents"], L, fx["leaf_codes"], fx["ambig"], fx["pattern_freq"])
        op.set_P(nodes, P)
        want = op.compute_block(nodes, fx["root_freqs"])
        assert abs(got - want) <= RTOL * abs(want), (got, want)


@pytest.mark.parametrize("name", ["codon_mix2", "codon_mix3"])
def test_explicit_form_mixture_built_on_the_device_matches_reference(name):
    """hyphy_hip_build_q + hyphy_hip_evaluate_mixture_built (r04): the same explicit-form likelihood function with every component's
    rate matrix formed on the device, Q_(b,m) = t_b T_0 + (omega_m t_b) T_1 over two uploaded templates — one coefficient row per
    (branch, component) crosses PCIe instead of a dense matrix.  Against the reference's log L and per-site log L, against the
    dense-matrix entry point (same device arithmetic behind the construction: equal to the last bits), and a partial update."""
    from hyphy_amd import models, tree
    fx = common.load(name)
    rev = dict(zip(common.REV_KEYS, (float(x) for x in fx["rev"])))
    t = np.asarray(fx["t"], dtype=np.float64)
    nodes = common.all_nodes(fx)
    omegas = np.asarray(fx["omegas"], dtype=np.float64)
    M = len(omegas)
    T = np.zeros((2, 61, 61))
    rv = dict(rev, AG=1.0)
    for (i, j, nm, ns, pf) in models.mg94rev_template(fx["pos_freqs"]):
        T[1 if ns else 0, i, j] = rv[nm] * pf
    Qc = np.stack([models.mg94rev_Q_batch(t, float(om), rev, fx["pos_freqs"]) for om in omegas], axis=1)   # [B, M, D, D]
    W = np.tile(np.asarray(fx["weights"], dtype=np.float64), (len(nodes), 1))
    coeffs = np.empty((len(nodes), M, 2))
    coeffs[:, :, 0] = t[:, None]
    coeffs[:, :, 1] = t[:, None] * omegas[None, :]
    ref = float(fx["logl"])
    with _mk(fx) as part:
        part.set_q_templates(T)
        ll, lik, sc = part.evaluate_mixture_built(nodes, nodes, coeffs, W, fx["root_freqs"], per_site=True)
        assert abs(ll - ref) <= RTOL * abs(ref), (ll, ref)
        site = (np.log(lik) - sc * 64 * np.log(2.0))[fx["site_to_pattern"]]
        assert np.max(np.abs(site - fx["site_logl"]) / np.abs(fx["site_logl"])) < RTOL
        dense = part.evaluate_mixture(nodes, nodes, Qc, W, fx["root_freqs"])
        assert abs(ll - dense) <= 1e-13 * abs(dense), (ll, dense)
        # partial update: one branch gets other coefficients and weights, through both entry points
        flat = tree.flat_from_parents(fx["flat_parents"], int(fx["L"]))
        node = 3
        scale = np.array([0.5, 2.0, 1.5][:M])
        w2 = np.asarray(fx["weights"], dtype=np.float64)[::-1].copy()
        upd = flat.path_update_nodes(node)
        want = part.evaluate_mixture(upd, np.array([node]), (Qc[node] * scale[:, None, None])[None], w2[None], fx["root_freqs"])
        part.evaluate_mixture(nodes, nodes, Qc, W, fx["root_freqs"])     # back to the start
        got = part.evaluate_mixture_built(upd, np.array([node]), (coeffs[node] * scale[:, None])[None], w2[None], fx["root_freqs"])
        assert abs(got - want) <= 1e-13 * abs(want), (got, want)


def test_rccl_allreduce_entry_points_single_rank():
    """The C-ABI's own all-reduce (hyphy_hip_comm_* / hyphy_hip_evaluate_allreduce, librccl loaded on first use): with a
    communicator of ONE rank — all this box offers, RCCL refuses two ranks on one device — the all-reduced value is the
    partition's own log-likelihood; the N-rank path differs only in the communicator (driver's multi-GPU runs)."""
    fx = common.load("codon_wide")
    Q = common.fixture_Q(fx)
    nodes = common.all_nodes(fx)
    hip = _hip()
    with _mk(fx) as part:
        ref = part.evaluate(nodes, nodes, Q, fx["root_freqs"])
        part.comm_init_rank(hip.HipPartition.comm_unique_id(), 0, 1)
        got = part.evaluate_allreduce(nodes, nodes, Q, fx["root_freqs"])
        assert got == ref or abs(got - ref) <= 1e-13 * abs(ref)
        assert abs(got - float(fx["logl"])) <= RTOL * abs(float(fx["logl"]))
        # partial update through the same entry point
        from hyphy_amd import tree
        flat = tree.flat_from_parents(fx["flat_parents"], int(fx["L"]))
        Q2 = Q[5:6] * 2.0
        upd = flat.path_update_nodes(5)
        a = part.evaluate_allreduce(upd, np.array([5]), Q2, fx["root_freqs"])
        b = part.evaluate(upd, np.array([5]), Q2, fx["root_freqs"])
        assert abs(a - b) <= 1e-13 * abs(b)


@pytest.mark.parametrize("kernel,pad", [("1", "0"), ("2", "0"), ("1", "1")])
@pytest.mark.parametrize("n_tiles", [41, 48, 7])
def test_chain_joins_within_and_across_xcds(n_tiles, kernel, pad, monkeypatch):
    """Chain schedules hand edge products between waves of ONE launch through global memory (write-through stores, drained,
    then an agent-scope arrival; the last arriver reads with L1-bypassing loads).  Workgroup b runs on XCD b mod 8 and
    the grid is (tiles, classes, sources): sibling chains of a tile sit n_tiles x (source distance) workgroups apart —
    with an odd tile count every join crosses XCDs, with a multiple of 8 every join stays inside one.  Smallest sources
    (m = 1: the most joins), poisoned allocations, 24 evaluations with changing parameters each against the oracle."""
    from hyphy_amd import data, models, tree
    from oracle import oracle
    monkeypatch.setenv("HYPHY_HIP_KERNEL", kernel)   # 1: one wave per chain, 2: a workgroup of four row-split waves per chain
    # pad = "0": the bare (tiles, classes, sources) grid, where the tile count decides which joins cross XCDs (the text above);
    # pad = "1" (the default since late r03): tile dimension padded to a multiple of 8, every join inside one XCD, surplus
    # workgroups retire at once
    monkeypatch.setenv("HYPHY_HIP_XCD_PAD", pad)
    monkeypatch.setenv("HYPHY_HIP_CHAIN_M", "1")
    monkeypatch.setenv("HYPHY_HIP_POISON", "1")
    rng = np.random.default_rng(100 + n_tiles)
    root = tree.random_tree(48, rng, trifurcating_root=True)
    flat = tree.flatten(root)
    S = n_tiles * 16 - 3
    states = rng.integers(0, 61, size=(flat.L, S))
    base = rng.integers(0, 61, size=S)
    states = np.where(rng.random((flat.L, S)) < 0.25, states, base[None, :])
    pd = data.from_states(states, 61, compress_patterns=False)
    pf = np.array([[0.3, 0.2, 0.25, 0.25], [0.2, 0.3, 0.3, 0.2], [0.25, 0.25, 0.2, 0.3]])
    rev = dict(AC=0.5, AT=0.4, CG=0.4, CT=1.2, GT=0.4)
    pi = models.f3x4_codon_freqs(pf)
    B = flat.n_branches
    nodes = np.arange(B, dtype=np.int64)
    tb = rng.uniform(0.01, 0.4, B)
    op = oracle.OraclePartition(61, flat.flat_parents, flat.L, pd.leaf_codes, None, pd.pattern_freq)
    hip = _hip()
    with hip.HipPartition(61, flat.flat_parents, flat.L, pd.leaf_codes, None, pd.pattern_freq) as part:
        assert part.S == S
        for k in range(24):
            Q = models.mg94rev_Q_batch(tb * (1.0 + 0.05 * k), 0.2 + 0.1 * k, rev, pf)
            got = part.evaluate(nodes, nodes, Q, pi)
            if k % 4 == 0 or k >= 20:   # (the oracle takes ~0.1 s per evaluation here)
                op.set_P(nodes, oracle.expm(Q, True))
                want = op.compute_block(nodes, pi)
                assert abs(got - want) <= RTOL * abs(want), (n_tiles, k, got, want)
            else:
                again = part.evaluate(nodes, nodes, Q, pi)   # same inputs, different arrival orders: same bits or rounding
                assert abs(got - again) <= 1e-13 * abs(got), (n_tiles, k, got, again)


def test_sorted_patterns_are_invisible_to_the_caller(monkeypatch):
    """The library keeps its patterns sorted on the device (api.hip: sort_patterns); every per-pattern input and output of the
    C-ABI is in the CALLER's order.  Same data through a sorting and a non-sorting partition: per-site likelihoods and
    exponents, downloaded node conditionals, a pinned-states evaluation and forced pattern shards must agree entry by entry."""
    from hyphy_amd import data, models, tree
    rng = np.random.default_rng(77)
    root = tree.random_tree(24, rng, trifurcating_root=True)
    flat = tree.flatten(root)
    S = 333
    base = rng.integers(0, 61, size=S)
    states = np.where(rng.random((flat.L, S)) < 0.2, rng.integers(0, 61, size=(flat.L, S)), base[None, :])
    pd = data.from_states(states, 61, compress_patterns=False)
    pf = np.array([[0.3, 0.2, 0.25, 0.25], [0.2, 0.3, 0.3, 0.2], [0.25, 0.25, 0.2, 0.3]])
    pi = models.f3x4_codon_freqs(pf)
    B = flat.n_branches
    nodes = np.arange(B, dtype=np.int64)
    Q = models.mg94rev_Q_batch(rng.uniform(0.02, 0.3, B), 0.4, dict(AC=0.5, AT=0.4, CG=0.4, CT=1.2, GT=0.4), pf)
    pinned = rng.integers(0, 61, size=S)
    hip = _hip()
    out = {}
    for label, env in (("sorted", {"HYPHY_HIP_SORT_PATTERNS": "1"}), ("caller", {"HYPHY_HIP_SORT_PATTERNS": "0"}),
                       ("sorted_sharded", {"HYPHY_HIP_SORT_PATTERNS": "1", "HYPHY_HIP_FORCE_SHARDS": "3"})):
        for k in ("HYPHY_HIP_SORT_PATTERNS", "HYPHY_HIP_FORCE_SHARDS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with hip.HipPartition(61, flat.flat_parents, flat.L, pd.leaf_codes, None, pd.pattern_freq) as part:
            ll, lik, sc = part.evaluate(nodes, nodes, Q, pi, per_site=True)
            cache, counts = part.download_partials()
            part.set_pinned_states(flat.L + 3, pinned)
            llp, likp, scp = part.evaluate(nodes, np.zeros(0, dtype=np.int64), np.zeros((0, 61, 61)), pi, per_site=True)
            part.set_pinned_states(None)
        out[label] = (ll, np.log(lik) - 64 * np.log(2.0) * sc, cache, counts, llp, np.log(np.maximum(likp, 1e-300)) - 64 * np.log(2.0) * scp)
    ref = out["caller"]
    for label in ("sorted", "sorted_sharded"):
        got = out[label]
        assert abs(got[0] - ref[0]) <= 1e-12 * abs(ref[0]), label
        assert np.allclose(got[1], ref[1], rtol=1e-12, atol=0), label          # per-site log-likelihoods, caller order
        assert np.allclose(got[2], ref[2], rtol=1e-10, atol=1e-300), label      # node conditionals [I][S][D]
        assert np.array_equal(got[3], ref[3]), label                            # exponents [I][S]
        assert abs(got[4] - ref[4]) <= 1e-12 * abs(ref[4]), label
        assert np.allclose(got[5], ref[5], rtol=1e-12, atol=0), label


@pytest.mark.parametrize("sort", ["1", "0"])
def test_exported_per_pattern_results_equal_the_copied_ones(sort, monkeypatch):
    """Per-pattern values and exponents reach a synchronous caller through site_export_kernel (caller's order, host-mapped memory,
    in front of the kernel that publishes the result record); HYPHY_HIP_SITE_EXPORT=0 keeps the two device-to-host copies and the
    host-side scatter.  Same partition data through both, full passes, a partial update, the template entry point and a pure
    re-evaluation: identical to the last bit."""
    from hyphy_amd import data, models, tree
    rng = np.random.default_rng(5)
    root = tree.random_tree(20, rng, trifurcating_root=True)
    flat = tree.flatten(root)
    S = 1777
    base = rng.integers(0, 61, size=S)
    states = np.where(rng.random((flat.L, S)) < 0.25, rng.integers(0, 61, size=(flat.L, S)), base[None, :])
    pd = data.from_states(states, 61, compress_patterns=False)
    pf = np.array([[0.3, 0.2, 0.25, 0.25], [0.2, 0.3, 0.3, 0.2], [0.25, 0.25, 0.2, 0.3]])
    pi = models.f3x4_codon_freqs(pf)
    B = flat.n_branches
    nodes = np.arange(B, dtype=np.int64)
    rev = dict(AC=0.5, AT=0.4, CG=0.4, CT=1.2, GT=0.4)
    tb = rng.uniform(0.02, 0.3, B)
    Q = models.mg94rev_Q_batch(tb, 0.4, rev, pf)
    Q2 = models.mg94rev_Q_batch(tb[:3] * 1.5, 0.4, rev, pf)
    import bench
    T, pi_b = bench.templates_for(3)
    coeffs = np.stack([tb, 0.4 * tb], axis=1)
    hip = _hip()
    monkeypatch.setenv("HYPHY_HIP_SORT_PATTERNS", sort)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("HYPHY_HIP_SITE_EXPORT", mode)
        got = []
        with hip.HipPartition(61, flat.flat_parents, flat.L, pd.leaf_codes, None, pd.pattern_freq) as part:
            got.append(part.evaluate(nodes, nodes, Q, pi, per_site=True))
            got.append(part.evaluate(nodes[:3], nodes[:3], Q2, pi, per_site=True))
            got.append(part.evaluate(nodes[:0], nodes[:0], Q[:0], pi, per_site=True))
            part.set_q_templates(T)
            part.build_q(coeffs)
            got.append(part.evaluate_built(nodes, nodes, pi_b, per_site=True))
        out[mode] = got
    for a, b in zip(out["1"], out["0"]):
        assert a[0] == b[0]
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert np.all(out["1"][0][1] > 0)


@pytest.mark.parametrize("kernel", ["1", "2"])
@pytest.mark.parametrize("shape", ["caterpillar", "long caterpillar", "random"])
def test_rerooted_schedules_match_oracle_without_reversibility(shape, kernel, monkeypatch):
    """Re-rooted schedules (api.hip: rr_path) hang the computation from the node that minimises the tree's height and walk the
    edges between the given root and that node with transposed matrices, pi folded in on the old root's edge.  Nothing about
    that needs a reversible model: random NON-reversible rate matrices and root frequencies that are not their stationary
    distribution, forced re-rooting, against the oracle — steady-state full passes (lazy persistence: the re-rooted form), new
    root frequencies alone, one path branch alone, a pinned evaluation in between (falls back to the given root) and per-site
    values."""
    from hyphy_amd import data, tree
    from oracle import oracle
    monkeypatch.setenv("HYPHY_HIP_REROOT", "1")
    monkeypatch.setenv("HYPHY_HIP_KERNEL", kernel)
    monkeypatch.setenv("HYPHY_HIP_CHAIN_M", "3")
    monkeypatch.setenv("HYPHY_HIP_POISON", "1")
    rng = np.random.default_rng(2024)
    root = (tree.caterpillar_tree(14) if shape == "caterpillar" else tree.caterpillar_tree(44) if shape == "long caterpillar"
            else tree.random_tree(40, rng, trifurcating_root=True))
    flat = tree.flatten(root)
    S, D = 16 * 9 + 5, 61
    base = rng.integers(0, D, size=S)
    states = np.where(rng.random((flat.L, S)) < 0.3, rng.integers(0, D, size=(flat.L, S)), base[None, :])
    pd = data.from_states(states, D, compress_patterns=False)
    B = flat.n_branches
    nodes = np.arange(B, dtype=np.int64)

    def random_q(scale):
        Q = rng.random((B, D, D)) * (rng.random((B, D, D)) < 0.15) * scale[:, None, None]
        for b in range(B):
            np.fill_diagonal(Q[b], 0.0)
            np.fill_diagonal(Q[b], -Q[b].sum(axis=1))
        return Q

    Q = random_q(rng.uniform(0.05, 0.4, B))
    pi = rng.random(D) + 0.1
    pi /= pi.sum()
    op = oracle.OraclePartition(D, flat.flat_parents, flat.L, pd.leaf_codes, None, pd.pattern_freq)
    op.set_P(nodes, oracle.expm(Q, True))
    hip = _hip()
    with hip.HipPartition(D, flat.flat_parents, flat.L, pd.leaf_codes, None, pd.pattern_freq) as part:
        def check(tag, q_nodes, q):
            want = op.compute_block(nodes, pi)
            got, lik, sc = part.evaluate(nodes, q_nodes, q, pi, per_site=True)
            assert abs(got - want) <= RTOL * abs(want), (shape, tag, got, want, part.schedule_info())
            return got
        check("first", nodes, Q)                       # persisting pass, given root
        for k in range(3):                             # steady state: lazy full passes -> re-rooted
            Q = random_q(rng.uniform(0.05, 0.4, B))
            op.set_P(nodes, oracle.expm(Q, True))
            check(f"full{k}", nodes, Q)
        assert "re-rooted" in part.schedule_info(), part.schedule_info()
        pi = rng.random(D) + 0.1                       # new root frequencies, no new matrix
        pi /= pi.sum()
        check("pi only", np.zeros(0, dtype=np.int64), np.zeros((0, D, D)))
        for b in (B - 1, B - 2, 0):                    # single branches (the last internal ones sit next to the given root)
            qb = random_q(rng.uniform(0.05, 0.4, B))[b:b + 1]
            Q[b] = qb[0]
            op.set_P(np.array([b]), oracle.expm(qb, True))
            check(f"branch {b}", np.array([b], dtype=np.int64), qb)
        st = rng.integers(0, D, size=S)
        part.set_pinned_states(flat.L + 1, st)
        op.set_branch(flat.L + 1, st)
        want = op.compute_block(nodes, pi)
        got = part.evaluate(nodes, np.zeros(0, dtype=np.int64), np.zeros((0, D, D)), pi)
        part.set_pinned_states(None)
        op.set_branch(None)
        if np.isfinite(want):
            assert abs(got - want) <= RTOL * abs(want), (shape, "pinned", got, want)
        check("after pin", np.zeros(0, dtype=np.int64), np.zeros((0, D, D)))


@pytest.mark.parametrize("kernel,n_tiles", [("0", 37), ("2", 131), ("2", 700), ("0", 513), ("1", 131)])
def test_fused_final_combine_equals_the_reduction_kernel(kernel, n_tiles, monkeypatch):
    """r03: the pruning launch sums the per-tile partial sums itself (prune.hip: publish_partial — the last root-finalising
    wave of the launch does what wg_reduce_kernel does in a launch of its own; HYPHY_HIP_FUSED_REDUCE=0 restores the separate
    kernel; the row-split kernels have the fused instantiation, the wave-per-tile kernel — case "1" — keeps the separate
    kernel whatever the variable says).  Tile counts below one load batch (37, 131), across several (700) and odd just past a batch boundary (513):
    the fused sum must equal the per-site values summed on the host and the separate kernel's result to rounding, must not
    depend on which wave arrives last (repeated evaluations: fixed summation order -> the chain joins' order is the only
    run-to-run freedom, 1e-13), and -inf / the scaler sum must come through."""
    from hyphy_amd import data, models, tree
    monkeypatch.setenv("HYPHY_HIP_KERNEL", kernel)
    if kernel != "0":
        monkeypatch.setenv("HYPHY_HIP_CHAIN_M", "3")
    monkeypatch.setenv("HYPHY_HIP_POISON", "1")
    rng = np.random.default_rng(7000 + n_tiles)
    root = tree.random_tree(24, rng, trifurcating_root=True)
    flat = tree.flatten(root)
    S = n_tiles * 16 - 5
    states = rng.integers(0, 61, size=(flat.L, S))
    base = rng.integers(0, 61, size=S)
    states = np.where(rng.random((flat.L, S)) < 0.3, states, base[None, :])
    pd = data.from_states(states, 61, compress_patterns=False)
    pf = np.array([[0.3, 0.2, 0.25, 0.25], [0.2, 0.3, 0.3, 0.2], [0.25, 0.25, 0.2, 0.3]])
    rev = dict(AC=0.5, AT=0.4, CG=0.4, CT=1.2, GT=0.4)
    pi = models.f3x4_codon_freqs(pf)
    B = flat.n_branches
    nodes = np.arange(B, dtype=np.int64)
    tb = rng.uniform(0.01, 0.4, B)
    freq = rng.integers(1, 4, size=S).astype(np.float64)
    hip = _hip()
    with hip.HipPartition(61, flat.flat_parents, flat.L, pd.leaf_codes, None, freq) as part:
        for k in range(6):
            Q = models.mg94rev_Q_batch(tb * (1.0 + 0.3 * k), 0.2 + 0.2 * k, rev, pf)
            monkeypatch.setenv("HYPHY_HIP_FUSED_REDUCE", "1")
            fused = part.evaluate(nodes, nodes, Q, pi)
            fused2, lik, sc = part.evaluate(nodes, nodes, Q, pi, per_site=True)
            monkeypatch.setenv("HYPHY_HIP_FUSED_REDUCE", "0")
            plain = part.evaluate(nodes, nodes, Q, pi)
            host = float(np.sum(freq * (np.log(lik) - sc * 64 * np.log(2.0))))
            assert np.isfinite(fused)
            assert abs(fused - fused2) <= 1e-13 * abs(fused), (k, fused, fused2)
            assert abs(fused - plain) <= 1e-13 * abs(plain), (k, fused, plain)
            assert abs(fused - host) <= 1e-12 * abs(host), (k, fused, host)
        monkeypatch.setenv("HYPHY_HIP_FUSED_REDUCE", "1")
        P = np.tile(np.eye(61), (B, 1, 1))   # identity transitions: any variable pattern is impossible
        assert part.evaluate(nodes, nodes, P, pi, q_is_probability=True) == -np.inf
        Q = models.mg94rev_Q_batch(tb, 0.5, rev, pf)
        a = part.evaluate(nodes, nodes, Q, pi)   # (and the arrival counter was left at zero by the -inf launch)
        monkeypatch.setenv("HYPHY_HIP_FUSED_REDUCE", "0")
        b = part.evaluate(nodes, nodes, Q, pi)
        assert np.isfinite(a) and abs(a - b) <= 1e-13 * abs(b)


@pytest.mark.parametrize("name,min_called", [("ref_fel_12x60", 5), ("ref_fel_10x48_all", 2)])
def test_fel_driver_matches_the_reference_fel(name, min_called):
    """hyphy_amd/fel.py (lockstep Nelder-Mead over hyphy_hip_site_fits_evaluate) against the reference's OWN analysis:
    tests/golden/ref_fel_12x60.npz holds the per-site table of the unmodified FEL.bf run by the unmodified binary
    (`python -m oracle.make_golden fel`; 12 taxa x 60 codons, internal branches tested, leaves nuisance) and the global fit
    its site phase starts from.  (1) the reconstructed global model evaluates on the device to the reference's global log L
    (1e-10); (2) per site: the likelihood-ratio statistic within 0.06 and the p-value within 0.05 of the reference's (its
    own per-site Optimize stops ~0.02 short: it reports LRTs down to -0.023 and alpha = 0.02 where the surface still falls
    towards 0), the rate estimates within the flatness of the surface, and the SAME sites called at p <= 0.1, in the same
    direction."""
    _hip()
    from tests import fel_reference_check as frc
    fx, glob, res = frc.run(name=name)   # (the second fixture tests EVERY branch: no nuisance rate in the site model)
    assert abs(glob - float(fx["global_logl"])) <= 1e-10 * abs(float(fx["global_logl"]))
    ref = fx["fel_table"]
    lrt_ref = np.maximum(ref[:, 3], 0.0)
    assert np.abs(res.lrt - lrt_ref).max() <= 0.06, np.abs(res.lrt - lrt_ref).max()
    assert np.abs(res.p_value - ref[:, 4]).max() <= 0.05
    for got, want in ((res.alpha, ref[:, 0]), (res.beta, ref[:, 1])):
        assert (np.abs(got - want) <= 0.06 + 0.2 * np.abs(want)).all(), np.abs(got - want).max()
    called_ref, called = ref[:, 4] <= 0.1, res.p_value <= 0.1
    assert np.array_equal(called, called_ref) and called.sum() >= min_called
    assert np.array_equal(np.sign(res.beta - res.alpha)[called], np.sign(ref[:, 1] - ref[:, 0])[called_ref])
    invariable = (ref[:, :4] == 0).all(1)      # (sites without substitutions: FEL.bf reports zeros without fitting)
    assert invariable.sum() >= 3 and np.abs(res.lrt[invariable]).max() <= 1e-6


def test_meme_driver_matches_the_reference_meme():
    """hyphy_amd/fel.py::meme against the reference's OWN MEME.bf (tests/golden/ref_meme_12x60.npz: unmodified batch file,
    unmodified binary, `python -m oracle.make_golden fel`).  The table carries the site's log-likelihood at the reference's
    alternative optimum ("MEME LogL"), so the anchor is at the likelihood level:
     * alternative: the device's optimum is never more than 0.03 below the reference's and at most 0.2 above (two-class
       mixtures are multi-modal: either side may find the better mode); alpha within 15 %, beta+ within 10 % where the
       reference puts >= 0.9 of the weight on it;
     * LRT / p-value: within 0.25 / 0.09 wherever the reference's null fit did not stall.  MEME.bf restarts the null
       (beta+ := alpha) from alpha = 1e-4 when the alternative has alpha = 0 (MEME.bf:1432-1436); at 5 of these 60 sites its
       Nelder-Mead ends 4-9 log units below the constrained optimum (tests/test_oracle_golden.py::
       test_reference_meme_null_fit_stalls_where_alpha_is_zero shows it with the CPU oracle), the device driver finds that
       optimum, and the statistics are not comparable there: the test then only requires device LRT <= reference LRT."""
    _hip()
    from tests import fel_reference_check as frc
    fx, res = frc.run_meme()
    ref = fx["fel_table"]   # alpha, beta-, p-, beta+, p+, LRT, p-value, MEME LogL, FEL LogL
    fitted = ref[:, 7] != 0
    assert fitted.sum() >= 45
    d = (res.logl_alt - ref[:, 7])[fitted]
    assert d.min() >= -0.03 and d.max() <= 0.2, (d.min(), d.max())
    assert (np.abs(res.alpha - ref[:, 0])[fitted] <= 0.06 + 0.15 * ref[fitted, 0]).all()
    heavy = fitted & (ref[:, 4] >= 0.9)
    assert heavy.sum() >= 15 and (np.abs(res.beta_plus - ref[:, 3])[heavy] <= 0.06 + 0.1 * ref[heavy, 3]).all()
    lrt_ref = np.maximum(ref[:, 5], 0.0)
    stalled = fitted & (ref[:, 0] <= 1e-2) & (lrt_ref > res.lrt + 0.5)
    assert stalled.sum() <= 6
    assert (res.lrt[stalled] <= lrt_ref[stalled]).all()
    ok = fitted & ~stalled
    assert np.abs(res.lrt - lrt_ref)[ok].max() <= 0.25, np.abs(res.lrt - lrt_ref)[ok].max()
    assert np.abs(res.p_value - ref[:, 6])[ok].max() <= 0.09
    assert np.abs(res.lrt[~fitted]).max() <= 1e-6      # (sites without substitutions: no test in either)


def test_busted_fit_of_the_reference_evaluates_on_the_device():
    """The unconstrained model of the reference's OWN BUSTED.bf (unmodified batch file + binary, `python -m
    oracle.make_golden busted`: 16 taxa x 150 codons simulated with site classes; test and background branches each with three
    omegas and stick-breaking weights) through hyphy_hip_evaluate_mixture at the reference's MLEs: its log L to 1e-10, and the
    per-site values against the oracle's explicitly mixed matrices."""
    from oracle import oracle
    fx = common.load("ref_busted_16x150")
    Qc, W = common.busted_components(fx)
    nodes = common.all_nodes(fx)
    with _mk(fx) as part:
        ll, lik, sc = part.evaluate_mixture(nodes, nodes, Qc, W, fx["root_freqs"], per_site=True)
    ref = float(fx["logl"])
    assert abs(ll - ref) <= RTOL * abs(ref), (ll, ref)
    P = sum(W[:, k, None, None] * oracle.expm(Qc[:, k], True) for k in range(3))
    op = oracle.OraclePartition(61, fx["flat_parents"], int(fx["L"]), fx["leaf_codes"], fx["ambig"], fx["pattern_freq"])
    op.set_P(nodes, P)
    want = op.site_log_likelihoods(nodes, fx["root_freqs"])
    got = np.log(lik) - sc * 64 * np.log(2.0)
    assert np.max(np.abs(got - want) / np.abs(want)) < RTOL
    # r04: the same MLEs through the BUILT path — no dense component matrix crosses PCIe: two templates (synonymous and
    # non-synonymous part of MG94xREV), the row of component k of branch b is (t_b, omega_k t_b) with the test or the background
    # distribution's omega_k (hyphy_hip_build_q + hyphy_hip_evaluate_mixture_built)
    from hyphy_amd import models
    rev = dict(zip(common.REV_KEYS, (float(x) for x in fx["rev"])), AG=1.0)
    T = np.zeros((2, 61, 61))
    for (i, j, nm, ns, pf) in models.mg94rev_template(fx["pos_freqs"]):
        T[1 if ns else 0, i, j] = rev[nm] * pf
    t = np.asarray(fx["t"], dtype=np.float64)
    om = np.where(np.asarray(fx["tested"], dtype=bool)[:, None], np.asarray(fx["omega_test"])[None, :], np.asarray(fx["omega_background"])[None, :])
    coeffs = np.stack([np.repeat(t[:, None], 3, axis=1), om * t[:, None]], axis=2)   # [B, 3, 2]
    with _mk(fx) as part:
        part.set_q_templates(T)
        ll2, lik2, sc2 = part.evaluate_mixture_built(nodes, nodes, coeffs, W, fx["root_freqs"], per_site=True)
    assert abs(ll2 - ref) <= RTOL * abs(ref), (ll2, ref)
    got2 = np.log(lik2) - sc2 * 64 * np.log(2.0)
    assert np.max(np.abs(got2 - want) / np.abs(want)) < RTOL


def test_rate_classes_spread_mode_on_the_device_single_rank():
    """hyphy_amd/dist.py::evaluate_classes_spread (SURVEY 8e-iii second form: class c on rank c mod world, one all-gather of
    the per-site rows, every rank mixes) with a world of one and the tensors on the GPU: the per-class, per-site outputs of
    hyphy_hip_evaluate(cat = c) mixed by torch ops must equal the library's own batched category evaluation and the
    reference's log L.  (World sizes 2 and 3: tests/test_distributed_cpu.py over gloo.)"""
    import torch
    from hyphy_amd import dist as hdist
    fx = common.load("codon_cat3")
    C = len(fx["cat_weights"])
    nodes = common.all_nodes(fx)
    Q = np.stack([common.fixture_Q(fx, float(v)) for v in fx["cat_values"]])
    ref = float(fx["logl"])
    with _mk(fx, C) as part:
        def evaluate_class(c):
            return part.evaluate(nodes, nodes, Q[c], fx["root_freqs"], cat=c, per_site=True)[1:]
        ll = hdist.evaluate_classes_spread(evaluate_class, fx["cat_weights"], fx["pattern_freq"], 0, 1, device="cuda")
        batched = part.evaluate_categories(nodes, nodes, Q, fx["cat_weights"], fx["root_freqs"])
    assert abs(ll - ref) <= RTOL * abs(ref), (ll, ref)
    assert abs(ll - batched) <= 1e-12 * abs(batched), (ll, batched)
    assert torch.cuda.is_available()


def test_expm_degree_follows_the_norm_without_losing_accuracy():
    """expm64_kernel picks the Taylor degree from the scaled infinity norm (12 / 9 / 6 for <= 1/4, 0.11, 1/64: one
    Paterson-Stockmeyer block less each).  Codon rate matrices scaled to sit just below and just above every threshold, and the
    headline's own (||Q t|| = 0.10): against the restatement of the reference's Taylor-to-convergence exponential, 2e-15
    absolute on every entry and row sums of 1 to 1e-14 — the same as the fixed degree 12 gives."""
    from hyphy_amd import models
    from oracle import oracle
    hip = _hip()
    pf = np.array([[0.3, 0.2, 0.25, 0.25], [0.2, 0.3, 0.3, 0.2], [0.25, 0.25, 0.2, 0.3]])
    rev = dict(AC=0.5, AT=0.4, CG=0.4, CT=1.2, GT=0.4)
    Q0 = models.mg94rev_Q(1.0, 0.7, rev, pf)
    n0 = np.abs(Q0).sum(1).max()
    norms = [0.9 / 64, 1.0 / 64 * 0.9999, 1.1 / 64, 0.10, 0.1099, 0.1101, 0.12, 0.2499, 0.2501, 0.4, 1e-6]
    Q = np.stack([Q0 * (nm / n0) for nm in norms])
    P = hip.expm_batch(Q)
    Po = oracle.expm(Q, True)
    assert np.max(np.abs(P - Po)) < 2e-15, np.max(np.abs(P - Po), axis=(1, 2))
    assert np.max(np.abs(P.sum(2) - 1.0)) < 1e-14
    # the 4-state exponential (expm4.h, one thread per matrix) follows the same rule on sqrt(||X||_1 ||X||_inf)
    Q4 = models.nuc_rev_Q(1.0, dict(AC=0.5, AT=0.4, CG=0.4, CT=1.2, GT=0.4), np.array([0.35, 0.15, 0.2, 0.3]))
    n4 = np.sqrt(np.abs(Q4).sum(1).max() * np.abs(Q4).sum(0).max())
    Qs = np.stack([Q4 * (nm / n4) for nm in norms + [3.0, 40.0]])
    P4 = hip.expm_batch(Qs)
    P4o = oracle.expm(Qs, False)
    err4 = np.max(np.abs(P4 - P4o), axis=(1, 2))
    assert err4[:len(norms)].max() < 2e-15, err4            # no squarings (|| <= 1/4) or one (0.4)
    assert err4[len(norms):].max() < 5e-14, err4            # 5 and 8 squarings: the reference goldens' tolerance
    assert np.max(np.abs(P4.sum(2) - 1.0)) < 1e-14


@pytest.mark.parametrize("lp,fold", [("0", "0"), ("1", "0"), ("1", "1"), ("0", "1")])
@pytest.mark.parametrize("name", ["nuc_small", "nuc_ambig", "nuc_wide"])
def test_four_state_kernel_with_and_without_the_lds_schedule(name, lp, fold, monkeypatch):
    """prune_nuc2_kernel fetches schedule words and internal-edge matrices through scalar loads (large shards) or from an LDS
    copy (LP: shards of at most two workgroups per CU, the default at these sizes); HYPHY_HIP_NUC_LP forces either.  Both
    against the reference's log L and per-site values."""
    monkeypatch.setenv("HYPHY_HIP_NUC_LP", lp)
    monkeypatch.setenv("HYPHY_HIP_NUC_FOLD", fold)   # 1: this evaluation's 4 x 4 exponentials computed inside the pruning launch
    fx = common.load(name)
    Q = common.fixture_Q(fx)
    nodes = common.all_nodes(fx)
    with _mk(fx) as part:
        assert part.prune_kernel_name() == "prune_nuc2_kernel"
        ll, lik, sc = part.evaluate(nodes, nodes, Q, fx["root_freqs"], per_site=True)
        again = part.evaluate(nodes, nodes, Q, fx["root_freqs"])      # (steady state: lazy persistence)
        # the LDS-schedule build also carries the fused final combine (last workgroup sums the partials): against the
        # separate reduction kernel, and -inf must come through it
        monkeypatch.setenv("HYPHY_HIP_FUSED_REDUCE", "0")
        plain = part.evaluate(nodes, nodes, Q, fx["root_freqs"])
        monkeypatch.setenv("HYPHY_HIP_FUSED_REDUCE", "1")
        P = np.tile(np.eye(4), (len(nodes), 1, 1))
        assert part.evaluate(nodes, nodes, P, fx["root_freqs"], q_is_probability=True) == -np.inf
        after = part.evaluate(nodes, nodes, Q, fx["root_freqs"])
    ref = float(fx["logl"])
    assert abs(ll - ref) <= RTOL * abs(ref), (ll, ref)
    assert again == ll
    assert abs(plain - ll) <= 1e-13 * abs(ll) and after == ll
    site = (np.log(lik) - sc * 64 * np.log(2.0))[fx["site_to_pattern"]]
    assert np.max(np.abs(site - fx["site_logl"]) / np.abs(fx["site_logl"])) < RTOL


def test_site_fits_reproduce_the_reference_fubar_grid():
    """The grid phase of the reference's OWN FUBAR.bf (unmodified batch file + binary, `python -m oracle.make_golden fubar`)
    evaluates every site of the alignment at every (alpha, beta) of a 10 x 10 rate grid; its cache file keeps the matrix
    (softmax columns + log normalisers).  hyphy_hip_site_fits_evaluate with one parameter set per grid point — the matrices
    never formed, exp(Q) applied by uniformisation — must return the same [grid point][site] log-likelihoods: 1e-9 relative
    wherever the reference's softmax did not underflow (5 489 of 6 000 entries), rates from 0 to 50 x the branch factors."""
    hip = _hip()
    fx = common.load("ref_fubar_12x60")
    T, group, coeffs, mult, codes, want = common.fubar_site_fit_args(fx)
    S = codes.shape[1]
    with hip.HipPartition(61, fx["flat_parents"], int(fx["L"]), codes, None, np.ones(S, dtype=np.int64)) as part:
        part.set_q_templates(T)
        got = part.site_fits_evaluate(group, coeffs, mult, fx["root_freqs"])
    assert got.shape == want.shape
    fin = np.isfinite(want)
    assert fin.sum() > 5000
    assert np.max(np.abs(got[fin] - want[fin]) / np.abs(want[fin])) < 1e-9
    # (and the conditionals FUBAR's inference consumes: per-site softmax over the grid)
    g2 = np.where(np.isfinite(got), got, -np.inf)     # (alpha = beta = 0 at a variable site: likelihood 0)
    mx = g2.max(0, keepdims=True)
    cond = np.exp(g2 - mx)
    cond /= cond.sum(0, keepdims=True)
    assert np.max(np.abs(cond - fx["conditionals"])) < 1e-9


def test_ordinary_per_site_evaluation_reproduces_the_reference_fubar_grid():
    """The same FUBAR matrix the cheap way: a grid point gives every SITE the same rates, so one ordinary evaluation with per-site
    outputs (hyphy_hip_evaluate: 21 exponentials + one pruning pass) yields a whole row of it — what an adapter-side FUBAR would
    call 100-400 times; the per-site batched kernel is for analyses whose sites carry DIFFERENT rates (FEL, MEME).  Six grid
    points, every site, 1e-9 relative."""
    _hip()
    fx = common.load("ref_fubar_12x60")
    T, group, coeffs, mult, codes, want = common.fubar_site_fit_args(fx)
    S, B = codes.shape[1], coeffs.shape[0]
    nodes = np.arange(B, dtype=np.int64)
    idx = np.arange(61)
    with _hip().HipPartition(61, fx["flat_parents"], int(fx["L"]), codes, None, np.ones(S, dtype=np.int64)) as part:
        for g in (11, 37, 55, 64, 99, 9):
            a, b = fx["grid"][g]
            Q = a * coeffs[:, 0, None, None] * T[0][None] + b * coeffs[:, 1, None, None] * T[1][None]
            Q[:, idx, idx] = 0.0
            Q[:, idx, idx] = -Q.sum(2)
            ll, lik, sc = part.evaluate(nodes, nodes, Q, fx["root_freqs"], per_site=True)
            got = np.log(lik) - sc * 64 * np.log(2.0)
            fin = np.isfinite(want[g])
            assert fin.sum() >= 40, (g, fin.sum())
            assert np.max(np.abs(got[fin] - want[g][fin]) / np.abs(want[g][fin])) < 1e-9, g
            assert abs(ll - got.sum()) <= 1e-10 * abs(ll)


def test_recycled_allocations_carry_nothing_over():
    """The library recycles the device / pinned blocks and the stream of a destroyed partition for the next one of the same
    shape (pool.hip: what FEL's one-likelihood-function-per-site life cycle needs).  Partitions of ONE shape and DIFFERENT data
    created and destroyed back to back — log-L and per-site values of each against the oracle, with a partial update in
    between, so that persisted nodes, arrival counters or the result record of a predecessor would show."""
    from hyphy_amd import data, models
    from oracle import oracle
    pf = np.array([[0.3, 0.2, 0.25, 0.25], [0.2, 0.3, 0.3, 0.2], [0.25, 0.25, 0.2, 0.3]])
    rev = dict(AC=0.5, AT=0.4, CG=0.4, CT=1.2, GT=0.4)
    pi = models.f3x4_codon_freqs(pf)
    hip = _hip()
    seen = set()
    for rep in range(6):
        syn = data.evolve(12, 40, 3, seed=100 + rep % 3, p_change=0.1 + 0.1 * (rep % 3))   # three alignments, each twice
        pd = data.from_states(syn.states, 61, compress_patterns=False)                       # (the same S every time)
        flat = syn.flat
        B = flat.n_branches
        ts = np.random.default_rng(rep).uniform(0.01, 0.6, size=B)
        Q = models.mg94rev_Q_batch(ts, 0.5 + 0.1 * rep, rev, pf)
        nodes = np.arange(B, dtype=np.int64)
        op = oracle.OraclePartition(61, flat.flat_parents, flat.L, pd.leaf_codes, None, pd.pattern_freq)
        op.set_P(nodes, oracle.expm(Q, True))
        ref = op.compute_block(nodes, pi)
        ref_site = op.site_log_likelihoods(nodes, pi)
        with hip.HipPartition(61, flat.flat_parents, flat.L, pd.leaf_codes, None, pd.pattern_freq, 1) as part:
            ll, lik, sc = part.evaluate(nodes, nodes, Q, pi, per_site=True)
            assert abs(ll - ref) <= RTOL * abs(ref), (rep, ll, ref)
            got_site = np.log(lik) - sc * 64 * np.log(2.0)
            assert np.max(np.abs(got_site - ref_site) / np.abs(ref_site)) < RTOL
            node = int(rep % B)
            Qn = models.mg94rev_Q_batch(ts[node:node + 1] * 1.7, 0.5 + 0.1 * rep, rev, pf)
            un = flat.path_update_nodes(node)
            ll2 = part.evaluate(un, [node], Qn, pi)
            op.set_P([node], oracle.expm(Qn, True))
            ref2 = op.compute_block(un, pi)
            assert abs(ll2 - ref2) <= RTOL * abs(ref2), (rep, ll2, ref2)
        seen.add(round(ll, 6))
    assert len(seen) >= 3


def test_staged_rows_are_claimed_by_their_first_consumer():
    """hyphy_hip_build_q stages coefficient rows without saying what they are — one row per (class, branch) for hyphy_hip_evaluate_built,
    one per (branch, mixture component) for hyphy_hip_evaluate_mixture_built.  The first evaluation that consumes a staging claims it; an
    evaluation of the OTHER kind that happens to need the same number of rows is refused instead of reading rows that mean something
    else, and a fresh hyphy_hip_build_q makes either kind possible again."""
    import ctypes as C
    from hyphy_amd import hip, models
    fx = common.load("codon_small")
    rev = dict(zip(common.REV_KEYS, (float(x) for x in fx["rev"])))
    t = np.asarray(fx["t"], dtype=np.float64)
    nodes = common.all_nodes(fx)
    B = len(nodes)
    T = np.zeros((2, 61, 61))
    rv = dict(rev, AG=1.0)
    for (i, j, nm, ns, pf) in models.mg94rev_template(fx["pos_freqs"]):
        T[1 if ns else 0, i, j] = rv[nm] * pf
    coeffs = np.ascontiguousarray(np.stack([t, t * float(fx["omega"])], axis=1))   # [B, 2]: one component per branch = the plain model
    W = np.ones((B, 1))
    ref = float(fx["logl"])
    rf = np.ascontiguousarray(fx["root_freqs"], dtype=np.float64)
    with _mk(fx) as part:
        part.set_q_templates(T)
        ll = part.evaluate_mixture_built(nodes, nodes, coeffs[:, None, :], W, rf)       # stages B rows, claims them as components
        assert abs(ll - ref) <= RTOL * abs(ref)
        out = C.c_double(0.0)
        rc = part._lib.hyphy_hip_evaluate_built(part._h, -1, hip._l(nodes), B, hip._l(nodes), B, hip._d(rf), C.byref(out))
        assert rc != 0 and b"hyphy_hip_build_q first" in part._lib.hyphy_hip_last_error()
        step = part.prepare_built_step(nodes, nodes, rf, coeffs)                        # stages afresh: now one row per branch
        assert abs(step() - ref) <= RTOL * abs(ref)
        cnt = np.ones(B, dtype=np.int64)
        rc = part._lib.hyphy_hip_evaluate_mixture_built(part._h, -1, hip._l(nodes), B, hip._l(nodes), B, hip._l(cnt), hip._d(W), hip._d(rf),
                                                        C.byref(out), None, None)
        assert rc != 0 and b"hyphy_hip_build_q first" in part._lib.hyphy_hip_last_error()
        assert abs(part.evaluate_mixture_built(nodes, nodes, coeffs[:, None, :], W, rf) - ref) <= RTOL * abs(ref)
