"""Subtree repeats (hyphy_amd/csrc/repeats.hip — the device form of the reference's `tcc` traversal masks, src/core/tree.cpp:2801-2858,
src/core/tree_evaluator.cpp:57-76, 240-256): the class-compressed evaluation against the plain one, against the reference's golden
vectors and against the CPU restatement, on every entry point that runs compressed and on every one that must fall back.

HYPHY_HIP_REPEATS=2 forces the compressed form where the library would not choose it (small fixtures); =0 turns it off; `set_repeats`
switches one partition.  Results of the two forms agree to rounding (the products are the same, multiplied in a different order)."""
import numpy as np
import pytest

from tests import common

pytestmark = pytest.mark.gpu

RTOL = 1e-10       # against the reference
SAME = 1e-13       # compressed against plain
LOG_SCALER = 64.0 * np.log(2.0)


def _hip():
    from hyphy_amd import hip
    return hip


def _mk(fx, C=1, **kw):
    hip = _hip()
    return hip.HipPartition(int(fx["D"]), fx["flat_parents"], int(fx["L"]), fx["leaf_codes"], fx["ambig"], fx["pattern_freq"], C, **kw)


def _site(lik, sc):
    return np.log(lik) - sc * LOG_SCALER


@pytest.mark.parametrize("levels", ["1", "0"])
@pytest.mark.parametrize("rho,inline", [("0", "1"), ("0", "0"), ("0.6", "0"), ("0.6", "1"), ("2", "0"), ("2", "1")])
@pytest.mark.parametrize("name", ["codon_small", "codon_ambig", "codon_deep", "codon_wide", "ref_smallcodon"])
def test_compressed_equals_plain_and_reference(name, rho, inline, levels, monkeypatch):
    """Full passes (first: every node stored; then steady state), per-pattern values and 2^64 exponents: compressed against plain
    against the reference.  rho = 2: one table per node; 0.6: paths while the child keeps 60 % of the classes; 0: longest paths.
    inline: leaf-only side chains walked inside the item of the path they hang off (the default with rho = 0) or as tables.
    levels: tables that read tables of the same pass run as one launch per level of that dependency (1, the default) or in one
    launch under the ticket protocol (0)."""
    monkeypatch.setenv("HYPHY_HIP_REP_LEVELS", levels)
    monkeypatch.setenv("HYPHY_HIP_REPEATS", "2")
    monkeypatch.setenv("HYPHY_HIP_REP_INLINE", inline)
    monkeypatch.setenv("HYPHY_HIP_REP_RHO", rho)
    monkeypatch.setenv("HYPHY_HIP_REP_THETA", "0.9")
    monkeypatch.setenv("HYPHY_HIP_POISON", "1")
    fx = common.load(name)
    Q = common.fixture_Q(fx)
    nodes = common.all_nodes(fx)
    with _mk(fx) as part:
        st = part.repeat_stats()
        assert st["available"] == 1 and st["in_use"] == 1 and st["tables"] > 0, st
        got = []
        for on in (True, False, True):
            part.set_repeats(on)
            for _ in range(3):   # persisting pass, then lazy steady state
                ll, lik, sc = part.evaluate(nodes, nodes, Q, fx["root_freqs"], per_site=True)
            got.append((ll, _site(lik, sc), sc.copy()))
    ref = float(fx["logl"])
    for ll, site, sc in got:
        assert abs(ll - ref) <= RTOL * abs(ref), (ll, ref)
        assert np.max(np.abs(site[fx["site_to_pattern"]] - fx["site_logl"]) / np.abs(fx["site_logl"])) < RTOL
    assert abs(got[0][0] - got[1][0]) <= SAME * abs(ref)
    assert np.max(np.abs(got[0][1] - got[1][1]) / np.abs(got[1][1])) < SAME
    assert got[0][0] == got[2][0]   # (the lower phase has no arrival-order joins: same bits when the trunk's schedule is the same)
    if name.endswith("deep"):
        assert got[0][2].max() > 0   # exponents per class were really exercised


@pytest.mark.parametrize("rho,inline", [("0", "1"), ("0.6", "0"), ("2", "1")])
@pytest.mark.parametrize("name", ["codon_small", "codon_ambig", "codon_deep", "ref_smallcodon"])
def test_row_split_walk_equals_one_wave_walk(name, rho, inline, monkeypatch):
    """r06: static lower-phase launches walk a path with a workgroup of NW row-split waves (class_table_team_kernel, the default);
    HYPHY_HIP_REP_TEAM=0 keeps the one-wave walk of r05 (class_table_kernel).  Same tables, same exponents: per-pattern values of the
    two forms agree to rounding (the rescale is applied on the other side of the product), both equal the reference's."""
    monkeypatch.setenv("HYPHY_HIP_REPEATS", "2")
    monkeypatch.setenv("HYPHY_HIP_REP_INLINE", inline)
    monkeypatch.setenv("HYPHY_HIP_REP_RHO", rho)
    monkeypatch.setenv("HYPHY_HIP_REP_THETA", "0.9")
    monkeypatch.setenv("HYPHY_HIP_POISON", "1")
    fx = common.load(name)
    Q = common.fixture_Q(fx)
    nodes = common.all_nodes(fx)
    got = {}
    for team in ("1", "0"):
        monkeypatch.setenv("HYPHY_HIP_REP_TEAM", team)
        with _mk(fx) as part:
            assert part.repeat_stats()["in_use"] == 1
            for _ in range(3):
                ll, lik, sc = part.evaluate(nodes, nodes, Q, fx["root_freqs"], per_site=True)
            got[team] = (ll, _site(lik, sc), sc.copy())
    ref = float(fx["logl"])
    for ll, site, sc in got.values():
        assert abs(ll - ref) <= RTOL * abs(ref), (ll, ref)
        assert np.max(np.abs(site[fx["site_to_pattern"]] - fx["site_logl"]) / np.abs(fx["site_logl"])) < RTOL
    assert abs(got["1"][0] - got["0"][0]) <= SAME * abs(ref)
    assert np.max(np.abs(got["1"][1] - got["0"][1]) / np.abs(got["0"][1])) < SAME
    assert np.array_equal(got["1"][2], got["0"][2]) or name.endswith("deep")   # (exponents may split differently only where a total sits on the threshold)


@pytest.mark.parametrize("kernel,m", [("2", "3"), ("2", "1"), ("0", None)])
@pytest.mark.parametrize("name", ["codon_small", "codon_ambig", "codon_deep", "codon_wide", "ref_smallcodon"])
def test_trunk_under_row_split_workgroups(name, kernel, m, monkeypatch):
    """r06: the trunk of a class-compressed partition under prune_mfma_kernel<.., REP> (generalised leaves gathered by class id, their
    exponents added) — on chain schedules (kernel 2) and one workgroup per tile (kernel 0) — against the wave-per-tile trunk, the
    plain form and the reference; first (persisting) pass and lazy steady state."""
    monkeypatch.setenv("HYPHY_HIP_REPEATS", "2")
    monkeypatch.setenv("HYPHY_HIP_REP_THETA", "0.9")
    monkeypatch.setenv("HYPHY_HIP_POISON", "1")
    monkeypatch.setenv("HYPHY_HIP_TRUNK_WALK", "0")   # (the pruning kernels on the trunk are the subject: without the tuner the walk would take the lazy passes)
    fx = common.load(name)
    Q = common.fixture_Q(fx)
    nodes = common.all_nodes(fx)
    got = {}
    for form in ("team", "wave"):
        if form == "team":
            monkeypatch.setenv("HYPHY_HIP_TRUNK_KERNEL", kernel)
            if m:
                monkeypatch.setenv("HYPHY_HIP_CHAIN_M", m)
            else:
                monkeypatch.setenv("HYPHY_HIP_TUNE", "0")
        else:
            monkeypatch.delenv("HYPHY_HIP_TRUNK_KERNEL")
            monkeypatch.delenv("HYPHY_HIP_CHAIN_M", raising=False)
            monkeypatch.setenv("HYPHY_HIP_TUNE", "0")   # (the tuner would offer — and on shards this small pick, or take from its cache — the
                                                        #  row-split form: without it the trunk stays with the wave-per-tile kernel)
        with _mk(fx) as part:
            assert part.repeat_stats()["in_use"] == 1
            for _ in range(3):
                ll, lik, sc = part.evaluate(nodes, nodes, Q, fx["root_freqs"], per_site=True)
            got[form] = (ll, _site(lik, sc), part.prune_kernel_name())
            part.set_repeats(False)
            ll0, lik0, sc0 = part.evaluate(nodes, nodes, Q, fx["root_freqs"], per_site=True)
            got[form + "_plain"] = (ll0, _site(lik0, sc0))
    assert got["team"][2] == "prune_mfma_kernel" and got["wave"][2] == "prune_wave_kernel", (got["team"][2], got["wave"][2])
    ref = float(fx["logl"])
    for k in ("team", "wave"):
        assert abs(got[k][0] - ref) <= RTOL * abs(ref), (k, got[k][0], ref)
        assert np.max(np.abs(got[k][1][fx["site_to_pattern"]] - fx["site_logl"]) / np.abs(fx["site_logl"])) < RTOL
    assert np.max(np.abs(got["team"][1] - got["wave"][1]) / np.abs(got["wave"][1])) < SAME
    assert np.max(np.abs(got["team"][1] - got["team_plain"][1]) / np.abs(got["team_plain"][1])) < SAME


@pytest.mark.parametrize("chains", ["1", "2"])
@pytest.mark.parametrize("theta", ["0.9", "0.3", "0.05"])
@pytest.mark.parametrize("name", ["codon_small", "codon_ambig", "codon_deep", "codon_wide", "ref_smallcodon"])
def test_trunk_as_one_row_split_walk_per_tile(name, theta, chains, monkeypatch):
    """r06: lazy full passes of the trunk under trunk_walk_kernel (repeats.hip) — one post-order walk per tile by a workgroup of NW
    row-split waves, internal side chains behind a push / pop of the running product, the root's conditionals straight into the
    per-pattern values — against the wave-per-tile trunk, the plain form and the reference.  theta moves the cut: a trunk of one or
    two nodes (0.9), a bushy one (0.05: most of the tree, nested side chains).  chains = 2: the subtrees below the root dealt to two
    workgroups per tile that meet at the root through memory (where the trunk has two of a size worth it; else one walk)."""
    monkeypatch.setenv("HYPHY_HIP_WALK_CHAINS", chains)
    monkeypatch.setenv("HYPHY_HIP_REPEATS", "2")
    monkeypatch.setenv("HYPHY_HIP_REP_THETA", theta)
    monkeypatch.setenv("HYPHY_HIP_POISON", "1")
    monkeypatch.setenv("HYPHY_HIP_TUNE", "0")
    fx = common.load(name)
    Q = common.fixture_Q(fx)
    nodes = common.all_nodes(fx)
    got = {}
    for form in ("walk", "wave"):
        monkeypatch.setenv("HYPHY_HIP_TRUNK_WALK", "1" if form == "walk" else "0")
        with _mk(fx) as part:
            if part.repeat_stats()["in_use"] != 1:
                pytest.skip("nothing to compress at this threshold")
            names = []
            for _ in range(3):   # (first pass persists: the pruning kernel; from the second on: the walk)
                ll, lik, sc = part.evaluate(nodes, nodes, Q, fx["root_freqs"], per_site=True)
                names.append(part.prune_kernel_name())
            got[form] = (ll, _site(lik, sc), names, sc.copy())
            # a partial update behind lazy passes (the walk persists nothing: the library restores the copies first), then lazy again
            some = nodes[: max(1, len(nodes) // 3)]
            ll_p, lik_p, sc_p = part.evaluate(some, some, Q[: len(some)], fx["root_freqs"], per_site=True)
            got[form + "_partial"] = (ll_p, _site(lik_p, sc_p))
            ll_b, lik_b, sc_b = part.evaluate(nodes, nodes, Q, fx["root_freqs"], per_site=True)
            ll_b, lik_b, sc_b = part.evaluate(nodes, nodes, Q, fx["root_freqs"], per_site=True)
            got[form + "_back"] = (ll_b, _site(lik_b, sc_b), part.prune_kernel_name())
    assert got["walk"][2][0] != "trunk_walk_kernel" and got["walk"][2][-1] == "trunk_walk_kernel", got["walk"][2]
    assert got["walk_back"][2] == "trunk_walk_kernel"
    assert "trunk_walk_kernel" not in got["wave"][2]
    ref = float(fx["logl"])
    for k in ("walk", "wave", "walk_partial", "walk_back"):
        assert abs(got[k][0] - ref) <= RTOL * abs(ref), (k, got[k][0], ref)
        assert np.max(np.abs(got[k][1][fx["site_to_pattern"]] - fx["site_logl"]) / np.abs(fx["site_logl"])) < RTOL, k
    assert np.max(np.abs(got["walk"][1] - got["wave"][1]) / np.abs(got["wave"][1])) < SAME


@pytest.mark.parametrize("seed,taxa,D", [(1, 40, 61), (2, 33, 61), (3, 24, 20), (4, 17, 48), (5, 30, 5)])
def test_random_trees_and_state_counts_against_the_oracle(seed, taxa, D, monkeypatch):
    """Random trees (multifurcating root), random ambiguity codes, 61 / 48 / 20 / 5 states (NW = 4, 3, 2, 1 row blocks), enough
    patterns for several tiles per table; compressed against plain against the CPU restatement."""
    from hyphy_amd import data
    from oracle import oracle
    monkeypatch.setenv("HYPHY_HIP_REPEATS", "2")
    monkeypatch.setenv("HYPHY_HIP_REP_THETA", "0.8")
    rng = np.random.default_rng(seed)
    syn = data.evolve(taxa, 700, 3, seed=seed, p_change=0.08)
    flat = syn.flat
    L, B = flat.L, flat.n_branches
    S = 500
    codes = rng.integers(0, min(D, 3), size=(L, S)).astype(np.int64)     # few states per column: many repeats below a node
    codes[:, S // 2:] = codes[:, : S - S // 2]                           # ... and whole patterns twice (with different weights)
    n_amb = 3
    ambig = (rng.random((n_amb, D)) < 0.5).astype(np.float64)
    ambig[:, 0] = 1.0
    mask = rng.random((L, S)) < 0.03
    codes[mask] = -rng.integers(1, n_amb + 1, size=int(mask.sum()))
    freq = rng.integers(1, 5, size=S).astype(np.int64)
    Q = np.zeros((B, D, D))
    for b in range(B):
        M = rng.random((D, D)) * rng.uniform(0.01, 0.4)
        np.fill_diagonal(M, 0.0)
        np.fill_diagonal(M, -M.sum(1))
        Q[b] = M
    pi = rng.random(D)
    pi /= pi.sum()
    nodes = np.arange(B, dtype=np.int64)
    hip = _hip()
    with hip.HipPartition(D, flat.flat_parents, L, codes, ambig, freq) as part:
        assert part.repeat_stats()["in_use"] == 1
        ll1, lik1, sc1 = part.evaluate(nodes, nodes, Q, pi, per_site=True)
        part.set_repeats(False)
        ll0, lik0, sc0 = part.evaluate(nodes, nodes, Q, pi, per_site=True)
    op = oracle.OraclePartition(D, flat.flat_parents, L, codes, ambig, freq)
    op.set_P(nodes, oracle.expm(Q, False))
    ref = op.compute_block(nodes, pi)
    assert abs(ll1 - ref) <= RTOL * abs(ref), (ll1, ref)
    assert abs(ll1 - ll0) <= SAME * abs(ref)
    assert np.max(np.abs(_site(lik1, sc1) - _site(lik0, sc0))) < 1e-11


@pytest.mark.parametrize("rho", ["0", "0.3", "2"])
@pytest.mark.parametrize("name", ["codon_deep", "codon_wide", "codon_ambig"])
def test_partial_updates_on_the_compressed_form(name, rho, monkeypatch):
    """DetermineNodesForUpdate-style dirty lists (a changed branch, its ancestors): the compressed form recomputes the class tables
    whose path contains a touched node or whose own branch changed, and the trunk above them; against a full pass and against plain.
    rho > 0: tables read tables — a partial pass launches the levels of the STALE tables only."""
    monkeypatch.setenv("HYPHY_HIP_REP_RHO", rho)
    monkeypatch.setenv("HYPHY_HIP_REPEATS", "2")
    monkeypatch.setenv("HYPHY_HIP_REP_THETA", "0.9")
    fx = common.load(name)
    Q = common.fixture_Q(fx)
    nodes = common.all_nodes(fx)
    B = len(nodes)
    rng = np.random.default_rng(7)
    with _mk(fx) as part, _mk(fx) as plain:
        plain.set_repeats(False)
        part.evaluate(nodes, nodes, Q, fx["root_freqs"])
        plain.evaluate(nodes, nodes, Q, fx["root_freqs"])
        Qc = Q.copy()
        for trial in range(12):
            k = int(rng.integers(1, 4))
            ch = np.sort(rng.choice(B, size=k, replace=False)).astype(np.int64)
            Qc[ch] *= rng.uniform(0.5, 1.8)
            a, la, sa = part.evaluate(ch, ch, Qc[ch], fx["root_freqs"], per_site=True)
            b, lb, sb = plain.evaluate(ch, ch, Qc[ch], fx["root_freqs"], per_site=True)
            assert abs(a - b) <= SAME * abs(b), (trial, ch, a, b)
            assert np.max(np.abs(_site(la, sa) - _site(lb, sb))) < 1e-11
        full = part.evaluate(nodes, nodes, Qc, fx["root_freqs"])
        assert abs(full - a) <= SAME * abs(a)
        # a pure re-evaluation (nothing changed) and new root frequencies only
        again = part.evaluate(np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64), None, fx["root_freqs"])
        assert again == full or abs(again - full) <= SAME * abs(full)


def test_rate_classes_batched_on_the_compressed_form(monkeypatch):
    """Three rate classes in one launch: class tables per class, one lower phase over all of them; against the reference."""
    monkeypatch.setenv("HYPHY_HIP_REPEATS", "2")
    monkeypatch.setenv("HYPHY_HIP_REP_THETA", "0.9")
    fx = common.load("codon_cat3")
    nodes = common.all_nodes(fx)
    w, vals = fx["cat_weights"], fx["cat_values"]
    Q = np.stack([common.fixture_Q(fx, float(v)) for v in vals])
    with _mk(fx, C=len(w)) as part:
        assert part.repeat_stats()["in_use"] == 1
        out = []
        for on in (True, False):
            part.set_repeats(on)
            for _ in range(2):
                ll, lik, sc = part.evaluate_categories(nodes, nodes, Q, w, fx["root_freqs"], per_site=True)
            out.append((ll, np.where(lik > 0, _site(np.maximum(lik, 1e-300), sc), -np.inf)))
    ref = float(fx["logl"])
    assert abs(out[0][0] - ref) <= RTOL * abs(ref)
    assert abs(out[0][0] - out[1][0]) <= SAME * abs(ref)
    assert np.max(np.abs(out[0][1] - out[1][1])) < 1e-11


def test_explicit_form_mixture_on_the_compressed_form(monkeypatch):
    """P_b = sum_m w_m exp(Q_b^(m)) formed on the device feeds the class tables like any other matrix image."""
    monkeypatch.setenv("HYPHY_HIP_REPEATS", "2")
    monkeypatch.setenv("HYPHY_HIP_REP_THETA", "0.9")
    from hyphy_amd import models
    fx = common.load("codon_mix3")
    nodes = common.all_nodes(fx)
    rev = dict(zip(common.REV_KEYS, (float(x) for x in fx["rev"])))
    t = np.asarray(fx["t"], dtype=np.float64)
    Qc = np.stack([models.mg94rev_Q_batch(t, float(om), rev, fx["pos_freqs"]) for om in fx["omegas"]], axis=1)   # [B, M, D, D]
    W = np.tile(np.asarray(fx["weights"], dtype=np.float64), (len(nodes), 1))
    with _mk(fx) as part:
        assert part.repeat_stats()["in_use"] == 1
        ll1 = part.evaluate_mixture(nodes, nodes, Qc, W, fx["root_freqs"])
        part.set_repeats(False)
        ll0 = part.evaluate_mixture(nodes, nodes, Qc, W, fx["root_freqs"])
    ref = float(fx["logl"])
    assert abs(ll1 - ref) <= RTOL * abs(ref), (ll1, ref)
    assert abs(ll1 - ll0) <= SAME * abs(ref)


@pytest.mark.parametrize("name", ["codon_ambig", "codon_deep"])
def test_plain_only_entry_points_on_a_compressed_partition(name, monkeypatch):
    """Pinned states, downloads of the per-pattern conditionals and the branch cache run on the partition's own tree; in between
    the ordinary evaluations go back to the compressed form (their tables and the trunk's copies are rebuilt by a full pass)."""
    from oracle import oracle
    monkeypatch.setenv("HYPHY_HIP_REPEATS", "2")
    monkeypatch.setenv("HYPHY_HIP_REP_THETA", "0.9")
    fx = common.load(name)
    Q = common.fixture_Q(fx)
    nodes = common.all_nodes(fx)
    D, L = int(fx["D"]), int(fx["L"])
    S = fx["leaf_codes"].shape[1]
    rng = np.random.default_rng(11)
    with _mk(fx) as part, _mk(fx) as plain:
        plain.set_repeats(False)
        ref = plain.evaluate(nodes, nodes, Q, fx["root_freqs"])
        assert abs(part.evaluate(nodes, nodes, Q, fx["root_freqs"]) - ref) <= SAME * abs(ref)
        # conditionals of every node, reference layout
        c1, n1 = part.download_partials()
        c0, n0 = plain.download_partials()
        assert np.array_equal(n1, n0)
        assert np.allclose(c1, c0, rtol=1e-12, atol=0)
        # pinned states at an internal node and at a leaf
        for node in (L + 2, 1):
            states = rng.integers(0, D, size=S).astype(np.int64)
            part.set_pinned_states(node, states)
            plain.set_pinned_states(node, states)
            a = part.evaluate(nodes, np.zeros(0, dtype=np.int64), None, fx["root_freqs"])
            b = plain.evaluate(nodes, np.zeros(0, dtype=np.int64), None, fx["root_freqs"])
            assert (np.isinf(a) and np.isinf(b)) or abs(a - b) <= SAME * abs(b), (node, a, b)
            part.set_pinned_states(None)
            plain.set_pinned_states(None)
        assert abs(part.evaluate(nodes, np.zeros(0, dtype=np.int64), None, fx["root_freqs"]) - ref) <= SAME * abs(ref)
        # branch cache: one branch varies, then an ordinary partial update of the same branch
        node = L + 1
        part.branch_cache_build(node)
        q1 = Q[node] * 1.3
        a = part.branch_cache_evaluate(node, q1)
        Q2 = Q.copy()
        Q2[node] = q1
        b = plain.evaluate(nodes, nodes, Q2, fx["root_freqs"])
        assert abs(a - b) <= 1e-12 * abs(b), (a, b)
        ch = np.array([node], dtype=np.int64)
        c = part.evaluate(ch, ch, Q2[ch], fx["root_freqs"])
        assert abs(c - b) <= SAME * abs(b), (c, b)
        assert part.repeat_stats()["in_use"] == 1


def test_library_chooses_by_measurement(monkeypatch):
    """Without HYPHY_HIP_REPEATS the library times both forms on the first steady-state pass and keeps the faster one; either way the
    values are the reference's."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tests.test_gpu_fullsize import _timed_path
    monkeypatch.delenv("HYPHY_HIP_REPEATS", raising=False)
    fx = common.load("full_mg94_64x10k_sweep")
    pts = [1, 2, 3, 4, 5]
    got, info, _ = _timed_path(fx, {}, monkeypatch, pts)
    assert "repeats: lower phase" in info, info
    for k in pts:
        assert abs(got[k] - fx["sweep_logl"][k - 1]) <= RTOL * abs(fx["sweep_logl"][k - 1]), (k, info)


def test_tables_that_do_not_fit_leave_a_plain_partition(monkeypatch):
    """Subtree repeats are an optimisation: when the class tables cannot be set up (here: a bound of 0 MB on them; in the field: a device
    short of memory) hyphy_hip_create still succeeds, nothing of the attempt stays allocated, and the partition evaluates plain."""
    monkeypatch.setenv("HYPHY_HIP_REPEATS", "2")
    monkeypatch.setenv("HYPHY_HIP_REP_THETA", "0.9")
    monkeypatch.setenv("HYPHY_HIP_REP_MAX_MB", "0")
    fx = common.load("codon_wide")
    Q = common.fixture_Q(fx)
    nodes = common.all_nodes(fx)
    with _mk(fx) as part:
        st = part.repeat_stats()
        assert st["available"] == 0 and st["in_use"] == 0 and st["tables"] == 0, st
        part.set_repeats(True)   # (asking for them changes nothing when they do not exist)
        ll = part.evaluate(nodes, nodes, Q, fx["root_freqs"])
        ch = np.array([3], dtype=np.int64)
        Q2 = Q.copy()
        Q2[3] *= 1.5
        ll2 = part.evaluate(ch, ch, Q2[ch], fx["root_freqs"])
        full2 = part.evaluate(nodes, nodes, Q2, fx["root_freqs"])
    ref = float(fx["logl"])
    assert abs(ll - ref) <= RTOL * abs(ref), (ll, ref)
    assert abs(ll2 - full2) <= SAME * abs(full2)


@pytest.mark.parametrize("name", ["nuc_small", "nuc_ambig", "nuc_deep", "nuc_wide", "ref_fluHA"])
def test_four_states_compressed_equals_plain_and_reference(name, monkeypatch):
    """4 states: every compressed subtree is walked whole by one thread per class of its root (class_table_nuc_kernel), the trunk goes
    through prune_nuc_kernel with generalised leaves.  Full passes, partial updates, ambiguity codes (inside walks and at the trunk's
    own leaves), a 300-taxon ladder that rescales (and whose walks exceed the stack: the roots move down), against plain and reference."""
    monkeypatch.setenv("HYPHY_HIP_REPEATS", "2")
    monkeypatch.setenv("HYPHY_HIP_POISON", "1")
    fx = common.load(name)
    Q = common.fixture_Q(fx)
    nodes = common.all_nodes(fx)
    B = len(nodes)
    rng = np.random.default_rng(3)
    with _mk(fx) as part, _mk(fx) as plain:
        st = part.repeat_stats()
        assert st["available"] == 1 and st["in_use"] == 1 and st["tables"] > 0, st
        assert part.prune_kernel_name() in ("prune_nuc_kernel", "prune_nuc2_kernel")
        plain.set_repeats(False)
        for _ in range(3):
            ll, lik, sc = part.evaluate(nodes, nodes, Q, fx["root_freqs"], per_site=True)
        ll0, lik0, sc0 = plain.evaluate(nodes, nodes, Q, fx["root_freqs"], per_site=True)
        ref = float(fx["logl"])
        assert abs(ll - ref) <= RTOL * abs(ref), (ll, ref)
        site = _site(lik, sc)
        assert np.max(np.abs(site[fx["site_to_pattern"]] - fx["site_logl"]) / np.abs(fx["site_logl"])) < RTOL
        assert abs(ll - ll0) <= SAME * abs(ref)
        assert np.max(np.abs(site - _site(lik0, sc0))) < 1e-11
        if name.endswith("deep"):
            assert sc.max() > 0
        Qc = Q.copy()
        for trial in range(8):
            ch = np.sort(rng.choice(B, size=int(rng.integers(1, 4)), replace=False)).astype(np.int64)
            Qc[ch] *= rng.uniform(0.5, 1.8)
            a = part.evaluate(ch, ch, Qc[ch], fx["root_freqs"])
            b = plain.evaluate(ch, ch, Qc[ch], fx["root_freqs"])
            assert abs(a - b) <= SAME * abs(b), (trial, ch, a, b)
        assert abs(part.evaluate(nodes, nodes, Qc, fx["root_freqs"]) - a) <= SAME * abs(a)
