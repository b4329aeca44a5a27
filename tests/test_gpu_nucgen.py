"""4 states: run-time generated straight-line kernels (hyphy_amd/csrc/nucgen.hip) against the schedule interpreter
(prune_nuc2_kernel) and against the reference (src/core/tree_evaluator.cpp:2253-2273, 3556-4171): the generated kernel executes the
same factors in the same order, so the two forms must agree BIT FOR BIT per pattern; both must equal the reference's golden values.

HYPHY_HIP_NUCGEN=0: interpreter only; =2: compile synchronously at the first full pass; default (1): a background thread compiles
after HYPHY_HIP_NUCGEN_AFTER evaluations under one schedule and the partition switches over when the code object is there."""
import time

import numpy as np
import pytest

from tests import common

pytestmark = pytest.mark.gpu

RTOL = 1e-10
LOG_SCALER = 64.0 * np.log(2.0)


def _hip():
    from hyphy_amd import hip
    return hip


def _mk(fx):
    hip = _hip()
    return hip.HipPartition(int(fx["D"]), fx["flat_parents"], int(fx["L"]), fx["leaf_codes"], fx["ambig"], fx["pattern_freq"])


def _run(fx, mode, monkeypatch, n_eval=3, small=None):
    monkeypatch.setenv("HYPHY_HIP_NUCGEN", mode)
    if small is None:
        monkeypatch.delenv("HYPHY_HIP_NUCGEN_SMALL", raising=False)
    else:
        monkeypatch.setenv("HYPHY_HIP_NUCGEN_SMALL", small)
    Q = common.fixture_Q(fx)
    nodes = common.all_nodes(fx)
    with _mk(fx) as part:
        for _ in range(n_eval):   # persisting pass, then the lazy steady state (the schedule a kernel is generated for)
            ll, lik, sc = part.evaluate(nodes, nodes, Q, fx["root_freqs"], per_site=True)
        return ll, lik.copy(), sc.copy(), part.prune_kernel_name()


@pytest.mark.parametrize("small", ["0", "1"])
@pytest.mark.parametrize("name", ["nuc_small", "nuc_ambig", "nuc_deep", "nuc_wide", "ref_fluHA"])
def test_generated_equals_interpreted_bit_for_bit_and_the_reference(name, small, monkeypatch):
    """small = 0: matrices through scalar loads, exponentials and combine in launches of their own (the form of large shards);
    1: every matrix in LDS, exponentials and final combine inside the launch (the form of shards of at most two workgroups per CU)."""
    fx = common.load(name)
    ll0, lik0, sc0, k0 = _run(fx, "0", monkeypatch)
    ll2, lik2, sc2, k2 = _run(fx, "2", monkeypatch, small=small)
    if int(fx["L"]) > 256:   # (the leaf matrices of such a tree do not fit the LDS table: prune_nuc_kernel, no generated form)
        assert k0 == k2 == "prune_nuc_kernel"
    else:
        assert k0 == "prune_nuc2_kernel" and k2 == "nucgen_kernel", (k0, k2)
    assert np.array_equal(lik0, lik2) and np.array_equal(sc0, sc2)          # same factors, same order: same bits
    ref = float(fx["logl"])
    assert abs(ll2 - ref) <= RTOL * abs(ref) and abs(ll0 - ref) <= RTOL * abs(ref), (ll0, ll2, ref)
    site = np.log(lik2) - LOG_SCALER * sc2
    assert np.max(np.abs(site[fx["site_to_pattern"]] - fx["site_logl"]) / np.abs(fx["site_logl"])) < RTOL
    if name.endswith("deep"):
        assert sc2.max() > 0   # (a tree that rescales)


def test_partial_updates_and_pinned_states_stay_with_the_interpreter(monkeypatch):
    """Only full passes run generated code: a one-branch update (its own schedule), then a full pass again — values against the
    CPU restatement at every step; the kernel name says which form ran."""
    from hyphy_amd import tree
    from oracle import oracle
    monkeypatch.setenv("HYPHY_HIP_NUCGEN", "2")
    fx = common.load("nuc_wide")
    flat = tree.flat_from_parents(fx["flat_parents"], int(fx["L"]))
    Q = common.fixture_Q(fx)
    nodes = common.all_nodes(fx)
    B = len(nodes)
    op = oracle.OraclePartition(4, fx["flat_parents"], int(fx["L"]), fx["leaf_codes"], fx["ambig"], fx["pattern_freq"])
    with _mk(fx) as part:
        for _ in range(3):
            ll = part.evaluate(nodes, nodes, Q, fx["root_freqs"])
        assert part.prune_kernel_name() == "nucgen_kernel"
        op.set_P(nodes, oracle.expm(Q, False))
        ref = op.compute_block(nodes, fx["root_freqs"])
        assert abs(ll - ref) <= RTOL * abs(ref)
        Q2 = Q.copy()
        b = B // 2
        Q2[b] = Q[b] * 1.7
        upd = np.array([b], dtype=np.int64)
        un = flat.path_update_nodes(int(b))   # DetermineNodesForUpdate-style list: the branch, its ancestors and their children
        # (the lazy full passes left no persisted copies: the library promotes this update to a persisting full pass — a schedule of
        #  its own, generated too in mode 2 — and the NEXT one-branch update is a true partial pass: the interpreter)
        ll_p = part.evaluate(un, upd, Q2[b:b + 1], fx["root_freqs"])
        op.set_P(upd, oracle.expm(Q2[b:b + 1], False))
        ref_p = op.compute_block(un, fx["root_freqs"])
        assert abs(ll_p - ref_p) <= RTOL * abs(ref_p), (ll_p, ref_p)
        Q2[b] = Q[b] * 0.6
        ll_p = part.evaluate(un, upd, Q2[b:b + 1], fx["root_freqs"])
        assert part.prune_kernel_name() == "prune_nuc2_kernel"
        op.set_P(upd, oracle.expm(Q2[b:b + 1], False))
        ref_p = op.compute_block(un, fx["root_freqs"])
        assert abs(ll_p - ref_p) <= RTOL * abs(ref_p), (ll_p, ref_p)
        for _ in range(3):
            ll_f = part.evaluate(nodes, nodes, Q2, fx["root_freqs"])
        assert part.prune_kernel_name() == "nucgen_kernel"
        assert abs(ll_f - ref_p) <= RTOL * abs(ref_p), (ll_f, ref_p)


def test_background_compilation_switches_over_without_changing_a_bit(monkeypatch):
    """Default mode: the interpreter runs until the background thread has the code object (requested after HYPHY_HIP_NUCGEN_AFTER
    evaluations); every evaluation on the way returns the same bits.  A tree no other test of this process uses (code objects are
    shared per process by schedule), values against the CPU restatement."""
    from hyphy_amd import data, hip, models
    from oracle import oracle
    monkeypatch.setenv("HYPHY_HIP_NUCGEN", "1")
    monkeypatch.setenv("HYPHY_HIP_NUCGEN_AFTER", "3")
    monkeypatch.delenv("HYPHY_HIP_NUCGEN_SMALL", raising=False)
    syn = data.evolve(23, 700, 1, seed=977, p_change=0.1)
    pd = data.from_states(syn.states, 4, compress_patterns=True)
    flat = syn.flat
    B = flat.n_branches
    pi = np.array([0.3, 0.2, 0.15, 0.35])
    rng = np.random.default_rng(9)
    Q = np.stack([models.nuc_rev_Q(float(rng.uniform(0.02, 0.3)), dict(AC=0.5, AT=0.4, CG=0.4, CT=1.2, GT=0.4), pi) for _ in range(B)])
    nodes = np.arange(B, dtype=np.int64)
    op = oracle.OraclePartition(4, flat.flat_parents, flat.L, pd.leaf_codes, None, pd.pattern_freq)
    op.set_P(nodes, oracle.expm(Q, False))
    ref = op.compute_block(nodes, pi)
    with hip.HipPartition(4, flat.flat_parents, flat.L, pd.leaf_codes, None, pd.pattern_freq) as part:
        seen = []
        first = None
        t0 = time.time()
        while time.time() - t0 < 60.0:
            ll, lik, sc = part.evaluate(nodes, nodes, Q, pi, per_site=True)
            if first is None:
                first = (ll, lik.copy(), sc.copy())
            assert ll == first[0] and np.array_equal(lik, first[1]) and np.array_equal(sc, first[2])
            k = part.prune_kernel_name()
            if not seen or seen[-1] != k:
                seen.append(k)
            if k == "nucgen_kernel":
                break
            time.sleep(0.01)
        assert seen == ["prune_nuc2_kernel", "nucgen_kernel"], seen
    assert abs(first[0] - ref) <= RTOL * abs(ref), (first[0], ref)


def test_full_size_configs_generated_against_reference(monkeypatch):
    """configs[0] (HKY85, 8 x 1 000) and one partition of configs[4] (GTR, 32 x 50 000) at their stated sizes under generated kernels."""
    import zlib
    from hyphy_amd import data, hip, models
    monkeypatch.setenv("HYPHY_HIP_NUCGEN", "2")
    fx = common.load("full_gtr_32x50k_x8")
    rev = dict(zip(("AC", "AT", "CG", "CT", "GT"), (float(x) for x in fx["rev"])))
    syn = data.evolve(int(fx["taxa"]), int(fx["sites"]), 1, seed=int(fx["seed0"]), p_change=float(fx["p_change"]))
    assert (zlib.crc32(np.ascontiguousarray(syn.states.astype(np.int16)).tobytes()) & 0xffffffff) == int(fx["states_crc"][0])
    pd = data.from_states(syn.states, 4, compress_patterns=True)
    B = syn.flat.n_branches
    Q = np.stack([models.nuc_rev_Q(float(fx["t"]), rev, fx["root_freqs"])] * B)
    nodes = np.arange(B, dtype=np.int64)
    with hip.HipPartition(4, syn.flat.flat_parents, syn.flat.L, pd.leaf_codes, None, pd.pattern_freq) as part:
        for _ in range(3):
            ll, lik, sc = part.evaluate(nodes, nodes, Q, fx["root_freqs"], per_site=True)
        assert part.prune_kernel_name() == "nucgen_kernel"
    ref = float(fx["part_logl"][0])
    assert abs(ll - ref) <= RTOL * abs(ref), (ll, ref)
    got = (np.log(lik) - LOG_SCALER * sc)[pd.site_to_pattern][fx["site_index"]]
    assert np.max(np.abs(got - fx["site_logl_part0"]) / np.abs(fx["site_logl_part0"])) < RTOL
