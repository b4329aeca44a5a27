"""N > 1 path on CPU: world_size-2 gloo.  Covers the host logic that bench.py --gpus N and a
multi-rank host adapter rely on: disjoint/exhaustive pattern shards, ONE all-reduce of the
partial log-likelihood per evaluation, all-gather of per-site vectors, -inf propagation.
The per-shard arithmetic here is done by the CPU oracle (the checker) — the HIP path itself is
exercised on the GPU by tests/test_gpu_parity.py::test_sharded_partition_equals_single."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hyphy_amd import dist as hdist
from tests import common


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, name, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle
    fx = common.load(name)
    codes, freq, (lo, hi) = hdist.shard_patterns(fx["leaf_codes"], fx["pattern_freq"], rank, world)
    part = oracle.OraclePartition(int(fx["D"]), fx["flat_parents"], int(fx["L"]), codes, fx["ambig"], freq)
    nodes = common.all_nodes(fx)
    part.set_P(nodes, oracle.expm(common.fixture_Q(fx), str(fx["kind"]) == "codon"))
    lik, sc = part.site_block(nodes, fx["root_freqs"])
    site_ll = np.log(lik) - sc * 64 * np.log(2.0)
    partial = torch.tensor([float((site_ll * freq).sum())], dtype=torch.float64)
    hdist.allreduce_logl(partial)
    full = hdist.allgather_sites(torch.from_numpy(site_ll), fx["leaf_codes"].shape[1], rank, world)
    # a rank whose shard contains a zero-likelihood pattern makes the whole evaluation -inf
    bad = torch.tensor([-float("inf") if rank == 1 else -1.0], dtype=torch.float64)
    hdist.allreduce_logl(bad)
    if rank == 0:
        out_q.put((float(partial[0]), full.numpy(), float(bad[0])))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["codon_small", "nuc_small"])
def test_two_rank_site_sharding_matches_reference(name):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, name, q)) for r in range(2)]
    for p in procs:
        p.start()
    total, sites, bad = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    fx = common.load(name)
    ref = float(fx["logl"])
    assert abs(total - ref) <= 1e-11 * abs(ref)
    got = sites[fx["site_to_pattern"]]
    assert np.max(np.abs(got - fx["site_logl"]) / np.abs(fx["site_logl"])) < 1e-11
    assert bad == -np.inf


def test_shard_ranges_are_a_partition():
    for n in (1, 2, 7, 16, 1000, 9974):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                lo, hi = hdist.shard_range(n, r, world)
                assert 0 <= lo <= hi <= n
                cover.extend(range(lo, hi))
            assert cover == list(range(n))
            sizes = [hdist.shard_range(n, r, world)[1] - hdist.shard_range(n, r, world)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def _worker_categories(rank, world, port, out_q):
    """Rate classes under site sharding (bench.py --gpus N --workload busted3_64x10k): every rank mixes the classes of ITS
    patterns (weighted-sum mode is per site, likefunc2.cpp:820-853), the mixed partial log-likelihoods add up."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle
    fx = common.load("codon_cat3")
    C = len(fx["cat_weights"])
    codes, freq, _ = hdist.shard_patterns(fx["leaf_codes"], fx["pattern_freq"], rank, world)
    part = oracle.OraclePartition(int(fx["D"]), fx["flat_parents"], int(fx["L"]), codes, fx["ambig"], freq, C)
    nodes = common.all_nodes(fx)
    liks, scs = [], []
    for c in range(C):
        part.set_P(nodes, oracle.expm(common.fixture_Q(fx, float(fx["cat_values"][c])), True), cat=c)
        lik, sc = part.site_block(nodes, fx["root_freqs"], cat=c)
        liks.append(lik)
        scs.append(sc)
    ll, _, _ = oracle.mix_categories(fx["cat_weights"], np.array(liks), np.array(scs), freq)
    partial = torch.tensor([float(ll)], dtype=torch.float64)
    hdist.allreduce_logl(partial)
    if rank == 0:
        out_q.put(float(partial[0]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_site_sharding_with_rate_classes():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_categories, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    total = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = float(common.load("codon_cat3")["logl"])
    assert abs(total - ref) <= 1e-11 * abs(ref)


def _worker_classes_spread(rank, world, port, out_q):
    """Rate classes dealt over ranks (SURVEY 8e-iii second form; the reference's MPI category mode, likefunc2.cpp:595-696):
    every rank holds the whole alignment, evaluates classes c = rank mod world, one all-gather of the per-site rows, every
    rank mixes."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle
    fx = common.load("codon_cat3")
    C = len(fx["cat_weights"])
    part = oracle.OraclePartition(int(fx["D"]), fx["flat_parents"], int(fx["L"]), fx["leaf_codes"], fx["ambig"], fx["pattern_freq"], C)
    nodes = common.all_nodes(fx)
    calls = []

    def evaluate_class(c):
        calls.append(c)
        part.set_P(nodes, oracle.expm(common.fixture_Q(fx, float(fx["cat_values"][c])), True), cat=c)
        return part.site_block(nodes, fx["root_freqs"], cat=c)

    ll = hdist.evaluate_classes_spread(evaluate_class, fx["cat_weights"], fx["pattern_freq"], rank, world)
    out_q.put((rank, ll, calls))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_rate_classes_spread_over_ranks(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_classes_spread, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = float(common.load("codon_cat3")["logl"])
    for rank, ll, calls in got:
        assert abs(ll - ref) <= 1e-11 * abs(ref), (rank, ll, ref)
        assert calls == hdist.local_classes(3, rank, world)      # every class evaluated exactly once, on its owner
    assert len({ll for _, ll, _ in got}) == 1                      # the same bits on every rank


def test_mix_classes_matches_the_oracle_mixing_with_exponents_and_floor():
    """hdist.mix_classes against the restatement of PopulateConditionalProbabilities / SumUpSiteLikelihoods on rows with
    different 2^64-exponents per class and a pattern whose mixed likelihood is zero (the myLog floor)."""
    from oracle import oracle
    rng = np.random.default_rng(5)
    C, S = 4, 50
    lik = rng.uniform(1e-8, 1.0, size=(C, S))
    sc = rng.integers(0, 3, size=(C, S))
    lik[:, 7] = 0.0
    w = np.array([0.4, 0.3, 0.2, 0.1])
    f = rng.integers(1, 5, size=S)
    want, _, _ = oracle.mix_categories(w, lik, sc, f)
    got = float(hdist.mix_classes(w, torch.from_numpy(lik), torch.from_numpy(sc), f))
    assert abs(got - want) <= 1e-12 * abs(want), (got, want)
    assert hdist.local_classes(5, 1, 2) == [1, 3] and hdist.local_classes(2, 2, 3) == []


def _worker_bench_line(rank, world, port, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    # every rank contributes ITS numbers; the slowest rank's wall clock is the job's
    table = bench.gather_per_rank(dist, "cpu", world, patterns=1000 + rank, kernel_ms=0.1 * (rank + 1), expm_ms=0.02, reduce_ms=0.005,
                                  allreduce_ms=0.03 + 0.01 * rank)
    dt = bench.max_over_ranks(dist, "cpu", 1.0 + rank)
    if rank == 0:
        out_q.put((table, dt))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_line_of_a_multi_gpu_run_carries_every_ranks_numbers():
    """First-contact hardening for the 8-GPU box: the pieces of bench.py that only run at N > 1 — the
    per-rank gather, the max-over-ranks clock and the line's N > 1 fields — over gloo with two ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bench_line, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    table, dt = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    assert dt == 2.0
    assert [r["rank"] for r in table] == [0, 1] and [r["patterns"] for r in table] == [1000, 1001]
    for r in table:
        assert set(r) == {"rank", *bench.PER_RANK_KEYS}
    assert table[1]["kernel_ms"] == pytest.approx(0.2) and table[1]["allreduce_ms"] == pytest.approx(0.04)
    ab = {"cabi_ms_per_step": 0.21, "torch_ms_per_step": 0.25, "cabi_again_ms_per_step": 0.21}
    cfg, roof, top = bench.multi_gpu_fields(2, "cabi", None, table, 0.03, ab)
    assert "ncclAllReduce" in cfg["collective"] and "x2" in cfg["parallelism"] and "collective_note" not in cfg
    assert roof == {"allreduce_ms": 0.03}
    assert top["per_rank"] is table and top["collective_ab"] == ab
    cfg, roof, top = bench.multi_gpu_fields(2, "torch", "C-ABI communicator not available", table, None, None)
    assert "torch.distributed" in cfg["collective"] and cfg["collective_note"] and roof == {} and "collective_ab" not in top


def _class_form_site_lik(parents, L, codes, ambig, P, pi, comp):
    """Pruning with per-node pattern classes on the nodes `comp` marks (what class_table_kernel + the trunk kernel do; plain numpy,
    no rescaling: small trees).  Returns per-pattern likelihoods and the edge products executed (rows of P x vector)."""
    N, S = len(parents), codes.shape[1]
    I = N - L
    kids = [[] for _ in range(I)]
    for n in range(N - 1):
        kids[int(parents[n])].append(n)
    cls = [None] * I      # class id per pattern (compressed nodes)
    edge = [None] * I     # E_n = P_n x (conditionals of n): [U_n][D] per class (compressed: what a class table holds) or [S][D]
    work = 0
    root = None

    def leaf_col(n, c):
        return P[n][:, c] if c >= 0 else P[n] @ ambig[-c - 1]

    for i in range(I):
        if comp[i]:       # children are leaves or compressed nodes: classes of the tuples of their class ids / codes
            key = np.stack([codes[c] if c < L else cls[c - L] for c in kids[i]], axis=1)
            uniq, inv = np.unique(key, axis=0, return_inverse=True)
            cls[i] = inv.reshape(-1)
            v = np.ones((len(uniq), P.shape[1]))
            for k, c in enumerate(kids[i]):
                for u in range(len(uniq)):
                    v[u] *= leaf_col(c, int(uniq[u][k])) if c < L else edge[c - L][int(uniq[u][k])]
        else:
            v = np.ones((S, P.shape[1]))
            for c in kids[i]:
                if c < L:
                    for s_ in range(S):
                        v[s_] *= leaf_col(c, int(codes[c, s_]))
                else:
                    v *= edge[c - L][cls[c - L]] if comp[c - L] else edge[c - L]   # a class table is gathered by class id
        if i == I - 1:
            root = v
        else:
            edge[i] = v @ P[L + i].T   # one edge product per class (compressed) / per pattern
            work += len(v)
    return root @ pi, work


def _worker_repeats(rank, world, port, name, theta, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle
    from hyphy_amd import hip
    fx = common.load(name)
    L = int(fx["L"])
    codes, freq, (lo, hi) = hdist.shard_patterns(fx["leaf_codes"], fx["pattern_freq"], rank, world)
    # every rank plans the classes of ITS patterns (host-only entry point of the C-ABI: no device needed)
    classes, comp, planned = hip.plan_repeats(fx["flat_parents"], L, codes, theta)
    P = oracle.expm(common.fixture_Q(fx), str(fx["kind"]) == "codon")
    lik, work = _class_form_site_lik(np.asarray(fx["flat_parents"]), L, np.asarray(codes), np.asarray(fx["ambig"]), P,
                                     np.asarray(fx["root_freqs"]), comp)
    site_ll = np.log(lik)
    partial = torch.tensor([float((site_ll * freq).sum())], dtype=torch.float64)
    hdist.allreduce_logl(partial)
    full = hdist.allgather_sites(torch.from_numpy(site_ll), fx["leaf_codes"].shape[1], rank, world)
    counts = torch.tensor([float(work), float(planned), float((len(fx["flat_parents"]) - L - 1) * codes.shape[1]), float(comp.sum())],
                          dtype=torch.float64)
    dist.all_reduce(counts)
    if rank == 0:
        out_q.put((float(partial[0]), full.numpy(), counts.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,theta", [("codon_wide", 0.9), ("codon_small", 1.0), ("nuc_small", 0.5)])
def test_two_rank_site_sharding_with_subtree_repeats(name, theta):
    """Subtree repeats under pattern sharding (one process per GPU): each rank plans the classes of ITS shard through the C-ABI's
    host-only hyphy_hip_plan_repeats, evaluates in class form, and the one all-reduce of the partial log-likelihoods / the
    all-gather of the per-pattern values give the reference's numbers; the executed edge products the plans announce are the
    ones the class form performs, and fewer than every-pattern-at-every-node."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_repeats, args=(r, 2, port, name, theta, q)) for r in range(2)]
    for p in procs:
        p.start()
    total, sites, counts = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    fx = common.load(name)
    ref = float(fx["logl"])
    assert abs(total - ref) <= 1e-10 * abs(ref)
    got = sites[fx["site_to_pattern"]]
    assert np.max(np.abs(got - fx["site_logl"]) / np.abs(fx["site_logl"])) < 1e-10
    work, planned, full, n_comp = counts
    assert work == planned and n_comp > 0 and work < full, counts


# ---------------------------------------------------------------------------------------------------------------------
# r06: the collective-free combine of one-process-per-GPU runs (hyphy_hip_xch_*, comm.hip: HostExchange) — host-only code
# (POSIX shared memory), so the REAL library runs it here: world 2 and 3, against gloo's all-reduce of the same partials.
# ---------------------------------------------------------------------------------------------------------------------
def _xch_worker(rank, world, port, name, tag, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hyphy_amd import hip
    from oracle import oracle
    fx = common.load(name)
    codes, freq, (lo, hi) = hdist.shard_patterns(fx["leaf_codes"], fx["pattern_freq"], rank, world)
    part = oracle.OraclePartition(int(fx["D"]), fx["flat_parents"], int(fx["L"]), codes, fx["ambig"], freq)
    nodes = common.all_nodes(fx)
    part.set_P(nodes, oracle.expm(common.fixture_Q(fx), str(fx["kind"]) == "codon"))
    x = hip.HostExchange(tag, rank, world)          # returns when every rank has attached
    got, want = [], []
    for it in range(40):                            # many exchanges back to back: the slots are reused every other one
        local = part.compute_block(nodes, fx["root_freqs"]) * (1.0 + 0.01 * it) + rank * 1e-3
        got.append(x.sum(local))
        t = torch.tensor([local], dtype=torch.float64)
        dist.all_reduce(t)
        want.append(float(t[0]))
    # a rank whose local evaluation failed posts NaN: every rank sees NaN, nobody waits
    nan_seen = x.sum(1.0, failed=(rank == world - 1))
    after = x.sum(float(rank + 1))                  # ... and the exchange goes on
    x.close()
    everyone = [None] * world
    dist.all_gather_object(everyone, got)           # same bits on every rank?
    if rank == 0:
        out_q.put((got, want, nan_seen, after, everyone))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_host_exchange_sums_like_an_allreduce(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    tag = f"test_{os.getpid()}_{port}"
    procs = [ctx.Process(target=_xch_worker, args=(r, world, port, "nuc_small", tag, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, want, nan_seen, after, everyone = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.allclose(got, want, rtol=1e-14, atol=0.0)
    assert all(e == got for e in everyone)          # bit-identical on every rank (summed in rank order)
    assert np.isnan(nan_seen)
    assert after == sum(range(1, world + 1))
