"""The N > 1 forms of the path on real devices, through the C-ABI — skipped on boxes with fewer than two GPUs:
 * one process per GPU: hyphy_hip_comm_unique_id on rank 0 -> the 128 bytes to every rank -> hyphy_hip_comm_init_rank ->
   hyphy_hip_build_q + hyphy_hip_evaluate_built_allreduce (what `bench.py --gpus N` times);
 * one process, N devices: hyphy_hip_create(device_count = N), shard partials combined on the host or by one RCCL group
   all-reduce (HYPHY_HIP_COMBINE=rccl behind hyphy_hip_comm_init_all).
Both against the single-device evaluation of the whole alignment."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _hip(min_devices):
    from hyphy_amd import hip
    if hip.device_count() < min_devices:
        pytest.skip(f"needs {min_devices} GPUs")
    return hip


def _single_device_values(hip):
    import multi_gpu_rank as mg
    syn, pd, T, pi, tb = mg.build_case()
    nodes = np.arange(syn.flat.n_branches, dtype=np.int64)
    vals = []
    with hip.HipPartition(61, syn.flat.flat_parents, syn.flat.L, pd.leaf_codes, None, pd.pattern_freq) as part:
        part.set_q_templates(T)
        co = mg.coeffs_for(tb, mg.OMEGAS[0])
        step = part.prepare_built_step(nodes, nodes, pi, co)
        for om in mg.OMEGAS:
            co[:] = mg.coeffs_for(tb, om)
            vals.append(step())
    return vals


def _run_ranks(world, tmp_path, extra_env=None):
    uid = str(tmp_path / "uid.bin")
    procs, outs = [], []
    for r in range(world):
        out = str(tmp_path / f"rank{r}.json")
        outs.append(out)
        env = dict(os.environ, **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "multi_gpu_rank.py"), str(r), str(world), uid, out],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=600)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("a rank hung (collective not matched?)")
    for p, lg in zip(procs, logs):
        assert p.returncode == 0, lg[-3000:]
    return [json.load(open(o)) for o in outs]


@pytest.mark.parametrize("world", [2, 4, 8])
def test_one_process_per_gpu_allreduce_through_the_c_abi(world, tmp_path):
    hip = _hip(world)
    want = _single_device_values(hip)
    res = _run_ranks(world, tmp_path)
    for r in res:
        got = np.array(r["values"]).reshape(len(want), 2)
        for k, w in enumerate(want):
            assert abs(got[k, 0] - w) <= 1e-12 * abs(w) and abs(got[k, 1] - w) <= 1e-12 * abs(w), (r["rank"], k, got[k], w)
    # every rank returns the same bits (one all-reduce result)
    assert all(res[0]["values"] == r["values"] for r in res[1:])


def test_a_failing_rank_does_not_leave_the_others_in_the_collective(tmp_path):
    _hip(2)
    res = _run_ranks(2, tmp_path, extra_env={"FAIL_RANK": "1"})
    by_rank = {r["rank"]: r["failure_case"] for r in res}
    assert by_rank[1].startswith("error:") and "twice" in by_rank[1], by_rank
    assert by_rank[0] == "nan", by_rank     # rank 0's own evaluation was fine; the sum carries rank 1's NaN


@pytest.mark.parametrize("world", [2, 3])
def test_host_exchange_between_processes_sharing_one_device(world, tmp_path):
    """r06: the collective-free combine (hyphy_hip_comm_init_host + hyphy_hip_evaluate_built_exchange) needs no device collective, so
    a one-GPU box runs it for real: `world` PROCESSES, each with its own partition over its pattern shard (all on device 0), every
    evaluation ends in one shared-memory exchange — the whole alignment's log-likelihood on every rank, same bits."""
    hip = _hip(1)
    want = _single_device_values(hip)
    res = _run_ranks(world, tmp_path, extra_env={"HOST_EXCHANGE": "share"})
    for r in res:
        got = np.array(r["values"]).reshape(len(want), 2)
        for k, w in enumerate(want):
            assert abs(got[k, 0] - w) <= 1e-12 * abs(w) and abs(got[k, 1] - w) <= 1e-12 * abs(w), (r["rank"], k, got[k], w)
    assert all(res[0]["values"] == r["values"] for r in res[1:])


def test_host_exchange_a_failing_rank_does_not_leave_the_others_waiting(tmp_path):
    _hip(1)
    res = _run_ranks(2, tmp_path, extra_env={"HOST_EXCHANGE": "share", "FAIL_RANK": "1"})
    by_rank = {r["rank"]: r["failure_case"] for r in res}
    assert by_rank[1].startswith("error:") and "twice" in by_rank[1], by_rank
    assert by_rank[0] == "nan", by_rank     # rank 0's own evaluation was fine; the sum carries rank 1's NaN


@pytest.mark.parametrize("world", [2, 4, 8])
def test_one_process_per_gpu_host_exchange(world, tmp_path):
    hip = _hip(world)
    want = _single_device_values(hip)
    res = _run_ranks(world, tmp_path, extra_env={"HOST_EXCHANGE": "1"})
    for r in res:
        got = np.array(r["values"]).reshape(len(want), 2)
        for k, w in enumerate(want):
            assert abs(got[k, 0] - w) <= 1e-12 * abs(w) and abs(got[k, 1] - w) <= 1e-12 * abs(w), (r["rank"], k, got[k], w)
    assert all(res[0]["values"] == r["values"] for r in res[1:])


@pytest.mark.parametrize("combine", ["host", "rccl"])
@pytest.mark.parametrize("n_dev", [2, 8])
def test_single_process_multi_device_combine(n_dev, combine, monkeypatch):
    hip = _hip(n_dev)
    import multi_gpu_rank as mg
    want = _single_device_values(hip)
    monkeypatch.setenv("HYPHY_HIP_COMBINE", combine)
    syn, pd, T, pi, tb = mg.build_case()
    nodes = np.arange(syn.flat.n_branches, dtype=np.int64)
    with hip.HipPartition(61, syn.flat.flat_parents, syn.flat.L, pd.leaf_codes, None, pd.pattern_freq, device_count=n_dev) as part:
        part.set_q_templates(T)
        if combine == "rccl":
            part.comm_init_all()
        co = mg.coeffs_for(tb, mg.OMEGAS[0])
        step = part.prepare_built_step(nodes, nodes, pi, co)
        for om, w in zip(mg.OMEGAS, want):
            co[:] = mg.coeffs_for(tb, om)
            for _ in range(2):
                got = step()
                assert abs(got - w) <= 1e-12 * abs(w), (n_dev, combine, om, got, w)
        # host-supplied matrices through hyphy_hip_evaluate take the same combine
        Q = np.einsum("bk,kij->bij", co, T)
        idx = np.arange(61)
        Q[:, idx, idx] = 0.0
        Q[:, idx, idx] = -Q.sum(2)
        got = part.evaluate(nodes, nodes, Q, pi)
        assert abs(got - want[-1]) <= 1e-12 * abs(want[-1])


def test_one_rank_communicator_on_one_gpu():
    """What a single-GPU box can check of the N > 1 step: the same entry point on a one-rank communicator."""
    hip = _hip(1)
    import multi_gpu_rank as mg
    want = _single_device_values(hip)
    syn, pd, T, pi, tb = mg.build_case()
    nodes = np.arange(syn.flat.n_branches, dtype=np.int64)
    with hip.HipPartition(61, syn.flat.flat_parents, syn.flat.L, pd.leaf_codes, None, pd.pattern_freq) as part:
        part.set_q_templates(T)
        part.comm_init_rank(hip.HipPartition.comm_unique_id(), 0, 1)
        co = mg.coeffs_for(tb, mg.OMEGAS[0])
        step = part.prepare_built_allreduce_step(nodes, nodes, pi, co)
        for om, w in zip(mg.OMEGAS, want):
            co[:] = mg.coeffs_for(tb, om)
            for _ in range(2):
                got = step()
                assert abs(got - w) <= 1e-13 * abs(w), (om, got, w)
        part.set_all_timings(True)
        step()
        assert part.last_allreduce_ms() > 0.0
        part.set_all_timings(False)
        bad = nodes.copy()
        bad[1] = bad[0]
        with pytest.raises(hip.HipError, match="twice"):      # a local failure still completes the collective and reports itself
            part.prepare_built_allreduce_step(nodes, bad, pi, co)()
        assert abs(step() - want[-1]) <= 1e-13 * abs(want[-1])  # ... and the partition is usable afterwards


def test_bench_two_ranks_sharing_one_device_walk_the_multi_rank_code(tmp_path):
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one rank per GPU) has never met a multi-GPU box.
    HYPHY_BENCH_SHARE_DEVICE=1 (diagnostic) puts both ranks on device 0 over gloo, so a 1-GPU box runs the N > 1 code of
    bench.py — pattern sharding, the per-evaluation sum across ranks, barrier/max timing, the per-rank gather, the line — and
    the summed log-L must equal the single-device value of the same sweep point."""
    _hip(1)
    import socket
    common = ["--workload", "mg94_64x1250", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-traffic", "--preheat-s", "0", "--cold-s", "0"]
    env = dict(os.environ, HYPHY_BENCH_SHARE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    for attempt in range(4):  # (the port is free when it is picked, not necessarily when the rendezvous binds it: EADDRINUSE -> another one)
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                              "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common,
                             env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
        if two.returncode == 0 or "address already in use" not in (two.stdout + two.stderr):
            break
    assert two.returncode == 0, (two.stdout + two.stderr)[-3000:]
    line2 = json.loads([ln for ln in two.stdout.split("\n") if ln.startswith("{")][-1])
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common, env=dict(os.environ),
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert one.returncode == 0, (one.stdout + one.stderr)[-3000:]
    line1 = json.loads([ln for ln in one.stdout.split("\n") if ln.startswith("{")][-1])
    assert line2["n_gpus"] == 2 and "DIAGNOSTIC" in line2["config"]
    assert len(line2["per_rank"]) == 2 and sum(r["patterns"] for r in line2["per_rank"]) == line1["config"]["patterns_rank0"]
    assert all(r["kernel_ms"] > 0 for r in line2["per_rank"])
    assert abs(line2["logl_last"] - line1["logl_last"]) <= 1e-12 * abs(line1["logl_last"]), (line2["logl_last"], line1["logl_last"])
    assert line2["scaling"] == "strong" and line2["value"] > 0
