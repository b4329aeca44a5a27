"""One rank of the multi-GPU C-ABI test (tests/test_gpu_multi.py): python tests/multi_gpu_rank.py RANK WORLD UID_FILE OUT_FILE
Rank r owns pattern shard r on device r; the RCCL unique id made by rank 0 travels through UID_FILE (a host with MPI would
broadcast it); every rank evaluates three parameter points through hyphy_hip_build_q + hyphy_hip_evaluate_built_allreduce and
writes the values it got — the log-likelihood of the WHOLE alignment — to OUT_FILE.  The last point is evaluated with a
deliberately invalid root-frequency pointer on rank 1 only when FAIL_RANK is set: the other ranks must not hang."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from hyphy_amd import data, hip, models
from hyphy_amd import dist as hdist


def build_case():
    syn = data.evolve(24, 900, 3, seed=17)
    pd = data.from_states(syn.states, 61)
    pf = np.array([[0.3, 0.2, 0.25, 0.25], [0.2, 0.3, 0.3, 0.2], [0.25, 0.25, 0.2, 0.3]])
    rv = dict(AC=0.5, AT=0.4, CG=0.4, CT=1.2, GT=0.4, AG=1.0)
    T = np.zeros((2, 61, 61))
    for (i, j, name, ns, f) in models.mg94rev_template(pf):
        T[1 if ns else 0, i, j] = rv[name] * f
    pi = models.f3x4_codon_freqs(pf)
    rng = np.random.default_rng(3)
    tb = rng.uniform(0.02, 0.3, syn.flat.n_branches)
    return syn, pd, T, pi, tb


def coeffs_for(tb, omega):
    return np.ascontiguousarray(np.stack([tb, tb * omega], axis=1))


OMEGAS = (0.25, 0.6, 1.4)

if __name__ == "__main__":
    rank, world, uid_file, out_file = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    syn, pd, T, pi, tb = build_case()
    codes, freq, _ = hdist.shard_patterns(pd.leaf_codes, pd.pattern_freq, rank, world)
    flat = syn.flat
    nodes = np.arange(flat.n_branches, dtype=np.int64)
    exchange = os.environ.get("HOST_EXCHANGE") is not None   # the collective-free combine (hyphy_hip_comm_init_host): no RCCL, and
                                                             # the ranks may share a device (HOST_EXCHANGE=share: all on device 0)
    if exchange:
        uid = None
    elif rank == 0:
        uid = hip.HipPartition.comm_unique_id()
        with open(uid_file + ".tmp", "wb") as fh:
            fh.write(uid)
        os.replace(uid_file + ".tmp", uid_file)
    else:
        t0 = time.time()
        while not os.path.exists(uid_file):
            if time.time() - t0 > 120:
                raise SystemExit("no unique id")
            time.sleep(0.05)
        uid = open(uid_file, "rb").read()
    dev = 0 if os.environ.get("HOST_EXCHANGE") == "share" else rank
    with hip.HipPartition(61, flat.flat_parents, flat.L, codes, None, freq, device_first=dev) as part:
        part.set_q_templates(T)
        if exchange:
            part.comm_init_host("t_" + os.path.basename(os.path.dirname(uid_file)) + "_" + os.path.basename(uid_file), rank, world)
            prepare = part.prepare_built_exchange_step
        else:
            part.comm_init_rank(uid, rank, world)
            prepare = part.prepare_built_allreduce_step
        co = coeffs_for(tb, OMEGAS[0])
        step = prepare(nodes, nodes, pi, co)
        vals = []
        for om in OMEGAS:
            co[:] = coeffs_for(tb, om)
            vals.append(step())
            vals.append(step())        # steady state (lazy persistence, tuned schedule)
        err = None
        if os.environ.get("FAIL_RANK") is not None:
            # one rank fails locally (a branch listed twice): it must still join the collective, the others see NaN
            bad_nodes = nodes.copy()
            if rank == int(os.environ["FAIL_RANK"]):
                bad_nodes[1] = bad_nodes[0]
            try:
                v = prepare(nodes, bad_nodes, pi, co)()
                err = "nan" if v != v else f"value {v!r}"
            except hip.HipError as e:
                err = f"error: {e}"
    json.dump({"rank": rank, "values": vals, "failure_case": err}, open(out_file, "w"))
