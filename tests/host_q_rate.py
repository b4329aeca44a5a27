"""Rate of hyphy_hip_evaluate with HOST rate matrices / HOST probability matrices (what an unmodified HyPhy host
calls, INTEGRATION.md mode A): 125 x 61 x 61 doubles cross PCIe every evaluation.  Headline workload."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from hyphy_amd import data, hip
from oracle import oracle
wl = bench.WORKLOADS["mg94_64x10k"]
syn = data.evolve(wl["taxa"], wl["sites"], 3, seed=wl["seed"])
pd = data.from_states(syn.states, 61)
flat = syn.flat; B = flat.n_branches
T, pi = bench.templates_for(3)
part = hip.HipPartition(61, flat.flat_parents, flat.L, pd.leaf_codes, None, pd.pattern_freq)
nodes = np.arange(B, dtype=np.int64)
def Q(om):
    q = np.empty((B, 61, 61))
    for b in range(B):
        q[b] = 0.05 * (T[0] + om * T[1]); np.fill_diagonal(q[b], 0.0); np.fill_diagonal(q[b], -q[b].sum(1))
    return q
Qs = [Q(0.3 + 0.001 * k) for k in range(8)]
Ps = [oracle.expm(q, True) for q in Qs]
for name, mats, prob in (("host P (host expm, mode A)", Ps, True), ("host Q (device expm)", Qs, False), ("host P again", Ps, True), ("host Q again", Qs, False)):
    for k in range(20): part.evaluate(nodes, nodes, mats[k % 8], pi, q_is_probability=prob)
    t0 = time.perf_counter(); n = 100
    for k in range(n): ll = part.evaluate(nodes, nodes, mats[k % 8], pi, q_is_probability=prob)
    dt = time.perf_counter() - t0
    print(f"{name}: {n/dt:.0f} evals/s ({1e6*dt/n:.0f} us per call), logL {ll:.6f}")
if os.environ.get("HYPHY_HIP_ALL_TIMINGS"):
    for name, mats, prob in (("host Q", Qs, False), ("host P", Ps, True)):
        part.evaluate(nodes, nodes, mats[1], pi, q_is_probability=prob)
        print(name, "last_timings (expm, prune, reduce) ms:", part.last_timings())
