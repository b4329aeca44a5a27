import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench
from hyphy_amd import data, hip
wl = bench.WORKLOADS["mg94_64x10k"]
syn = data.evolve(wl["taxa"], wl["sites"], 3, seed=wl["seed"])
pd = data.from_states(syn.states, 61)
flat = syn.flat; B = flat.n_branches
T, pi = bench.templates_for(3)
part = hip.HipPartition(61, flat.flat_parents, flat.L, pd.leaf_codes, None, pd.pattern_freq)
nodes = np.arange(B, dtype=np.int64)
q = np.empty((B, 61, 61))
for b in range(B):
    q[b] = 0.05 * (T[0] + 0.3 * T[1]); np.fill_diagonal(q[b], 0.0); np.fill_diagonal(q[b], -q[b].sum(1))
for k in range(6):
    if k == 4: os.environ["HYPHY_HIP_TRACE"] = "1"
    part.evaluate(nodes, nodes, q, pi)
    sys.stderr.write("--- call\n")
