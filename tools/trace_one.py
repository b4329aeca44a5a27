import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from hyphy_amd import data, hip
wl = bench.WORKLOADS["mg94_64x10k"]
syn = data.evolve(wl["taxa"], wl["sites"], 3, seed=wl["seed"])
pd = data.from_states(syn.states, 61)
flat = syn.flat; B = flat.n_branches
T, pi = bench.templates_for(3)
part = hip.HipPartition(61, flat.flat_parents, flat.L, pd.leaf_codes, None, pd.pattern_freq)
part.set_q_templates(T)
use_torch = len(sys.argv) > 1 and sys.argv[1] == "torch"
if use_torch:
    st = torch.cuda.Stream(); torch.cuda.set_stream(st); part.set_stream(st.cuda_stream)
d = torch.zeros(2, dtype=torch.float64, device="cuda")
nodes = np.arange(B, dtype=np.int64); tb = np.full(B, 0.05); co = np.empty((B, 2)); q = part.q_buffer()
for k in range(6):
    co[:, 0] = tb; co[:, 1] = tb * (0.3 + 0.001 * k)
    part.build_q(co)
    if k >= 4: os.environ["HYPHY_HIP_TRACE"] = "1"
    part.evaluate_device(nodes, nodes, q, pi, d.data_ptr())
    part.synchronize(); torch.cuda.synchronize()
    sys.stderr.write("---\n")
