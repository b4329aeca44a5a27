"""Where does the host time of one synchronous evaluation go? (diagnostic)"""
import os, sys, time
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
st = torch.cuda.Stream(); torch.cuda.set_stream(st); part.set_stream(st.cuda_stream)
d = torch.zeros(2, dtype=torch.float64, device="cuda")
nodes = np.arange(B, dtype=np.int64); tb = np.full(B, 0.05); co = np.empty((B, 2)); q = part.q_buffer()
acc = {}
def tick(name, t0):
    t = time.perf_counter(); acc[name] = acc.get(name, 0) + t - t0; return t
for k in range(120):
    if k == 20: acc.clear()
    t = time.perf_counter()
    co[:, 0] = tb; co[:, 1] = tb * (0.3 + 0.001 * k); t = tick("numpy", t)
    if k == 110: os.environ["HYPHY_HIP_TRACE"] = "1"
    if k == 113: os.environ.pop("HYPHY_HIP_TRACE", None)
    part.build_q(co); t = tick("build_q call", t)
    part.evaluate_device(nodes, nodes, q, pi, d.data_ptr()); t = tick("evaluate_device call", t)
    v = d[0].item(); t = tick("item (sync)", t)
    part.last_timings(); t = tick("last_timings", t)
for k, v in acc.items(): print(f"{k:24s} {1e6*v/100:8.1f} us")
# host-pointer synchronous C-ABI call
import bench as b
Q = b.models.mg94rev_Q_batch(tb, 0.3, b.REV, b.POS_FREQS)
t0 = time.perf_counter()
for k in range(50): ll = part.evaluate(nodes, nodes, Q, pi)
print("hyphy_hip_evaluate (host Q, 3.7 MB H2D, sync):", 1e6 * (time.perf_counter() - t0) / 50, "us", ll)
