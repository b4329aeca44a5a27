import os, sys, time, ctypes as C
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
lib = hip.load()
mode = sys.argv[1] if len(sys.argv) > 1 else "torch"
d = torch.zeros(2, dtype=torch.float64, device="cuda")
if mode == "torch":
    st = torch.cuda.Stream(); torch.cuda.set_stream(st); part.set_stream(st.cuda_stream)
nodes = np.arange(B, dtype=np.int64); tb = np.full(B, 0.05); co = np.empty((B, 2)); q = part.q_buffer()
pnodes = nodes.ctypes.data_as(C.POINTER(C.c_int64)); ppi = pi.ctypes.data_as(C.POINTER(C.c_double))
pco = co.ctypes.data_as(C.POINTER(C.c_double)); h = part._h; dptr = C.c_void_p(d.data_ptr()); qptr = C.c_void_p(q)
acc = {}
def tick(name, t0):
    t = time.perf_counter(); acc[name] = acc.get(name, 0) + t - t0; return t
host_out = np.zeros(2)
for k in range(220):
    if k == 20: acc.clear()
    t = time.perf_counter()
    co[:, 0] = tb; co[:, 1] = tb * (0.3 + 0.001 * k); t = tick("numpy", t)
    lib.hyphy_hip_build_q(h, B, pco); t = tick("build_q raw", t)
    lib.hyphy_hip_evaluate_device(h, -1, pnodes, B, pnodes, B, qptr, 0, ppi, dptr); t = tick("evaluate_device raw", t)
    if mode == "torch":
        v = d[0].item(); t = tick("item (sync)", t)
    else:
        lib.hyphy_hip_synchronize(h); t = tick("synchronize", t)
for k, v in acc.items(): print(f"{k:24s} {1e6*v/200:8.1f} us")
print("total per step us", 1e6 * sum(acc.values()) / 200)
