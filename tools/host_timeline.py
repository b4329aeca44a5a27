"""Host-side cost of one synchronous step (HYPHY_HIP_TRACE laps of the C-ABI calls), headline workload."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
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
    nodes = np.arange(B, dtype=np.int64); co = np.empty((B, 2)); co[:, 0] = 0.05
    step = part.prepare_built_step(nodes, nodes, pi, co)
    for k in range(40):
        co[:, 1] = 0.05 * (0.3 + 0.001 * k)
        if k == 36: os.environ["HYPHY_HIP_TRACE"] = "1"
        step()
        if k >= 36: sys.stderr.write("--- step\n")
else:
    r = subprocess.run([sys.executable, __file__, "child"], stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True)
    lines = [l for l in r.stderr.split("\n") if "trace" in l or "--- step" in l]
    print("\n".join(lines[-40:]))
