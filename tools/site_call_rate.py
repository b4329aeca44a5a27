#!/usr/bin/env python3
"""Cost of a synchronous evaluation that also returns per-pattern values and exponents (what a host that mixes rate classes itself
asks for once per class), against the same evaluation without them.  usage: tools/site_call_rate.py [workload] [steps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from hyphy_amd import data, hip  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "mg94_64x10k"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
wl = bench.WORKLOADS[name]
syn = data.evolve(wl["taxa"], wl["sites"], wl["unit"], seed=wl["seed"], p_change=wl.get("p_change", 0.04))
D = 61 if wl["unit"] == 3 else 4
pd = data.from_states(syn.states, D, compress_patterns=(D > 4))
flat = syn.flat
B = flat.n_branches
T, pi = bench.templates_for(wl["unit"])
tb = np.full(B, 0.05)
nodes = np.arange(B, dtype=np.int64)
part = hip.HipPartition(D, flat.flat_parents, flat.L, pd.leaf_codes, None, pd.pattern_freq)
part.set_q_templates(T)
coeffs = np.empty((B, 2))
coeffs[:, 0] = tb
for per_site in (False, True, False, True):
    for k in range(80 + steps):
        if k == 80:
            t0 = time.perf_counter()
        np.multiply(tb, 0.3 + 0.001 * k, out=coeffs[:, 1])
        part.build_q(coeffs)
        r = part.evaluate_built(nodes, nodes, pi, per_site=per_site)
    dt = time.perf_counter() - t0
    print(f"per_site={per_site}: {1e6 * dt / steps:.1f} us per call (python wrapper included), export "
          f"{os.environ.get('HYPHY_HIP_SITE_EXPORT', '1')}, logL {r[0] if per_site else r!r}", flush=True)
part.close()
