#!/usr/bin/env python3
"""Subtree repeats on / off held to each other on a bench workload: log L, per-pattern values, a partial update, timings.

    python tools/rep_check.py [workload] [steps]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from hyphy_amd import data, hip, models  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "mg94_64x10k"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    wl = bench.WORKLOADS[name]
    syn = data.evolve(wl["taxa"], wl["sites"], wl["unit"], seed=wl["seed"], p_change=wl.get("p_change", 0.04))
    D = 61 if wl["unit"] == 3 else 4
    pd = data.from_states(syn.states, D, compress_patterns=(D > 4))
    flat = syn.flat
    L, I, B = flat.L, flat.I, flat.n_branches
    T, pi = bench.templates_for(wl["unit"])
    tb = np.full(B, 0.05)
    nodes = np.arange(B, dtype=np.int64)
    part = hip.HipPartition(D, flat.flat_parents, L, pd.leaf_codes, None, pd.pattern_freq)
    part.set_q_templates(T)
    print("repeat stats:", part.repeat_stats(), flush=True)
    Q = models.mg94rev_Q_batch(tb, 0.3, bench.REV, bench.POS_FREQS)
    res = {}
    for on in (1, 0, 1):
        part.set_repeats(bool(on))
        ll, sl, sc = part.evaluate(nodes, nodes, Q, pi, per_site=True)
        site = np.log(sl) - 64.0 * np.log(2.0) * sc
        # a partial update: one leaf branch and one internal branch change
        Q2 = Q.copy()
        ch = np.array([3, L + 5], dtype=np.int64)
        Q2[ch] *= 1.7
        upd = np.array(sorted({3, L + 5}), dtype=np.int64)
        ll2, sl2, sc2 = part.evaluate(upd, ch, Q2[ch], pi, per_site=True)
        site2 = np.log(sl2) - 64.0 * np.log(2.0) * sc2
        llf, slf, scf = part.evaluate(nodes, nodes, Q2, pi, per_site=True)
        sitef = np.log(slf) - 64.0 * np.log(2.0) * scf
        print(f"repeats {on}: logL {ll!r}  partial {ll2!r}  full-after {llf!r}  |partial-full| {abs(ll2 - llf):.3e} per-site {np.max(np.abs(site2 - sitef)):.3e}", flush=True)
        res[on] = (ll, site, ll2, site2)
    print(f"on vs off: dlogL {abs(res[1][0] - res[0][0]) / abs(res[0][0]):.3e} rel, per-site max {np.max(np.abs(res[1][1] - res[0][1])):.3e}; "
          f"partial dlogL {abs(res[1][2] - res[0][2]) / abs(res[0][2]):.3e}, per-site {np.max(np.abs(res[1][3] - res[0][3])):.3e}", flush=True)
    coeffs = np.empty((B, 2))
    coeffs[:, 0] = tb
    for on in (0, 1):
        part.set_repeats(bool(on))
        stepf = part.prepare_built_step(nodes, nodes, pi, coeffs)
        vals = []
        for k in range(60):
            np.multiply(tb, 0.3 + 0.001 * k, out=coeffs[:, 1])
            vals.append(stepf())
        t0 = time.perf_counter()
        for k in range(steps):
            np.multiply(tb, 0.3 + 0.001 * k, out=coeffs[:, 1])
            v = stepf()
        dt = time.perf_counter() - t0
        pt = part.prune_timings(32)
        print(f"repeats {on}: {steps / dt:.0f} evals/s, {1e6 * dt / steps:.1f} us/step, pruning launches {1e3 * np.median(pt):.1f} us (median of {len(pt)}), "
              f"logL[0] {vals[0]!r}  schedule: {part.schedule_info()}", flush=True)
    part.close()


if __name__ == "__main__":
    main()
