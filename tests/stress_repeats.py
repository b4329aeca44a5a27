"""Randomised GPU stress test of the class-compressed form under the row-split walks (repeats.hip: class_table_team_kernel +
trunk_walk_kernel, one or two workgroups per tile): random trees and alignment sizes, random thresholds (the trunk from one node to most
of the tree), random sequences of full passes with new matrices, partial updates and per-pattern evaluations — every result held to a
second partition of the same data that runs the plain form.  Usage (GPU box): python tests/stress_repeats.py [n_cases] [seed0]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
from hyphy_amd import data, hip, models   # noqa: E402

PF = np.array([[0.3, 0.2, 0.25, 0.25], [0.2, 0.3, 0.3, 0.2], [0.25, 0.25, 0.2, 0.3]])
REV = dict(AC=0.5, AT=0.4, CG=0.4, CT=1.2, GT=0.4, AG=1.0)
LOG_SCALER = 64.0 * np.log(2.0)
t0 = time.time()
checks = walks = 0
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    taxa, codons = int(rng.integers(6, 56)), int(rng.integers(300, 4000))
    theta, chains = float(rng.choice([0.05, 0.15, 0.3, 0.6, 0.9])), str(int(rng.choice([1, 2])))
    os.environ.update(HYPHY_HIP_REPEATS="2", HYPHY_HIP_REP_THETA=str(theta), HYPHY_HIP_TRUNK_WALK="1", HYPHY_HIP_WALK_CHAINS=chains,
                      HYPHY_HIP_POISON="1", HYPHY_HIP_REP_RHO=str(float(rng.choice([0.0, 0.0, 0.5]))))
    syn = data.evolve(taxa, codons, 3, seed=seed0 + case, p_change=float(rng.choice([0.02, 0.05, 0.15])))
    pd = data.from_states(syn.states, 61)
    flat = syn.flat
    B = flat.n_branches
    pi = models.f3x4_codon_freqs(PF)
    tmpl = models.mg94rev_template(PF)

    def Q_for(tb, omega):
        Q = np.zeros((len(tb), 61, 61))
        for (i, j, name, ns, f) in tmpl:
            Q[:, i, j] = tb * REV[name] * f * (omega if ns else 1.0)
        Q[:, np.arange(61), np.arange(61)] = -Q.sum(axis=2)
        return Q

    nodes = np.arange(B, dtype=np.int64)
    tb = rng.uniform(0.01, 0.4, B)
    with hip.HipPartition(61, flat.flat_parents, flat.L, pd.leaf_codes, None, pd.pattern_freq) as part, \
            hip.HipPartition(61, flat.flat_parents, flat.L, pd.leaf_codes, None, pd.pattern_freq) as plain:
        plain.set_repeats(False)
        on = part.repeat_stats()["in_use"] == 1
        Q = Q_for(tb, 0.5)
        for op in range(10):
            kind = "full" if op < 2 else str(rng.choice(["full", "full", "partial", "again"]))
            if kind == "full":
                tb = tb * rng.uniform(0.8, 1.25, B)
                Q = Q_for(tb, float(rng.uniform(0.2, 1.5)))
                un = qn = nodes
                Qs = Q
            elif kind == "partial":
                ch = np.sort(rng.choice(B, size=int(rng.integers(1, 4)), replace=False)).astype(np.int64)
                tb[ch] *= rng.uniform(0.5, 1.8)
                Q[ch] = Q_for(tb[ch], 0.7)
                un = qn = ch
                Qs = Q[ch]
            else:
                un = qn = nodes
                Qs = Q
            a, la, sa = part.evaluate(un, qn, Qs, pi, per_site=True)
            b, lb, sb = plain.evaluate(un, qn, Qs, pi, per_site=True)
            assert abs(a - b) <= 1e-12 * abs(b), (case, op, kind, a, b, taxa, codons, theta, chains)
            d = np.max(np.abs((np.log(la) - sa * LOG_SCALER) - (np.log(lb) - sb * LOG_SCALER)))
            assert d < 1e-10, (case, op, kind, d)
            checks += 1
            walks += part.prune_kernel_name() == "trunk_walk_kernel"
    print(f"case {case}: {taxa} taxa x {codons} codons, theta {theta}, chains {chains}, compressed {on}: ok", flush=True)
print(f"{n_cases} cases, {checks} evaluations checked ({walks} of them ended in trunk_walk_kernel) in {time.time() - t0:.0f} s")
