"""Shared helpers for the parity tests: load a golden fixture and rebuild the numeric rate
matrices the reference evaluated (same templates as oracle/hbl.py wrote into the HBL)."""
import os

import numpy as np

from hyphy_amd import models

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REV_KEYS = ("AC", "AT", "CG", "CT", "GT")


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def fixture_Q(fx, cat_value=1.0):
    """[B, D, D] rate matrices (already multiplied by branch length) for a fixture."""
    rev = dict(zip(REV_KEYS, (float(x) for x in fx["rev"])))
    t = np.asarray(fx["t"], dtype=np.float64) * cat_value
    if str(fx["kind"]) == "codon":
        return models.mg94rev_Q_batch(t, float(fx["omega"]), rev, fx["pos_freqs"])
    return np.stack([models.nuc_rev_Q(float(tt), rev, fx["root_freqs"]) for tt in t])


def all_nodes(fx):
    return np.arange(len(fx["flat_parents"]) - 1, dtype=np.int64)


def busted_components(fx):
    """Rate matrices [B, 3, 61, 61] and weights [B, 3] of the unconstrained BUSTED model stored in a `ref_busted_*` fixture
    (test branches and background branches carry their own omega distribution; BS_REL.bf: P_b = sum_k w_k Exp(Q_b(omega_k)))."""
    from hyphy_amd import models
    rev = dict(zip(REV_KEYS, (float(x) for x in fx["rev"])))
    t = np.asarray(fx["t"], dtype=np.float64)
    B = len(t)
    Qc = np.zeros((B, 3, 61, 61))
    W = np.zeros((B, 3))
    by_set = {True: (fx["omega_test"], fx["weights_test"]), False: (fx["omega_background"], fx["weights_background"])}
    for b in range(B):
        om, w = by_set[bool(fx["tested"][b])]
        W[b] = w
        for k in range(3):
            Qc[b, k] = models.mg94rev_Q(t[b], float(om[k]), rev, fx["pos_freqs"])
    return Qc, W


def fubar_site_fit_args(fx, grid_points=None, sites=None):
    """Arguments of hyphy_hip_site_fits_evaluate for a `ref_fubar_*` fixture: one branch group, per-branch (synonymous, non-synonymous)
    factors, and one parameter set per grid point — every site of a set carries that point's (alpha, beta).
    Returns (templates [2, 61, 61], branch_group [B], branch_coeffs [B, 2], site_mult [n, S, 1, 2], leaf_codes [L, S], expected [n, S])."""
    from hyphy_amd import models
    rev = dict(zip(REV_KEYS, (float(x) for x in fx["rev"])), AG=1.0)
    T = np.zeros((2, 61, 61))
    for (i, j, name, ns, f) in models.mg94rev_template(fx["pos_freqs"]):
        T[1 if ns else 0, i, j] = rev[name] * f
    gp = np.arange(fx["grid"].shape[0]) if grid_points is None else np.asarray(grid_points)
    st = np.arange(fx["leaf_codes"].shape[1]) if sites is None else np.asarray(sites)
    B = len(fx["syn_factor"])
    coeffs = np.ascontiguousarray(np.stack([fx["syn_factor"], fx["nonsyn_factor"]], axis=1))
    mult = np.ascontiguousarray(np.broadcast_to(fx["grid"][gp][:, None, None, :], (len(gp), len(st), 1, 2)))
    return T, np.zeros(B, dtype=np.int64), coeffs, mult, np.ascontiguousarray(fx["leaf_codes"][:, st]), fx["site_logl"][np.ix_(gp, st)]
