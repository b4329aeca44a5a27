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
