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
