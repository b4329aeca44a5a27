"""Numeric rate matrices for the substitution models the benchmark configs name.

In the reference the numeric Q of a branch is produced by the HBL formula engine
(``_CalcNode::RecomputeMatrix`` ``calcnode.cpp:526-704`` → ``_Matrix::MultByFreqs``
``matrix.cpp:1546-1677``) — serial scalar host work that is *out of scope* (SURVEY §2);
its output, one dense D×D matrix per branch **already multiplied by the branch length**,
is the input of the hot path (``_Matrix::Exponentiate(1., true, …)`` ``calcnode.cpp:729``).
This module restates the few model templates the configs use so tests and ``bench.py`` can
produce those inputs without the reference; the same templates are written out as HBL by
``oracle/hbl.py`` so the real reference evaluates *identical* matrices.

Codon state order = HyPhy's: 64 codons in base-4 order with A,C,G,T = 0..3
(index = 16·n1 + 4·n2 + n3), stop codons removed (``CreateFilter(ds,3,"","","TAA,TAG,TGA")``
gives D = 61).
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np

NUC = "ACGT"
_AA = "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSS*CWCLFLF"  # universal code, ACGT order

# REV exchangeability parameter names; AG is the reference rate (=1) as in HyPhy's REV.
REV_NAMES = {(0, 1): "AC", (0, 2): "AG", (0, 3): "AT", (1, 2): "CG", (1, 3): "CT", (2, 3): "GT"}


def sense_codons() -> List[str]:
    out = []
    for i in range(64):
        c = NUC[i >> 4] + NUC[(i >> 2) & 3] + NUC[i & 3]
        if _AA[i] != "*":
            out.append(c)
    return out


def codon_aa() -> List[str]:
    return [_AA[i] for i in range(64) if _AA[i] != "*"]


SENSE = sense_codons()
SENSE_AA = codon_aa()
CODON_INDEX: Dict[str, int] = {c: i for i, c in enumerate(SENSE)}


def codon_neighbors() -> List[Tuple[int, int, int, int, int, bool]]:
    """(i, j, pos, from_nt, to_nt, nonsynonymous) for all single-nucleotide codon pairs."""
    out = []
    for i, a in enumerate(SENSE):
        for j, b in enumerate(SENSE):
            if i == j:
                continue
            diff = [p for p in range(3) if a[p] != b[p]]
            if len(diff) != 1:
                continue
            p = diff[0]
            out.append((i, j, p, NUC.index(a[p]), NUC.index(b[p]), SENSE_AA[i] != SENSE_AA[j]))
    return out


_NEIGH = codon_neighbors()


def f3x4_codon_freqs(pos_freqs: np.ndarray) -> np.ndarray:
    """π_codon ∝ Π_pos π_pos[nt], renormalised over the 61 sense codons."""
    pf = np.asarray(pos_freqs, dtype=np.float64).reshape(3, 4)
    pi = np.array([pf[0, NUC.index(c[0])] * pf[1, NUC.index(c[1])] * pf[2, NUC.index(c[2])] for c in SENSE])
    return pi / pi.sum()


def finish_rate_matrix(Q: np.ndarray) -> np.ndarray:
    """Diagonal = −Σ off-diagonals, subtracted in column order like the dense branch of
    ``_Matrix::MultByFreqs`` (``matrix.cpp:1664-1674``)."""
    D = Q.shape[0]
    for r in range(D):
        d = 0.0
        for c in range(D):
            if c != r:
                d -= Q[r, c]
        Q[r, r] = d
    return Q


def mg94rev_template(pos_freqs: np.ndarray) -> List[Tuple[int, int, str, bool, float]]:
    """Sparse template entries (i, j, rev_name, nonsyn, π_pos[to]) of MG94×REV:
    Q_ij = θ_xy · (ω if nonsynonymous) · t · π^{pos}_y   for single-nucleotide changes
    (libv3 ``models/codon/MG_REV.bf``; the constants play the role of the frequency factors
    written into ``tests/hbltests/SimpleOptimizations/SmallCodon.bf``'s matrix)."""
    pf = np.asarray(pos_freqs, dtype=np.float64).reshape(3, 4)
    out = []
    for (i, j, p, x, y, ns) in _NEIGH:
        name = REV_NAMES[(min(x, y), max(x, y))]
        out.append((i, j, name, ns, float(pf[p, y])))
    return out


def mg94rev_Q(t: float, omega: float, rev: Dict[str, float], pos_freqs: np.ndarray) -> np.ndarray:
    """Dense 61×61 numeric rate matrix for one branch.  Product order
    ((θ·ω)·t)·π matches the HBL formula text written by ``oracle/hbl.py``
    (``AC*R*t*const``, evaluated left to right) so the real reference sees bit-identical
    entries."""
    Q = np.zeros((61, 61), dtype=np.float64)
    rv = dict(rev)
    rv.setdefault("AG", 1.0)
    for (i, j, name, ns, pf) in mg94rev_template(pos_freqs):
        if name == "AG":
            v = (omega * t) * pf if ns else t * pf
        else:
            v = ((rv[name] * omega) * t) * pf if ns else (rv[name] * t) * pf
        Q[i, j] = v
    return finish_rate_matrix(Q)


def mg94rev_Q_batch(ts: Sequence[float], omega: float, rev: Dict[str, float], pos_freqs: np.ndarray) -> np.ndarray:
    """[B, 61, 61] rate matrices, one per branch length (vectorised; same product order)."""
    ts = np.asarray(ts, dtype=np.float64)
    B = ts.shape[0]
    Q = np.zeros((B, 61, 61), dtype=np.float64)
    rv = dict(rev)
    rv.setdefault("AG", 1.0)
    for (i, j, name, ns, pf) in mg94rev_template(pos_freqs):
        if name == "AG":
            v = (omega * ts) * pf if ns else ts * pf
        else:
            v = ((rv[name] * omega) * ts) * pf if ns else (rv[name] * ts) * pf
        Q[:, i, j] = v
    # diagonal: sequential subtraction in column order (vectorised over branches)
    d = np.zeros(B)
    for r in range(61):
        d[:] = 0.0
        for c in range(61):
            if c != r:
                d -= Q[:, r, c]
        Q[:, r, r] = d
    return Q


def nuc_rev_Q(t: float, rev: Dict[str, float], freqs: np.ndarray) -> np.ndarray:
    """GTR/REV (and HKY85 as the special case AC=AT=CG=GT=1/κ·…): Q_ij = θ_ij·t·π_j.
    Mirrors ``Model M = (Q, freqs, 1)``: template entry ``θ*t`` then ``MultByFreqs``."""
    rv = dict(rev)
    rv.setdefault("AG", 1.0)
    Q = np.zeros((4, 4), dtype=np.float64)
    for i in range(4):
        for j in range(4):
            if i == j:
                continue
            name = REV_NAMES[(min(i, j), max(i, j))]
            base = t if name == "AG" else rv[name] * t
            Q[i, j] = base * freqs[j]
    return finish_rate_matrix(Q)


def hky85_rev(kappa: float) -> Dict[str, float]:
    """HKY85 as REV with transversions scaled by 1/κ relative to transitions (AG, CT = 1):
    the parameterisation of ``tests/hbltests/SimpleOptimizations/IntermediateNuc.bf``
    (``global kappa``; trv = kappa·trst)."""
    return {"AC": kappa, "AT": kappa, "CG": kappa, "GT": kappa, "CT": 1.0}


def expected_subs_per_site(Q: np.ndarray, pi: np.ndarray) -> float:
    return float(-(np.diag(Q) * pi).sum())
