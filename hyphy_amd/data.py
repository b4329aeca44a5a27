"""Alignment → (patterns, frequencies, leaf codes, ambiguity table): the host-side mirror
of what ``_LikelihoodFunction::SetupLFCaches`` hands to the pruning code.

Reference: ``src/core/likefunc.cpp:4253-4311`` builds, per partition,
``conditionalTerminalNodeStateFlag`` (``long[L][S]``: state ≥ 0, or −(k+1) selecting the
k-th ambiguity vector) and ``conditionalTerminalNodeLikelihoodCaches`` (``double[nAmbig][D]``)
from the ``_DataSetFilter`` (unique site patterns + ``theFrequencies``).  Character
resolution follows ``_DataSetFilter::Translate2Frequencies`` /
``_TranslationTable`` semantics: an ambiguous nucleotide resolves to the 0/1 indicator of
the compatible bases; for unit = 3 the codon indicator is the product over positions
restricted to sense codons; a fully unresolved character is all ones.

Synthetic data generation: random tree, sequences evolved from a uniform
random root with a per-branch single-nucleotide change probability, stop codons rejected.
"""
from __future__ import annotations

import dataclasses
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import models
from .tree import FlatTree, Node, flatten, random_tree

IUPAC = {
    "A": "A", "C": "C", "G": "G", "T": "T", "U": "T",
    "R": "AG", "Y": "CT", "S": "CG", "W": "AT", "K": "GT", "M": "AC",
    "B": "CGT", "D": "AGT", "H": "ACT", "V": "ACG",
    "N": "ACGT", "X": "ACGT", "?": "ACGT", "-": "ACGT", ".": "ACGT", "*": "ACGT",
}


@dataclasses.dataclass
class PatternData:
    """One partition's data in the layout the C-ABI consumes (pattern-indexed)."""

    D: int
    leaf_codes: np.ndarray      # int64[L, S]
    ambig: np.ndarray           # float64[n_ambig, D]
    pattern_freq: np.ndarray    # int64[S]
    site_to_pattern: np.ndarray  # int64[n_sites]

    @property
    def S(self) -> int:
        return int(self.leaf_codes.shape[1])

    @property
    def L(self) -> int:
        return int(self.leaf_codes.shape[0])


def _char_vector(chars: str, unit: int) -> Tuple[int, Optional[np.ndarray]]:
    """(state, None) for a resolved character, (−1, indicator) otherwise."""
    chars = chars.upper()
    if unit == 1:
        res = IUPAC.get(chars, "ACGT")
        if len(res) == 1:
            return models.NUC.index(res), None
        v = np.zeros(4)
        for c in res:
            v[models.NUC.index(c)] = 1.0
        return -1, v
    sets = [IUPAC.get(c, "ACGT") for c in chars]
    if all(len(s) == 1 for s in sets):
        cod = "".join(sets)
        if cod in models.CODON_INDEX:
            return models.CODON_INDEX[cod], None
        # a stop codon in the data: HyPhy treats an unmappable character as fully missing
        return -1, np.ones(61)
    v = np.zeros(61)
    for i, cod in enumerate(models.SENSE):
        if cod[0] in sets[0] and cod[1] in sets[1] and cod[2] in sets[2]:
            v[i] = 1.0
    if v.sum() == 0:
        v[:] = 1.0
    return -1, v


def compress(seqs: Sequence[str], unit: int) -> PatternData:
    """Site-pattern compression + leaf-code table for sequences given in ``flatLeaves``
    order.  Patterns are numbered in order of first appearance."""
    L = len(seqs)
    n_sites = len(seqs[0]) // unit
    D = 4 if unit == 1 else 61
    cols: Dict[Tuple[str, ...], int] = {}
    site_to_pattern = np.empty(n_sites, dtype=np.int64)
    pats: List[Tuple[str, ...]] = []
    for s in range(n_sites):
        key = tuple(seq[s * unit:(s + 1) * unit].upper() for seq in seqs)
        idx = cols.get(key)
        if idx is None:
            idx = len(pats)
            cols[key] = idx
            pats.append(key)
        site_to_pattern[s] = idx
    S = len(pats)
    freq = np.bincount(site_to_pattern, minlength=S).astype(np.int64)
    codes = np.empty((L, S), dtype=np.int64)
    amb_index: Dict[bytes, int] = {}
    amb_rows: List[np.ndarray] = []
    cache: Dict[str, Tuple[int, Optional[np.ndarray]]] = {}
    for p, key in enumerate(pats):
        for l, ch in enumerate(key):
            r = cache.get(ch)
            if r is None:
                r = _char_vector(ch, unit)
                cache[ch] = r
            st, vec = r
            if vec is None:
                codes[l, p] = st
            else:
                kb = vec.tobytes()
                k = amb_index.get(kb)
                if k is None:
                    k = len(amb_rows)
                    amb_index[kb] = k
                    amb_rows.append(vec)
                codes[l, p] = -(k + 1)
    ambig = np.array(amb_rows, dtype=np.float64).reshape(len(amb_rows), D)
    return PatternData(D, codes, ambig, freq, site_to_pattern)


def from_states(states: np.ndarray, D: int, compress_patterns: bool = True) -> PatternData:
    """Build PatternData directly from an integer state matrix [L, n_sites] (no
    ambiguities) — the fast path used for large synthetic workloads."""
    states = np.ascontiguousarray(states, dtype=np.int64)
    L, n = states.shape
    if compress_patterns:
        uniq, first, inv, counts = np.unique(states.T, axis=0, return_index=True, return_inverse=True, return_counts=True)
        # renumber patterns in order of first appearance (as the reference's filter does)
        order = np.argsort(first, kind="stable")
        rank = np.empty_like(order)
        rank[order] = np.arange(len(order))
        codes = np.ascontiguousarray(uniq[order].T)
        freq = counts[order].astype(np.int64)
        s2p = rank[inv.reshape(-1)].astype(np.int64)
    else:
        codes, freq, s2p = states.copy(), np.ones(n, dtype=np.int64), np.arange(n, dtype=np.int64)
    return PatternData(D, codes, np.zeros((0, D)), freq, s2p)


@dataclasses.dataclass
class Synthetic:
    tree: Node
    flat: FlatTree
    states: np.ndarray          # int64[L, n_sites] in flatLeaves order (codon or nucleotide states)
    seqs: List[str]             # nucleotide strings in flatLeaves order
    unit: int


def evolve(n_leaves: int, n_sites: int, unit: int, seed: int, p_change: float = 0.04,
           tree: Optional[Node] = None) -> Synthetic:
    """Deterministic synthetic alignment: uniform random root sequence,
    per-branch per-site single-nucleotide change with probability ``p_change``, stop
    codons rejected (the change is redrawn)."""
    rng = np.random.default_rng(seed)
    root = tree if tree is not None else random_tree(n_leaves, rng)
    flat = flatten(root)
    sense_mask = np.array([a != "*" for a in models._AA])
    sense64 = np.nonzero(sense_mask)[0]
    to61 = -np.ones(64, dtype=np.int64)
    to61[sense64] = np.arange(61)

    if unit == 3:
        root_seq = sense64[rng.integers(0, 61, size=n_sites)]
    else:
        root_seq = rng.integers(0, 4, size=n_sites)

    def mutate(seq: np.ndarray) -> np.ndarray:
        out = seq.copy()
        hit = np.nonzero(rng.random(n_sites) < p_change)[0]
        if unit == 1:
            out[hit] = (out[hit] + rng.integers(1, 4, size=len(hit))) % 4
            return out
        for s in hit:
            if unit == 1:
                out[s] = (out[s] + rng.integers(1, 4)) % 4
            else:
                while True:
                    pos = int(rng.integers(0, 3))
                    shift = 4 ** (2 - pos)
                    cur = (out[s] // shift) % 4
                    new = (cur + int(rng.integers(1, 4))) % 4
                    cand = out[s] + (new - cur) * shift
                    if sense_mask[cand]:
                        out[s] = cand
                        break
        return out

    leaf_seq: Dict[str, np.ndarray] = {}
    stack = [(root, root_seq)]
    while stack:
        node, seq = stack.pop()
        if node.is_leaf:
            leaf_seq[node.name] = seq
            continue
        for ch in node.children:
            stack.append((ch, mutate(seq)))
    raw = np.stack([leaf_seq[n] for n in flat.leaf_names])
    if unit == 3:
        states = to61[raw]
        lut = np.array([models.NUC[i >> 4] + models.NUC[(i >> 2) & 3] + models.NUC[i & 3] for i in range(64)])
        seqs = ["".join(lut[row]) for row in raw]
    else:
        states = raw.astype(np.int64)
        lut = np.array(list(models.NUC))
        seqs = ["".join(lut[row]) for row in raw]
    return Synthetic(root, flat, states, seqs, unit)


def inject_missing(seqs: Sequence[str], unit: int, frac: float, seed: int) -> List[str]:
    """Replace a fraction of characters with gaps / N / partial ambiguities (test input)."""
    rng = np.random.default_rng(seed)
    out = []
    for seq in seqs:
        arr = list(seq)
        n_units = len(arr) // unit
        for u in np.nonzero(rng.random(n_units) < frac)[0]:
            kind = rng.integers(0, 3)
            if kind == 0:
                for k in range(unit):
                    arr[u * unit + k] = "-"
            elif kind == 1:
                arr[u * unit + int(rng.integers(0, unit))] = "N"
            else:
                arr[u * unit + int(rng.integers(0, unit))] = "RYSWKM"[int(rng.integers(0, 6))]
        out.append("".join(arr))
    return out
