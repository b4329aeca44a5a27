"""TEST INFRASTRUCTURE — ctypes front-end of the C restatement ``oracle/hyphy_oracle.c``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module (as the checker).  The product path (``hyphy_amd``) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "hyphy_oracle.c")
LIB = os.path.join(HERE, "_build", "libhyphy_oracle.so")

_lib = None


def build(force: bool = False) -> str:
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        # -ffp-contract=off: keep the reference's non-FMA scalar semantics in the restatement
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", LIB, SRC, "-lm"])
    return LIB


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB)
        L = _lib
        dp, lp = C.POINTER(C.c_double), C.POINTER(C.c_long)
        L.hy_oracle_log_scaler.restype = C.c_double
        L.hy_oracle_expm.restype = C.c_int
        L.hy_oracle_expm.argtypes = [C.c_long, dp, C.c_int, dp, lp, lp]
        L.hy_oracle_expm_batch.restype = C.c_int
        L.hy_oracle_expm_batch.argtypes = [C.c_long, C.c_long, dp, C.c_int, dp]
        L.hy_oracle_tree_block.restype = C.c_double
        L.hy_oracle_tree_block.argtypes = [C.c_long] * 4 + [lp, lp, C.c_long, dp, lp, dp, lp, dp, dp, dp, lp,
                                                            C.c_long, C.c_long, dp, lp]
        L.hy_oracle_compute_block.restype = C.c_double
        L.hy_oracle_compute_block.argtypes = [C.c_long] * 4 + [lp, lp, C.c_long, dp, lp, dp, lp, dp, dp, dp, lp, C.c_long]
        L.hy_oracle_mix_categories.restype = C.c_double
        L.hy_oracle_mix_categories.argtypes = [C.c_long, C.c_long, dp, dp, lp, lp, dp, lp]
        L.hy_oracle_set_branch.restype = None
        L.hy_oracle_set_branch.argtypes = [C.c_long, lp]
    return _lib


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _l(a):
    return a.ctypes.data_as(C.POINTER(C.c_long)) if a is not None else None


def expm(Q: np.ndarray, sparse_hint: bool) -> np.ndarray:
    Q = np.ascontiguousarray(Q, dtype=np.float64)
    if Q.ndim == 2:
        P = np.empty_like(Q)
        st = lib().hy_oracle_expm(Q.shape[0], _d(Q), int(sparse_hint), _d(P), None, None)
        if st < 0:
            raise FloatingPointError("oracle expm failed")
        return P
    P = np.empty_like(Q)
    st = lib().hy_oracle_expm_batch(Q.shape[1], Q.shape[0], _d(Q), int(sparse_hint), _d(P))
    if st < 0:
        raise FloatingPointError("oracle expm failed")
    return P


def expm_counts(Q: np.ndarray, sparse_hint: bool):
    Q = np.ascontiguousarray(Q, dtype=np.float64)
    P = np.empty_like(Q)
    nt, ns = C.c_long(0), C.c_long(0)
    lib().hy_oracle_expm(Q.shape[0], _d(Q), int(sparse_hint), _d(P), C.byref(nt), C.byref(ns))
    return P, nt.value, ns.value


class OraclePartition:
    """Stateful mirror of one (filter, tree) partition of ``_LikelihoodFunction`` as far as
    ``ComputeBlock`` is concerned: owns iNodeCache / siteScalingFactors / overallScaler /
    siteCorrections exactly like ``SetupLFCaches`` (``likefunc.cpp:4163-4319``) allocates them."""

    def __init__(self, D, flat_parents, L, leaf_codes, ambig, pattern_freq, C_cat: int = 1):
        self.D = int(D)
        self.L = int(L)
        self.fp = np.ascontiguousarray(flat_parents, dtype=np.int64)
        self.I = len(self.fp) - self.L
        self.codes = np.ascontiguousarray(leaf_codes, dtype=np.int64)
        self.S = int(self.codes.shape[1])
        self.ambig = np.ascontiguousarray(ambig if ambig is not None and len(ambig) else np.zeros((1, D)), dtype=np.float64)
        self.freq = np.ascontiguousarray(pattern_freq, dtype=np.int64)
        self.C = int(C_cat)
        self.cache = np.zeros((self.C, self.I, self.S, self.D))
        self.scal = np.ones((self.C, self.I, self.S))
        self.overall = [np.zeros(1, dtype=np.int64) for _ in range(self.C)]
        self.site_corr = np.zeros((self.C, self.S), dtype=np.int64)
        self.P = np.zeros((self.C, self.L + self.I - 1, self.D, self.D))

    def set_P(self, node_codes, P, cat: int = 0):
        self.P[cat, np.asarray(node_codes)] = P

    def set_branch(self, node_code=None, states=None):
        """Pin node ``node_code`` (leaf l -> l, internal i -> L + i) to ``states[pattern]`` for the following
        evaluations (``ComputeBlock``'s branchIndex / branchValues); ``None`` removes the pin."""
        if node_code is None:
            self._pin = None
            lib().hy_oracle_set_branch(-1, None)
            return
        self._pin = np.ascontiguousarray(states, dtype=np.int64)
        code = int(node_code)
        ref_code = code - self.L if code >= self.L else self.I + code   # the reference's own encoding
        lib().hy_oracle_set_branch(ref_code, _l(self._pin))

    def compute_block(self, update_nodes, root_freqs, cat: int = 0, np_blocks: int = 1) -> float:
        un = np.ascontiguousarray(update_nodes, dtype=np.int64)
        rf = np.ascontiguousarray(root_freqs, dtype=np.float64)
        return lib().hy_oracle_compute_block(
            self.D, self.S, self.L, self.I, _l(self.fp), _l(un), len(un), _d(self.P[cat]), _l(self.codes),
            _d(self.ambig), _l(self.freq), _d(rf), _d(self.cache[cat]), _d(self.scal[cat]),
            _l(self.overall[cat]), np_blocks)

    def site_block(self, update_nodes, root_freqs, cat: int = 0):
        """Per-site mode (``siteRes != nil``): returns (site likelihoods, cumulative scaler counts)."""
        un = np.ascontiguousarray(update_nodes, dtype=np.int64)
        rf = np.ascontiguousarray(root_freqs, dtype=np.float64)
        out = np.zeros(self.S)
        lib().hy_oracle_tree_block(
            self.D, self.S, self.L, self.I, _l(self.fp), _l(un), len(un), _d(self.P[cat]), _l(self.codes),
            _d(self.ambig), _l(self.freq), _d(rf), _d(self.cache[cat]), _d(self.scal[cat]),
            _l(self.overall[cat]), 0, self.S, _d(out), _l(self.site_corr[cat]))
        return out, self.site_corr[cat].copy()

    def site_log_likelihoods(self, update_nodes, root_freqs, cat: int = 0) -> np.ndarray:
        lik, sc = self.site_block(update_nodes, root_freqs, cat)
        return np.log(lik) - sc * lib().hy_oracle_log_scaler()


def mix_categories(weights, site_lik, site_scalers, pattern_freq):
    w = np.ascontiguousarray(weights, dtype=np.float64)
    sl = np.ascontiguousarray(site_lik, dtype=np.float64)
    ss = np.ascontiguousarray(site_scalers, dtype=np.int64)
    pf = np.ascontiguousarray(pattern_freq, dtype=np.int64)
    Cc, S = sl.shape
    mixed = np.zeros(S)
    sc = np.zeros(S, dtype=np.int64)
    ll = lib().hy_oracle_mix_categories(S, Cc, _d(w), _d(sl), _l(ss), _l(pf), _d(mixed), _l(sc))
    return ll, mixed, sc
