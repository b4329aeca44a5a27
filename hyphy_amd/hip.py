"""ctypes binding of the C-ABI in ``include/hyphy_hip.h`` (``hyphy_amd/lib/libhyphy_hip.so``).

This is the only way Python reaches the device code: there is no PyTorch/numpy/CPU fallback —
if the shared library is missing or no MI355X is visible, construction raises.  The class
mirrors the reference-side lifetime: ``HipPartition(...)`` ≙ the hook in
``_LikelihoodFunction::SetupLFCaches`` (``likefunc.cpp:4313-4316``), ``evaluate`` ≙ the call
from ``ComputeBlock`` (``likefunc.cpp:10978-11123``), ``close`` ≙ ``DeleteCaches``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence, Tuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HYPHY_HIP_LIB") or os.path.join(HERE, "lib", "libhyphy_hip.so")  # (override: A/B builds)

EXPORTS = [
    "hyphy_hip_device_count", "hyphy_hip_create", "hyphy_hip_destroy", "hyphy_hip_evaluate",
    "hyphy_hip_evaluate_device", "hyphy_hip_fetch_device_scalar", "hyphy_hip_plan_reroot", "hyphy_hip_plan_pattern_order", "hyphy_hip_plan_schedule", "hyphy_hip_evaluate_async", "hyphy_hip_collect", "hyphy_hip_evaluate_mixture", "hyphy_hip_evaluate_mixture_built", "hyphy_hip_comm_unique_id", "hyphy_hip_comm_init_rank", "hyphy_hip_comm_init_all", "hyphy_hip_allreduce_device", "hyphy_hip_evaluate_allreduce", "hyphy_hip_evaluate_built_allreduce", "hyphy_hip_last_allreduce_ms", "hyphy_hip_evaluate_categories", "hyphy_hip_download_partials",
    "hyphy_hip_expm_batch", "hyphy_hip_set_q_templates", "hyphy_hip_build_q", "hyphy_hip_q_buffer",
    "hyphy_hip_evaluate_built", "hyphy_hip_evaluate_built_sites", "hyphy_hip_update_q_templates", "hyphy_hip_evaluate_categories_built", "hyphy_hip_evaluate_categories_built_sites", "hyphy_hip_prune_timings", "hyphy_hip_prune_launches",
    "hyphy_hip_prune_kernel_name", "hyphy_hip_branch_cache_build", "hyphy_hip_branch_cache_evaluate",
    "hyphy_hip_set_pinned_states", "hyphy_hip_site_fits_evaluate", "hyphy_hip_site_fits_evaluate_mixture", "hyphy_hip_site_fits_kernel_ms",
    "hyphy_hip_synchronize", "hyphy_hip_stream", "hyphy_hip_set_stream", "hyphy_hip_last_timings", "hyphy_hip_set_timing_detail", "hyphy_hip_schedule_info", "hyphy_hip_set_repeats", "hyphy_hip_repeat_stats", "hyphy_hip_plan_repeats", "hyphy_hip_plan_trunk_walk", "hyphy_hip_plan_nucgen", "hyphy_hip_comm_init_host", "hyphy_hip_evaluate_exchange", "hyphy_hip_evaluate_built_exchange",
    "hyphy_hip_xch_open", "hyphy_hip_xch_sum", "hyphy_hip_xch_close", "hyphy_hip_last_error",
    "hyphy_hip_version",
]

_lib = None


class HipError(RuntimeError):
    pass


class HipUnsupported(HipError):
    """The library answered "> 0": use the CPU path of the host application."""


def load():
    """Load libhyphy_hip.so; raise loudly if it has not been built (``__graft_entry__.build()``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipError(f"{LIB_PATH} not built — run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    dp, lp, vp = C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_void_p
    lib.hyphy_hip_device_count.restype = C.c_int
    lib.hyphy_hip_create.restype = C.c_int
    lib.hyphy_hip_create.argtypes = [C.POINTER(vp)] + [C.c_int64] * 5 + [lp, lp, dp, C.c_int64, lp, C.c_int, C.c_int]
    lib.hyphy_hip_destroy.restype = None
    lib.hyphy_hip_destroy.argtypes = [vp]
    lib.hyphy_hip_evaluate.restype = C.c_int
    lib.hyphy_hip_evaluate.argtypes = [vp, C.c_int64, lp, C.c_int64, lp, C.c_int64, dp, C.c_int, dp, dp, dp, lp]
    lib.hyphy_hip_comm_unique_id.restype = C.c_int
    lib.hyphy_hip_comm_unique_id.argtypes = [vp]
    lib.hyphy_hip_comm_init_rank.restype = C.c_int
    lib.hyphy_hip_comm_init_rank.argtypes = [vp, vp, C.c_int, C.c_int]
    lib.hyphy_hip_comm_init_all.restype = C.c_int
    lib.hyphy_hip_comm_init_all.argtypes = [vp]
    lib.hyphy_hip_allreduce_device.restype = C.c_int
    lib.hyphy_hip_allreduce_device.argtypes = [vp, vp]
    lib.hyphy_hip_evaluate_allreduce.restype = C.c_int
    lib.hyphy_hip_evaluate_allreduce.argtypes = [vp, C.c_int64, lp, C.c_int64, lp, C.c_int64, dp, C.c_int, dp, dp]
    lib.hyphy_hip_evaluate_mixture.restype = C.c_int
    lib.hyphy_hip_evaluate_mixture.argtypes = [vp, C.c_int64, lp, C.c_int64, lp, C.c_int64, lp, dp, dp, dp, dp, dp, lp]
    lib.hyphy_hip_evaluate_mixture_built.restype = C.c_int
    lib.hyphy_hip_evaluate_mixture_built.argtypes = [vp, C.c_int64, lp, C.c_int64, lp, C.c_int64, lp, dp, dp, dp, dp, lp]
    lib.hyphy_hip_evaluate_async.restype = C.c_int
    lib.hyphy_hip_evaluate_async.argtypes = [vp, C.c_int64, lp, C.c_int64, lp, C.c_int64, dp, C.c_int, dp]
    lib.hyphy_hip_collect.restype = C.c_int
    lib.hyphy_hip_collect.argtypes = [vp, dp, dp, lp]
    lib.hyphy_hip_plan_reroot.restype = C.c_int64
    lib.hyphy_hip_plan_reroot.argtypes = [C.c_int64, C.c_int64, lp, C.c_int64, lp, C.c_int64]
    lib.hyphy_hip_plan_pattern_order.restype = C.c_int
    lib.hyphy_hip_plan_pattern_order.argtypes = [C.c_int64, C.c_int64, C.c_int64, lp, lp]
    lib.hyphy_hip_plan_schedule.restype = C.c_int
    lib.hyphy_hip_plan_schedule.argtypes = [C.c_int64, C.c_int64, lp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, lp]
    lib.hyphy_hip_fetch_device_scalar.restype = C.c_int
    lib.hyphy_hip_fetch_device_scalar.argtypes = [vp, vp, C.POINTER(C.c_double)]
    lib.hyphy_hip_evaluate_device.restype = C.c_int
    lib.hyphy_hip_evaluate_device.argtypes = [vp, C.c_int64, lp, C.c_int64, lp, C.c_int64, vp, C.c_int, dp, vp]
    lib.hyphy_hip_evaluate_categories.restype = C.c_int
    lib.hyphy_hip_evaluate_categories.argtypes = [vp, lp, C.c_int64, lp, C.c_int64, dp, C.c_int, dp, dp, dp, dp, lp]
    lib.hyphy_hip_download_partials.restype = C.c_int
    lib.hyphy_hip_download_partials.argtypes = [vp, C.c_int64, dp, lp]
    lib.hyphy_hip_expm_batch.restype = C.c_int
    lib.hyphy_hip_expm_batch.argtypes = [C.c_int64, C.c_int64, dp, dp]
    lib.hyphy_hip_set_q_templates.restype = C.c_int
    lib.hyphy_hip_set_q_templates.argtypes = [vp, C.c_int64, dp]
    lib.hyphy_hip_build_q.restype = C.c_int
    lib.hyphy_hip_build_q.argtypes = [vp, C.c_int64, dp]
    lib.hyphy_hip_evaluate_built.restype = C.c_int
    lib.hyphy_hip_evaluate_built.argtypes = [vp, C.c_int64, lp, C.c_int64, lp, C.c_int64, dp, dp]
    lib.hyphy_hip_evaluate_built_sites.restype = C.c_int
    lib.hyphy_hip_evaluate_built_sites.argtypes = [vp, C.c_int64, lp, C.c_int64, lp, C.c_int64, dp, dp, dp, lp]
    lib.hyphy_hip_evaluate_categories_built.restype = C.c_int
    lib.hyphy_hip_evaluate_categories_built.argtypes = [vp, lp, C.c_int64, lp, C.c_int64, dp, dp, dp]
    lib.hyphy_hip_evaluate_categories_built_sites.restype = C.c_int
    lib.hyphy_hip_evaluate_categories_built_sites.argtypes = [vp, lp, C.c_int64, lp, C.c_int64, dp, dp, dp, dp, lp]
    lib.hyphy_hip_site_fits_evaluate.restype = C.c_int
    lib.hyphy_hip_site_fits_evaluate.argtypes = [vp, C.c_int64, C.c_int64, lp, dp, dp, dp, dp]
    lib.hyphy_hip_site_fits_evaluate_mixture.restype = C.c_int
    lib.hyphy_hip_site_fits_evaluate_mixture.argtypes = [vp, C.c_int64, C.c_int64, C.c_int64, lp, dp, dp, dp, dp, dp]
    lib.hyphy_hip_site_fits_kernel_ms.restype = C.c_double
    lib.hyphy_hip_site_fits_kernel_ms.argtypes = [vp]
    lib.hyphy_hip_q_buffer.restype = vp
    lib.hyphy_hip_q_buffer.argtypes = [vp]
    lib.hyphy_hip_synchronize.restype = C.c_int
    lib.hyphy_hip_synchronize.argtypes = [vp]
    lib.hyphy_hip_stream.restype = vp
    lib.hyphy_hip_stream.argtypes = [vp]
    lib.hyphy_hip_set_stream.restype = C.c_int
    lib.hyphy_hip_set_stream.argtypes = [vp, vp]
    lib.hyphy_hip_evaluate_built_allreduce.restype = C.c_int
    lib.hyphy_hip_evaluate_built_allreduce.argtypes = [vp, C.c_int64, lp, C.c_int64, lp, C.c_int64, dp, dp]
    lib.hyphy_hip_last_allreduce_ms.restype = C.c_double
    lib.hyphy_hip_last_allreduce_ms.argtypes = [vp]
    lib.hyphy_hip_last_timings.restype = C.c_int
    lib.hyphy_hip_last_timings.argtypes = [vp, dp]
    lib.hyphy_hip_schedule_info.restype = C.c_char_p
    lib.hyphy_hip_schedule_info.argtypes = [vp]
    lib.hyphy_hip_set_timing_detail.restype = C.c_int
    lib.hyphy_hip_set_timing_detail.argtypes = [vp, C.c_int]
    lib.hyphy_hip_prune_timings.restype = C.c_int64
    lib.hyphy_hip_prune_timings.argtypes = [vp, dp, C.c_int64]
    lib.hyphy_hip_prune_launches.restype = C.c_int
    lib.hyphy_hip_prune_launches.argtypes = [vp]
    lib.hyphy_hip_set_pinned_states.restype = C.c_int
    lib.hyphy_hip_set_pinned_states.argtypes = [vp, C.c_int64, lp]
    lib.hyphy_hip_branch_cache_build.restype = C.c_int
    lib.hyphy_hip_branch_cache_build.argtypes = [vp, C.c_int64, C.c_int64]
    lib.hyphy_hip_branch_cache_evaluate.restype = C.c_int
    lib.hyphy_hip_branch_cache_evaluate.argtypes = [vp, C.c_int64, C.c_int64, dp, C.c_int, dp, dp, lp]
    lib.hyphy_hip_prune_kernel_name.restype = C.c_char_p
    lib.hyphy_hip_prune_kernel_name.argtypes = [vp]
    lib.hyphy_hip_set_repeats.restype = C.c_int
    lib.hyphy_hip_set_repeats.argtypes = [vp, C.c_int]
    lib.hyphy_hip_plan_trunk_walk.restype = C.c_int64
    lib.hyphy_hip_plan_trunk_walk.argtypes = [C.c_int64, C.c_int64, lp, lp, C.c_int64]
    lib.hyphy_hip_plan_repeats.restype = C.c_int64
    lib.hyphy_hip_plan_repeats.argtypes = [C.c_int64, C.c_int64, lp, C.c_int64, lp, C.c_double, lp, lp]
    lib.hyphy_hip_repeat_stats.restype = C.c_int
    lib.hyphy_hip_repeat_stats.argtypes = [vp, lp]
    lib.hyphy_hip_comm_init_host.restype = C.c_int
    lib.hyphy_hip_comm_init_host.argtypes = [vp, C.c_char_p, C.c_int, C.c_int]
    lib.hyphy_hip_evaluate_exchange.restype = C.c_int
    lib.hyphy_hip_evaluate_exchange.argtypes = [vp, C.c_int64, lp, C.c_int64, lp, C.c_int64, dp, C.c_int, dp, dp]
    lib.hyphy_hip_evaluate_built_exchange.restype = C.c_int
    lib.hyphy_hip_evaluate_built_exchange.argtypes = [vp, C.c_int64, lp, C.c_int64, lp, C.c_int64, dp, dp]
    lib.hyphy_hip_xch_open.restype = C.c_int
    lib.hyphy_hip_xch_open.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    lib.hyphy_hip_xch_sum.restype = C.c_int
    lib.hyphy_hip_xch_sum.argtypes = [C.c_void_p, C.c_double, C.c_int, dp]
    lib.hyphy_hip_xch_close.restype = None
    lib.hyphy_hip_xch_close.argtypes = [C.c_void_p]
    lib.hyphy_hip_plan_nucgen.restype = C.c_int64
    lib.hyphy_hip_plan_nucgen.argtypes = [C.c_int64, C.c_int64, lp, lp, C.c_int64, C.c_char_p, C.c_int64, lp]
    lib.hyphy_hip_last_error.restype = C.c_char_p
    lib.hyphy_hip_version.restype = C.c_char_p
    _lib = lib
    return lib


def _check(rc: int):
    if rc == 0:
        return
    msg = load().hyphy_hip_last_error().decode()
    if rc > 0:
        raise HipUnsupported(msg)
    raise HipError(msg)


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _l(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64)) if a is not None else None


def plan_reroot(flat_parents, L: int, candidate: int = 0) -> np.ndarray:
    """Host-only: internal indices from the given root to the ``candidate``-th height-minimising node (empty: none)."""
    fp = np.ascontiguousarray(flat_parents, dtype=np.int64)
    out = np.zeros(64, dtype=np.int64)
    n = int(load().hyphy_hip_plan_reroot(int(L), len(fp) - int(L), _l(fp), int(candidate), _l(out), len(out)))
    if n < 0:
        raise HipError("plan_reroot: bad arguments")
    return out[:n].copy()


def plan_pattern_order(D: int, leaf_codes) -> np.ndarray:
    """Host-only: the order in which the library stores the patterns on the device (caller's pattern indices)."""
    lc = np.ascontiguousarray(leaf_codes, dtype=np.int64)
    out = np.zeros(lc.shape[1], dtype=np.int64)
    if load().hyphy_hip_plan_pattern_order(int(D), lc.shape[0], lc.shape[1], _l(lc), _l(out)):
        raise HipError("plan_pattern_order: bad arguments")
    return out


def plan_schedule(flat_parents, L: int, kernel: int = 1, chain_m: int = 0, ntiles: int = 64, reroot: bool = False) -> dict:
    """Host-only: compile the steady-state full-pass schedule of a 61-state partition of this shape and decode its join
    table the way the kernels do (include/hyphy_hip.h: hyphy_hip_plan_schedule)."""
    fp = np.ascontiguousarray(flat_parents, dtype=np.int64)
    info = np.zeros(8, dtype=np.int64)
    if load().hyphy_hip_plan_schedule(int(L), len(fp) - int(L), _l(fp), int(kernel), int(chain_m), int(ntiles), int(bool(reroot)), _l(info)):
        raise HipError("plan_schedule: bad arguments")
    keys = ("chain", "programs", "entries", "max_slot", "max_need", "decode_errors", "rerooted", "trunk_nodes")
    return dict(zip(keys, (int(v) for v in info)))


def plan_repeats(flat_parents, L: int, leaf_codes, theta: float = 0.0):
    """Host-only: (classes per internal node, compressed flags, edge products of a full pass with one table per compressed node)."""
    lib = load()
    fp = np.ascontiguousarray(flat_parents, dtype=np.int64)
    lc = np.ascontiguousarray(leaf_codes, dtype=np.int64)
    I = len(fp) - L
    cl = np.zeros(I, dtype=np.int64)
    cp = np.zeros(I, dtype=np.int64)
    work = lib.hyphy_hip_plan_repeats(L, I, _l(fp), lc.shape[1], _l(lc), float(theta), _l(cl), _l(cp))
    if work < 0:
        raise HipError("plan_repeats: bad arguments")
    return cl, cp.astype(bool), int(work)


def plan_trunk_walk(parents, L: int):
    """Host-only: the program trunk_walk_kernel runs for a trunk over ``L`` generalised leaves (``parents[c]`` = node code of c's
    parent; internal node i has code L + i, children before parents, the root last).  Returns a dict: ``nodes`` [(internal index or
    -1, leaf children, first input, flags)], ``inputs`` (leaf codes), ``depth``, ``one`` = (first, end) node range of the one-chain
    form, ``two`` = ((first, end), (first, end)) of the two chains or None."""
    lib = load()
    par = np.ascontiguousarray(parents, dtype=np.int64)
    I = len(par) + 1 - L
    cap = 16 + 8 * (L + I) + 4 * L
    out = np.zeros(cap, dtype=np.int64)
    n = lib.hyphy_hip_plan_trunk_walk(L, I, _l(par), _l(out), cap)
    if n < 0:
        raise HipError("plan_trunk_walk: bad arguments")
    nn, ni, depth, two = (int(x) for x in out[:4])
    nodes = [tuple(int(x) for x in out[10 + 4 * k: 14 + 4 * k]) for k in range(nn)]
    inputs = [int(x) for x in out[10 + 4 * nn: 10 + 4 * nn + ni]]
    return {"nodes": nodes, "inputs": inputs, "depth": depth, "one": (int(out[4]), int(out[5])),
            "two": ((int(out[6]), int(out[7])), (int(out[8]), int(out[9]))) if two else None}


def plan_nucgen(flat_parents, L: int, leaf_has_ambig=None, compile_it: bool = True, small: bool = False):
    """Host-only (libhiprtc, no device): (source of the run-time generated 4-state kernel for this tree's steady-state full pass,
    compiled for gfx950?).  Empty source: the generator does not cover the schedule."""
    lib = load()
    fp = np.ascontiguousarray(flat_parents, dtype=np.int64)
    I = len(fp) - L
    amb = None if leaf_has_ambig is None else np.ascontiguousarray(leaf_has_ambig, dtype=np.int64)
    cap = 1 << 22
    buf = C.create_string_buffer(cap)
    ok = np.zeros(1, dtype=np.int64)
    n = lib.hyphy_hip_plan_nucgen(L, I, _l(fp), _l(amb), 1 if small else 0, buf, cap, _l(ok) if compile_it else None)
    if n < 0:
        raise HipError("plan_nucgen: bad arguments")
    return buf.value.decode(), bool(ok[0])


class HostExchange:
    """The collective-free combine of one-process-per-GPU runs on one node (hyphy_hip_xch_*): every rank posts its partial
    log-likelihood into a shared-memory segment and reads the others'; ``sum`` returns the total (same bits on every rank)."""

    def __init__(self, name: str, rank: int, n_ranks: int):
        self._lib = load()
        h = C.c_void_p()
        _check(self._lib.hyphy_hip_xch_open(name.encode(), int(rank), int(n_ranks), C.byref(h)))
        self._h = h

    def sum(self, local: float, failed: bool = False) -> float:
        out = C.c_double(0.0)
        _check(self._lib.hyphy_hip_xch_sum(self._h, float(local), 1 if failed else 0, C.byref(out)))
        return out.value

    def close(self):
        if self._h:
            self._lib.hyphy_hip_xch_close(self._h)
            self._h = None


def device_count() -> int:
    return int(load().hyphy_hip_device_count())


def expm_batch(Q: np.ndarray) -> np.ndarray:
    """Batched P = exp(Q) on the device (drop-in for the loop at ``tree.cpp:3011-3037``)."""
    Q = np.ascontiguousarray(Q, dtype=np.float64)
    single = Q.ndim == 2
    Qb = Q[None] if single else Q
    P = np.empty_like(Qb)
    _check(load().hyphy_hip_expm_batch(Qb.shape[1], Qb.shape[0], _d(Qb), _d(P)))
    return P[0] if single else P


class HipPartition:
    """Device-resident state of one (filter, tree) partition."""

    def __init__(self, D: int, flat_parents, L: int, leaf_codes, ambig, pattern_freq, C_cat: int = 1,
                 device_first: int = 0, device_count: int = 1):
        lib = load()
        self.D, self.L = int(D), int(L)
        fp = np.ascontiguousarray(flat_parents, dtype=np.int64)
        self.I = len(fp) - self.L
        codes = np.ascontiguousarray(leaf_codes, dtype=np.int64)
        self.S = int(codes.shape[1])
        self.C = int(C_cat)
        self.B = self.L + self.I - 1
        amb = np.ascontiguousarray(ambig, dtype=np.float64) if ambig is not None and len(ambig) else None
        freq = np.ascontiguousarray(pattern_freq, dtype=np.int64)
        h = C.c_void_p()
        _check(lib.hyphy_hip_create(C.byref(h), self.D, self.S, self.L, self.I, self.C, _l(fp), _l(codes),
                                    _d(amb), 0 if amb is None else amb.shape[0], _l(freq), device_first, device_count))
        self._h = h
        self._lib = lib

    def close(self):
        if getattr(self, "_h", None):
            self._lib.hyphy_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- evaluation -------------------------------------------------------------------------
    def evaluate(self, update_nodes, q_nodes, q_dense, root_freqs, cat: int = -1, q_is_probability: bool = False,
                 per_site: bool = False):
        un = np.ascontiguousarray(update_nodes, dtype=np.int64)
        qn = np.ascontiguousarray(q_nodes, dtype=np.int64)
        q = np.ascontiguousarray(q_dense, dtype=np.float64) if len(qn) else None
        rf = np.ascontiguousarray(root_freqs, dtype=np.float64)
        out = C.c_double(0.0)
        sl = np.zeros(self.S) if per_site else None
        sc = np.zeros(self.S, dtype=np.int64) if per_site else None
        _check(self._lib.hyphy_hip_evaluate(self._h, cat, _l(un), len(un), _l(qn), len(qn), _d(q),
                                            int(q_is_probability), _d(rf), C.byref(out), _d(sl), _l(sc)))
        return (out.value, sl, sc) if per_site else out.value

    # -- RCCL (one process per GPU: the C-ABI's own all-reduce of the partition log-likelihood) ---------------
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        _check(load().hyphy_hip_comm_unique_id(C.cast(buf, C.c_void_p)))
        return buf.raw

    def comm_init_rank(self, unique_id: bytes, rank: int, n_ranks: int):
        buf = C.create_string_buffer(unique_id, 128)
        _check(self._lib.hyphy_hip_comm_init_rank(self._h, C.cast(buf, C.c_void_p), rank, n_ranks))

    def comm_init_all(self):
        _check(self._lib.hyphy_hip_comm_init_all(self._h))

    def evaluate_allreduce(self, update_nodes, q_nodes, q_dense, root_freqs, cat: int = -1, q_is_probability: bool = False):
        """``evaluate`` on this rank's pattern shard + one RCCL all-reduce of the partial log-L: the whole alignment's value."""
        un = np.ascontiguousarray(update_nodes, dtype=np.int64)
        qn = np.ascontiguousarray(q_nodes, dtype=np.int64)
        q = np.ascontiguousarray(q_dense, dtype=np.float64) if len(qn) else None
        rf = np.ascontiguousarray(root_freqs, dtype=np.float64)
        out = C.c_double(0.0)
        _check(self._lib.hyphy_hip_evaluate_allreduce(self._h, cat, _l(un), len(un), _l(qn), len(qn), _d(q), int(q_is_probability),
                                                      _d(rf), C.byref(out)))
        return out.value

    def evaluate_mixture(self, update_nodes, q_nodes, q_components, weights, root_freqs, cat: int = -1, per_site: bool = False):
        """Branch-site mixture on every listed branch: ``q_components`` [n_q, M, D, D] rate matrices, ``weights`` [n_q, M];
        the device forms ``P_k = sum_m w_km exp(Q_km)`` (explicit-form models: BUSTED / BS-REL)."""
        un = np.ascontiguousarray(update_nodes, dtype=np.int64)
        qn = np.ascontiguousarray(q_nodes, dtype=np.int64)
        q = np.ascontiguousarray(q_components, dtype=np.float64)
        w = np.ascontiguousarray(weights, dtype=np.float64)
        assert q.ndim == 4 and w.shape == q.shape[:2] and q.shape[0] == len(qn)
        cnt = np.full(len(qn), q.shape[1], dtype=np.int64)
        rf = np.ascontiguousarray(root_freqs, dtype=np.float64)
        out = C.c_double(0.0)
        sl = np.zeros(self.S) if per_site else None
        sc = np.zeros(self.S, dtype=np.int64) if per_site else None
        _check(self._lib.hyphy_hip_evaluate_mixture(self._h, cat, _l(un), len(un), _l(qn), len(qn), _l(cnt), _d(q), _d(w),
                                                    _d(rf), C.byref(out), _d(sl), _l(sc)))
        return (out.value, sl, sc) if per_site else out.value

    def evaluate_mixture_built(self, update_nodes, q_nodes, coeffs, weights, root_freqs, cat: int = -1, per_site: bool = False):
        """The same with the component rate matrices formed on the device from the uploaded templates: ``coeffs`` [n_q, M, K]
        (one coefficient row per branch and component: ``hyphy_hip_build_q`` + ``hyphy_hip_evaluate_mixture_built``)."""
        un = np.ascontiguousarray(update_nodes, dtype=np.int64)
        qn = np.ascontiguousarray(q_nodes, dtype=np.int64)
        c = np.ascontiguousarray(coeffs, dtype=np.float64)
        w = np.ascontiguousarray(weights, dtype=np.float64)
        assert c.ndim == 3 and w.shape == c.shape[:2] and c.shape[0] == len(qn)
        cnt = np.full(len(qn), c.shape[1], dtype=np.int64)
        rf = np.ascontiguousarray(root_freqs, dtype=np.float64)
        out = C.c_double(0.0)
        sl = np.zeros(self.S) if per_site else None
        sc = np.zeros(self.S, dtype=np.int64) if per_site else None
        _check(self._lib.hyphy_hip_build_q(self._h, c.shape[0] * c.shape[1], _d(c)))
        _check(self._lib.hyphy_hip_evaluate_mixture_built(self._h, cat, _l(un), len(un), _l(qn), len(qn), _l(cnt), _d(w), _d(rf),
                                                          C.byref(out), _d(sl), _l(sc)))
        return (out.value, sl, sc) if per_site else out.value

    def prepare_mixture_built_step(self, update_nodes, q_nodes, weights, root_freqs, coeffs: np.ndarray, cat: int = -1):
        """Zero-argument callable: ``build_q`` of one coefficient row per (branch, component) + ``evaluate_mixture_built`` -> log-L.
        ``coeffs`` [n_q, M, K] and ``weights`` [n_q, M] are read at call time."""
        un = np.ascontiguousarray(update_nodes, dtype=np.int64)
        qn = np.ascontiguousarray(q_nodes, dtype=np.int64)
        rf = np.ascontiguousarray(root_freqs, dtype=np.float64)
        assert coeffs.flags.c_contiguous and coeffs.dtype == np.float64 and coeffs.ndim == 3 and coeffs.shape[0] == len(qn)
        assert weights.flags.c_contiguous and weights.dtype == np.float64 and weights.shape == coeffs.shape[:2]
        cnt = np.full(len(qn), coeffs.shape[1], dtype=np.int64)
        keep = (un, qn, rf, coeffs, weights, cnt)
        lib, h = self._lib, self._h
        pun, pqn, prf, pco, pw, pcnt = _l(un), _l(qn), _d(rf), _d(coeffs), _d(weights), _l(cnt)
        nun, nqn, nrows = len(un), len(qn), coeffs.shape[0] * coeffs.shape[1]
        out = C.c_double(0.0)
        pout = C.byref(out)

        def step(_keep=keep):
            rc = lib.hyphy_hip_build_q(h, nrows, pco)
            if rc == 0:
                rc = lib.hyphy_hip_evaluate_mixture_built(h, cat, pun, nun, pqn, nqn, pcnt, pw, prf, pout, None, None)
            if rc:
                _check(rc)
            return out.value
        return step

    def evaluate_async(self, update_nodes, q_nodes, q_dense, root_freqs, cat: int = -1, q_is_probability: bool = False):
        """Enqueue an evaluation and return at once (``collect`` waits for it): lets the partitions of one
        likelihood function overlap on different devices / streams."""
        un = np.ascontiguousarray(update_nodes, dtype=np.int64)
        qn = np.ascontiguousarray(q_nodes, dtype=np.int64)
        q = np.ascontiguousarray(q_dense, dtype=np.float64) if len(qn) else None
        rf = np.ascontiguousarray(root_freqs, dtype=np.float64)
        _check(self._lib.hyphy_hip_evaluate_async(self._h, cat, _l(un), len(un), _l(qn), len(qn), _d(q),
                                                  int(q_is_probability), _d(rf)))

    def collect(self, per_site: bool = False):
        out = C.c_double(0.0)
        sl = np.zeros(self.S) if per_site else None
        sc = np.zeros(self.S, dtype=np.int64) if per_site else None
        _check(self._lib.hyphy_hip_collect(self._h, C.byref(out), _d(sl), _l(sc)))
        return (out.value, sl, sc) if per_site else out.value

    def evaluate_device(self, update_nodes, q_nodes, d_q_ptr: int, root_freqs, d_logl_ptr: int, cat: int = -1,
                        q_is_probability: bool = False):
        un = np.ascontiguousarray(update_nodes, dtype=np.int64)
        qn = np.ascontiguousarray(q_nodes, dtype=np.int64)
        rf = np.ascontiguousarray(root_freqs, dtype=np.float64)
        _check(self._lib.hyphy_hip_evaluate_device(self._h, cat, _l(un), len(un), _l(qn), len(qn),
                                                   C.c_void_p(d_q_ptr), int(q_is_probability), _d(rf),
                                                   C.c_void_p(d_logl_ptr)))

    def prepare_device_step(self, update_nodes, q_nodes, root_freqs, d_logl_ptr: int, coeffs: np.ndarray,
                            cat: int = -1):
        """Returns a zero-argument callable that enqueues ``build_q(coeffs)`` + ``evaluate_device`` with
        all ctypes arguments marshalled ONCE (numpy's ``.ctypes.data_as`` costs tens of microseconds per
        call — more than the C-ABI calls themselves).  ``coeffs`` is read at call time (update it in
        place between calls)."""
        un = np.ascontiguousarray(update_nodes, dtype=np.int64)
        qn = np.ascontiguousarray(q_nodes, dtype=np.int64)
        rf = np.ascontiguousarray(root_freqs, dtype=np.float64)
        assert coeffs.flags.c_contiguous and coeffs.dtype == np.float64
        keep = (un, qn, rf, coeffs)
        lib, h = self._lib, self._h
        pun, pqn, prf, pco = _l(un), _l(qn), _d(rf), _d(coeffs)
        nun, nqn, nco = len(un), len(qn), coeffs.shape[0]
        dq, dl = C.c_void_p(self.q_buffer()), C.c_void_p(d_logl_ptr)

        def step(_keep=keep):
            rc = lib.hyphy_hip_build_q(h, nco, pco)
            if rc == 0:
                rc = lib.hyphy_hip_evaluate_device(h, cat, pun, nun, pqn, nqn, dq, 0, prf, dl)
            if rc:
                _check(rc)
        return step

    def prepare_fetch(self, d_ptr: int):
        """Zero-argument callable: the double at device address ``d_ptr`` (written by work queued on this partition's
        stream, e.g. an all-reduce behind ``evaluate_device``) through the host-mapped result record -> float."""
        lib, h, src, out = self._lib, self._h, C.c_void_p(d_ptr), C.c_double(0.0)
        ref = C.byref(out)

        def fetch():
            rc = lib.hyphy_hip_fetch_device_scalar(h, src, ref)
            if rc:
                _check(rc)
            return out.value
        return fetch

    def prepare_built_step(self, update_nodes, q_nodes, root_freqs, coeffs: np.ndarray, cat: int = -1):
        """Zero-argument callable: ``build_q(coeffs)`` + synchronous ``evaluate_built`` -> log-L (float).
        ctypes arguments are marshalled once; ``coeffs`` is read at call time."""
        un = np.ascontiguousarray(update_nodes, dtype=np.int64)
        qn = np.ascontiguousarray(q_nodes, dtype=np.int64)
        rf = np.ascontiguousarray(root_freqs, dtype=np.float64)
        assert coeffs.flags.c_contiguous and coeffs.dtype == np.float64
        keep = (un, qn, rf, coeffs)
        lib, h = self._lib, self._h
        pun, pqn, prf, pco = _l(un), _l(qn), _d(rf), _d(coeffs)
        nun, nqn, nco = len(un), len(qn), coeffs.shape[0]
        out = C.c_double(0.0)
        pout = C.byref(out)

        def step(_keep=keep):
            rc = lib.hyphy_hip_build_q(h, nco, pco)
            if rc == 0:
                rc = lib.hyphy_hip_evaluate_built(h, cat, pun, nun, pqn, nqn, prf, pout)
            if rc:
                _check(rc)
            return out.value
        return step

    def prepare_built_allreduce_step(self, update_nodes, q_nodes, root_freqs, coeffs: np.ndarray, cat: int = -1):
        """Zero-argument callable: ``build_q(coeffs)`` + ``evaluate_built_allreduce`` -> log-L of the WHOLE alignment on every
        rank (local evaluation of this rank's pattern shard, one ncclAllReduce in the partition's stream, value back through
        the host-mapped record).  Needs ``comm_init_rank`` first."""
        un = np.ascontiguousarray(update_nodes, dtype=np.int64)
        qn = np.ascontiguousarray(q_nodes, dtype=np.int64)
        rf = np.ascontiguousarray(root_freqs, dtype=np.float64)
        assert coeffs.flags.c_contiguous and coeffs.dtype == np.float64
        keep = (un, qn, rf, coeffs)
        lib, h = self._lib, self._h
        pun, pqn, prf, pco = _l(un), _l(qn), _d(rf), _d(coeffs)
        nun, nqn, nco = len(un), len(qn), coeffs.shape[0]
        out = C.c_double(0.0)
        pout = C.byref(out)

        def step(_keep=keep):
            rc = lib.hyphy_hip_build_q(h, nco, pco)
            if rc == 0:
                rc = lib.hyphy_hip_evaluate_built_allreduce(h, cat, pun, nun, pqn, nqn, prf, pout)
            if rc:
                _check(rc)
            return out.value
        return step

    def comm_init_host(self, name: str, rank: int, n_ranks: int):
        """Attach this rank's partition to the run's shared-memory exchange (collective-free combine, one node)."""
        _check(self._lib.hyphy_hip_comm_init_host(self._h, name.encode(), int(rank), int(n_ranks)))

    def prepare_built_exchange_step(self, update_nodes, q_nodes, root_freqs, coeffs: np.ndarray, cat: int = -1):
        """``build_q`` + ``evaluate_built_exchange``: the local evaluation, then ONE host-side exchange of the partial log-likelihoods
        through shared memory (no collective on the device); the total on every rank.  Needs ``comm_init_host`` first."""
        un = np.ascontiguousarray(update_nodes, dtype=np.int64)
        qn = np.ascontiguousarray(q_nodes, dtype=np.int64)
        rf = np.ascontiguousarray(root_freqs, dtype=np.float64)
        assert coeffs.flags.c_contiguous and coeffs.dtype == np.float64
        keep = (un, qn, rf, coeffs)
        lib, h = self._lib, self._h
        pun, pqn, prf, pco = _l(un), _l(qn), _d(rf), _d(coeffs)
        nun, nqn, nco = len(un), len(qn), coeffs.shape[0]
        out = C.c_double(0.0)
        pout = C.byref(out)

        def step(_keep=keep):
            rc = lib.hyphy_hip_build_q(h, nco, pco)
            if rc == 0:
                rc = lib.hyphy_hip_evaluate_built_exchange(h, cat, pun, nun, pqn, nqn, prf, pout)
            if rc:
                _check(rc)
            return out.value
        return step

    def last_allreduce_ms(self) -> float:
        """Event-timed duration of the last in-stream all-reduce (``set_all_timings(True)``), ms."""
        return float(self._lib.hyphy_hip_last_allreduce_ms(self._h))

    def prepare_built_categories_step(self, update_nodes, q_nodes, weights, root_freqs, coeffs: np.ndarray):
        """``build_q`` (C*n_q coefficient rows, class-major) + ``evaluate_categories_built`` -> log-L."""
        un = np.ascontiguousarray(update_nodes, dtype=np.int64)
        qn = np.ascontiguousarray(q_nodes, dtype=np.int64)
        rf = np.ascontiguousarray(root_freqs, dtype=np.float64)
        wt = np.ascontiguousarray(weights, dtype=np.float64)
        assert coeffs.flags.c_contiguous and coeffs.dtype == np.float64 and coeffs.shape[0] == self.C * len(qn)
        keep = (un, qn, rf, wt, coeffs)
        lib, h = self._lib, self._h
        pun, pqn, prf, pwt, pco = _l(un), _l(qn), _d(rf), _d(wt), _d(coeffs)
        nun, nqn, nco = len(un), len(qn), coeffs.shape[0]
        out = C.c_double(0.0)
        pout = C.byref(out)

        def step(_keep=keep):
            rc = lib.hyphy_hip_build_q(h, nco, pco)
            if rc == 0:
                rc = lib.hyphy_hip_evaluate_categories_built(h, pun, nun, pqn, nqn, pwt, prf, pout)
            if rc:
                _check(rc)
            return out.value
        return step

    def evaluate_categories_built_sites(self, update_nodes, q_nodes, weights, root_freqs, coeffs: np.ndarray):
        """``build_q`` (C*n_q coefficient rows, class-major) + ``evaluate_categories_built_sites`` -> (log-L, mixed per-pattern
        likelihoods, mixed 2^64 exponents) — what the host's weighted-sum category loop leaves in its buffer and scalers."""
        un = np.ascontiguousarray(update_nodes, dtype=np.int64)
        qn = np.ascontiguousarray(q_nodes, dtype=np.int64)
        rf = np.ascontiguousarray(root_freqs, dtype=np.float64)
        wt = np.ascontiguousarray(weights, dtype=np.float64)
        co = np.ascontiguousarray(coeffs, dtype=np.float64)
        assert co.shape[0] == self.C * len(qn)
        out = C.c_double(0.0)
        sl = np.zeros(self.S)
        sc = np.zeros(self.S, dtype=np.int64)
        _check(self._lib.hyphy_hip_build_q(self._h, co.shape[0], _d(co)))
        _check(self._lib.hyphy_hip_evaluate_categories_built_sites(self._h, _l(un), len(un), _l(qn), len(qn), _d(wt), _d(rf),
                                                                   C.byref(out), _d(sl), _l(sc)))
        return out.value, sl, sc

    def evaluate_categories(self, update_nodes, q_nodes, q_dense, weights, root_freqs, q_is_probability: bool = False,
                            per_site: bool = False):
        un = np.ascontiguousarray(update_nodes, dtype=np.int64)
        qn = np.ascontiguousarray(q_nodes, dtype=np.int64)
        q = np.ascontiguousarray(q_dense, dtype=np.float64) if len(qn) else None
        w = np.ascontiguousarray(weights, dtype=np.float64)
        rf = np.ascontiguousarray(root_freqs, dtype=np.float64)
        out = C.c_double(0.0)
        sl = np.zeros(self.S) if per_site else None
        sc = np.zeros(self.S, dtype=np.int64) if per_site else None
        _check(self._lib.hyphy_hip_evaluate_categories(self._h, _l(un), len(un), _l(qn), len(qn), _d(q),
                                                       int(q_is_probability), _d(w), _d(rf), C.byref(out), _d(sl),
                                                       _l(sc)))
        return (out.value, sl, sc) if per_site else out.value

    def download_partials(self, cat: int = 0) -> Tuple[np.ndarray, np.ndarray]:
        cache = np.zeros((self.I, self.S, self.D))
        counts = np.zeros((self.I, self.S), dtype=np.int64)
        _check(self._lib.hyphy_hip_download_partials(self._h, cat, _d(cache), _l(counts)))
        return cache, counts

    def set_pinned_states(self, node=None, states=None):
        """Fix node ``node`` (node code) to ``states[pattern]`` for the evaluations that follow; ``None`` removes it."""
        if node is None:
            _check(self._lib.hyphy_hip_set_pinned_states(self._h, -1, None))
            return
        st = np.ascontiguousarray(states, dtype=np.int64)
        assert st.shape == (self.S,)
        _check(self._lib.hyphy_hip_set_pinned_states(self._h, int(node), _l(st)))

    # -- branch cache (one-branch line searches) --------------------------------------------------
    def branch_cache_build(self, node: int, cat: int = 0):
        """Prepare the device branch cache for branch ``node`` (call after an ``evaluate``)."""
        _check(self._lib.hyphy_hip_branch_cache_build(self._h, cat, int(node)))

    def branch_cache_evaluate(self, node: int, q_dense: np.ndarray, cat: int = 0, q_is_probability: bool = False,
                              per_site: bool = False):
        """log-L with branch ``node``'s matrix replaced by exp(q_dense); everything else as at build time."""
        q = np.ascontiguousarray(q_dense, dtype=np.float64)
        out = C.c_double(0.0)
        sl = np.zeros(self.S) if per_site else None
        sc = np.zeros(self.S, dtype=np.int64) if per_site else None
        _check(self._lib.hyphy_hip_branch_cache_evaluate(self._h, cat, int(node), _d(q), int(q_is_probability),
                                                         C.byref(out), _d(sl), _l(sc)))
        return (out.value, sl, sc) if per_site else out.value

    def prepare_branch_cache_step(self, node: int, q_dense: np.ndarray, cat: int = 0):
        """Zero-argument callable evaluating the cached branch with the CURRENT contents of ``q_dense``
        (update it in place between calls); arguments marshalled once."""
        assert q_dense.flags.c_contiguous and q_dense.dtype == np.float64
        lib, h, pq, out = self._lib, self._h, _d(q_dense), C.c_double(0.0)
        ref = C.byref(out)

        def step(_keep=q_dense):
            _check(lib.hyphy_hip_branch_cache_evaluate(h, cat, int(node), pq, 0, ref, None, None))
            return out.value
        return step

    # -- device-side Q construction -------------------------------------------------------------
    def evaluate_built(self, update_nodes, q_nodes, root_freqs, cat: int = -1, per_site: bool = False):
        """Synchronous evaluation over the rate matrices staged by ``build_q`` (one coefficient row per entry of q_nodes);
        ``per_site``: also the per-pattern values and 2^64 exponents (hyphy_hip_evaluate_built_sites)."""
        un = np.ascontiguousarray(update_nodes, dtype=np.int64)
        qn = np.ascontiguousarray(q_nodes, dtype=np.int64)
        rf = np.ascontiguousarray(root_freqs, dtype=np.float64)
        out = C.c_double(0.0)
        sl = np.zeros(self.S) if per_site else None
        sc = np.zeros(self.S, dtype=np.int64) if per_site else None
        _check(self._lib.hyphy_hip_evaluate_built_sites(self._h, cat, _l(un), len(un), _l(qn), len(qn), _d(rf), C.byref(out), _d(sl), _l(sc)))
        return (out.value, sl, sc) if per_site else out.value

    def set_q_templates(self, templates: np.ndarray):
        t = np.ascontiguousarray(templates, dtype=np.float64)
        _check(self._lib.hyphy_hip_set_q_templates(self._h, t.shape[0], _d(t)))

    def build_q(self, coeffs: np.ndarray):
        c = np.ascontiguousarray(coeffs, dtype=np.float64)
        _check(self._lib.hyphy_hip_build_q(self._h, c.shape[0], _d(c)))

    def site_fits_evaluate(self, branch_group, branch_coeffs, site_mult, root_freqs) -> np.ndarray:
        """Per-site batched fits: ``site_mult`` is [n_sets, S, n_groups, K] (or [S, n_groups, K]);
        returns the site log-likelihoods [n_sets, S] (or [S]) of every pattern under its own multipliers."""
        bg = np.ascontiguousarray(branch_group, dtype=np.int64)
        bc = np.ascontiguousarray(branch_coeffs, dtype=np.float64)
        sm = np.ascontiguousarray(site_mult, dtype=np.float64)
        single = sm.ndim == 3
        if single:
            sm = sm[None]
        n_sets, S, G, K = sm.shape
        assert S == self.S and bg.shape == (self.B,) and bc.shape == (self.B, K)
        rf = np.ascontiguousarray(root_freqs, dtype=np.float64)
        out = np.zeros((n_sets, S))
        _check(self._lib.hyphy_hip_site_fits_evaluate(self._h, n_sets, G, _l(bg), _d(bc), _d(sm), _d(rf), _d(out)))
        return out[0] if single else out

    def site_fits_evaluate_mixture(self, branch_group, branch_coeffs, site_mult, site_weights, root_freqs) -> np.ndarray:
        """Branch-site mixture per site: ``site_mult`` [n_sets, S, n_mix, n_groups, K], ``site_weights`` [n_sets, S, n_mix]
        (or both without the leading n_sets axis); returns site log-likelihoods [n_sets, S] (or [S])."""
        bg = np.ascontiguousarray(branch_group, dtype=np.int64)
        bc = np.ascontiguousarray(branch_coeffs, dtype=np.float64)
        sm = np.ascontiguousarray(site_mult, dtype=np.float64)
        sw = np.ascontiguousarray(site_weights, dtype=np.float64)
        single = sm.ndim == 4
        if single:
            sm, sw = sm[None], sw[None]
        n_sets, S, M, G, K = sm.shape
        assert S == self.S and sw.shape == (n_sets, S, M) and bg.shape == (self.B,) and bc.shape == (self.B, K)
        rf = np.ascontiguousarray(root_freqs, dtype=np.float64)
        out = np.zeros((n_sets, S))
        _check(self._lib.hyphy_hip_site_fits_evaluate_mixture(self._h, n_sets, G, M, _l(bg), _d(bc), _d(sm), _d(sw), _d(rf), _d(out)))
        return out[0] if single else out

    def site_fits_kernel_ms(self) -> float:
        return float(self._lib.hyphy_hip_site_fits_kernel_ms(self._h))

    def q_buffer(self) -> int:
        return int(self._lib.hyphy_hip_q_buffer(self._h))

    def synchronize(self):
        _check(self._lib.hyphy_hip_synchronize(self._h))

    def stream(self) -> int:
        return int(self._lib.hyphy_hip_stream(self._h) or 0)

    def set_stream(self, stream_ptr: int):
        _check(self._lib.hyphy_hip_set_stream(self._h, C.c_void_p(stream_ptr)))

    def prune_timings(self, n: int) -> np.ndarray:
        """Durations (ms) of the pruning launches of the last ``n`` evaluations (HIP event ring; queried
        only now, never while evaluations run)."""
        out = np.zeros(int(n))
        m = int(self._lib.hyphy_hip_prune_timings(self._h, _d(out), int(n)))
        return out[:m]

    def prune_launches(self) -> int:
        return int(self._lib.hyphy_hip_prune_launches(self._h))

    def prune_kernel_name(self) -> str:
        return self._lib.hyphy_hip_prune_kernel_name(self._h).decode()

    def schedule_info(self) -> str:
        return self._lib.hyphy_hip_schedule_info(self._h).decode()

    def set_repeats(self, on: bool):
        """Subtree repeats (class-compressed lower phase) on / off for this partition; same results either way."""
        _check(self._lib.hyphy_hip_set_repeats(self._h, 1 if on else 0))

    def repeat_stats(self) -> dict:
        out = np.zeros(8, dtype=np.int64)
        _check(self._lib.hyphy_hip_repeat_stats(self._h, _l(out)))
        keys = ("available", "tables", "table_rows", "trunk_internal", "trunk_leaves", "edge_products_on", "edge_products_off", "in_use")
        return {k: int(v) for k, v in zip(keys, out)}

    def set_all_timings(self, on: bool):
        """Also stamp the expm and reduction kernels of the evaluations that follow (``last_timings()[0]``, ``[2]``)."""
        _check(self._lib.hyphy_hip_set_timing_detail(self._h, 1 if on else 0))

    def last_timings(self) -> np.ndarray:
        out = np.zeros(3)
        _check(self._lib.hyphy_hip_last_timings(self._h, _d(out)))
        return out
