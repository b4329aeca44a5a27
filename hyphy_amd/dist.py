"""Site-pattern sharding across ranks (one process per GPU) and the single collective of the path.

The reference splits a partition's patterns into contiguous blocks over OpenMP threads
(``likefunc.cpp:10995-11044``) and combines the block results on the host (``:11046-11093``);
across processes it only has point-to-point MPI.  Here every rank owns one
contiguous pattern range, computes its partial log-likelihood on its own MI355X, and the partials
are summed with ONE all-reduce per evaluation (``torch.distributed`` backend "nccl" == RCCL over
xGMI on ROCm; "gloo" in the CPU tests).  The payload is a single double: latency-, not
bandwidth-bound.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


def shard_range(n_patterns: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, disjoint, exhaustive pattern ranges; sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return (n_patterns * rank) // world, (n_patterns * (rank + 1)) // world


def shard_patterns(leaf_codes: np.ndarray, pattern_freq: np.ndarray, rank: int, world: int):
    lo, hi = shard_range(leaf_codes.shape[1], rank, world)
    return np.ascontiguousarray(leaf_codes[:, lo:hi]), np.ascontiguousarray(pattern_freq[lo:hi]), (lo, hi)


def allreduce_logl(partial, group=None):
    """Sum the per-rank partial log-likelihoods in place (``partial``: 1-element float64 tensor that
    lives where the backend expects it — device memory for RCCL).  -inf (a zero-likelihood pattern
    on some rank) and NaN propagate through the sum exactly as in the single-process combine."""
    import torch.distributed as dist
    dist.all_reduce(partial, op=dist.ReduceOp.SUM, group=group)
    return partial


def allgather_sites(local_vals, n_patterns: int, rank: int, world: int, group=None):
    """Per-site mode (``storageVec`` / ``siteCorrections``): concatenate the shards' vectors.
    Shards differ in length by at most one, so pad to the longest and trim."""
    import torch
    import torch.distributed as dist
    sizes = [shard_range(n_patterns, r, world)[1] - shard_range(n_patterns, r, world)[0] for r in range(world)]
    m = max(sizes)
    buf = torch.zeros(m, dtype=local_vals.dtype, device=local_vals.device)
    buf[: local_vals.numel()] = local_vals
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    return torch.cat([o[:n] for o, n in zip(out, sizes)])


# ---- rate classes spread over ranks (second form) ---------------------------------------------------
# The reference's MPI "category" mode (likefunc.cpp:2708-2768 set-up, likefunc2.cpp:595-696 per evaluation): every MPI node
# holds the WHOLE alignment, computes the per-site conditional likelihoods of the rate classes it was dealt, and the master
# mixes them (weighted-sum mode, likefunc2.cpp:820-853).  Here: class c lives on rank c mod world; every rank evaluates its
# classes over all patterns (hyphy_hip_evaluate with `cat` and per-site outputs), ONE all-gather moves the per-site
# (likelihood, exponent) rows — 12 bytes per pattern and class, the only data-path collective of this mode — and every rank
# mixes (no master: the result is on every rank, like the all-reduce of the site-sharded mode).  The preferred form for the
# headline workloads stays the inner batch dimension on each GPU (hyphy_hip_evaluate_categories); this one is for C >= G
# when the alignment is too small to shard by sites.

def local_classes(n_classes: int, rank: int, world: int):
    """Rate classes owned by ``rank``: c with c mod world == rank (ascending)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return [c for c in range(n_classes) if c % world == rank]


def mix_classes(weights, site_lik, site_scalers, pattern_freq):
    """Weighted-sum category mixing on tensors (the device analogue is mix_categories_kernel + site_reduce_kernel):
    ``site_lik`` / ``site_scalers`` [C, S] per-pattern likelihoods and 2^64-exponents of every class, ``weights`` [C],
    ``pattern_freq`` [S].  mixed_s = sum_c w_c l_cs 2^(-64 (k_cs - min_c k_cs)) (likefunc2.cpp:820-853), log L =
    sum_s f_s log mixed_s - 64 ln2 sum_s f_s min_c k_cs, a pattern with mixed_s <= 0 contributes the reference's myLog
    floor -1e6 f_s and no exponent (likefunc.cpp:644-661).  Returns a 0-dim float64 tensor."""
    import math
    import torch
    w = torch.as_tensor(weights, dtype=torch.float64, device=site_lik.device)
    k = site_scalers.to(torch.float64)
    kmin = k.min(dim=0).values
    mixed = (w[:, None] * site_lik * torch.exp2(-64.0 * (k - kmin[None, :]))).sum(dim=0)
    f = torch.as_tensor(pattern_freq, dtype=torch.float64, device=site_lik.device)
    pos = mixed > 0
    lg = torch.where(pos, torch.log(torch.where(pos, mixed, torch.ones_like(mixed))), torch.full_like(mixed, -1000000.0))
    return (lg * f).sum() - 64.0 * math.log(2.0) * (torch.where(pos, kmin, torch.zeros_like(kmin)) * f).sum()


def allgather_classes(lik_local, sc_local, n_classes: int, rank: int, world: int, group=None):
    """All ranks' per-class rows in class order: ``lik_local`` / ``sc_local`` [len(local_classes), S] on this rank ->
    ([C, S], [C, S]) on every rank.  Ranks own ceil(C / world) or floor(C / world) classes: rows are padded to the larger
    count for the collective and dropped afterwards."""
    import torch
    import torch.distributed as dist
    mine = local_classes(n_classes, rank, world)
    if lik_local.shape[0] != len(mine) or sc_local.shape != lik_local.shape:
        raise ValueError("one row per local class expected")
    per = -(-n_classes // world)
    S = lik_local.shape[1]
    buf = torch.zeros((per, 2, S), dtype=torch.float64, device=lik_local.device)
    buf[: len(mine), 0] = lik_local
    buf[: len(mine), 1] = sc_local.to(torch.float64)       # (exponents are small integers: exact in a double)
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    lik = torch.zeros((n_classes, S), dtype=torch.float64, device=lik_local.device)
    sc = torch.zeros((n_classes, S), dtype=torch.float64, device=lik_local.device)
    for r in range(world):
        for row, c in enumerate(local_classes(n_classes, r, world)):
            lik[c], sc[c] = out[r][row, 0], out[r][row, 1]
    return lik, sc


def evaluate_classes_spread(evaluate_class, weights, pattern_freq, rank: int, world: int, device=None, group=None):
    """One evaluation of a C-class model with the classes dealt over the ranks.  ``evaluate_class(c)`` returns this rank's
    per-pattern (likelihood [S], exponent [S]) of class c over the WHOLE alignment (``HipPartition.evaluate(..., cat=c,
    per_site=True)[1:]`` on a GPU rank; a CPU stand-in in the gloo tests); ``device``: where the collective's tensors live
    ("cuda" for RCCL).  Returns log L (float), identical on every rank."""
    import torch
    C = len(weights)
    rows = [evaluate_class(c) for c in local_classes(C, rank, world)]
    S = len(pattern_freq)
    lik = torch.zeros((len(rows), S), dtype=torch.float64, device=device)
    sc = torch.zeros((len(rows), S), dtype=torch.float64, device=device)
    for k, (l, e) in enumerate(rows):
        lik[k] = torch.as_tensor(np.asarray(l, dtype=np.float64), device=device)
        sc[k] = torch.as_tensor(np.asarray(e, dtype=np.float64), device=device)
    if world > 1:
        lik, sc = allgather_classes(lik, sc, C, rank, world, group=group)
    return float(mix_classes(weights, lik, sc, pattern_freq))
