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
