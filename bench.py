#!/usr/bin/env python3
"""bench.py — full-tree log-L evaluations / second of the MI355X likelihood core.

One "step" = one full-tree log-likelihood evaluation after a GLOBAL parameter change (omega is
swept 0.3 + 0.001 k), i.e. per step: device-side rate-matrix build for every branch,
batched matrix exponential of all L+I-1 branches, one full Felsenstein pruning pass, root
reduction, (N > 1: one RCCL all-reduce of the partition log-likelihood), log-L back on the host.
That is exactly what the reference's timing loop `R = ...; LFCompute (lf, res);` does per
iteration.  Inputs (alignment, tree, rate-matrix templates) are resident in HBM
before the timed region starts; each step is synchronous (the value is needed by the caller).

Workloads:  mg94_64x10k (default; BASELINE.json metric: 61-state MG94 codon, 64 taxa x 10k codons)
            mg94_32x5k (configs[1]),  mg94_128x100k (configs[3]),  hky_8x1k (configs[0])

Launch for N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
                   --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W
Patterns are sharded contiguously over the ranks (one process per GPU).  The collective is the library's own
(`--collective cabi`, default): rank 0 makes an RCCL unique id through the C-ABI, the 128 bytes travel over the
torch.distributed store, every rank calls hyphy_hip_comm_init_rank, and a step is hyphy_hip_build_q +
hyphy_hip_evaluate_built_allreduce — local evaluation, ONE ncclAllReduce of one double on the partition's stream, the sum
back through the host-mapped record.  (`--collective torch`: torch.distributed.all_reduce on the same stream instead;
torch.distributed is otherwise only used for the barriers around the timed region and to gather the ranks' timings.)
Total work is fixed -> "strong".  `--single-process --gpus N [--combine host|rccl]` is the other form (one host thread
drives N devices, HyPhy proper): shard partials combined on the host (Neumaier) or by one RCCL group all-reduce.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from hyphy_amd import data, hip, models  # noqa: E402
from hyphy_amd import dist as hdist  # noqa: E402

POS_FREQS = np.array([[0.30, 0.20, 0.25, 0.25], [0.20, 0.30, 0.30, 0.20], [0.25, 0.25, 0.20, 0.30]])
REV = dict(AC=0.5, AT=0.4, CG=0.4, CT=1.2, GT=0.4)
NUC_FREQS = np.array([0.35, 0.15, 0.2, 0.3])
WORKLOADS = {
    "mg94_64x10k": dict(taxa=64, sites=10000, unit=3, seed=3),
    "mg94_32x5k": dict(taxa=32, sites=5000, unit=3, seed=2),
    "mg94_64x1250": dict(taxa=64, sites=1250, unit=3, seed=3),    # one rank's share of the headline workload at 8 GPUs
    "mg94_64x2500": dict(taxa=64, sites=2500, unit=3, seed=3),    # ... at 4 GPUs
    "mg94_64x5000": dict(taxa=64, sites=5000, unit=3, seed=3),    # ... at 2 GPUs
    "mg94_128x100k": dict(taxa=128, sites=100000, unit=3, seed=4),
    "mg94_64x20k": dict(taxa=64, sites=20000, unit=3, seed=3),    # mid sizes: where the lower phase's defaults change (repeats.hip: rho)
    "mg94_64x40k": dict(taxa=64, sites=40000, unit=3, seed=3),
    "hky_8x1k": dict(taxa=8, sites=1000, unit=1, seed=1),
    # configs[2]: BUSTED-style, 3 omega classes (weights .7/.25/.05, omega .1/1/5 scaled by the swept factor),
    # classes batched into one expm launch + one pruning launch, mixed on the device
    "busted3_64x10k": dict(taxa=64, sites=10000, unit=3, seed=3, classes=3),
    # configs[2] in the form BUSTED really runs: explicit-form branch-site mixture, P_b = sum_m w_m exp(Q_b(omega_m)), M = 3 components
    # (weights .6/.3/.1, omegas .1/1/5 x the swept factor); 375 component matrices built and exponentiated on the device per step,
    # mixed into 125 transition matrices, ONE pruning pass (hyphy_hip_build_q + hyphy_hip_evaluate_mixture_built)
    "mix3_64x10k": dict(taxa=64, sites=10000, unit=3, seed=3, mixture=3),
    "gtr_32x50k": dict(taxa=32, sites=50000, unit=1, seed=5, p_change=0.25),   # one partition of configs[4]
    "gtr_32x1m": dict(taxa=32, sites=1000000, unit=1, seed=6, p_change=0.25),  # same shape, large enough to leave the L2/MALL
}
FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X FP64 matrix (== vector) peak, AMD datasheet; see DESIGN.md §roofline
HBM_PEAK_GBS = 8000.0
HBM_ACHIEVABLE_GBS = 6300.0    # what a streaming kernel reaches (MI355X_MICROARCH.md)


def measured_instruction_peak():
    """Chip-wide rate v_mfma_f64_16x16x4_f64 sustains with VGPR accumulators (tools/ubench/mfma4_skew.hip, committed run of r04):
    the best VGPR line, or None when the file is not there.  (r01-r03 quoted 49 TFLOP/s from profiles/r01_ubench_mfma_f64.txt:
    that microbenchmark had its accumulators in AGPRs, which halves the issue rate — profiles/r04_ubench_mfma_agpr_vs_vgpr.txt.)"""
    import re
    path = os.path.join(ROOT, "profiles", "r04_ubench_mfma_agpr_vs_vgpr.txt")
    try:
        vals = [float(m.group(1)) for m in (re.search(r"VGPR accumulators.*?([0-9.]+) TF", ln) for ln in open(path)) if m]
        return (max(vals), "profiles/r04_ubench_mfma_agpr_vs_vgpr.txt") if vals else (None, None)
    except OSError:
        return None, None


# The library brackets the pruning launches of an evaluation with a HIP event pair on its stream (two barrier
# packets, ~5 us per step at the headline size).  The bench keeps one evaluation in TIMING_EVERY stamped: the
# kernel durations are still measured live inside the timed region, on a quarter of its steps.
TIMING_EVERY = 4
os.environ.setdefault("HYPHY_HIP_TIMING_EVERY", str(TIMING_EVERY))
TIMING_EVERY = max(1, int(os.environ["HYPHY_HIP_TIMING_EVERY"]))


def templates_for(unit):
    """Q_b = t_b * T0 + (t_b * omega) * T1  (off-diagonal); device builds the diagonal."""
    if unit == 3:
        T = np.zeros((2, 61, 61))
        rv = dict(REV, AG=1.0)
        for (i, j, name, ns, pf) in models.mg94rev_template(POS_FREQS):
            T[1 if ns else 0, i, j] = rv[name] * pf
        return T, models.f3x4_codon_freqs(POS_FREQS)
    T = np.zeros((2, 4, 4))
    rv = dict(models.hky85_rev(0.35), AG=1.0)
    for i in range(4):
        for j in range(4):
            if i != j:
                T[0, i, j] = rv[models.REV_NAMES[(min(i, j), max(i, j))]] * NUC_FREQS[j]
    return T, NUC_FREQS


COLLECTIVE_LABEL = {"cabi": "hyphy_hip_evaluate_built_allreduce (in-stream ncclAllReduce of one double, C-ABI communicator)",
                    "host": "hyphy_hip_evaluate_built_exchange (no device collective: every rank's partial through a shared-memory segment, "
                            "Neumaier sum in rank order on every rank)",
                    "torch": "torch.distributed.all_reduce on the partition's stream", "none": None}
PER_RANK_KEYS = ("patterns", "kernel_ms", "expm_ms", "reduce_ms", "allreduce_ms")


def gather_per_rank(dist, ctl, world, patterns, kernel_ms, expm_ms, reduce_ms, allreduce_ms):
    """Every rank's own numbers on every rank (the line is printed by rank 0): shard size and kernel / expm / reduction /
    all-reduce milliseconds per step.  Only torch.distributed is touched: tests/test_distributed_cpu.py runs it over gloo."""
    import torch
    mine = torch.tensor([float(patterns), float(kernel_ms), float(expm_ms or 0.0), float(reduce_ms or 0.0), float(allreduce_ms or 0.0)],
                        dtype=torch.float64, device=ctl)
    allr = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    return [dict(rank=r, patterns=int(v[0].item()), **{k: float(v[i + 1].item()) for i, k in enumerate(PER_RANK_KEYS[1:])})
            for r, v in enumerate(allr)]


def max_over_ranks(dist, ctl, seconds):
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=ctl)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def multi_gpu_fields(world, collective, collective_note, per_rank, allreduce_ms, collective_ab):
    """What the line of an N > 1 run carries on top of the N = 1 line: (config additions, roofline additions, top-level
    additions).  `collective_ab`: ms per step of the SAME rank set with the library's in-stream all-reduce and with
    torch.distributed's, measured back to back behind the timed region (None: not measured)."""
    cfg = {"parallelism": f"site-shard x{world}, one process per GPU", "collective": COLLECTIVE_LABEL[collective]}
    if collective_note:
        cfg["collective_note"] = collective_note
    roof = {}
    if allreduce_ms is not None:
        roof["allreduce_ms"] = allreduce_ms
    top = {"per_rank": per_rank}
    if collective_ab:
        top["collective_ab"] = collective_ab
    return cfg, roof, top


def alg_work(D, S, L, I):
    """SURVEY §8d algorithmic work of one pruning pass (per full-tree evaluation)."""
    flops = S * ((I - 1) * (2 * D * D + 2 * D) + L * D + 2 * D)
    bytes_ = S * ((2 * (I - 1) + 1) * 8 * D + L)
    return flops, bytes_


def cpu_thread_sweep(wl, syn, omega0, t_branch, candidates, seconds_each=3.0):
    """Short differenced runs of the reference binary at each thread count (same script as cpu_baseline): evals/s per count.
    The unmodified reference has a sweet spot (its OpenMP blocks: 16 of 128 cores won the r01 sweep); the count that the
    long run uses is chosen HERE, on this box, not taken from a profile of another one."""
    from oracle import hbl
    from hyphy_amd import tree as htree
    tmpl = models.mg94rev_template(POS_FREQS)
    pi = models.f3x4_codon_freqs(POS_FREQS)
    g = dict(R=omega0, **REV)
    bt = {n: t_branch for n in syn.flat.branch_names()}
    scale = (syn.flat.L / 64.0) * (syn.states.shape[1] / 10000.0)
    out = {}
    for thr in candidates:
        rate0 = 7.0 * min(thr, 8)              # rough expectation, only sizes the run
        n_long = int(max(8, min(2000, seconds_each * rate0 / scale)))
        n_short = max(2, n_long // 8)
        secs = {}
        for n in (n_short, n_long):
            t0 = time.perf_counter()
            hbl.evaluate(names=syn.flat.leaf_names, seqs=syn.seqs, newick=htree.to_newick(syn.tree), unit=3,
                         model_block=hbl.codon_model_block(tmpl, pi), model_name="MGM", globals_=g, branch_t=bt,
                         sweep=dict(param="R", start=omega0, step=0.001, n=n, record=0), threads=thr, per_site=False, timeout=600.0)
            secs[n] = time.perf_counter() - t0
        out[thr] = (n_long - n_short) / max(secs[n_long] - secs[n_short], 1e-3)
    return out


def cpu_baseline(wl, syn, omega0, t_branch, n_threads, steps, budget_s=20.0):
    """Reference CPU path timed on this host: the real hyphy binary (oracle/_ref).  Bounded sample of the same
    workload: the LFCompute sweep of SURVEY A.8 at ONE thread and at `n_threads` (the best count of the sweep in
    profiles/), each timed as the difference of two runs of different length (wall clock of the whole process:
    start-up, data reading and the thread-benchmark Optimize cancel out) with >= ~30 s between them.  The long
    best-thread run also records the log-likelihood of every point the GPU loop evaluates, the one-thread run the
    per-site log-likelihoods at the first point."""
    from oracle import hbl
    from hyphy_amd import tree as htree
    if not (hbl.have_reference() and wl["unit"] == 3):
        return None, None
    tmpl = models.mg94rev_template(POS_FREQS)
    pi = models.f3x4_codon_freqs(POS_FREQS)
    g = dict(R=omega0, **REV)
    bt = {n: t_branch for n in syn.flat.branch_names()}
    scale = (syn.flat.L / 64.0) * (syn.states.shape[1] / 10000.0)   # cost of one evaluation relative to the headline workload
    out = {}
    ref = {}
    for thr, rate0 in ((1, 8.0), (n_threads, 75.0)):
        if thr in out:
            continue
        n_long = int(max(12, min(20000, budget_s * rate0 / scale)))
        n_short = max(2, n_long // 10)
        secs = {}
        for n in (n_short, n_long):
            rec = min(steps, n) if (thr == n_threads and n == n_long) else 0
            t0 = time.perf_counter()
            res = hbl.evaluate(names=syn.flat.leaf_names, seqs=syn.seqs, newick=htree.to_newick(syn.tree), unit=3,
                               model_block=hbl.codon_model_block(tmpl, pi), model_name="MGM", globals_=g, branch_t=bt,
                               sweep=dict(param="R", start=omega0, step=0.001, n=n, record=rec), threads=thr,
                               per_site=(thr == 1), timeout=1800.0)
            secs[n] = time.perf_counter() - t0
            ref["logl"] = res["logl"]
            if rec:
                ref["sweep_values"] = res.get("sweep_values")
            if thr == 1 and "site_logl" in res:
                ref["site_logl"] = res["site_logl"]
        dt = max(secs[n_long] - secs[n_short], 1e-3)
        out[thr] = dict(value=(n_long - n_short) / dt, cores=thr, evals=n_long - n_short, seconds=dt)
    best = out[n_threads]
    cb = dict(value=best["value"], unit="evals/s", cores=n_threads, kind="reference",
              sample=f"{best['evals']} LFCompute calls with R swept in {best['seconds']:.1f} s (difference of two runs of "
                     f"the same script with different loop lengths, process wall clock) on the same alignment/tree, reference "
                     f"hyphy 2.5.100 built by oracle/Makefile.ref, NUMBER_THREADS={n_threads} (best of this run's thread sweep: cpu_baseline.thread_sweep)",
              seconds=best["seconds"],
              one_thread=dict(value=out[1]["value"], unit="evals/s", cores=1, evals=out[1]["evals"], seconds=out[1]["seconds"]))
    return cb, ref


MIX_OMEGAS = (0.1, 1.0, 5.0)
MIX_WEIGHTS = (0.6, 0.3, 0.1)


def cpu_baseline_mixture(wl, syn, t_branch, steps, n_threads=16, budget_s=20.0):
    """The reference on the explicit-form mixture workload (BS_REL.bf:48; tree.cpp:3047-3090): LFCompute with the second component's
    omega swept exactly as the GPU loop sweeps it, timed as the difference of two runs of different length, NUMBER_THREADS fixed at 16
    (the best count of the headline workload's sweep; no sweep of its own: one evaluation costs 3 x the headline's exponentials)."""
    from oracle import hbl
    from hyphy_amd import tree as htree
    if not hbl.have_reference():
        return None, None
    tmpl = models.mg94rev_template(POS_FREQS)
    pi = models.f3x4_codon_freqs(POS_FREQS)
    M = wl["mixture"]
    g = dict(REV)
    for m in range(M):
        g[f"R{m + 1}"] = MIX_OMEGAS[m]
    for m in range(M - 1):
        g[f"W{m + 1}"] = MIX_WEIGHTS[m]
    wexpr = [f"W{m}" for m in range(1, M)] + ["(1" + "".join(f"-W{m}" for m in range(1, M)) + ")"]
    block = hbl.codon_mixture_model_block(tmpl, pi, [f"R{m}" for m in range(1, M + 1)], wexpr)
    bt = {n: t_branch for n in syn.flat.branch_names()}
    n_long = int(max(12, budget_s * 14.0))
    n_short = max(2, n_long // 10)
    secs, ref = {}, {}
    for n in (n_short, n_long):
        rec = min(steps, n) if n == n_long else 0
        t0 = time.perf_counter()
        res = hbl.evaluate(names=syn.flat.leaf_names, seqs=syn.seqs, newick=htree.to_newick(syn.tree), unit=3, model_block=block,
                           model_name="MGM", globals_=g, branch_t=bt, upper_bounds={f"W{m}": 1.0 for m in range(1, M)},
                           sweep=dict(param="R2", start=MIX_OMEGAS[1], step=0.001, n=n, record=rec), threads=n_threads, per_site=False,
                           timeout=1800.0)
        secs[n] = time.perf_counter() - t0
        ref["logl"] = res["logl"]
        if rec:
            ref["sweep_values"] = res.get("sweep_values")
    dt = max(secs[n_long] - secs[n_short], 1e-3)
    cb = dict(value=(n_long - n_short) / dt, unit="evals/s", cores=n_threads, kind="reference",
              sample=f"{n_long - n_short} LFCompute calls of the explicit-form {M}-component mixture with R2 swept in {dt:.1f} s (difference of two "
                     f"runs of the same script with different loop lengths) on the same alignment/tree, reference hyphy built by "
                     f"oracle/Makefile.ref, NUMBER_THREADS={n_threads}", seconds=dt)
    return cb, ref


def measure_traffic(workload, kernels):
    """HBM traffic of the dominant kernel, measured by re-running THIS script under rocprofv3 with the L2's memory-side
    counters — two passes (FETCH_SIZE needs 3 of the 4 TCC slots, WRITE_SIZE 2: MI355X_MICROARCH.md, PMC section), counter
    collection only (no trace domains besides the kernel trace).  Returns per-launch bytes with the guide's gfx950
    correction (FETCH_SIZE counts 128-byte requests as 64 bytes for wide coalesced streams: doubled; WRITE_SIZE as is),
    or None when rocprofv3 is not available / the passes fail."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return None
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        tmp = tempfile.mkdtemp(prefix="hyprof_", dir="/tmp")
        env = dict(os.environ, TMPDIR="/tmp", HYPHY_HIP_TIMING_EVERY="1")
        if len(kernels) > 1:
            # a class-compressed pass: keep the counter run on that form — left to itself the library would first time the plain form too
            # (rep_decide), and those launches of the same pruning kernel move twice the bytes (they were most of an 18-step run's rows)
            env["HYPHY_HIP_REPEATS"] = "1"
        try:
            subprocess.run([exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", tmp, "--", sys.executable,
                            os.path.abspath(__file__), "--workload", workload, "--steps", "12", "--warmup", "6", "--no-cpu-baseline",
                            "--no-traffic"], cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240)
            total = 0.0
            per_kernel = {}
            for kernel in kernels:   # (a class-compressed pass is two launches or more: lower phase + trunk; their traffic adds up)
                rows = []
                for f in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
                    for r in csv.DictReader(open(f)):
                        if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
                            rows.append(float(r["Counter_Value"]))
                if len(rows) < 4:
                    return None
                per_kernel[kernel] = rows
            n_last = min(len(r) for r in per_kernel.values())   # launches of the kernel that runs once per evaluation
            for kernel, rows in per_kernel.items():
                per_eval = max(1, int(round(len(rows) / n_last)))   # (a lower phase of one launch per table level: several per evaluation)
                tail = rows[-8 * per_eval:]   # (steady state is the END of the run: the first passes persist every node / tune)
                total += float(np.median(tail)) if per_eval == 1 else per_eval * float(np.mean(tail))
            vals[counter] = total
        except Exception:
            return None
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return {"bytes_per_launch": (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, "FETCH_SIZE_KB": vals["FETCH_SIZE"],
            "WRITE_SIZE_KB": vals["WRITE_SIZE"], "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two passes of this script, median over "
            "the steady-state launches); bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB, the gfx950 correction of MI355X_MICROARCH.md"}


def cpu_port_baseline(pd, flat, Q, pi, sparse):
    from oracle import oracle
    t0 = time.time()
    P = oracle.expm(Q, sparse)
    op = oracle.OraclePartition(pd.D, flat.flat_parents, flat.L, pd.leaf_codes, None, pd.pattern_freq)
    nodes = np.arange(flat.n_branches, dtype=np.int64)
    op.set_P(nodes, P)
    ll = op.compute_block(nodes, pi)
    dt = time.time() - t0
    return dict(value=1.0 / dt, unit="evals/s", cores=1, kind="port",
                sample="1 full evaluation (expm of every branch + full pruning pass) by oracle/hyphy_oracle.c, scalar C"), ll


def site_fit_terms(mu):
    """Series length of sitefit.hip for a tile whose largest uniformisation rate (per sub-step) is mu."""
    mu = float(mu)
    if mu <= 0.:
        return 0
    n_sub = int(np.ceil(mu / 64.0))
    m = mu / n_sub
    wgt, j = np.exp(-m), 0
    while True:
        j += 1
        wgt *= m / j
        r = m / (j + 1)
        if r < 0.5 and wgt * r / (1.0 - r) < 1e-18:
            return j * n_sub


def time_site_fits(part, args, wl, pd, flat, T, pi, tb):
    """FEL-style per-site fits: every pattern has its own (alpha, beta_test, beta_nuisance); one launch evaluates
    SETS candidate vectors per pattern.  Reports site-evaluations/s, the MFMA rate of the kernel (flops counted from
    the series lengths the kernel uses) and the reference's way timed on this host: one single-site likelihood
    function per site = (L+I-1) matrix exponentials + a one-pattern pruning pass (oracle restatement, 1 core)."""
    from oracle import oracle
    rng = np.random.default_rng(7)
    S, B, D = part.S, part.B, part.D
    n_sets = args.site_fits
    bgroup = (rng.random(B) < 0.25).astype(np.int64)           # a quarter of the branches "tested"
    bcoef = np.stack([tb, 0.3 * tb], axis=1)                    # synonymous / non-synonymous lengths of the global fit
    smult = np.exp(rng.uniform(np.log(0.1), np.log(5.0), (n_sets, S, 2, 2)))   # (alpha_s, beta_s) per group, log-uniform
    smult[:, :, 1, 0] = smult[:, :, 0, 0]                      # alpha is shared by the two groups
    part.site_fits_evaluate(bgroup, bcoef, smult, pi)          # warm-up (allocations, schedule)
    reps = max(1, min(args.steps, 10))
    t0 = time.perf_counter()
    kern = 0.0
    for _ in range(reps):
        out = part.site_fits_evaluate(bgroup, bcoef, smult, pi)
        kern += part.site_fits_kernel_ms()
    dt = (time.perf_counter() - t0) / reps
    kern /= reps
    # flops: per (set, tile, branch) terms x K templates x NW*NKK MFMAs x 2048
    dmax = np.array([T[k].sum(1).max() for k in range(T.shape[0])])
    S_pad = (S + 15) // 16 * 16
    x = np.zeros((n_sets, S_pad, B, 2))
    x[:, :S] = smult[:, :, bgroup, :] * bcoef[None, None]
    mu_tile = (x @ dmax).reshape(n_sets, S_pad // 16, 16, B).max(2)
    mus, inv = np.unique(np.round(mu_tile, 6), return_inverse=True)
    terms = np.array([site_fit_terms(m) for m in mus])[inv.reshape(-1)]
    NW = (D + 15) // 16
    mfma = int(terms.sum()) * T.shape[0] * NW * 4 * NW
    flops = mfma * 2048.0
    # the reference's way on one host core, a few sites
    n_cpu = 6
    nodes = np.arange(B, dtype=np.int64)
    idx = np.arange(D)
    t1 = time.perf_counter()
    worst = 0.0
    for s_ in range(n_cpu):
        xs = smult[0, s_][bgroup] * bcoef
        Q = np.einsum("bk,kij->bij", xs, T)
        Q[:, idx, idx] = 0.0
        Q[:, idx, idx] = -Q.sum(2)
        op = oracle.OraclePartition(D, flat.flat_parents, flat.L, pd.leaf_codes[:, s_:s_ + 1], None, np.ones(1, dtype=np.int64))
        op.set_P(nodes, oracle.expm(Q, True))
        ref = op.site_log_likelihoods(nodes, pi)[0]
        worst = max(worst, abs(out[0, s_] - ref) / abs(ref))
    cpu_rate = n_cpu / (time.perf_counter() - t1)
    fel_fit = None
    if args.fel:
        from hyphy_amd import fel
        t2 = time.perf_counter()
        res = fel.fel(part, bgroup == 0, bcoef[:, 0], bcoef[:, 1], pi, max_iter=300)
        fel_fit = {"seconds": time.perf_counter() - t2, "launches": res.launches, "patterns": int(S),
                   "median_alpha": float(np.median(res.alpha)), "median_beta": float(np.median(res.beta)),
                   "sites_p_below_0.1": int((res.p_value < 0.1).sum()), "sum_logl_alt": float(res.logl_alt.sum()),
                   "sum_logl_null": float(res.logl_null.sum())}
        t3 = time.perf_counter()
        mres = fel.meme(part, bgroup == 0, bcoef[:, 0], bcoef[:, 1], pi, max_iter=150)
        fel_fit["meme_seconds"] = time.perf_counter() - t3
        fel_fit["meme_launches"] = mres.launches
        fel_fit["meme_sites_p_below_0.1"] = int((mres.p_value < 0.1).sum())
        fel_fit["meme_sum_logl_alt"] = float(mres.logl_alt.sum())
    return {**({"fel_fit": fel_fit} if fel_fit else {}), "sets_per_launch": n_sets, "patterns": int(S), "site_evals_per_s": n_sets * S / dt, "ms_per_launch": 1e3 * dt,
            "kernel_ms": kern, "kernel_site_evals_per_s": n_sets * S / (kern * 1e-3), "mean_series_terms": float(terms.mean()),
            "mfma_tflops": flops / (kern * 1e-3) / 1e12, "mfma_frac_of_peak": flops / (kern * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
            "cpu_reference_way": {"site_evals_per_s": cpu_rate, "cores": 1, "kind": "port",
                                  "sample": f"{n_cpu} sites: {B} matrix exponentials + one-pattern pruning each"},
            "max_rel_err_vs_cpu_sample": worst}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="mg94_64x10k", choices=sorted(WORKLOADS))
    ap.add_argument("--preheat-s", type=float, default=0.3,
                    help="seconds of untimed steps BEFORE the --warmup steps: the chip ramps its clocks over the first ~50 steps "
                         "after an idle period (per-step 191 -> 167 us), longer than a 5-step warm-up; reported as preheat_s")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="do not re-run under rocprofv3 for the HBM traffic of the dominant kernel")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--pipelined", action="store_true", help="also report throughput with no per-step host sync")
    ap.add_argument("--branch-cache", action="store_true",
                    help="also time one-branch line-search evaluations through the device branch cache (SURVEY 8f-1)")
    ap.add_argument("--site-fits", type=int, default=0, metavar="SETS",
                    help="also time per-site batched fits (SURVEY 8f-4): every pattern under its own (alpha, beta), "
                         "SETS candidate parameter vectors per pattern and launch")
    ap.add_argument("--collective", choices=["auto", "cabi", "host", "torch"], default="auto",
                    help="N > 1: who sums the ranks' partial log-likelihoods.  cabi (auto for N > 1): the library's own in-stream "
                         "ncclAllReduce (hyphy_hip_evaluate_built_allreduce); torch: torch.distributed.all_reduce on the same stream.  "
                         "With --gpus 1, cabi runs the same entry point on a one-rank communicator (its overhead on one GPU)")
    ap.add_argument("--single-process", action="store_true",
                    help="--gpus N from ONE process (hyphy_hip_create with device_count = N; not under torch.distributed.run)")
    ap.add_argument("--combine", choices=["host", "rccl"], default="host",
                    help="--single-process: shard partials summed on the host (Neumaier) or by one RCCL group all-reduce")
    ap.add_argument("--cold-s", type=float, default=0.25,
                    help="idle seconds before the un-preheated measurement that is reported as value_cold (0: skip it)")
    ap.add_argument("--fel", action="store_true",
                    help="with --site-fits: also run the whole FEL-style analysis (alternative + null fit of every "
                         "pattern, hyphy_amd/fel.py) and report its wall time")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    N = args.gpus
    dist = None
    single = bool(args.single_process)
    multi = N > 1 and not single           # one process per GPU
    share = multi and bool(os.environ.get("HYPHY_BENCH_SHARE_DEVICE"))
    if single:
        if world != 1:
            raise SystemExit("--single-process is one process driving N devices: do not launch it with torch.distributed.run")
        os.environ["HYPHY_HIP_COMBINE"] = args.combine
    elif N > 1:
        if world != N:
            raise SystemExit(f"--gpus {N} needs WORLD_SIZE={N} (launch with torch.distributed.run)")
        import torch
        import torch.distributed as dist
        if share:
            # DIAGNOSTIC (HYPHY_BENCH_SHARE_DEVICE=1): N ranks on ONE device — a 1-GPU box walks the N > 1 code of this file
            # (sharding, fall-backs, per-rank gather, the line).  RCCL refuses two ranks on a device, so the control plane is
            # gloo and the sum goes through torch.distributed; the numbers mean nothing, the line says so.
            local = 0
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl")   # == RCCL on ROCm (barriers, timing gather, the id broadcast)
    import torch
    torch.cuda.set_device(local if multi else 0)
    ctl = "cpu" if share else "cuda"       # where the control-plane tensors (flags, counts, timings) live
    collective = args.collective
    if share:
        collective = "torch"
    # (N > 1, auto: the C-ABI's RCCL all-reduce AND the host-side exchange are both set up, both are timed on a few untimed steps, the
    #  faster one runs the timed region — the line says which and carries both timings, `collective_choice`)
    want_host = multi and args.collective in ("host", "auto")
    collective_auto = collective == "auto" and multi and not share
    if collective == "auto":
        collective = "cabi" if multi else "none"
    if single or (N == 1 and collective in ("torch", "host")):
        collective = "none"

    wl = WORKLOADS[args.workload]
    syn = data.evolve(wl["taxa"], wl["sites"], wl["unit"], seed=wl["seed"], p_change=wl.get("p_change", 0.04))
    D = 61 if wl["unit"] == 3 else 4
    # nucleotide workloads keep every site as its own pattern (4^32 possible columns: the interesting
    # regime for the HBM-bound kernel is S = sites; compression would leave a few thousand patterns)
    pd_all = data.from_states(syn.states, D, compress_patterns=(D > 4))
    flat = syn.flat
    S_all, L, I, B = pd_all.S, flat.L, flat.I, flat.n_branches
    # contiguous pattern shard of this rank (the reference's OpenMP site blocks, likefunc.cpp:10995-11044)
    codes, freq, (lo, hi) = hdist.shard_patterns(pd_all.leaf_codes, pd_all.pattern_freq, rank, N if multi else 1)

    T, pi = templates_for(wl["unit"])
    t_branch = 0.05
    tb = np.full(B, t_branch)
    nodes = np.arange(B, dtype=np.int64)
    omega0 = 0.3

    n_classes = wl.get("classes", 1)
    n_mix = wl.get("mixture", 0)
    part = hip.HipPartition(D, flat.flat_parents, L, codes, None, freq, C_cat=n_classes,
                            device_first=(local if multi else 0), device_count=(N if single else 1))
    part.set_q_templates(T)
    if single and args.combine == "rccl":
        part.comm_init_all()
    stream = torch.cuda.Stream()             # everything (kernels, RCCL, the .item() copy) in ONE stream
    torch.cuda.set_stream(stream)
    if not single:
        part.set_stream(stream.cuda_stream)
    d_logl = torch.zeros(2, dtype=torch.float64, device="cuda")
    collective_note = None
    if collective == "cabi":
        # the library's own communicator: rank 0 makes the id, the 128 bytes travel through torch.distributed's store.
        # Safety net for the first run on a multi-GPU box: if any rank cannot set it up, ALL ranks fall back to
        # torch.distributed's all-reduce (agreed on by a MIN all-reduce of a flag) and the line says so.
        ok, why = 1, ""
        uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
        try:
            if rank == 0:
                uid.copy_(torch.frombuffer(bytearray(hip.HipPartition.comm_unique_id()), dtype=torch.uint8))
        except Exception as e:
            ok, why = 0, f"comm_unique_id: {e}"
        if multi:
            flag = torch.tensor([ok], dtype=torch.int32, device=ctl)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item())
            if ok:
                dist.broadcast(uid, src=0)
        if ok:
            try:
                part.comm_init_rank(bytes(uid.cpu().numpy().tobytes()), rank, N if multi else 1)
            except Exception as e:
                ok, why = 0, f"comm_init_rank: {e}"
            if multi:
                flag = torch.tensor([ok], dtype=torch.int32, device=ctl)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ok = int(flag.item())
        if not ok:
            collective = "torch" if multi else "none"
            collective_note = "C-ABI communicator not available (" + (why or "another rank failed") + "): fell back to torch.distributed.all_reduce"
            sys.stderr.write(f"[bench] rank {rank}: {collective_note}\n")
    have_host = False
    if multi and want_host:
        # the collective-free combine: a shared-memory segment named after this run (rank 0's pid travels over torch.distributed)
        tag_t = torch.tensor([os.getpid() if rank == 0 else 0], dtype=torch.int64, device=ctl)
        dist.broadcast(tag_t, src=0)
        ok = 1
        try:
            part.comm_init_host(f"bench_{int(tag_t.item())}_{os.environ.get('MASTER_PORT', '0')}", rank, N)
        except Exception as e:
            ok = 0
            sys.stderr.write(f"[bench] rank {rank}: host exchange not available ({e})\n")
        flag = torch.tensor([ok], dtype=torch.int32, device=ctl)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        have_host = bool(int(flag.item()))
        if share and have_host:
            collective = "host"   # (two ranks on one device: no RCCL — the host exchange needs no device collective at all)
        if not have_host and collective == "host":
            collective = "cabi"
            collective_note = "host exchange not available on every rank: the C-ABI's RCCL all-reduce instead"
    coeffs = np.empty((B * n_classes, 2))
    coeffs[:, 0] = np.tile(tb, n_classes)
    class_omega = np.array([0.1, 1.0, 5.0][:n_classes]) / 0.3 if n_classes > 1 else np.array([1.0])
    class_w = np.array([0.7, 0.25, 0.05])
    ar_step = None
    if n_mix:
        # explicit-form mixture: one coefficient row per (branch, component); the second component's omega is what the loop sweeps
        mix_coeffs = np.empty((B, n_mix, 2))
        mix_coeffs[:, :, 0] = tb[:, None]
        mix_coeffs[:, :, 1] = tb[:, None] * np.array(MIX_OMEGAS[:n_mix])[None, :]
        mix_w = np.ascontiguousarray(np.tile(np.array(MIX_WEIGHTS[:n_mix]), (B, 1)))
        mix_step = part.prepare_mixture_built_step(nodes, nodes, mix_w, pi, mix_coeffs)
        sync_step = enqueue = fetch = None
    elif n_classes > 1:
        cat_step = part.prepare_built_categories_step(nodes, nodes, class_w, pi, coeffs)
    else:
        enqueue = part.prepare_device_step(nodes, nodes, pi, d_logl.data_ptr(), coeffs)
        fetch = part.prepare_fetch(d_logl.data_ptr())   # log-L behind the all-reduce -> host (host-mapped record, no D2H copy)
        sync_step = part.prepare_built_step(nodes, nodes, pi, coeffs)   # N == 1: synchronous C-ABI entry point
        ar_step = part.prepare_built_allreduce_step(nodes, nodes, pi, coeffs) if collective == "cabi" else None
        xch_step = part.prepare_built_exchange_step(nodes, nodes, pi, coeffs) if have_host else None

    use_host = [collective == "host" and have_host]   # (a cell: the auto choice below flips it)

    def step(k, sync=True, force_torch=False, force=None):
        if n_mix:
            np.multiply(tb, MIX_OMEGAS[1] + 0.001 * k, out=mix_coeffs[:, 1, 1])   # R2 = 1.0 + 0.001 k on every branch
            v = mix_step()
            if multi:   # (a rank's partial log-L over its patterns)
                d_logl[0] = v
                hdist.allreduce_logl(d_logl[:1])
                v = float(d_logl[0].item())
            return v
        omega = omega0 + 0.001 * k
        if n_classes > 1:
            coeffs[:, 1] = np.repeat(class_omega * omega, B) * coeffs[:, 0]
        else:
            np.multiply(tb, omega, out=coeffs[:, 1])   # nonSynRate = omega * synRate on every branch (one numpy call: < 1 us)
        if n_classes > 1:
            v = cat_step()         # build_q (3 x 125 matrices) + expm + batched pruning + mixing + reduction
            if share:
                t = torch.tensor([v], dtype=torch.float64)
                dist.all_reduce(t)
                return float(t.item())
            if multi:              # (classes are mixed per site on the rank that owns the site; the partial log-Ls add up)
                d_logl[0] = v
                hdist.allreduce_logl(d_logl[:1])
                v = float(d_logl[0].item())
            return v
        if sync and not force_torch and xch_step is not None and (force == "host" or (force is None and use_host[0])):
            return xch_step()      # build_q + evaluate_built_exchange: local pass, partials through shared memory, total on every rank
        if sync and ar_step is not None and not force_torch:
            return ar_step()       # build_q + evaluate_built_allreduce: local pass, in-stream ncclAllReduce, value on every rank
        if (not multi) and sync and not os.environ.get("HYPHY_BENCH_DEVICE_STEP"):
            return sync_step()     # build_q + evaluate_built: log-L returned by the C-ABI call itself (all shards if --single-process)
        enqueue()      # device-side Q for every branch, then expm + pruning + reduction (C-ABI calls)
        if share:      # (diagnostic: the sum over ranks through a host tensor and gloo)
            t = torch.tensor([fetch()], dtype=torch.float64)
            dist.all_reduce(t)
            return float(t.item())
        if multi:
            hdist.allreduce_logl(d_logl[:1])                   # one RCCL all-reduce per evaluation (torch.distributed)
        if sync:
            if os.environ.get("HYPHY_BENCH_READBACK") == "item":
                return float(d_logl[0].item())                 # (torch's device-to-host copy + synchronisation)
            return fetch()                                     # log-L back on the host (synchronises)
        return None

    if n_mix or n_classes > 1:
        xch_step = None
    if not multi and not n_mix and n_classes == 1 and not os.environ.get("HYPHY_BENCH_DEVICE_STEP"):
        # (one GPU, one rate class: the same work as the general step above — new coefficients, build_q + evaluate_built, log-L back —
        #  without its case analysis: what a C host's loop does, and ≈ 1 us of interpreter per step less)
        _general, _col = step, coeffs[:, 1]

        def step(k, sync=True, force_torch=False, force=None):   # noqa: F811
            if not sync or force_torch or force is not None:
                return _general(k, sync=sync, force_torch=force_torch, force=force)
            np.multiply(tb, omega0 + 0.001 * k, out=_col)
            return sync_step()
    if os.environ.get("HYPHY_BENCH_STEP_TIMES") == "all":   # (diagnostic: wall clock of EVERY step of the run, from the first)
        _step0, _marks = step, []

        def step(k, **kw):   # noqa: F811
            t_ = time.perf_counter()
            v_ = _step0(k, **kw)
            _marks.append((time.perf_counter() - t_) * 1e6)
            if len(_marks) in (40, 120):
                sys.stderr.write("[bench] every step so far, us: " + " ".join(f"{x:.0f}" for x in _marks) + "\n")
            return v_
    ll0 = step(0)
    collective_choice = None
    if collective_auto and ar_step is not None and xch_step is not None:
        # which collective for the timed region?  20 untimed steps of each (max over ranks), rank 0's verdict for everybody
        collective_choice = {}
        for name in ("cabi", "host", "cabi", "host"):
            step(1, force=name)
            dist.barrier()
            torch.cuda.synchronize()
            ta = time.perf_counter()
            for k in range(20):
                step(k + 1, force=name)
            dist.barrier()
            torch.cuda.synchronize()
            ms = 1e3 * max_over_ranks(dist, ctl, time.perf_counter() - ta) / 20
            collective_choice[name + "_ms_per_step"] = min(ms, collective_choice.get(name + "_ms_per_step", 1e30))
        pick = torch.tensor([1 if collective_choice["host_ms_per_step"] < collective_choice["cabi_ms_per_step"] else 0], dtype=torch.int32, device=ctl)
        dist.broadcast(pick, src=0)
        use_host[0] = bool(int(pick.item()))
        collective = "host" if use_host[0] else "cabi"
        collective_choice["chosen"] = collective
    if ar_step is not None and not np.isfinite(ll0):
        # (the all-reduced value is the same on every rank, so every rank takes this branch or none does)
        collective_note = "hyphy_hip_evaluate_built_allreduce returned a non-finite value on the first evaluation: fell back to torch.distributed.all_reduce"
        sys.stderr.write(f"[bench] rank {rank}: {collective_note}\n")
        ar_step = None
        collective = "torch" if multi else "none"
        ll0 = step(0)
    # 4 states: the library compiles the schedule into straight-line code on a background thread once it has come back a few times
    # (hyphy_amd/csrc/nucgen.hip) and switches over when the code object is there: reach that steady state before anything is timed
    # (untimed evaluations, at most 30 s; HYPHY_HIP_NUCGEN=0 keeps the interpreter and skips the wait)
    nucgen_wait_s = None
    if D == 4 and os.environ.get("HYPHY_HIP_NUCGEN", "1") != "0" and hasattr(part, "prune_kernel_name"):
        t_ng = time.perf_counter()
        while part.prune_kernel_name() != "nucgen_kernel" and time.perf_counter() - t_ng < 30.0:
            step(1)
        nucgen_wait_s = time.perf_counter() - t_ng
    # device preheat (clock ramp): the same number of steps on every rank (a step contains a collective when N > 1)
    t_pre = time.perf_counter()
    for _ in range(3):
        step(1)
    per = (time.perf_counter() - t_pre) / 3.0
    # Python's cyclic garbage collector is kept out of the timed windows (r06): a generation-2 collection of a process that has torch
    # loaded takes 30-40 ms and fires at an allocation COUNT — two more ctypes prototypes in hyphy_amd/hip.py moved it from the untimed
    # part of this script into the 2 ms timed window of the driver's protocol (value 10 000 -> 490 evals/s, `git log` of this file).
    # Collected once here, then disabled until the measurements are done.
    import gc
    gc.collect()
    gc.disable()
    # value_cold: the driver's exact protocol (W warm-up steps, K timed steps) on a chip that has just idled — no preheat
    value_cold = None
    if args.cold_s > 0:
        time.sleep(args.cold_s)
        for k in range(args.warmup):
            step(k + 1)
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        tc0 = time.perf_counter()
        cold_marks = [] if os.environ.get("HYPHY_BENCH_STEP_TIMES") else None
        for k in range(args.steps):
            step(k + 1)
            if cold_marks is not None:
                cold_marks.append(time.perf_counter())
        if multi:
            dist.barrier()
        t_sync = time.perf_counter()
        torch.cuda.synchronize()
        dtc = time.perf_counter() - tc0
        if cold_marks is not None and rank == 0:
            sys.stderr.write(f"[bench] un-preheated window: torch.cuda.synchronize() behind the last step took {1e6 * (time.perf_counter() - t_sync):.0f} us\n")
        if cold_marks and rank == 0:
            sys.stderr.write("[bench] un-preheated window, per-step us: " + " ".join(f"{x:.0f}" for x in np.diff(np.array([tc0] + cold_marks)) * 1e6) + "\n")
        if multi:
            tcm = torch.tensor([dtc], dtype=torch.float64, device=ctl)
            dist.all_reduce(tcm, op=dist.ReduceOp.MAX)
            dtc = float(tcm.item())
        value_cold = args.steps / dtc
    n_pre = int(min(20000, max(0.0, args.preheat_s) / max(per, 1e-6)))
    if multi:
        cnt = torch.tensor([n_pre], dtype=torch.int64, device=ctl)
        dist.broadcast(cnt, src=0)
        n_pre = int(cnt.item())
    for _ in range(n_pre):
        step(1)
    n_pre += 3
    for k in range(args.warmup):
        step(k + 1)
    t_exp = t_prune = t_red = 0.0
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    timed_values = [0.0] * args.steps
    step_marks = [0.0] * args.steps if os.environ.get("HYPHY_BENCH_STEP_TIMES") else None   # (diagnostic: per-step wall clock)
    for k in range(args.steps):
        last = timed_values[k] = step(k + 1)
        if step_marks is not None:
            step_marks[k] = time.perf_counter()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gc.enable()
    if step_marks is not None and rank == 0:
        marks = np.diff(np.array([t0] + step_marks)) * 1e6
        sys.stderr.write("[bench] per-step us: " + " ".join(f"{x:.0f}" for x in marks) + "\n")
    # kernel durations of the timed steps: HIP event pairs recorded by the library on ITS stream around the
    # pruning launches of every evaluation, read back only now (querying inside the loop perturbs it)
    pt = part.prune_timings(min(max(1, args.steps // TIMING_EVERY), 1024))
    t_prune = float(pt.sum()) * (args.steps / max(1, len(pt)))
    t_exp = t_red = None
    if n_classes == 1 and not n_mix:
        # expm (incl. the fused rate-matrix build) and reduction kernels: event-timed on a few extra steps AFTER the
        # timed region (two more event records per step would perturb it)
        part.set_all_timings(True)
        te, tr_, ta = [], [], []
        for k in range(8):
            step(args.steps + 1 + k)
            tm = part.last_timings()
            te.append(tm[0])
            tr_.append(tm[2])
            ta.append(part.last_allreduce_ms())
        part.set_all_timings(False)
        t_exp, t_red = float(np.median(te)), float(np.median(tr_))
        t_ar = float(np.median(ta)) if collective == "cabi" else None
    else:
        t_ar = None
    per_rank = None
    collective_ab = None
    if multi:
        dt = max_over_ranks(dist, ctl, dt)
        per_rank = gather_per_rank(dist, ctl, N, hi - lo, t_prune / max(1, args.steps), t_exp, t_red, t_ar)
        if ar_step is not None and n_classes == 1 and not share:
            # both collectives in ONE run (which of the two costs less per step on this node?): 8 + 8 more steps
            # behind the timed region, the library's in-stream all-reduce and torch.distributed's on the same stream
            collective_ab = {}
            for name, ft in (("cabi", False), ("torch", True), ("cabi_again", False)) + ((("host", False),) if xch_step is not None else ()):
                fc = "host" if name == "host" else ("cabi" if not ft else None)
                step(1, force_torch=ft, force=fc)
                dist.barrier()
                torch.cuda.synchronize()
                ta = time.perf_counter()
                for k in range(8):
                    step(k + 1, force_torch=ft, force=fc)
                dist.barrier()
                torch.cuda.synchronize()
                collective_ab[name + "_ms_per_step"] = 1e3 * max_over_ranks(dist, ctl, time.perf_counter() - ta) / 8

    branch_cache = None
    if args.branch_cache and n_classes == 1 and not n_mix and N == 1 and D > 4 and collective == "none":
        # one-branch line search (the optimiser's inner loop): all parameters fixed, ONE branch
        # length varies; each evaluation = 1 expm + 1 [D x D] x [D x S] contraction + reduction
        node = L + I // 2                                  # an internal branch in the middle of the tree
        omega = omega0
        Qb = np.zeros((B, D, D))
        for b in range(B):
            Qb[b] = tb[b] * (T[0] + omega * T[1])
            np.fill_diagonal(Qb[b], 0.0)
            np.fill_diagonal(Qb[b], -Qb[b].sum(1))
        full = part.evaluate(nodes, nodes, Qb, pi)
        tb0 = time.perf_counter()
        part.branch_cache_build(node)
        part.synchronize()
        build_ms = 1e3 * (time.perf_counter() - tb0)
        qn = np.ascontiguousarray(Qb[node])
        bstep = part.prepare_branch_cache_step(node, qn)
        same = bstep()
        for k in range(args.warmup):
            bstep()
        tb1 = time.perf_counter()
        for k in range(args.steps):
            qn[:] = Qb[node] * (1.0 + 0.002 * (k + 1))
            lastb = bstep()
        dtb = time.perf_counter() - tb1
        branch_cache = {"node": int(node), "evals_per_s": args.steps / dtb, "ms_per_eval": 1e3 * dtb / args.steps,
                        "build_ms": build_ms, "logl_full": full, "logl_cached_same_length": same,
                        "rel_err_vs_full": abs(same - full) / abs(full), "logl_last": lastb}

    site_fits = None
    if args.site_fits > 0 and n_classes == 1 and not n_mix and N == 1 and D > 4:
        site_fits = time_site_fits(part, args, wl, pd_all, flat, T, pi, tb)

    pipelined = None
    if args.pipelined and n_classes == 1 and not n_mix:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(args.steps):
            step(k + 1, sync=False)
        torch.cuda.synchronize()
        pipelined = args.steps / (time.perf_counter() - t1)

    if rank == 0:
        S_rank = hi - lo
        flops, bytes_ = alg_work(D, S_rank, L, I)
        flops *= n_classes
        bytes_ *= n_classes
        prune_ms = max(t_prune / args.steps, 1e-9)
        bound = "mfma" if D > 4 else "hbm"
        # subtree repeats (repeats.hip): a class-compressed pass EXECUTES fewer edge products than the algorithmic count (every internal
        # edge at every pattern).  `achieved` / `frac` are on the executed flops — what the matrix pipes did; the algorithmic flops
        # over the same time are `effective_tflops` (what the caller got; may exceed what any pipe could do without the reuse).
        rs = part.repeat_stats()
        rep_on = bool(rs["available"] and rs["in_use"])
        exec_flops = flops
        if rep_on and D > 4:
            exec_flops = n_classes * (rs["edge_products_on"] * (2 * D * D + 2 * D) + S_rank * (L * D + 2 * D))
        if bound == "mfma":
            ach = exec_flops / (prune_ms * 1e-3) / 1e12
            roof = dict(bound="mfma", achieved=ach, peak=FP64_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                        frac=ach / FP64_MFMA_PEAK_TFLOPS, traffic=None)
            roof["executed_flops_per_step"] = exec_flops
            roof["repeat_ratio"] = rs["edge_products_on"] / max(1, rs["edge_products_off"]) if rep_on else 1.0
            roof["effective_tflops"] = flops / (prune_ms * 1e-3) / 1e12
            roof["subtree_repeats"] = dict(rs, note="first shard; edge products per full pass with / without class compression (tables padded to "
                                                    "tiles of 16 classes, a path of k nodes walked once per 16 classes of its top node)")
        else:
            # 4 states.  SURVEY 8d's algorithmic bytes assume every internal conditional vector goes through HBM once each
            # way; the kernel keeps them on chip (registers / LDS parking, lazy persistence), so that model is not what the
            # memory system sees.  `achieved` / `frac` stay the contract's algorithmic figure only while it is meaningful
            # (<= the peak); the numbers to hold the kernel against are `traffic_rate_gbs` (counter bytes / time, vs the
            # ~6.3 TB/s a streaming kernel reaches) and `valu_tflops` (algorithmic flops / time, vs the 78.6 TFLOP/s FP64
            # vector peak): the kernel is bound by instruction issue, between the two roofs.
            ach = bytes_ / (prune_ms * 1e-3) / 1e9
            roof = dict(bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=(ach / HBM_PEAK_GBS if ach <= HBM_PEAK_GBS else None),
                        traffic=None)
            roof["achievable_gbs"] = HBM_ACHIEVABLE_GBS
            roof["valu_tflops"] = flops / (prune_ms * 1e-3) / 1e12
            roof["valu_frac"] = roof["valu_tflops"] / FP64_MFMA_PEAK_TFLOPS
            if roof["frac"] is None:
                roof["note"] = ("algorithmic bytes / time exceeds the HBM peak: the modelled traffic (every conditional vector through "
                                "HBM) does not exist — see traffic_rate_gbs (PMC) and valu_tflops")
        roof["kernel"] = part.prune_kernel_name()
        kernels = [roof["kernel"]]
        if rep_on and D > 4:   # two launches per pass: the class tables (lower phase), then the trunk through the pruning kernel
            # (r06: a workgroup of row-split waves walks a path — class_table_team_kernel; HYPHY_HIP_REP_TEAM=0: the one-wave walk)
            lower = "class_table_team_kernel" if os.environ.get("HYPHY_HIP_REP_TEAM", "1") != "0" else "class_table_kernel"
            kernels = [lower, roof["kernel"]]
            roof["kernel"] = lower + " + " + roof["kernel"]
        if bound == "mfma":
            # measured ceiling of the instruction the kernel issues (tools/ubench/mfma4_skew.hip): v_mfma_f64_16x16x4_f64 with VGPR
            # accumulators sustains 73-78 TFLOP/s chip-wide from one operand pair and 66-72 with the kernel's own operand stream
            # (profiles/r04_ubench_*.txt); `peak` stays the datasheet figure
            roof["instruction"] = "v_mfma_f64_16x16x4_f64"
            ipk, ipk_src = measured_instruction_peak()
            if ipk is not None:
                roof["instruction_peak_measured"] = ipk   # TFLOP/s, one operand pair, VGPR accumulators
                roof["instruction_peak_source"] = ipk_src
        # forest scheduling: the pruning pass of ONE evaluation is `launches_per_step` launches of the same
        # kernel (levels of subtree fragments).  achieved = (algorithmic work of the pass / launches) / (mean
        # launch duration) = work of the pass / time of the pass; rocprofv3's per-launch average x launches
        # per step must agree with kernel_ms_per_step.
        nl = part.prune_launches() + (1 if rep_on and D > 4 else 0)
        roof["launches_per_step"] = nl
        roof["kernel_ms_per_launch"] = prune_ms / nl
        roof["kernel_ms"] = prune_ms
        roof["timed_steps_sampled"] = f"1 in {TIMING_EVERY}"
        roof["expm_ms"] = t_exp       # median of 8 event-timed steps after the timed region; None: not measured
        roof["reduce_ms"] = t_red
        if t_ar is not None and not multi:
            roof["allreduce_ms"] = t_ar   # the in-stream ncclAllReduce of one double (same 8 steps)
        roof["alg_flops_per_step"] = flops
        roof["alg_bytes_per_step"] = bytes_
        # HBM traffic per launch: measured now (two rocprofv3 counter passes of this same script), else the value of the
        # round's committed profile of the same workload and kernel, else null
        if N == 1 and not args.no_traffic and not args.no_cpu_baseline:
            tr_ = measure_traffic(args.workload, kernels)
            if tr_:
                roof["traffic"] = tr_["bytes_per_launch"]
                roof["traffic_detail"] = tr_
        if roof.get("traffic") is None:
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(pmc):
                try:
                    rec = json.load(open(pmc)).get(args.workload, {})
                    if rec.get("kernel") == roof["kernel"]:
                        roof["traffic"] = rec.get("hbm_bytes_per_launch")
                        roof["traffic_detail"] = {"source": "profiles/pmc_traffic.json (committed PMC run of this workload)"}
                except Exception:
                    pass
        if roof.get("traffic"):
            # counter bytes per launch / launch duration: what the memory system really moved (for the 4-state kernel this,
            # not the algorithmic figure, is the number to hold against the ~6.3 TB/s achievable HBM rate)
            # (a class-compressed pass: `traffic` is the sum over its two launches, so the time is the pass's, not a launch's)
            t_ms = roof["kernel_ms"] if rep_on and D > 4 else roof["kernel_ms_per_launch"]
            roof["traffic_rate_gbs"] = roof["traffic"] / (t_ms * 1e-3) / 1e9
            if bound == "hbm":
                roof["traffic_frac_of_achievable"] = roof["traffic_rate_gbs"] / HBM_ACHIEVABLE_GBS
        out = {
            "metric": "full-tree log-L evals/sec, 61-state MG94 codon, 64 taxa x 10k codons" if args.workload == "mg94_64x10k"
                      else f"full-tree log-L evals/sec ({args.workload})",
            "value": args.steps / dt, "unit": "evals/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic", "preheat_s": args.preheat_s, "preheat_steps": n_pre,
            "value_cold": value_cold,   # the same K steps behind the same W warm-up steps on an idle chip (no preheat); None: skipped
            "config": {"workload": args.workload, "states": D, "taxa": L, "codons" if D > 4 else "sites": wl["sites"],
                       "unique_patterns": int(S_all), "branches": int(B), "rate_classes": n_classes,
                       "parallelism": (f"site-shard x{N}, one process per GPU" if multi else
                                       f"site-shard x{N}, one process, shard partials combined by {args.combine}" if single and N > 1 else "single GPU"),
                       "collective": COLLECTIVE_LABEL[collective],
                       **({"collective_note": collective_note} if collective_note else {}),
                       **({"DIAGNOSTIC": "HYPHY_BENCH_SHARE_DEVICE: all ranks ran on ONE device over gloo — the N > 1 code path was walked, the rate means nothing"} if share else {}),
                       "patterns_rank0": int(S_rank),
                       **({"mixture_components": n_mix, "mixture_weights": list(MIX_WEIGHTS[:n_mix]), "mixture_omegas": list(MIX_OMEGAS[:n_mix])} if n_mix else {}),
                       "step": ("device build + expm of every component of every branch + mixing into the branches' matrices + full pruning pass + reduction" if n_mix else
                                "device Q build + expm of all branches + full pruning pass + reduction") +
                               (" + host-side exchange of the ranks' partials (shared memory)" if collective == "host" else
                                (" + RCCL all-reduce" if (multi or collective == "cabi") else "")) + ", log-L returned to host every step"},
            "logl_first": ll0, "logl_last": last,
            "roofline": roof,
            **({"branch_cache": branch_cache} if branch_cache else {}),
            **({"site_fits": site_fits} if site_fits else {}),
        }
        if multi:   # the N > 1 additions: which collective, every rank's numbers, the two collectives side by side
            cfg_m, roof_m, top_m = multi_gpu_fields(N, collective, collective_note, per_rank, t_ar, collective_ab)
            out["config"].update(cfg_m)
            roof.update(roof_m)
            out.update(top_m)
            if collective_choice:   # (auto: both collectives timed on untimed steps, the faster one ran the timed region)
                out["collective_choice"] = collective_choice
        if pipelined:
            out["value_pipelined_no_host_sync"] = pipelined
        if not args.no_cpu_baseline and N == 1 and n_mix:
            try:
                cb, ref = cpu_baseline_mixture(wl, syn, t_branch, args.steps)
            except Exception as e:
                sys.stderr.write(f"[bench] reference baseline unavailable: {e}\n")
                cb = ref = None
            if cb is not None:
                out["cpu_baseline"] = cb
                par = {"logl_gpu": ll0, "logl_cpu": ref["logl"], "rel_err": abs(ll0 - ref["logl"]) / abs(ref["logl"]), "tolerance": 1e-6}
                sv = ref.get("sweep_values")
                if sv is not None and len(sv):
                    n = min(len(sv), args.steps)
                    rel = np.abs(np.array(timed_values[:n]) - sv[:n]) / np.abs(sv[:n])
                    par["timed_points_checked"] = int(n)
                    par["timed_points_max_rel_err"] = float(rel.max())
                out["parity"] = par
        elif not args.no_cpu_baseline and N == 1 and n_classes == 1:
            cb = ref = sweep = None
            try:
                from oracle import hbl
                nthr = args.cpu_threads
                if not nthr and hbl.have_reference() and wl["unit"] == 3:
                    ncpu = os.cpu_count() or 1
                    cands = sorted({c for c in (8, 16, 32, 64) if 1 < c <= ncpu}) or [ncpu]   # (all 256 hardware threads: 4 evals/s)
                    sweep = cpu_thread_sweep(wl, syn, omega0, t_branch, cands) if cands else {}
                    nthr = max(sweep, key=sweep.get) if sweep else 1
                cb, ref = cpu_baseline(wl, syn, omega0, t_branch, nthr or 1, args.steps)
                if cb is not None and sweep:
                    cb["thread_sweep"] = {str(k): v for k, v in sweep.items()}   # evals/s of short runs at each count (this box)
            except Exception as e:  # reference binary missing / failed: fall back to the C restatement
                sys.stderr.write(f"[bench] reference baseline unavailable: {e}\n")
            if cb is None:
                Q0 = (models.mg94rev_Q_batch(tb, omega0, REV, POS_FREQS) if D > 4 else
                      np.stack([models.nuc_rev_Q(t_branch, models.hky85_rev(0.35), NUC_FREQS)] * B))
                cb, ref_ll = cpu_port_baseline(pd_all, flat, Q0, pi, D > 4)
                ref = dict(logl=ref_ll)
            out["cpu_baseline"] = cb
            par = {"logl_gpu": ll0, "logl_cpu": ref["logl"], "rel_err": abs(ll0 - ref["logl"]) / abs(ref["logl"]),
                   "tolerance": 1e-6}
            sv = ref.get("sweep_values")
            if sv is not None and len(sv):   # every timed parameter point the reference also evaluated
                n = min(len(sv), args.steps)
                rel = np.abs(np.array(timed_values[:n]) - sv[:n]) / np.abs(sv[:n])
                par["timed_points_checked"] = int(n)
                par["timed_points_max_rel_err"] = float(rel.max())
            if ref.get("site_logl") is not None and D > 4:
                # per-site log-likelihoods at the first point: GPU per-pattern (l_s, c_s) -> log L_s, mapped to sites
                Q0 = models.mg94rev_Q_batch(tb, omega0, REV, POS_FREQS)
                _, sl, sc = part.evaluate(nodes, nodes, Q0, pi, per_site=True)
                gpu_site = (np.log(sl) - 64.0 * np.log(2.0) * sc)[pd_all.site_to_pattern]
                par["per_site_max_abs_dlogl"] = float(np.max(np.abs(gpu_site - ref["site_logl"])))
                par["per_site_sites"] = int(len(gpu_site))
            out["parity"] = par
        # (RCCL prints its version banner through C stdio: flush that first, so that the JSON line is the LAST line on stdout)
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    part.close()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
