"""TEST INFRASTRUCTURE — drives the *real* reference binary (``oracle/_ref/hyphy``, built by
``oracle/Makefile.ref`` from the sources under /root/reference) through self-contained HBL
scripts written here, to (a) generate the golden vectors committed under ``tests/golden/``
and (b) time the reference's CPU path as ``bench.py``'s ``cpu_baseline`` (kind "reference").

Nothing in the product path imports this module.  The scripts are our own HBL (syntax
precedents: ``tests/hbltests/SimpleOptimizations/SmallCodon.bf`` for the model/LF block,
``res/TemplateBatchFiles/libv3/tasks/estimators.bf:1392-1394`` for the
LF_START_COMPUTE / LFCompute / LF_DONE_COMPUTE bracket, ``tests/hbltests/HMM/TreeHMM.bf:17``
for a discrete ``category`` variable, SURVEY A.8 for the thread-pinning trick).
They need no ``res/`` library, so the binary also runs on the GPU box where
/root/reference does not exist.
"""
from __future__ import annotations

import os
import re
import subprocess
import tempfile
from typing import Dict, List, Optional, Sequence

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_BIN = os.path.join(HERE, "_ref", "hyphy")


def have_reference() -> bool:
    return os.path.isfile(REF_BIN) and os.access(REF_BIN, os.X_OK)


def _fmt(x: float) -> str:
    return repr(float(x))


def write_fasta(path: str, names: Sequence[str], seqs: Sequence[str]) -> None:
    with open(path, "w") as fh:
        for n, s in zip(names, seqs):
            fh.write(f">{n}\n{s}\n")


def codon_model_block(template, codon_freqs, rate_expr: str = "t", omega: str = "R") -> str:
    """``MGQ[i][j] := AC*R*t*const`` lines + Model statement (multiply-by-freqs = 0, the
    constants carry the position-specific target-nucleotide frequency, as in SmallCodon.bf)."""
    lines = ["MGQ = {61,61};"]
    for (i, j, name, ns, pf) in template:
        parts = []
        if name != "AG":
            parts.append(name)
        if ns:
            parts.append(omega)
        parts.append(rate_expr)
        parts.append(_fmt(pf))
        lines.append(f"MGQ[{i}][{j}] := {'*'.join(parts)};")
    fr = ",\n".join("{" + _fmt(v) + "}" for v in codon_freqs)
    lines.append("vectorOfFrequencies = {\n" + fr + "};")
    lines.append("Model MGM = (MGQ, vectorOfFrequencies, 0);")
    return "\n".join(lines)


def codon_mixture_model_block(template, codon_freqs, omegas: Sequence[str], weights: Sequence[str], rate_expr: str = "t") -> str:
    """Branch-site mixture in the reference's "explicit form" (syntax precedents:
    ``tests/hbltests/RegressionTesting/expModelCrash.bf:1159``, ``res/TemplateBatchFiles/BranchSiteREL.bf:276``): one rate
    matrix per omega class, ``Model = ("Exp(Q1)*w1+Exp(Q2)*w2...", freqs, EXPLICIT_FORM_MATRIX_EXPONENTIAL)`` — the
    transition matrix of every branch is the weighted sum of the classes' exponentials (``tree.cpp:3047-3090``)."""
    lines = []
    for m, om in enumerate(omegas, start=1):
        lines.append(f"MGQ{m} = {{61,61}};")
        for (i, j, name, ns, pf) in template:
            parts = []
            if name != "AG":
                parts.append(name)
            if ns:
                parts.append(om)
            parts.append(rate_expr)
            parts.append(_fmt(pf))
            lines.append(f"MGQ{m}[{i}][{j}] := {'*'.join(parts)};")
    fr = ",\n".join("{" + _fmt(v) + "}" for v in codon_freqs)
    lines.append("vectorOfFrequencies = {\n" + fr + "};")
    expr = "+".join(f"Exp(MGQ{m})*({w})" for m, w in enumerate(weights, start=1))
    lines.append(f'Model MGM = ("{expr}", vectorOfFrequencies, EXPLICIT_FORM_MATRIX_EXPONENTIAL);')
    return "\n".join(lines)


def nuc_model_block(freqs, rate_expr: str = "t") -> str:
    from hyphy_amd.models import REV_NAMES
    lines = ["NQ = {4,4};"]
    for i in range(4):
        for j in range(4):
            if i == j:
                continue
            name = REV_NAMES[(min(i, j), max(i, j))]
            e = rate_expr if name == "AG" else f"{name}*{rate_expr}"
            lines.append(f"NQ[{i}][{j}] := {e};")
    fr = ",".join("{" + _fmt(v) + "}" for v in freqs)
    lines.append("nucFreqs = {" + fr + "};")
    lines.append("Model NM = (NQ, nucFreqs, 1);")
    return "\n".join(lines)


def build_script(*, fasta: str, newick: str, unit: int, model_block: str, model_name: str,
                 globals_: Dict[str, float], branch_t: Dict[str, float],
                 out_path: str, sweep: Optional[Dict] = None, threads: int = 0,
                 category: Optional[Dict] = None, per_site: bool = True, optimize: bool = False,
                 constraints: Optional[Dict[str, str]] = None, extra_partitions: Optional[List[Dict]] = None,
                 upper_bounds: Optional[Dict[str, float]] = None) -> str:
    """One self-contained batch file.  ``sweep`` = {"param": "R", "start": .3, "step": .001,
    "n": N} runs the SURVEY A.8 timing loop and reports wall-clock seconds via Time(1)."""
    L: List[str] = ["VERBOSITY_LEVEL = -1;", "PRINT_DIGITS = 17;"]
    for k, v in globals_.items():
        if constraints and k in constraints:
            L.append(f"global {k} := {constraints[k]};")   # (tied parameter, e.g. CG := AT)
        else:
            L.append(f"global {k} = {_fmt(v)};")
        if upper_bounds and k in upper_bounds:
            L.append(f"{k} :< {_fmt(upper_bounds[k])};")   # (e.g. mixture weights live in [0, 1])
    if category:
        w = ",".join(_fmt(x) for x in category["weights"])
        v = ",".join(_fmt(x) for x in category["values"])
        L.append(f"category {category['name']} = ({len(category['weights'])}, {{{{{w}}}}}, MEAN, , {{{{{v}}}}}, 0, 1e25);")
    L.append(model_block)
    L.append(f"UseModel ({model_name});")
    L.append(f"Tree givenTree = {newick};")
    L.append(f'DataSet ds = ReadDataFile ("{fasta}");')
    if unit == 3:
        L.append('DataSetFilter filteredData = CreateFilter (ds,3,"","","TAA,TAG,TGA");')
    else:
        L.append("DataSetFilter filteredData = CreateFilter (ds,1);")
    for name, t in branch_t.items():
        L.append(f"givenTree.{name}.t = {_fmt(t)};")
    pairs = ["filteredData, givenTree"]
    # further partitions of the same likelihood function, each with its own alignment and tree (syntax precedent:
    # res/TemplateBatchFiles/REL/MultiplePartitions.bf builds `LikelihoodFunction lf = (filter_1, tree_1, filter_2, tree_2, ...)`)
    for k, part in enumerate(extra_partitions or [], start=1):
        L.append(f"Tree givenTree{k} = {part['newick']};")
        L.append(f'DataSet ds{k} = ReadDataFile ("{part["fasta"]}");')
        if unit == 3:
            L.append(f'DataSetFilter filteredData{k} = CreateFilter (ds{k},3,"","","TAA,TAG,TGA");')
        else:
            L.append(f"DataSetFilter filteredData{k} = CreateFilter (ds{k},1);")
        for name, t in part["branch_t"].items():
            L.append(f"givenTree{k}.{name}.t = {_fmt(t)};")
        pairs.append(f"filteredData{k}, givenTree{k}")
    L.append(f"LikelihoodFunction lf = ({', '.join(pairs)});")
    if threads and threads > 1:
        # SURVEY A.8 / BASELINE.md §3: a fresh LF is single-threaded until Optimize runs
        # BenchmarkThreads (likefunc.cpp:223-227); do a throw-away 1-iteration Optimize.
        L.append(f"NUMBER_THREADS = {threads}; MAXIMUM_ITERATIONS_PER_VARIABLE = 1; OPTIMIZATION_TIME_HARD_LIMIT = 1;")
        L.append("Optimize (mles_, lf);")
        for k, v in globals_.items():
            L.append(f"{k} = {_fmt(v)};")
        for name, t in branch_t.items():
            L.append(f"givenTree.{name}.t = {_fmt(t)};")
        for k, part in enumerate(extra_partitions or [], start=1):
            for name, t in part["branch_t"].items():
                L.append(f"givenTree{k}.{name}.t = {_fmt(t)};")
    L.append("LFCompute (lf, LF_START_COMPUTE);")
    L.append("LFCompute (lf, res0);")
    L.append(f'fprintf ("{out_path}", CLEAR_FILE, "LOGL ", Format (res0, 30, 17), "\\n");')
    if sweep:
        p = sweep["param"]
        rec = int(sweep.get("record", 0))   # keep the log-likelihoods of the first `record` points (parity at every timed point)
        if rec > 0:
            L.append(f"swv_ = {{{rec},1}};")
        keep = f" if (k_ < {rec}) {{ swv_[k_] = res_; }}" if rec > 0 else ""
        L.append("t0_ = Time (1);")
        L.append(f"for (k_ = 0; k_ < {int(sweep['n'])}; k_ += 1) {{ {p} = {_fmt(sweep['start'])} + {_fmt(sweep['step'])}*(k_+1); LFCompute (lf, res_);{keep} }}")
        L.append("t1_ = Time (1);")
        if rec > 0:
            L.append(f'fprintf ("{out_path}", "SWEEP_VALUES ", {min(rec, int(sweep["n"]))}, "\\n");')
            L.append(f'for (k_ = 0; k_ < {min(rec, int(sweep["n"]))}; k_ += 1) {{ fprintf ("{out_path}", Format (swv_[k_], 30, 17), "\\n"); }}')
        L.append(f'fprintf ("{out_path}", "SWEEP_SECONDS ", Format (t1_-t0_, 20, 6), "\\n", "SWEEP_LAST ", Format (res_, 30, 17), "\\n");')
        L.append(f"{p} = {_fmt(globals_[p])};")
    L.append("LFCompute (lf, LF_DONE_COMPUTE);")
    if optimize:   # full maximum-likelihood fit (Optimize = the reference's "train()" analogue)
        L.append("OPTIMIZATION_PRECISION = 0.001; VERBOSITY_LEVEL = -1;")
        L.append("Optimize (mles2_, lf);")
        L.append(f'fprintf ("{out_path}", "OPT_LOGL ", Format (mles2_[1][0], 30, 17), "\\n");')
    if per_site:
        L.append("ConstructCategoryMatrix (sl_, lf, SITE_LOG_LIKELIHOODS);")
        L.append(f'fprintf ("{out_path}", "SITES ", Columns (sl_), "\\n");')
        L.append(f'for (k_ = 0; k_ < Columns (sl_); k_ += 1) {{ fprintf ("{out_path}", Format (sl_[k_], 30, 17), "\\n"); }}')
    return "\n".join(L) + "\n"


def run_script(script_text: str, workdir: str, cpus: int = 1, timeout: float = 3600.0, binary: Optional[str] = None,
               extra_env: Optional[Dict[str, str]] = None) -> str:
    bf = os.path.join(workdir, "driver.bf")
    with open(bf, "w") as fh:
        fh.write(script_text)
    lib = os.path.join(workdir, "emptylib")
    os.makedirs(lib, exist_ok=True)
    env = dict(os.environ)
    env.pop("OMP_PROC_BIND", None)
    if extra_env:
        env.update(extra_env)
    r = subprocess.run([binary or REF_BIN, f"LIBPATH={lib}", f"CPU={cpus}", bf], cwd=workdir, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)
    if r.returncode != 0:
        raise RuntimeError(f"reference hyphy failed ({r.returncode}):\n{r.stdout[-4000:]}")
    return r.stdout


def parse_output(path: str) -> Dict:
    out: Dict = {}
    with open(path) as fh:
        lines = fh.read().split("\n")
    i = 0
    while i < len(lines):
        ln = lines[i].strip()
        if ln.startswith("LOGL "):
            out["logl"] = float(ln.split()[1])
        elif ln.startswith("OPT_LOGL "):
            out["opt_logl"] = float(ln.split()[1])
        elif ln.startswith("SWEEP_SECONDS "):
            out["sweep_seconds"] = float(ln.split()[1])
        elif ln.startswith("SWEEP_LAST "):
            out["sweep_last"] = float(ln.split()[1])
        elif ln.startswith("SWEEP_VALUES "):
            n = int(ln.split()[1])
            out["sweep_values"] = np.array([float(x) for x in lines[i + 1:i + 1 + n]])
            i += n
        elif ln.startswith("SITES "):
            n = int(ln.split()[1])
            out["site_logl"] = np.array([float(x) for x in lines[i + 1:i + 1 + n]])
            i += n
        i += 1
    return out


def evaluate(*, names, seqs, newick, unit, model_block, model_name, globals_, branch_t,
             sweep=None, threads=0, category=None, per_site=True, workdir=None, timeout=3600.0,
             binary=None, extra_env=None, optimize=False, constraints=None, extra_partitions=None, upper_bounds=None) -> Dict:
    """Write fasta + script into a scratch dir, run the reference, parse the results."""
    own = workdir is None
    tmp = tempfile.mkdtemp(prefix="hyref_") if own else workdir
    fasta = os.path.join(tmp, "aln.fasta")
    outp = os.path.join(tmp, "out.txt")
    write_fasta(fasta, names, seqs)
    xparts = []
    for k, part in enumerate(extra_partitions or [], start=1):   # dicts with names, seqs, newick, branch_t
        fk = os.path.join(tmp, f"aln{k}.fasta")
        write_fasta(fk, part["names"], part["seqs"])
        xparts.append(dict(fasta=fk, newick=part["newick"], branch_t=part["branch_t"]))
    txt = build_script(fasta=fasta, newick=newick, unit=unit, model_block=model_block, extra_partitions=xparts, upper_bounds=upper_bounds,
                       model_name=model_name, globals_=globals_, branch_t=branch_t,
                       out_path=outp, sweep=sweep, threads=threads, category=category,
                       per_site=per_site, optimize=optimize, constraints=constraints)
    stdout = run_script(txt, tmp, cpus=max(1, threads), timeout=timeout, binary=binary, extra_env=extra_env)
    res = parse_output(outp)
    res["stdout"] = stdout
    return res


def expm_via_reference(Q: np.ndarray, workdir: Optional[str] = None) -> np.ndarray:
    """P = Exp(Q) through the reference's ``_Matrix::Exponentiate`` (HBL ``Exp``)."""
    D = Q.shape[0]
    tmp = tempfile.mkdtemp(prefix="hyexp_") if workdir is None else workdir
    outp = os.path.join(tmp, "p.txt")
    rows = ",\n".join("{" + ",".join(_fmt(v) for v in row) + "}" for row in Q)
    txt = (f"Q_ = {{\n{rows}}};\nP_ = Exp (Q_);\n"
           f'fprintf ("{outp}", CLEAR_FILE);\n'
           f'for (i_ = 0; i_ < {D}; i_ += 1) {{ for (j_ = 0; j_ < {D}; j_ += 1) {{ fprintf ("{outp}", Format (P_[i_][j_], 30, 20), "\\n"); }} }}\n')
    run_script(txt, tmp)
    vals = np.loadtxt(outp)
    return vals.reshape(D, D)
