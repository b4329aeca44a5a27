"""Through-the-host rates: the patched hyphy binary (integration/_build/hyphy_hip, HYPHY_HIP=1) runs HBL written
by oracle/hbl.py, next to the unmodified reference —
  headline  64 taxa x 10 000 codons, MG94 with one local parameter per branch (t) and a global omega: LFCompute sweep of R
            (template mode), and the same with HYPHY_HIP_TEMPLATES=0 (dense mode B);
  class2    the same alignment, MG94 with LOCAL synRate / nonSynRate and `nonSynRate := R*synRate` on the background branches,
            `:= R2*synRate` on every third (foreground) branch — two branch classes (r04: template mode with a group per class);
  cat3      64 taxa x 10 000 codons, MG94 with a 3-class omega category variable (BUSTED-shaped, weighted-sum category mode):
            LFCompute sweep of the global scaling R;
  mix3      the same alignment under the reference's EXPLICIT-FORM 3-component branch-site mixture
            ("Exp(Q1)*W1+Exp(Q2)*W2+Exp(Q3)*(1-W1-W2)"): LFCompute sweep of W1 with device exponentials forced (mode B, mixture mode);
  manylf    N single-codon likelihood functions on the 64-taxon tree (what FEL does per site): create, 50 LFCompute calls with R
            swept, destroy — through the device (with and without the schedule tuner's cache), with host exponentials, and on the CPU.
One JSON line per measurement.  Usage (GPU box): python tools/adapter_rate.py [headline,class2,cat3,mix3,manylf] [n_evals] [n_lfs]"""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from hyphy_amd import data, models, tree as htree
from oracle import hbl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP_BIN = os.path.join(ROOT, "integration", "_build", "hyphy_hip")
which = (sys.argv[1] if len(sys.argv) > 1 else "headline,class2,cat3,mix3,manylf").split(",")
n_evals = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
n_lfs = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
ENV = dict(HYPHY_HIP="1", HYPHY_HIP_VERBOSE="1", **{k: v for k, v in os.environ.items() if k.startswith("HYPHY_HIP_")})

wl = bench.WORKLOADS["mg94_64x10k"]
syn = data.evolve(wl["taxa"], wl["sites"], 3, seed=wl["seed"], p_change=0.04)
tmpl = models.mg94rev_template(bench.POS_FREQS)
pi = models.f3x4_codon_freqs(bench.POS_FREQS)
bt = {nm: 0.05 for nm in syn.flat.branch_names()}
common = dict(names=syn.flat.leaf_names, seqs=syn.seqs, newick=htree.to_newick(syn.tree), unit=3, model_name="MGM", branch_t=bt,
              per_site=False, timeout=1800.0)


def emit(tag, host, res, count, t0, extra=None):
    secs = max(res.get("sweep_seconds", 0.0), 1.0)   # (HBL's Time(1) has 1 s resolution)
    mode = [ln for ln in res.get("stdout", "").split("\n") if "mode:" in ln or "template analysis" in ln or "explicit-form" in ln or "template mode" in ln]
    print(json.dumps({"case": tag, "host": host, "evals": count, "sweep_seconds": secs, "evals_per_s": count / secs, "logl": res["logl"],
                      "wall": time.time() - t0, "adapter_says": mode[-3:], **(extra or {}),
                      **({"tuner_says": [ln[:400] for ln in res.get("stdout", "").split("\n") if "schedule tuner" in ln][:3]} if os.environ.get("ADAPTER_RATE_TUNER") else {})}),
          flush=True)


def measure(tag, host, run, n_pilot, target_s=10.0, min_evals=0, extra=None):
    """evaluations per second as the DIFFERENCE of the wall clock of two runs of the same script with different sweep lengths
    (process start, data reading, likelihood-function set-up and the adapter's learning calls cancel): a pilot of `n_pilot`
    evaluations sizes the long run for >= target_s seconds and >= min_evals evaluations of difference.  (r04 divided by HBL's Time(1),
    whole seconds: most rows were one or two ticks.)"""
    t0 = time.time()
    res0 = run(n_pilot)
    w0 = time.time() - t0
    secs0 = max(res0.get("sweep_seconds", 0.0), 1.0)
    n_long = n_pilot + int(max(min_evals, (n_pilot / secs0) * target_s))
    t1 = time.time()
    res = run(n_long)
    w1 = time.time() - t1
    dt = max(w1 - w0, 1e-3)
    mode = [ln for ln in res.get("stdout", "").split("\n") if "mode:" in ln or "template analysis" in ln or "explicit-form" in ln or "template mode" in ln]
    print(json.dumps({"case": tag, "host": host, "evals": n_long - n_pilot, "seconds": dt, "evals_per_s": (n_long - n_pilot) / dt,
                      "method": f"wall clock of two runs, {n_pilot} and {n_long} evaluations: ({n_long} - {n_pilot}) / ({w1:.2f} s - {w0:.2f} s)",
                      "hbl_timer_seconds_long_run": res.get("sweep_seconds"), "logl": res["logl"], "adapter_says": mode[-3:], **(extra or {})}),
          flush=True)


if "headline" in which:
    block = hbl.codon_model_block(tmpl, pi)
    for host, binary, env, count, floor in (("adapter (template mode)", HIP_BIN, dict(ENV, HYPHY_HIP_DEVICE_EXPM="always"), n_evals, 50000),
                                             ("adapter, HYPHY_HIP_TEMPLATES=0 (dense mode B)", HIP_BIN, dict(ENV, HYPHY_HIP_DEVICE_EXPM="always", HYPHY_HIP_TEMPLATES="0"), n_evals // 2, 0),
                                             ("reference 16 threads", None, None, max(40, n_evals // 100), 0)):
        measure("mg94_64x10k", host, lambda n: hbl.evaluate(model_block=block, globals_=dict(R=0.3, **bench.REV),
                                                           sweep=dict(param="R", start=0.3, step=0.00001, n=n), threads=(1 if binary else 16),
                                                           binary=binary, extra_env=env, **common), count, min_evals=floor)

if "class2" in which:
    # MG94 with two local parameters per branch and branch-specific constraints: foreground / background omega
    lines = ["MGQ = {61,61};"]
    for (i, j, name, ns, pf) in tmpl:
        parts = ([name] if name != "AG" else []) + ["nonSynRate" if ns else "synRate", repr(float(pf))]
        lines.append(f"MGQ[{i}][{j}] := {'*'.join(parts)};")
    lines.append("vectorOfFrequencies = {\n" + ",\n".join("{" + repr(float(v)) + "}" for v in pi) + "};")
    lines.append("Model MGM = (MGQ, vectorOfFrequencies, 0);")
    tmp = tempfile.mkdtemp(prefix="hyclass_")
    fasta, outp = os.path.join(tmp, "aln.fasta"), os.path.join(tmp, "out.txt")
    hbl.write_fasta(fasta, syn.flat.leaf_names, syn.seqs)
    for host, binary, env, count, thr, floor in (("adapter (template mode, one group per branch class)", HIP_BIN, dict(ENV, HYPHY_HIP_DEVICE_EXPM="always"), n_evals, 1, 50000),
                                                  ("adapter, HYPHY_HIP_TEMPLATES=0 (dense mode B)", HIP_BIN, dict(ENV, HYPHY_HIP_DEVICE_EXPM="always", HYPHY_HIP_TEMPLATES="0"), n_evals // 2, 1, 0),
                                                  ("reference 16 threads", None, None, max(40, n_evals // 100), 16, 0)):
        def run_class2(count):
            txt = hbl.build_script(fasta=fasta, newick=htree.to_newick(syn.tree), unit=3, model_block="\n".join(lines), model_name="MGM",
                                   globals_=dict(R=0.3, R2=0.5, **bench.REV), branch_t=bt, out_path=outp, per_site=False,
                                   sweep=dict(param="R", start=0.3, step=0.00001, n=count), threads=thr)
            for k, (nm, t) in enumerate(bt.items()):
                om = "R2" if k % 3 == 0 else "R"
                txt = txt.replace(f"givenTree.{nm}.t = {hbl._fmt(t)};",
                                  f"givenTree.{nm}.synRate = {hbl._fmt(t)}; givenTree.{nm}.nonSynRate := {om}*givenTree.{nm}.synRate;")
            assert ".t = " not in txt
            stdout = hbl.run_script(txt, tmp, cpus=thr, timeout=1800.0, binary=binary, extra_env=env)
            res = hbl.parse_output(outp)
            res["stdout"] = stdout
            return res
        try:
            measure("mg94_two_omega_classes_64x10k", host, run_class2, count, min_evals=floor)
        except Exception as e:
            print(json.dumps({"case": "class2", "host": host, "error": str(e)[-600:]}), flush=True)

if "cat3" in which:
    block = hbl.codon_model_block(tmpl, pi, omega="R*cc")
    cat = dict(name="cc", weights=[0.7, 0.25, 0.05], values=[0.1 / 0.3, 1.0 / 0.3, 5.0 / 0.3])
    # (LFCompute outside Optimize: device exponentials have to be asked for, as in every other case of this script)
    for host, binary, env, count, floor in (("adapter", HIP_BIN, dict(ENV, HYPHY_HIP_DEVICE_EXPM="always"), n_evals // 2, 20000),
                                             ("adapter, HYPHY_HIP_DEVICE_EXPM unset (mode A outside Optimize: host exponentials)", HIP_BIN, ENV, max(100, n_evals // 40), 0),
                                             ("reference 16 threads", None, None, max(12, n_evals // 400), 0)):
        if os.environ.get("ADAPTER_RATE_ROWS") == "first" and host != "adapter":
            continue
        measure("cat3_64x10k", host, lambda n: hbl.evaluate(model_block=block, globals_=dict(R=0.3, **bench.REV), category=cat,
                                                           sweep=dict(param="R", start=0.3, step=0.00001, n=n), threads=(1 if binary else 16),
                                                           binary=binary, extra_env=env, **common), count, min_evals=floor)

if "mix3" in which:
    block = hbl.codon_mixture_model_block(tmpl, pi, ["R1", "R2", "R3"], ["W1", "W2", "(1-W1-W2)"])
    g = dict(R1=0.1, R2=1.0, R3=5.0, W1=0.6, W2=0.3, **bench.REV)
    # three sweeps: a mixture weight (no rate matrix changes: the reference re-mixes cached exponentials), one component's omega
    # (one component of every branch changes), a nucleotide rate (every component of every branch changes)
    for param, start, step in (("W1", 0.5, 0.000001), ("R2", 1.0, 0.00001), ("AC", 0.5, 0.00001)):
        for host, binary, env, count, floor in (("adapter", HIP_BIN, dict(ENV, HYPHY_HIP_DEVICE_EXPM="always"), n_evals // 2, 20000),
                                                 ("adapter, HYPHY_HIP_TEMPLATES=0 (dense components)", HIP_BIN, dict(ENV, HYPHY_HIP_DEVICE_EXPM="always", HYPHY_HIP_TEMPLATES="0"), max(200, n_evals // 10), 0),
                                                 ("reference 16 threads", None, None, max(8, n_evals // 500), 0)):
            measure(f"mix3_64x10k sweep of {param}", host,
                    lambda n: hbl.evaluate(model_block=block, globals_=g, upper_bounds=dict(W1=1.0, W2=1.0), sweep=dict(param=param, start=start, step=step, n=n),
                                           threads=(1 if binary else 16), binary=binary, extra_env=env, **common), count, min_evals=floor)

if "manylf" in which:
    # N single-codon likelihood functions, 50 evaluations each (FEL's shape: one LF per site)
    tmp = tempfile.mkdtemp(prefix="hymany_")
    fasta = os.path.join(tmp, "aln.fasta")
    hbl.write_fasta(fasta, syn.flat.leaf_names, [s[: 3 * n_lfs] for s in syn.seqs])
    outp = os.path.join(tmp, "out.txt")
    L = ["VERBOSITY_LEVEL = -1;", "PRINT_DIGITS = 17;"] + [f"global {k} = {v!r};" for k, v in dict(R=0.3, **bench.REV).items()]
    L.append(hbl.codon_model_block(tmpl, pi))
    L.append("UseModel (MGM);")
    L.append(f"Tree givenTree = {htree.to_newick(syn.tree)};")
    L.append(f'DataSet ds = ReadDataFile ("{fasta}");')
    for nm, t in bt.items():
        L.append(f"givenTree.{nm}.t = {t!r};")
    L.append("tot_ = 0; t0_ = Time (1);")
    L.append("for (s_ = 0; s_ < N_LFS_; s_ += 1) {")
    L.append('  DataSetFilter sf_ = CreateFilter (ds, 3, "" + (3*s_) + "-" + (3*s_+2), "", "TAA,TAG,TGA");')
    L.append("  LikelihoodFunction slf_ = (sf_, givenTree);")
    L.append("  LFCompute (slf_, LF_START_COMPUTE);")
    L.append("  for (k_ = 0; k_ < 50; k_ += 1) { R = 0.3 + 0.001*k_; LFCompute (slf_, r_); }")
    L.append("  LFCompute (slf_, LF_DONE_COMPUTE);")
    L.append("  tot_ += r_;")
    L.append('  if (s_ % 200 == 199) { fprintf (stdout, "PROGRESS ", s_ + 1, " likelihood functions after ", Time (1) - t0_, " s\\n"); }')
    L.append("}")
    L.append("t1_ = Time (1);")
    L.append(f'fprintf ("{outp}", CLEAR_FILE, "LOGL ", Format (tot_, 30, 17), "\\n", "SWEEP_SECONDS ", Format (t1_-t0_, 20, 6), "\\n");')
    script = "\n".join(L) + "\n"
    n_ref = max(4, n_lfs // 40)   # (the host spends ~10 ms per evaluation on the 125 exponentials of a one-codon LF)
    for host, binary, env, n_here in (("adapter, every LF on the device", HIP_BIN, dict(ENV, HYPHY_HIP_MIN_PATTERNS="0", HYPHY_HIP_VERBOSE="0", HYPHY_HIP_DEVICE_EXPM="always"), n_lfs),
                                      ("adapter, schedule tuner cache off", HIP_BIN, dict(ENV, HYPHY_HIP_MIN_PATTERNS="0", HYPHY_HIP_TUNE_CACHE="0", HYPHY_HIP_VERBOSE="0", HYPHY_HIP_DEVICE_EXPM="always"), n_lfs),
                                      ("adapter, HYPHY_HIP_DEVICE_EXPM unset (host exponentials outside Optimize)", HIP_BIN, dict(ENV, HYPHY_HIP_VERBOSE="0"), max(4, n_lfs // 10)),
                                      ("reference 1 thread", None, None, n_ref)):
        if os.environ.get("ADAPTER_RATE_ROWS") == "first" and not host.startswith("adapter, every"):
            continue
        t0 = time.time()
        try:
            stdout = hbl.run_script(script.replace("N_LFS_", str(n_here)), tmp, cpus=1, timeout=1800.0, binary=binary, extra_env=env)
            res = hbl.parse_output(outp)
            res["stdout"] = stdout
            prog = [ln.replace("PROGRESS ", "") for ln in stdout.split("\n") if ln.startswith("PROGRESS")]
            emit("manylf_64taxa_1codon_x50", host, res, 50 * n_here, t0, extra={"likelihood_functions": n_here, "progress": prog})
        except Exception as e:  # (report and go on: the other hosts still run)
            print(json.dumps({"case": "manylf", "host": host, "error": str(e)[-600:]}), flush=True)
