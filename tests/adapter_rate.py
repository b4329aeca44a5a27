"""End-to-end rate of the REAL host through the adapter: the patched hyphy binary
(integration/_build/hyphy_hip, HYPHY_HIP=1) runs the headline LFCompute sweep — HBL formula engine, rate-matrix
construction and (mode A) OpenMP exponentials on the host, pruning on the device — next to the unmodified reference.
Usage (GPU box): python tests/adapter_rate.py [n_evals] [threads,threads,...]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from hyphy_amd import data, models, tree as htree
from oracle import hbl

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
threads = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 8, 16]
wl = bench.WORKLOADS[os.environ.get("WORKLOAD", "mg94_64x10k")]
syn = data.evolve(wl["taxa"], wl["sites"], wl["unit"], seed=wl["seed"], p_change=wl.get("p_change", 0.04))
tmpl = models.mg94rev_template(bench.POS_FREQS)
pi = models.f3x4_codon_freqs(bench.POS_FREQS)
g = dict(R=0.3, **bench.REV)
bt = {nm: 0.05 for nm in syn.flat.branch_names()}
HIP_BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "integration", "_build", "hyphy_hip")
for thr in threads:
    for label, binary, env, count in (("adapter", HIP_BIN, dict(HYPHY_HIP="1", **{k: v for k, v in os.environ.items() if k.startswith("HYPHY_HIP_")}), n),
                                      ("reference", None, None, max(8, n // 40))):
        if label == "reference" and (thr not in (16,) or os.environ.get("NO_REFERENCE")):
            continue
        t0 = time.time()
        res = hbl.evaluate(names=syn.flat.leaf_names, seqs=syn.seqs, newick=htree.to_newick(syn.tree), unit=3,
                           model_block=hbl.codon_model_block(tmpl, pi), model_name="MGM", globals_=g, branch_t=bt,
                           sweep=dict(param="R", start=0.3, step=0.0001, n=count), threads=thr, per_site=False,
                           timeout=900.0, binary=binary, extra_env=env)
        secs = max(res.get("sweep_seconds", 0.0), 1.0)
        print(json.dumps({"host": label, "threads": thr, "evals": count, "sweep_seconds": secs, "evals_per_s": count / secs,
                          "logl": res["logl"], "wall": time.time() - t0}), flush=True)
