"""GPU box diagnostic: a short headline sweep through the patched host with HYPHY_HIP_TRACE=1 (per-stage host microseconds of
every C-ABI call); prints the tail of the host's output.  usage: python tools/r03_adapter_trace.py [n] [extra ENV=val ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from hyphy_amd import data, models, tree as htree
from oracle import hbl
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
env = dict(HYPHY_HIP="1", HYPHY_HIP_VERBOSE="1", HYPHY_HIP_TRACE="1")
for kv in sys.argv[2:]:
    k, v = kv.split("=", 1)
    env[k] = v
wl = bench.WORKLOADS["mg94_64x10k"]
syn = data.evolve(wl["taxa"], wl["sites"], 3, seed=wl["seed"], p_change=0.04)
tmpl = models.mg94rev_template(bench.POS_FREQS)
pi = models.f3x4_codon_freqs(bench.POS_FREQS)
bt = {nm: 0.05 for nm in syn.flat.branch_names()}
t0 = time.time()
res = hbl.evaluate(names=syn.flat.leaf_names, seqs=syn.seqs, newick=htree.to_newick(syn.tree), unit=3, model_block=hbl.codon_model_block(tmpl, pi),
                   model_name="MGM", globals_=dict(R=0.3, **bench.REV), branch_t=bt, sweep=dict(param="R", start=0.3, step=0.0001, n=n), threads=1,
                   per_site=False, timeout=900.0, binary=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "integration", "_build", "hyphy_hip"),
                   extra_env=env)
print("wall", time.time() - t0, "sweep_seconds", res.get("sweep_seconds"), "logl", res["logl"])
lines = res["stdout"].split("\n")
print("\n".join(lines[:25]))
print("...")
print("\n".join(lines[-70:]))
