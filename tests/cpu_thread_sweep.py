"""Times the real reference (oracle/_ref/hyphy) on the bench workload at several OpenMP thread
counts to pick the strongest CPU baseline configuration for this host."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from hyphy_amd import data
wl = bench.WORKLOADS["mg94_64x10k"]
syn = data.evolve(wl["taxa"], wl["sites"], wl["unit"], seed=wl["seed"])
for thr in [int(x) for x in sys.argv[1:]] or [1, 8, 16, 32, 64]:
    cb, ll, _ = bench.cpu_baseline(wl, syn, 0.3, 0.05, thr, budget_s=12.0)
    print(json.dumps(dict(threads=thr, evals_per_s=cb["value"], sample=cb["sample"][:40], logl=ll)), flush=True)
