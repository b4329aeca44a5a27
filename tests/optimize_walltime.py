"""Wall time of a complete maximum-likelihood fit (HBL Optimize: all branch lengths + global rates) of the headline
alignment through the REAL host: patched hyphy (device) or the unmodified reference (CPU).
Usage: python tests/optimize_walltime.py adapter|reference [threads] [workload]"""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from hyphy_amd import data, models, tree as htree
from oracle import hbl

which = sys.argv[1] if len(sys.argv) > 1 else "adapter"
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 8
wl_name = sys.argv[3] if len(sys.argv) > 3 else "mg94_64x10k"
wl = bench.WORKLOADS[wl_name]
syn = data.evolve(wl["taxa"], wl["sites"], wl["unit"], seed=wl["seed"], p_change=wl.get("p_change", 0.04))
tmpl = models.mg94rev_template(bench.POS_FREQS)
pi = models.f3x4_codon_freqs(bench.POS_FREQS)
tmp = tempfile.mkdtemp(prefix="optwall_")
fasta, outp = os.path.join(tmp, "aln.fasta"), os.path.join(tmp, "out.txt")
hbl.write_fasta(fasta, syn.flat.leaf_names, syn.seqs)
txt = hbl.build_script(fasta=fasta, newick=htree.to_newick(syn.tree), unit=3, model_block=hbl.codon_model_block(tmpl, pi),
                       model_name="MGM", globals_=dict(R=0.3, **bench.REV), branch_t={n: 0.05 for n in syn.flat.branch_names()},
                       out_path=outp, threads=threads, per_site=False)
txt += ("OPTIMIZATION_PRECISION = 0.001; VERBOSITY_LEVEL = -1; MAXIMUM_ITERATIONS_PER_VARIABLE = 10000; OPTIMIZATION_TIME_HARD_LIMIT = 100000;\n"
        "t0_ = Time (1); c0_ = Time (0);\nOptimize (mles2_, lf);\nt1_ = Time (1); c1_ = Time (0);\n"
        f'fprintf ("{outp}", "OPT_LOGL ", Format (mles2_[1][0], 30, 17), "\\n", "SWEEP_SECONDS ", Format (t1_-t0_, 20, 6), "\\n", "SWEEP_LAST ", Format (c1_-c0_, 20, 6), "\\n");\n')
binary = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "integration", "_build", "hyphy_hip") if which == "adapter" else None
env = dict(HYPHY_HIP="1", HYPHY_HIP_VERBOSE="1") if which == "adapter" else None
t0 = time.time()
out = hbl.run_script(txt, tmp, cpus=threads, timeout=6 * 3600.0, binary=binary, extra_env=env)
res = hbl.parse_output(outp)
tail = [l for l in out.splitlines() if "hyphy_hip" in l][-1:] if which == "adapter" else []
print(json.dumps({"host": which, "workload": wl_name, "threads": threads, "optimize_seconds": res.get("sweep_seconds"), "optimize_cpu_seconds": res.get("sweep_last"),
                  "opt_logl": res.get("opt_logl"), "start_logl": res.get("logl"), "wall_total": time.time() - t0, "device_counters": tail}))
