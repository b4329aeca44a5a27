"""Turn a tools/profile_round.sh output directory (gpurun_out/<tag>) into the committed summaries under
profiles/: bench lines, rocprofv3 kernel stats, per-kernel PMC means, HBM traffic of the pruning kernel
(2*FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md HBM section), microbenchmark text.
Usage: python tools/summarize_profiles.py gpurun_out/r01b r01"""
import collections, csv, glob, json, os, re, shutil, sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "profiles")


def kname(full):
    m = re.search(r"(\w+_kernel)", full)
    return m.group(1) if m else full.split("(")[0][:48]


for f in ("bench.json", "bench_alltimings.json"):
    shutil.copy(os.path.join(src, f), os.path.join(out, f"{tag}_{f}"))
for f, dst in (("ubench_mfma_f64.txt", f"{tag}_ubench_mfma_f64.txt"),
               ("kernel_choice_by_shard_size.txt", f"{tag}_kernel_choice_by_shard_size.txt"),
               ("all_workloads.txt", f"{tag}_all_workloads.txt"), ("adapter_rate.jsonl", f"{tag}_adapter_rate.jsonl"),
               ("stress.txt", f"{tag}_stress.txt")):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(out, dst))
st = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
if st:
    shutil.copy(st[0], os.path.join(out, f"{tag}_rocprofv3_kernel_stats.csv"))

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, "pmc", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[kname(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
means = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
json.dump(means, open(os.path.join(out, f"{tag}_pmc_per_kernel_means.json"), "w"), indent=1, sort_keys=True)

bench = json.load(open(os.path.join(src, "bench.json")))
kern = bench["roofline"]["kernel"]
launches = bench["roofline"].get("launches_per_step", 1)
if kern not in means:  # (older bench lines named the kernel family, not the variant)
    kern = next((k for k in means if k.startswith("prune_")), kern)
pm = means.get(kern, {})
if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
    per_launch = (2 * pm["FETCH_SIZE"] + pm["WRITE_SIZE"]) * 1024.0
    traffic = {bench["config"]["workload"]: {
        "kernel": kern, "launches_per_evaluation": launches,
        "FETCH_SIZE_KB": pm["FETCH_SIZE"] * launches, "WRITE_SIZE_KB": pm["WRITE_SIZE"] * launches,
        "hbm_bytes_per_launch": per_launch * launches,
        "note": "per evaluation (all pruning launches of one evaluation)",
        "correction": "(2*FETCH_SIZE + WRITE_SIZE) KB: MI355X_MICROARCH.md HBM section says FETCH_SIZE reads 1/2 of a wide "
                      "coalesced stream on gfx950; WRITE_SIZE matches the known persist volume here",
    }}
    json.dump(traffic, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
    print("traffic per evaluation: %.1f MB" % (per_launch * launches / 1e6))
print("kernels:", sorted(means))
