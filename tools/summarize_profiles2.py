"""Turn a tools/profile_round2.sh output directory (gpurun_out/<tag>) into the committed summaries under profiles/:
bench lines, rocprofv3 kernel stats, per-workload / per-kernel PMC means, HBM traffic of the dominant kernel
((2*FETCH_SIZE + WRITE_SIZE) KB, MI355X_MICROARCH.md HBM section), per-workload table, adapter rates, stress, wave timeline.
Usage: python tools/summarize_profiles2.py gpurun_out/r02 r02"""
import collections, csv, glob, json, os, re, shutil, subprocess, sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "profiles")


def kname(full):
    m = re.search(r"(\w+_kernel)", full)
    return m.group(1) if m else full.split("(")[0][:48]


def last_json(path):
    return json.loads([l for l in open(path) if l.startswith("{")][-1])


for f, dst in (("bench.json", f"{tag}_bench.json"), ("bench_driver_line.json", f"{tag}_bench_driver_line.json"),
               ("all_workloads.txt", f"{tag}_all_workloads.txt"), ("adapter_rate.jsonl", f"{tag}_adapter_rate.jsonl"),
               ("stress.txt", f"{tag}_stress.txt")):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(out, dst))
st = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
if st:
    shutil.copy(st[0], os.path.join(out, f"{tag}_rocprofv3_kernel_stats.csv"))

means_all, traffic = {}, {}
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    wl = os.path.basename(d)[4:]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[kname(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    # steady state: drop the first two dispatches of every kernel (persisting first pass, schedule tuner)
    # steady state: the MEDIAN over the dispatches of every kernel (the first evaluations persist everything and the schedule
    # tuner launches the pruning kernel under a dozen candidate cuts; means would mix those in)
    def med(v):
        w = sorted(v)
        return w[len(w) // 2] if len(w) % 2 else 0.5 * (w[len(w) // 2 - 1] + w[len(w) // 2])
    means = {k: {c: med(v) for c, v in dd.items()} for k, dd in acc.items()}
    means_all[wl] = means
    wj = os.path.join(src, f"wl_{wl}.json")
    if os.path.exists(wj):
        b = last_json(wj)
        kern, ms = b["roofline"]["kernel"], b["roofline"]["kernel_ms_per_launch"]
        pm = means.get(kern, {})
        if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
            bytes_ = (2 * pm["FETCH_SIZE"] + pm["WRITE_SIZE"]) * 1024.0
            traffic[wl] = {"kernel": kern, "FETCH_SIZE_KB": pm["FETCH_SIZE"], "WRITE_SIZE_KB": pm["WRITE_SIZE"],
                           "hbm_bytes_per_launch": bytes_, "kernel_ms_per_launch": ms,
                           "traffic_rate_GBs": bytes_ / (ms * 1e-3) / 1e9,
                           "alg_bytes_per_launch": b["roofline"]["alg_bytes_per_step"] / b["roofline"]["launches_per_step"],
                           "correction": "(2*FETCH_SIZE + WRITE_SIZE) KB: FETCH_SIZE counts 128-byte requests as 64 bytes for wide "
                                         "coalesced streams on gfx950 (MI355X_MICROARCH.md, HBM); WRITE_SIZE as reported"}
json.dump(means_all, open(os.path.join(out, f"{tag}_pmc_per_kernel_means.json"), "w"), indent=1, sort_keys=True)
json.dump(traffic, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
for wl, t in traffic.items():
    print(f"{wl:18s} {t['kernel']:18s} traffic {t['hbm_bytes_per_launch']/1e6:8.1f} MB/launch = {t['traffic_rate_GBs']:7.1f} GB/s  (algorithmic {t['alg_bytes_per_launch']/1e6:8.1f} MB)")
tl = os.path.join(src, "timeline_m12.txt")
if os.path.exists(tl):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "timeline_waves.py"), tl, "624"], stdout=subprocess.PIPE, text=True)
    open(os.path.join(out, f"{tag}_wave_timeline_headline_m12.txt"), "w").write(r.stdout)
# MFMA pipe utilisation of the pruning kernel from the SQ counters
util = {}
for wl in ("mg94_64x10k", "mg94_128x100k"):
    pm = means_all.get(wl, {}).get("prune_wave_kernel", {})
    if "SQ_VALU_MFMA_BUSY_CYCLES" in pm and "GRBM_GUI_ACTIVE" in pm:
        cycles = pm["GRBM_GUI_ACTIVE"] / 8.0   # (the counter is summed over the 8 XCDs)
        util[wl] = {"kernel_cycles": cycles, "SQ_INSTS_MFMA": pm.get("SQ_INSTS_MFMA"), "SQ_VALU_MFMA_BUSY_CYCLES": pm["SQ_VALU_MFMA_BUSY_CYCLES"],
                    "mfma_pipe_busy_fraction": pm["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cycles),
                    "issue_stalled_fraction_of_wave_cycles": pm.get("SQ_WAIT_INST_ANY", 0) / max(pm.get("SQ_WAVE_CYCLES", 1), 1),
                    "waiting_fraction_of_wave_cycles": pm.get("SQ_WAIT_ANY", 0) / max(pm.get("SQ_WAVE_CYCLES", 1), 1),
                    "waves": pm.get("SQ_WAVES"), "mean_resident_waves_per_simd": pm.get("SQ_WAVE_CYCLES", 0) * 4 / (1024 * cycles),
                    "LDS_bank_conflict_cycles": pm.get("SQ_LDS_BANK_CONFLICT"), "SQ_INSTS_LDS": pm.get("SQ_INSTS_LDS")}
        print(wl, json.dumps(util[wl]))
json.dump(util, open(os.path.join(out, f"{tag}_mfma_utilisation.json"), "w"), indent=1, sort_keys=True)
