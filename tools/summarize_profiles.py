"""Turn a tools/profile_round.sh output directory (gpurun_out/<tag>) into the committed summaries under profiles/: the driver's
bench line, the per-workload table, rocprofv3 kernel stats, per-workload / per-kernel PMC medians, HBM traffic of the dominant kernel
((2*FETCH_SIZE + WRITE_SIZE) KB, MI355X_MICROARCH.md HBM section), MFMA-pipe utilisation and instruction mix of the pruning kernel,
the wave-cycle breakdown of the 4-state kernel, adapter rates, phase budgets.  Usage: python tools/summarize_profiles.py gpurun_out/r06 r06 [out dir]
(tools/profile_round.sh runs it on the GPU box into gpurun_out/<tag>/summary and deletes the raw counter files: they exceed what travels back)"""
import collections, csv, glob, json, os, re, shutil, sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles")   # (on the GPU box: a directory under gpurun_out/, copied to profiles/ here)
os.makedirs(out, exist_ok=True)


def kname(full):
    m = re.search(r"(\w+_kernel)", full)
    return m.group(1) if m else full.split("(")[0][:48]


def last_json(path):
    return json.loads([l for l in open(path) if l.startswith("{")][-1])


def med(v):
    w = sorted(v)
    return w[len(w) // 2] if len(w) % 2 else 0.5 * (w[len(w) // 2 - 1] + w[len(w) // 2])


for f, dst in (("bench_driver_line.json", f"{tag}_bench_driver_line.json"), ("all_workloads.txt", f"{tag}_all_workloads.txt"),
               ("adapter_rate.jsonl", f"{tag}_adapter_rate.jsonl")):
    if os.path.exists(os.path.join(src, f)) and os.path.getsize(os.path.join(src, f)) > 0:
        shutil.copy(os.path.join(src, f), os.path.join(out, dst))
st = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
if st:
    shutil.copy(st[0], os.path.join(out, f"{tag}_rocprofv3_kernel_stats.csv"))
for wl in ("gtr_32x1m", "gtr_32x50k"):
    st = glob.glob(os.path.join(src, f"stats_{wl}", "**", "*kernel_stats.csv"), recursive=True)
    if st:
        shutil.copy(st[0], os.path.join(out, f"{tag}_rocprofv3_kernel_stats_{wl}.csv"))
ph = sorted(glob.glob(os.path.join(src, "phases_*.txt")))
if ph:
    with open(os.path.join(out, f"{tag}_wave_phase_budget.txt"), "w") as fh:
        for f in ph:
            fh.write(open(f).read() + "\n")

means_all, traffic = {}, {}
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    wl = os.path.basename(d)[4:]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[kname(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    # steady state: the MEDIAN over the dispatches of every kernel (the first evaluations persist everything).  A lower phase of
    # several launches per evaluation (one per level of table dependencies, r05): the MEAN over the later half of its dispatches
    # times the launches per evaluation, so that every entry of the class kernel is per EVALUATION like the trunk's
    means_all[wl] = {k: {c: med(v) for c, v in dd.items()} for k, dd in acc.items()}
    trunk = "trunk_walk_kernel" if "trunk_walk_kernel" in acc else "prune_wave_kernel"   # (r06: the trunk as a row-split walk per tile)
    n_trunk = max((len(v) for v in acc.get(trunk, {}).values()), default=0)
    # (r06: the lower phase is class_table_team_kernel — a workgroup of row-split waves per item — unless HYPHY_HIP_REP_TEAM=0)
    lower = "class_table_team_kernel" if "class_table_team_kernel" in acc else "class_table_kernel"
    n_class = max((len(v) for v in acc.get(lower, {}).values()), default=0)
    if n_trunk and n_class > 1.2 * n_trunk:
        per_eval = round(n_class / n_trunk)
        means_all[wl][lower] = {c: per_eval * sum(v[len(v) // 2 // per_eval * per_eval:]) / max(1, len(v[len(v) // 2 // per_eval * per_eval:]))
                                for c, v in acc[lower].items()}
        means_all[wl][lower]["launches_per_evaluation"] = per_eval
    wj = os.path.join(src, f"wl_{wl}.json")
    if os.path.exists(wj):
        b = last_json(wj)
        kern, ms = b["roofline"]["kernel"], b["roofline"]["kernel_ms_per_launch"]
        pm = means_all[wl].get(kern, {})
        if " + " in kern:   # a class-compressed pass: lower phase + trunk, two launches whose traffic adds up
            parts = [means_all[wl].get(k.strip(), {}) for k in kern.split("+")]
            if all("FETCH_SIZE" in q and "WRITE_SIZE" in q for q in parts):
                pm = {c: sum(q[c] for q in parts) for c in ("FETCH_SIZE", "WRITE_SIZE")}
            ms = b["roofline"]["kernel_ms"]
        if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
            bytes_ = (2 * pm["FETCH_SIZE"] + pm["WRITE_SIZE"]) * 1024.0
            traffic[wl] = {"kernel": kern, "FETCH_SIZE_KB": pm["FETCH_SIZE"], "WRITE_SIZE_KB": pm["WRITE_SIZE"],
                           "hbm_bytes_per_launch": bytes_, "kernel_ms_per_launch": ms, "traffic_rate_GBs": bytes_ / (ms * 1e-3) / 1e9,
                           "alg_bytes_per_launch": b["roofline"]["alg_bytes_per_step"] / b["roofline"]["launches_per_step"],
                           "correction": "(2*FETCH_SIZE + WRITE_SIZE) KB: FETCH_SIZE counts 128-byte requests as 64 bytes for wide "
                                         "coalesced streams on gfx950 (MI355X_MICROARCH.md, HBM); WRITE_SIZE as reported"}
if means_all:
    json.dump(means_all, open(os.path.join(out, f"{tag}_pmc_per_kernel_means.json"), "w"), indent=1, sort_keys=True)
if traffic:
    old = {}
    try:
        old = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except Exception:
        pass
    old.update(traffic)
    json.dump(old, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
    for wl, t in traffic.items():
        print(f"{wl:18s} {t['kernel']:18s} traffic {t['hbm_bytes_per_launch']/1e6:8.1f} MB/launch = {t['traffic_rate_GBs']:7.1f} GB/s  (algorithmic {t['alg_bytes_per_launch']/1e6:8.1f} MB)")


def wave_cycle_table(pm, n_simd=1024):
    """SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md); GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
    wc = max(pm.get("SQ_WAVE_CYCLES", 1.0), 1.0)
    cycles = pm.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    waves = pm.get("SQ_WAVES", 0.0)
    t = {"kernel_cycles": cycles, "waves": waves, "wave_cycles_per_wave": 4.0 * wc / max(waves, 1.0),
         "mean_resident_waves_per_simd": 4.0 * wc / (n_simd * cycles) if cycles else None,
         "fraction_of_wave_cycles": {"issuing (SQ_ACTIVE_INST_ANY)": pm.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                                     "issue-stalled (SQ_WAIT_INST_ANY)": pm.get("SQ_WAIT_INST_ANY", 0) / wc,
                                     "waiting on a counter / barrier (SQ_WAIT_ANY)": pm.get("SQ_WAIT_ANY", 0) / wc,
                                     "issuing VALU": pm.get("SQ_ACTIVE_INST_VALU", 0) / wc, "issuing scalar": pm.get("SQ_ACTIVE_INST_SCA", 0) / wc,
                                     "issuing LDS": pm.get("SQ_ACTIVE_INST_LDS", 0) / wc, "issuing vector memory": pm.get("SQ_ACTIVE_INST_VMEM", 0) / wc,
                                     "issuing misc (branch, waitcnt, nop)": pm.get("SQ_ACTIVE_INST_MISC", 0) / wc,
                                     "stalled on the LDS queue (SQ_WAIT_INST_LDS)": pm.get("SQ_WAIT_INST_LDS", 0) / wc},
         "instructions_per_wave": {k[len("SQ_INSTS_"):]: pm[k] / max(waves, 1.0) for k in sorted(pm) if k.startswith("SQ_INSTS_")}}
    return t


util = {}
for wl in ("mg94_64x10k", "mg94_128x100k"):
    trunk = "trunk_walk_kernel" if "trunk_walk_kernel" in means_all.get(wl, {}) else "prune_wave_kernel"
    pm = means_all.get(wl, {}).get(trunk, {})
    if "SQ_VALU_MFMA_BUSY_CYCLES" in pm and "GRBM_GUI_ACTIVE" in pm:
        cycles = pm["GRBM_GUI_ACTIVE"] / 8.0
        u = wave_cycle_table(pm)
        u.update({"SQ_INSTS_MFMA": pm.get("SQ_INSTS_MFMA"), "SQ_VALU_MFMA_BUSY_CYCLES": pm["SQ_VALU_MFMA_BUSY_CYCLES"],
                  "mfma_pipe_busy_fraction (counter: 64 cycles per v_mfma_f64_16x16x4_f64)": pm["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cycles),
                  "mfma_instruction": "v_mfma_f64_16x16x4_f64 only (VGPR accumulators; tools/ubench/mfma4_skew.hip: 66-72 TFLOP/s with the kernel's operand stream)",
                  "LDS_bank_conflict_cycles": pm.get("SQ_LDS_BANK_CONFLICT")})
        lower = "class_table_team_kernel" if "class_table_team_kernel" in means_all.get(wl, {}) else "class_table_kernel"
        pc = means_all.get(wl, {}).get(lower, {})
        if "SQ_INSTS_MFMA" in pc:   # subtree repeats: the lower phase is a launch of its own
            uc = wave_cycle_table(pc)
            cyc = pc["GRBM_GUI_ACTIVE"] / 8.0
            uc.update({"SQ_INSTS_MFMA": pc["SQ_INSTS_MFMA"], "SQ_VALU_MFMA_BUSY_CYCLES": pc.get("SQ_VALU_MFMA_BUSY_CYCLES"),
                       "mfma_pipe_busy_fraction (counter: 64 cycles per v_mfma_f64_16x16x4_f64)": pc.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * cyc) if cyc else None})
            u = {"method": "counters only: SQ_VALU_MFMA_BUSY_CYCLES / (1 024 SIMDs x GRBM_GUI_ACTIVE / 8), collected in passes of their own "
                           "(profiled launches run a few per cent slower than production: compare fractions, not microseconds)",
                 trunk + " (trunk)": u, lower + " (lower phase)": uc,
                 "SQ_INSTS_MFMA_per_evaluation": pm.get("SQ_INSTS_MFMA", 0.0) + pc["SQ_INSTS_MFMA"],
                 "SQ_INSTS_MFMA_per_evaluation_without_repeats (r04, every internal edge at every pattern)": 2436096 if wl == "mg94_64x10k" else None,
                 "mfma_pipe_busy_fraction (counter: 64 cycles per v_mfma_f64_16x16x4_f64)":
                     (pm["SQ_VALU_MFMA_BUSY_CYCLES"] + pc.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)) / (1024 * (cycles + cyc)),
                 "mean_resident_waves_per_simd": u["mean_resident_waves_per_simd"]}
        util[wl] = u
        print(wl, "MFMA pipe busy", round(u["mfma_pipe_busy_fraction (counter: 64 cycles per v_mfma_f64_16x16x4_f64)"], 3),
              "resident waves/SIMD", round(u["mean_resident_waves_per_simd"] or 0, 2))
if util:
    json.dump(util, open(os.path.join(out, f"{tag}_mfma_utilisation.json"), "w"), indent=1, sort_keys=True)
nuc = {}
for wl in ("gtr_32x1m", "gtr_32x50k"):
    for kern in ("nucgen_kernel", "prune_nuc2_kernel"):   # (r06: the run-time generated kernel once it is compiled, the interpreter before)
        pm = means_all.get(wl, {}).get(kern, {})
        if "SQ_WAVE_CYCLES" in pm:
            nuc.setdefault(wl, {})[kern] = wave_cycle_table(pm)
    if wl in traffic and wl in nuc:
        nuc[wl]["hbm_traffic"] = traffic[wl]
if nuc:
    json.dump(nuc, open(os.path.join(out, f"{tag}_nuc_wave_cycles.json"), "w"), indent=1, sort_keys=True)
for f, dst in (("ubench_agpr_vs_vgpr.txt", None), ("ubench_edge_product.txt", None), ("ubench_edge_plus_leaf.txt", None)):
    pass  # (the microbenchmark outputs are committed by hand with their headers: profiles/r04_ubench_*.txt)
