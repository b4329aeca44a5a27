#!/usr/bin/env python3
"""Summarise a wave timeline of prune_wave_kernel (HYPHY_HIP_TIMELINE): dispatch profile, per-source durations,
resident waves over time, SIMD sharing.  usage: tools/timeline_waves.py file n_tiles"""
import sys
import numpy as np
path, ntiles = sys.argv[1], int(sys.argv[2])
a = np.loadtxt(path, comments="#")
a = a[a[:, 1] > 0]
t0 = a[:, 1].min()
st, pro, prg, en = (a[:, 1] - t0) / 100.0, (a[:, 2] - t0) / 100.0, (a[:, 3] - t0) / 100.0, (a[:, 4] - t0) / 100.0   # us
lev, how = a[:, 5].astype(int), a[:, 6].astype(int)
hw, xcc = a[:, 7].astype(np.int64), a[:, 8].astype(np.int64)
idx = a[:, 0].astype(int)
src = idx // ntiles            # chain grid: x = tile fastest
print(f"{len(a)} waves; launch span {en.max():.1f} us; starts: p50 {np.median(st):.1f} p90 {np.percentile(st,90):.1f} max {st.max():.1f}")
print(f"prologue: mean {np.mean(pro-st):.2f} us  p90 {np.percentile(pro-st,90):.2f};  total wave-time {np.sum(en-st)/1e3:.2f} ms = {np.sum(en-st)/2048:.1f} us per slot (2048 slots)")
# resident waves over time
ts = np.linspace(0, en.max(), 60)
res = [(np.sum((st <= t) & (en > t))) for t in ts]
print("resident waves every %.1f us:" % (ts[1]-ts[0]), " ".join(str(r) for r in res))
for s_ in range(src.max() + 1):
    m = src == s_
    if not m.any(): continue
    d = en[m] - st[m]
    nlev = lev[m] % 100
    pre = (lev[m] // 100) % 10
    raced = lev[m] // 1000
    print(f"source {s_:2d}: n {m.sum():4d} start {np.median(st[m]):6.1f}  program {np.median(prg[m]-pro[m]):6.1f} us  total {np.median(d):6.1f} (p90 {np.percentile(d,90):6.1f})  "
          f"levels {nlev.mean():.2f} prefetched {pre.mean():.2f} raced {raced.mean():.2f}  root {np.mean(how[m]==1):.2f}")
# SIMD sharing: hw id -> (xcc, se, cu, simd)
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; se = (hw >> 13) & 7
key = xcc * 100000 + se * 1000 + cu * 10 + simd
print("distinct SIMDs used:", len(np.unique(key)), " distinct CUs:", len(np.unique(key // 10)))
# phase cycles (records with 24 words, r04): where a wave's shader cycles go (prune.hip HYPHY_TR buckets)
if a.shape[1] >= 25:
    ph = a[:, 9:25]
    names = {0: "edge products", 2: "leaf entries", 4: "finalisations", 5: "trunk joins (arrive/deposit)", 6: "child tiles from global memory",
             7: "wave prologue", 8: "schedule-entry decode", 9: "trunk bookkeeping before the edge", 10: "deposits multiplied in",
             11: "between entry and finalisation", 12: "ahead of an edge product", 13: "epilogue / retirement"}
    tot = sum(ph[:, i].sum() for i in names)
    nsimd = len(np.unique(key))
    print(f"shader cycles, all waves: {tot/1e6:.1f} M = {tot/nsimd/1e3:.1f} k per SIMD used; mean clock {tot/np.sum(en-st)/1e3:.2f} GHz")
    for i in names:
        print(f"  {names[i]:36s} {ph[:, i].sum()/tot*100:5.1f} %   {ph[:, i].sum()/nsimd/1e3:7.1f} k cycles per SIMD")
    ne, nl = ph[:, 1].sum(), ph[:, 3].sum()
    print(f"  cycles per edge product {ph[:, 0].sum()/ne:.0f} ({ne:.0f} edges; 64 MFMAs each -> {ph[:, 0].sum()/ne/64:.1f} per MFMA);  per leaf {ph[:, 2].sum()/max(nl,1):.0f} ({nl:.0f})")
    print(f"  MFMA issue floor: {ne*64/nsimd:.0f} MFMAs per SIMD x 69 cycles = {ne*64/nsimd*69/1e3:.1f} k cycles per SIMD")
