#!/usr/bin/env python3
"""Design-space simulator for the pruning kernels (host-side model, no GPU): how long does ONE full evaluation take
for a given decomposition of (tiles x tree) into workgroups / waves / chained fragments, under the measured behaviour
of v_mfma_f64_16x16x4_f64 on gfx950 (profiles/r01_ubench_mfma_f64.txt):

  * a SIMD issues at most one f64 MFMA per ~100 cycles; ONE wave alone gets one per ~143 cycles;
  * waves resident per SIMD limited by registers (occ);
  * a hand-off through global memory at agent scope costs the consumer ~4 us, through LDS / same-CU L2 ~0.2-0.5 us.

The model is processor sharing per SIMD (waves in an MFMA phase share the pipe equally), plus fixed latencies for
the non-MFMA steps.  It is a planning tool: it reproduces the measured r01 numbers of the wave-per-tile kernel to
~10 % (see `--validate`) and is used to choose team size, fragment cut and occupancy before building a kernel.
"""
import argparse
import heapq
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

CLK = 2.4e3  # cycles per microsecond


class Sim:
    def __init__(self, n_cu=256, occ=2, solo=143.0, shared=100.0):
        self.n_cu, self.occ, self.solo, self.shared = n_cu, occ, solo, shared
        self.nsimd = n_cu * 4
        self.active = [dict() for _ in range(self.nsimd)]   # simd -> {wave: remaining mfma}
        self.last_t = [0.0] * self.nsimd
        self.ver = [0] * self.nsimd
        self.resident = [0] * self.nsimd                   # waves resident per simd
        self.ev = []                                       # (time, seq, kind, payload)
        self.seq = 0
        self.t = 0.0
        self.busy_integral = 0.0                           # sum over simds of pipe-busy cycles

    def push(self, t, kind, payload):
        self.seq += 1
        heapq.heappush(self.ev, (t, self.seq, kind, payload))

    def rate(self, k):  # MFMAs per cycle per wave
        if k <= 0:
            return 0.0
        if k == 1:
            return 1.0 / self.solo
        return 1.0 / (self.shared * k)

    def _advance(self, s):
        a = self.active[s]
        dt = self.t - self.last_t[s]
        if a and dt > 0:
            r = self.rate(len(a))
            for w in a:
                a[w] -= dt * r
            self.busy_integral += dt * min(1.0, len(a) * r * self.shared)
        self.last_t[s] = self.t

    def _resched(self, s):
        self.ver[s] += 1
        a = self.active[s]
        if not a:
            return
        r = self.rate(len(a))
        w = min(a, key=lambda x: a[x])
        self.push(self.t + max(a[w], 0.0) / r, "mfma_done", (s, self.ver[s], w))

    def start_mfma(self, wave, n):
        s = wave.simd
        self._advance(s)
        self.active[s][wave] = float(n)
        self._resched(s)

    def run(self):
        while self.ev:
            t, _, kind, p = heapq.heappop(self.ev)
            if kind == "mfma_done":
                s, v, w = p
                if v != self.ver[s]:
                    continue
                self.t = t
                self._advance(s)
                del self.active[s][w]
                self._resched(s)
                w.step()
            elif kind == "wake":
                self.t = t
                p.step()
            elif kind == "call":
                self.t = t
                p()
        return self.t


class Wave:
    """A wave runs a generator yielding ('mfma', n) / ('delay', cycles) / ('block',) actions."""
    def __init__(self, sim, simd, gen_fn):
        self.sim, self.simd = sim, simd
        self.gen = gen_fn(self)
        self.done = False
        self.on_exit = None

    def step(self):
        try:
            act = next(self.gen)
        except StopIteration:
            self.done = True
            if self.on_exit:
                self.on_exit(self)
            return
        if act[0] == "mfma":
            if act[1] <= 0:
                self.sim.push(self.sim.t, "wake", self)
            else:
                self.sim.start_mfma(self, act[1])
        elif act[0] == "delay":
            self.sim.push(self.sim.t + act[1], "wake", self)
        elif act[0] == "block":
            pass  # someone else will call sim.push(..., "wake", self)

    def wake(self, delay=0.0):
        self.sim.push(self.sim.t + delay, "wake", self)


class Dispatcher:
    """Workgroups are dispatched in order to the CU with the most free wave slots (ties: lowest index), W waves on
    SIMDs (k + start) % 4, limited by `occ` waves per SIMD and wgs_per_cu (LDS)."""
    def __init__(self, sim, W, wgs_per_cu=99):
        self.sim, self.W, self.wgs_per_cu = sim, W, wgs_per_cu
        self.queue = []
        self.cu_wgs = [0] * sim.n_cu
        self.rr = 0

    def submit(self, wg_factory):
        self.queue.append(wg_factory)

    def try_dispatch(self):
        sim = self.sim
        while self.queue:
            best, best_free = -1, -1
            for k in range(sim.n_cu):  # round-robin start so that equal CUs are filled evenly
                cu = (self.rr + k) % sim.n_cu
                if self.cu_wgs[cu] >= self.wgs_per_cu:
                    continue
                simds = sorted(range(4), key=lambda s: sim.resident[cu * 4 + s])
                need = [0, 0, 0, 0]
                for i in range(self.W):
                    need[simds[i % 4]] += 1
                if all(sim.resident[cu * 4 + s] + need[s] <= sim.occ for s in range(4)):
                    free = sum(sim.occ - sim.resident[cu * 4 + s] for s in range(4))
                    if free > best_free:
                        best, best_free = cu, free
            if best < 0:
                return
            self.rr = (best + 1) % sim.n_cu
            cu = best
            fac = self.queue.pop(0)
            simds = sorted(range(4), key=lambda s: sim.resident[cu * 4 + s])
            self.cu_wgs[cu] += 1
            left = [self.W]

            def on_exit(w, cu=cu):
                sim.resident[w.simd] -= 1
                left[0] -= 1
                if left[0] == 0:
                    self.cu_wgs[cu] -= 1
                self.sim.push(self.sim.t, "call", self.try_dispatch)

            waves = fac(sim, [cu * 4 + simds[i % 4] for i in range(self.W)])
            for w in waves:
                sim.resident[w.simd] += 1
                w.on_exit = on_exit
                w.wake(0.0)


# ------------------------------------------------------------------------------------------------
# tree helpers
# ------------------------------------------------------------------------------------------------
def tree_arrays(flat_parents, L):
    I = len(flat_parents) - L
    par = [int(flat_parents[L + i]) for i in range(I)]
    ich = [[] for _ in range(I)]   # internal children (internal indices)
    nleaf = [0] * I
    for n in range(L + I):
        p = int(flat_parents[n])
        if p < 0:
            continue
        if n < L:
            nleaf[p] += 1
        else:
            ich[p].append(n - L)
    return I, par, ich, nleaf


def cut_fragments(I, par, ich, max_frag):
    """api.hip build_schedule level peeling; returns programs = list of node lists, prog_of[node], parent prog."""
    done = [False] * I
    progs = []
    while True:
        size = [0] * I
        for n in range(I):
            if not done[n]:
                size[n] = 1 + sum(size[c] for c in ich[n])
        root = I - 1
        if size[root] <= max_frag:
            frags = [[n for n in range(I) if not done[n]]]
        else:
            fr = [-1] * I
            for n in range(I - 1, -1, -1):
                if done[n]:
                    continue
                p = par[n]
                if p >= 0 and not done[p] and fr[p] >= 0:
                    fr[n] = fr[p]
                elif size[n] <= max_frag:
                    fr[n] = n
            d = {}
            for n in range(I):
                if not done[n] and fr[n] >= 0:
                    d.setdefault(fr[n], []).append(n)
            frags = sorted(d.values(), key=lambda f: -len(f))
        for f in frags:
            for n in f:
                done[n] = True
            progs.append(f)
        if done[root]:
            break
    prog_of = [-1] * I
    for k, f in enumerate(progs):
        for n in f:
            prog_of[n] = k
    pparent = [prog_of[par[f[-1]]] if par[f[-1]] >= 0 else -1 for f in progs]
    return progs, prog_of, pparent


# ------------------------------------------------------------------------------------------------
# the "flow" design: a unit = (program, tile) run by a team of W waves; chains start at sources (nodes without
# in-program internal children, or handed-off child fragments), the last arriver at a node continues upwards
# ------------------------------------------------------------------------------------------------
class Costs:
    def __init__(self, **kw):
        self.mfma_edge = 64          # MFMAs per internal edge (whole tile rows, one wave)
        self.t_ticket = 250          # LDS atomic + descriptor fetch
        self.t_source = 700          # leaf gathers of a cherry-like source node
        self.t_global_src = 1200     # child vector from global memory (persisted / handed off), same CU or L2
        self.t_edge = 350            # per edge: operand stream start-up, acc multiply
        self.t_join_last = 700       # last arriver: read deposit(s), multiply, leaf gathers, finalise
        self.t_deposit = 900         # non-last arriver: store 8 KiB + drain + arrival
        self.t_publish = 3500        # fragment root: sc1 stores + drain + global arrival
        self.t_handoff = 9600        # consumer side latency of a cross-WG hand-off (~4 us)
        self.t_root = 600            # root epilogue
        self.__dict__.update(kw)


def run_flow(flat_parents, L, ntiles, n_cat=1, W=4, occ=4, max_frag=99, wgs_per_cu=99, rendezvous=True, costs=None,
             n_cu=256, solo=143.0, shared=100.0, order="tile", verbose=False):
    c = costs or Costs()
    I, par, ich, nleaf = tree_arrays(flat_parents, L)
    progs, prog_of, pparent = cut_fragments(I, par, ich, max_frag)
    nprog = len(progs)
    need_prog = [sum(1 for k in range(nprog) if pparent[k] == q) for q in range(nprog)]
    # per program: sources and arrival needs
    info = []
    for q, f in enumerate(progs):
        fs = set(f)
        need = {}
        sources = []
        for n in f:
            in_ch = [x for x in ich[n] if x in fs]
            out_ch = [x for x in ich[n] if x not in fs]      # handed-off fragment roots
            need[n] = len(in_ch) + len(out_ch)
            if need[n] == 0:
                sources.append(("leaf", n))
            for x in out_ch:
                sources.append(("global", x, n))
        info.append((f, need, sources))
    sim = Sim(n_cu=n_cu, occ=occ, solo=solo, shared=shared)
    disp = Dispatcher(sim, W, wgs_per_cu)
    frag_arrivals = {}
    finish = [0.0]

    def make_unit(tile, cat, q0):
        def factory(sim, simds):
            state = {"prog": q0, "ticket": 0, "arr": {}, "idle": [], "alive": W}

            def body(wave):
                while True:
                    q = state["prog"]
                    f, need, sources = info[q]
                    tk = state["ticket"]
                    if tk >= len(sources):
                        if not rendezvous:
                            return
                        # wait for the program's fate (next program or exit)
                        state["idle"].append(wave)
                        yield ("block",)
                        if state["prog"] is None:
                            return
                        continue
                    state["ticket"] += 1
                    yield ("delay", c.t_ticket)
                    src = sources[tk]
                    if src[0] == "leaf":
                        node = src[1]
                        yield ("delay", c.t_source)
                    else:
                        node = None
                        child, into = src[1], src[2]
                        yield ("delay", c.t_global_src + c.t_handoff * 0.0)
                    # chain upwards
                    while True:
                        if node is None:
                            p = into
                        else:
                            if node == f[-1]:
                                break
                            p = par[node]
                        yield ("delay", c.t_edge)
                        yield ("mfma", c.mfma_edge)
                        a = state["arr"].get((q, p), 0) + 1
                        state["arr"][(q, p)] = a
                        if a < need[p]:
                            yield ("delay", c.t_deposit)
                            node = "stop"
                            break
                        yield ("delay", c.t_join_last + 300 * (need[p] - 1))
                        node = p
                    if node == "stop":
                        continue
                    # finished the program root
                    if pparent[q] < 0:
                        yield ("delay", c.t_root)
                        finish[0] = max(finish[0], sim.t)
                        state["prog"] = None
                    else:
                        yield ("delay", c.t_publish)
                        key = (tile, cat, pparent[q])
                        frag_arrivals[key] = frag_arrivals.get(key, 0) + 1
                        if frag_arrivals[key] < need_prog[pparent[q]]:
                            state["prog"] = None
                        else:
                            yield ("delay", c.t_handoff)
                            state["prog"] = pparent[q]
                            state["ticket"] = 0
                    idle, state["idle"] = state["idle"], []
                    for w in idle:
                        w.wake(100.0)
                    if state["prog"] is None:
                        return
            return [Wave(sim, s, body) for s in simds]
        return factory

    leaf_progs = [q for q in range(nprog) if need_prog[q] == 0]
    if order == "tile":
        for tile in range(ntiles):
            for cat in range(n_cat):
                for q in leaf_progs:
                    disp.submit(make_unit(tile, cat, q))
    else:
        for q in leaf_progs:
            for cat in range(n_cat):
                for tile in range(ntiles):
                    disp.submit(make_unit(tile, cat, q))
    disp.try_dispatch()
    sim.run()
    T = finish[0]
    if verbose:
        print("programs:", [len(f) for f in progs], "leaf programs:", len(leaf_progs))
    total_mfma = ntiles * n_cat * (I - 1) * c.mfma_edge
    util = total_mfma * shared / (T * sim.nsimd)
    return T / CLK, util


def bench_tree(taxa, seed):
    from hyphy_amd import data
    syn = data.evolve(taxa, 30, 3, seed=seed)
    return np.asarray(syn.flat.flat_parents), syn.flat.L


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--taxa", type=int, default=64)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--tiles", type=int, default=624)
    ap.add_argument("--cat", type=int, default=1)
    ap.add_argument("--validate", action="store_true")
    args = ap.parse_args()
    fp, L = bench_tree(args.taxa, args.seed)
    if args.validate:
        # r01 wave-per-tile kernel: W = 1, 2 waves/SIMD, hand-off ~4 us, register hand-over inside a fragment
        wave = Costs(t_ticket=0, t_deposit=900, t_join_last=900)
        for tiles, mf, cat, meas in [(624, 25, 1, 152), (624, 25, 3, 360), (313, 12, 1, 160), (157, 6, 1, 104), (79, 4, 1, 92)]:
            us, util = run_flow(fp, L, tiles, n_cat=cat, W=1, occ=2, max_frag=mf, rendezvous=False, costs=wave)
            print(f"wave kernel model: tiles={tiles} max_frag={mf} classes={cat}: {us:7.1f} us (measured {meas}), pipe util {util:.2f}")
        sys.exit(0)
    for W in (1, 2, 4, 8):
        for occ in (2, 3, 4):
            for mf in (8, 12, 16, 25, 99):
                us, util = run_flow(fp, L, args.tiles, n_cat=args.cat, W=W, occ=occ, max_frag=mf)
                print(f"W={W} occ={occ} max_frag={mf:3d}: {us:7.1f} us  pipe util {util:.2f}")
