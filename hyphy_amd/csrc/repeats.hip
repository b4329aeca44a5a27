// Subtree repeats on the device — the counterpart of the reference's `tcc` traversal masks (src/core/tree.cpp:2801-2858
// marks, per internal node and site position, a subtree whose leaf states equal the previous site's;
// src/core/likefunc.cpp:10854 passes the mask on every ComputeBlock; src/core/tree_evaluator.cpp:57-76 skips the node and
// :240-256 copies the previous site's child vector; OptimalOrder, likefunc.cpp:11463+, orders sites to lengthen the runs).
//
// The reference's form is a run-length one (it helps a serial walk over sorted sites).  The device form is per-node
// pattern-CLASS compression: two patterns are in the same class of internal node n when the leaves below n carry the
// same states.  Classes are static per partition shard (they depend on the alignment and the topology only):
//   host, once per partition   class ids bottom-up (class of n at pattern s = id of the tuple of its children's classes),
//                              U_n = number of classes of n; the COMPRESSED nodes are the maximal subtrees with U_n <=
//                              theta * patterns, every other internal node is a TRUNK node;
//   lower phase, per evaluation  class_table_kernel: for every compressed node n and class u
//                                    E_n[u] = P_n x prod_c E_c[class_c(u)]
//                              — one [D x D] x [D x 16 classes] MFMA product per tile of 16 classes, children read by
//                              class index from THEIR tables (a leaf child: the column of P_leaf of its state; a leaf with
//                              ambiguity codes: a table of its own over its distinct codes), result stored edge-applied in
//                              the column-gather layout of the leaf matrices ([class][w][g][r]) with its 2^64 exponent;
//   trunk phase                the pruning kernels (prune.hip) over the trunk, in which a compressed subtree root is a
//                              GENERALISED LEAF: a 64-vector gather by class id + an exponent.
// Work per evaluation: sum of U_n over the compressed nodes + (trunk edges) x patterns edge products instead of
// (internal edges) x patterns: 0.26 of them on the headline alignment at theta = 0.5 (tools/repeat_stats.py).
//
// Scheduling inside the lower phase.  A tile needs rows of its input tables from anywhere in them, so the dependency is
// table -> table.  The work items (table, tile) form one sequence in which every table precedes its readers, dealt round-robin
// over 32 queues.  A wave takes tickets with one returning atomic on the head of its XCD's queue (another queue when that one is
// empty) and, before it works on an item, makes sure that the tickets of all the item's inputs are SOLD (every queue head beyond
// the item's bound) — unsold earlier tickets it takes and runs first (a small stack of held tickets, earliest on top).  Inside an
// item it then waits, where the walk reaches them, on per-table counters of finished tiles.  A wave therefore only sleeps on
// work that has a holder, and the holder of the earliest unfinished item never sleeps: no dispatch order, residency or workgroup
// -> XCD placement is assumed (MI355X_MICROARCH.md, "placement-independent protocols only").
// Hand-off of table rows between waves of the launch: 16-byte sc1 (write-through) stores -> asm "s_waitcnt vmcnt(0)" ->
// relaxed agent-scope RMW on the node's counter; consumer: relaxed agent-scope load that proves the count -> sc1 loads (the
// same contract as the chain joins of prune.hip).
#include <functional>
#include <map>
#include <mutex>
#include <tuple>
#include <unordered_map>

#include "devutil.h"
#include "combine.h"
#include "partition.h"

namespace hyhip {

// Paths.  A table per compressed node means a memory hand-off per tree level (publish -> drain -> counter -> poll -> gather:
// ~6 us of fabric round trips), and the lower phase is bound by exactly that chain, not by its arithmetic (the headline's
// 5 000 tiles are 5 us of the chip's matrix pipes).  A descriptor is therefore a PATH: a compressed node together with the
// chain of heaviest compressed children below it (while the child keeps >= rho of the parent's classes); one wave takes 16
// classes of the path's TOP node and walks the path bottom-up in registers — the edge product of one node is the next node's
// first factor, as in the pruning kernels — and only the top's rows are stored.  Children off the path are tops of their own
// paths (tables).  A path of k nodes costs k products per 16 top classes instead of one per 16 classes of each node (8 900
// against 5 075 tile products at the headline with rho = 0), and turns k hand-offs into one (launch 190 -> ~25 us).
//
// Device descriptors (int4 words in Shard::rep_desc).  Descriptor d: header words 2 d, 2 d + 1.
//   h0 = (first table row, classes U of the top node, path nodes | kind << 16, first node entry)
//   h1 = (offset in rep_map of the code list (kind 1) / of the first input's index map (kind 0; the others follow, `rows` entries each),
//         tiles, level, kind 1: matrix slot of the leaf's branch / kind 0: inputs of the path)
//   node entry  = (matrix slot of the node's branch, inputs, first input entry, flags), in the order of the walk (bottom of the path
//                 first; a side chain walked inline sits in front of the path node it hangs off).  flags 1: first node of an inline
//                 chain (park the running product, start from ones), 2: its last node (multiply the parked product back in behind
//                 the edge product).  The path's input entries follow its node entries, in the same order
//   input entry = (first row of the input's table or -1: ordinary leaf, matrix slot of an ordinary leaf,
//                  offset of the index map in rep_map (class of the TOP node -> row of the input / state of the leaf),
//                  descriptor of the input's table | its tiles << 16, or -1)
// kind 0: a path of compressed internal nodes, kind 1: leaf with ambiguity codes (table over its distinct codes).
// Work item = (descriptor or -1: padding, tile | rate class << 20, position in the launch's item sequence, queue index every head must
// have passed before all inputs of the item are sold).
constexpr int kRepQueues = 32;                       // (one word saturates at ~88 returning atomics per microsecond: 2 048 waves start at once)
// Every word the waves of the launch meet at — queue heads, exit counter, per (class, descriptor) finished tiles — sits 4 352 bytes
// from the next: device-scope atomics execute at the memory side, one channel serves ~88 of them per microsecond, and 64-byte
// spacing put all 32 heads and every counter on one channel (measured: 17 us per ticket with 2 048 waves, the CU's memory queue
// backed up behind them and the A-operand stream of the products with it).
constexpr int kRepHeadStride = 1088;                 // ints between the words
constexpr int kRepSyncExit = kRepQueues * kRepHeadStride;
constexpr int kRepSyncDone = kRepSyncExit + kRepHeadStride;  // per (class, descriptor) finished tiles from here, same spacing
constexpr int kRepStack = 24;                        // held tickets per wave (tree heights beyond that fall back to waiting)
constexpr int kRepMaxInputs = 64;                    // inputs of a path whose class indices are staged in LDS (longer paths: cut by the host)
constexpr int kRepAStages = 3;                       // A-operand chunks in flight ahead of the MFMAs of an edge product
constexpr int kRepBStages = 2;                       // (row-split walk) B-operand chunks on their way out of LDS ahead of them

struct RepArgs {
  const int4 *desc;
  const int4 *items;          // [8][qcap]
  int qcap;                   // items per queue (the same in every queue: levels are padded to multiples of kRepQueues)
  const int *live;            // [n_desc] 1: the table is recomputed by this pass (a child table that is not counts as finished)
  int n_desc;
  int *sync;
  double *tab;                // [classes][rows][DP]
  int32_t *cnt;               // [classes][rows]
  const int32_t *map;
  long long rows;             // table rows per class
  const double *Pfrag;        // class 0
  const double *PTg;
  size_t cs_P;
  const double *ambig;
  int n_waves;                // grid size
  int cat0;                   // first rate class of the pass
  int n_static;               // > 0: no item of this pass reads a table of this pass — the items (this many) are dealt to the waves by
                              // position (wave b: items b, b + n_waves, ..), nobody takes tickets, counts tiles or drains its stores
  long long *dbg;             // diagnostic (HYPHY_HIP_REP_TIMELINE): per wave [wall start, wall end, items, failed polls, shader cycles in
                              // tickets, waiting, gathers, product, publish, + first-item wall stamps] (16 words), or nullptr
};

namespace {

#define REP_TR(b)                          \
  if constexpr (TRACE) {                   \
    const long long n_ = clock64();        \
    tr[b] += n_ - tr_last, tr_last = n_;   \
  }
// (descriptors, items and live flags are __restrict__ kernel arguments of their own: read-only for the launch, they then come
//  through scalar loads; as members of the argument struct every descriptor word was a vector load + vmcnt(0) + readfirstlane —
//  four to five L2 round trips in series per path node)
template <int NW, bool TRACE = false>
__global__ __launch_bounds__(64, 2) void class_table_kernel(const int4 *__restrict__ desc, const int4 *__restrict__ items,
                                                            const int *__restrict__ live, RepArgs a) {
  [[maybe_unused]] long long tr[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  [[maybe_unused]] long long tr_last = 0;
  if constexpr (TRACE) {
    tr[0] = wall_clock64();
    tr_last = clock64();
  }
  constexpr int NKK = 4 * NW, DP = 16 * NW, TILE = NKK * 64;
  __shared__ int4 held[kRepStack];
  // a finished tile on its way out: [16 classes][DP + 2] (rows padded to keep the 16-byte writes of the 16 class lanes on
  // different banks) + the 16 exponents
  __shared__ __align__(16) double stage[16 * (DP + 2)];
  __shared__ __align__(16) int stage_cnt[16];
  __shared__ int sidx[kRepMaxInputs * 16];  // the class indices of a path's inputs, fetched at the item's start
  const int lane = threadIdx.x, g = lane >> 4, sl = lane & 15;
  // this wave's queue: four per XCD (HW_REG_XCC_ID)
  const int own = (((__builtin_amdgcn_s_getreg((31 << 11) | 20) & 7) << 2) | ((blockIdx.x >> 3) & 3)) & (kRepQueues - 1);
  auto uni = [](int4 v) -> int4 {  // (items are wave-uniform: keep them in SGPRs wherever they come from)
    return make_int4(__builtin_amdgcn_readfirstlane(v.x), __builtin_amdgcn_readfirstlane(v.y), __builtin_amdgcn_readfirstlane(v.z),
                     __builtin_amdgcn_readfirstlane(v.w));
  };
  int sp = 0;
  int sold_below = 0;  // every queue's head has been seen at or beyond this index: all tickets below it have holders
  bool own_empty = false;
  const f64x4 ones = (f64x4){1., 1., 1., 1.}, zeros = (f64x4){0., 0., 0., 0.};

  // one ticket from queue q: the item, or x = -2 when the queue is sold out
  auto ticket = [&](int q) -> int4 {
    int k = 0;
    if (lane == 0) k = __hip_atomic_fetch_add(a.sync + (size_t)q * kRepHeadStride, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    k = __builtin_amdgcn_readfirstlane(k);
    if (k >= a.qcap) return make_int4(-2, 0, 0, 0);
    return uni(items[(size_t)q * a.qcap + k]);
  };
  // the heads of all queues in one load: queues with tickets below index `bound` left (bit mask), and the lowest head
  auto heads = [&](int bound, int &hmin) -> unsigned {
    int h = 0x7fffffff;
    if (lane < kRepQueues) h = __hip_atomic_load(a.sync + (size_t)lane * kRepHeadStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned m = (unsigned)__ballot(lane < kRepQueues && h < bound && h < a.qcap);
    int mn = h;
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) mn = min(mn, __shfl_xor(mn, off));  // (lanes 0..31 hold the heads)
    hmin = __builtin_amdgcn_readfirstlane(mn);
    return m;
  };
  // next item: this wave's own queue, then whichever queue still has tickets (one look at all heads instead of 31 more atomics:
  // every wave of the launch comes through here when the queues run dry)
  auto pull = [&]() -> int4 {
    if (!own_empty) {
      const int4 it = ticket(own);
      if (it.x != -2) return it;
      own_empty = true;
    }
    for (;;) {
      int hmin;
      const unsigned m = heads(0x7fffffff, hmin);
      if (m == 0u) return make_int4(-2, 0, 0, 0);
      const unsigned rot = (m >> own) | (own ? m << (kRepQueues - own) : 0u);  // start looking behind the own queue
      const int4 it = ticket((own + __builtin_ctz(rot)) & (kRepQueues - 1));
      if (it.x != -2) return it;
    }
  };
  // held tickets, the earliest of the sequence (item.z) on top
  auto push = [&](const int4 &it) {
    int pos = sp;
    while (pos > 0 && held[pos - 1].z < it.z) {
      held[pos] = held[pos - 1];
      pos--;
    }
    held[pos] = it;
    sp++;
  };

  int spos = blockIdx.x, round = 0;  // (n_static: this wave's next item)
  auto next_item = [&]() -> int4 {
    if (a.n_static > 0) {
      if (spos >= a.n_static) return make_int4(-2, 0, 0, 0);
      const int4 it = items[(size_t)(spos % kRepQueues) * a.qcap + spos / kRepQueues];
      // (the items are in descending order of cost: back and forth over the waves, so that the wave with the longest first item
      //  gets the shortest second one)
      round++;
      spos = round * a.n_waves + ((round & 1) ? a.n_waves - 1 - (int)blockIdx.x : (int)blockIdx.x);
      return it;
    }
    return sp > 0 ? uni(held[--sp]) : pull();
  };
  int4 cur = next_item();
  REP_TR(4)
  while (cur.x != -2) {
    if (cur.x < 0) {  // padding behind the last item
      cur = next_item();
      continue;
    }
    // Never wait for work nobody holds: an item is only worked on once the tickets of all its inputs are sold (every head beyond
    // the item's bound; the heads only grow, so what was seen once stays true) — unsold earlier tickets are taken and run first.
    if (cur.w > sold_below) {
      int hmin;
      const unsigned m = heads(cur.w, hmin);
      sold_below = max(sold_below, hmin);
      if (m != 0u && sp < kRepStack - 1) {
        const int4 it = ticket(__builtin_ctz(m));
        if (it.x >= 0) {
          if (it.z < cur.z) {
            push(cur);
            cur = it;
          } else {
            push(it);  // (another wave was faster: a later ticket than this one — keep it for afterwards)
          }
        }
        REP_TR(5)
        continue;
      }
    }
    REP_TR(5)
    const int4 h0 = desc[2 * cur.x], h1 = desc[2 * cur.x + 1];
    const int n_nodes = h0.z & 0xffff, kind = h0.z >> 16;
    const int cat = a.cat0 + (cur.y >> 20);  // (items name their class RELATIVE to the pass's first: one list serves every class)
    int *done = a.sync + kRepSyncDone + (size_t)cat * a.n_desc * kRepHeadStride;
    // ---- one tile of 16 classes of the path's top node ----
    const int u0 = (cur.y & 0xfffff) * 16;
    const long long row = (long long)cat * a.rows + h0.x + u0;
    f64x4 acc[NW];  // conditionals of the node being assembled; behind its edge product: the next node's first factor
    int cnt = 0;
    // E = P x T for the branch in matrix slot `slot`: the A-operand image streamed from L2, T from `bsrc`
    // (kRepAStages chunks of the image in flight: a wave of this launch is mostly alone on its SIMD — 2 000 items on 1 024 SIMDs —
    //  and with one chunk ahead every k-step waited a full L2 round trip: 8 300 cycles per product against 2 048 of MFMA issue)
    auto product = [&](int slot, auto bsrc) {
      const double *Pf = a.Pfrag + (size_t)cat * a.cs_P + (size_t)slot * NW * TILE;  // uniform
      const __amdgpu_buffer_rsrc_t pfr = agent_rsrc(Pf);
      const unsigned lane16 = (unsigned)lane * 16u;
      constexpr int NS = NKK / 2, PF = kRepAStages < NS ? kRepAStages : NS;
      f64x4 D[NW];
#pragma unroll
      for (int w = 0; w < NW; w++) D[w] = zeros;
      f64x2 A[PF][NW];
#pragma unroll
      for (int st = 0; st < PF; st++)
#pragma unroll
        for (int w = 0; w < NW; w++) A[st][w] = ld16_buf(pfr, lane16, (unsigned)((w * TILE + st * 128) * 8));
#pragma unroll
      for (int k2 = 0; k2 < NS; k2++) {
        const f64x2 bc = bsrc(k2);
        f64x2 Ac[NW];
#pragma unroll
        for (int w = 0; w < NW; w++) Ac[w] = A[k2 % PF][w];
        asm volatile("" ::"v"(Ac[0]));  // (the chunk has arrived: its registers are free for the one PF steps ahead)
        if (k2 + PF < NS) {
#pragma unroll
          for (int w = 0; w < NW; w++) A[k2 % PF][w] = ld16_buf(pfr, lane16, (unsigned)((w * TILE + (k2 + PF) * 128) * 8));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int w = 0; w < NW; w++) D[w] = mfma(Ac[w][0], bc[0], D[w]);
#pragma unroll
        for (int w = 0; w < NW; w++) D[w] = mfma(Ac[w][1], bc[1], D[w]);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int w = 0; w < NW; w++) acc[w] = D[w];
    };
    if (kind == 0) {
      // walk the path bottom-up: the edge product of node k is node k + 1's first factor, straight from the registers (the MFMA's
      // C/D image is its B-operand image, common.h); everything else a node needs comes by class index from tables below the path
      // class indices of every input of the path (class of the top node -> row of the input's table / state of the leaf): one pass at
      // the start, four inputs per load instruction, parked in LDS
      // (a descriptor's index maps lie one behind the other, `rows` entries each, in input order)
      const int n_in = h1.w, map0 = h1.x, rows = h1.y * 16;
      for (int e0 = 0; e0 < n_in; e0 += 4) {
        const int e = e0 + g;
        if (e < n_in) sidx[e * 16 + sl] = a.map[map0 + e * rows + u0 + sl];
      }
      __syncthreads();
      // is the table behind input entry `ie` complete?  (one look; `block`: sleep until it is)
      auto table_ready = [&](const int4 &ie, bool block) -> bool {
        if (a.n_static > 0 || ie.w < 0 || !live[ie.w & 0xffff]) return true;  // (n_static: the tables below were finished by an earlier launch)
        const int *ctr = done + (size_t)(ie.w & 0xffff) * kRepHeadStride;
        for (;;) {
          int v = 0;
          if (lane == 0) v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (__builtin_amdgcn_readfirstlane(v) >= (ie.w >> 16)) break;
          if (!block) return false;
          __builtin_amdgcn_s_sleep(8);
          if constexpr (TRACE) tr[3]++;
        }
        asm volatile("" ::: "memory");
        return true;
      };
      auto gather = [&](const int4 &ie, int e, f64x2 (&v)[2 * NW], int &ec) {
        const int idx = sidx[e * 16 + sl];
        const double *src = ie.x >= 0 ? a.tab + ((size_t)cat * a.rows + ie.x) * DP
                                      : a.PTg + (size_t)cat * a.cs_P + (size_t)ie.y * DP * DP;  // uniform
        ec = 0;
        if (ie.x >= 0) {  // rows written by another wave of this launch (write-through): L1-bypassing loads
#pragma unroll
          for (int w = 0; w < NW; w++) {
            const unsigned off = (unsigned)((idx * NW + w) * 16 + g * 4) * 8u;
            v[2 * w] = ld16_agent(src, off), v[2 * w + 1] = ld16_agent(src, off + 16u);
          }
          ec = __hip_atomic_load(a.cnt + (size_t)cat * a.rows + ie.x + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {  // columns of a leaf's matrix (written by the exponential kernel): plain loads, L1 / L2 hits
#pragma unroll
          for (int w = 0; w < NW; w++) {
            const unsigned off = (unsigned)((idx * NW + w) * 16 + g * 4) * 8u;
            v[2 * w] = ld16(src, off), v[2 * w + 1] = ld16(src, off + 16u);
          }
        }
      };
      f64x2 pv[2 * NW];  // the first input of the NEXT node, requested before this node's edge product
      int pcnt = 0, park_cnt = 0;
      bool pf = false;
      int e_node = 0;    // first input (path-wide numbering) of the current node
      for (int k = 0; k < n_nodes; k++) {
        const int4 ne = desc[h0.w + k];  // (matrix slot of the node's branch, inputs, first input entry)
        if (k == 0) {
#pragma unroll
          for (int w = 0; w < NW; w++) acc[w] = ones;
        }
        if (ne.w & 1) {  // an inline side chain starts: the running product (ones in front of the walk's first node) waits in the wave's LDS tile
#pragma unroll
          for (int w = 0; w < NW; w++) {
            *reinterpret_cast<f64x2 *>(stage + ((2 * w) * 64 + lane) * 2) = (f64x2){acc[w][0], acc[w][1]};
            *reinterpret_cast<f64x2 *>(stage + ((2 * w + 1) * 64 + lane) * 2) = (f64x2){acc[w][2], acc[w][3]};
          }
          park_cnt = cnt;
          cnt = 0;
#pragma unroll
          for (int w = 0; w < NW; w++) acc[w] = ones;
        }
        for (int j = 0; j < ne.y; j++) {
          if (j == 0 && pf) {
#pragma unroll
            for (int w = 0; w < NW; w++) acc[w] *= (f64x4){pv[2 * w][0], pv[2 * w][1], pv[2 * w + 1][0], pv[2 * w + 1][1]};
            cnt += pcnt;
            continue;
          }
          const int4 ie = desc[ne.z + j];  // uniform
          if (!table_ready(ie, true)) {}
          REP_TR(5)
          f64x2 v[2 * NW];
          int ec;
          gather(ie, e_node + j, v, ec);
#pragma unroll
          for (int w = 0; w < NW; w++) acc[w] *= (f64x4){v[2 * w][0], v[2 * w][1], v[2 * w + 1][0], v[2 * w + 1][1]};
          cnt += ec;
        }
        e_node += ne.y;
        // the node's conditionals of these classes: per-class 2^64 rescale (tested at every compressed node)
        double s = 0.;
#pragma unroll
        for (int w = 0; w < NW; w++) s += (acc[w][0] + acc[w][1]) + (acc[w][2] + acc[w][3]);
        const double tot = row_sum4(s);
        double sc = 1.0;
        int m = 0;
        if (__any(!(tot >= kScalerThreshold && tot <= kScalerUp))) m = rescale_decision(tot, sc);  // rare
        cnt += m;
#pragma unroll
        for (int w = 0; w < NW; w++) acc[w] *= sc;
        if constexpr (TRACE) asm volatile("" ::"v"(acc[0][0]));
        REP_TR(6)
        pf = false;
        if (k + 1 < n_nodes) {  // the next node's first input rides under this node's product (if its table is complete already)
          const int4 ne2 = desc[h0.w + k + 1];
          if (ne2.y > 0) {
            const int4 ie2 = desc[ne2.z];
            if (table_ready(ie2, false)) {
              gather(ie2, e_node, pv, pcnt);
              pf = true;
            }
          }
        }
        product(ne.x, [&](int k2) -> f64x2 { return (f64x2){acc[k2 >> 1][(k2 & 1) * 2], acc[k2 >> 1][(k2 & 1) * 2 + 1]}; });
        if (ne.w & 2) {  // the chain's edge product joins the product it was walked for
#pragma unroll
          for (int w = 0; w < NW; w++) {
            const f64x2 p0 = *reinterpret_cast<const f64x2 *>(stage + ((2 * w) * 64 + lane) * 2);
            const f64x2 p1 = *reinterpret_cast<const f64x2 *>(stage + ((2 * w + 1) * 64 + lane) * 2);
            acc[w] *= (f64x4){p0[0], p0[1], p1[0], p1[1]};
          }
          cnt += park_cnt;
        }
        if constexpr (TRACE) asm volatile("" ::"v"(acc[0][0]));
        REP_TR(7)
      }
    } else {
      // a leaf with ambiguity codes: E[u] = P_leaf x (resolution vector of the u-th distinct code)
      const int code = a.map[h1.x + u0 + sl];
      const double *av = a.ambig + (size_t)(code < 0 ? -code - 1 : 0) * DP;
      product(h1.w, [&](int k2) -> f64x2 {
        f64x2 b;
        b[0] = (code >= 0) ? ((8 * k2 + g == code) ? 1.0 : 0.0) : av[8 * k2 + g];
        b[1] = (code >= 0) ? ((8 * k2 + 4 + g == code) ? 1.0 : 0.0) : av[8 * k2 + 4 + g];
        return b;
      });
      if constexpr (TRACE) asm volatile("" ::"v"(acc[0][0]));
      REP_TR(7)
    }
    // ---- publish the rows ([class][w][g][r] = E[16 w + 4 r + g][class]) and their exponents, then the tile ----
    // Through LDS: a lane holds 4 doubles of each of NW row blocks of ONE class; written from the registers, a store
    // instruction scatters 64 half-sectors over 16 lines.  Transposed through the wave's LDS tile, every store instruction
    // writes 1 KiB of consecutive bytes (write-through stores go to the fabric as they are).
    {
      double *out = a.tab + (size_t)row * DP;  // uniform
#pragma unroll
      for (int w = 0; w < NW; w++) {
        *reinterpret_cast<f64x2 *>(stage + sl * (DP + 2) + w * 16 + g * 4) = (f64x2){acc[w][0], acc[w][1]};
        *reinterpret_cast<f64x2 *>(stage + sl * (DP + 2) + w * 16 + g * 4 + 2) = (f64x2){acc[w][2], acc[w][3]};
      }
      if (g == 0) stage_cnt[sl] = cnt;
      __syncthreads();  // (one wave per workgroup: orders the LDS round trip)
#pragma unroll
      for (int i = 0; i < 2 * NW; i++) {
        const int e = i * 128 + lane * 2, r = e / DP, c = e % DP;
        st16_agent(out, (unsigned)e * 8u, *reinterpret_cast<const f64x2 *>(stage + r * (DP + 2) + c));
      }
      if (lane < 4) {
        const int4 cv = *reinterpret_cast<const int4 *>(stage_cnt + 4 * lane);
        u32x4_t v;
        __builtin_memcpy(&v, &cv, 16);
        __builtin_amdgcn_raw_buffer_store_b128(v, agent_rsrc(reinterpret_cast<const double *>(a.cnt + row)), (unsigned)lane * 16u, 0, 16);
      }
      __syncthreads();  // (the tile is free for the next item)
      if (a.n_static == 0) {  // (somebody in this launch may be waiting for the table)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(done + (size_t)cur.x * kRepHeadStride, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    REP_TR(8)
    if constexpr (TRACE) tr[2]++;
    cur = next_item();
    REP_TR(4)
  }
  if constexpr (TRACE) {
    if (a.dbg && lane == 0) {
      tr[1] = wall_clock64();
      for (int i = 0; i < 16; i++) a.dbg[(size_t)blockIdx.x * 16 + i] = tr[i];
    }
  }
  // (queue heads and counters are reset by the launch that follows on the stream: the trunk's pruning kernel, prune.hip)
}

// The ROW-SPLIT walk (r06): the same item — 16 classes of a path's top node, walked bottom-up — by a WORKGROUP of NW waves, wave w
// owning rows 16 w .. 16 w + 15 of every node on the way: 4 NW matrix instructions per edge product and wave instead of 4 NW^2, its
// share of every gathered input (two 16-byte loads) instead of the whole row, its row block of the A-operand stream.  What the
// one-wave walk kept in registers between two nodes — the product's C/D image as the next product's B operand — crosses an LDS tile
// here: every wave leaves its four k-steps (two 16-byte writes per lane, conflict-free), ONE barrier, every wave reads the whole
// B image back (2 NW 16-byte reads per lane).  The per-class sum of the rescale test travels the same way (partial row sums of the
// NW row blocks, added in a fixed order by everybody; the power of 2^64 is applied to B as it leaves LDS — exact, so any wave
// agreeing on it is all that matters).  Why: the one-wave walk is a chain of 4-6 products of 64 dependent matrix instructions with
// the gathers in between — 36-50 k cycles — and 1 205 of them on 1 024 SIMDs end with 181 SIMDs running two (30 us of launch for
// 10 us of matrix-pipe work).  A team's walk is a chain a quarter as long in matrix time, registers drop from 256 to < 128 per wave,
// so four to five teams share a CU and every SIMD interleaves the products of several walks.  One item per workgroup, workgroups in
// descending order of estimated cost: the dispatcher hands the next item to whichever CU frees a slot (longest-processing-time-first
// without a host-side placement).  Static passes only (no item reads a table of its own launch: the production form since the
// per-level launches of r05); the ticket protocol stays with class_table_kernel.
template <int NW, bool TRACE = false, int AST = kRepAStages, int BST = kRepBStages, int OCC = 6, int CHAINS = 1>
__global__ __launch_bounds__(64 * NW, OCC) void class_table_team_kernel(const int4 *__restrict__ desc, const int4 *__restrict__ items, RepArgs a) {
  [[maybe_unused]] long long tr[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  [[maybe_unused]] long long tr_last = 0;
  if constexpr (TRACE) {
    tr[0] = wall_clock64();
    tr_last = clock64();
  }
  constexpr int NKK = 4 * NW, DP = 16 * NW, TILE = NKK * 64, NS = NKK / 2;
  constexpr int PF = AST < NS ? AST : NS;
  // [2][TILE]: the B-operand image of the node being handed over (double-buffered: a wave may write node k + 1 while another still
  // reads node k); behind the last product the same bytes carry the finished tile on its way out ([16 classes][DP + 2])
  __shared__ __align__(16) double bx[2 * TILE + 64];
  __shared__ double psum[2][NW][16];
  __shared__ __align__(16) int stage_cnt[16];
  __shared__ int sidx[kRepMaxInputs * 16];
  static_assert(16 * (DP + 2) <= 2 * TILE + 64, "the outgoing tile fits the exchange buffers");
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, sl = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pos = blockIdx.x;
  const int4 cur = items[(size_t)(pos % kRepQueues) * a.qcap + pos / kRepQueues];
  if (cur.x < 0) return;  // padding
  const int4 h0 = desc[2 * cur.x], h1 = desc[2 * cur.x + 1];
  const int n_nodes = h0.z & 0xffff, kind = h0.z >> 16;
  const int cat = a.cat0 + (cur.y >> 20);
  const int u0 = (cur.y & 0xfffff) * 16;
  const long long row = (long long)cat * a.rows + h0.x + u0;
  const f64x4 ones = (f64x4){1., 1., 1., 1.}, zeros = (f64x4){0., 0., 0., 0.};
  const unsigned lane16 = (unsigned)lane * 16u;
  f64x4 acc = ones;  // this wave's rows of the node being assembled; behind its edge product: of the next node's first factor
  int cnt = 0;
  // this wave's rows of E = P x B for the branch in matrix slot `slot`; B: the whole operand, k2 -> k-steps 2 k2, 2 k2 + 1
  auto a_rsrc = [&](int slot) { return agent_rsrc(a.Pfrag + (size_t)cat * a.cs_P + ((size_t)slot * NW + w) * TILE); };
  auto a_first = [&](__amdgpu_buffer_rsrc_t pfr, f64x2 (&A)[PF]) {
#pragma unroll
    for (int st = 0; st < PF; st++) A[st] = ld16_buf(pfr, lane16, (unsigned)(st * 128 * 8));
  };
  // (B through `bsrc`: the operand leaves LDS a few k-steps ahead of the matrix instructions that read it — kRepBStages chunks in
  //  flight — instead of standing in 4 NKK registers: 108 -> under 96 registers, five teams per CU instead of four)
  auto product = [&](__amdgpu_buffer_rsrc_t pfr, f64x2 (&A)[PF], auto bsrc) {
    constexpr int PB = BST < NS ? BST : NS;
    f64x4 D0 = zeros, D1 = zeros;  // (two accumulator chains: even / odd k-steps)
    f64x2 Bq[PB];
#pragma unroll
    for (int st = 0; st < PB; st++) Bq[st] = bsrc(st);
#pragma unroll
    for (int k2 = 0; k2 < NS; k2++) {
      const f64x2 Ac = A[k2 % PF], Bc = Bq[k2 % PB];
      asm volatile("" ::"v"(Ac), "v"(Bc));
      if (k2 + PF < NS) A[k2 % PF] = ld16_buf(pfr, lane16, (unsigned)((k2 + PF) * 128 * 8));
      if (k2 + PB < NS) Bq[k2 % PB] = bsrc(k2 + PB);
      D0 = mfma(Ac[0], Bc[0], D0);
      if constexpr (CHAINS == 2) D1 = mfma(Ac[1], Bc[1], D1);
      else D0 = mfma(Ac[1], Bc[1], D0);
    }
    if constexpr (CHAINS == 2) acc = D0 + D1;
    else acc = D0;
  };
  if (kind == 0) {
    const int n_in = h1.w, map0 = h1.x, rows = h1.y * 16;
    for (int e0 = 0; e0 < n_in; e0 += 4 * NW) {
      const int e = e0 + (tid >> 4);
      if (e < n_in) sidx[e * 16 + sl] = a.map[map0 + e * rows + u0 + sl];
    }
    __syncthreads();
    REP_TR(4)
    // this wave's share of an input's rows (the tables of a static pass were finished by an earlier launch)
    auto gather = [&](const int4 &ie, int e, f64x2 (&v)[2], int &ec) {
      const int idx = sidx[e * 16 + sl];
      const unsigned off = (unsigned)((idx * NW + w) * 16 + g * 4) * 8u;
      ec = 0;
      if (ie.x >= 0) {
        const double *src = a.tab + ((size_t)cat * a.rows + ie.x) * DP;  // uniform
        v[0] = ld16_agent(src, off), v[1] = ld16_agent(src, off + 16u);
        ec = __hip_atomic_load(a.cnt + (size_t)cat * a.rows + ie.x + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        const double *src = a.PTg + (size_t)cat * a.cs_P + (size_t)ie.y * DP * DP;  // uniform
        v[0] = ld16(src, off), v[1] = ld16(src, off + 16u);
      }
    };
    f64x2 pv[2];  // the first input of the NEXT node, requested before this node's edge product
    int pcnt = 0, park_cnt = 0;
    f64x4 parked = ones;  // an inline side chain is being walked: the running product waits here
    bool pf = false;
    int e_node = 0;
    for (int k = 0; k < n_nodes; k++) {
      const int4 ne = desc[h0.w + k];
      if (ne.w & 1) {
        parked = acc;
        park_cnt = cnt;
        cnt = 0;
        acc = ones;
      }
      for (int j = 0; j < ne.y; j++) {
        if (j == 0 && pf) {
          acc *= (f64x4){pv[0][0], pv[0][1], pv[1][0], pv[1][1]};
          cnt += pcnt;
          continue;
        }
        const int4 ie = desc[ne.z + j];  // uniform
        f64x2 v[2];
        int ec;
        gather(ie, e_node + j, v, ec);
        acc *= (f64x4){v[0][0], v[0][1], v[1][0], v[1][1]};
        cnt += ec;
      }
      e_node += ne.y;
      if constexpr (TRACE) asm volatile("" ::"v"(acc[0]));
      REP_TR(5)
      // hand the node over: this wave's k-steps 4 w .. 4 w + 3 of the B image and its share of the per-class totals
      double *bb = bx + (k & 1) * TILE;
      *reinterpret_cast<f64x2 *>(bb + ((2 * w) * 64 + lane) * 2) = (f64x2){acc[0], acc[1]};
      *reinterpret_cast<f64x2 *>(bb + ((2 * w + 1) * 64 + lane) * 2) = (f64x2){acc[2], acc[3]};
      const double ps = row_sum4((acc[0] + acc[1]) + (acc[2] + acc[3]));
      if (g == 0) psum[k & 1][w][sl] = ps;
      // in flight across the barrier: the head of this product's A stream and the next node's first input
      const __amdgpu_buffer_rsrc_t pfr = a_rsrc(ne.x);
      f64x2 A[PF];
      a_first(pfr, A);
      pf = false;
      if (k + 1 < n_nodes) {
        const int4 ne2 = desc[h0.w + k + 1];
        if (ne2.y > 0) {
          gather(desc[ne2.z], e_node, pv, pcnt);
          pf = true;
        }
      }
      lds_barrier();  // (not __syncthreads(): that would also wait for the A chunks and the gather just requested; measured: no difference)
      double tot = psum[k & 1][0][sl];
#pragma unroll
      for (int ww = 1; ww < NW; ww++) tot += psum[k & 1][ww][sl];
      if constexpr (TRACE) asm volatile("" ::"v"(tot));
      REP_TR(6)
      product(pfr, A, [&](int k2) -> f64x2 { return *reinterpret_cast<const f64x2 *>(bb + (k2 * 64 + lane) * 2); });
      if (__any(!(tot >= kScalerThreshold && tot <= kScalerUp))) {  // rare: some class needs (or cannot have) a rescale
        double sc = 1.0;                                            // (behind the product: a power of 2^64 commutes with it exactly)
        cnt += rescale_decision(tot, sc);
        acc *= sc;
      }
      if (ne.w & 2) {  // the chain's edge product joins the product it was walked for
        acc *= parked;
        cnt += park_cnt;
      }
      if constexpr (TRACE) asm volatile("" ::"v"(acc[0]));
      REP_TR(7)
    }
  } else {
    // a leaf with ambiguity codes: E[u] = P_leaf x (resolution vector of the u-th distinct code)
    const int code = a.map[h1.x + u0 + sl];
    const double *av = a.ambig + (size_t)(code < 0 ? -code - 1 : 0) * DP;
    const __amdgpu_buffer_rsrc_t pfr = a_rsrc(h1.w);
    f64x2 A[PF];
    a_first(pfr, A);
    product(pfr, A, [&](int k2) -> f64x2 {
      f64x2 b;
      b[0] = (code >= 0) ? ((8 * k2 + g == code) ? 1.0 : 0.0) : av[8 * k2 + g];
      b[1] = (code >= 0) ? ((8 * k2 + 4 + g == code) ? 1.0 : 0.0) : av[8 * k2 + 4 + g];
      return b;
    });
  }
  // ---- the rows ([class][w][g][r] = E[16 w + 4 r + g][class]) and their exponents, through the LDS tile: every store
  //      instruction writes 1 KiB of consecutive bytes ----
  __syncthreads();  // (nobody reads the exchange buffers any more)
  double *stage = bx;
  *reinterpret_cast<f64x2 *>(stage + sl * (DP + 2) + w * 16 + g * 4) = (f64x2){acc[0], acc[1]};
  *reinterpret_cast<f64x2 *>(stage + sl * (DP + 2) + w * 16 + g * 4 + 2) = (f64x2){acc[2], acc[3]};
  if (tid < 16) stage_cnt[tid] = cnt;  // (wave 0, g = 0: lane = class)
  __syncthreads();
  double *out = a.tab + (size_t)row * DP;  // uniform
#pragma unroll
  for (int i2 = 0; i2 < 2; i2++) {
    const int e = (2 * w + i2) * 128 + lane * 2, r = e / DP, c = e % DP;
    st16_agent(out, (unsigned)e * 8u, *reinterpret_cast<const f64x2 *>(stage + r * (DP + 2) + c));
  }
  if (tid < 4) {
    const int4 cv = *reinterpret_cast<const int4 *>(stage_cnt + 4 * tid);
    u32x4_t v;
    __builtin_memcpy(&v, &cv, 16);
    __builtin_amdgcn_raw_buffer_store_b128(v, agent_rsrc(reinterpret_cast<const double *>(a.cnt + row)), (unsigned)tid * 16u, 0, 16);
  }
  if constexpr (TRACE) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    REP_TR(8)
    if (a.dbg && tid == 0) {  // one record per workgroup (wave 0): wall start / end, walked nodes, where it ran, cycles per phase
      tr[1] = wall_clock64();
      tr[2] = n_nodes;
      tr[3] = __builtin_amdgcn_s_getreg((31 << 11) | 4) | ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
      for (int i = 0; i < 16; i++) a.dbg[(size_t)blockIdx.x * 16 + i] = tr[i];
    }
  }
}


// The TRUNK under the row-split walk (r06): the rest of the tree — every node above the class tables, one value per PATTERN — walked by
// the same workgroup of NW waves, one tile of 16 patterns each.  The wave-per-tile trunk (prune_wave_kernel<.., REP>) gives a tile three
// waves that each issue 64 matrix instructions per edge product from one SIMD; here a product is split over the CU's four SIMDs and
// crosses the LDS tile, exactly as in the class tables' walk.  A trunk is a tree, not a path: a node's second internal child starts a
// new chain while the product walked so far waits — the inline side chains of the class tables (flags 1 / 2) nested, the waiting
// products on a small register stack (kWalkDepth; deeper trunks keep the pruning kernels).  Generalised leaves are gathered exactly as
// the pruning kernels' REP builds do (class id of the pattern -> row of the leaf's table, or the pattern's state -> column of the
// leaf's matrix).  The root has no edge product: its conditionals meet the root frequencies, and wave 0 leaves the tile's per-pattern
// values and its share of sum f log L (tree_evaluator.cpp:4046-4128, likefunc.cpp:11123) where the reduction kernel expects them.
// TWO CHAINS per tile (grid.z) where tiles are scarce (the headline: 624 tiles on 256 CUs): the subtrees below the root are dealt to
// two workgroups; each ends with the product of its subtrees' edge products, leaves it in memory (16-byte sc1 stores -> s_waitcnt
// vmcnt(0) -> relaxed agent RMW on the tile's counter: the protocol of the pruning kernels' joins), and the one that arrives LAST
// multiplies the other's in and finishes the root (two factors: the order of arrival does not change a bit).  Nobody waits.
// Nothing is persisted: the launch serves full passes under lazy persistence (api.hip); everything else runs the pruning kernels on
// the same view.
// Program (int4 words, per shard; absolute word indices): [0] = (first word of the one-chain form, of the two-chain form or 0, 0, 0).
// A form: (chains, 0, 0, 0), then per chain (first node word, nodes, 0, 0); node = (matrix slot of the node's branch or -1: the chain's
// end at the root, leaf children, first input word, flags); input = (view leaf, first row of its table or -1: ordinary leaf, first
// exponent row / matrix slot, 0).
struct TrunkWalkArgs {
  const double *Pfrag, *PTg;  // class 0 (blockIdx.y = rate class)
  size_t cs_P;
  const double *gtab;         // class tables, class 0
  const int32_t *gcnt;
  size_t cs_gtab, cs_gcnt;
  const int16_t *codes_tile;  // [tile][view leaves][16]
  int L;                      // view leaves
  int form;                   // first word of the form in use
  const double *pi;           // [DP]
  double *site_lik;
  int32_t *site_cnt;
  size_t cs_site;
  const double *freq;
  double *wg_sum;
  long long *wg_cnt;
  int *wg_flag;
  size_t cs_wg;
  // two chains: products handed over at the root
  double *deposits;           // [class][chain][tile][TILE]
  size_t cs_deposits;
  int32_t *hand_cnt;          // [class][chain][tile][16]
  int *arrivals;              // [class][tile], zero between launches
  int ntiles;
  // fused final combine (FUSE builds, red_out != nullptr): the workgroup that finishes the launch's LAST root sums the per-tile
  // partial sums and publishes the result record (prune.hip: publish_partial) — no reduction kernel behind the launch
  double *red_out, *red_rec;
  const int *red_status;
  double red_seq;
  int *red_done;
  int red_n;
  long long *dbg;             // diagnostic (HYPHY_HIP_WALK_TIMELINE): the lower phase's record per workgroup, or nullptr
};
constexpr int kWalkDepth = 3;

// (AST: A chunks ahead of a product, DEPTH: waiting products, OCC: workgroups per CU the build allows — one production form: 3 / 3 / 5)
template <int NW, bool TRACE = false, bool FUSE = false, int AST = kRepAStages, int DEPTH = kWalkDepth, int OCC = 5, int BST = kRepBStages>
__global__ __launch_bounds__(64 * NW, OCC) void trunk_walk_kernel(const int4 *__restrict__ walk, TrunkWalkArgs a) {
  [[maybe_unused]] long long tr[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  [[maybe_unused]] long long tr_last = 0;
  if constexpr (TRACE) {
    tr[0] = wall_clock64();
    tr_last = clock64();
  }
  constexpr int NKK = 4 * NW, DP = 16 * NW, TILE = NKK * 64, NS = NKK / 2;
  constexpr int PF = AST < NS ? AST : NS;
  __shared__ __align__(16) double bx[2 * TILE];
  __shared__ double psum[2][NW][16];
  __shared__ double epi[NW][16];
  __shared__ int arrived;
  extern __shared__ __align__(16) int16_t walk_codes[];  // [view leaves][16]
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, sl = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = blockIdx.x, chain = blockIdx.z;
  const size_t cat = blockIdx.y;
  {
    const int *src = reinterpret_cast<const int *>(a.codes_tile + (size_t)tile * a.L * 16);
    int *dst = reinterpret_cast<int *>(walk_codes);
    for (int i = tid; i < a.L * 8; i += 64 * NW) dst[i] = src[i];
  }
  const int n_chains = walk[a.form].x;
  const int4 ch = walk[a.form + 1 + chain];
  const int node0 = ch.x, n_nodes = ch.y;
  if (ch.z == 3) __builtin_amdgcn_s_setprio(3);  // (the longer of two chains: its instructions first wherever the SIMDs are shared)
  const f64x4 ones = (f64x4){1., 1., 1., 1.}, zeros = (f64x4){0., 0., 0., 0.};
  const unsigned lane16 = (unsigned)lane * 16u;
  const double *const Pfrag = a.Pfrag + cat * a.cs_P, *const PTg = a.PTg + cat * a.cs_P, *const gtab = a.gtab + cat * a.cs_gtab;
  const int32_t *const gcnt = a.gcnt + cat * a.cs_gcnt;
  f64x4 acc = ones;
  int cnt = 0;
  auto a_rsrc = [&](int slot) { return agent_rsrc(Pfrag + ((size_t)slot * NW + w) * TILE); };
  auto a_first = [&](__amdgpu_buffer_rsrc_t pfr, f64x2 (&A)[PF]) {
#pragma unroll
    for (int st = 0; st < PF; st++) A[st] = ld16_buf(pfr, lane16, (unsigned)(st * 128 * 8));
  };
  auto product = [&](__amdgpu_buffer_rsrc_t pfr, f64x2 (&A)[PF], const double *bb) {
    constexpr int PB = BST < NS ? BST : NS;
    f64x4 D0 = zeros;
    f64x2 Bq[PB];
#pragma unroll
    for (int st = 0; st < PB; st++) Bq[st] = *reinterpret_cast<const f64x2 *>(bb + (st * 64 + lane) * 2);
#pragma unroll
    for (int k2 = 0; k2 < NS; k2++) {
      const f64x2 Ac = A[k2 % PF], Bc = Bq[k2 % PB];
      asm volatile("" ::"v"(Ac), "v"(Bc));
      if (k2 + PF < NS) A[k2 % PF] = ld16_buf(pfr, lane16, (unsigned)((k2 + PF) * 128 * 8));
      if (k2 + PB < NS) Bq[k2 % PB] = *reinterpret_cast<const f64x2 *>(bb + ((k2 + PB) * 64 + lane) * 2);
      D0 = mfma(Ac[0], Bc[0], D0);
      D0 = mfma(Ac[1], Bc[1], D0);
    }
    acc = D0;
  };
  __syncthreads();
  REP_TR(4)
  // this wave's share of a generalised leaf's columns for the tile's patterns, and their 2^64 exponents
  auto gather = [&](const int4 &in, f64x2 (&v)[2], int &ec) {
    const int idx = (int)walk_codes[in.x * 16 + sl];
    const unsigned off = (unsigned)((idx * NW + w) * 16 + g * 4) * 8u;
    ec = 0;
    if (in.y >= 0) {
      const double *src = gtab + (size_t)in.y * DP;  // uniform
      v[0] = ld16(src, off), v[1] = ld16(src, off + 16u);
      ec = gcnt[in.z + idx];
    } else {
      const double *src = PTg + (size_t)in.z * DP * DP;  // uniform
      v[0] = ld16(src, off), v[1] = ld16(src, off + 16u);
    }
  };
  f64x2 pv[2];  // the first input of the NEXT node, requested before this node's edge product
  int pcnt = 0;
  bool pf = false;
  f64x4 pk0 = ones, pk1 = ones, pk2 = ones;  // products waiting for the chain walked for them
  int pc0 = 0, pc1 = 0, pc2 = 0, depth = 0;
  for (int k = 0; k < n_nodes; k++) {
    const int4 ne = walk[node0 + k];
    if (ne.w & 1) {
      if (DEPTH == 1 || depth == 0) pk0 = acc, pc0 = cnt;
      else if (DEPTH == 2 || depth == 1) pk1 = acc, pc1 = cnt;
      else pk2 = acc, pc2 = cnt;
      depth++;
      cnt = 0;
      acc = ones;
    }
    for (int j = 0; j < ne.y; j++) {
      if (j == 0 && pf) {
        acc *= (f64x4){pv[0][0], pv[0][1], pv[1][0], pv[1][1]};
        cnt += pcnt;
        continue;
      }
      f64x2 v[2];
      int ec;
      gather(walk[ne.z + j], v, ec);
      acc *= (f64x4){v[0][0], v[0][1], v[1][0], v[1][1]};
      cnt += ec;
    }
    if constexpr (TRACE) asm volatile("" ::"v"(acc[0]));
    REP_TR(5)
    if (ne.x < 0) break;  // the root: this chain's share of its conditionals is complete
    double *bb = bx + (k & 1) * TILE;
    *reinterpret_cast<f64x2 *>(bb + ((2 * w) * 64 + lane) * 2) = (f64x2){acc[0], acc[1]};
    *reinterpret_cast<f64x2 *>(bb + ((2 * w + 1) * 64 + lane) * 2) = (f64x2){acc[2], acc[3]};
    const double ps = row_sum4((acc[0] + acc[1]) + (acc[2] + acc[3]));
    if (g == 0) psum[k & 1][w][sl] = ps;
    const __amdgpu_buffer_rsrc_t pfr = a_rsrc(ne.x);
    f64x2 A[PF];
    a_first(pfr, A);
    pf = false;
    {
      const int4 ne2 = walk[node0 + k + 1];  // (the chain's end at the root follows every other node)
      if (ne2.y > 0) {
        gather(walk[ne2.z], pv, pcnt);
        pf = true;
      }
    }
    lds_barrier();  // (not __syncthreads(): that would also wait for the A chunks and the columns just requested)
    double tot = psum[k & 1][0][sl];
#pragma unroll
    for (int ww = 1; ww < NW; ww++) tot += psum[k & 1][ww][sl];
    if constexpr (TRACE) asm volatile("" ::"v"(tot));
    REP_TR(6)
    product(pfr, A, bb);
    if (__any(!(tot >= kScalerThreshold && tot <= kScalerUp))) {  // rare: some pattern needs (or cannot have) a rescale
      double sc = 1.0;                                            // (behind the product: a power of 2^64 commutes with it exactly)
      cnt += rescale_decision(tot, sc);
      acc *= sc;
    }
    if (ne.w & 2) {  // the chain's edge product joins the product it was walked for
      depth--;
      if (DEPTH == 1 || depth == 0) acc *= pk0, cnt += pc0;
      else if (DEPTH == 2 || depth == 1) acc *= pk1, cnt += pc1;
      else acc *= pk2, cnt += pc2;
    }
    if constexpr (TRACE) asm volatile("" ::"v"(acc[0]));
    REP_TR(7)
  }
  if (n_chains > 1) {
    // ---- two chains: leave this chain's product, arrive; the last arriver takes the other's and goes on to the root ----
    const size_t slot = (cat * 2 + chain) * (size_t)a.ntiles + tile, other = (cat * 2 + (1 - chain)) * (size_t)a.ntiles + tile;
    double *dst = a.deposits + slot * TILE;
    st16_agent(dst, (unsigned)((2 * w) * 64 + lane) * 16u, (f64x2){acc[0], acc[1]});
    st16_agent(dst, (unsigned)((2 * w + 1) * 64 + lane) * 16u, (f64x2){acc[2], acc[3]});
    if (w == 0 && g == 0) __hip_atomic_store(a.hand_cnt + slot * 16 + sl, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) arrived = __hip_atomic_fetch_add(a.arrivals + cat * a.ntiles + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (arrived == 0) {
      if constexpr (TRACE) {
        if (a.dbg && tid == 0) {
          tr[1] = wall_clock64();
          tr[2] = n_nodes;
          tr[3] = __builtin_amdgcn_s_getreg((31 << 11) | 4) | ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
          for (int i = 0; i < 16; i++) a.dbg[(((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 + i] = tr[i];
        }
      }
      return;
    }
    if (tid == 0) __hip_atomic_store(a.arrivals + cat * a.ntiles + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next launch
    const double *src = a.deposits + other * TILE;
    const f64x2 o0 = ld16_agent(src, (unsigned)((2 * w) * 64 + lane) * 16u), o1 = ld16_agent(src, (unsigned)((2 * w + 1) * 64 + lane) * 16u);
    cnt += __hip_atomic_load(a.hand_cnt + other * 16 + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    acc *= (f64x4){o0[0], o0[1], o1[0], o1[1]};
  }
  // ---- root: L_s = sum_k root[k][s] pi[k] over the NW row blocks (fixed order), then the tile's share of the log-likelihood ----
  {
    double pr = 0.;
#pragma unroll
    for (int r = 0; r < 4; r++) pr = fma(acc[r], a.pi[16 * w + 4 * r + g], pr);
    pr = row_sum4(pr);
    if (g == 0) epi[w][sl] = pr;
  }
  __syncthreads();
  if (w == 0) {
    double s = epi[0][sl];
#pragma unroll
    for (int ww = 1; ww < NW; ww++) s += epi[ww][sl];
    double wsum = 0.;
    long long wcnt = 0;
    int wflag = 0;
    if (g == 0) {
      const size_t site = (size_t)tile * 16 + sl;
      a.site_lik[cat * a.cs_site + site] = s;
      a.site_cnt[cat * a.cs_site + site] = cnt;
      const double f = a.freq[site];
      if (f != 0.) {
        if (s != s || isinf(s)) wflag |= 2;
        else if (s <= 0.) wflag |= 1;
        else {
          wsum = log(s) * f;
          wcnt = (long long)cnt * (long long)f;
        }
      }
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {  // fixed-order butterfly over the 16 pattern lanes (lanes >= 16 hold zeros)
      wsum += __shfl_xor(wsum, off);
      wcnt += __shfl_xor(wcnt, off);
      wflag |= __shfl_xor(wflag, off);
    }
    if (FUSE && a.red_out != nullptr) {
      if (lane == 0) {
        __hip_atomic_store(a.wg_sum + tile, wsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.wg_cnt + tile, wcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.wg_flag + tile, wflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (payload written through before the arrival: the protocol of the joins)
      int old = 0;
      if (lane == 0) old = __hip_atomic_fetch_add(a.red_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      old = __builtin_amdgcn_readfirstlane(old);
      asm volatile("" ::: "memory");
      if (old + 1 == a.red_n) {
        if (lane == 0) __hip_atomic_store(a.red_done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next launch
        combine_partials(a.wg_sum, a.wg_cnt, a.wg_flag, a.red_n, a.red_out, a.red_rec, a.red_status, a.red_seq, lane);
      }
    } else if (lane == 0) {
      a.wg_sum[cat * a.cs_wg + tile] = wsum;
      a.wg_cnt[cat * a.cs_wg + tile] = wcnt;
      a.wg_flag[cat * a.cs_wg + tile] = wflag;
    }
  }
  if constexpr (TRACE) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    REP_TR(8)
    if (a.dbg && tid == 0) {
      tr[1] = wall_clock64();
      tr[2] = n_nodes;
      tr[3] = __builtin_amdgcn_s_getreg((31 << 11) | 4) | ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
      for (int i = 0; i < 16; i++) a.dbg[(((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 + i] = tr[i];
    }
  }
}

}  // namespace

// ---- 4 states ----------------------------------------------------------------------------------------------------------
// A conditional vector is 32 bytes there and an edge product 16 multiply-adds: a compressed subtree is not cut into paths and
// tables but walked WHOLE, in post-order, by one thread per class of its root — inputs are leaf columns only (by the state the
// class's representative pattern has at the leaf; an ambiguity code: the product with its resolution vector), a finished
// child's edge product waits on a small per-thread stack in LDS (children are visited heaviest first: a subtree of n leaves needs
// at most log2 n + 1 slots) — and only the root's rows are stored.  No table reads another: one launch, no synchronisation.
constexpr int kNucStack = 8;
struct RepNucArgs {
  const int32_t *map;
  double *tab;        // [rows][4]
  int32_t *cnt;       // [rows]
  const double *P;    // [B][16] row-major transition matrices of the class being evaluated
  const double *ambig;
};

namespace {
__global__ __launch_bounds__(128) void class_table_nuc_kernel(const int4 *__restrict__ desc, const int *__restrict__ list, RepNucArgs a) {
  const int d = list[blockIdx.y];
  const int4 h0 = desc[2 * d], h1 = desc[2 * d + 1];  // (first row, classes, walked nodes, first node entry), (first index map, rows, -, inputs)
  const int rows = h1.y, n_nodes = h0.z;
  if ((int)blockIdx.x * 128 >= rows) return;
  const int tid = threadIdx.x, u = blockIdx.x * 128 + tid;  // (rows are padded to 16: threads beyond them do nothing but follow)
  const bool live = u < rows;
  __shared__ double stk[kNucStack][4][128];
  __shared__ int stk_cnt[kNucStack][128];
  int sp = 0, e = 0;
  double E[4] = {1., 1., 1., 1.};
  int cnt = 0;
  for (int k = 0; k < n_nodes; k++) {
    const int4 ne = desc[h0.w + k];  // (matrix slot of the node's branch, leaf inputs, first input entry, finished children to take off the stack)
    double acc[4] = {1., 1., 1., 1.};
    cnt = 0;
    for (int j = 0; j < ne.y; j++, e++) {
      const int4 ie = desc[ne.z + j];
      const int code = live ? a.map[h1.x + (size_t)e * rows + u] : 0;
      const double *Pl = a.P + (size_t)ie.y * 16;  // uniform
      if (code >= 0) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const double p0 = Pl[4 * i], p1 = Pl[4 * i + 1], p2 = Pl[4 * i + 2], p3 = Pl[4 * i + 3];
          acc[i] *= (code == 0) ? p0 : (code == 1) ? p1 : (code == 2) ? p2 : p3;
        }
      } else {
        const double *av = a.ambig + (size_t)(-code - 1) * 4;
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] *= fma(Pl[4 * i + 3], av[3], fma(Pl[4 * i + 2], av[2], fma(Pl[4 * i + 1], av[1], Pl[4 * i] * av[0])));
      }
    }
    for (int j = 0; j < ne.w; j++) {
      sp--;
#pragma unroll
      for (int i = 0; i < 4; i++) acc[i] *= stk[sp][i][tid];
      cnt += stk_cnt[sp][tid];
    }
    const double tot = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    double sc;
    const int m = rescale_decision(tot, sc);
    if (m != 0) {
#pragma unroll
      for (int i = 0; i < 4; i++) acc[i] *= sc;
    }
    cnt += m;
    const double *Pn = a.P + (size_t)ne.x * 16;  // uniform
#pragma unroll
    for (int i = 0; i < 4; i++) E[i] = fma(Pn[4 * i + 3], acc[3], fma(Pn[4 * i + 2], acc[2], fma(Pn[4 * i + 1], acc[1], Pn[4 * i] * acc[0])));
    if (k + 1 < n_nodes) {
#pragma unroll
      for (int i = 0; i < 4; i++) stk[sp][i][tid] = E[i];
      stk_cnt[sp][tid] = cnt;
      sp++;
    }
  }
  if (live) {
    double *out = a.tab + ((size_t)h0.x + u) * 4;
    *reinterpret_cast<f64x2 *>(out) = (f64x2){E[0], E[1]};
    *reinterpret_cast<f64x2 *>(out + 2) = (f64x2){E[2], E[3]};
    a.cnt[h0.x + u] = cnt;
  }
}
}  // namespace

// team != 0: the row-split walk, one workgroup of NW waves per item (static passes only; a.n_static items)
void launch_class_tables(const RepArgs &a, int NW, hipStream_t stream, int team = 0) {
  if (team && a.n_static > 0 && NW >= 2) {
    const dim3 tgrid(a.n_static), tblock(64 * NW);
    if (a.dbg && NW == 4) {
      hipLaunchKernelGGL((class_table_team_kernel<4, true>), tgrid, tblock, 0, stream, a.desc, a.items, a);
      return;
    }
    // (measured, r06, headline / 32 x 5 k / 128 x 100 k pruning launches in us — two accumulator chains, 3 + 3 chunks ahead, five teams
    //  per CU: 67.5 / 37.9 / 564.8; all eight A chunks ahead of the barrier, four teams: 71.2 / 37.5 / 569.0; ONE chain, 3 + 2 chunks,
    //  SIX teams per CU (80 registers): 65.8 / 37.3 / 550.4 — the form kept: what a team waits for is its partners, and a sixth team on
    //  the CU fills more of those waits than deeper prefetch does)
    switch (NW) {
      case 2: hipLaunchKernelGGL((class_table_team_kernel<2>), tgrid, tblock, 0, stream, a.desc, a.items, a); break;
      case 3: hipLaunchKernelGGL((class_table_team_kernel<3>), tgrid, tblock, 0, stream, a.desc, a.items, a); break;
      default: hipLaunchKernelGGL((class_table_team_kernel<4>), tgrid, tblock, 0, stream, a.desc, a.items, a); break;
    }
    return;
  }
  const dim3 grid(a.n_waves), block(64);
  if (a.dbg && NW == 4) {
    hipLaunchKernelGGL((class_table_kernel<4, true>), grid, block, 0, stream, a.desc, a.items, a.live, a);
    return;
  }
  switch (NW) {
    case 1: hipLaunchKernelGGL((class_table_kernel<1>), grid, block, 0, stream, a.desc, a.items, a.live, a); break;
    case 2: hipLaunchKernelGGL((class_table_kernel<2>), grid, block, 0, stream, a.desc, a.items, a.live, a); break;
    case 3: hipLaunchKernelGGL((class_table_kernel<3>), grid, block, 0, stream, a.desc, a.items, a.live, a); break;
    default: hipLaunchKernelGGL((class_table_kernel<4>), grid, block, 0, stream, a.desc, a.items, a.live, a); break;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Host side: classes, compressed set, the trunk view, tables and maps (once per partition); item queues per pass.
// ------------------------------------------------------------------------------------------------------------------

namespace {

// class ids of the pairs (a[s], b[s]) in order of first occurrence
int pair_classes(const std::vector<int> &a, const std::vector<int> &b, std::vector<int> &out) {
  std::unordered_map<uint64_t, int> seen;
  seen.reserve(a.size() / 2 + 16);
  out.resize(a.size());
  int n = 0;
  for (size_t s = 0; s < a.size(); s++) {
    const uint64_t key = ((uint64_t)(uint32_t)a[s] << 32) | (uint32_t)b[s];
    auto it = seen.find(key);
    if (it == seen.end()) it = seen.emplace(key, n++).first;
    out[s] = it->second;
  }
  return n;
}

// ... of a single column
int column_classes(const std::vector<int> &a, std::vector<int> &out) {
  std::unordered_map<int, int> seen;
  out.resize(a.size());
  int n = 0;
  for (size_t s = 0; s < a.size(); s++) {
    auto it = seen.find(a[s]);
    if (it == seen.end()) it = seen.emplace(a[s], n++).first;
    out[s] = it->second;
  }
  return n;
}

// class of every (internal node, pattern) bottom-up — the id of the tuple of the children's classes (a leaf: its code, ambiguity codes
// included) in order of first occurrence — and the number of classes per node
void count_classes(const std::vector<std::vector<int>> &children, int L, int I, const std::vector<int16_t> &codes, size_t SP,
                   std::vector<std::vector<int>> &cls, std::vector<int> &U) {
  cls.assign(I, std::vector<int>());
  U.assign(I, 0);
  std::vector<int> col(SP), tmp;
  for (int n = 0; n < I; n++) {
    std::vector<int> run;
    bool first = true;
    for (int c : children[n]) {
      const std::vector<int> *cc;
      if (c < L) {
        for (size_t j = 0; j < SP; j++) col[j] = (int)codes[(size_t)c * SP + j];
        cc = &col;
      } else {
        cc = &cls[c - L];
      }
      if (first) {
        run = *cc;
        first = false;
      } else {
        pair_classes(run, *cc, tmp);
        run.swap(tmp);
      }
    }
    U[n] = column_classes(run, cls[n]);
  }
}

// the compressed set: internal nodes (never the root) with at most theta x patterns classes on every shard, all of whose internal
// children are compressed too
std::vector<char> compressed_set(const std::vector<std::vector<int>> &children, int L, int I, const std::vector<std::vector<int>> &U,
                                 const std::vector<int> &S_pad, double theta) {
  std::vector<char> comp(I, 0);
  for (int n = 0; n < I - 1; n++) {
    bool ok = true;
    for (size_t k = 0; k < U.size(); k++)
      if ((double)U[k][n] > theta * S_pad[k] || U[k][n] > 32000) ok = false;
    for (int c : children[n])
      if (c >= L && !comp[c - L]) ok = false;
    if ((int)children[n].size() > kRepMaxInputs - 8) ok = false;  // (a star: its children's indices would not fit the wave's LDS list)
    comp[n] = ok ? 1 : 0;
  }
  return comp;
}

}  // namespace

// Decide the compressed set, build views[1] and every shard's tables.  `codes[k]`: shard k's leaf table [L][S_pad] in device
// pattern order (padding patterns included: they are patterns like any other).  Leaves p->rep_on false when compression
// would not pay (or cannot be used) — nothing else changes then.
namespace {
void rep_release(hyphy_hip_partition *p) {  // everything rep_setup_impl may have left behind: the partition runs plain
  for (Shard &s : p->shards) {
    if (hipSetDevice(s.device) != hipSuccess) continue;
    void **bufs[] = {(void **)&s.rep_tab, (void **)&s.rep_cnt, (void **)&s.rep_map, (void **)&s.rep_desc, (void **)&s.rep_sync,
                     (void **)&s.rep_codes_tile, (void **)&s.rep_leaf, (void **)&s.rep_walk};
    for (void **b : bufs)
      if (*b) {
        pool_free_sync(*b);
        *b = nullptr;
      }
    s.dev_bytes -= std::min(s.dev_bytes, s.rep_bytes);
    s.rep_bytes = 0;
    s.rep_tabs.clear();
    s.rep_rows = 0;
  }
  p->rep_nodes.clear();
  p->rep_desc_of.clear();
  p->views[1] = hyphy_hip_partition::View();
  p->rep_on = false;
}
int rep_setup_impl(hyphy_hip_partition *p, const std::vector<std::vector<int16_t>> &codes);
}  // namespace

// Subtree repeats are an optimisation: whatever goes wrong while they are set up (tables that do not fit the device, a failed
// copy) leaves a partition that evaluates every node at every pattern — hyphy_hip_create never fails because of them.
int rep_setup(hyphy_hip_partition *p, const std::vector<std::vector<int16_t>> &codes) {
  const int rc = rep_setup_impl(p, codes);
  if (rc != 0 || !p->rep_on) {
    if (rc != 0 && getenv("HYPHY_HIP_VERBOSE")) fprintf(stderr, "[hyphy_hip] subtree repeats not set up (%s): plain evaluation\n", g_last_error.c_str());
    rep_release(p);
    (void)hipGetLastError();
    g_last_error.clear();
  }
  return 0;
}

namespace {
// 4 states (see class_table_nuc_kernel): one descriptor per compressed subtree ROOT, walked whole; theta 0.9 by default (a walk costs
// 16 multiply-adds per node: everything that repeats at all is worth a table).
int rep_setup_nuc(hyphy_hip_partition *p, const std::vector<std::vector<int16_t>> &codes, const std::vector<std::vector<std::vector<int>>> &cls,
                  const std::vector<std::vector<int>> &U, std::vector<char> &comp, bool forced) {
  const int L = (int)p->L, I = (int)p->I;
  const size_t nsh = p->shards.size();
  // stack slots a subtree's walk needs when every node visits its internal children heaviest first
  std::vector<int> need(I, 1);
  std::vector<std::vector<int>> order(I);  // internal children (internal indices) in visiting order
  for (int n = 0; n < I; n++) {
    for (int c : p->children[n])
      if (c >= L) order[n].push_back(c - L);
    std::stable_sort(order[n].begin(), order[n].end(), [&](int x, int y) { return need[x] > need[y]; });
    for (size_t j = 0; j < order[n].size(); j++) need[n] = std::max(need[n], (int)j + need[order[n][j]]);
  }
  for (int n = I - 1; n >= 0; n--)  // (parents first: a subtree too deep for the stack hands the role of root to its children)
    if (comp[n] && (p->parents[L + n] < 0 || !comp[p->parents[L + n]]) && need[n] > kNucStack) comp[n] = 0;
  {
    double full = 0., repd = 0.;
    for (int n = 0; n < I - 1; n++) {
      full += p->shards[0].S_pad;
      const bool root = comp[n] && !comp[p->parents[L + n]];
      repd += comp[n] ? (root ? (double)U[0][n] : 0.) : (double)p->shards[0].S_pad;  // (nodes inside a walk cost next to nothing here)
    }
    if (!forced && repd > 0.5 * full) return 0;
  }
  p->rep_nodes.clear();
  p->rep_desc_of.assign((size_t)L + I, -1);
  for (int n = 0; n < I; n++) {
    if (!comp[n] || comp[p->parents[L + n]]) continue;  // roots only
    hyphy_hip_partition::RepNode rn;
    rn.node = L + n;
    rn.level = 0;
    // post-order over the subtree, heaviest child first
    std::vector<std::pair<int, size_t>> stack(1, std::make_pair(n, (size_t)0));
    while (!stack.empty()) {
      std::pair<int, size_t> &t = stack.back();
      if (t.second < order[t.first].size()) {
        const int c = order[t.first][t.second++];
        stack.push_back(std::make_pair(c, (size_t)0));
      } else {
        const int x = t.first;
        rn.path.push_back(L + x);
        rn.flags.push_back((int)order[x].size());  // (4 states: finished children this node takes off the stack)
        rn.kids.push_back(std::vector<int>());
        rn.kid_desc.push_back(std::vector<int>());
        for (int c : p->children[x])
          if (c < L) {
            rn.kids.back().push_back(c);
            rn.kid_desc.back().push_back(-1);
          }
        stack.pop_back();
      }
    }
    p->rep_desc_of[L + n] = (int)p->rep_nodes.size();
    p->rep_nodes.push_back(rn);
  }
  const int ND = (int)p->rep_nodes.size();
  if (ND == 0 || ND > 65535) return 0;
  // the trunk view: ordinary leaves (ambiguity codes stay with the trunk kernel's own resolution-vector path) and compressed roots
  hyphy_hip_partition::View &v = p->views[1];
  v = hyphy_hip_partition::View();
  std::vector<int> vint(I, -1);
  for (int n = 0; n < I; n++)
    if (!comp[n]) vint[n] = v.I++;
  std::vector<int> leaf_nodes, vleaf((size_t)L + I, -1);
  for (int n = 0; n < I; n++)
    if (!comp[n])
      for (int c : p->children[n])
        if (c < L || comp[c - L]) {
          vleaf[c] = (int)leaf_nodes.size();
          leaf_nodes.push_back(c);
        }
  v.L = (int)leaf_nodes.size();
  if (v.L > 65535 || v.L < 1) return 0;
  v.parents.assign((size_t)v.L + v.I, -1);
  v.children.assign(v.I, std::vector<int>());
  v.slot.assign((size_t)v.L + v.I, 0);
  v.leaf_has_ambig.assign(v.L, 0);
  for (int n = 0; n < I; n++) {
    if (comp[n]) continue;
    v.slot[v.L + vint[n]] = L + n;
    const int64_t par = p->parents[L + n];
    v.parents[v.L + vint[n]] = par < 0 ? -1 : vint[par];
    for (int c : p->children[n]) {
      if (c < L || comp[c - L]) {
        v.children[vint[n]].push_back(vleaf[c]);
        v.parents[vleaf[c]] = vint[n];
        v.slot[vleaf[c]] = c;
        if (c < L) v.leaf_has_ambig[vleaf[c]] = p->leaf_has_ambig[c];
      } else {
        v.children[vint[n]].push_back(v.L + vint[c - L]);
      }
    }
    std::sort(v.children[vint[n]].begin(), v.children[vint[n]].end());
  }
  for (size_t k = 0; k < nsh; k++) {
    Shard &s = p->shards[k];
    if (hipSetDevice(s.device) != hipSuccess) return fail("hipSetDevice failed");
    const size_t SP = (size_t)s.S_pad;
    s.rep_tabs.assign(ND, RepTable());
    std::vector<int32_t> maps;
    int64_t rows = 0;
    std::vector<int4> desc((size_t)2 * ND);
    for (int d = 0; d < ND; d++) {
      const hyphy_hip_partition::RepNode &rn = p->rep_nodes[d];
      RepTable &t = s.rep_tabs[d];
      const std::vector<int> &cl = cls[k][rn.node - L];
      t.U = U[k][rn.node - L];
      t.rows = (t.U + 15) / 16 * 16;
      t.row0 = rows;
      rows += t.rows;
      std::vector<int> first_pat(t.rows, -1);
      for (size_t j = 0; j < SP; j++)
        if (first_pat[cl[j]] < 0) first_pat[cl[j]] = (int)j;
      for (int u = t.U; u < t.rows; u++) first_pat[u] = first_pat[0];
      const int64_t map0 = (int64_t)maps.size();
      int n_in = 0;
      for (size_t x = 0; x < rn.path.size(); x++)
        for (int c : rn.kids[x]) {
          for (int u = 0; u < t.rows; u++) maps.push_back((int32_t)codes[k][(size_t)c * SP + first_pat[u]]);
          n_in++;
        }
      t.map0.push_back(map0);
      const int node0 = (int)desc.size();
      desc[2 * d] = make_int4((int)t.row0, t.U, (int)rn.path.size(), node0);
      desc[2 * d + 1] = make_int4((int)map0, t.rows, 0, n_in);
      desc.resize(desc.size() + rn.path.size());
      for (size_t x = 0; x < rn.path.size(); x++) {
        desc[node0 + x] = make_int4(rn.path[x], (int)rn.kids[x].size(), (int)desc.size(), rn.flags[x]);
        for (int c : rn.kids[x]) desc.push_back(make_int4(-1, c, 0, -1));
      }
    }
    s.rep_rows = rows;
    if (maps.size() > 0x7fffffffull || rows > 0x7fffffffll) return 0;
    std::vector<int16_t> ct((size_t)v.L * SP, 0);  // the trunk's leaf table, row-major [view leaf][pattern]: state codes / class ids
    std::vector<int2> lt(v.L);
    for (int vl = 0; vl < v.L; vl++) {
      const int c = leaf_nodes[vl], d = p->rep_desc_of[c];
      lt[vl] = d >= 0 ? make_int2((int)s.rep_tabs[d].row0, (int)s.rep_tabs[d].row0) : make_int2(-1, c);
      for (size_t j = 0; j < SP; j++) ct[(size_t)vl * SP + j] = (int16_t)(d < 0 ? (int)codes[k][(size_t)c * SP + j] : cls[k][c - L][j]);
    }
    {
      size_t free_b = 0, total_b = 0, limit = (size_t)1 << 30;
      const size_t need_b = (size_t)p->C * rows * 36 + maps.size() * 4 + ct.size() * 2;
      if (const char *e = getenv("HYPHY_HIP_REP_MAX_MB")) limit = (size_t)std::max(0L, atol(e)) << 20;
      else if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) limit = free_b / 4;
      if (need_b > limit) return fail("class tables would take more than a quarter of the free device memory (or than HYPHY_HIP_REP_MAX_MB)");
    }
#define R_(ptr, bytes)                                                                  \
  if (pool_malloc((void **)&(ptr), (bytes)) != hipSuccess) return fail("hipMalloc failed (" #ptr ")"); \
  s.dev_bytes += (size_t)(bytes);                                                                      \
  s.rep_bytes += (size_t)(bytes);
    R_(s.rep_tab, (size_t)p->C * rows * 4 * sizeof(double));
    R_(s.rep_cnt, (size_t)p->C * rows * sizeof(int32_t));
    R_(s.rep_map, std::max<size_t>(1, maps.size()) * sizeof(int32_t));
    R_(s.rep_desc, desc.size() * sizeof(int4));
    R_(s.rep_codes_tile, ct.size() * sizeof(int16_t));
    R_(s.rep_leaf, lt.size() * sizeof(int2));
#undef R_
    if (getenv("HYPHY_HIP_POISON")) {
      hipMemset(s.rep_tab, 0xff, (size_t)p->C * rows * 4 * sizeof(double));
      hipMemset(s.rep_cnt, 0xff, (size_t)p->C * rows * sizeof(int32_t));
    }
    hipMemcpy(s.rep_map, maps.data(), maps.size() * sizeof(int32_t), hipMemcpyHostToDevice);
    hipMemcpy(s.rep_desc, desc.data(), desc.size() * sizeof(int4), hipMemcpyHostToDevice);
    hipMemcpy(s.rep_codes_tile, ct.data(), ct.size() * sizeof(int16_t), hipMemcpyHostToDevice);
    hipMemcpy(s.rep_leaf, lt.data(), lt.size() * sizeof(int2), hipMemcpyHostToDevice);
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) return fail("repeats: device initialisation failed");
  }
  p->rep_resident.assign(p->C, 0);
  p->rep_cached_valid = false;
  hyphy_hip_partition::ModeState &ms = p->saved_mode[1];
  ms.variant = p->variant;
  ms.wave_variant = 0;
  ms.n_slots = p->n_slots;   // (2 + kNucParkSlots)
  ms.chain_m_forced = 0;
  ms.rr_use = false;
  ms.kernel_forced = true;
  ms.tuned_for = 0;
  ms.nuc_leaf_pairs = false;  // (the trunk goes through prune_nuc_kernel: one leaf per entry)
  p->rep_on = true;
  p->rep_decided = true;       // (4 states: decided here, by the arithmetic above)
  if (getenv("HYPHY_HIP_VERBOSE")) {
    long long rows = 0, lower = 0;
    for (int d = 0; d < ND; d++) {
      rows += p->shards[0].rep_tabs[d].rows;
      lower += (long long)p->shards[0].rep_tabs[d].rows * (long long)p->rep_nodes[d].path.size();
    }
    fprintf(stderr, "[hyphy_hip] subtree repeats (4 states): %d compressed subtrees walked per class (%lld classes on shard 0, %lld node visits), trunk of %d "
                    "internal nodes over %d leaves (every pattern at every node: %lld)\n",
            ND, rows, lower, v.I, v.L, (long long)(I - 1) * p->shards[0].S_pad);
  }
  return 0;
}
}  // namespace

// The trunk as one post-order walk per tile (trunk_walk_kernel): heaviest internal child first — its chain stays in the running
// product —, every other internal child a chain of its own behind a push (flag 1 on its first walked node, 2 on the child itself).
// Two forms: ONE workgroup per tile walks everything; TWO split the subtrees below the root (longest-processing-time-first); each
// chain ends with a node -1: the root's own leaf children (first chain only) and the hand-over / epilogue.
struct WalkPlan {
  struct Node { int node, n_in, in0, flags; };  // node: view-internal index (-1: the chain's end at the root); in0: first entry of `inputs`
  std::vector<Node> nodes;
  std::vector<int> inputs;                      // view leaves
  std::vector<int> one_range, two_range;        // [first, end) node ranges: the one-chain form; the two chains
  int max_depth = 0;
  bool two = false;
};
static WalkPlan plan_trunk_walk(const hyphy_hip_partition::View &v) {
  WalkPlan wp;
  std::vector<int> weight(v.I, 1);
  for (int i = 0; i < v.I; i++)  // (children before parents)
    for (int c : v.children[i])
      if (c >= v.L) weight[i] += weight[c - v.L];
  std::vector<WalkPlan::Node> &nodes = wp.nodes;
  std::vector<int> &inputs = wp.inputs;
  int depth = 0;
  auto leaf_inputs = [&](int i, int &n_in, int &in0) {
    n_in = 0, in0 = (int)inputs.size();
    for (int c : v.children[i])
      if (c < v.L) inputs.push_back(c), n_in++;
  };
  std::function<void(int)> emit = [&](int i) {
    std::vector<int> kids;
    for (int c : v.children[i])
      if (c >= v.L) kids.push_back(c - v.L);
    std::stable_sort(kids.begin(), kids.end(), [&](int x, int y) { return weight[x] > weight[y]; });
    for (size_t j = 0; j < kids.size(); j++) {
      const size_t first = nodes.size();
      if (j > 0) wp.max_depth = std::max(wp.max_depth, ++depth);
      emit(kids[j]);
      if (j > 0) {
        nodes[first].flags |= 1;
        nodes.back().flags |= 2;
        depth--;
      }
    }
    int n_in, in0;
    leaf_inputs(i, n_in, in0);
    nodes.push_back(WalkPlan::Node{i, n_in, in0, 0});
  };
  auto emit_chain = [&](const std::vector<int> &subs, bool root_leaves, std::vector<int> &range) {
    range.push_back((int)nodes.size());
    for (size_t j = 0; j < subs.size(); j++) {
      const size_t first = nodes.size();
      if (j > 0) wp.max_depth = std::max(wp.max_depth, ++depth);
      emit(subs[j]);
      if (j > 0) {
        nodes[first].flags |= 1;
        nodes.back().flags |= 2;
        depth--;
      }
    }
    int n_in = 0, in0 = (int)inputs.size();
    if (root_leaves) leaf_inputs(v.I - 1, n_in, in0);
    nodes.push_back(WalkPlan::Node{-1, n_in, in0, 0});
    range.push_back((int)nodes.size());
  };
  std::vector<int> subs;
  for (int c : v.children[v.I - 1])
    if (c >= v.L) subs.push_back(c - v.L);
  std::stable_sort(subs.begin(), subs.end(), [&](int x, int y) { return weight[x] > weight[y]; });
  emit_chain(subs, true, wp.one_range);
  if (subs.size() >= 2) {
    std::vector<int> c0, c1;
    int w0 = 0, w1 = 0;
    for (int sb : subs)
      if (w0 <= w1) c0.push_back(sb), w0 += weight[sb];
      else c1.push_back(sb), w1 += weight[sb];
    if (std::min(w0, w1) >= 2 && 4 * std::min(w0, w1) >= std::max(w0, w1)) {  // (a second chain of a node or two is not worth its workgroup)
      emit_chain(c0, true, wp.two_range);
      emit_chain(c1, false, wp.two_range);
      wp.two = true;
    }
  }
  return wp;
}

namespace {
int rep_setup_impl(hyphy_hip_partition *p, const std::vector<std::vector<int16_t>> &codes) {
  p->rep_on = false;
  const char *env = getenv("HYPHY_HIP_REPEATS");
  if (env && atoi(env) == 0) return 0;
  if (p->shards.empty()) return 0;
  if (!env)  // (the diagnostic switches that force a kernel, an instantiation or a cut are about the per-pattern kernels)
    for (const char *sw : {"HYPHY_HIP_KERNEL", "HYPHY_HIP_WAVE_VARIANT", "HYPHY_HIP_CUT", "HYPHY_HIP_CHAIN_M", "HYPHY_HIP_FRAGMENT",
                           "HYPHY_HIP_SLOTS", "HYPHY_HIP_REROOT", "HYPHY_HIP_TILES", "HYPHY_HIP_TIMELINE", "HYPHY_HIP_ABLATE"})
      if (getenv(sw)) return 0;
  for (const Shard &s : p->shards)
    if (s.T != 1) return 0;
  if (!p->nuc && p->variant != 1 && !(env && atoi(env) == 2)) return 0;  // (tiny shards: the workgroup-per-tile kernel keeps the whole tree)
  // 4 states: only on request (HYPHY_HIP_REPEATS=1 / 2).  Measured, it loses: a class row is 36 bytes gathered past the L2 per pattern
  // and generalised leaf, where prune_nuc2_kernel computes a whole node from registers for 12 multiply-adds — gtr_32x50k 35.0 us plain
  // against 16.5 (walks) + 26.0 (trunk of 5 nodes) compressed, gtr_32x1m 123.8 against 14.5 + 238.9
  if (p->nuc && !env) return 0;
  const int L = (int)p->L, I = (int)p->I, DP = p->DP;
  if (I < 2) return 0;
  const bool forced = env && atoi(env) == 2;
  // (before any work: a partition of a few tiles — FEL makes one per site — has nothing to compress, and the class arrays of a
  //  very large one are host memory this analysis should not take: 4 bytes per internal node and pattern)
  if (!forced && (p->nuc ? p->shards[0].S_pad < 8192 : p->shards[0].ntiles < 8)) return 0;  // (4 states: a small shard is one launch
                                                                                              //  with exponentials and combine folded in)
  {
    double cells = 0.;
    for (const Shard &s : p->shards) cells += (double)I * s.S_pad;
    // (2.5e8 cells = 1 GB of transient host memory and a few seconds of hashing inside hyphy_hip_create, before it is known
    //  whether compression pays — 128 taxa x 100 000 codons is 1.3e7; beyond the bound the partition runs plain)
    if (cells > 2.5e8) return 0;
  }
  // theta: the share of the patterns below which a node's classes are worth a table.  0.3-0.4 is the flat optimum of the headline
  // alignment (0.2: 100 us of pruning launches, 0.3: 89, 0.4: 89, 0.5: 101, 0.7: 106): above it the walks of the lower phase get long
  // and few, below it the trunk keeps nodes that repeat heavily
  double theta = 0.35;
  if (const char *e = getenv("HYPHY_HIP_REP_THETA")) theta = atof(e);
  const size_t nsh = p->shards.size();
  // ---- classes per shard ----
  std::vector<std::vector<std::vector<int>>> cls(nsh);  // [shard][internal node][pattern]
  std::vector<std::vector<int>> U(nsh, std::vector<int>(I, 0));
  for (size_t k = 0; k < nsh; k++) count_classes(p->children, L, I, codes[k], (size_t)p->shards[k].S_pad, cls[k], U[k]);
  // ---- compressed set: the same on every shard (one schedule serves them all) ----
  std::vector<int> spads;
  for (const Shard &s : p->shards) spads.push_back(s.S_pad);
  std::vector<char> comp = compressed_set(p->children, L, I, U, spads, p->nuc && !getenv("HYPHY_HIP_REP_THETA") ? 0.9 : theta);
  for (size_t k = 0; k < nsh; k++)  // (only the classes of compressed nodes are read again: tables, maps, the trunk's leaf table)
    for (int n = 0; n < I; n++)
      if (!comp[n]) std::vector<int>().swap(cls[k][n]);
  if (p->nuc) return rep_setup_nuc(p, codes, cls, U, comp, forced);
  {  // worth it?  compare the edge products of the two forms on the first shard
    double full = 0., rep = 0.;
    for (int n = 0; n < I - 1; n++) {
      full += p->shards[0].S_pad;
      rep += comp[n] ? (U[0][n] + 15) / 16 * 16 : p->shards[0].S_pad;
    }
    const double need = getenv("HYPHY_HIP_REP_MIN_GAIN") ? atof(getenv("HYPHY_HIP_REP_MIN_GAIN")) : 0.15;
    if (!forced && rep > (1.0 - need) * full) return 0;
  }
  // ---- descriptors: leaves with ambiguity codes first (level 0), then the paths, inputs before the paths that read them ----
  // rho: a path goes on into the heaviest compressed child while that child keeps at least rho of the node's classes.  0: paths
  // run to the bottom of the subtree — no table reads a table, one launch, more products (3.4 x those of one table per node at
  // 128 x 100 k).  Larger values trade products for hand-offs; a hand-off is a launch boundary (rep_build_items: one launch per
  // level).  Where the lower phase is bound by its walks' lengths (a shard of a few hundred tiles: 1.5 walks per wave) rho = 0 wins
  // (headline pruning launches 74.5 us, 79.0 at 0.3, 77.0 at 0.6, 83.2 at 2); where it is bound by throughput (128 x 100 k: 6 250
  // tiles, eight rounds of walks) the saved products win: 663.6 us at 0, 584.2 at 0.15, 537.7 at 0.3, 549.6 at 0.45, 552.2 at 2.
  // In between (64 taxa): 1 250 tiles 110.0 us at 0 against 119.1 at 0.3; 2 500 tiles 173.8 against 169.3 — the default changes at
  // 2 048.  (Before the per-level launches, with the ticket protocol: 785 us at 0.6 against 726 at 0.)  HYPHY_HIP_REP_RHO overrides.
  // (r06, row-split walks in both phases, pruning launches at 128 x 100 k: 625 us at 0, 537 at 0.15, 501 at 0.3, 489 at 0.5, 485 at 0.6,
  //  504 at 0.7, 508 at 1 and 2: the default there moved from 0.3 to 0.5)
  double rho = p->shards[0].ntiles >= 2048 ? 0.5 : 0.;
  if (const char *e = getenv("HYPHY_HIP_REP_RHO")) rho = atof(e);
  std::vector<int> heavy(I, -1);     // the compressed child a node's path continues into
  std::vector<char> continued(I, 0); // the node is inside its parent's path (no table of its own)
  std::vector<int> path_inputs(I, 0), path_len(I, 0);
  for (int n = 0; n < I; n++) {
    if (!comp[n]) continue;
    int best = -1;
    for (int c : p->children[n])
      if (c >= L && (best < 0 || U[0][c - L] > U[0][best])) best = c - L;
    // (a path's inputs are staged in LDS by index: long caterpillars / wide multifurcations are cut into several paths)
    const int own_inputs = (int)p->children[n].size();
    path_inputs[n] = own_inputs;
    path_len[n] = 1;
    if (best >= 0 && (double)U[0][best] >= rho * (double)U[0][n] && path_inputs[best] + own_inputs - 1 <= kRepMaxInputs - 8 &&
        path_len[best] < 32 && own_inputs <= kRepMaxInputs / 2) {
      heavy[n] = best;
      continued[best] = 1;
      path_inputs[n] = path_inputs[best] + own_inputs - 1;
      path_len[n] = path_len[best] + 1;
    }
  }
  p->rep_nodes.clear();
  p->rep_desc_of.assign((size_t)L + I, -1);
  for (int l = 0; l < L; l++)
    if (p->leaf_has_ambig[l]) {
      hyphy_hip_partition::RepNode rn;
      rn.node = l;
      rn.level = 0;
      p->rep_desc_of[l] = (int)p->rep_nodes.size();
      p->rep_nodes.push_back(rn);
    }
  // Side chains walked inline (latency mode, rho = 0).  A side path whose nodes read leaves only — a cherry, a short caterpillar —
  // hanging off another path is not worth a table of its own: its table would be one more hand-off in the dependency chain of the
  // path that reads it (publish, drain, counter, poll, gather: ~8 us with the item's fixed costs) to save a few products.  Such a
  // chain is walked INSIDE the item of the path it hangs off: the running product is parked in the wave's LDS tile, the chain
  // is walked from ones, and its edge product is multiplied back in (one parking slot: inlined chains do not nest).
  const bool inline_chains = getenv("HYPHY_HIP_REP_INLINE") ? atoi(getenv("HYPHY_HIP_REP_INLINE")) != 0 : rho == 0.;
  auto chain_of = [&](int top) {
    std::vector<int> down;
    for (int x = top; x >= 0; x = heavy[x]) down.push_back(x);
    return std::vector<int>(down.rbegin(), down.rend());  // bottom first
  };
  std::vector<char> leafy(I, 0);  // a path top whose whole path reads ordinary leaves / leaf tables only
  for (int n = 0; n < I; n++)
    if (comp[n] && !continued[n]) {
      bool ok = true;
      for (int x : chain_of(n))
        for (int c : p->children[x])
          if (c >= L && c - L != heavy[x]) ok = false;
      leafy[n] = ok ? 1 : 0;
    }
  // (first which chains go inline — the path they hang off decides, within what its LDS index list holds —, then the descriptors)
  std::vector<char> absorbed(I, 0);
  std::vector<std::vector<int>> inline_at(I);  // per path node: the side tops walked inline in front of it
  if (inline_chains)
    for (int n = 0; n < I; n++)
      if (comp[n] && !continued[n]) {
        int n_inputs = 0, n_entries = 0;
        for (int x : chain_of(n)) n_inputs += (int)p->children[x].size() - (heavy[x] >= 0 ? 1 : 0), n_entries++;
        for (int x : chain_of(n))
          for (int c : p->children[x]) {
            if (c < L || c - L == heavy[x] || !comp[c - L] || !leafy[c - L]) continue;
            const std::vector<int> side = chain_of(c - L);
            int side_inputs = 0;
            for (int y : side) side_inputs += (int)p->children[y].size();
            if (n_inputs + side_inputs > kRepMaxInputs - 8 || n_entries + (int)side.size() > 40) continue;
            n_inputs += side_inputs - 1;  // (the chain's table would have been one input)
            n_entries += (int)side.size();
            absorbed[c - L] = 1;
            inline_at[x].push_back(c - L);
          }
      }
  for (int n = 0; n < I; n++)
    if (comp[n] && !continued[n] && !absorbed[n]) {  // a path's top (ascending: every input's top comes first)
      hyphy_hip_partition::RepNode rn;
      rn.node = L + n;
      rn.level = 0;
      auto add_node = [&](int x, int skip_child, int flags) {
        rn.path.push_back(L + x);
        rn.flags.push_back(flags);
        rn.kids.push_back(std::vector<int>());
        rn.kid_desc.push_back(std::vector<int>());
        for (int c : p->children[x]) {
          if (c == skip_child) continue;  // (comes through the registers / was walked inline just before)
          if (c >= L && absorbed[c - L]) continue;
          rn.kids.back().push_back(c);
          const int d = p->rep_desc_of[c];
          rn.kid_desc.back().push_back(d);
          if (d >= 0) rn.level = std::max(rn.level, p->rep_nodes[d].level + 1);
        }
      };
      const std::vector<int> main_path = chain_of(n);
      for (size_t k = 0; k < main_path.size(); k++) {
        const int x = main_path[k];
        for (int c : inline_at[x]) {
          const std::vector<int> side = chain_of(c);
          for (size_t j = 0; j < side.size(); j++)
            add_node(side[j], j > 0 ? L + side[j - 1] : -1, (j == 0 ? 1 : 0) | (j + 1 == side.size() ? 2 : 0));
        }
        add_node(x, k > 0 ? L + main_path[k - 1] : -1, 0);
      }
      p->rep_desc_of[L + n] = (int)p->rep_nodes.size();
      p->rep_nodes.push_back(rn);
    }
  const int ND = (int)p->rep_nodes.size();
  if (ND == 0 || ND > 65535) return 0;
  // ---- the trunk view ----
  hyphy_hip_partition::View &v = p->views[1];
  v = hyphy_hip_partition::View();
  std::vector<int> vint(I, -1);  // view-internal index of a trunk node
  for (int n = 0; n < I; n++)
    if (!comp[n]) vint[n] = v.I++;
  std::vector<int> leaf_nodes;  // node code (partition's tree) of every view leaf
  std::vector<int> vleaf((size_t)L + I, -1);
  for (int n = 0; n < I; n++)
    if (!comp[n])
      for (int c : p->children[n])
        if (c < L || comp[c - L]) {
          vleaf[c] = (int)leaf_nodes.size();
          leaf_nodes.push_back(c);
        }
  v.L = (int)leaf_nodes.size();
  if (v.L > 65535 || v.L < 1) return 0;
  v.parents.assign((size_t)v.L + v.I, -1);
  v.children.assign(v.I, std::vector<int>());
  v.slot.assign((size_t)v.L + v.I, 0);
  v.leaf_has_ambig.assign(v.L, 0);  // (ambiguity codes live in class tables: no leaf of the view takes the resolution-vector path)
  for (int n = 0; n < I; n++) {
    if (comp[n]) continue;
    v.slot[v.L + vint[n]] = L + n;
    const int64_t par = p->parents[L + n];
    v.parents[v.L + vint[n]] = par < 0 ? -1 : vint[par];
    for (int c : p->children[n]) {
      if (c < L || comp[c - L]) {
        v.children[vint[n]].push_back(vleaf[c]);
        v.parents[vleaf[c]] = vint[n];
        v.slot[vleaf[c]] = c;
      } else {
        v.children[vint[n]].push_back(v.L + vint[c - L]);
      }
    }
    std::sort(v.children[vint[n]].begin(), v.children[vint[n]].end());
  }
  // ---- per shard: table rows, index maps, descriptors, the trunk's leaf table ----
  for (size_t k = 0; k < nsh; k++) {
    Shard &s = p->shards[k];
    if (hipSetDevice(s.device) != hipSuccess) return fail("hipSetDevice failed");
    const size_t SP = (size_t)s.S_pad;
    s.rep_tabs.assign(ND, RepTable());
    std::vector<std::vector<int>> leaf_cls(L);  // classes of the leaves that own a table
    std::vector<int32_t> maps;
    int64_t rows = 0;
    std::vector<std::vector<int>> first_pat(ND);  // representative pattern of every class
    for (int d = 0; d < ND; d++) {
      const hyphy_hip_partition::RepNode &rn = p->rep_nodes[d];
      RepTable &t = s.rep_tabs[d];
      const std::vector<int> *cl;
      std::vector<int> col(SP);
      if (rn.node < L) {
        for (size_t j = 0; j < SP; j++) col[j] = (int)codes[k][(size_t)rn.node * SP + j];
        t.U = column_classes(col, leaf_cls[rn.node]);
        cl = &leaf_cls[rn.node];
      } else {
        t.U = U[k][rn.node - L];
        cl = &cls[k][rn.node - L];
      }
      t.rows = (t.U + 15) / 16 * 16;
      t.row0 = rows;
      rows += t.rows;
      first_pat[d].assign(t.rows, -1);
      for (size_t j = 0; j < SP; j++)
        if (first_pat[d][(*cl)[j]] < 0) first_pat[d][(*cl)[j]] = (int)j;
      for (int u = t.U; u < t.rows; u++) first_pat[d][u] = first_pat[d][0];  // padding rows repeat class 0
      if (rn.node < L) {  // the code of every class
        t.map0.push_back((int64_t)maps.size());
        for (int u = 0; u < t.rows; u++) maps.push_back((int32_t)codes[k][(size_t)rn.node * SP + first_pat[d][u]]);
      } else {
        for (size_t x = 0; x < rn.path.size(); x++)
          for (size_t j = 0; j < rn.kids[x].size(); j++) {
            t.map0.push_back((int64_t)maps.size());
            const int c = rn.kids[x][j], cd = rn.kid_desc[x][j];
            for (int u = 0; u < t.rows; u++) {
              const int pat = first_pat[d][u];  // (a class of the top node fixes the class of everything below it)
              int32_t idx;
              if (cd < 0) idx = (int32_t)codes[k][(size_t)c * SP + pat];        // ordinary leaf: its state
              else if (c < L) idx = (int32_t)leaf_cls[c][pat];                  // leaf with a table: class of its code
              else idx = (int32_t)cls[k][c - L][pat];                           // compressed child: its class
              maps.push_back(idx);
            }
          }
      }
    }
    s.rep_rows = rows;
    if (maps.size() > 0x7fffffffull || rows > 0x7fffffffll) return 0;
    std::vector<int4> desc((size_t)2 * ND);
    for (int d = 0; d < ND; d++) {
      const hyphy_hip_partition::RepNode &rn = p->rep_nodes[d];
      const RepTable &t = s.rep_tabs[d];
      const int kind = rn.node < L ? 1 : 0;
      const int node0 = (int)desc.size();
      desc[2 * d] = make_int4((int)t.row0, t.U, (int)rn.path.size() | (kind << 16), node0);
      int n_in = 0;
      for (const std::vector<int> &kk : rn.kids) n_in += (int)kk.size();
      desc[2 * d + 1] = make_int4(t.map0.empty() ? 0 : (int)t.map0[0], t.rows / 16, rn.level, kind ? rn.node : n_in);
      desc.resize(desc.size() + rn.path.size());
      size_t mi = 0;
      for (size_t x = 0; x < rn.path.size(); x++) {
        desc[node0 + x] = make_int4(rn.path[x], (int)rn.kids[x].size(), (int)desc.size(), rn.flags[x]);
        for (size_t j = 0; j < rn.kids[x].size(); j++, mi++) {
          const int cd = rn.kid_desc[x][j];
          desc.push_back(make_int4(cd >= 0 ? (int)s.rep_tabs[cd].row0 : -1, rn.kids[x][j], (int)t.map0[mi],
                                   cd >= 0 ? (cd | ((s.rep_tabs[cd].rows / 16) << 16)) : -1));
        }
      }
    }
    // the trunk's leaf table, tile-major, and where each view leaf gathers from
    std::vector<int16_t> ct((size_t)v.L * SP, 0);
    std::vector<int2> lt(v.L);
    for (int vl = 0; vl < v.L; vl++) {
      const int c = leaf_nodes[vl], d = p->rep_desc_of[c];
      lt[vl] = d >= 0 ? make_int2((int)s.rep_tabs[d].row0, (int)s.rep_tabs[d].row0) : make_int2(-1, c);
      for (size_t j = 0; j < SP; j++) {
        int val;
        if (d < 0) val = (int)codes[k][(size_t)c * SP + j];
        else if (c < L) val = leaf_cls[c][j];
        else val = cls[k][c - L][j];
        ct[((j >> 4) * (size_t)v.L + vl) * 16 + (j & 15)] = (int16_t)val;
      }
    }
    const size_t sync_words = ((size_t)kRepQueues + 1 + (size_t)p->C * ND) * kRepHeadStride;  // (= rep_sync_words(p) words)
    {  // the tables must leave the device room for everything else the partition allocates later (deposits, mixtures, site fits)
      size_t free_b = 0, total_b = 0;
      const size_t need_b = (size_t)p->C * rows * (DP * sizeof(double) + sizeof(int32_t)) + maps.size() * sizeof(int32_t) + ct.size() * sizeof(int16_t);
      size_t limit = 0;
      if (const char *e = getenv("HYPHY_HIP_REP_MAX_MB")) limit = (size_t)std::max(0L, atol(e)) << 20;  // (the caller's bound on the tables)
      else if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) limit = free_b / 4;
      else limit = (size_t)1 << 30;
      if (need_b > limit) return fail("class tables would take more than a quarter of the free device memory (or than HYPHY_HIP_REP_MAX_MB)");
    }
#define R_(ptr, bytes)                                                                  \
  if (pool_malloc((void **)&(ptr), (bytes)) != hipSuccess) return fail("hipMalloc failed (" #ptr ")"); \
  s.dev_bytes += (size_t)(bytes);                                                                      \
  s.rep_bytes += (size_t)(bytes);
    R_(s.rep_tab, (size_t)p->C * rows * DP * sizeof(double));
    R_(s.rep_cnt, (size_t)p->C * rows * sizeof(int32_t));
    R_(s.rep_map, std::max<size_t>(1, maps.size()) * sizeof(int32_t));
    R_(s.rep_desc, desc.size() * sizeof(int4));
    R_(s.rep_sync, sync_words * sizeof(int));
    R_(s.rep_codes_tile, ct.size() * sizeof(int16_t));
    R_(s.rep_leaf, lt.size() * sizeof(int2));
#undef R_
    if (getenv("HYPHY_HIP_POISON")) {
      hipMemset(s.rep_tab, 0xff, (size_t)p->C * rows * DP * sizeof(double));
      hipMemset(s.rep_cnt, 0xff, (size_t)p->C * rows * sizeof(int32_t));
    }
    hipMemset(s.rep_sync, 0, sync_words * sizeof(int));
    hipMemcpy(s.rep_map, maps.data(), maps.size() * sizeof(int32_t), hipMemcpyHostToDevice);
    hipMemcpy(s.rep_desc, desc.data(), desc.size() * sizeof(int4), hipMemcpyHostToDevice);
    hipMemcpy(s.rep_codes_tile, ct.data(), ct.size() * sizeof(int16_t), hipMemcpyHostToDevice);
    hipMemcpy(s.rep_leaf, lt.data(), lt.size() * sizeof(int2), hipMemcpyHostToDevice);
    s.rep_leaf_host = lt;
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) return fail("repeats: device initialisation failed");
  }
  // the trunk as one post-order walk per tile (trunk_walk_kernel): plan_trunk_walk below
  p->rep_walk_host.clear();
  if (!p->nuc && p->NW >= 2 && v.L <= 1024) {
    const WalkPlan wp = plan_trunk_walk(v);
    const std::vector<WalkPlan::Node> &nodes = wp.nodes;
    const std::vector<int> &inputs = wp.inputs;
    const std::vector<int> &one_range = wp.one_range, &two_range = wp.two_range;
    const int max_depth = wp.max_depth;
    const bool two = wp.two;
    if (max_depth <= kWalkDepth && nodes.size() < 4096 && !inputs.empty()) {
      for (Shard &s : p->shards) {
        const std::vector<int2> &lt = s.rep_leaf_host;
        // [0] header, one-chain form (1 + 1 words), two-chain form (1 + 2 words), node words, input words
        const int form1 = 1, form2 = two ? 3 : 0, node_base = two ? 6 : 3, in_base = node_base + (int)nodes.size();
        std::vector<int4> prog((size_t)in_base + inputs.size());
        prog[0] = make_int4(form1, form2, max_depth, (int)prog.size());
        prog[form1] = make_int4(1, 0, 0, 0);
        prog[form1 + 1] = make_int4(node_base + one_range[0], one_range[1] - one_range[0], 0, 0);
        if (two) {
          prog[form2] = make_int4(2, 0, 0, 0);
          const int n0 = two_range[1] - two_range[0], n1 = two_range[3] - two_range[2];
          // (.z: wave priority — the longer chain's instructions first wherever the two share a SIMD: 59.5 against 60.4 us of pruning
          //  launches at the headline; the same by walk length in the lower phase: no difference, not kept)
          prog[form2 + 1] = make_int4(node_base + two_range[0], n0, n0 > n1 ? 3 : 0, 0);
          prog[form2 + 2] = make_int4(node_base + two_range[2], n1, n1 > n0 ? 3 : 0, 0);
        }
        for (size_t k = 0; k < nodes.size(); k++)
          prog[node_base + k] = make_int4(nodes[k].node < 0 ? -1 : v.slot[v.L + nodes[k].node], nodes[k].n_in, in_base + nodes[k].in0, nodes[k].flags);
        for (size_t k = 0; k < inputs.size(); k++) prog[in_base + k] = make_int4(inputs[k], lt[inputs[k]].x, lt[inputs[k]].y, 0);
        if (&s == &p->shards[0]) {
          p->rep_walk_host = prog;
          if (getenv("HYPHY_HIP_VERBOSE")) {
            fprintf(stderr, "[hyphy_hip] trunk walk: %d nodes, waiting products at most %d deep", one_range[1] - one_range[0], max_depth);
            if (two) fprintf(stderr, "; two chains per tile: %d + %d nodes", two_range[1] - two_range[0], two_range[3] - two_range[2]);
            fprintf(stderr, "\n");
          }
        }
        if (hipSetDevice(s.device) != hipSuccess) return fail("hipSetDevice failed");
        const size_t bytes = prog.size() * sizeof(int4);
        if (pool_malloc((void **)&s.rep_walk, bytes) != hipSuccess) return fail("hipMalloc failed (s.rep_walk)");
        s.dev_bytes += bytes;
        s.rep_bytes += bytes;
        hipMemcpy(s.rep_walk, prog.data(), bytes, hipMemcpyHostToDevice);
      }
    }
  }
  p->rep_resident.assign(p->C, 0);
  p->rep_cached_valid = false;
  // what the trunk's schedules start from (the tuner refines them): the wave-per-tile kernel, no re-rooting
  hyphy_hip_partition::ModeState &ms = p->saved_mode[1];
  ms.variant = 1;
  ms.wave_variant = 0;
  ms.n_slots = p->n_slots_wave;
  ms.chain_m_forced = 0;
  ms.rr_use = false;
  ms.kernel_forced = true;
  ms.tuned_for = 0;
  ms.rr_path.clear();
  ms.rr_cands.clear();
  // (r06: the trunk under the row-split workgroup kernel, prune_mfma_kernel<.., REP> — the tuner offers it on shards below eight tiles
  //  per CU; HYPHY_HIP_TRUNK_KERNEL=0 / 2 forces it — whole trunk per workgroup / on chain schedules — for tests and A/B runs)
  if (const char *tk = getenv("HYPHY_HIP_TRUNK_KERNEL")) {
    const int kv = atoi(tk);
    const hyphy_hip_partition::View &tv = p->views[1];
    if ((kv == 0 || kv == 2) && p->NW == 4 && !p->nuc && (size_t)tv.L * 32 + (size_t)(tv.L + tv.I) * 16 <= 24576) {
      ms.variant = kv;
      ms.n_slots = lds_slots(1);
    }
  }
  ms.trunk_walk = false;
  {
    // HYPHY_HIP_TRUNK_WALK=1: the walk without asking the tuner (tests, A/B runs).  No tuner at all (HYPHY_HIP_TUNE=0): the walk too —
    // it is the faster form wherever the static rule turns compression on, and with the one-workgroup-per-tile kernel behind it for
    // the other passes nothing in this mode depends on an order of arrival: the same bits on every run (tested)
    const char *tw = getenv("HYPHY_HIP_TRUNK_WALK");
    const bool tune_off = getenv("HYPHY_HIP_TUNE") && atoi(getenv("HYPHY_HIP_TUNE")) == 0;
    const hyphy_hip_partition::View &tv = p->views[1];
    if ((tw ? atoi(tw) == 1 : tune_off) && p->NW == 4 && !p->rep_walk_host.empty() && (size_t)tv.L * 32 + (size_t)(tv.L + tv.I) * 16 <= 24576) {
      ms.trunk_walk = true;
      ms.variant = 0;
      ms.n_slots = lds_slots(1);
    }
  }
  if (p->C == 1 && !p->nuc) {  // the trunk's own height-minimising roots (the tuner's third stage times the re-rooted schedules)
    const int mode0 = p->mode;
    std::vector<int> path0;
    std::vector<std::vector<int>> cands0;
    path0.swap(p->rr_path);
    cands0.swap(p->rr_cands);
    p->mode = 1;
    reroot_path(p);
    ms.rr_path.swap(p->rr_path);
    ms.rr_cands.swap(p->rr_cands);
    p->mode = mode0;
    p->rr_path.swap(path0);
    p->rr_cands.swap(cands0);
  }
  p->rep_on = true;
  if (getenv("HYPHY_HIP_VERBOSE")) {
    long long rows = 0, lower = 0;
    for (int d = 0; d < ND; d++) {
      rows += p->shards[0].rep_tabs[d].rows;
      lower += (long long)p->shards[0].rep_tabs[d].rows * std::max<size_t>(1, p->rep_nodes[d].path.size());
    }
    fprintf(stderr, "[hyphy_hip] subtree repeats: theta %.2f, rho %.2f, %d class tables (%lld rows on shard 0), trunk of %d internal nodes over %d leaves; "
                    "edge products per pass %lld + %lld (every pattern at every node: %lld)\n",
            theta, rho, ND, rows, v.I, v.L, lower, (long long)(v.I - 1) * p->shards[0].S_pad, (long long)(I - 1) * p->shards[0].S_pad);
  }
  return 0;
}
}  // namespace

// The view in use: saves what the tuner decided for the view that is left and restores the other's.
void switch_mode(hyphy_hip_partition *p, int mode) {
  if (p->mode == mode) return;
  hyphy_hip_partition::ModeState &out = p->saved_mode[p->mode];
  out.variant = p->variant;
  out.wave_variant = p->wave_variant;
  out.n_slots = p->n_slots;
  out.chain_m_forced = p->chain_m_forced;
  out.rr_use = p->rr_use;
  out.kernel_forced = p->kernel_forced;
  out.nuc_leaf_pairs = p->nuc_leaf_pairs;
  out.trunk_walk = p->trunk_walk;
  out.tuned_for = p->tuned_for;
  out.tune_report = p->tune_report;
  out.rr_path = p->rr_path;
  out.rr_cands = p->rr_cands;
  const hyphy_hip_partition::ModeState &in = p->saved_mode[mode];
  p->variant = in.variant;
  p->wave_variant = in.wave_variant;
  p->n_slots = in.n_slots;
  p->chain_m_forced = in.chain_m_forced;
  p->rr_use = in.rr_use;
  p->kernel_forced = in.kernel_forced;
  if (p->nuc) p->nuc_leaf_pairs = in.nuc_leaf_pairs;
  p->trunk_walk = in.trunk_walk;
  p->tuned_for = in.tuned_for;
  p->tune_report = in.tune_report;
  p->rr_path = in.rr_path;
  p->rr_cands = in.rr_cands;
  p->mode = mode;
  p->cached_valid = 0;
  p->rr_active = false;
  for (Shard &sh : p->shards) sh.twins_dirty = true;  // (each view's re-rooted schedules reverse their own edges)
  std::fill(p->last_full.begin(), p->last_full.end(), 0);  // (the first full pass under a view stores every node it keeps)
}

// Translate the caller's update list (node codes of the partition's tree) for a pass under views[1]: the descriptors whose
// tables must be recomputed (children before parents) and the update list of the trunk (view node codes).  A table E_n =
// P_n x (conditionals below n) is stale when n's conditionals are (n is the parent of a listed node, or an ancestor of one)
// or when its own branch's matrix changed (n among q_nodes).
void rep_translate_update(hyphy_hip_partition *p, const int64_t *update_nodes, int64_t n_update, const int64_t *q_nodes, int64_t n_q,
                          bool full, std::vector<int> &dirty_desc, std::vector<int64_t> &view_update) {
  const int L = (int)p->L, I = (int)p->I;
  const int ND = (int)p->rep_nodes.size();
  dirty_desc.clear();
  view_update.clear();
  if (full) {
    for (int d = 0; d < ND; d++) dirty_desc.push_back(d);
    return;
  }
  std::vector<char> touched(I, 0), qd((size_t)L + I, 0);
  for (int64_t k = 0; k < n_update; k++) {
    const int64_t n = update_nodes[k];
    if (n < 0 || n >= L + I) continue;
    int64_t par = p->parents[n];
    while (par >= 0 && !touched[par]) {
      touched[par] = 1;
      par = p->parents[L + par];
    }
  }
  for (int64_t k = 0; k < n_q; k++)
    if (q_nodes[k] >= 0 && q_nodes[k] < L + I) qd[q_nodes[k]] = 1;
  if (p->rep_stale_branch >= 0 && p->rep_stale_branch < L + I) qd[p->rep_stale_branch] = 1;
  const hyphy_hip_partition::View &v = p->views[1];
  for (int d = 0; d < ND; d++) {
    const hyphy_hip_partition::RepNode &rn = p->rep_nodes[d];
    bool dirty = rn.node < L && qd[rn.node];
    for (int n : rn.path) dirty = dirty || qd[n] || touched[n - L];
    if (dirty) dirty_desc.push_back(d);
  }
  // the trunk: list the view nodes whose parents must be recomputed — a trunk node that is touched lists itself through any of
  // its view children; simplest exact form: every view node whose parent (a trunk node) is touched in the partition's tree
  for (int n = 0; n < v.L + v.I; n++) {
    const int64_t par = v.parents[n];
    if (par < 0) continue;
    const int pn = v.slot[v.L + par] - L;  // the parent's internal index in the partition's tree
    if (touched[pn]) view_update.push_back(n);
  }
}

// Item queues of a pass over the descriptors `dirty` for `n_classes` rate classes starting at `cat0`.  The items form ONE sequence,
// dealt round-robin over the queues (item at position g: queue g % 32, index g / 32), in an order in which every table comes before
// the paths that read it — by descending PRIORITY = products of the path + the longest chain of paths that wait for it: the side
// tables of the long paths first, then the long paths (they are what the launch waits for at its end), short independent ones last
// to fill the gaps.  An item's `bound` = the queue index every head must have passed for all its inputs to be sold.
// Returns items per queue.
// Waves of a lower-phase launch over n items: one per SIMD while the items fit two rounds of that (a walk is bound by the SIMD's
// matrix pipe: two walks on one SIMD take twice as long each, and the launch ends with its longest walk), two per SIMD beyond.
static int rep_wave_count(const Shard &s, size_t n_items) {
  const int n = (int)((n_items + kRepQueues - 1) / kRepQueues * kRepQueues);
  int w = n <= 8 * s.cus ? std::min(n, s.cus * 4) : s.cus * 8;
  if (const char *e = getenv("HYPHY_HIP_REP_WAVES")) w = std::max(1, std::min(n, atoi(e)));
  return std::max(1, w);
}

int rep_build_items(const hyphy_hip_partition *p, const Shard &s, const std::vector<int> &dirty, int cat0, int n_classes,
                    std::vector<int4> &queues /* [kRepQueues][qcap] */, int *n_static = nullptr, int *n_waves = nullptr,
                    std::vector<RepLaunch> *launches = nullptr, bool team = false) {
  const int ND = (int)p->rep_nodes.size();
  std::vector<char> live(ND, 0);
  for (int d : dirty) live[d] = 1;
  std::vector<long> prio(ND, 0);
  for (int d = ND - 1; d >= 0; d--) {  // (descriptors are stored inputs first: consumers have larger numbers)
    if (!live[d]) continue;
    prio[d] += (long)std::max<size_t>(1, p->rep_nodes[d].path.size());
    for (const std::vector<int> &kk : p->rep_nodes[d].kid_desc)
      for (int dd : kk)
        if (dd >= 0 && live[dd]) prio[dd] = std::max(prio[dd], prio[d]);  // (so far: the longest chain above dd; its own length is added when its turn comes)
  }
  std::vector<int> order(dirty);
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return prio[x] > prio[y]; });
  std::vector<int4> all;
  std::vector<long> pos0((size_t)ND * n_classes, -1);
  for (int d : order)
    for (int c = 0; c < n_classes; c++) {
      long last_dep = -1;
      for (const std::vector<int> &kk : p->rep_nodes[d].kid_desc)
        for (int dd : kk)
          if (dd >= 0 && live[dd]) last_dep = std::max(last_dep, pos0[(size_t)dd * n_classes + c] + s.rep_tabs[dd].rows / 16 - 1);
      const int bound = last_dep < 0 ? 0 : (int)(last_dep / kRepQueues) + 1;
      pos0[(size_t)d * n_classes + c] = (long)all.size();
      for (int t = 0; t < s.rep_tabs[d].rows / 16; t++) all.push_back(make_int4(d, t | (c << 20), (int)all.size(), bound));
    }
  if (n_static) {  // no item waits for another one of this pass?
    *n_static = (int)all.size();
    for (const int4 &it : all)
      if (it.w > 0) *n_static = 0;
  }
  const int nw = rep_wave_count(s, all.size());
  if (n_waves) *n_waves = nw;
  const bool lpt_on = !(getenv("HYPHY_HIP_REP_LPT") && atoi(getenv("HYPHY_HIP_REP_LPT")) == 0);
  // Items dealt by position: wave w takes positions w, 2 w_n - 1 - w, 2 w_n + w, ... (class_table_kernel).  Which item sits where
  // decides when the launch ends: 1 519 walks of 8-24 us on 1 024 waves are two rounds at most, and in order of descriptor
  // priority the waves whose first walk was a 15-us one got a second 15-us one (31.6 us) while 23-us walks sat alone and 16-us
  // ones too.  Longest-processing-time-first instead: items in descending order of estimated cost (k cycles per walk, fitted to the
  // kernel's timeline at the headline: 8.5 + 7.0 per walked node — walks of 6 / 4 / 3 nodes took 50.5 / 36.5 / 27.5), each to the
  // wave with the least so far.  Returns the positions (padding items in the gaps).
  std::vector<double> cost(ND, 0.);
  for (int d : dirty) cost[d] = 8.5 + 7.0 * (double)std::max<size_t>(1, p->rep_nodes[d].path.size());
  auto place = [&](std::vector<int4> &its, int w_n, double *longest) {
    std::vector<int> by_cost(its.size());
    for (size_t i = 0; i < its.size(); i++) by_cost[i] = (int)i;
    std::stable_sort(by_cost.begin(), by_cost.end(), [&](int x, int y) { return cost[its[x].x] > cost[its[y].x]; });
    std::vector<std::vector<int>> mine(w_n);
    std::vector<std::pair<double, int>> heap;  // (load so far, wave): min-heap
    for (int w = 0; w < w_n; w++) heap.push_back(std::make_pair(0., w));
    auto later = [](const std::pair<double, int> &x, const std::pair<double, int> &y) { return x > y; };
    std::make_heap(heap.begin(), heap.end(), later);
    for (int i : by_cost) {
      std::pop_heap(heap.begin(), heap.end(), later);
      std::pair<double, int> &top = heap.back();
      mine[top.second].push_back(i);
      top.first += cost[its[i].x];
      std::push_heap(heap.begin(), heap.end(), later);
    }
    size_t rounds = 0;
    for (const std::vector<int> &m : mine) rounds = std::max(rounds, m.size());
    std::vector<int4> placed(rounds * (size_t)w_n);
    for (size_t i = 0; i < placed.size(); i++) placed[i] = make_int4(-1, 0, (int)i, 0);
    for (int w = 0; w < w_n; w++)
      for (size_t r = 0; r < mine[w].size(); r++) {
        const size_t pos = r * (size_t)w_n + ((r & 1) ? (size_t)(w_n - 1 - w) : (size_t)w);
        placed[pos] = its[mine[w][r]];
        placed[pos].z = (int)pos;
        placed[pos].w = 0;
      }
    if (longest) {
      *longest = 0.;
      for (const std::pair<double, int> &h : heap) *longest = std::max(*longest, h.first);
    }
    its.swap(placed);
  };
  // the row-split walk (class_table_team_kernel): one workgroup per item, dispatched in the order of the list — descending cost
  auto by_cost_desc = [&](std::vector<int4> &its) {
    std::stable_sort(its.begin(), its.end(), [&](const int4 &x, const int4 &y) { return cost[x.x] > cost[y.x]; });
    for (size_t i = 0; i < its.size(); i++) its[i].z = (int)i, its[i].w = 0;
  };
  const bool verbose2 = getenv("HYPHY_HIP_VERBOSE") && atoi(getenv("HYPHY_HIP_VERBOSE")) >= 2;
  if (launches) launches->clear();
  const bool levels_on = launches && !(getenv("HYPHY_HIP_REP_LEVELS") && atoi(getenv("HYPHY_HIP_REP_LEVELS")) == 0) &&
                         !(getenv("HYPHY_HIP_REP_STATIC") && atoi(getenv("HYPHY_HIP_REP_STATIC")) == 0);
  if (n_static && *n_static == 0 && !all.empty() && levels_on) {
    // Tables that read tables of the same pass: ONE LAUNCH PER LEVEL of that dependency (level 0: no table of this pass among the
    // inputs), each dealt by position like a pass without dependencies — the launch boundary is the hand-off (2-3 us for everybody at
    // once) where the ticket protocol paid a drain, a counter and the polling of a thousand waiting waves per table (~6 us each).
    std::vector<int> lvl(ND, 0);
    int n_lvl = 1;
    for (int d = 0; d < ND; d++) {  // (descriptors are stored inputs first)
      if (!live[d]) continue;
      for (const std::vector<int> &kk : p->rep_nodes[d].kid_desc)
        for (int dd : kk)
          if (dd >= 0 && live[dd]) lvl[d] = std::max(lvl[d], lvl[dd] + 1);
      n_lvl = std::max(n_lvl, lvl[d] + 1);
    }
    queues.clear();
    for (int lv = 0; lv < n_lvl; lv++) {
      std::vector<int4> part;
      for (const int4 &it : all)
        if (lvl[it.x] == lv) part.push_back(make_int4(it.x, it.y, (int)part.size(), 0));
      if (part.empty()) continue;
      const int w_n = rep_wave_count(s, part.size());
      double longest = 0.;
      if (team) by_cost_desc(part);
      else if ((int)part.size() > w_n && lpt_on) place(part, w_n, &longest);
      const int n_pos = (int)part.size();
      while (part.size() % kRepQueues) part.push_back(make_int4(-1, 0, (int)part.size(), 0));
      const int pq = (int)part.size() / kRepQueues;
      const size_t off = queues.size();
      queues.resize(off + part.size(), make_int4(-1, 0, 0, 0));
      for (size_t i = 0; i < part.size(); i++) queues[off + (i % kRepQueues) * (size_t)pq + i / kRepQueues] = part[i];
      launches->push_back(RepLaunch{off, pq, n_pos, w_n});
      if (verbose2) fprintf(stderr, "[hyphy_hip] lower phase, level %d: %d positions on %d waves (longest estimated wave %.1f k cycles)\n", lv, n_pos, w_n, longest);
    }
    *n_static = launches->empty() ? 0 : (*launches)[0].n_static;
    if (n_waves && !launches->empty()) *n_waves = (*launches)[0].n_waves;
    return (int)(queues.size() / kRepQueues);
  }
  if (n_static && *n_static > 0 && team) {
    by_cost_desc(all);
  } else if (n_static && *n_static > nw && lpt_on) {
    double longest = 0.;
    place(all, nw, &longest);
    if (verbose2) {
      for (int d : dirty) fprintf(stderr, "[hyphy_hip] lower phase: descriptor %d walks %zu nodes, %d tiles, cost %.1f\n", d, p->rep_nodes[d].path.size(), s.rep_tabs[d].rows / 16, cost[d]);
      fprintf(stderr, "[hyphy_hip] lower phase: %zu positions on %d waves, longest estimated wave %.1f k cycles\n", all.size(), nw, longest);
    }
    *n_static = (int)all.size();
  }
  while (all.size() % kRepQueues) all.push_back(make_int4(-1, 0, (int)all.size(), 0));
  const int per_q = (int)all.size() / kRepQueues;
  queues.assign((size_t)kRepQueues * std::max(1, per_q), make_int4(-1, 0, 0, 0));
  for (size_t i = 0; i < all.size(); i++) queues[(i % kRepQueues) * (size_t)std::max(1, per_q) + i / kRepQueues] = all[i];
  return per_q;
}

// Everything a pass under views[1] needs on the device besides the trunk's schedule: the item queues of the tables that are
// stale (uploaded when they differ from the ones the device holds) — and the trunk's update list for the schedule compiler.
int rep_prepare_pass(hyphy_hip_partition *p, const int64_t *update_nodes, int64_t n_update, const int64_t *q_nodes, int64_t n_q, bool full,
                     int cat0, int n_classes, std::vector<int64_t> &view_update) {
  std::vector<int> dirty;
  rep_translate_update(p, update_nodes, n_update, q_nodes, n_q, full, dirty, view_update);
  p->rep_stale_branch = -1;
  const int pass_key = n_classes * 65536;  // (the lists do not name the first class: a host that evaluates its rate classes one
                                           //  ComputeBlock at a time — the reference's category loop — re-uses one list for all of them)
  if (!p->rep_cached_valid) {
    for (auto &k : p->rep_pass) k.valid = false;
    p->rep_cached_valid = true;
  }
  // (a slot's event says "everything that read these queues has run".  It used to be recorded behind every lower-phase launch: a marker
  //  between the class tables' kernel and the trunk's — 5.6 us of every evaluation by the kernel trace.  Now: when the pass leaves the
  //  slot, or when the slot is about to be rewritten — steady-state evaluations record nothing.)
  auto leave = [&](Shard &s, RepPassSlot &ps) -> int {
    if (ps.used && ps.ev) {
      HIPCHK(hipSetDevice(s.device));
      HIPCHK(hipEventRecord(ps.ev, s.stream));
      ps.ev_recorded = true;
    }
    ps.used = false;
    return 0;
  };
  auto activate = [&](int slot) {
    for (Shard &s : p->shards) {
      if (s.rep_slot_cur >= 0 && s.rep_slot_cur != slot) leave(s, s.rep_slot[s.rep_slot_cur]);
      RepPassSlot &ps = s.rep_slot[slot];
      s.rep_slot_cur = slot;
      s.rep_items = ps.items;
      s.rep_qcap = ps.qcap;
      s.rep_waves = ps.waves;
      s.rep_static = ps.n_static;
      s.rep_team = ps.team;
      s.rep_launches = ps.launches;
    }
    p->rep_pass[slot].stamp = ++p->rep_pass_clock;
  };
  for (int k = 0; k < kRepPassSlots; k++)
    if (p->rep_pass[k].valid && p->rep_pass[k].classes == pass_key && p->rep_pass[k].dirty == dirty) {
      activate(k);  // (a dirty set seen before: its queues are still on the device)
      return 0;
    }
  int slot = 0;  // an unused slot, else the least recently used one
  for (int k = 1; k < kRepPassSlots; k++)
    if (!p->rep_pass[k].valid ? p->rep_pass[slot].valid : (p->rep_pass[slot].valid && p->rep_pass[k].stamp < p->rep_pass[slot].stamp)) slot = k;
  p->rep_pass[slot].valid = false;
  const int ND = (int)p->rep_nodes.size();
  for (Shard &s : p->shards) {
    HIPCHK(hipSetDevice(s.device));
    RepPassSlot &ps = s.rep_slot[slot];
    if (!ps.ev) HIPCHK(hipEventCreateWithFlags(&ps.ev, hipEventDisableTiming));
    if (leave(s, ps)) return -1;
    if (ps.ev_recorded) {  // (the launches that read this slot last — and the copy that filled its staging block — have they run?)
      HIPCHK(hipEventSynchronize(ps.ev));
      ps.ev_recorded = false;
    }
    auto ensure = [&](size_t words, size_t cap_wanted) -> int {
      if (words <= ps.cap) return 0;
      if (ps.items) pool_free_sync(ps.items);
      if (ps.h_items) pool_host_free(ps.h_items);
      ps.items = nullptr;
      ps.h_items = nullptr;
      ps.cap = 0;
      const size_t cap = std::max(words, cap_wanted);
      HIPCHK(pool_malloc((void **)&ps.items, cap * sizeof(int4)));
      HIPCHK(pool_host_malloc((void **)&ps.h_items, cap * sizeof(int4)));
      ps.cap = cap;
      return 0;
    };
    if (p->nuc) {  // 4 states: the list of subtrees to walk (no items, no queues: one thread per class, no table reads another)
      const size_t words = std::max<size_t>(1, (dirty.size() + 3) / 4);
      if (ensure(words, ((size_t)ND + 3) / 4)) return -1;
      int *lst = reinterpret_cast<int *>(ps.h_items);
      for (size_t k = 0; k < dirty.size(); k++) lst[k] = dirty[k];
      if (!dirty.empty()) HIPCHK(hipMemcpyAsync(ps.items, ps.h_items, words * sizeof(int4), hipMemcpyHostToDevice, s.stream));
      ps.qcap = (int)dirty.size();
      ps.waves = 0;
      ps.n_static = 0;
      ps.team = false;
      ps.launches.clear();
      continue;
    }
    std::vector<int4> queues;
    int n_static = 0, n_waves = 1;
    ps.team = p->NW >= 2 && !(getenv("HYPHY_HIP_REP_TEAM") && atoi(getenv("HYPHY_HIP_REP_TEAM")) == 0) &&
              !(getenv("HYPHY_HIP_REP_STATIC") && atoi(getenv("HYPHY_HIP_REP_STATIC")) == 0);
    ps.launches.clear();
    const int per_q = dirty.empty() ? 0 : rep_build_items(p, s, dirty, cat0, n_classes, queues, &n_static, &n_waves, &ps.launches, ps.team);
    ps.n_static = getenv("HYPHY_HIP_REP_STATIC") && atoi(getenv("HYPHY_HIP_REP_STATIC")) == 0 ? 0 : n_static;
    const size_t words = (size_t)kRepQueues * std::max(1, per_q) + (size_t)(ND + 3) / 4;  // queues, then the live flags
    {
      // (sized for a full pass of every class: partial passes never need more)
      size_t all = 0;
      for (const RepTable &t : s.rep_tabs) all += (size_t)t.rows / 16;
      if (ensure(words, (all * (size_t)p->C + (size_t)kRepQueues * (ND + 2)) + (size_t)(ND + 3) / 4 + kRepQueues)) return -1;
    }
    if (per_q > 0) memcpy(ps.h_items, queues.data(), (size_t)kRepQueues * per_q * sizeof(int4));
    int *live = reinterpret_cast<int *>(ps.h_items + (size_t)kRepQueues * std::max(1, per_q));
    for (int d = 0; d < ND; d++) live[d] = 0;
    for (int d : dirty) live[d] = getenv("HYPHY_HIP_REP_NODEPS") ? 0 : 1;  // (diagnostic: nobody waits, results invalid)
    HIPCHK(hipMemcpyAsync(ps.items, ps.h_items, words * sizeof(int4), hipMemcpyHostToDevice, s.stream));
    HIPCHK(hipEventRecord(ps.ev, s.stream));  // (re-recorded behind every launch that reads the slot: rep_launch)
    ps.ev_recorded = true;
    ps.qcap = per_q;
    ps.waves = n_waves;  // (rep_wave_count)
  }
  p->rep_pass[slot].dirty = dirty;
  p->rep_pass[slot].classes = pass_key;
  p->rep_pass[slot].valid = true;
  activate(slot);
  return 0;
}

size_t rep_sync_words(const hyphy_hip_partition *p) { return (size_t)kRepQueues + 1 + (size_t)p->C * p->rep_nodes.size(); }
int rep_sync_stride() { return kRepHeadStride; }

// The lower phase of a pass: one launch over the item queues prepared by rep_prepare_pass (behind the exponentials, ahead of
// the trunk's pruning launch, on the shard's stream).
static int rep_launch_impl(hyphy_hip_partition *p, Shard &s, int cat0);
int rep_launch(hyphy_hip_partition *p, Shard &s, int cat0) {
  const int rc = rep_launch_impl(p, s, cat0);
  if (rc == 0 && s.rep_qcap > 0 && s.rep_slot_cur >= 0) s.rep_slot[s.rep_slot_cur].used = true;  // (rep_prepare_pass: the slot's event)
  return rc;
}
static int rep_launch_impl(hyphy_hip_partition *p, Shard &s, int cat0) {
  if (s.rep_qcap <= 0) return 0;
  if (p->nuc) {
    RepNucArgs a;
    a.map = s.rep_map;
    a.tab = s.rep_tab + (size_t)cat0 * s.rep_rows * 4;
    a.cnt = s.rep_cnt + (size_t)cat0 * s.rep_rows;
    a.P = s.Prow + (size_t)cat0 * p->B * 16;
    a.ambig = s.ambig;
    int max_rows = 16;
    for (const RepTable &t : s.rep_tabs) max_rows = std::max(max_rows, t.rows);
    hipLaunchKernelGGL(class_table_nuc_kernel, dim3((max_rows + 127) / 128, s.rep_qcap), dim3(128), 0, s.stream, (const int4 *)s.rep_desc,
                       (const int *)s.rep_items, a);
    return 0;
  }
  RepArgs a;
  a.cat0 = cat0;
  a.desc = s.rep_desc;
  a.items = s.rep_items;
  a.qcap = s.rep_qcap;
  a.live = reinterpret_cast<const int *>(s.rep_items + (size_t)kRepQueues * s.rep_qcap);
  a.n_desc = (int)p->rep_nodes.size();
  a.sync = s.rep_sync;
  a.tab = s.rep_tab;
  a.cnt = s.rep_cnt;
  a.map = s.rep_map;
  a.rows = s.rep_rows;
  a.Pfrag = s.Pfrag;
  a.PTg = s.PTg;
  a.cs_P = (size_t)p->B * p->DP * p->DP;
  a.ambig = s.ambig;
  a.n_waves = s.rep_waves;
  a.n_static = s.rep_static;
  a.dbg = nullptr;
  // queue heads and counters are zero between launches.  A launch whose items are dealt by position (n_static) never touches
  // them; behind any other one they are cleared in front of the next launch (a memset node on the stream: 2-3 us in front of passes
  // that take hundreds — class tables that read class tables of the same pass, rho > 0 or partial updates)
  if (s.rep_sync_dirty) HIPCHK(hipMemsetAsync(s.rep_sync, 0, rep_sync_words(p) * kRepHeadStride * sizeof(int), s.stream));
  s.rep_sync_dirty = a.n_static == 0 && s.rep_launches.empty();
  const char *tl = getenv("HYPHY_HIP_REP_TIMELINE");
  if (tl && p->NW == 4) {  // diagnostic: synchronous, one file per launch (the last launch survives)
    size_t traced = s.rep_launches.size();  // (several launches per pass: the LAST level is the traced one, the others run plain before it)
    for (size_t k = 0; k < s.rep_launches.size(); k++) {
      const RepLaunch &lv = s.rep_launches[k];
      a.items = s.rep_items + lv.off;
      a.qcap = lv.qcap;
      a.n_static = lv.n_static;
      a.n_waves = lv.n_waves;
      if (k + 1 < traced) launch_class_tables(a, p->NW, s.stream, s.rep_team);
    }
    const bool team_tr = s.rep_team && a.n_static > 0;
    const size_t n_rec = team_tr ? (size_t)a.n_static : (size_t)a.n_waves;
    const size_t n = n_rec * 16;
    HIPCHK(pool_malloc((void **)&a.dbg, n * sizeof(long long)));
    HIPCHK(hipMemsetAsync(a.dbg, 0, n * sizeof(long long), s.stream));
    launch_class_tables(a, p->NW, s.stream, s.rep_team);
    std::vector<long long> h(n);
    HIPCHK(hipStreamSynchronize(s.stream));
    HIPCHK(hipMemcpy(h.data(), a.dbg, n * sizeof(long long), hipMemcpyDeviceToHost));
    pool_free_sync(a.dbg);
    if (FILE *f = fopen(tl, "w")) {
      if (team_tr) fprintf(f, "# team (one workgroup of NW waves per item, class_table_team_kernel) wall_start wall_end(100MHz) walked_nodes hw_id|xcc<<32 cycles: index-maps gathers exchange product publish\n");
      else fprintf(f, "# wave wall_start wall_end(100MHz) items failed_polls cycles: tickets waiting gathers product publish\n");
      for (int w = 0; w < (int)n_rec; w++) {
        const long long *r = &h[(size_t)w * 16];
        fprintf(f, "%d %lld %lld %lld %lld %lld %lld %lld %lld %lld\n", w, r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8]);
      }
      fclose(f);
    }
    return 0;
  }
  if (!s.rep_launches.empty()) {  // one launch per level of table-reads-table dependencies (rep_build_items), each dealt by position
    for (const RepLaunch &lv : s.rep_launches) {
      a.items = s.rep_items + lv.off;
      a.qcap = lv.qcap;
      a.n_static = lv.n_static;
      a.n_waves = lv.n_waves;
      launch_class_tables(a, p->NW, s.stream, s.rep_team);
    }
    return 0;
  }
  launch_class_tables(a, p->NW, s.stream, s.rep_team);
  return 0;
}

// ---- the trunk's lazy full pass as one row-split walk per tile (trunk_walk_kernel) ----
static bool trunk_walk_allowed() {  // (per call: tests and A/B runs switch it)
  const char *e = getenv("HYPHY_HIP_TRUNK_WALK");
  return !(e && atoi(e) == 0);
}
bool trunk_walk_applies(const hyphy_hip_partition *p, const Shard &s) {
  return p->mode == 1 && p->trunk_walk && !p->nuc && s.rep_walk && s.T == 1 && p->sched_full && !p->sched_persist && p->pin_node < 0 &&
         (size_t)p->views[1].L * 32 <= 32768 && trunk_walk_allowed();
}
bool trunk_walk_fuses_reduce(const hyphy_hip_partition *p) { return p->NW == 4; }
int launch_trunk_walk(hyphy_hip_partition *p, Shard &s, int cat, int n_cat_batch, bool timeline, const PruneArgs *red) {
  const int DP = p->DP, n_cat = std::max(1, n_cat_batch);
  // two workgroups per tile where that still fits the chip in one round (five workgroups per CU); HYPHY_HIP_WALK_CHAINS=1/2 forces
  int chains = (p->rep_walk_host[0].y > 0 && (size_t)2 * s.ntiles * n_cat <= (size_t)5 * s.cus) ? 2 : 1;
  if (const char *e = getenv("HYPHY_HIP_WALK_CHAINS")) chains = (atoi(e) == 2 && p->rep_walk_host[0].y > 0) ? 2 : 1;
  TrunkWalkArgs a;
  a.Pfrag = s.Pfrag + (size_t)cat * p->B * DP * DP;
  a.PTg = s.PTg + (size_t)cat * p->B * DP * DP;
  a.cs_P = (size_t)p->B * DP * DP;
  a.gtab = s.rep_tab + (size_t)cat * s.rep_rows * DP;
  a.gcnt = s.rep_cnt + (size_t)cat * s.rep_rows;
  a.cs_gtab = (size_t)s.rep_rows * DP;
  a.cs_gcnt = (size_t)s.rep_rows;
  a.codes_tile = s.rep_codes_tile;
  a.L = p->views[1].L;
  a.form = chains == 2 ? p->rep_walk_host[0].y : p->rep_walk_host[0].x;
  a.pi = s.pi;
  a.site_lik = s.site_lik + (size_t)cat * s.S_pad;
  a.site_cnt = s.site_cnt + (size_t)cat * s.S_pad;
  a.cs_site = (size_t)s.S_pad;
  a.freq = s.freq;
  a.wg_sum = s.wg_sum;
  a.wg_cnt = s.wg_cnt;
  a.wg_flag = s.wg_flag;
  a.cs_wg = (size_t)s.ntiles;
  a.deposits = nullptr;
  a.cs_deposits = 0;
  a.hand_cnt = s.hand_cnt;
  a.arrivals = s.frag_ctr;
  a.ntiles = s.ntiles;
  a.red_out = nullptr, a.red_rec = nullptr, a.red_status = nullptr, a.red_seq = 0., a.red_done = nullptr, a.red_n = 0;
  if (red && red->red_out && n_cat == 1) {
    a.red_out = red->red_out, a.red_rec = red->red_rec, a.red_status = red->red_status, a.red_seq = red->red_seq;
    a.red_done = red->red_done, a.red_n = s.ntiles;
  }
  a.dbg = nullptr;
  if (chains == 2) {  // (the buffers of the pruning kernels' chain schedules: [class][node][tile] tiles, exponents, arrival counters)
    if (ensure_deposits(p, s)) return -1;
    if (s.deposits_cap < (size_t)p->C * 2 * s.ntiles * 16 * DP) return fail("internal: trunk walk: deposits too small");
    a.deposits = s.deposits;
  }
  const dim3 grid(s.ntiles, n_cat, chains), block(64 * p->NW);
  const size_t lds = (size_t)a.L * 32;
  const char *tl = timeline ? getenv("HYPHY_HIP_WALK_TIMELINE") : nullptr;
  if (tl && p->NW == 4) {  // diagnostic: synchronous, one record per workgroup (the lower phase's format: tools/rep_team_timeline.py)
    const size_t n = (size_t)grid.x * grid.y * grid.z * 16;
    HIPCHK(pool_malloc((void **)&a.dbg, n * sizeof(long long)));
    HIPCHK(hipMemsetAsync(a.dbg, 0, n * sizeof(long long), s.stream));
    hipLaunchKernelGGL((trunk_walk_kernel<4, true>), grid, block, lds, s.stream, (const int4 *)s.rep_walk, a);
    std::vector<long long> h(n);
    HIPCHK(hipStreamSynchronize(s.stream));
    HIPCHK(hipMemcpy(h.data(), a.dbg, n * sizeof(long long), hipMemcpyDeviceToHost));
    pool_free_sync(a.dbg);
    if (FILE *f = fopen(tl, "w")) {
      fprintf(f, "# team (one workgroup of NW waves per tile and chain, trunk_walk_kernel) wall_start wall_end(100MHz) walked_nodes hw_id|xcc<<32 cycles: leaf-table gathers exchange product epilogue\n");
      for (size_t w = 0; w < n / 16; w++) {
        const long long *r = &h[w * 16];
        fprintf(f, "%zu %lld %lld %lld %lld %lld %lld %lld %lld %lld\n", w, r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8]);
      }
      fclose(f);
    }
    return 0;
  }
  // (measured, r06: an instantiation with ONE waiting product — 76 registers, six workgroups per CU — is not faster anywhere: 61.9 against
  //  60.5 us at the headline, 148 / 145 with three classes; two waiting products + two A chunks ahead, six per CU: 510 against 499 us at
  //  128 x 100 k.  One form.)
#define WALK_(nw, fuse) hipLaunchKernelGGL((trunk_walk_kernel<nw, false, fuse>), grid, block, lds, s.stream, (const int4 *)s.rep_walk, a)
  if (p->NW == 4) {
    if (a.red_out) WALK_(4, true);
    else WALK_(4, false);
  } else if (p->NW == 3) {
    WALK_(3, false);
  } else {
    WALK_(2, false);
  }
#undef WALK_
  return 0;
}

// Settle, by measurement, whether this partition's evaluations run class-compressed.  Called on the first steady-state full pass
// under views[1], behind the schedule tuner (which has just timed the trunk's pruning pass under its best cut): the lower phase is
// timed on the resident matrices (its items are on the device), then the plain form is tuned and timed the same way, and the faster
// of the two stays.  Compression costs a second launch and a chain of table hand-offs; on small shards (a rank's share of an
// alignment at 4 or 8 GPUs) that outweighs the edge products it saves.  What was decided for the same tree, shard and class
// batch earlier in this process is taken over without timing anything.
namespace {
//  the decision itself is kept per process — keyed like the tuner's choice: states, trunk, shard size, class batch, the
// tree AND the trunk's leaves (which depend on the alignment) — so that a second partition of the same shape does not time anything
struct RepDecisionKey {
  int64_t D, L, I, ntiles, classes;
  uint64_t topo;
  bool operator<(const RepDecisionKey &o) const {
    return std::tie(D, L, I, ntiles, classes, topo) < std::tie(o.D, o.L, o.I, o.ntiles, o.classes, o.topo);
  }
};
std::map<RepDecisionKey, std::pair<bool, std::string>> g_rep_decisions;
std::mutex g_rep_decisions_mutex;
RepDecisionKey rep_decision_key(const hyphy_hip_partition *p, int n_classes) {
  uint64_t topo = 1469598103934665603ull;
  for (int64_t v : p->views[1].parents) topo = (topo ^ (uint64_t)v) * 1099511628211ull;
  for (int sl : p->views[1].slot) topo = (topo ^ (uint64_t)sl) * 1099511628211ull;
  for (int64_t v : p->parents) topo = (topo ^ (uint64_t)v) * 1099511628211ull;
  return RepDecisionKey{p->D, p->L, p->I, p->shards[0].ntiles, n_classes, topo};
}
}  // namespace

// Without a measurement (HYPHY_HIP_TUNE=0, a forced cut): the static rule the measurements of r05 / r06 agree with — compression pays
// from ~128 tiles per shard upwards (64 x 2 500 and 32 x 5 k: on; 64 x 1 250, 79 tiles: off — a second launch outweighs the products
// it saves).  Called from hyphy_hip_create's side of things (api.hip) when no measurement will ever run.
bool rep_static_decision(const hyphy_hip_partition *p) { return !p->shards.empty() && p->shards[0].ntiles >= 128; }

int rep_decide(hyphy_hip_partition *p, int cat, int n_classes) {
  p->rep_decided = true;
  Shard &s = p->shards[0];
  HIPCHK(hipSetDevice(s.device));
  static const bool cache_on = !(getenv("HYPHY_HIP_TUNE_CACHE") && atoi(getenv("HYPHY_HIP_TUNE_CACHE")) == 0);
  const RepDecisionKey dkey = rep_decision_key(p, n_classes);
  if (cache_on) {
    std::lock_guard<std::mutex> lock(g_rep_decisions_mutex);
    auto hit = g_rep_decisions.find(dkey);
    if (hit != g_rep_decisions.end()) {
      p->rep_report = "(same tree, trunk, shard size and class batch as an earlier partition of this process) " + hit->second.second;
      if (getenv("HYPHY_HIP_VERBOSE")) fprintf(stderr, "[hyphy_hip] %s\n", p->rep_report.c_str());
      if (!hit->second.first) {
        p->rep_enabled = false;
        switch_mode(p, 0);
        p->cached_valid = 0;
      }
      return 0;
    }
  }
  const double trunk_ms = p->tuned_ms;
  float a_ms = 0.f, b_ms = 0.f;
  if (rep_launch(p, s, cat)) return -1;  // warm-up
  HIPCHK(hipEventRecord(s.ev[0], s.stream));
  if (rep_launch(p, s, cat)) return -1;
  HIPCHK(hipEventRecord(s.ev[1], s.stream));
  if (rep_launch(p, s, cat)) return -1;
  HIPCHK(hipEventRecord(s.ev[2], s.stream));
  HIPCHK(hipStreamSynchronize(s.stream));
  if (hipEventElapsedTime(&a_ms, s.ev[0], s.ev[1]) != hipSuccess || hipEventElapsedTime(&b_ms, s.ev[1], s.ev[2]) != hipSuccess) return 0;
  const double lower_ms = std::min(a_ms, b_ms);
  // the plain form under its own best schedule
  switch_mode(p, 0);
  p->cached_valid = 0;
  if (p->tuned_for != p->batch_classes) {
    if (tune_schedule(p, cat, n_classes)) return -1;
  } else {
    p->tuned_ms = 0.;  // (tuned earlier without a recorded time: keep compression on)
  }
  const double plain_ms = p->tuned_ms;
  const double on_ms = trunk_ms + lower_ms;
  // (the plain form has to win by 7 %: behind the exponential kernel of a real evaluation its launch runs slower than in these
  //  back-to-back passes, the two launches of the compressed form do not — 32 x 5 k: plain 44.2 us here, 50.3 in production;
  //  compressed 45.4 here, 44.4 in production — and at parity the measurement flipped from run to run)
  const bool off = plain_ms > 0. && plain_ms <= 0.93 * on_ms;
  char b[160];
  snprintf(b, sizeof b, "repeats: lower phase %.1fus + trunk %.1fus against %.1fus without -> %s", 1e3 * lower_ms, 1e3 * trunk_ms, 1e3 * plain_ms,
           off ? "off" : "on");
  p->rep_report = b;
  if (getenv("HYPHY_HIP_VERBOSE")) fprintf(stderr, "[hyphy_hip] %s\n", b);
  if (cache_on) {
    std::lock_guard<std::mutex> lock(g_rep_decisions_mutex);
    g_rep_decisions[dkey] = std::make_pair(!off, std::string(b));
  }
  if (off) {
    p->rep_enabled = false;  // (stays under views[0]; the caller rebuilds the schedule)
  } else {
    switch_mode(p, 1);
    for (Shard &sh : p->shards)  // (the plain form's tuning sized the chain deposits for every internal node: back to the trunk's)
      if (sh.deposits) {
        HIPCHK(hipSetDevice(sh.device));
        HIPCHK(hipStreamSynchronize(sh.stream));
        pool_free_sync(sh.deposits);
        sh.dev_bytes -= sh.deposits_cap * sizeof(double);
        sh.deposits = nullptr;
        sh.deposits_cap = 0;
      }
  }
  return 0;
}

}  // namespace hyhip

using namespace hyhip;

extern "C" {

/* Host-only (no device needed): what hyphy_hip_create decides about subtree repeats from the topology and the leaf table alone.
 * classes_out[I]: classes of every internal node over the S patterns as given (no padding, one shard); compressed_out[I]: 1 where
 * the node's subtree is evaluated per class (theta <= 0: the library's default).  Returns the edge products a full pass executes
 * with one table per compressed node (sum of their classes + S per trunk edge), < 0 on bad arguments. */
/* Host-only: the walk program of a trunk given as a tree over generalised leaves (node codes 0 .. L - 1: leaves, L + i: internal node
 * i, children before parents, the root last; parents[c] = node code of c's parent, the root's entry is ignored).
 * out: [nodes, inputs, stack depth, two chains?, first / end node of the one-chain form, of chain 0, of chain 1], then per node
 * (internal index or -1, leaf children, first input, flags), then the inputs (leaf codes).  Returns the words written, < 0: cap too small. */
int64_t hyphy_hip_plan_trunk_walk(int64_t L, int64_t I, const int64_t *parents, int64_t *out, int64_t cap) {
  if (L < 1 || I < 1 || !parents || !out) return -1;
  hyphy_hip_partition::View v;
  v.L = (int)L, v.I = (int)I;
  v.children.assign((size_t)I, std::vector<int>());
  for (int64_t c = 0; c < L + I - 1; c++) {
    if (parents[c] < L || parents[c] >= L + I) return -1;
    v.children[(size_t)(parents[c] - L)].push_back((int)c);
  }
  const hyhip::WalkPlan wp = hyhip::plan_trunk_walk(v);
  const int64_t need = 10 + 4 * (int64_t)wp.nodes.size() + (int64_t)wp.inputs.size();
  if (need > cap) return -need;
  int64_t *o = out;
  *o++ = (int64_t)wp.nodes.size(), *o++ = (int64_t)wp.inputs.size(), *o++ = wp.max_depth, *o++ = wp.two ? 1 : 0;
  *o++ = wp.one_range[0], *o++ = wp.one_range[1];
  for (int k = 0; k < 4; k++) *o++ = wp.two ? wp.two_range[(size_t)k] : 0;
  for (const auto &n : wp.nodes) *o++ = n.node, *o++ = n.n_in, *o++ = n.in0, *o++ = n.flags;
  for (int x : wp.inputs) *o++ = x;
  return need;
}

int64_t hyphy_hip_plan_repeats(int64_t L, int64_t I, const int64_t *flat_parents, int64_t S, const int64_t *leaf_codes, double theta,
                               int64_t *classes_out, int64_t *compressed_out) {
  if (L < 2 || I < 1 || S < 1 || !flat_parents || !leaf_codes) return -1;
  std::vector<std::vector<int>> children((size_t)I);
  for (int64_t n = 0; n < L + I - 1; n++) {
    const int64_t par = flat_parents[n];
    if (par < 0 || par >= I || (n >= L && par <= n - L)) return -1;
    children[(size_t)par].push_back((int)n);
  }
  std::vector<int16_t> codes((size_t)L * S);
  for (int64_t k = 0; k < L * S; k++) {
    if (leaf_codes[k] > 32767 || leaf_codes[k] < -32768) return -1;
    codes[(size_t)k] = (int16_t)leaf_codes[k];
  }
  std::vector<std::vector<int>> cls;
  std::vector<std::vector<int>> U(1);
  count_classes(children, (int)L, (int)I, codes, (size_t)S, cls, U[0]);
  const std::vector<char> comp = compressed_set(children, (int)L, (int)I, U, std::vector<int>(1, (int)S), theta > 0. ? theta : 0.35);
  int64_t work = 0;
  for (int64_t n = 0; n < I; n++) {
    if (classes_out) classes_out[n] = U[0][(size_t)n];
    if (compressed_out) compressed_out[n] = comp[(size_t)n];
    if (n < I - 1) work += comp[(size_t)n] ? U[0][(size_t)n] : S;
  }
  return work;
}

}  // extern "C"
