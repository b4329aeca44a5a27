// Felsenstein pruning on gfx950 — device counterpart of _TheTree::ComputeTreeBlockByBranch
// (src/core/tree_evaluator.cpp:3556-4171) and of the combine step of ComputeBlock
// (src/core/likefunc.cpp:11046-11123).
//
// Design (MI355X-first, not a translation of the per-site CPU loop):
//  * A workgroup owns T tiles of 16 site patterns and walks the WHOLE post-order schedule for
//    them, so conditionals flow child -> parent through registers/LDS; HBM sees each finished
//    node once (persist, for later partial updates) instead of a write + read per tree level.
//  * Per child edge the product  [DP x DP] x [DP x 16 sites]  runs on the FP64 matrix cores
//    (v_mfma_f64_16x16x4_f64).  Wave w of the workgroup owns parent-state rows 16w..16w+15; the
//    MFMA C/D register image equals the B-operand image (common.h), so a node's result feeds its
//    parent's product with no shuffle.  The four row blocks are exchanged through LDS once per
//    node, together with the per-site sums that drive the 2^64 underflow rescaling.
//  * Leaf edges are a column gather from a transposed image of P (K4 in SURVEY §2.1); leaves
//    with ambiguity codes take the MFMA path with their resolution vector as the B operand.
//  * Rescaling is stateless per evaluation: integer exponents per (node, pattern) are carried up
//    the tree; observable contract of SURVEY A.5 (l_s, c_s with L_s = l_s 2^(-64 c_s)).
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "devutil.h"
#include "expm4.h"
#include "combine.h"

#ifndef HYPHY_OCC
#define HYPHY_OCC 3  // waves per SIMD the T = 1 pruning kernel is compiled for
#endif

namespace hyhip {

namespace {

// A root-finalising wave's share of the log-likelihood sum (all 64 lanes active, values wave-uniform): entry `idx` of the
// per-tile partial sums.  PruneArgs::red_out == nullptr: plain stores, wg_reduce_kernel combines them behind the launch.
// Otherwise (r03, fused final combine) the partials go out with agent-scope stores, the wave arrives at red_done, and the
// LAST arriver of the launch reads all red_n partials back (sc1 loads, all in flight together; entries past red_n read as
// zero through the buffer bounds), sums them in a FIXED order — lane l takes entries 128 j + 2 l, 128 j + 2 l + 1 for
// j = 0, 1, .. with a Kahan sum, then a compensated shuffle tree over the lanes — and publishes the result record exactly
// like wg_reduce_kernel does (same flags, same record, same sequence word).  The order of arrival decides only WHO sums.
// FUSE is a template parameter of the kernels: the combine's registers must not exist in the builds that do not use it (inlined
// behind a run-time test it cost the production wave kernel 15 scratch instructions and 3 us of 119 at the headline size).
template <bool FUSE>
__device__ __forceinline__ void publish_partial(const PruneArgs &a, int idx, double wsum, long long wcnt, int wflag, int lane) {
  if (!FUSE || a.red_out == nullptr) {
    if (lane == 0) {
      a.wg_sum[idx] = wsum;
      a.wg_cnt[idx] = wcnt;
      a.wg_flag[idx] = wflag;
    }
    return;
  }
  if constexpr (FUSE) {
  if (lane == 0) {
    __hip_atomic_store(a.wg_sum + idx, wsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(a.wg_cnt + idx, wcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(a.wg_flag + idx, wflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (payload written through before the arrival: the protocol of the chain joins)
  int old = 0;
  if (lane == 0) old = __hip_atomic_fetch_add(a.red_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  old = __builtin_amdgcn_readfirstlane(old);
  asm volatile("" ::: "memory");
  if (old + 1 < a.red_n) return;
  if (lane == 0) __hip_atomic_store(a.red_done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next launch
  combine_partials(a.wg_sum, a.wg_cnt, a.wg_flag, a.red_n, a.red_out, a.red_rec, a.red_status, a.red_seq, lane);
  }  // FUSE
}

// Operand bundle fetched one schedule entry ahead: 16 doubles per lane, either the A-operand image
// of the next internal edge's transition matrix or the gathered columns of the next leaf group.
struct Payload {
  f64x2 v[8];
  int c0 = 0, c1 = 0;  // (REP builds) 2^64 exponents of the class-table rows gathered for the entry's leaves
};

// Root epilogue shared by the pruning kernels: L_s = sum_k root[s][k] pi[k]; this workgroup's share of
// sum_s f_s log L_s (tree_evaluator.cpp:4046-4128) and of the integer scaler sum (likefunc.cpp:11123).
// `rootv` = the root's unscaled conditional tiles in LDS (fragment layout), `rscale`/`rcnt` = this wave's
// copy of the root's per-site scale and 2^64-exponent ([T][16]).
template <int NW, int T, bool FUSE = false>
__device__ __forceinline__ void root_epilogue(const PruneArgs &a, const double *rootv0, const double *rscale,
                                              const int *rcntv, int tile0, int w, int lane) {
  constexpr int NKK = 4 * NW, TILE = NKK * 64;
  const int g = lane >> 4, sl = lane & 15;
  double pk[NKK];
#pragma unroll
  for (int kk = 0; kk < NKK; kk++) pk[kk] = a.pi[4 * kk + g];
  double wsum = 0.;
  long long wcnt = 0;
  int wflag = 0;
#pragma unroll
  for (int t = 0; t < T; t++) {
    double s = 0.;
    const double *rootv = rootv0 + t * TILE;
#pragma unroll
    for (int kk = 0; kk < NKK; kk++) s = fma(rootv[frag_index(kk, lane)], pk[kk], s);
    s *= rscale[t * 16 + sl];
    const int rcnt = rcntv[t * 16 + sl];
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (w == 0 && g == 0) {
      const int site = (tile0 + t) * 16 + sl;
      a.site_lik[site] = s;
      a.site_cnt[site] = rcnt;
      const double f = a.freq[site];
      if (f != 0.) {
        if (s != s || isinf(s)) wflag |= 2;
        else if (s <= 0.) wflag |= 1;
        else {
          wsum += log(s) * f;
          wcnt += (long long)rcnt * (long long)f;
        }
      }
    }
  }
  if (w == 0) {  // fixed-order butterfly over the 16 site lanes (lanes >= 16 hold zeros)
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
      wsum += __shfl_xor(wsum, off);
      wcnt += __shfl_xor(wcnt, off);
      wflag |= __shfl_xor(wflag, off);
    }
    publish_partial<FUSE>(a, (int)blockIdx.x, wsum, wcnt, wflag, lane);
  }
}

// The pruning kernel interprets a host-compiled schedule (api.hip: build_schedule).  Everything that
// can be decided on the host is: entry kind, which LDS slot a finished node goes to (including the
// ping-pong parity of the two exchange slots), where a child vector is read from.  The device loop is
// kept as branch-poor as possible — with only ~2.4 waves per SIMD at the benchmark size, every
// dependent scalar/LDS round trip in the interpreter is exposed latency (measured: an interpreter with
// ~160 branches cost more than the MFMAs themselves).
//
// CLDS: leaf codes of the workgroup's tiles are staged in LDS (the common case); the !CLDS variant
// (thousands of taxa) reads them from global memory.
// CHAIN (r03, T = 1): chain schedules for the row-split workgroup (api: "team" kernel, PruneArgs::variant 2).  grid.z indexes
// the SOURCE programs of a chain schedule (schedule.hip); behind its source the workgroup walks the trunk exactly like the
// wave-per-tile kernel does — edge product towards the parent, arrival at the parent's counter, the last arriver multiplies
// the deposited products of its siblings in and goes on — but every product is split over the NW waves (16 MFMAs each
// instead of 64 from one wave), the waves agree on every arrival through one LDS word, and each wave deposits / fetches only
// its own 16 rows.  A tile's critical path is then (height of the tree) x (a quarter of the wave kernel's edge latency):
// what small shards — a rank's share of an alignment at 4 or 8 GPUs — are bound by.
// REP (r06): the tree is the TRUNK of a class-compressed partition (repeats.hip) — a leaf of the schedule may be a generalised leaf: its
// columns are rows of a class table, gathered by class id, and carry a 2^64 exponent (what prune_wave_kernel<.., REP> does with one wave
// per tile).  The trunk is a handful of nodes: a tile's critical path, not the chip's arithmetic, bounds the launch, and a team's edge
// product is a quarter of a wave's.  T = 1, leaf table of the view staged in LDS.
template <int NW, int T, bool CLDS, bool TRACE, int ABL = 0, bool CHAIN = false, bool FUSE = false, bool REP = false>
__global__ __launch_bounds__(64 * NW, (T == 1 ? HYPHY_OCC : 1)) void prune_mfma_kernel(const int4 *__restrict__ ops,
                                                                               PruneArgs a) {
  static_assert(!CHAIN || T == 1, "chain schedules: one tile per workgroup");
  static_assert(!REP || (T == 1 && CLDS), "class-compressed trunk: one tile per workgroup, leaf table in LDS");
  // forest scheduling: grid.z = subtree fragment of this level, each with its own program
  const int4 prg = a.prog[blockIdx.z];
  const int4 *const ops0 = ops;
  ops += prg.x;
  {  // rate-class batching: one grid row per class, same schedule, class-strided buffers
    const size_t cat = blockIdx.y;
    if (CHAIN) {
      a.frag_ctr += cat * (size_t)a.n_prog_total * a.ntiles;
      a.hand_cnt += cat * (size_t)(a.root_inode + 1) * a.ntiles * 32;
      a.deposits += cat * a.cs_deposits;
    }
    a.Pfrag += cat * a.cs_P;
    a.PTg += cat * a.cs_P;
    a.partials += cat * a.cs_partials;
    a.counts += cat * a.cs_counts;
    a.site_lik += cat * a.cs_site;
    a.site_cnt += cat * a.cs_site;
    a.wg_sum += cat * a.cs_wg;
    a.wg_cnt += cat * a.cs_wg;
    a.wg_flag += cat * a.cs_wg;
    if constexpr (REP) {
      a.gtab += cat * a.cs_gtab;
      a.gcnt += cat * a.cs_gcnt;
    }
  }
  constexpr int NKK = 4 * NW, DP = 16 * NW, TILE = NKK * 64;
  constexpr int G = (T <= 2) ? 2 : 1;  // leaves per leaf-group entry (T*G*4 doubles <= 16)
  static_assert(NW == 4 || NKK <= 16, "payload sized for DP <= 64");
  // LDS: NS slots of T tiles.  Slots 0/1 are the ping-pong exchange buffers of successive node
  // finalisations (one barrier per node); slots >= 2 park finished nodes whose parent is not the next
  // schedule entry.  Slot data is unscaled; each wave keeps its own copy of the per-site scale and
  // exponent of every slot (written and read by the same wave: no cross-wave ordering needed).
  constexpr int NS = lds_slots(T);
  __shared__ __align__(16) double xbuf[NS * T * TILE];
  __shared__ double psum[2][T][NW * 64];            // per-lane partial site sums of a finalisation
  __shared__ double slot_scale[NS][NW][T][16];
  __shared__ int slot_cnt[NS][NW][T][16];
  extern __shared__ __align__(16) int16_t codes_lds[];  // CLDS: [L][T*16] leaf codes

  if constexpr (CHAIN) {
    if ((int)blockIdx.x >= a.ntiles) return;  // (tile dimension padded to a multiple of 8: launch_prune_T)
  }
  const int lane = threadIdx.x & 63, g = lane >> 4, sl = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave index in an SGPR: scalar bases
  const int tile0 = blockIdx.x * T;
  const int S_pad = a.S_pad;
  constexpr int ablate = ABL;  // DIAGNOSTIC builds only (HYPHY_HIP_ABLATE); 0 in production

  if (CLDS) {
    const int n = a.L * T * 16;
    for (int i = threadIdx.x; i < n; i += 64 * NW) {
      const int leaf = i / (T * 16), off = i - leaf * (T * 16);
      if constexpr (REP) {  // (the view's leaf table is tile-major [tile][view leaves][16]: state codes / class ids)
        codes_lds[i] = a.codes_tile[((size_t)tile0 * a.L + leaf) * 16 + off];
      } else {
        codes_lds[i] = (leaf == a.pin_leaf) ? a.pin[tile0 * 16 + off]  // pinned leaf: its states replace the data
                                            : a.codes[(size_t)leaf * S_pad + tile0 * 16 + off];
      }
    }
    __syncthreads();
  }
  auto leaf_code = [&](int leaf, int t) -> int {
    if (CLDS) return (int)codes_lds[leaf * (T * 16) + t * 16 + sl];
    if (leaf == a.pin_leaf) return (int)a.pin[(tile0 + t) * 16 + sl];
    return (int)a.codes[(size_t)leaf * S_pad + (tile0 + t) * 16 + sl];
  };
  // Issue the global loads for a schedule entry (they complete while the previous entry computes).
  // STRAIGHT-LINE on purpose: exactly 8 x 16-byte loads per lane for every entry kind, addresses
  // chosen with wave-uniform selects.  With an if/else per kind the two paths load into different
  // registers and the compiler joins them with "s_waitcnt vmcnt(0) + 16 x v_mov_b64" — i.e. it waits
  // for the prefetch right after issuing it (seen in the ISA; it made every earlier attempt at
  // hiding the operand latency a no-op).
  auto prefetch = [&](const int4 &op, Payload &pay) {
    if (ablate & 16) return;
    const bool is_leaf = (op.x & 3) == OPK_LEAF;
    const int leaf0 = is_leaf ? (op.z & 0xffff) : 0, leaf1 = is_leaf ? ((op.z >> 16) & 0xffff) : 0;
    int code[2][T];
#pragma unroll
    for (int t = 0; t < T; t++) {
      const int c0 = leaf_code(leaf0, t), c1 = (G > 1) ? leaf_code(leaf1, t) : 0;
      code[0][t] = c0 < 0 ? 0 : c0;  // ambiguous sites: gathered value unused (slow path)
      code[1][t] = c1 < 0 ? 0 : c1;
    }
    const double *bl0 = a.PTg + ((size_t)leaf0 * DP * NW + w) * 16;     // uniform
    const double *bl1 = a.PTg + ((size_t)leaf1 * DP * NW + w) * 16;     // uniform
    if constexpr (REP) {
      // (first row of the leaf's class table or -1, first exponent row / matrix slot; the table is a member of the by-value argument
      //  block — vector loads: back into scalar registers, or every address derived from them counts as divergent)
      int2 lt0 = a.leaf_tab[leaf0], lt1 = a.leaf_tab[leaf1];
      lt0.x = __builtin_amdgcn_readfirstlane(lt0.x), lt0.y = __builtin_amdgcn_readfirstlane(lt0.y);
      lt1.x = __builtin_amdgcn_readfirstlane(lt1.x), lt1.y = __builtin_amdgcn_readfirstlane(lt1.y);
      bl0 = (lt0.x >= 0 ? a.gtab + (size_t)lt0.x * DP : a.PTg + (size_t)lt0.y * DP * DP) + (size_t)w * 16;
      bl1 = (lt1.x >= 0 ? a.gtab + (size_t)lt1.x * DP : a.PTg + (size_t)lt1.y * DP * DP) + (size_t)w * 16;
      pay.c0 = (is_leaf && lt0.x >= 0) ? a.gcnt[lt0.y + code[0][0]] : 0;
      pay.c1 = (is_leaf && lt1.x >= 0) ? a.gcnt[lt1.y + code[1][0]] : 0;
    }
    const double *bfr = a.Pfrag + ((size_t)op.z * NW + w) * TILE;       // uniform (internal entries)
    constexpr int NLOADS = G * T * 2;  // gather loads of a leaf group (<= 8)
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int kl = k % NLOADS, i = kl / (2 * T), t = (kl >> 1) % T, h = kl & 1;
      const double *base = is_leaf ? (i ? bl1 : bl0) : bfr;
      const unsigned off_leaf = (unsigned)(code[i][t] * NW * 16 + g * 4) * 8u + (unsigned)h * 16u;
      const unsigned off_frag = (unsigned)((k % (NKK / 2)) * 64 + lane) * 16u;
      if (k < NLOADS) {
        pay.v[k] = ld16(base, is_leaf ? off_leaf : off_frag);  // both kinds: same registers, one path
      } else if (!is_leaf) {
        pay.v[k] = ld16(bfr, off_frag);  // only internal entries need the rest of the A image; a leaf
      }                                  // entry leaves these registers untouched (no join copies)
    }
  };
  // ARRIVE: a dummy use of a payload.  It makes the compiler put its vector-memory wait HERE instead
  // of (as vmcnt(0), because the number of loads in flight differs per path) before the first MFMA of
  // the next entry, where it would also drain the loads just issued for the entry after that.
  auto arrive = [](const Payload &q) {
    asm volatile("" ::"v"(q.v[0]), "v"(q.v[1]), "v"(q.v[2]), "v"(q.v[3]), "v"(q.v[4]), "v"(q.v[5]), "v"(q.v[6]),
                 "v"(q.v[7]));
    if constexpr (REP) asm volatile("" ::"v"(q.c0), "v"(q.c1));
  };

  f64x4 acc[T];  // this wave's 16 parent states x 16 sites running product
  int cnt[T];    // 2^64-exponent of the running product
#pragma unroll
  for (int t = 0; t < T; t++) {
    acc[t] = (f64x4){1., 1., 1., 1.};
    cnt[t] = 0;
  }

  // Finalise the parent: publish this wave's 16 rows and its partial site sums, ONE barrier, decide the rescale, persist.
  // `opx` = flag word of the parent's last entry, `mid` = what to do between the LDS part and the persist stores.
  auto finalise = [&](int opx, int parent, long long *tl, bool trace, auto mid) {
      // Finalise the parent: publish this wave's 16 rows and its partial site sums, ONE barrier,
      // decide the rescale, persist.  The exchange slot and psum buffer alternate between successive
      // finalisations (parity chosen by the host), so a wave that runs ahead writes into the other
      // buffer and cannot get two nodes ahead (it has to pass the next node's barrier first).
      const int par2 = (opx >> 4) & 1;
      const int slot = (opx >> 16) & 0xff;
      
      if (parent == a.pin_inode) {  // pinned internal node: only the pinned state survives (tree_evaluator.cpp:589-594)
#pragma unroll
        for (int t = 0; t < T; t++) {
          const int ps = (int)a.pin[(tile0 + t) * 16 + sl];
#pragma unroll
          for (int r = 0; r < 4; r++) acc[t][r] = (16 * w + 4 * r + g == ps) ? acc[t][r] : 0.;
        }
      }
#pragma unroll
      for (int t = 0; t < T; t++) {
        psum[par2][t][w * 64 + lane] = (acc[t][0] + acc[t][1]) + (acc[t][2] + acc[t][3]);
        double *dst = xbuf + (slot * T + t) * TILE;
        if (!(ablate & 32)) {  // kk = 4w + r  ->  frag_index(kk, lane): two 16-byte stores
          *reinterpret_cast<f64x2 *>(dst + ((2 * w) * 64 + lane) * 2) = (f64x2){acc[t][0], acc[t][1]};
          *reinterpret_cast<f64x2 *>(dst + ((2 * w + 1) * 64 + lane) * 2) = (f64x2){acc[t][2], acc[t][3]};
        }
      }
      if (!(ablate & 2)) lds_barrier();
      if (trace && lane == 0) tl[2] = clock64();
      double sc[T];
#pragma unroll
      for (int t = 0; t < T; t++) {
        // site total over all NW*4 row groups, fixed order (identical in every wave)
        double tot = 0.;
#pragma unroll
        for (int q = 0; q < NW * 4; q++) tot += psum[par2][t][q * 16 + sl];
        sc[t] = 1.0;
        int m = 0;
        if (__any(!(tot >= kScalerThreshold && tot <= kScalerUp))) m = rescale_decision(tot, sc[t]);  // rare
        cnt[t] += m;
        slot_scale[slot][w][t][sl] = sc[t];  // (the 4 row-group lanes of a site store the same value)
        slot_cnt[slot][w][t][sl] = cnt[t];
      }
      // ARRIVE before the persist stores are issued: the vector-memory counter is in-order and counts
      // stores; this way the stores have the whole following entry to complete.
      mid();
#pragma unroll
      for (int t = 0; t < T; t++) {
        // persist this wave's rows (== its own accumulator, times the exact power-of-two scale) and
        // the exponent: fire and forget
        double *out = a.partials + ((size_t)parent * a.ntiles + tile0 + t) * TILE + (size_t)w * 256;  // uniform
        const f64x4 q = acc[t] * sc[t];
        if (!(ablate & 4) && !(opx & OPF_NOPERSIST)) {  // (lazy persistence: the host knows nobody re-reads this node)
          st16(out, (unsigned)lane * 16u, (f64x2){q[0], q[1]});
          st16(out, (unsigned)(64 + lane) * 16u, (f64x2){q[2], q[3]});
          if (w == 0 && g == 0) a.counts[(size_t)parent * S_pad + (tile0 + t) * 16 + sl] = cnt[t];
        }
        acc[t] = (f64x4){1., 1., 1., 1.};  // the next entry starts a new parent
        cnt[t] = 0;
      }
      if (trace && lane == 0) tl[3] = clock64();
  };

  // one schedule entry: multiply one child edge (or leaf group) into the parent's running product and,
  // after the parent's last child, finalise it.  `nxt` = payload of the following entry (in flight).
  auto body = [&](const int4 &op, const Payload &pay, const Payload &nxt, int oi) {
    const int kind = op.x & 3;
    const bool trace = TRACE && a.timeline != nullptr && blockIdx.x < kTraceWG;
    long long *tl = trace ? a.timeline + (((size_t)blockIdx.x * NW + w) * a.n_ops + oi) * 4 : nullptr;
    if (trace && lane == 0) tl[0] = clock64();

    if (kind == OPK_LEAF) {
      // K4: parent[k] *= P[k][state] — the columns were gathered one entry ahead
      const int nl = (op.x >> 8) & 0x7f;
      if (!(op.x & OPF_AMBIG)) {
        {
          const bool one = nl > 0;  // (nl == 0: padding entry)
#pragma unroll
          for (int t = 0; t < T; t++) {
            const f64x4 m = (f64x4){pay.v[t * 2][0], pay.v[t * 2][1], pay.v[t * 2 + 1][0], pay.v[t * 2 + 1][1]};
            acc[t] *= one ? m : (f64x4){1., 1., 1., 1.};
          }
        }
        if (G > 1) {  // second leaf of the group: uniform select instead of a branch
          const bool two = nl > 1;
#pragma unroll
          for (int t = 0; t < T; t++) {
            const f64x4 m = (f64x4){pay.v[(T + t) * 2][0], pay.v[(T + t) * 2][1], pay.v[(T + t) * 2 + 1][0],
                                    pay.v[(T + t) * 2 + 1][1]};
            acc[t] *= two ? m : (f64x4){1., 1., 1., 1.};
          }
        }
        if constexpr (REP) cnt[0] += (nl > 0 ? pay.c0 : 0) + (nl > 1 ? pay.c1 : 0);
      } else {
        // slow path: some leaf of the group carries ambiguity codes.  Tiles containing one take the
        // full product with the resolution vector as B operand (operands streamed, not staged).
#pragma unroll
        for (int i = 0; i < G; i++) {
          if (i < nl) {
            const int leaf = (op.z >> (16 * i)) & 0xffff;
            const double *Af = a.Pfrag + ((size_t)leaf * NW + w) * TILE;
#pragma unroll
            for (int t = 0; t < T; t++) {
              const int c = leaf_code(leaf, t);
              if (!__any(c < 0)) {
                acc[t] *= (f64x4){pay.v[(i * T + t) * 2][0], pay.v[(i * T + t) * 2][1], pay.v[(i * T + t) * 2 + 1][0],
                                  pay.v[(i * T + t) * 2 + 1][1]};
              } else {
                f64x4 d = (f64x4){0., 0., 0., 0.};
                const double *av = a.ambig + (size_t)(c < 0 ? -c - 1 : 0) * DP;
#pragma unroll 2
                for (int kk = 0; kk < NKK; kk++) {
                  const double bv = (c >= 0) ? ((4 * kk + g == c) ? 1.0 : 0.0) : av[4 * kk + g];
                  d = mfma(Af[frag_index(kk, lane)], bv, d);
                }
                acc[t] *= d;
              }
            }
          }
        }
      }
    } else {
      // internal child: its conditional vector is the B operand, streamed from its LDS slot (exchange
      // slot of the previous finalisation, or a parking slot) while the MFMAs run; the per-site
      // power-of-two scale is applied to the product (columns are sites).  Two accumulator chains per
      // tile: one wave cannot issue f64 MFMAs back to back on one accumulator.
      f64x4 d0[T], d1[T];
      double csc[T];
      int ccnt[T];
#pragma unroll
      for (int t = 0; t < T; t++) d0[t] = d1[t] = (f64x4){0., 0., 0., 0.};
      if (kind == OPK_INTERNAL) {
        const int slot = (op.x >> 24) & 0xff;
#pragma unroll
        for (int t = 0; t < T; t++) {
          csc[t] = slot_scale[slot][w][t][sl];
          ccnt[t] = slot_cnt[slot][w][t][sl];
        }
        if (!(ablate & 1)) {
#pragma unroll
          for (int k2 = 0; k2 < NKK / 2; k2++)
#pragma unroll
            for (int t = 0; t < T; t++) {
              const f64x2 bv =
                  *reinterpret_cast<const f64x2 *>(xbuf + (slot * T + t) * TILE + (k2 * 64 + lane) * 2);
              d0[t] = mfma(pay.v[k2 % 8][0], bv[0], d0[t]);
              d1[t] = mfma(pay.v[k2 % 8][1], bv[1], d1[t]);
            }
        }
      } else {
        // child not recomputed in this call (or the LDS slots ran out): persisted copy in HBM
        const int cinode = op.w;
        if (op.x & OPF_GSYNC) __syncthreads();
#pragma unroll
        for (int t = 0; t < T; t++) {
          const double *src = a.partials + ((size_t)cinode * a.ntiles + tile0 + t) * TILE;  // uniform
          csc[t] = 1.;
          ccnt[t] = a.counts[(size_t)cinode * S_pad + (tile0 + t) * 16 + sl];
#pragma unroll
          for (int k2 = 0; k2 < NKK / 2; k2++) {
            const f64x2 bv = ld16(src, (unsigned)(k2 * 64 + lane) * 16u);
            d0[t] = mfma(pay.v[k2 % 8][0], bv[0], d0[t]);
            d1[t] = mfma(pay.v[k2 % 8][1], bv[1], d1[t]);
          }
          // consume every load of this (rarer) path INSIDE the branch: a load still in flight at the
          // join would make the compiler guard the common path with vmcnt(0) as well
          asm volatile("" ::"v"(ccnt[t]));
        }
      }
#pragma unroll
      for (int t = 0; t < T; t++) {
        acc[t] *= (d0[t] + d1[t]) * csc[t];
        cnt[t] += ccnt[t];
      }
    }

    if (trace) {
      asm volatile("" ::"v"(acc[0][0]));
      if (lane == 0) tl[1] = clock64();
    }
    if (op.x & OPF_LAST) {
      finalise(op.x, op.y, tl, trace, [&]() { arrive(nxt); });
    } else {
      arrive(nxt);
    }
  };

  // Software pipeline over the schedule, unrolled by two with ping-pong operand registers.
  // Per entry i:   issue loads(i+1)  ->  compute(i)  ->  ARRIVE(i+1).
  // Schedule entries come through scalar loads (SGPRs, uniform control flow), fetched two ahead.
  // The host pads the schedule to an even number of entries and appends two more no-op entries
  // (empty leaf groups), so the loop needs no bounds tests besides its own.
  const int n_ops = prg.y;  // even
  int4 opA = ops[0];
  int4 opB = ops[1];
  Payload pA, pB;
#pragma unroll
  for (int k = 0; k < 8; k++) pA.v[k] = pB.v[k] = (f64x2){0., 0.};
  prefetch(opA, pA);
  arrive(pA);
  for (int oi = 0; oi < ((ablate & 64) ? 0 : n_ops); oi += 2) {
    prefetch(opB, pB);
    const int4 opC = ops[oi + 2];
    body(opA, pA, pB, oi);
    prefetch(opC, pA);
    const int4 opD = ops[oi + 3];
    body(opB, pB, pA, oi + 1);
    opA = opC;
    opB = opD;
  }

  if constexpr (CHAIN) {
    // ---- the trunk (see the wave-per-tile kernel for the protocol and its memory-ordering contract) ----
    __shared__ int s_word[2];  // arrival counter values, read / updated by thread 0 and agreed on through LDS
    int cur = prg.z & 0xff;    // exchange slot the source's root was finalised into
    int c = prg.w;             // internal index of the node whose conditionals are in slot `cur`
    const int tile = tile0;
    const int4 *__restrict__ jn = a.jn;
    // this wave's rows of the A-operand image of one branch: 8 x 16 bytes per lane, requested a whole level ahead
    auto load_image = [&](int branch, Payload &pa) {
      const double *bfr = a.Pfrag + ((size_t)branch * NW + w) * TILE;  // uniform
#pragma unroll
      for (int k = 0; k < NKK / 2 && k < 8; k++) pa.v[k] = ld16(bfr, (unsigned)(k * 64 + lane) * 16u);
    };
    // One trunk entry's operands, requested one entry ahead of their use.  Straight-line on purpose (always four 16-byte
    // loads and one 4-byte load, addresses chosen with uniform selects): a leaf group's two column gathers, or this
    // wave's rows of a deposited product (+ its exponents).  L1-bypassing loads throughout — deposits come from other CUs.
    struct Item {
      f64x2 v[4];
      int cnt;
    };
    auto issue = [&](const int4 &op, Item &it) {
      const bool is_leaf = (op.x & 3) == OPK_LEAF;
      const int nl = (op.x >> 8) & 0x7f;
      const int leaf0 = is_leaf ? (op.z & 0xffff) : 0, leaf1 = (is_leaf && nl > 1) ? ((op.z >> 16) & 0xffff) : leaf0;
      const int c0 = leaf_code(leaf0, 0), c1 = leaf_code(leaf1, 0);
      const int child = is_leaf ? 0 : op.w;
      const double *dep = a.deposits + ((size_t)child * a.ntiles + tile) * TILE + (size_t)w * 256;             // uniform
      const double *b0 = is_leaf ? a.PTg + ((size_t)leaf0 * DP * NW + w) * 16 : dep;                         // uniform
      const double *b1 = is_leaf ? a.PTg + ((size_t)leaf1 * DP * NW + w) * 16 : dep;                         // uniform
      [[maybe_unused]] int gc = 0;  // (REP) exponents of the class-table rows of the entry's generalised leaves
      if constexpr (REP) {
        if (is_leaf) {
          int2 lt0 = a.leaf_tab[leaf0], lt1 = a.leaf_tab[leaf1];
          lt0.x = __builtin_amdgcn_readfirstlane(lt0.x), lt0.y = __builtin_amdgcn_readfirstlane(lt0.y);
          lt1.x = __builtin_amdgcn_readfirstlane(lt1.x), lt1.y = __builtin_amdgcn_readfirstlane(lt1.y);
          b0 = (lt0.x >= 0 ? a.gtab + (size_t)lt0.x * DP : a.PTg + (size_t)lt0.y * DP * DP) + (size_t)w * 16;
          b1 = (lt1.x >= 0 ? a.gtab + (size_t)lt1.x * DP : a.PTg + (size_t)lt1.y * DP * DP) + (size_t)w * 16;
          if (lt0.x >= 0 && nl > 0) gc += a.gcnt[lt0.y + (c0 < 0 ? 0 : c0)];
          if (lt1.x >= 0 && nl > 1) gc += a.gcnt[lt1.y + (c1 < 0 ? 0 : c1)];
        }
      }
      const unsigned g0 = (unsigned)((c0 < 0 ? 0 : c0) * NW * 16 + g * 4) * 8u, g1 = (unsigned)((c1 < 0 ? 0 : c1) * NW * 16 + g * 4) * 8u;
      const unsigned o0 = is_leaf ? g0 : (unsigned)lane * 16u, o0b = is_leaf ? g0 + 16u : (unsigned)(64 + lane) * 16u;
      const unsigned o1 = is_leaf ? g1 : (unsigned)lane * 16u, o1b = is_leaf ? g1 + 16u : (unsigned)(64 + lane) * 16u;
      it.v[0] = ld16_agent(b0, o0);
      it.v[1] = ld16_agent(b0, o0b);
      it.v[2] = ld16_agent(b1, o1);
      it.v[3] = ld16_agent(b1, o1b);
      it.cnt = __hip_atomic_load(a.hand_cnt + ((size_t)child * a.ntiles + tile) * 32 + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if constexpr (REP) it.cnt = is_leaf ? gc : it.cnt;
    };
    Payload pimg;
#ifdef HYPHY_TRUNK_PRIO
    __builtin_amdgcn_s_setprio(HYPHY_TRUNK_PRIO);  // (experiment: see the wave-per-tile kernel)
#endif
    // `s_early[k & 1]`: arrival counter of level k's parent, sampled by thread 0 a whole level ahead (while the node below is
    // finalised: the round trip is covered) and agreed on through LDS.  If every sibling had arrived by then, this workgroup
    // WILL be the last arriver: it skips its own deposit and has the siblings' deposits in flight under its own product.
    __shared__ int s_early[2];
    int lvl = 0;
    {
      const int4 j0 = jn[c];
      if (j0.x >= 0) {
        load_image(j0.x >> 16, pimg);
        const int par0 = j0.x & 0xffff;
        if ((jn[par0].y & 0xff) > 1 && threadIdx.x == 0)
          s_early[0] = __hip_atomic_load(a.frag_ctr + (size_t)par0 * a.ntiles + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      lds_barrier();
    }
    for (;; lvl++) {
      const int4 jc = jn[c];  // x = parent | matrix-image slot of the edge c -> parent << 16; -1: c is the root
      if (jc.x < 0) break;
      const int par = jc.x & 0xffff;
      const int4 jp = jn[par];
      const int need = jp.y & 0xff;
      int *ctr = a.frag_ctr + (size_t)par * a.ntiles + tile;
      const bool pred = need <= 1 || s_early[lvl & 1] == need - 1;  // (uniform over the workgroup)
      asm volatile("" ::: "memory");
      const int4 *tops = ops0 + jp.z;  // (followed by a no-op entry: one entry ahead is always readable)
      int4 op = tops[0];
      Item itA, itB;
      if (pred && jp.w > 0) issue(op, itA);
      // this wave's 16 rows of the edge product P_branch x (conditionals of c)
      f64x4 prod;
      int pcnt;
      {
        const double csc = slot_scale[cur][w][0][sl];
        pcnt = slot_cnt[cur][w][0][sl];
        f64x4 d0 = (f64x4){0., 0., 0., 0.}, d1 = d0;
#pragma unroll
        for (int k2 = 0; k2 < NKK / 2; k2++) {
          const f64x2 bv = *reinterpret_cast<const f64x2 *>(xbuf + cur * TILE + (k2 * 64 + lane) * 2);
          d0 = mfma(pimg.v[k2 % 8][0], bv[0], d0);
          d1 = mfma(pimg.v[k2 % 8][1], bv[1], d1);
        }
        prod = (d0 + d1) * csc;
      }
      if (jp.x >= 0) load_image(jp.x >> 16, pimg);  // the edge above `par`: in flight while this level joins and finalises
      if (!pred) {
        // deposit this workgroup's product (each wave its rows), drain, count the arrival
        double *out = a.deposits + ((size_t)c * a.ntiles + tile) * TILE + (size_t)w * 256;  // uniform
        st16_agent(out, (unsigned)lane * 16u, (f64x2){prod[0], prod[1]});
        st16_agent(out, (unsigned)(64 + lane) * 16u, (f64x2){prod[2], prod[3]});
        if (w == 0 && g == 0)
          __hip_atomic_store(a.hand_cnt + ((size_t)c * a.ntiles + tile) * 32 + sl, pcnt, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();  // every wave's payload has left for L2
        if (threadIdx.x == 0) s_word[1] = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lds_barrier();
        if (s_word[1] + 1 < need) return;  // somebody else will finish `par`
        asm volatile("" ::: "memory");
        if (jp.w > 0) issue(op, itA);
      }
      if (threadIdx.x == 0) {
        if (need > 1) __hip_atomic_store(ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
        if (jp.x >= 0 && (jn[jp.x & 0xffff].y & 0xff) > 1)  // the NEXT level's arrivals, sampled now
          s_early[(lvl + 1) & 1] = __hip_atomic_load(a.frag_ctr + (size_t)(jp.x & 0xffff) * a.ntiles + tile, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT);
      }
      // `par`: own product, leaf columns, the deposited products of the other children
      acc[0] = prod;
      cnt[0] = pcnt;
      int lastx = 0;
      auto consume = [&](const int4 &o, const Item &it) {
        lastx = o.x;
        if ((o.x & 3) == OPK_LEAF) {
          const int nl = (o.x >> 8) & 0x7f;
          bool slow = false;
          if (o.x & OPF_AMBIG) {
            const int leaf0 = o.z & 0xffff;
            slow = __any(leaf_code(leaf0, 0) < 0) != 0;  // (a leaf with ambiguity codes forms a group of its own)
          }
          if (!slow) {
            if (nl > 0) acc[0] *= (f64x4){it.v[0][0], it.v[0][1], it.v[1][0], it.v[1][1]};
            if (nl > 1) acc[0] *= (f64x4){it.v[2][0], it.v[2][1], it.v[3][0], it.v[3][1]};
            if constexpr (REP) cnt[0] += it.cnt;
          } else {  // ambiguity codes in this tile: product with the resolution vectors
            const int leaf = o.z & 0xffff;
            const int cd = leaf_code(leaf, 0);
            const double *Af = a.Pfrag + ((size_t)leaf * NW + w) * TILE;
            const double *av = a.ambig + (size_t)(cd < 0 ? -cd - 1 : 0) * DP;
            f64x4 d = (f64x4){0., 0., 0., 0.};
#pragma unroll 2
            for (int kk = 0; kk < NKK; kk++) {
              const double bv = (cd >= 0) ? ((4 * kk + g == cd) ? 1.0 : 0.0) : av[4 * kk + g];
              d = mfma(Af[frag_index(kk, lane)], bv, d);
            }
            acc[0] *= d;
          }
        } else if (o.w != c) {  // OPK_DEP: deposited by the workgroup that computed it (its arrival was counted)
          acc[0] *= (f64x4){it.v[0][0], it.v[0][1], it.v[1][0], it.v[1][1]};
          cnt[0] += it.cnt;
        }
      };
      for (int oi = 0; oi < jp.w; oi += 2) {  // unrolled by two: ping-pong operand registers, no copies
        const int4 opn = tops[oi + 1];
        if (oi + 1 < jp.w) issue(opn, itB);
        consume(op, itA);
        if (oi + 1 < jp.w) {
          op = tops[oi + 2];
          if (oi + 2 < jp.w) issue(op, itA);
          consume(opn, itB);
        }
      }
      cur ^= 1;
      finalise((lastx & (OPF_NOPERSIST | OPF_LAST)) | (cur ? OPF_PARITY : 0) | (cur << 16), par, nullptr, false, []() {});
      c = par;
    }
    root_epilogue<NW, T, FUSE>(a, xbuf + (size_t)cur * T * TILE, &slot_scale[cur][w][0][0], &slot_cnt[cur][w][0][0], tile0, w, lane);
    return;
  }
  if (a.do_root)
    root_epilogue<NW, T, FUSE>(a, xbuf + (size_t)a.root_slot * T * TILE, &slot_scale[a.root_slot][w][0][0],
                         &slot_cnt[a.root_slot][w][0][0], tile0, w, lane);
}

// ---------------------------------------------------------------------------------------------
// Wave-per-tile variant (T = 1): ONE wave owns a 16-pattern tile and all DP parent rows, so there is
// no cross-wave exchange at all — no barrier, no LDS round trip for the common child -> parent hand
// over.  The MFMA C/D image equals the B-operand image (kk = 4w + r), so the node finalised last is
// the next product's B operand straight from registers; per child edge the wave issues NW*NKK MFMAs
// on NW independent accumulator chains while streaming P's A-operand image from L2.  Nodes whose
// parent is not the next schedule entry are parked in a wave-private LDS slot (NP of them) or re-read
// from their persisted copy.  Workgroup = one wave; the grid is (tiles, classes, fragments).
// ---------------------------------------------------------------------------------------------
#ifndef HYPHY_OCC3
#define HYPHY_OCC3 2
#endif

// TRACE (HYPHY_HIP_TIMELINE, a separate diagnostic instantiation): per wave [start, prologue done, source program done,
// end] in 100 MHz ticks, trunk levels walked (+100 per prefetched join, +1000 per join won after depositing), how it
// ended (0 deposited and retired, 1 reached the root, 2 retired at a fragment hand-off), HW_ID and XCC_ID; it also honours
// HYPHY_HIP_ABLATE bits 256 / 512 (no deposit stores / no deposit reads: results invalid).
// HYPHY_ABL (compile-time bitmask, diagnostic builds only, results invalid): 1 = with HYPHY_HIP_ABLATE=4096 every edge reads
// branch 0's matrix image (A operand always cache-hot), 2 = one A chunk per edge instead of eight (no operand stream),
// 4 = no leaf gathers, 8 = no rescaling test at a node's finalisation, 16 = no gather for the first leaf behind an edge
// product (what a perfect prefetch under that product could hide).
#ifndef HYPHY_ABL
#define HYPHY_ABL 0
#endif
// phase accounting of the TRACE build: every shader cycle of a wave goes to exactly one bucket (the time since the previous
// mark goes to bucket b): 0 edge products, 1 their number, 2 leaf entries, 3 leaves, 4 finalisations, 5 trunk joins (arrival,
// deposit), 6 child tiles fetched from global memory, 7 wave prologue, 8 schedule-entry decode (scalar loads of the entry),
// 9 trunk bookkeeping before the edge (join records, arrival counter), 10 deposits multiplied in, 11 between entries and
// their finalisation, 12 ahead of an edge product inside an entry, 13 root epilogue / retirement
#define HYPHY_TR(b)                       \
  if constexpr (TRACE) {                  \
    const long long n_ = clock64();       \
    tr_ph[b] += n_ - tr_last, tr_last = n_; \
  }
#define HYPHY_TRACE_STAMP(k) \
  if constexpr (TRACE) tr_t[k] = wall_clock64();
#define HYPHY_TRACE_FINISH(how)                                                                                          \
  if constexpr (TRACE) {                                                                                                 \
    if (a.timeline && threadIdx.x == 0) {                                                                                \
      long long *r_ = a.timeline + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 24;        \
      r_[0] = tr_t[0], r_[1] = tr_t[1], r_[2] = tr_t[2], r_[3] = wall_clock64(), r_[4] = tr_levels, r_[5] = (how);      \
      HYPHY_TR(13)                                                                                                       \
      for (int i_ = 0; i_ < 16; i_++) r_[8 + i_] = tr_ph[i_];                                                            \
      r_[6] = __builtin_amdgcn_s_getreg((31 << 11) | 4);                                                                 \
      r_[7] = __builtin_amdgcn_s_getreg((31 << 11) | 20);                                                                \
    }                                                                                                                    \
  }

// LB: the node finalised last lives in the wave's LDS tile (`stage`) instead of 32 registers (frees registers; the next
//     product reads its B operand from LDS like a parked node's).  OCC: waves per SIMD the instantiation is compiled for.
// PRE: a sibling's deposit is streamed into registers under the wave's own product (32 registers).
// APF: the first A-operand chunk of the NEXT edge product is requested during the last k-step of the current one (the
//      schedule names it), so that an edge does not start with an exposed L2 round trip.
// REP: the tree is the TRUNK of a class-compressed partition (repeats.hip): a leaf of the schedule may be a generalised leaf —
//      its columns are rows of a class table (gathered by class id instead of state code) and carry a 2^64 exponent.
template <int NW, int NP, bool CLDS, bool TRACE = false, bool LB = false, int OCC = HYPHY_OCC3, bool PRE = true, bool APF = false, bool FUSE = false,
          bool REP = false>
__global__ __launch_bounds__(64, OCC) void prune_wave_kernel(const int4 *__restrict__ ops, const int4 *__restrict__ prog,
                                                                             const int4 *__restrict__ jn, PruneArgs a) {
  if (a.chain && (int)blockIdx.x >= a.ntiles) return;  // (tile dimension padded to a multiple of 8: launch_prune_T)
  [[maybe_unused]] long long tr_t[3] = {0, 0, 0};
  [[maybe_unused]] int tr_levels = 0;
  // TRACE: shader cycles (s_memtime) in [0] internal edge products, [1] their number, [2] leaf entries, [3] leaves,
  // [4] finalisations, [5] trunk joins (arrival bookkeeping, deposits), [6] child tiles fetched from global memory
  [[maybe_unused]] long long tr_ph[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  [[maybe_unused]] long long tr_last = 0;
  if constexpr (TRACE) tr_last = clock64();
  HYPHY_TRACE_STAMP(0)
  // legacy grid = (leaf programs, classes, tiles): tile-major dispatch order, so that a tile's chained parent
  // programs start while other tiles still run their leaf fragments (no low-occupancy tail);
  // chain grid = (tiles, classes, sources), sources sorted by their distance to the root: every tile's critical path
  // is dispatched first, the short chains that join close to the root fill the end of the launch
  int cur = a.chain ? blockIdx.z : blockIdx.x;  // program (subtree fragment / source) this wave starts with
  {
    const size_t cat = blockIdx.y;
    a.frag_ctr += cat * (size_t)a.n_prog_total * a.ntiles;
    a.hand_cnt += cat * (size_t)(a.root_inode + 1) * a.ntiles * 32;
    a.Pfrag += cat * a.cs_P;
    a.PTg += cat * a.cs_P;
    a.partials += cat * a.cs_partials;
    a.deposits += cat * a.cs_deposits;
    a.counts += cat * a.cs_counts;
    a.site_lik += cat * a.cs_site;
    a.site_cnt += cat * a.cs_site;
    a.wg_sum += cat * a.cs_wg;
    a.wg_cnt += cat * a.cs_wg;
    a.wg_flag += cat * a.cs_wg;
    if constexpr (REP) {
      a.gtab += cat * a.cs_gtab;
      a.gcnt += cat * a.cs_gcnt;
      // (NOT here: resetting the lower phase's queue heads and counters.  r05 had workgroup 0 of this launch store the zeros in
      //  its prologue; behind that store loop the compiler no longer took the schedule words for unclobbered — every schedule
      //  entry came through vector loads, every branch on it was a divergent one and each of the 312 buffer loads of the A operands
      //  sat in a waterfall loop: 7 700 instructions instead of 4 400.  repeats.hip clears the words itself where a launch used them.)
    }
  }
  constexpr int NKK = 4 * NW, DP = 16 * NW, TILE = NKK * 64;
  // LDS of a wave: NP parked nodes (scaled, fragment layout) + the tile's leaf codes; LB builds also keep the node finalised last
  // (and a child tile fetched from global memory) in `stage`.  r04: the other builds fetch child tiles straight into registers
  // and keep the parked nodes' exponents in registers, so that two parking slots cost what one slot + the staging tile did
  // (16 KiB + codes: eight waves per CU up to 128 taxa; r03's 8 + 8 KiB + 64 B + 4 KiB of codes admitted seven).
  __shared__ __align__(16) double park[(NP > 0 ? NP : 1) * TILE];
  __shared__ __align__(16) double stage_[LB ? TILE : 2];
  double *const stage = stage_;
  [[maybe_unused]] int pcnt[NP > 0 ? NP : 1];  // exponents of the parked nodes (this lane's site)
  extern __shared__ __align__(16) int16_t codes_lds[];  // CLDS: [L][16]

  const int lane = threadIdx.x, g = lane >> 4, sl = lane & 15;
  const int tile0 = a.chain ? blockIdx.x : blockIdx.z;
  const int S_pad = a.S_pad;

  if (CLDS) {  // the tile's leaf codes: one contiguous run of 32 L bytes in the tile-major table
    const int4 *src = reinterpret_cast<const int4 *>(a.codes_tile + (size_t)tile0 * a.L * 16);
    int4 *dst = reinterpret_cast<int4 *>(codes_lds);
    for (int i = lane; i < a.L * 2; i += 64) dst[i] = src[i];
    __syncthreads();
    if (a.pin_leaf >= 0 && lane < 16) codes_lds[a.pin_leaf * 16 + lane] = a.pin[tile0 * 16 + lane];  // pinned leaf
    __syncthreads();
  }
  auto leaf_code = [&](int leaf) -> int {
    if (CLDS) return (int)codes_lds[leaf * 16 + sl];
    if (leaf == a.pin_leaf) return (int)a.pin[tile0 * 16 + sl];
    return (int)a.codes_tile[((size_t)tile0 * a.L + leaf) * 16 + sl];
  };

  HYPHY_TRACE_STAMP(1)
  HYPHY_TR(7)
  const f64x4 ones = (f64x4){1., 1., 1., 1.}, zeros = (f64x4){0., 0., 0., 0.};
  f64x4 acc[NW], bch[NW];  // running product of the current parent / the node finalised last (scaled)
  int cnt = 0, bcnt = 0;
#pragma unroll
  for (int w = 0; w < NW; w++) acc[w] = ones, bch[w] = zeros;

  // acc[w'] *= sum_kk A[w'][kk] * B[kk]: `bsrc(k2)` yields the B operands of k-steps 2*k2, 2*k2 + 1.
  // Explicit two-stage software pipeline: the operands of step k2 + 1 are requested before the MFMAs of
  // step k2 are issued (sched_barrier: left alone, the scheduler sinks the loads below the MFMAs to save
  // registers and then waits for them with vmcnt(0) — eight exposed L2 round trips per edge).
  // `pre` != nullptr (wave-uniform): one 16-byte agent-scope load of a sibling's deposit rides along with every step and is
  // multiplied into the (idle) running product two steps later, behind the wait for that step's A operands, which the in-order
  // return of vector loads makes a wait for the deposit chunk too: no register image of the deposit exists (r04; the 32
  // registers of r02's image kept the kernel on the 256-register edge).
  // Leaf column gathers in flight (r04).  A gather is 2 NW 16-byte loads per lane from the transposed image of the leaf's
  // matrix; its latency (1.1 k cycles alone, up to 4 k when every wave of the CU gathers) used to be exposed once per leaf.
  // r04: the two gathers of a leaf pair are in flight together (two waves per SIMD only: the 168-register builds spill).
  // (Built, measured and removed in r04: the first leaf's gather requested inside the preceding edge product of the same parent —
  //  into registers of its own, or into the running product where that is still all ones; DESIGN §4.1, profiles/r04_wave_kernel_steps.txt.)
  auto gather_issue = [&](f64x2 (&dst)[2 * NW], int lf, int c) {
    const double *bl = a.PTg + (size_t)lf * DP * DP;  // uniform; [code][w][g][r]
    if constexpr (REP) {
      // (first row of the leaf's class table or -1, first exponent row / matrix slot.  The table is a member of the by-value
      //  argument block: its loads are VECTOR loads, and without the readfirstlane the compiler takes every address derived from
      //  them — after merging the identical product loops of the call sites, the A operands' buffer resources too — for
      //  divergent: 312 buffer loads in waterfall loops, 8 500 instructions instead of 4 000)
      int2 lt = a.leaf_tab[lf];
      lt.x = __builtin_amdgcn_readfirstlane(lt.x);
      lt.y = __builtin_amdgcn_readfirstlane(lt.y);
      bl = lt.x >= 0 ? a.gtab + (size_t)lt.x * DP : a.PTg + (size_t)lt.y * DP * DP;
      if (lt.x >= 0) cnt += a.gcnt[lt.y + c];
    }
#pragma unroll
    for (int w = 0; w < NW; w++) {
      const unsigned off = (unsigned)((c * NW + w) * 16 + g * 4) * 8u;
      dst[2 * w] = ld16(bl, off), dst[2 * w + 1] = ld16(bl, off + 16u);
    }
  };
  auto gather_apply = [&](const f64x2 (&src)[2 * NW]) {
#pragma unroll
    for (int w = 0; w < NW; w++) acc[w] *= (f64x4){src[2 * w][0], src[2 * w][1], src[2 * w + 1][0], src[2 * w + 1][1]};
  };
  [[maybe_unused]] bool abl_after_edge = false;  // (HYPHY_ABL & 16: the first leaf behind an edge product is free)
  [[maybe_unused]] const int abl_mask = (a.ablate & 4096) ? 0 : -1;  // (HYPHY_ABL == 1 builds only)
  int polled = 0;  // (lane 0) arrival counter sampled near the end of an edge product, see the trunk loop
  [[maybe_unused]] f64x2 Apre[NW];   // APF: first A chunk of branch `apre_branch`, requested under the previous product
  [[maybe_unused]] int apre_branch = -1;
  auto edge_product = [&](int branch, auto bsrc, const double *pre, const int *poll = nullptr, int next_branch = -1) {
    HYPHY_TR(12)
    const double *pf = a.Pfrag + (size_t)((HYPHY_ABL & 1) ? (branch & abl_mask) : branch) * NW * TILE;  // uniform
    const __amdgpu_buffer_rsrc_t pfr = agent_rsrc(pf);
    const unsigned lane16 = (unsigned)lane * 16u;
    f64x4 D[NW];
#pragma unroll
    for (int w = 0; w < NW; w++) D[w] = zeros;
    f64x2 Ac[NW], An[NW], bc, bn;
    [[maybe_unused]] f64x2 dq[NKK / 2];  // deposit chunks in flight (three at a time)
    if (APF && apre_branch == branch) {
#pragma unroll
      for (int w = 0; w < NW; w++) Ac[w] = Apre[w];
    } else {
#pragma unroll
      for (int w = 0; w < NW; w++) Ac[w] = ld16_buf(pfr, lane16, (unsigned)(w * TILE * 8));
    }
    bc = bsrc(0);
#pragma unroll
    for (int k2 = 0; k2 < NKK / 2; k2++) {
      if (k2 + 1 < NKK / 2) {
#pragma unroll
        for (int w = 0; w < NW; w++) An[w] = (HYPHY_ABL & 2) ? Ac[w] : ld16_buf(pfr, lane16, (unsigned)((w * TILE + (k2 + 1) * 128) * 8));
        bn = bsrc(k2 + 1);
      }
      if (pre) {
        dq[k2] = ld16_agent(pre, (unsigned)(k2 * 64 + lane) * 16u);
        if (k2 >= 2) acc[(k2 - 2) >> 1][((k2 - 2) & 1) * 2] *= dq[k2 - 2][0], acc[(k2 - 2) >> 1][((k2 - 2) & 1) * 2 + 1] *= dq[k2 - 2][1];
      }
      if (APF && k2 == NKK / 2 - 1 && next_branch >= 0) {
        const double *pn = a.Pfrag + (size_t)next_branch * NW * TILE;  // uniform
#pragma unroll
        for (int w = 0; w < NW; w++) Apre[w] = ld16(pn, (unsigned)((w * TILE + lane * 2) * 8));
      }
      if (poll && k2 == NKK / 2 - 2 && lane == 0) polled = __hip_atomic_load(poll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int w = 0; w < NW; w++) D[w] = mfma(Ac[w][0], bc[0], D[w]);
#pragma unroll
      for (int w = 0; w < NW; w++) D[w] = mfma(Ac[w][1], bc[1], D[w]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int w = 0; w < NW; w++) Ac[w] = An[w];
      bc = bn;
    }
    if (pre) {
#pragma unroll
      for (int k2 = (NKK / 2 >= 2 ? NKK / 2 - 2 : 0); k2 < NKK / 2; k2++)
        acc[k2 >> 1][(k2 & 1) * 2] *= dq[k2][0], acc[k2 >> 1][(k2 & 1) * 2 + 1] *= dq[k2][1];
    }
#pragma unroll
    for (int w = 0; w < NW; w++) acc[w] *= D[w];
    if (HYPHY_ABL & 16) abl_after_edge = true;
    if (APF) apre_branch = next_branch;
    if constexpr (TRACE) {
      asm volatile("" ::"v"(acc[0][0]));  // (the stamp waits for the product)
      tr_ph[1]++;
    }
    HYPHY_TR(0)
  };

  // Interpreter of one run of schedule entries.  `own`: (trunk nodes of a chain schedule) internal index of the child
  // whose edge product this wave holds in `acc` already; `have_pre`: the deposit of the first OTHER child is in dreg.
  auto run_ops = [&](const int4 *__restrict__ pops, int n_ops, int own, bool have_pre, int pre_cnt, int last_next = -1) {
  int4 op = pops[0];
  for (int oi = 0; oi < n_ops; oi++) {
    const int4 nxt = pops[oi + 1];
    const int kind = op.x & 3;
    if constexpr (TRACE) asm volatile("" ::"s"(kind));
    HYPHY_TR(8)
    // APF: the edge after this one, when the schedule's next entry is an internal edge (or, behind the run's last entry,
    // the caller's hint: the first trunk edge above a source)
    int nb = -1;
    if (APF) {
      const int nk = nxt.x & 3;
      if (oi + 1 < n_ops) nb = (nk == OPK_INTERNAL || nk == OPK_INTERNAL_GLOBAL) ? nxt.z : -1;
      else nb = last_next;
    }
    if (kind == OPK_LEAF) {
      const int nl = (op.x >> 8) & 0x3f;
      if (nl == 0) {
        // (padding entry of a program)
      } else if (!(op.x & OPF_AMBIG) && !(HYPHY_ABL & (4 | 16))) {
        // no ambiguity codes in this leaf group: column gathers only.  A second leaf's columns are requested before the
        // first's are consumed.
        const int lf0 = op.z & 0xffff, lf1 = (op.z >> 16) & 0xffff;
        f64x2 g0[2 * NW];
        {
          const int c0 = leaf_code(lf0);
          gather_issue(g0, lf0, c0 < 0 ? 0 : c0);
        }
        if constexpr (OCC < 3) {
          f64x2 g1[2 * NW];
          if (nl > 1) {
            const int c1 = leaf_code(lf1);
            gather_issue(g1, lf1, c1 < 0 ? 0 : c1);
          }
          gather_apply(g0);
          if (nl > 1) gather_apply(g1);
        } else {
          gather_apply(g0);
          if (nl > 1) {
            const int c1 = leaf_code(lf1);
            gather_issue(g0, lf1, c1 < 0 ? 0 : c1);
            gather_apply(g0);
          }
        }
      } else
      for (int i = 0; i < nl; i++) {
        const int lf = (op.z >> (16 * i)) & 0xffff;
        const int c = leaf_code(lf);
        if (!(op.x & OPF_AMBIG) || !__any(c < 0)) {
          if (!(HYPHY_ABL & 4) && !((HYPHY_ABL & 16) && i == 0 && abl_after_edge)) {
            f64x2 g0[2 * NW];
            gather_issue(g0, lf, c < 0 ? 0 : c);
            gather_apply(g0);
          }
          abl_after_edge = false;
        } else {  // ambiguity codes in this tile: full product with the resolution vector
          const double *av = a.ambig + (size_t)(c < 0 ? -c - 1 : 0) * DP;
          edge_product(REP ? __builtin_amdgcn_readfirstlane(a.leaf_tab[lf].y) : lf, [&](int k2) -> f64x2 {
            f64x2 b;
            b[0] = (c >= 0) ? ((8 * k2 + g == c) ? 1.0 : 0.0) : av[8 * k2 + g];
            b[1] = (c >= 0) ? ((8 * k2 + 4 + g == c) ? 1.0 : 0.0) : av[8 * k2 + 4 + g];
            return b;
          }, nullptr);
        }
      }
      if constexpr (TRACE) {
        asm volatile("" ::"v"(acc[0][0]));
        tr_ph[3] += nl;
      }
      HYPHY_TR(2)
    } else if (kind == OPK_INTERNAL) {
      const int slot = (op.x >> 24) & 0xff;
      if (slot < 2) {  // the node finalised by the previous entry: operand straight from registers (LB: from its LDS tile)
        edge_product(op.z, [&](int k2) -> f64x2 {
          if constexpr (LB) return *reinterpret_cast<const f64x2 *>(stage + (k2 * 64 + lane) * 2);
          else return (f64x2){bch[k2 >> 1][(k2 & 1) * 2], bch[k2 >> 1][(k2 & 1) * 2 + 1]};
        }, nullptr, nullptr, nb);
        cnt += bcnt;
      } else {
        const double *src = park + (slot - 2) * TILE;
        int ccnt = pcnt[0];
#pragma unroll
        for (int q = 1; q < NP; q++) ccnt = (slot - 2 == q) ? pcnt[q] : ccnt;
        edge_product(op.z, [&](int k2) -> f64x2 {
          return *reinterpret_cast<const f64x2 *>(src + (k2 * 64 + lane) * 2);
        }, nullptr, nullptr, nb);
        cnt += ccnt;
      }
    } else if (kind == OPK_DEP) {
      // trunk node of a chain schedule: the product of the edge from internal child op.w was deposited (agent-scope
      // stores, drained before its arrival was counted) by the wave that computed it — unless that wave is this one
      if (op.w != own) {
        const double *src = a.deposits + ((size_t)op.w * a.ntiles + tile0) * TILE;  // uniform
        int dcnt = pre_cnt;  // (have_pre: the deposit went into acc under this wave's own product already)
        bool skip_load = have_pre;
        if constexpr (TRACE) skip_load = skip_load || (a.ablate & 512) != 0;
        if (!skip_load) {
          dcnt = __hip_atomic_load(a.hand_cnt + ((size_t)op.w * a.ntiles + tile0) * 32 + sl, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
          f64x2 dreg[NKK / 2];
#pragma unroll
          for (int k2 = 0; k2 < NKK / 2; k2++) dreg[k2] = ld16_agent(src, (unsigned)(k2 * 64 + lane) * 16u);
#pragma unroll
          for (int w = 0; w < NW; w++)
            acc[w] *= (f64x4){dreg[2 * w][0], dreg[2 * w][1], dreg[2 * w + 1][0], dreg[2 * w + 1][1]};
        }
        have_pre = false;
        cnt += dcnt;
      }
      if constexpr (TRACE) asm volatile("" ::"v"(acc[0][0]));
      HYPHY_TR(10)
    } else {
      // The child's tile is in global memory — the root of a child fragment finished by another workgroup
      // of this launch (agent-scope loads; its arrival was counted before this program started), or a node
      // not recomputed by this program / one that found no parking slot (persisted copy).  Bulk copy into
      // LDS first (all loads in flight together: ONE memory round trip instead of one per k-step).
      const double *src = a.partials + ((size_t)op.w * a.ntiles + tile0) * TILE;  // uniform
      int ccnt;
      f64x2 gq[NKK / 2];  // the child's tile: the B-operand image, in registers (LB builds: through `stage`)
      if (op.x & OPF_HANDOFF) {
        ccnt = __hip_atomic_load(a.hand_cnt + ((size_t)op.w * a.ntiles + tile0) * 32 + sl, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int k2 = 0; k2 < NKK / 2; k2++) gq[k2] = ld16_agent(src, (unsigned)(k2 * 64 + lane) * 16u);
      } else {
        if (op.x & OPF_GSYNC) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // own stores visible in L2
        ccnt = a.counts[(size_t)op.w * S_pad + tile0 * 16 + sl];
#pragma unroll
        for (int k2 = 0; k2 < NKK / 2; k2++) gq[k2] = ld16(src, (unsigned)(k2 * 64 + lane) * 16u);
      }
      if constexpr (TRACE) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      HYPHY_TR(6)
      if constexpr (LB) {
#pragma unroll
        for (int k2 = 0; k2 < NKK / 2; k2++) *reinterpret_cast<f64x2 *>(stage + (k2 * 64 + lane) * 2) = gq[k2];
        edge_product(op.z, [&](int k2) -> f64x2 { return *reinterpret_cast<const f64x2 *>(stage + (k2 * 64 + lane) * 2); },
                     nullptr, nullptr, nb);
      } else {
        edge_product(op.z, [&](int k2) -> f64x2 { return gq[k2]; }, nullptr, nullptr, nb);
      }
      cnt += ccnt;
    }

    if (op.x & OPF_LAST) {
      const int slot = (op.x >> 16) & 0xff;
      HYPHY_TR(11)
      if (op.y == a.pin_inode) {  // pinned internal node: only the pinned state survives (tree_evaluator.cpp:589-594)
        const int ps = (int)a.pin[tile0 * 16 + sl];
#pragma unroll
        for (int w = 0; w < NW; w++)
#pragma unroll
          for (int r = 0; r < 4; r++) acc[w][r] = (16 * w + 4 * r + g == ps) ? acc[w][r] : 0.;
      }
      double sc = 1.0;
      if (!(op.x & OPF_NOSCALE)) {  // (the host thins the tests out where underflow is impossible: api.hip thin_rescale_tests)
        double s = 0.;
#pragma unroll
        for (int w = 0; w < NW; w++) s += (acc[w][0] + acc[w][1]) + (acc[w][2] + acc[w][3]);
        const double tot = (HYPHY_ABL & 8) ? 1.0 : row_sum4(s);
        int m = 0;
        if (__any(!(tot >= kScalerThreshold && tot <= kScalerUp))) m = rescale_decision(tot, sc);  // rare
        cnt += m;
      }
      double *out = a.partials + ((size_t)op.y * a.ntiles + tile0) * TILE;  // uniform
#pragma unroll
      for (int w = 0; w < NW; w++) {
        bch[w] = acc[w] * sc;
        acc[w] = ones;
      }
      if constexpr (LB) {
#pragma unroll
        for (int w = 0; w < NW; w++) {
          *reinterpret_cast<f64x2 *>(stage + ((2 * w) * 64 + lane) * 2) = (f64x2){bch[w][0], bch[w][1]};
          *reinterpret_cast<f64x2 *>(stage + ((2 * w + 1) * 64 + lane) * 2) = (f64x2){bch[w][2], bch[w][3]};
        }
      }
      if (op.x & OPF_PUBLISH) {  // fragment root: another workgroup may consume it in this launch
#pragma unroll
        for (int w = 0; w < NW; w++) {
          st16_agent(out, (unsigned)((2 * w) * 64 + lane) * 16u, (f64x2){bch[w][0], bch[w][1]});
          st16_agent(out, (unsigned)((2 * w + 1) * 64 + lane) * 16u, (f64x2){bch[w][2], bch[w][3]});
        }
        if (g == 0)
          __hip_atomic_store(a.hand_cnt + ((size_t)op.y * a.ntiles + tile0) * 32 + sl, cnt, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
      } else if (!(op.x & OPF_NOPERSIST)) {  // (lazy persistence: the host knows nobody re-reads this node)
#pragma unroll
        for (int w = 0; w < NW; w++) {
          st16(out, (unsigned)((2 * w) * 64 + lane) * 16u, (f64x2){bch[w][0], bch[w][1]});
          st16(out, (unsigned)((2 * w + 1) * 64 + lane) * 16u, (f64x2){bch[w][2], bch[w][3]});
        }
      }
      if (g == 0 && !(op.x & OPF_NOPERSIST)) a.counts[(size_t)op.y * S_pad + tile0 * 16 + sl] = cnt;
      if (NP > 0 && slot >= 2) {  // park for a later parent
        double *dst = park + (slot - 2) * TILE;
#pragma unroll
        for (int w = 0; w < NW; w++) {
          *reinterpret_cast<f64x2 *>(dst + ((2 * w) * 64 + lane) * 2) = (f64x2){bch[w][0], bch[w][1]};
          *reinterpret_cast<f64x2 *>(dst + ((2 * w + 1) * 64 + lane) * 2) = (f64x2){bch[w][2], bch[w][3]};
        }
#pragma unroll
        for (int q = 0; q < NP; q++) pcnt[q] = (slot - 2 == q) ? cnt : pcnt[q];
      }
      bcnt = cnt;
      cnt = 0;
      if constexpr (TRACE) asm volatile("" ::"v"(bch[0][0]));
      HYPHY_TR(4)
    }
    op = nxt;
  }
  };

 int4 prg;
 for (;;) {  // chained fragments: run program `cur`, then possibly its parent program
  prg = prog[cur];  // (scalar load: uniform control flow, schedule entries in SGPRs)
  run_ops(ops + prg.x, prg.y, -1, false, 0, (a.chain && jn[prg.w].x >= 0) ? (jn[prg.w].x >> 16) : -1);
  if (a.chain || prg.z < 0) break;  // a chain source / the root program (or a stand-alone one)
  // arrival at the parent program: the wave that completes the parent's last child fragment (for this
  // tile) continues with the parent; every other wave retires.  Payload stores were write-through.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int *ctr = a.frag_ctr + (size_t)prg.z * a.ntiles + tile0;
  int old = 0;
  if (lane == 0) old = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  old = __builtin_amdgcn_readfirstlane(old);
  asm volatile("" ::: "memory");
  if (old + 1 < prog[prg.z].w) {
    HYPHY_TRACE_FINISH(2)
    return;
  }
  if (lane == 0) __hip_atomic_store(ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
  cur = prg.z;
 }
  HYPHY_TRACE_STAMP(2)

  if (a.chain) {
    // The trunk: `c` = internal index of the node whose (scaled) conditionals are in bch.  Per level: the edge product
    // towards the parent p, the arrival at p, and — for the last arriver only — p's remaining children and its
    // finalisation.  A wave that finds every sibling already arrived (relaxed read of the arrival counter BEFORE its
    // own product) knows it will be last: it skips its own deposit and streams the sibling's deposit in under its
    // MFMAs, so that a join costs the critical path one counter read instead of a ~4 us hand-off.
    int c = prg.w;
    // (experiment, -DHYPHY_TRUNK_PRIO=3: trunk waves carry their tile's critical path — let them win the SIMD's issue arbitration)
#ifdef HYPHY_TRUNK_PRIO
    __builtin_amdgcn_s_setprio(HYPHY_TRUNK_PRIO);
#endif
    int early = -1;  // arrival counter of c's parent, sampled (lane 0) while c itself was still being finalised
    for (;;) {
      const int4 jc = jn[c];  // x = parent (internal index) | matrix-image slot of the edge c -> parent << 16; -1: c is the root
      if (jc.x < 0) break;  // c is the root: epilogue below
      const int p = jc.x & 0xffff;
      const int4 jp = jn[p];
      const int need = jp.y & 0xff;
      int *ctr = a.frag_ctr + (size_t)p * a.ntiles + tile0;
      bool last = need <= 1;
      const double *pre = nullptr;
      int pre_cnt = 0;
      if (!last) {
        int seen = early;
        if (seen < 0 && lane == 0) seen = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        seen = __builtin_amdgcn_readfirstlane(seen);
        asm volatile("" ::: "memory");
        if (seen == need - 1) {
          last = true;
          if (PRE && need == 2) {  // the one sibling: its deposit streams in under this wave's own product
            const int sib = (jp.y >> 8) - c;
            pre = a.deposits + ((size_t)sib * a.ntiles + tile0) * TILE;
            pre_cnt = __hip_atomic_load(a.hand_cnt + ((size_t)sib * a.ntiles + tile0) * 32 + sl, __ATOMIC_RELAXED,
                                        __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
      HYPHY_TR(9)
      edge_product(jc.x >> 16, [&](int k2) -> f64x2 {
        if constexpr (LB) return *reinterpret_cast<const f64x2 *>(stage + (k2 * 64 + lane) * 2);
        else return (f64x2){bch[k2 >> 1][(k2 & 1) * 2], bch[k2 >> 1][(k2 & 1) * 2 + 1]};
      }, pre, last ? nullptr : ctr, jp.x >= 0 ? (jp.x >> 16) : -1);
      cnt = bcnt;
      if (!last) {
        // (the counter was sampled two k-steps before the end of the product: its round trip is covered)
        const int seen = __builtin_amdgcn_readfirstlane(polled);
        asm volatile("" ::: "memory");
        if (seen != need - 1) {
          // deposit the product (write-through), drain, count the arrival.
          // Memory-ordering contract of every hand-off between waves of this launch (deposits here, fragment roots above):
          //   producer  16-byte sc1 (write-through) payload stores -> asm "s_waitcnt vmcnt(0)" (an asm statement with a memory
          //             clobber: the hardware drain AND a compiler barrier, invisible to the waitcnt-elision pass) -> relaxed
          //             agent-scope RMW on the arrival counter;
          //   consumer  the RMW / relaxed load that proves every producer has arrived -> compiler barrier -> sc1 loads (they
          //             bypass the reader's L1, so no agent-scope acquire — no L1 invalidate — is needed).
          // This is the "sc1 payload -> asm vmcnt(0) -> flag, sc1 loads on the reading side" form of MI355X_MICROARCH.md
          // ("Valid forms"), valid under any workgroup -> XCD placement; a release/acquire pair on the counter would add a
          // buffer_wbl2 and a buffer_inv (1.7 us each) to every join for accesses that never touch L1 / dirty L2 lines.
          // tests/test_gpu_parity.py::test_chain_joins_within_and_across_xcds drives it with joins forced across / inside XCDs.
          double *out = a.deposits + ((size_t)c * a.ntiles + tile0) * TILE;  // uniform
          bool skip_store = false;
          if constexpr (TRACE) skip_store = (a.ablate & 256) != 0;
          if (!skip_store)
#pragma unroll
          for (int w = 0; w < NW; w++) {
            st16_agent(out, (unsigned)((2 * w) * 64 + lane) * 16u, (f64x2){acc[w][0], acc[w][1]});
            st16_agent(out, (unsigned)((2 * w + 1) * 64 + lane) * 16u, (f64x2){acc[w][2], acc[w][3]});
          }
          if (g == 0)
            __hip_atomic_store(a.hand_cnt + ((size_t)c * a.ntiles + tile0) * 32 + sl, cnt, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          int old = 0;
          if (lane == 0) old = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          old = __builtin_amdgcn_readfirstlane(old);
          asm volatile("" ::: "memory");
          if (old + 1 < need) {  // somebody else will finish p
            HYPHY_TRACE_FINISH(0)
            return;
          }
          if constexpr (TRACE) tr_levels += 1000;
        }
      }
      if (need > 1 && lane == 0) __hip_atomic_store(ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next launch
      // sample the NEXT level's arrival counter now: the round trip hides behind p's leaves, deposits and finalisation
      early = -1;
      if (jp.x >= 0 && (jn[jp.x & 0xffff].y & 0xff) > 1 && lane == 0)
        early = __hip_atomic_load(a.frag_ctr + (size_t)(jp.x & 0xffff) * a.ntiles + tile0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      HYPHY_TR(5)
      run_ops(ops + jp.z, jp.w, c, pre != nullptr, pre_cnt);
      if constexpr (TRACE) tr_levels += 1 + (pre ? 100 : 0);
      c = p;
    }
  }

  if (a.do_root) {
    // root: L_s = sum_k root[s][k] pi[k]; bch holds the (scaled) root conditionals, bcnt its exponent
    double s = 0.;
#pragma unroll
    for (int kk = 0; kk < NKK; kk++) {
      double rv;
      if constexpr (LB) rv = stage[frag_index(kk, lane)];
      else rv = bch[kk >> 2][kk & 3];
      s = fma(rv, a.pi[4 * kk + g], s);
    }
    s = row_sum4(s);
    double wsum = 0.;
    long long wcnt = 0;
    int wflag = 0;
    if (g == 0) {
      const int site = tile0 * 16 + sl;
      a.site_lik[site] = s;
      a.site_cnt[site] = bcnt;
      const double f = a.freq[site];
      if (f != 0.) {
        if (s != s || isinf(s)) wflag |= 2;
        else if (s <= 0.) wflag |= 1;
        else {
          wsum += log(s) * f;
          wcnt += (long long)bcnt * (long long)f;
        }
      }
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {  // fixed-order butterfly over the 16 site lanes
      wsum += __shfl_xor(wsum, off);
      wcnt += __shfl_xor(wcnt, off);
      wflag |= __shfl_xor(wflag, off);
    }
    publish_partial<FUSE>(a, tile0, wsum, wcnt, wflag, lane);
  }
  HYPHY_TRACE_FINISH(1)
}

// ---------------------------------------------------------------------------------------------
// Branch cache (device counterpart of _TheTree::ComputeBranchCache tree_evaluator.cpp:4286-4845 and
// _TheTree::ComputeLLWithBranchCache tree.cpp:3383-3936).  While the optimiser varies ONE branch length,
// L_s(t) = sum_i A_s[i] sum_j P_c(t)[i][j] B_s[j]: B = conditionals of the branch's child node, A = the
// rest of the tree seen from the branch's parent (root frequencies and all other subtrees folded in).
// A is produced by the ordinary pruning kernel run over the tree RE-ROOTED at the parent: the edges
// between the old root and the parent are traversed upside down, i.e. with the transposed transition
// matrix (no reversibility assumption, unlike the reference).  transpose_frag_kernel makes the A-operand
// image of M[j][i] = scale[i] P[i][j] from the image of P; bc_eval_kernel is the one-edge evaluation.
// ---------------------------------------------------------------------------------------------
__global__ void transpose_frag_kernel(const double *__restrict__ src, double *__restrict__ dst,
                                      const double *__restrict__ row_scale, int NW) {
  const int NKK = 4 * NW, DP = 16 * NW, TILE = NKK * 64;
  for (int idx = threadIdx.x; idx < DP * DP; idx += blockDim.x) {
    const int w = idx / TILE, rem = idx - w * TILE;
    const int k2 = rem >> 7, l = (rem >> 1) & 63, kk = 2 * k2 + (rem & 1);
    const int r = 16 * w + (l & 15), c = 4 * kk + (l >> 4);  // dst element M[r][c] = scale[c] * P[c][r]
    const double v = src[(c >> 4) * TILE + frag_index(r >> 2, (r & 3) * 16 + (c & 15))];
    dst[idx] = row_scale ? v * row_scale[c] : v;
  }
}

template <int NW>
__global__ __launch_bounds__(64) void bc_eval_kernel(BcArgs a) {
  constexpr int NKK = 4 * NW, DP = 16 * NW, TILE = NKK * 64;
  const int lane = threadIdx.x, g = lane >> 4, sl = lane & 15;
  const int tile0 = blockIdx.x;
  f64x4 prod[NW];
  int ccnt = 0;
  if (a.child_internal >= 0) {
    const double *src = a.partials + ((size_t)a.child_internal * a.ntiles + tile0) * TILE;
    ccnt = a.counts[(size_t)a.child_internal * a.S_pad + tile0 * 16 + sl];
#pragma unroll
    for (int w = 0; w < NW; w++) prod[w] = (f64x4){0., 0., 0., 0.};
#pragma unroll 2
    for (int k2 = 0; k2 < NKK / 2; k2++) {
      const f64x2 b = ld16(src, (unsigned)(k2 * 64 + lane) * 16u);
#pragma unroll
      for (int w = 0; w < NW; w++) {
        const f64x2 av = ld16(a.Pfrag + w * TILE, (unsigned)(k2 * 64 + lane) * 16u);
        prod[w] = mfma(av[0], b[0], prod[w]);
        prod[w] = mfma(av[1], b[1], prod[w]);
      }
    }
  } else {
    const int c = (int)a.codes_tile[((size_t)tile0 * a.L + a.child_leaf) * 16 + sl];
    if (!__any(c < 0)) {  // column gather, [code][w][g][r] = P[16w + 4r + g][code]
#pragma unroll
      for (int w = 0; w < NW; w++) {
        const unsigned off = (unsigned)((c * NW + w) * 16 + g * 4) * 8u;
        const f64x2 v0 = ld16(a.PTg, off), v1 = ld16(a.PTg, off + 16u);
        prod[w] = (f64x4){v0[0], v0[1], v1[0], v1[1]};
      }
    } else {  // ambiguity codes in this tile: product with the resolution vectors
      const double *av = a.ambig + (size_t)(c < 0 ? -c - 1 : 0) * DP;
#pragma unroll
      for (int w = 0; w < NW; w++) prod[w] = (f64x4){0., 0., 0., 0.};
      for (int kk = 0; kk < NKK; kk++) {
        const double bv = (c >= 0) ? ((4 * kk + g == c) ? 1.0 : 0.0) : av[4 * kk + g];
#pragma unroll
        for (int w = 0; w < NW; w++) prod[w] = mfma(a.Pfrag[w * TILE + frag_index(kk, lane)], bv, prod[w]);
      }
    }
  }
  // L_s = sum_i A_s[i] prod_s[i]  (rows 16w + 4r + g of the C/D image; A is stored as a persisted node)
  const double *Asrc = a.partials + ((size_t)a.node_A * a.ntiles + tile0) * TILE;
  double s = 0.;
#pragma unroll
  for (int w = 0; w < NW; w++) {
    const f64x2 a0 = ld16(Asrc, (unsigned)((2 * w) * 64 + lane) * 16u), a1 = ld16(Asrc, (unsigned)((2 * w + 1) * 64 + lane) * 16u);
    const double av[4] = {a0[0], a0[1], a1[0], a1[1]};
#pragma unroll
    for (int r = 0; r < 4; r++) {
      double t = av[r] * prod[w][r];
      if (a.use_pi) t *= a.pi[16 * w + 4 * r + g];
      s += t;
    }
  }
  s = row_sum4(s);
  const int rcnt = ccnt + a.counts[(size_t)a.node_A * a.S_pad + tile0 * 16 + sl];
  double wsum = 0.;
  long long wcnt = 0;
  int wflag = 0;
  if (g == 0) {
    const int site = tile0 * 16 + sl;
    a.site_lik[site] = s;
    a.site_cnt[site] = rcnt;
    const double f = a.freq[site];
    if (f != 0.) {
      if (s != s || isinf(s)) wflag |= 2;
      else if (s <= 0.) wflag |= 1;
      else {
        wsum += log(s) * f;
        wcnt += (long long)rcnt * (long long)f;
      }
    }
  }
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) {
    wsum += __shfl_xor(wsum, off);
    wcnt += __shfl_xor(wcnt, off);
    wflag |= __shfl_xor(wflag, off);
  }
  if (lane == 0) {
    a.wg_sum[tile0] = wsum;
    a.wg_cnt[tile0] = wcnt;
    a.wg_flag[tile0] = wflag;
  }
}

// ---------------------------------------------------------------------------------------------
// 4-state (nucleotide) kernel: one thread per site pattern walks the whole schedule; P matrices
// are wave-uniform (scalar loads), conditionals live in registers and are persisted as
// state-major planes so every global access is a coalesced 512-byte line per wave.
// HBM-bound: per node 32 B/site written (+ re-read of children that are not in registers).
// ---------------------------------------------------------------------------------------------
// (ops and P are separate __restrict__ kernel arguments: schedule entries and transition matrices then come
//  through scalar loads; as members of the by-value argument struct they were 30 vector loads per entry)
// REP: the tree is the trunk of a class-compressed partition (repeats.hip): a leaf of the schedule may be a generalised leaf — its
//      column is a row of a class table (4 doubles, looked up by class id) and carries a 2^64 exponent.
template <bool PIN, bool REP = false>  // PIN: a node's states are pinned (hyphy_hip_set_pinned_states); compiled apart — the extra
                      // select per leaf entry costs the HBM-bound kernel 30 % (54 vs 41 us at gtr_32x50k)
__global__ __launch_bounds__(256) void prune_nuc_kernel(const int4 *__restrict__ ops, const double *__restrict__ Pm,
                                                        NucArgs a) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;  // S_pad is a multiple of the block size
  const size_t S_pad = a.S_pad;
  // pending nodes (finished, parent not next) stay on chip: every thread parks its own column in LDS
  __shared__ double park[kNucParkSlots][4][256];
  __shared__ int park_cnt[kNucParkSlots][256];
  double acc[4] = {1., 1., 1., 1.}, b[4] = {0., 0., 0., 0.};
  int cnt = 0, bcnt = 0;
  // One entry ahead: the schedule word, the 16 entries of its transition matrix (scalar loads) and, for a leaf,
  // this thread's code — at small shard sizes (< 1 wave per SIMD) the entry-to-entry chain of dependent loads
  // is the whole run time.  Programs are followed by two no-op entries, so oi + 1 is always readable.
  auto child_of = [&](const int4 &o) {  // (matrix slot of the entry's branch)
    if ((o.x & 3) != OPK_LEAF) return o.z;
    if constexpr (REP) {
      const int2 lt = a.leaf_tab[o.z & 0xffff];
      return lt.x >= 0 ? 0 : lt.y;  // (a generalised leaf needs no matrix: its table rows are edge-applied already)
    }
    return o.z & 0xffff;
  };
  int4 op = ops[0];
  double P[16];
#pragma unroll
  for (int e = 0; e < 16; e++) P[e] = Pm[(size_t)child_of(op) * 16 + e];
  auto code_of = [&](const int4 &o) -> int {
    if ((o.x & 3) != OPK_LEAF) return 0;
    const int lf = o.z & 0xffff;
    if (PIN && lf == a.pin_leaf) return (int)a.pin[s];  // (pinned leaf: its states replace the data)
    return (int)a.codes[(size_t)lf * S_pad + s];
  };
  int code = code_of(op);
  for (int oi = 0; oi < a.n_ops; oi++) {
    const int4 nxt = ops[oi + 1];
    double Pn[16];
#pragma unroll
    for (int e = 0; e < 16; e++) Pn[e] = Pm[(size_t)child_of(nxt) * 16 + e];
    const int code_n = code_of(nxt);
    const int kind = op.x & 3, parent = op.y;
    const bool is_leaf = kind == OPK_LEAF;
    if (!(is_leaf && ((op.x >> 8) & 0x7f) == 0)) {  // (else: padding entry)
    double cv[4];
    bool matvec = true;
    bool gen = false;
    if constexpr (REP) {
      if (is_leaf) {
        const int2 lt = a.leaf_tab[op.z & 0xffff];
        if (lt.x >= 0) {  // generalised leaf: the class's row of the subtree root's table
          gen = true;
          matvec = false;
          const double *row = a.gtab + ((size_t)lt.x + (size_t)code) * 4;
          const f64x2 r0 = *reinterpret_cast<const f64x2 *>(row), r1 = *reinterpret_cast<const f64x2 *>(row + 2);
          acc[0] *= r0[0], acc[1] *= r0[1], acc[2] *= r1[0], acc[3] *= r1[1];
          cnt += a.gcnt[lt.x + code];
        }
      }
    }
    if (gen) {
      // (done above)
    } else if (is_leaf) {
      if (code >= 0) {
        matvec = false;
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const double p0 = P[4 * i], p1 = P[4 * i + 1], p2 = P[4 * i + 2], p3 = P[4 * i + 3];
          acc[i] *= (code == 0) ? p0 : (code == 1) ? p1 : (code == 2) ? p2 : p3;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++) cv[j] = a.ambig[(size_t)(-code - 1) * 4 + j];
      }
    } else {
      if (kind == OPK_INTERNAL) {  // parked by this thread in LDS (its parent was not the next entry)
        const int ps = ((op.x >> 24) & 0xff) - 2;
#pragma unroll
        for (int j = 0; j < 4; j++) b[j] = park[ps][j][threadIdx.x];
        bcnt = park_cnt[ps][threadIdx.x];
      } else if (!(op.x & OPF_INREGS)) {
        const size_t base = (size_t)op.w * 4 * S_pad + s;
#pragma unroll
        for (int j = 0; j < 4; j++) b[j] = a.partials[base + j * S_pad];
        bcnt = a.counts[(size_t)op.w * S_pad + s];
      }
#pragma unroll
      for (int j = 0; j < 4; j++) cv[j] = b[j];
      cnt += bcnt;
    }
    if (matvec) {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        double m = P[4 * i] * cv[0];
        m = fma(P[4 * i + 1], cv[1], m);
        m = fma(P[4 * i + 2], cv[2], m);
        m = fma(P[4 * i + 3], cv[3], m);
        acc[i] *= m;
      }
    }
    if (op.x & OPF_LAST) {
      if (PIN && parent == a.pin_inode) {  // pinned internal node: only the pinned state survives
        const int ps = (int)a.pin[s];
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] = (i == ps) ? acc[i] : 0.;
      }
      const double tot = (acc[0] + acc[1]) + (acc[2] + acc[3]);
      double sc;
      const int m = rescale_decision(tot, sc);
#pragma unroll
      for (int j = 0; j < 4; j++) b[j] = (m != 0) ? acc[j] * sc : acc[j];
      cnt += m;
      bcnt = cnt;
      const int dslot = (op.x >> 16) & 0xff;
      if (dslot >= 2) {  // pending node: its parent comes later in the schedule
#pragma unroll
        for (int j = 0; j < 4; j++) park[dslot - 2][j][threadIdx.x] = b[j];
        park_cnt[dslot - 2][threadIdx.x] = cnt;
      }
      if (!(op.x & OPF_NOPERSIST_NUC)) {  // (lazy persistence: the host knows nobody re-reads this node)
        const size_t base = (size_t)parent * 4 * S_pad + s;
#pragma unroll
        for (int j = 0; j < 4; j++) a.partials[base + j * S_pad] = b[j];
        a.counts[(size_t)parent * S_pad + s] = cnt;
      }
      acc[0] = acc[1] = acc[2] = acc[3] = 1.;  // the next entry starts a new parent
      cnt = 0;
    }
    }
    op = nxt;
    code = code_n;
#pragma unroll
    for (int e = 0; e < 16; e++) P[e] = Pn[e];
  }
  __shared__ double rs[256];
  __shared__ long long rc[256];
  __shared__ int rf;
  if (threadIdx.x == 0) rf = 0;
  __syncthreads();
  double term = 0.;
  long long tc = 0;
  if (a.n_ops > 0 && s < a.S_pad) {
    double L = b[0] * a.pi[0];
    L = fma(b[1], a.pi[1], L);
    L = fma(b[2], a.pi[2], L);
    L = fma(b[3], a.pi[3], L);
    a.site_lik[s] = L;
    a.site_cnt[s] = bcnt;
    const double f = a.freq[s];
    if (f != 0.) {
      if (L != L || isinf(L)) atomicOr(&rf, 2);
      else if (L <= 0.) atomicOr(&rf, 1);
      else {
        term = log(L) * f;
        tc = (long long)bcnt * (long long)f;
      }
    }
  }
  rs[threadIdx.x] = term;
  rc[threadIdx.x] = tc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      rs[threadIdx.x] += rs[threadIdx.x + off];
      rc[threadIdx.x] += rc[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0 && a.n_ops > 0) {
    a.wg_sum[blockIdx.x] = rs[0];
    a.wg_cnt[blockIdx.x] = rc[0];
    a.wg_flag[blockIdx.x] = rf;
  }
}

// ---------------------------------------------------------------------------------------------
// 4-state kernel, r03 (`prune_nuc2_kernel<NP, PIN>`).  r02's kernel above was bound by neither memory (0.5 TB/s) nor
// arithmetic: ~40 scalar + vector instructions per schedule entry and pattern, most of them moving the matrix between
// SGPR sets and selecting a leaf's column with compare/select chains.  What changed:
//  * a leaf edge is a LOOKUP: the transposed leaf matrices sit in LDS ([leaf][state][row], 128 bytes per leaf, filled once per
//    workgroup) and a thread fetches the four entries of its pattern's column with two 16-byte reads — the four possible
//    columns of a leaf cover all 32 banks exactly once, so any mix of states in a wave is conflict-free;
//  * an internal edge uses the row-stochastic form of the reference's 4-state path (_handle4x4_pruning_case_direct,
//    tree_evaluator.cpp:2253-2273): (P v)_i = (c0 P_i0 + c1 P_i1) + (v_3 + c2 P_i2), c_j = v_j - v_3 — 12 multiply-adds and
//    3 subtractions, the matrix arriving through scalar loads one entry ahead (loop unrolled by two: no register moves);
//  * NP patterns per thread share every scalar instruction, branch and wait of an entry and give the vector unit NP
//    independent chains;
//  * the rescaling test costs one wave ballot per finalisation unless some pattern really is out of range;
//  * block reduction by wave shuffles + one LDS round instead of an 8-barrier tree.
// Same schedule format, same exponent bookkeeping, same outputs as prune_nuc_kernel (which stays for trees whose leaf
// matrices do not fit LDS).
// ---------------------------------------------------------------------------------------------
// FOLD: the matrix exponentials of this evaluation are computed HERE, by the first threads of every workgroup (each writes the
// transposed copies it reads itself; workgroup 0 also leaves the row-major ones), instead of by a launch of their own in
// front — for shards of a few hundred workgroups the 10 us of that launch were a fifth of the step, 2 us of one wave are not.
// LP (late r03): the schedule words and the transposed matrices of ALL branches are copied to LDS once per workgroup and every
// per-entry fetch is an LDS read (uniform address, one entry / two entries ahead as before).  Scalar loads return out of order
// and share their counter with LDS: every wait for a leaf's LDS lookup or for this entry's matrix is an lgkmcnt(0) that also
// waits for the scalar loads just issued for the NEXT entry — with (less than) one wave per SIMD that exposed a scalar-load
// latency per entry.  LDS reads return in order: the waits become partial.  Costs ~7 wide LDS reads per entry and wave, so
// only shards of at most two workgroups per CU use it (launch_prune_nuc); the 10^6-site launches keep the scalar path.
template <int NP, bool PIN, bool FOLD = false, bool LP = false>
__global__ __launch_bounds__(256) void prune_nuc2_kernel(const int4 *__restrict__ ops, const double *__restrict__ PTm,
                                                         NucArgs a, ExpmArgs ex, typename CoefArg<FOLD>::type ci) {
  constexpr int WGP = 256 * NP;  // patterns per workgroup
  extern __shared__ __align__(16) double nlds[];
  const int nPT = LP ? a.L + a.root_inode : a.L;                       // matrices kept in LDS (LP: every branch)
  double *PT = nlds;                                                   // [nPT][4 states][4 rows]
  double *park = nlds + (size_t)nPT * 16;                              // [slot][NP][4][256]
  int *park_cnt = reinterpret_cast<int *>(park + kNucParkSlots * NP * 4 * 256);  // [slot][NP][256]
  [[maybe_unused]] int4 *ops_l = reinterpret_cast<int4 *>(park_cnt + kNucParkSlots * NP * 256);  // LP: [n_ops + 4]
  const int tid = threadIdx.x;
  const size_t S_pad = a.S_pad;
  const int s0 = blockIdx.x * WGP + tid;  // pattern q of this thread: s0 + 256 q
  if constexpr (FOLD && !LP) {
    for (int m = tid; m < ex.n; m += 256) {
      double R[16];
      expm4_one(ex, m, R, ex.coef_inline ? ci.c : ex.coeffs);
      const int slot = ex.slots ? ex.slots[m] : m;
      if (blockIdx.x == 0 && ex.Prow) {
#pragma unroll
        for (int k = 0; k < 16; k++) ex.Prow[(size_t)slot * 16 + k] = R[k];
      }
#pragma unroll
      for (int k = 0; k < 16; k++) ex.PTrow[(size_t)slot * 16 + k] = R[4 * (k & 3) + (k >> 2)];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this workgroup's copies have reached L2 before any of its waves reads them
    __syncthreads();
  }
  for (int idx = tid; idx < nPT * 16; idx += 256) PT[idx] = PTm[idx];  // (leaves are branches 0 .. L-1)
  if constexpr (LP) {
    for (int idx = tid; idx < a.n_ops + 4; idx += 256) ops_l[idx] = ops[idx < a.n_ops + 2 ? idx : a.n_ops + 1];
  }
  if constexpr (FOLD && LP) {
    // this evaluation's exponentials straight into the LDS copy every entry reads (behind the copy of the resident matrices:
    // branches that did not change keep theirs); workgroup 0 also leaves the global copies for later partial updates
    __syncthreads();
    for (int m = tid; m < ex.n; m += 256) {
      double R[16];
      expm4_one(ex, m, R, ex.coef_inline ? ci.c : ex.coeffs);
      const int slot = ex.slots ? ex.slots[m] : m;
#pragma unroll
      for (int k = 0; k < 16; k++) PT[slot * 16 + k] = R[4 * (k & 3) + (k >> 2)];
      if (blockIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
          if (ex.Prow) ex.Prow[(size_t)slot * 16 + k] = R[k];
          ex.PTrow[(size_t)slot * 16 + k] = R[4 * (k & 3) + (k >> 2)];
        }
      }
    }
  }
  __syncthreads();
  auto fetch_op = [&](int k) -> int4 {
    if constexpr (LP) {
      int4 o = ops_l[k];  // (uniform address; the words steer branches: back into scalar registers)
      o.x = __builtin_amdgcn_readfirstlane(o.x);
      o.y = __builtin_amdgcn_readfirstlane(o.y);
      o.z = __builtin_amdgcn_readfirstlane(o.z);
      o.w = __builtin_amdgcn_readfirstlane(o.w);
      return o;
    } else {
      return ops[k];
    }
  };
  double acc[NP][4], b[NP][4];
  int cnt[NP], bcnt[NP];
#pragma unroll
  for (int q = 0; q < NP; q++) {
    cnt[q] = bcnt[q] = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) acc[q][j] = 1., b[q][j] = 0.;
  }
  // a leaf entry carries one or two leaves (api: leaf groups; a leaf with ambiguity codes forms a group of its own)
  auto codes_of = [&](const int4 &o, int (&code)[2][NP]) {
    const int nl = (o.x & 3) == OPK_LEAF ? ((o.x >> 8) & 0x7f) : 0;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int lf = (o.z >> (16 * i)) & 0xffff;
#pragma unroll
      for (int q = 0; q < NP; q++) {
        code[i][q] = 0;
        if (i < nl) {
          if (PIN && lf == a.pin_leaf) code[i][q] = (int)a.pin[s0 + 256 * q];  // (pinned leaf: its states replace the data)
          else code[i][q] = (int)a.codes[(size_t)lf * S_pad + s0 + 256 * q];
        }
      }
    }
  };
  // columns 0..2 of a child's matrix = the first 12 doubles of its TRANSPOSED copy (PTm: [branch][state j][row i]): uniform
  // address, three 32-byte scalar loads, requested one entry ahead.  (Tried and measured slower: the same through vector
  // loads — 96 bytes x 64 lanes per entry made the kernel TA-bound, 205 vs 120 us at 10^6 sites; schedule words through
  // vector loads + readfirstlane, 38 vs 27 us at 50 000 sites.)
  auto load_P = [&](const int4 &o, double (&P)[12]) {
    const int br = (o.x & 3) == OPK_LEAF ? (o.z & 0xffff) : o.z;
    if constexpr (LP) {
      const f64x2 *src = reinterpret_cast<const f64x2 *>(PT + (size_t)br * 16);
#pragma unroll
      for (int e = 0; e < 6; e++) {
        const f64x2 v = src[e];
        P[2 * e] = v[0], P[2 * e + 1] = v[1];
      }
    } else {
      const double *src = PTm + (size_t)br * 16;
#pragma unroll
      for (int e = 0; e < 12; e++) P[e] = src[e];
    }
  };
  // one schedule entry: `code` = this thread's leaf codes (leaf entries), P = columns 0..2 of the child's transition matrix
  auto entry = [&](const int4 &op, const double (&P)[12], const int (&code)[2][NP]) {
    const int kind = op.x & 3, parent = op.y;
    if (kind == OPK_LEAF) {
      const int nl = (op.x >> 8) & 0x7f;
      if (nl == 0) return;  // padding entry
      const int lf = op.z & 0xffff;
      bool any_ambig = false;
#pragma unroll
      for (int q = 0; q < NP; q++) any_ambig = any_ambig || code[0][q] < 0;
      if (!__any(any_ambig)) {
        const int lf1 = (op.z >> 16) & 0xffff;
        f64x2 c01[2][NP], c23[2][NP];
#pragma unroll
        for (int q = 0; q < NP; q++) {  // (all lookups of the group in flight together)
          const f64x2 *col = reinterpret_cast<const f64x2 *>(PT + lf * 16 + code[0][q] * 4);
          c01[0][q] = col[0], c23[0][q] = col[1];
          const f64x2 *col1 = reinterpret_cast<const f64x2 *>(PT + lf1 * 16 + (nl > 1 ? code[1][q] : 0) * 4);
          c01[1][q] = col1[0], c23[1][q] = col1[1];
        }
#pragma unroll
        for (int q = 0; q < NP; q++) {
          acc[q][0] *= c01[0][q][0];
          acc[q][1] *= c01[0][q][1];
          acc[q][2] *= c23[0][q][0];
          acc[q][3] *= c23[0][q][1];
        }
        if (nl > 1) {
#pragma unroll
          for (int q = 0; q < NP; q++) {
            acc[q][0] *= c01[1][q][0];
            acc[q][1] *= c01[1][q][1];
            acc[q][2] *= c23[1][q][0];
            acc[q][3] *= c23[1][q][1];
          }
        }
      } else {  // (a leaf with ambiguity codes: always a group of one)
#pragma unroll
        for (int q = 0; q < NP; q++) {
          double cv[4];
          if (code[0][q] >= 0) {
#pragma unroll
            for (int j = 0; j < 4; j++) cv[j] = (j == code[0][q]) ? 1. : 0.;
          } else {
#pragma unroll
            for (int j = 0; j < 4; j++) cv[j] = a.ambig[(size_t)(-code[0][q] - 1) * 4 + j];
          }
#pragma unroll
          for (int i = 0; i < 4; i++) {  // (rare path: the leaf's full matrix from its LDS copy, [state j][row i])
            double m = PT[lf * 16 + i] * cv[0];
            m = fma(PT[lf * 16 + 4 + i], cv[1], m);
            m = fma(PT[lf * 16 + 8 + i], cv[2], m);
            m = fma(PT[lf * 16 + 12 + i], cv[3], m);
            acc[q][i] *= m;
          }
        }
      }
    } else {
      if (kind == OPK_INTERNAL) {  // parked by this thread in LDS (its parent was not the next entry)
        const int ps = ((op.x >> 24) & 0xff) - 2;
#pragma unroll
        for (int q = 0; q < NP; q++) {
#pragma unroll
          for (int j = 0; j < 4; j++) b[q][j] = park[((ps * NP + q) * 4 + j) * 256 + tid];
          bcnt[q] = park_cnt[(ps * NP + q) * 256 + tid];
        }
      } else if (!(op.x & OPF_INREGS)) {
#pragma unroll
        for (int q = 0; q < NP; q++) {
          const size_t base = (size_t)op.w * 4 * S_pad + s0 + 256 * q;
#pragma unroll
          for (int j = 0; j < 4; j++) b[q][j] = a.partials[base + j * S_pad];
          bcnt[q] = a.counts[(size_t)op.w * S_pad + s0 + 256 * q];
        }
      }
#pragma unroll
      for (int q = 0; q < NP; q++) {
        cnt[q] += bcnt[q];
        const double c0 = b[q][0] - b[q][3], c1 = b[q][1] - b[q][3], c2 = b[q][2] - b[q][3], c3 = b[q][3];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const double t0 = fma(c1, P[4 + i], c0 * P[i]);
          const double t1 = fma(c2, P[8 + i], c3);
          acc[q][i] *= t0 + t1;
        }
      }
    }
    if (op.x & OPF_LAST) {
      bool odd = false;
      double tot[NP];
#pragma unroll
      for (int q = 0; q < NP; q++) {
        if (PIN && parent == a.pin_inode) {  // pinned internal node: only the pinned state survives
          const int ps = (int)a.pin[s0 + 256 * q];
#pragma unroll
          for (int i = 0; i < 4; i++) acc[q][i] = (i == ps) ? acc[q][i] : 0.;
        }
        tot[q] = (acc[q][0] + acc[q][1]) + (acc[q][2] + acc[q][3]);
        odd = odd || !(tot[q] >= kScalerThreshold && tot[q] <= kScalerUp);
      }
      if (__any(odd)) {  // rare: some pattern of the wave needs (or cannot have) a rescale
#pragma unroll
        for (int q = 0; q < NP; q++) {
          double sc;
          const int m = rescale_decision(tot[q], sc);
          if (m != 0) {
#pragma unroll
            for (int j = 0; j < 4; j++) acc[q][j] *= sc;
            cnt[q] += m;
          }
        }
      }
      const int dslot = (op.x >> 16) & 0xff;
#pragma unroll
      for (int q = 0; q < NP; q++) {
#pragma unroll
        for (int j = 0; j < 4; j++) b[q][j] = acc[q][j];
        bcnt[q] = cnt[q];
        if (dslot >= 2) {  // pending node: its parent comes later in the schedule
#pragma unroll
          for (int j = 0; j < 4; j++) park[(((dslot - 2) * NP + q) * 4 + j) * 256 + tid] = b[q][j];
          park_cnt[((dslot - 2) * NP + q) * 256 + tid] = cnt[q];
        }
        if (!(op.x & OPF_NOPERSIST_NUC)) {  // (lazy persistence: the host knows nobody re-reads this node)
          const size_t base = (size_t)parent * 4 * S_pad + s0 + 256 * q;
#pragma unroll
          for (int j = 0; j < 4; j++) a.partials[base + j * S_pad] = b[q][j];
          a.counts[(size_t)parent * S_pad + s0 + 256 * q] = cnt[q];
        }
        acc[q][0] = acc[q][1] = acc[q][2] = acc[q][3] = 1.;  // the next entry starts a new parent
        cnt[q] = 0;
      }
    }
  };
  // Software pipeline, unrolled by two: matrix and leaf codes of entry i + 1 are requested before entry i is processed.
  // Programs are padded to an even entry count and followed by two no-op entries.
  // (schedule words two entries ahead: their scalar load has returned by the time they address the matrix / code loads)
  int4 opA = fetch_op(0), opB = fetch_op(1);
  double PA[12], PB[12];
  int cA[2][NP], cB[2][NP];
  load_P(opA, PA);
  codes_of(opA, cA);
  for (int oi = 0; oi < a.n_ops; oi += 2) {
    const int4 opC = fetch_op(oi + 2);
    load_P(opB, PB);
    codes_of(opB, cB);
    entry(opA, PA, cA);
    const int4 opD = fetch_op(oi + 3);
    load_P(opC, PA);
    codes_of(opC, cA);
    entry(opB, PB, cB);
    opA = opC;
    opB = opD;
  }
  // root: L_s = sum_k root[s][k] pi[k]; this workgroup's share of sum_s f_s log L_s and of the integer scaler sum
  double term = 0.;
  long long tc = 0;
  int fl = 0;
  if (a.n_ops > 0) {
#pragma unroll
    for (int q = 0; q < NP; q++) {
      const int s = s0 + 256 * q;
      double Lk = b[q][0] * a.pi[0];
      Lk = fma(b[q][1], a.pi[1], Lk);
      Lk = fma(b[q][2], a.pi[2], Lk);
      Lk = fma(b[q][3], a.pi[3], Lk);
      a.site_lik[s] = Lk;
      a.site_cnt[s] = bcnt[q];
      const double f = a.freq[s];
      if (f != 0.) {
        if (Lk != Lk || isinf(Lk)) fl |= 2;
        else if (Lk <= 0.) fl |= 1;
        else {
          term += log(Lk) * f;
          tc += (long long)bcnt[q] * (long long)f;
        }
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {  // fixed-order butterfly inside the wave, then the four waves in order
    term += __shfl_xor(term, off);
    tc += __shfl_xor(tc, off);
    fl |= __shfl_xor(fl, off);
  }
  __syncthreads();  // (the parking area is dead: its first bytes carry the four waves' partial sums)
  double *rs = park;
  long long *rc = reinterpret_cast<long long *>(park + 4);
  int *rf = reinterpret_cast<int *>(park + 8);
  if ((tid & 63) == 0) {
    rs[tid >> 6] = term;
    rc[tid >> 6] = tc;
    rf[tid >> 6] = fl;
  }
  __syncthreads();
  if constexpr (LP) {
    if (a.red_out && a.n_ops > 0) {  // fused final combine (small shards): the last workgroup to arrive sums all partials
      if (tid < 64) {                // (wave 0, all lanes: the same protocol as the codon kernels' publish_partial)
        if (tid == 0) {
          __hip_atomic_store(a.wg_sum + blockIdx.x, (rs[0] + rs[1]) + (rs[2] + rs[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(a.wg_cnt + blockIdx.x, (rc[0] + rc[1]) + (rc[2] + rc[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(a.wg_flag + blockIdx.x, rf[0] | rf[1] | rf[2] | rf[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int old = 0;
        if (tid == 0) old = __hip_atomic_fetch_add(a.red_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        old = __builtin_amdgcn_readfirstlane(old);
        asm volatile("" ::: "memory");
        if (old + 1 == (int)gridDim.x) {
          if (tid == 0) __hip_atomic_store(a.red_done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next launch
          combine_partials(a.wg_sum, a.wg_cnt, a.wg_flag, (int)gridDim.x, a.red_out, a.red_rec, a.red_status, a.red_seq, tid);
        }
      }
      return;
    }
  }
  if (tid == 0 && a.n_ops > 0) {
    a.wg_sum[blockIdx.x] = (rs[0] + rs[1]) + (rs[2] + rs[3]);
    a.wg_cnt[blockIdx.x] = (rc[0] + rc[1]) + (rc[2] + rc[3]);
    a.wg_flag[blockIdx.x] = rf[0] | rf[1] | rf[2] | rf[3];
  }
}

// ---------------------------------------------------------------------------------------------
// logL = sum_s f_s log L_s  -  64 ln2 * sum_s f_s c_s      (tree_evaluator.cpp:4114-4128 Kahan sum,
// likefunc.cpp:11123 scaler correction).  One workgroup; per-thread Kahan accumulation over a
// fixed stride, then a fixed-order tree: deterministic run to run.  The scaler part is summed in
// exact integer arithmetic like the reference's `long overallScaler`.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void site_reduce_kernel(const double *__restrict__ site_lik,
                                                           const int32_t *__restrict__ site_cnt,
                                                           const double *__restrict__ freq, int S_pad,
                                                           int floor_log, double *__restrict__ out,
                                                           double *__restrict__ out_cnt,
                                                           const int *__restrict__ status, double seq) {
  __shared__ double ssum[1024];
  __shared__ double scomp[1024];
  __shared__ long long scnt[1024];
  __shared__ int sflags;
  const int tid = threadIdx.x;
  if (tid == 0) sflags = 0;
  __syncthreads();
  double sum = 0., comp = 0.;
  long long c = 0;
  int fl = 0;  // 1: a pattern with zero likelihood (-> -inf, tree_evaluator.cpp:4094-4112), 2: NaN
  for (int s = tid; s < S_pad; s += 1024) {
    const double f = freq[s];
    if (f == 0.) continue;
    const double L = site_lik[s];
    if (L != L) { fl |= 2; continue; }
    if (L <= 0. || isinf(L)) {
      if (floor_log && L <= 0.) {  // myLog floor, no scaler (likefunc.cpp:644-661) in category mode
        const double y0 = -1000000. * f - comp;
        const double t0 = sum + y0;
        comp = (t0 - sum) - y0;
        sum = t0;
      } else {
        fl |= (L <= 0.) ? 1 : 2;
      }
      continue;
    }
    const double y = log(L) * f - comp;  // Kahan
    const double t = sum + y;
    comp = (t - sum) - y;
    sum = t;
    c += (long long)site_cnt[s] * (long long)f;
  }
  if (fl) atomicOr(&sflags, fl);
  ssum[tid] = sum;
  scomp[tid] = comp;
  scnt[tid] = c;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if (tid < off) {
      const double a0 = ssum[tid], b0 = ssum[tid + off];
      const double t = a0 + b0;
      const double e = (fabs(a0) >= fabs(b0)) ? (a0 - t) + b0 : (b0 - t) + a0;  // a0 + b0 = t + e exactly
      ssum[tid] = t;
      scomp[tid] = scomp[tid] + scomp[tid + off] - e;
      scnt[tid] += scnt[tid + off];
    }
    __syncthreads();
  }
  if (tid == 0) {
    double r = (ssum[0] - scomp[0]) - kLogScaler * (double)scnt[0];
    if (sflags & 2) r = NAN;
    else if (sflags & 1) r = -INFINITY;
    out[0] = r;
    out_cnt[0] = (double)scnt[0];
    out_cnt[1] = status ? (double)*status : 0.;  // [log-L, scaler sum, expm status]: one host-visible record
    if (seq != 0.) {  // host spins on this word instead of waiting for the stream (record complete before it)
      __threadfence_system();
      reinterpret_cast<volatile double *>(out_cnt)[2] = seq;
    }
  }
}

// Final combine of the per-workgroup partial sums: Neumaier-compensated, fixed order (the device
// analogue of ComputeBlock's combine of its thread blocks, likefunc.cpp:11046-11123).
__global__ __launch_bounds__(256) void wg_reduce_kernel(const double *__restrict__ wg_sum,
                                                        const long long *__restrict__ wg_cnt,
                                                        const int *__restrict__ wg_flag, int n,
                                                        double *__restrict__ out, double *__restrict__ out_cnt,
                                                        const int *__restrict__ status, double seq) {
  __shared__ double ssum[256];
  __shared__ double scomp[256];
  __shared__ long long scnt[256];
  __shared__ int sflags;
  const int tid = threadIdx.x;
  if (tid == 0) sflags = 0;
  __syncthreads();
  double sum = 0., comp = 0.;
  long long c = 0;
  int fl = 0;
  for (int k = tid; k < n; k += 256) {
    const double y = wg_sum[k] - comp;
    const double t = sum + y;
    comp = (t - sum) - y;
    sum = t;
    c += wg_cnt[k];
    fl |= wg_flag[k];
  }
  if (fl) atomicOr(&sflags, fl);
  ssum[tid] = sum;
  scomp[tid] = comp;
  scnt[tid] = c;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) {
      const double a0 = ssum[tid], b0 = ssum[tid + off];
      const double t = a0 + b0;
      const double e = (fabs(a0) >= fabs(b0)) ? (a0 - t) + b0 : (b0 - t) + a0;
      ssum[tid] = t;
      scomp[tid] = scomp[tid] + scomp[tid + off] - e;
      scnt[tid] += scnt[tid + off];
    }
    __syncthreads();
  }
  if (tid == 0) {
    double r = (ssum[0] - scomp[0]) - kLogScaler * (double)scnt[0];
    if (sflags & 2) r = NAN;
    else if (sflags & 1) r = -INFINITY;
    out[0] = r;
    out_cnt[0] = (double)scnt[0];
    out_cnt[1] = status ? (double)*status : 0.;  // [log-L, scaler sum, expm status]: one host-visible record
    if (seq != 0.) {  // host spins on this word instead of waiting for the stream (record complete before it)
      __threadfence_system();
      reinterpret_cast<volatile double *>(out_cnt)[2] = seq;
    }
  }
}

// The same combine by ONE wave (r04): combine_partials — what the fused launches' last arriver runs — in a launch of its own.
// All partials of a 512-entry block are in flight together (the 256-thread kernel above walks them in dependent round trips
// and then an 8-step LDS tree with barriers); same record, same flags, a different (fixed) order of summation.
__global__ __launch_bounds__(64) void wave_reduce_kernel(double *wg_sum, long long *wg_cnt, int *wg_flag, int n, double *out,
                                                         double *out_cnt, const int *status, double seq) {
  combine_partials(wg_sum, wg_cnt, wg_flag, n, out, out_cnt, status, seq, (int)threadIdx.x);
}

// Thousands of partials (128 x 100 k: 6 250 tiles; 10^6 nucleotide sites: 3 907 workgroups): NWV waves take a contiguous range each —
// the one-wave kernel's sum over that range — and wave 0 adds the NWV results in wave order with the same compensated step.  A
// different (fixed) order of summation than one wave's: used from 2 048 partials upwards, where no fused launch exists to disagree with
// (one wave: 10.5 us for 6 250 partials, 8.6 for 3 907).
template <int NWV>
__global__ __launch_bounds__(64 * NWV) void multi_wave_reduce_kernel(double *wg_sum, long long *wg_cnt, int *wg_flag, int n, double *out,
                                                                     double *out_cnt, const int *status, double seq) {
  __shared__ double ssum[NWV], scomp[NWV];
  __shared__ long long sc[NWV];
  __shared__ int sfl[NWV];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int chunk = ((n + NWV - 1) / NWV + 511) / 512 * 512;
  double sum, comp;
  long long c;
  int fl;
  combine_range(wg_sum, wg_cnt, wg_flag, n, min(n, w * chunk), min(n, (w + 1) * chunk), lane, sum, comp, c, fl);
  if (lane == 0) ssum[w] = sum, scomp[w] = comp, sc[w] = c, sfl[w] = fl;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < NWV; k++) {
      const double b0 = ssum[k], t = sum + b0;
      const double e = (fabs(sum) >= fabs(b0)) ? (sum - t) + b0 : (b0 - t) + sum;  // sum + b0 = t + e exactly
      comp = comp + scomp[k] - e;
      sum = t;
      c += sc[k];
      fl |= sfl[k];
    }
    combine_publish(sum, comp, c, fl, out, out_cnt, status, seq);
  }
}

// Category mixing on the device: PopulateConditionalProbabilities weighted-sum mode
// (likefunc2.cpp:820-853): buf[s] = sum_c w_c L_c[s] 2^(-64 (c_c[s] - min_c c_c[s])).
__global__ void mix_categories_kernel(const double *__restrict__ site_lik, const int32_t *__restrict__ site_cnt,
                                      const double *__restrict__ w, int C, int S_pad, double *__restrict__ mixed,
                                      int32_t *__restrict__ mixed_cnt) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S_pad) return;
  double buf = 0.;
  int sc = 0;
  for (int c = 0; c < C; c++) {
    const double v = site_lik[(size_t)c * S_pad + s] * w[c];
    const int scv = site_cnt[(size_t)c * S_pad + s];
    if (c == 0) {
      buf = v;
      sc = scv;
    } else if (scv < sc) {
      buf = v + buf * exp(-kLogScaler * (double)(sc - scv));
      sc = scv;
    } else if (scv > sc) {
      buf += v * exp(-kLogScaler * (double)(scv - sc));
    } else {
      buf += v;
    }
  }
  mixed[s] = buf;
  mixed_cnt[s] = sc;
}

// fragment layout -> reference iNodeCache layout [(node*S + pattern)*D + state]
__global__ void unpack_partials_kernel(const double *__restrict__ partials, int I, int ntiles, int NW, int D, int S,
                                       double *__restrict__ out) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)I * S * D;
  if (idx >= total) return;
  const int state = idx % D;
  const size_t rest = idx / D;
  const int pat = rest % S;
  const int node = rest / S;
  const int tile = pat >> 4, sl = pat & 15;
  const int kk = state >> 2, lane = (state & 3) * 16 + sl;
  const int TILE = NW * 4 * 64;
  const int within = frag_index(kk, lane);
  out[idx] = partials[((size_t)node * ntiles + tile) * TILE + within];
}

template <int NW, bool CLDS>
void launch_prune_T(const PruneArgs &a, hipStream_t stream) {
  // Chain grids (tiles, classes, sources): workgroup b runs on XCD b mod 8 and the grid is linearised x-fastest, so with the
  // tile dimension padded to a multiple of 8 (the surplus workgroups retire at once) every chain of a tile — and so every
  // join of its trunk — sits on ONE XCD: deposits and arrival counters are then L2 hits instead of trips to memory.
  // (HYPHY_HIP_XCD_PAD=0: the bare tile count.)
  const char *pad_env = getenv("HYPHY_HIP_XCD_PAD");  // (read per launch: the tests run both placements in one process)
  const bool xcd_pad = !(pad_env && atoi(pad_env) == 0);
  const int gx_chain = (a.chain && a.T == 1 && xcd_pad && !a.timeline) ? ((a.ntiles + 7) & ~7) : a.ntiles / a.T;
  const dim3 grid(a.chain ? gx_chain : a.ntiles / a.T, a.n_cat > 0 ? a.n_cat : 1, a.n_prog > 0 ? a.n_prog : 1), block(64 * NW);
  const size_t lds = CLDS ? (size_t)a.L * a.T * 16 * sizeof(int16_t) : 0;
  const dim3 gridw = a.chain ? dim3(gx_chain, a.n_cat > 0 ? a.n_cat : 1, a.n_prog > 0 ? a.n_prog : 1)   // chains: source-major
                             : dim3(a.n_prog > 0 ? a.n_prog : 1, a.n_cat > 0 ? a.n_cat : 1, a.ntiles);  // fragments: tile-major
  if (a.variant == 1 && a.T == 1) {  // wave-per-tile kernel: one wave per workgroup
    const dim3 block1(64);
    const size_t lds1 = CLDS ? (size_t)a.L * 16 * sizeof(int16_t) : 0;
    if (a.leaf_tab && a.timeline && NW == 4 && CLDS) {  // tracing build of the trunk (HYPHY_HIP_TIMELINE with HYPHY_HIP_REPEATS=1)
      if (a.n_slots <= 3) hipLaunchKernelGGL((prune_wave_kernel<4, 1, true, true, false, HYPHY_OCC3, true, false, false, true>), gridw, block1, lds1, stream, a.ops, a.prog, a.jn, a);
      else hipLaunchKernelGGL((prune_wave_kernel<4, 2, true, true, false, HYPHY_OCC3, true, false, false, true>), gridw, block1, lds1, stream, a.ops, a.prog, a.jn, a);
      return;
    }
    if (a.leaf_tab && NW == 4 && CLDS && (a.wave_variant == 2 || a.wave_variant == 3) && a.n_slots <= 2) {  // three waves per SIMD, see below
      if (a.wave_variant == 2) hipLaunchKernelGGL((prune_wave_kernel<4, 0, true, false, true, 3, false, false, false, true>), gridw, block1, lds1, stream, a.ops, a.prog, a.jn, a);
      else hipLaunchKernelGGL((prune_wave_kernel<4, 0, true, false, true, 3, true, false, false, true>), gridw, block1, lds1, stream, a.ops, a.prog, a.jn, a);
      return;
    }
    if (a.leaf_tab) {  // the trunk of a class-compressed partition
      if (a.n_slots <= 2) hipLaunchKernelGGL((prune_wave_kernel<NW, 0, CLDS, false, false, HYPHY_OCC3, true, false, false, true>), gridw, block1, lds1, stream, a.ops, a.prog, a.jn, a);
      else if (a.n_slots == 3) hipLaunchKernelGGL((prune_wave_kernel<NW, 1, CLDS, false, false, HYPHY_OCC3, true, false, false, true>), gridw, block1, lds1, stream, a.ops, a.prog, a.jn, a);
      else hipLaunchKernelGGL((prune_wave_kernel<NW, 2, CLDS, false, false, HYPHY_OCC3, true, false, false, true>), gridw, block1, lds1, stream, a.ops, a.prog, a.jn, a);
      return;
    }
    if (a.timeline && NW == 4 && CLDS) {  // tracing build (HYPHY_HIP_TIMELINE)
      if (a.n_slots <= 3) hipLaunchKernelGGL((prune_wave_kernel<4, 1, true, true>), gridw, block1, lds1, stream, a.ops, a.prog, a.jn, a);
      else hipLaunchKernelGGL((prune_wave_kernel<4, 2, true, true>), gridw, block1, lds1, stream, a.ops, a.prog, a.jn, a);
      return;
    }
    // three waves per SIMD (the tuner's second stage): finalised node in LDS instead of 32 registers, no parking slot; 2 without,
    // 3 with the deposit prefetch
    if (NW == 4 && CLDS && a.wave_variant == 2 && a.n_slots <= 2) {
      hipLaunchKernelGGL((prune_wave_kernel<4, 0, true, false, true, 3, false>), gridw, block1, lds1, stream, a.ops, a.prog, a.jn, a);
      return;
    }
    if (NW == 4 && CLDS && a.wave_variant == 3 && a.n_slots <= 2) {
      hipLaunchKernelGGL((prune_wave_kernel<4, 0, true, false, true, 3, true>), gridw, block1, lds1, stream, a.ops, a.prog, a.jn, a);
      return;
    }
    // production: two waves per SIMD, 0 / 1 / 2 parking slots in LDS (the schedule was compiled for a.n_slots - 2 of them)
    if (a.n_slots <= 2) hipLaunchKernelGGL((prune_wave_kernel<NW, 0, CLDS>), gridw, block1, lds1, stream, a.ops, a.prog, a.jn, a);
    else if (a.n_slots == 3) hipLaunchKernelGGL((prune_wave_kernel<NW, 1, CLDS>), gridw, block1, lds1, stream, a.ops, a.prog, a.jn, a);
    else hipLaunchKernelGGL((prune_wave_kernel<NW, 2, CLDS>), gridw, block1, lds1, stream, a.ops, a.prog, a.jn, a);
    return;
  }
  if (a.leaf_tab && a.variant != 1) {  // the trunk of a class-compressed partition under the row-split workgroup kernel (r06)
    if constexpr (NW == 4 && CLDS) {
      if (a.T == 1 && a.chain) {
        hipLaunchKernelGGL((prune_mfma_kernel<4, 1, true, false, 0, true, false, true>), grid, block, lds, stream, a.ops, a);
        return;
      }
      if (a.T == 1) {
        hipLaunchKernelGGL((prune_mfma_kernel<4, 1, true, false, 0, false, false, true>), grid, block, lds, stream, a.ops, a);
        return;
      }
    }
    return;  // (no other form of this mode exists: the tuner only offers the two above)
  }
  if (a.variant == 2 && a.chain && a.T == 1) {  // row-split workgroups on a chain schedule: grid = (tiles, classes, sources)
    if constexpr (NW == 4 && CLDS) {
      if (a.red_out) {
        hipLaunchKernelGGL((prune_mfma_kernel<4, 1, true, false, 0, true, true>), grid, block, lds, stream, a.ops, a);
        return;
      }
    }
    hipLaunchKernelGGL((prune_mfma_kernel<NW, 1, CLDS, false, 0, true>), grid, block, lds, stream, a.ops, a);
    return;
  }
  if (a.timeline) {  // tracing build of the kernel (HYPHY_HIP_TIMELINE), T = 1 only
    hipLaunchKernelGGL((prune_mfma_kernel<NW, 1, CLDS, true>), grid, block, lds, stream, a.ops, a);
    return;
  }
  if (NW == 4 && CLDS && a.T == 1 && a.ablate) {  // diagnostic ablation builds (results invalid)
#define ABL_CASE(v)                                                                                     \
  case v:                                                                                               \
    hipLaunchKernelGGL((prune_mfma_kernel<4, 1, true, false, v>), grid, block, lds, stream, a.ops, a); \
    return;
    switch (a.ablate) {
      ABL_CASE(1) ABL_CASE(2) ABL_CASE(4) ABL_CASE(16) ABL_CASE(62) ABL_CASE(63) ABL_CASE(127)
      default: break;
    }
#undef ABL_CASE
  }
  switch (a.T) {
    case 1:
      if constexpr (NW == 4 && CLDS) {
        if (a.red_out) {
          hipLaunchKernelGGL((prune_mfma_kernel<4, 1, true, false, 0, false, true>), grid, block, lds, stream, a.ops, a);
          break;
        }
      }
      hipLaunchKernelGGL((prune_mfma_kernel<NW, 1, CLDS, false>), grid, block, lds, stream, a.ops, a);
      break;
    case 2:
      hipLaunchKernelGGL((prune_mfma_kernel<NW, 2, CLDS, false>), grid, block, lds, stream, a.ops, a);
      break;
    case 3:
      hipLaunchKernelGGL((prune_mfma_kernel<NW, 3, CLDS, false>), grid, block, lds, stream, a.ops, a);
      break;
    default:
      hipLaunchKernelGGL((prune_mfma_kernel<NW, 4, CLDS, false>), grid, block, lds, stream, a.ops, a);
      break;
  }
}

template <int NW>
void launch_prune_NW(const PruneArgs &a, hipStream_t stream) {
  if (a.codes_in_lds) launch_prune_T<NW, true>(a, stream);
  else launch_prune_T<NW, false>(a, stream);
}

}  // namespace

void launch_prune_mfma(const PruneArgs &a, hipStream_t stream) {
  if (a.n_ops <= 0) return;
  switch (a.NW) {
    case 1:
      launch_prune_NW<1>(a, stream);
      break;
    case 2:
      launch_prune_NW<2>(a, stream);
      break;
    case 3:
      launch_prune_NW<3>(a, stream);
      break;
    default:
      launch_prune_NW<4>(a, stream);
      break;
  }
}

// prune_nuc2_kernel: NP patterns per thread while the leaf matrices (128 bytes each) and the parking slots fit LDS
static int nuc_forced() {
  static const int forced = getenv("HYPHY_HIP_NUC") ? atoi(getenv("HYPHY_HIP_NUC")) : -1;  // 0: r02 kernel, 1 / 2: patterns per thread
  return forced;
}
bool prune_nuc_takes_leaf_pairs(int L) { return nuc_forced() != 0 && L <= 256; }
static int nuc2_np(const NucArgs &a) {
  const int forced = nuc_forced();
  if (!prune_nuc_takes_leaf_pairs(a.L) || !a.PT) return 0;
  if (forced == 1 || a.S_pad % 512 != 0) return 1;
  if (forced == 2) return 2;
  return 1;  // (measured: one pattern per thread and four workgroups per CU beat two patterns and two workgroups at every size)
}
static size_t nuc2_lds(const NucArgs &a, int np, bool lp = false) {
  return (size_t)(lp ? a.L + a.root_inode : a.L) * 16 * sizeof(double) +
         (size_t)kNucParkSlots * np * 256 * (4 * sizeof(double) + sizeof(int)) + (lp ? (size_t)(a.n_ops + 4) * sizeof(int4) : 0);
}

static bool nuc_uses_lp(const NucArgs &a, int np, bool folded, int dev) {
  static int cus_n[64];
  if (!cus_n[dev]) {
    hipDeviceProp_t pr;
    cus_n[dev] = (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
  }
  const char *lp_env = getenv("HYPHY_HIP_NUC_LP");
  (void)folded;
  return np == 1 && (lp_env ? atoi(lp_env) != 0 : a.S_pad / 256 <= 2 * cus_n[dev]) && nuc2_lds(a, 1, true) <= (size_t)(112 * 1024);
}

// true when launch_prune_nuc can take the evaluation's matrix exponentials along (ex != nullptr): the r03 kernel on a shard of
// at most two workgroups per CU
bool prune_nuc_folds_expm(int L, int S_pad, int n_ops) {
  // (late r03: on by default — with the Paterson-Stockmeyer 4 x 4 exponential and the coefficients in the kernel-argument block the
  //  folded launch is 41 us per step at 50 000 sites against 42.5-43 with a launch of its own; HYPHY_HIP_NUC_FOLD=0 turns it off)
  const char *fe = getenv("HYPHY_HIP_NUC_FOLD");
  const bool on = !(fe && atoi(fe) == 0);
  return on && n_ops > 0 && prune_nuc_takes_leaf_pairs(L) && nuc_forced() != 2 && S_pad % 256 == 0 && S_pad / 256 <= 512;
}

// true when launch_prune_nuc will pick the LDS-schedule instantiation (LP) — the one that can also carry the fused final
// combine (NucArgs::red_out) — for this launch
bool prune_nuc_fuses_reduce(const NucArgs &a, bool folded) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  return a.n_ops > 0 && nuc_uses_lp(a, nuc2_np(a), folded, dev);
}

void launch_prune_nuc(const NucArgs &a, hipStream_t stream, const ExpmArgs *ex) {
  if (a.n_ops <= 0) return;
  const bool pin = a.pin_leaf >= 0 || a.pin_inode >= 0;
  const int np = nuc2_np(a);
  if (a.leaf_tab) {  // the trunk of a class-compressed partition (no pinned states in that mode)
    hipLaunchKernelGGL((prune_nuc_kernel<false, true>), dim3((a.S_pad + 255) / 256), dim3(256), 0, stream, a.ops, a.P, a);
    return;
  }
  if (np == 0) {
    if (pin) hipLaunchKernelGGL(prune_nuc_kernel<true>, dim3((a.S_pad + 255) / 256), dim3(256), 0, stream, a.ops, a.P, a);
    else hipLaunchKernelGGL(prune_nuc_kernel<false>, dim3((a.S_pad + 255) / 256), dim3(256), 0, stream, a.ops, a.P, a);
    return;
  }
  static bool attr_done[64];
  constexpr int cap_bytes = 112 * 1024;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!attr_done[dev]) {  // (2 patterns per thread: 74 KiB of parking + up to 32 KiB of leaf matrices)
    const int cap = cap_bytes;
    hipFuncSetAttribute(reinterpret_cast<const void *>(prune_nuc2_kernel<1, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    hipFuncSetAttribute(reinterpret_cast<const void *>(prune_nuc2_kernel<1, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    hipFuncSetAttribute(reinterpret_cast<const void *>(prune_nuc2_kernel<1, false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    hipFuncSetAttribute(reinterpret_cast<const void *>(prune_nuc2_kernel<1, true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    hipFuncSetAttribute(reinterpret_cast<const void *>(prune_nuc2_kernel<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    hipFuncSetAttribute(reinterpret_cast<const void *>(prune_nuc2_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    hipFuncSetAttribute(reinterpret_cast<const void *>(prune_nuc2_kernel<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    hipFuncSetAttribute(reinterpret_cast<const void *>(prune_nuc2_kernel<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    hipFuncSetAttribute(reinterpret_cast<const void *>(prune_nuc2_kernel<1, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    hipFuncSetAttribute(reinterpret_cast<const void *>(prune_nuc2_kernel<1, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    attr_done[dev] = true;
  }
  const dim3 grid(a.S_pad / (256 * np)), block(256);
  const size_t lds = nuc2_lds(a, np);
  ExpmArgs none;
  none.n = 0;
  CoefInline ci;          // (contents only matter to the folded launches, which fill it below; copied into the kernel arguments at launch)
  ExpmArgs exb;
  if (ex && ex->n > 0) {
    exb = *ex;
    fill_coef_inline(exb, ci);
  }
  // schedule words + every branch's matrix from LDS (LP): shards of at most two workgroups per CU
  const bool lp = nuc_uses_lp(a, np, ex && ex->n > 0, dev);
  if (lp) {
    const size_t ldl = nuc2_lds(a, 1, true);
    if (ex && ex->n > 0) {  // (with this evaluation's exponentials folded in)
      if (pin) hipLaunchKernelGGL((prune_nuc2_kernel<1, true, true, true>), grid, block, ldl, stream, a.ops, a.PT, a, exb, ci);
      else hipLaunchKernelGGL((prune_nuc2_kernel<1, false, true, true>), grid, block, ldl, stream, a.ops, a.PT, a, exb, ci);
      return;
    }
    if (pin) hipLaunchKernelGGL((prune_nuc2_kernel<1, true, false, true>), grid, block, ldl, stream, a.ops, a.PT, a, none, CoefNone{0});
    else hipLaunchKernelGGL((prune_nuc2_kernel<1, false, false, true>), grid, block, ldl, stream, a.ops, a.PT, a, none, CoefNone{0});
    return;
  }
  if (ex && ex->n > 0 && np == 1) {
    if (pin) hipLaunchKernelGGL((prune_nuc2_kernel<1, true, true>), grid, block, lds, stream, a.ops, a.PT, a, exb, ci);
    else hipLaunchKernelGGL((prune_nuc2_kernel<1, false, true>), grid, block, lds, stream, a.ops, a.PT, a, exb, ci);
    return;
  }
  if (np == 2) {
    if (pin) hipLaunchKernelGGL((prune_nuc2_kernel<2, true>), grid, block, lds, stream, a.ops, a.PT, a, none, CoefNone{0});
    else hipLaunchKernelGGL((prune_nuc2_kernel<2, false>), grid, block, lds, stream, a.ops, a.PT, a, none, CoefNone{0});
  } else {
    if (pin) hipLaunchKernelGGL((prune_nuc2_kernel<1, true>), grid, block, lds, stream, a.ops, a.PT, a, none, CoefNone{0});
    else hipLaunchKernelGGL((prune_nuc2_kernel<1, false>), grid, block, lds, stream, a.ops, a.PT, a, none, CoefNone{0});
  }
}

// Per-pattern results in the CALLER's order straight into host-mapped pinned memory (posted writes, 512 contiguous bytes per wave):
// a host that mixes rate classes itself asks for them once per class and evaluation, and two SDMA copies + a stream wait + a
// scatter loop on the host cost it more than the evaluation's own host share.  inv[i] = device pattern of caller pattern i.
__global__ __launch_bounds__(256) void site_export_kernel(const double *__restrict__ lik, const int32_t *__restrict__ cnt,
                                                          const int32_t *__restrict__ inv, int S, double *out_lik, long long *out_cnt) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= S) return;
  const int j = inv ? inv[i] : i;
  if (out_lik) out_lik[i] = lik[j];
  if (out_cnt) out_cnt[i] = (long long)cnt[j];
  //  the host reads these rows as soon as it sees the sequence word the NEXT kernel publishes, without waiting for the
  // stream: every row is pushed out at system scope by the thread that wrote it, so the record's own system-scope fence + store
  // (behind the kernel boundary) can only be seen after them — not left to the ordering of posted writes of two different kernels
  __threadfence_system();
}
void launch_site_export(const double *lik, const int32_t *cnt, const int32_t *inv, int S, double *out_lik, long long *out_cnt,
                        hipStream_t stream) {
  hipLaunchKernelGGL(site_export_kernel, dim3((S + 255) / 256), dim3(256), 0, stream, lik, cnt, inv, S, out_lik, out_cnt);
}

void launch_site_reduce(const double *site_lik, const int32_t *site_cnt, const double *freq, int S_pad, int floor_log,
                        double *out, double *out_cnt, const int *status, hipStream_t stream, double seq) {
  hipLaunchKernelGGL(site_reduce_kernel, dim3(1), dim3(1024), 0, stream, site_lik, site_cnt, freq, S_pad, floor_log,
                     out, out_cnt, status, seq);
}

void launch_wg_reduce(const double *wg_sum, const long long *wg_cnt, const int *wg_flag, int n, double *out_logl,
                      double *out_cnt, const int *status, hipStream_t stream, double seq) {
  static const bool block_form = getenv("HYPHY_HIP_REDUCE") && !strcmp(getenv("HYPHY_HIP_REDUCE"), "block");
  static const bool multi_off = getenv("HYPHY_HIP_REDUCE") && !strcmp(getenv("HYPHY_HIP_REDUCE"), "wave");
  if (block_form)
    hipLaunchKernelGGL(wg_reduce_kernel, dim3(1), dim3(256), 0, stream, wg_sum, wg_cnt, wg_flag, n, out_logl, out_cnt, status, seq);
  else if (n >= 2048 && !multi_off)
    hipLaunchKernelGGL(multi_wave_reduce_kernel<8>, dim3(1), dim3(512), 0, stream, const_cast<double *>(wg_sum), const_cast<long long *>(wg_cnt),
                       const_cast<int *>(wg_flag), n, out_logl, out_cnt, status, seq);
  else
    hipLaunchKernelGGL(wave_reduce_kernel, dim3(1), dim3(64), 0, stream, const_cast<double *>(wg_sum), const_cast<long long *>(wg_cnt),
                       const_cast<int *>(wg_flag), n, out_logl, out_cnt, status, seq);
}

int prune_mfma_grid(const PruneArgs &a) { return a.ntiles / a.T; }

// true when launch_prune_mfma has an instantiation with the fused final combine for this launch form (PruneArgs::red_out):
// 49-64 states with the leaf codes in LDS, one tile per workgroup, the row-split kernels (workgroup per tile, team)
bool prune_fuses_reduce(const PruneArgs &a) {
  // (not the wave-per-tile kernel: at 231 VGPRs the combine's registers push 15 spill instructions into its main loop —
  //  what the fusion saves on a small shard, the spills cost)
  if (a.NW != 4 || !a.codes_in_lds || a.T != 1 || a.timeline || a.ablate) return false;
  if (a.leaf_tab) return false;  // (the trunk of a class-compressed partition: its instantiations carry no fused combine)
  return a.variant == 0 || (a.variant == 2 && a.chain);
}

void launch_transpose_frag(const double *src_image, double *dst_image, const double *row_scale, int NW, hipStream_t stream) {
  hipLaunchKernelGGL(transpose_frag_kernel, dim3(1), dim3(256), 0, stream, src_image, dst_image, row_scale, NW);
}
void launch_bc_eval(const BcArgs &a, hipStream_t stream) {
  const dim3 grid(a.ntiles), block(64);
  switch (a.NW) {
    case 1: hipLaunchKernelGGL(bc_eval_kernel<1>, grid, block, 0, stream, a); break;
    case 2: hipLaunchKernelGGL(bc_eval_kernel<2>, grid, block, 0, stream, a); break;
    case 3: hipLaunchKernelGGL(bc_eval_kernel<3>, grid, block, 0, stream, a); break;
    default: hipLaunchKernelGGL(bc_eval_kernel<4>, grid, block, 0, stream, a); break;
  }
}
int prune_nuc_grid(const NucArgs &a) {
  const int np = nuc2_np(a);
  return np ? a.S_pad / (256 * np) : (a.S_pad + 255) / 256;
}

void launch_mix_categories(const double *site_lik, const int32_t *site_cnt, const double *weights_dev, int C,
                           int S_pad, double *mixed_lik, int32_t *mixed_cnt, hipStream_t stream) {
  hipLaunchKernelGGL(mix_categories_kernel, dim3((S_pad + 255) / 256), dim3(256), 0, stream, site_lik, site_cnt,
                     weights_dev, C, S_pad, mixed_lik, mixed_cnt);
}

void launch_unpack_partials_mfma(const double *partials, int I, int ntiles, int NW, int D, int S, double *out,
                                 hipStream_t stream) {
  const size_t total = (size_t)I * S * D;
  hipLaunchKernelGGL(unpack_partials_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, partials, I,
                     ntiles, NW, D, S, out);
}

}  // namespace hyhip
