// Internal host-side definitions shared by the host translation units of libhyphy_hip.so:
//   api.hip       the C-ABI (include/hyphy_hip.h): partition life cycle, evaluation entry points, kernel sequencing
//   schedule.hip  the schedule compiler (post-order programs, chain / level-peeled cuts, re-rooting, pattern order)
//   tuner.hip     the run-time schedule tuner and the launch of the current schedule
//   comm.hip      RCCL (loaded on first use) and the all-reduce entry points
// Host-side bookkeeping only — all arithmetic of the hot path happens in expm.hip / prune.hip / sitefit.hip.
#pragma once
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <set>
#include <string>
#include <vector>

#include "../../include/hyphy_hip.h"
#include "common.h"

namespace hyhip {

extern thread_local std::string g_last_error;
int fail(const std::string &msg);

// Recycling allocator (pool.hip): analyses that create one likelihood function after another on the same tree (FEL: one per
// site) allocate the same ~45 device / pinned-host blocks and a stream over and over; hipMalloc / hipHostMalloc / hipFree /
// stream creation cost that life cycle more than its 50 evaluations.  Freed blocks go to a free list keyed by (device, size)
// and are handed out again on an exact size match.  pool_free does NOT synchronise (hipFree does, implicitly): callers that
// free while work may be in flight use pool_free_sync.  HYPHY_HIP_POOL_MB (default 1024; 0: off) caps the cached bytes per
// kind; blocks above 64 MiB are never cached.
hipError_t pool_malloc(void **p, size_t bytes);
void pool_free(void *p);
void pool_free_sync(void *p);
hipError_t pool_host_malloc(void **p, size_t bytes);
void pool_host_free(void *p);
hipError_t pool_stream_get(hipStream_t *s);   // a non-blocking stream of the current device
void pool_stream_put(hipStream_t s);          // (idle: the caller has synchronised it)

// ---- RCCL, loaded on first use (librccl.so is part of ROCm; a host that never all-reduces does not need it) -------------
struct Rccl {
  void *lib = nullptr;
  int (*GetUniqueId)(void *) = nullptr;
  int (*CommInitAll)(void **comms, int ndev, const int *devlist) = nullptr;
  int (*CommDestroy)(void *comm) = nullptr;
  int (*AllReduce)(const void *send, void *recv, size_t count, int dtype, int op, void *comm, hipStream_t stream) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};
struct RcclUniqueId {
  char internal[128];  // NCCL_UNIQUE_ID_BYTES
};
typedef int (*rccl_init_rank_fn)(void **comm, int nranks, RcclUniqueId id, int rank);
extern Rccl g_rccl;
extern rccl_init_rank_fn g_rccl_init_rank;
constexpr int kNcclDouble = 8, kNcclSum = 0;
int rccl_load();
#define RCCLCHK(expr)                                                                                         \
  do {                                                                                                        \
    int r_ = (expr);                                                                                          \
    if (r_ != 0) return fail(std::string(#expr) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "RCCL error")); \
  } while (0)

struct Trace {
  bool on;
  double t0;
  const char *what;
  static double now() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
  }
  explicit Trace(const char *w) : on(getenv("HYPHY_HIP_TRACE") != nullptr), t0(0), what(w) {
    if (on) t0 = now();
  }
  void lap(const char *stage) {
    if (!on) return;
    double t = now();
    fprintf(stderr, "[hyphy_hip trace] %s/%s %.1f us\n", what, stage, t - t0);
    t0 = t;
  }
};

#define HIPCHK(expr)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) {                                                                       \
      return fail(std::string(#expr) + ": " + hipGetErrorString(e_));                             \
    }                                                                                             \
  } while (0)

constexpr int kTimingRing = 1024;

// Subtree repeats: one class table per compressed internal node / leaf with ambiguity codes, per shard
struct RepTable {
  int U = 0, rows = 0;       // classes of this shard, padded to a multiple of 16
  int64_t row0 = 0;          // first row in the shard's table buffer
  std::vector<int64_t> map0; // per child: offset of its index map in rep_map
};

struct RepLaunch { size_t off; int qcap, n_static, n_waves; };  // one launch of the lower phase: its items in Shard::rep_items (repeats.hip)
// The item queues of the last few distinct passes (r06): an optimiser's partial updates move from branch to branch and
// come back — each dirty set keeps its queues on the device in a slot of its own (device block + pinned staging block + an event
// recorded behind the launches that read them), so a set that returns costs nothing and a new one overwrites the least recently used
// slot without waiting for the stream.
constexpr int kRepPassSlots = 4;
struct RepPassSlot {
  int4 *items = nullptr, *h_items = nullptr;
  size_t cap = 0;
  int qcap = 0, waves = 0, n_static = 0;
  bool team = false;
  std::vector<RepLaunch> launches;
  hipEvent_t ev = nullptr;
  bool ev_recorded = false;
  bool used = false;                 // launches have read the slot since `ev` was recorded last (recorded when the pass moves to another slot)
};

struct Shard {
  int device = 0;
  hipStream_t stream = nullptr;      // stream in use
  hipStream_t own_stream = nullptr;  // stream created (and destroyed) by the library
  int64_t s0 = 0, S = 0;  // pattern range [s0, s0+S) of the partition
  int S_pad = 0, ntiles = 0, T = 1, cus = 256;
  int16_t *codes = nullptr;
  double *freq = nullptr;
  double *ambig = nullptr;
  double *partials = nullptr;  // [C] x per-class block
  int32_t *counts = nullptr;   // [C][I][S_pad]
  double *site_lik = nullptr;  // [C][S_pad]
  int32_t *site_cnt = nullptr;
  double *mixed_lik = nullptr;
  int32_t *mixed_cnt = nullptr;
  double *Pfrag = nullptr, *PTg = nullptr, *Prow = nullptr;  // [C][B]...
  double *qbuf = nullptr;                                   // [C*B*D*D]
  int32_t *slots = nullptr;                                 // [C*B]
  int4 *ops = nullptr;
  int16_t *codes_tile = nullptr;  // [tile][L][16] copy of the leaf table (wave-per-tile kernels)
  unsigned char *expm_need = nullptr;  // [B] which matrix images a branch's consumers read (ExpmArgs::need), or nullptr
  int16_t *pin = nullptr;         // [S_pad] pinned states (hyphy_hip_set_pinned_states)
  int4 *bc_ops = nullptr;         // branch cache: schedule of the re-rooted chain, its one-entry program table,
  int4 *bc_prog = nullptr;        //   the slot word and the rate matrix of the cached branch
  int32_t *bc_slot = nullptr;
  double *bc_q = nullptr;
  int4 *prog = nullptr;       // program table (forest scheduling): (offset, entries, parent program, child programs)
  int4 *h_prog = nullptr;
  double *ar_buf = nullptr;   // device scalar: this shard's partial log-L, all-reduced in place over RCCL
  void *comm = nullptr;       // ncclComm_t of this shard (hyphy_hip_comm_init_rank / single-process group)
  double *mix_q = nullptr, *mix_p = nullptr, *mix_w = nullptr;  // branch-site mixtures: component rate matrices, their exponentials, weights
  int *mix_off = nullptr;
  size_t mix_cap = 0, mix_nq_cap = 0;
  int4 *jn = nullptr;         // chain schedules: per internal node (parent, arrivals needed | child sum << 8, trunk entries offset, count)
  int4 *h_jn = nullptr;
  double *deposits = nullptr; // chain schedules: [C][view's I][ntiles][TILE] edge products of non-last arrivers (allocated on first use: ensure_deposits)
  size_t deposits_cap = 0, deposits_class_stride = 0;  // doubles
  size_t dev_bytes = 0;       // device memory this shard holds
  size_t rep_bytes = 0;       // ... of which class tables and their maps (repeats.hip)
  int *frag_ctr = nullptr;    // [classes][programs][tiles] arrivals of child fragments (wave-per-tile kernel)
  int32_t *hand_cnt = nullptr;  // [classes][I][tiles][32] 2^64-exponents of fragment roots (own 128-byte line each)
  double *pi = nullptr;       // [DP]
  double *out = nullptr;      // [2]
  double *pi_ones = nullptr;  // [DP] 1 for real states, 0 for padding: root frequencies of a re-rooted schedule (pi sits in a twin image)
  bool twins_dirty = true;    // the transposed twin images (expm.hip) may lag behind the matrices they mirror
  double *wg_sum = nullptr;   // per-workgroup partial sums of the pruning kernel
  long long *wg_cnt = nullptr;
  int *wg_flag = nullptr;
  int wg_cap = 0;
  int32_t *status = nullptr;  // [1]
  double *weights = nullptr;  // [C]
  double *templates = nullptr;
  double *templates_pad = nullptr;  // [K][64*64] zero-padded, zero diagonals (expm64_kernel; DP == 64 only)
  double *coeffs = nullptr;
  // pinned host staging
  int4 *h_ops = nullptr;
  double *h_out = nullptr;    // pinned, host-mapped: the reduction kernel writes [log-L, scaler sum, status] here
  double *d_hout = nullptr;   // device-side address of h_out
  double *h_export = nullptr; // per-pattern results in the caller's order (host-mapped pinned: [S] doubles, [S] int64), see site_export_kernel
  bool export_failed = false;     // the export's buffers could not be set up once: this shard keeps the copies of gather_sites
  double *d_export = nullptr; // device-side address of h_export
  int32_t *d_inv = nullptr;   // device pattern of caller pattern i (nullptr: identity)
  int exported = 0;           // what the last enqueued evaluation exported (1: values, 2: exponents)
  int32_t *h_slots = nullptr;
  double *h_coeffs = nullptr;  // pinned ring (4 x C*B*K) for build_q coefficients
  double *d_hcoeffs = nullptr; // ... as the device sees it (host-mapped): the fused expm kernel reads it directly
  const double *coeffs_cur = nullptr;  // coefficients of the pending fused build (device-visible pointer)
  unsigned coeff_turn = 0;
  // The fused expm kernel reads a ring slot over PCIe when it EXECUTES.  On the asynchronous path
  // (hyphy_hip_evaluate_device) the host may run ahead of the device: an event recorded behind the consuming launch
  // guards the slot, and hyphy_hip_build_q waits for it before rewriting the slot.
  hipEvent_t coeff_ev[4] = {nullptr, nullptr, nullptr, nullptr};
  double *h_site = nullptr;       // pinned staging of per-pattern results on their way to the caller (gather_sites): [S_pad] doubles + [S_pad] int32
  double *h_tstage = nullptr;     // pinned ring of 2: [K D D unpadded | K 64 64 padded] template images on their way to the device (update_q_templates)
  size_t tstage_slot = 0;         // doubles per ring slot
  hipEvent_t tstage_ev[2] = {nullptr, nullptr};
  bool tstage_busy[2] = {false, false};
  unsigned tstage_turn = 0;
  bool coeff_busy[4] = {false, false, false, false};
  int coeff_slot = -1;                 // ring slot of the staged coefficients
  int64_t coeff_rows = 0;              // rows staged by the last hyphy_hip_build_q (0: nothing staged)
  int coeff_kind = 0;                  // what consumed them first: 0 nobody yet, 1 one row per (class, branch), 2 one row per (branch, mixture component)
  bool qbuf_built = false;             // ... and materialised in qbuf (HYPHY_HIP_MATERIALIZE_Q)
  double *h_small = nullptr;  // pi / weights staging
  size_t h_small_cap = 0;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_ar[2] = {nullptr, nullptr};  // around the all-reduce of hyphy_hip_evaluate*_allreduce (timing detail on)
  std::vector<hipEvent_t> ring;  // kTimingRing pairs (start, end) around the pruning launches
  uint64_t ring_count = 0;       // evaluations stamped so far
  uint64_t eval_count = 0;
  bool last_stamped = false;     // the most recent evaluation carried a kernel-duration stamp (hyphy_hip_last_timings)
  size_t partial_stride = 0;  // doubles per class
  // completion of the synchronous entry points: the reduction kernel writes a sequence number behind the result
  // record in host-mapped memory and the host spins on it (hipStreamSynchronize costs several microseconds more)
  double seq_next = 1., seq_wait = 0.;
  // per-site batched fits (hyphy_hip_site_fits_evaluate), allocated on first use
  double *fit_Timg = nullptr, *fit_bcoef = nullptr, *fit_smult = nullptr, *fit_smix = nullptr, *fit_out = nullptr,
         *fit_scratch = nullptr, *fit_pi = nullptr;
  int *fit_bgroup = nullptr;
  int32_t *fit_scratch_cnt = nullptr;
  int4 *fit_ops = nullptr;
  size_t fit_sets_cap = 0, fit_scratch_sets = 0;
  bool fit_static_current = false;  // template images + schedule on the device match the host copies
  // subtree repeats (repeats.hip): class tables of the compressed nodes of this shard
  std::vector<RepTable> rep_tabs;    // per descriptor (hyphy_hip_partition::rep_nodes order): classes of this shard
  int64_t rep_rows = 0;              // table rows (classes, padded to whole tiles of 16) over all descriptors
  double *rep_tab = nullptr;         // [C][rep_rows][DP]   E_n[u] = P_n x (conditionals of class u below n), column-gather layout
  int32_t *rep_cnt = nullptr;        // [C][rep_rows]       their 2^64 exponents
  int32_t *rep_map = nullptr;        // per (descriptor, child): [rows of the descriptor] class of the child / leaf code
  int4 *rep_desc = nullptr;          // descriptor headers and child entries (repeats.hip: RepDesc)
  RepPassSlot rep_slot[kRepPassSlots];
  int rep_slot_cur = -1;             // slot the fields below mirror
  int4 *rep_items = nullptr;         // work items of the current pass, eight queues (= rep_slot[rep_slot_cur].items)
  int *rep_sync = nullptr;           // queue heads, exit counter, per (class, descriptor) finished tiles (zero between launches)
  int16_t *rep_codes_tile = nullptr; // [tile][view leaves][16] leaf table of the trunk: state codes / class ids of generalised leaves
  int2 *rep_leaf = nullptr;          // [view leaves] (table row0 or -1: ordinary leaf, exponent row0 / matrix slot)
  int4 *rep_walk = nullptr;          // the trunk as ONE post-order walk per tile (repeats.hip: trunk_walk_kernel), or nullptr
  std::vector<int2> rep_leaf_host;   // host copy of rep_leaf (the walk's program carries it)
  int rep_qcap = 0;                  // items per queue of the pass the device queues hold
  int rep_waves = 0;                 // waves its launch runs
  std::vector<RepLaunch> rep_launches; // non-empty: the pass runs one launch per level of table-reads-table dependencies
  int rep_static = 0;                // > 0: its items do not depend on one another and are dealt by position (RepArgs::n_static)
  bool last_walk = false;            // the last trunk pass ran trunk_walk_kernel (hyphy_hip_prune_kernel_name)
  bool last_nucgen = false;          // the last 4-state pruning launch ran a generated kernel (hyphy_hip_prune_kernel_name)
  bool rep_team = false;             // its static launches run the row-split walk (class_table_team_kernel), one workgroup per item
  bool rep_sync_dirty = false;       // a lower-phase launch has run and no trunk launch has reset rep_sync behind it yet
};

}  // namespace hyhip

struct hyphy_hip_partition {
  int64_t D = 0, S = 0, L = 0, I = 0, C = 1, B = 0;
  int DP = 0, NW = 0;
  bool nuc = false;
  bool nuc_leaf_pairs = false;               // 4 states: leaf entries carry up to two leaves (prune_nuc2_kernel; prune_nuc_kernel takes one)
  std::vector<int64_t> parents;              // [L+I]
  std::vector<std::vector<int>> children;    // per internal node, ascending node codes
  // The tree the schedule compiler and the pruning kernels walk.  views[0]: the partition's own tree.  views[1] (subtree
  // repeats, repeats.hip): the TRUNK — the internal nodes whose subtrees are not class-compressed — over generalised
  // leaves (ordinary leaves below trunk nodes, compressed subtree roots, leaves with ambiguity codes).  Internal
  // indices / leaf numbers are the view's own; `slot` maps a view node back to its transition-matrix slot (= node code
  // of the partition's tree).
  struct View {
    int L = 0, I = 0;
    std::vector<int64_t> parents;            // [L+I] view-internal index of the parent
    std::vector<std::vector<int>> children;  // per view-internal node: view node codes, ascending
    std::vector<char> leaf_has_ambig;        // [L]
    std::vector<int> slot;                   // [L+I] matrix slot (node code in the partition's tree)
  };
  View views[2];
  int mode = 0;                              // view in use (switch_mode)
  const View &vw() const { return views[mode]; }
  struct ModeState {                         // what the tuner / re-rooting decided, per view
    int variant = 0, wave_variant = 0, n_slots = 0, chain_m_forced = 0;
    bool rr_use = false, kernel_forced = false, nuc_leaf_pairs = false, trunk_walk = false;
    int64_t tuned_for = 0;
    std::string tune_report;
    std::vector<int> rr_path;
    std::vector<std::vector<int>> rr_cands;
  };
  ModeState saved_mode[2];
  // subtree repeats
  int export_sites = 0;                      // the evaluation being enqueued exports per-pattern results (1: values, 2: exponents)
  bool rep_on = false;                       // views[1] exists
  bool rep_enabled = true;                   // ... and ordinary evaluations use it (hyphy_hip_set_repeats)
  struct RepNode {                           // one class table (descriptor): a path of compressed internal nodes or a leaf with ambiguity codes
    int node = 0;                            // node code (partition's tree) of the node whose classes the table has: the path's top / the leaf
    int level = 0;                           // 0: no table among its inputs
    std::vector<int> path;                   // compressed nodes walked by one wave, in walking order (empty: a leaf's table)
    std::vector<int> flags;                  // per walked node: 1 first node of a side chain walked inline, 2 its last node
    std::vector<std::vector<int>> kids;      // per path node: its children off the path (node codes)
    std::vector<std::vector<int>> kid_desc;  // ... descriptor of the child's table, -1: ordinary leaf (gathered by state code)
  };
  std::vector<RepNode> rep_nodes;            // children before parents
  std::vector<int> rep_desc_of;              // [L+I] descriptor of a node's table, -1: none
  std::vector<char> rep_resident;            // per class: tables and trunk copies are current (mode 1 counterpart of `resident`)
  struct RepPassKey { std::vector<int> dirty; int classes = 0; bool valid = false; uint64_t stamp = 0; };
  RepPassKey rep_pass[hyhip::kRepPassSlots];  // what each slot of the shards' item-queue rings was built for
  uint64_t rep_pass_clock = 0;
  bool rep_cached_valid = false;             // (false: every slot is stale — the tables or the trunk changed)
  int64_t rep_stale_branch = -1;             // a branch whose matrix image was rewritten outside an evaluation (branch cache)
  double rep_kernel_ms = 0.;
  std::vector<hyhip::Shard> shards;
  void *xch = nullptr;                       // host-side exchange of one-process-per-GPU runs (comm.hip: HostExchange), or nullptr
  std::vector<char> initialized;             // per class: a full evaluation has populated the caches
  std::vector<char> leaf_has_ambig;          // per leaf: any ambiguity code in its row of the leaf table
  std::vector<int4> ops_host;
  uint64_t nucgen_key = 0;                   // 4 states: key of the run-time generated kernel of the current schedule (nucgen.hip), 0: none
  int nucgen_uses = 0;                       // ... evaluations under it so far
  bool nucgen_small = false;                 // ... in its small-shard form (matrices in LDS, exponentials and combine inside the launch)
  bool nucgen_asked = false;                 // ... its compilation has been requested
  std::vector<int64_t> cached_update;        // update list the device schedule was built for
  bool cached_full = false;
  int cached_valid = 0;
  // Lazy persistence of the conditionals (cache_policy 1, default; HYPHY_HIP_CACHE=always turns it off): a full
  // pass that follows a full pass (a sweep over a global parameter) keeps its nodes in registers / LDS only —
  // nothing reads the persisted copies before the next full pass overwrites them — except the nodes some later
  // schedule entry of the same pass re-reads.  `resident[c]`: the persisted copies of class c are current;
  // a partial update, a branch-cache build or a download that finds them stale first re-runs a persisting pass.
  int cache_policy = 1;
  std::vector<char> resident, last_full;
  bool cached_persist = true, sched_persist = true, sched_full = true;
  std::vector<double> cached_pi;             // root frequencies currently on the device
  std::vector<double> cached_weights;        // category weights currently on the device
  std::vector<std::vector<int64_t>> cached_slots;  // per class: q_nodes list currently on the device
  int root_slot = 0;
  // Re-rooted schedules.  The likelihood does not depend on where the pruning recursion is rooted if the edges between the given
  // root and the new one are traversed with the transposed matrices (and pi is folded in on the old root's edge) — no
  // reversibility assumed, the same identity the branch cache uses.  A root in the middle of the tree shortens every tile's
  // critical path (the tree's height), which is what small and medium shards are bound by.  rr_path = internal indices from the
  // given root (front) to the node the computation is rooted at (back); empty: the given root is already the best one.
  std::vector<int> rr_path;
  std::vector<std::vector<int>> rr_cands;    // the (at most two) height-minimising nodes' paths; rr_path is the one in use
  int emit_skip_par = -1, emit_skip_child = -1;  // (transient, build_schedule -> emit_program)
  bool rr_use = false;                       // ask build_schedule for the re-rooted form (tuner / HYPHY_HIP_REROOT)
  bool rr_active = false;                    // the current schedule is a re-rooted one
  std::vector<int64_t> perm;                 // internal pattern j = caller's pattern perm[j] (empty: identity); see sort_patterns()
  int64_t pin_node = -1;                     // node code whose states are pinned for the evaluations that follow (-1: none)
  std::vector<int64_t> bc_node;              // per rate class: branch whose outside vector is resident (-1: none)
  std::vector<int> bc_use_pi;                // ... hangs off the root (frequencies applied at evaluation)
  int variant = 0;                           // pruning kernel variant (common.h PruneArgs::variant)
  bool trunk_walk = false;                   // (views[1]) lazy full passes of the trunk run trunk_walk_kernel; `variant` (0) serves the rest
  std::vector<int4> rep_walk_host;           // its program (empty: the trunk is deeper than the walk's stack, or has too many leaves)
  int n_slots = 0;                           // LDS slots the schedules are compiled for (0: lds_slots(T))
  struct Prog { int off, n, parent = -1, need = 0; };
  bool chain = false;                        // the current schedule is a chain schedule (common.h PruneArgs::chain)
  bool kernel_forced = false;                // HYPHY_HIP_KERNEL / T > 1: the tuner must not switch kernels
  int n_slots_wave = 3;                      // LDS slot budget of the wave-per-tile kernel's schedules
  double *h_qstage = nullptr;                // pinned copy of the caller's matrices for hyphy_hip_evaluate_async
  size_t h_qstage_cap = 0;
  bool async_pending = false;                // an asynchronous evaluation has not been collected yet
  int64_t async_cat = 0;
  int wave_variant = 0;                      // instantiation of the wave-per-tile kernel (0: 2 waves per SIMD; 2: 3 waves per SIMD,
                                             // finalised node in LDS, no parking slot) — chosen by the schedule tuner
  int chain_m_forced = 0;                    // cut chosen by the schedule tuner: > 0 source size limit m, -1 level-peeled fragments, 0 heuristic
  int64_t tuned_for = 0;                     // batch_classes the tuner ran for (0: not yet)
  std::string tune_report;                   // what the tuner measured (hyphy_hip_schedule_info)
  double tuned_ms = 0.;                      // duration of the pruning pass under the schedule it chose
  bool rep_decided = false;                  // subtree repeats on / off has been settled by measurement (or by the caller)
  std::string rep_report;                    // ... and what was measured
  std::vector<int4> jn_host;                 // ... its per-node join table
  struct Level { int first, count; };
  std::vector<Prog> programs;                // (offset, padded entry count) into ops_host
  std::vector<Level> levels;                 // launches: programs [first, first+count) run concurrently
  int64_t batch_classes = 1;                 // rate classes batched into the pruning launch being scheduled
  int slots_batch_mode = -1;                 // whether the slot table on the device was written for a class batch
  bool all_timings = getenv("HYPHY_HIP_ALL_TIMINGS") != nullptr;  // also stamp expm / reduction (2 more event records)
  bool coeffs_pending = false;               // build_q staged coefficients; the next evaluate_device(q_buffer) fuses
                                             // the rate-matrix construction into the expm kernel
  int64_t K = 0;                             // Q templates
  std::vector<double> templates_host;        // [K][D][D] as passed to hyphy_hip_set_q_templates
  std::vector<int4> fit_ops_host;            // per-site fits: full schedule compiled for kSiteFitParkSlots parking slots
  int fit_n_ops = 0;
  bool fit_spills = false;                   // ... some node goes through the scratch copy
  double fit_kernel_ms = 0.;                 // duration of the last site-fit kernel (max over shards)
  double timings[3] = {0, 0, 0};
  double allreduce_ms = 0.;                  // duration of the last in-stream all-reduce (timing detail on; else 0)
};

namespace hyhip {

inline size_t ops_capacity(const hyphy_hip_partition *p) { return (size_t)(p->L + p->I) + 4 * (size_t)p->I + 8; }
inline int64_t caller_pattern(const hyphy_hip_partition *p, int64_t j) { return p->perm.empty() ? j : p->perm[j]; }
inline int twin_slot0(const hyphy_hip_partition *p) { return (int)(p->B + (p->I + 2)); }

// "use the shard's own Q buffer" (filled / staged by hyphy_hip_build_q on every shard): compared by address
extern const double kOwnQBuffer;

// branch-site mixture: matrix k of the evaluation is sum_m weights[off_k + m] exp(q[off_k + m]), count[k] components
constexpr int64_t kMixRows = 16;  // most components of an explicit-form mixture per branch
struct MixSpec {
  const int64_t *count;
  const double *weights;
  int64_t n_tot;
};

// schedule.hip
void init_plain_view(hyphy_hip_partition *p);
int emit_program(hyphy_hip_partition *p, const std::vector<int> &nodes, int *offset_out, int *n_out, bool handoff = false,
                 bool is_root_program = true);
void build_schedule(hyphy_hip_partition *p, const int64_t *update_nodes, int64_t n_update, bool full);
void reroot_path(hyphy_hip_partition *p);
void sort_patterns(hyphy_hip_partition *p, const int64_t *leaf_codes, int64_t L, int64_t S);
// tuner.hip
int upload_schedule(hyphy_hip_partition *p, Shard &s);
void launch_prune_current(hyphy_hip_partition *p, Shard &s, int cat, int n_cat_batch);
int tune_schedule(hyphy_hip_partition *p, int cat, int n_cat_batch);
// api.hip
void refresh_twins(hyphy_hip_partition *p, Shard &s);
PruneArgs base_prune_args(hyphy_hip_partition *p, Shard &s, int cat, int n_cat_batch);
int ensure_deposits(hyphy_hip_partition *p, Shard &s);
int eval_common(hyphy_hip_partition *p, int64_t cat, const int64_t *update_nodes, int64_t n_update, const int64_t *q_nodes,
                int64_t n_q, const double *q, bool q_on_device, int q_is_probability, const double *root_freqs,
                double *d_logl_out, bool reduce, bool floor_log, bool batch = false, bool force_persist = false,
                const MixSpec *mix = nullptr);
int finish_pending_async(hyphy_hip_partition *p);
int collect_status(hyphy_hip_partition *p);
int publish_and_collect(hyphy_hip_partition *p, const double *d_value, double *value_out);  // (single-shard partitions)
void record_timings(hyphy_hip_partition *p);
double combine(const std::vector<double> &parts);
// repeats.hip
struct RepArgs;
int rep_setup(hyphy_hip_partition *p, const std::vector<std::vector<int16_t>> &codes);
void switch_mode(hyphy_hip_partition *p, int mode);
int rep_prepare_pass(hyphy_hip_partition *p, const int64_t *update_nodes, int64_t n_update, const int64_t *q_nodes, int64_t n_q, bool full,
                     int cat0, int n_classes, std::vector<int64_t> &view_update);
int rep_launch(hyphy_hip_partition *p, Shard &s, int cat0);
// the trunk's lazy full pass as one row-split walk per tile (trunk_walk_kernel); false: not applicable to this pass — the caller runs the
// pruning kernels.  `timeline`: HYPHY_HIP_WALK_TIMELINE diagnostics allowed (evaluations, not the tuner's passes)
bool trunk_walk_applies(const hyphy_hip_partition *p, const Shard &s);
bool trunk_walk_fuses_reduce(const hyphy_hip_partition *p);  // its launch can carry the fused final combine (PruneArgs::red_*)
int launch_trunk_walk(hyphy_hip_partition *p, Shard &s, int cat, int n_cat_batch, bool timeline, const hyhip::PruneArgs *red = nullptr);
int rep_decide(hyphy_hip_partition *p, int cat, int n_classes);
bool rep_static_decision(const hyphy_hip_partition *p);  // on / off without a measurement (tuner disabled, forced cut)
size_t rep_sync_words(const hyphy_hip_partition *p);
int rep_sync_stride();
// comm.hip
int combine_shards(hyphy_hip_partition *p, double *logl_out);

}  // namespace hyhip
