// Internal definitions shared by the HIP translation units of libhyphy_hip.so.
// gfx950 (MI355X / CDNA4) only: wave64, v_mfma_f64_16x16x4_f64, 160 KiB LDS per CU.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

namespace hyhip {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

// Rescaling constants of the reference (src/core/tree.cpp:126-129).
constexpr double kScalerUp = 18446744073709551616.0;            // 2^64   _lfScalerUpwards
constexpr double kScalerThreshold = 1.0 / 18446744073709551616.0;  // 2^-64  _lfScalingFactorThreshold
constexpr double kLogScaler = 64.0 * 0.69314718055994530942;    // _logLFScaler

// ---------------------------------------------------------------------------------------
// Device-private "fragment" layout of one tile of 16 site patterns x DP states
// (DP = 16*NW, NKK = DP/4 k-steps of v_mfma_f64_16x16x4_f64).
//
//   element (state j, site s) lives with lane = (j & 3) * 16 + s, k-step kk = j >> 2
//   at double index FRAG(kk, lane) = ((kk >> 1) * 64 + lane) * 2 + (kk & 1)
//
// i.e. exactly the B-operand register image of the MFMA (lane l holds B[k = l>>4][n = l&15]
// for each k-step), stored so that one 16-byte access per lane fetches two k-steps and a
// wave's access is one contiguous 1 KiB line.  The MFMA's C/D image (row = 4*r + (l>>4),
// col = l&15 for accumulator register r of row-block w) is the SAME map with kk = 4*w + r,
// so a node's freshly computed conditionals feed its parent's product with no data movement.
// ---------------------------------------------------------------------------------------
__host__ __device__ inline int frag_index(int kk, int lane) { return (((kk >> 1) * 64 + lane) << 1) + (kk & 1); }

// Host-compiled schedule entry (int4), built by build_schedule() in api.hip:
//   x  bits 0-1  kind: OPK_LEAF (group of <= 2 leaf children), OPK_INTERNAL (child vector in an LDS
//                slot), OPK_INTERNAL_GLOBAL (child vector = persisted copy in HBM), OPK_DEP (chain schedules of the
//                wave-per-tile kernel: the edge product of internal child w was deposited by the wave that computed it)
//      bit  2    OPF_HANDOFF (wave-per-tile kernel, chained fragments) internal-global entry: the child is the root of
//                a fragment finished by ANOTHER workgroup of this launch — read it with agent-scope (sc1) loads
//      bit  3    OPF_LAST    last child of its parent: finalise the parent
//      bit  4    parity of this finalisation (which psum buffer / exchange slot pair member)
//      bit  5    OPF_GSYNC   (internal-global) copy was written earlier in THIS launch: full fence first
//      bit  6    OPF_AMBIG   (leaf group) some leaf of the group carries ambiguity codes in this shard
//      bit  7    OPF_INREGS  (nucleotide kernel only) child is the node finalised by the previous entry
//                OPF_NOPERSIST (MFMA kernels, OPF_LAST entries) lazy persistence: do not store the finalised parent
//      bits 8-13 number of leaves in a leaf group
//      bit  14   OPF_NOSCALE (wave-per-tile kernel, OPF_LAST entries) no rescaling test at this node (schedule.hip: thin_rescale_tests)
//      bit  15   OPF_PUBLISH (wave-per-tile kernel, OPF_LAST entries) the finalised parent is the root of a fragment that
//                another workgroup of this launch consumes — publish it with sc1 stores.  (Its own bit: the last entry
//                of a fragment root may itself be an internal-global entry, with or without OPF_HANDOFF.)
//      bits 16-23 destination LDS slot of the finalised parent (0/1 exchange slots, >= 2 parking)
//      bits 24-31 source LDS slot of an OPK_INTERNAL child
//   y  parent internal index
//   z  internal: child node code (= transition-matrix slot);  leaf group: leaf0 | leaf1 << 16
//   w  internal: child internal index
// A parent's first entry needs no flag: the running product is reset when a parent is finalised.
enum : int { OPK_LEAF = 0, OPK_INTERNAL = 1, OPK_INTERNAL_GLOBAL = 2, OPK_DEP = 3 };
enum : int { OPF_HANDOFF = 4, OPF_LAST = 8, OPF_PARITY = 16, OPF_GSYNC = 32, OPF_AMBIG = 64, OPF_INREGS = 128,
             OPF_NOPERSIST = 128, OPF_NOSCALE = 0x4000 /* (wave-per-tile kernel, OPF_LAST entries: no rescaling test at this node) */, OPF_PUBLISH = 0x8000, OPF_NOPERSIST_NUC = 4 /* (nucleotide kernel: bit 7 is OPF_INREGS there) */ };
#ifndef HYPHY_SLOTS1
#define HYPHY_SLOTS1 5
#endif
__host__ __device__ constexpr int lds_slots(int T) { return T == 1 ? HYPHY_SLOTS1 : (T == 2 ? 4 : (T == 3 ? 3 : 2)); }

struct PruneArgs {
  const int4 *ops;
  const int4 *prog;          // program table: (offset into ops, padded entry count, parent program or -1, number of
                             // child programs); grid.z indexes the programs of this launch
  int n_prog;                // programs (subtree fragments) in this launch = grid.z
  int do_root;               // this launch finalises the root: run the root / log-sum epilogue
  int n_ops;                 // longest program of the launch (trace buffer stride); every program is padded to
                             // an even count and followed by two no-op entries
  int root_slot;             // LDS slot the root was finalised into
  int NW;                    // row blocks (waves per workgroup) = DP/16
  int T;                     // 16-pattern tiles per workgroup
  int S_pad;                 // patterns padded to 16*T
  int ntiles;                // S_pad / 16
  int root_inode;            // I - 1
  int L;                     // leaves
  int codes_in_lds;          // leaf codes of the workgroup's tiles are staged in LDS (L*T*32 bytes fit)
  int n_prog_total;          // programs in the whole table (chained fragments: stride of frag_ctr)
  int *frag_ctr;             // [class][program][tile] arrivals of finished child fragments (zero between launches)
  int32_t *hand_cnt;         // [class][I][tile][32] exponents of fragment roots handed between workgroups
  int variant;               // 0: workgroup-per-tile kernel (prune_mfma_kernel), 1: wave-per-tile kernel (T = 1)
  // chain schedules (wave-per-tile kernel, full passes): the tree is cut into SOURCE programs (small bottom subtrees,
  // walked serially) and a TRUNK of join nodes above them.  A wave starts at a source, computes the edge product
  // towards the parent and arrives at it; the LAST arriver of a node multiplies the deposited products of its
  // siblings in, finalises the node and goes on upwards, everybody else deposits its product and retires.
  int wave_variant;          // 0: production instantiation of prune_wave_kernel; > 0: experimental ones (HYPHY_HIP_WAVE_VARIANT)
  int chain;                 // 1: grid = (tiles, classes, sources), programs sorted by distance to the root
  const int4 *jn;            // [I] per internal node: (parent internal index or -1, arrivals needed | sum of internal child
                             //     indices << 8, offset of the node's trunk entries in ops, number of entries)
  double *deposits;          // [I][ntiles][NKK*64] edge product of (node -> parent), written by non-last arrivers
  size_t cs_deposits = 0;    //     class stride (I = internal nodes of the view the schedule was cut from)
                             //     (exponents: hand_cnt; arrival counters: frag_ctr indexed by node)
  int n_slots;               // LDS slots the schedule was compiled for (2 exchange + parking)
  const double *Pfrag;       // [B][NW][NKK*64]      A-operand images of the transition matrices
  const double *PTg;         // [B][DP][NW][4][4]    column-gather images (leaf edges)
  const int16_t *codes;      // [L][S_pad]           >= 0 state, < 0 -> -(k+1) ambiguity row
  const int16_t *codes_tile; // [ntiles][L][16]      the same table, tile-major (wave-per-tile kernels)
  // pinned node states (hyphy_hip_set_pinned_states; ComputeBlock's branchIndex / branchValues): at most one node
  int pin_inode, pin_leaf;   // internal index / leaf index of the pinned node, -1 = none
  const int16_t *pin;        // [S_pad] state the node is fixed to at each pattern
  const double *ambig;       // [n_ambig][DP]
  double *partials;          // [I][ntiles][NKK*64]  conditionals, fragment layout
  int32_t *counts;           // [I][S_pad]           cumulative 2^64-exponent of the subtree
  const double *pi;          // [DP] root frequencies (zero padded)
  double *site_lik;          // [S_pad]
  int32_t *site_cnt;         // [S_pad]
  const double *freq;        // [S_pad] pattern frequencies (0 for padding)
  double *wg_sum;            // [n workgroups] sum_s f_s log L_s over the workgroup's patterns
  long long *wg_cnt;         // [n workgroups] sum_s f_s c_s
  int *wg_flag;              // [n workgroups] 1: zero-likelihood pattern, 2: NaN
  // rate-class batching: blockIdx.y = class; element strides between consecutive classes
  int n_cat;
  size_t cs_P;               // Pfrag / PTg       (B*DP*DP)
  size_t cs_partials;        // partials
  size_t cs_counts;          // counts            (I*S_pad)
  size_t cs_site;            // site_lik/site_cnt (S_pad)
  size_t cs_wg;              // wg_sum/cnt/flag   (workgroups per class)
  int ablate;                // DIAGNOSTIC ONLY (HYPHY_HIP_ABLATE bitmask, results invalid): 1 no MFMA, 2 no barrier,
                             // 4 no persist stores, 8 no leaf gathers, 16 no operand prefetch, 32 no LDS exchange
  long long *timeline;       // optional tracing: [kTraceWG][NW][n_ops][4] s_memtime stamps (HYPHY_HIP_TIMELINE)
  // fused final combine (r03): the wave that finalises the LAST root of the launch sums the per-tile partial sums itself
  // (fixed order, compensated) and publishes the result record — no reduction kernel behind the pruning launch.
  // red_out == nullptr: off (the caller launches wg_reduce_kernel).
  double *red_out;           // [1] log-L
  double *red_rec;           // [3] scaler sum, expm status, sequence word (the host spins on it when red_seq != 0)
  const int *red_status;     // expm status word of the shard (or nullptr)
  double red_seq;
  int *red_done;             // arrivals of finalised roots (zero between launches)
  int red_n;                 // roots the launch finalises = entries of wg_sum
  // subtree repeats (repeats.hip; prune_wave_kernel<REP>): the schedule's leaves are those of the trunk view
  const int2 *leaf_tab = nullptr;   // [L] (first row of the leaf's class table or -1: ordinary leaf, first exponent row / matrix slot)
  const double *gtab = nullptr;     // [rows][DP] class tables, column-gather layout
  const int32_t *gcnt = nullptr;    // [rows] their 2^64 exponents
  size_t cs_gtab = 0, cs_gcnt = 0;  // class strides
};
constexpr int kTraceWG = 8;
constexpr int kNucParkSlots = 4;  // LDS parking slots of the 4-state kernel (nodes whose parent is not the next entry)

constexpr int kCoefInline = 400;  // coefficients that fit the kernel-argument block (n * K doubles)
struct CoefInline {
  double c[kCoefInline];
};
struct CoefNone {  // (kernels that take the coefficient block only in some instantiations)
  int unused;
};
template <bool ON>
struct CoefArg {
  typedef CoefNone type;
};
template <>
struct CoefArg<true> {
  typedef CoefInline type;
};
struct ExpmArgs;
bool fill_coef_inline(ExpmArgs &b, CoefInline &ci);  // expm.hip

struct NucArgs {
  const int4 *ops;
  int n_ops;
  int S_pad;
  int L;                     // leaves (prune_nuc2_kernel keeps their transposed matrices in LDS)
  int root_inode;
  const double *P;           // [B][16] row-major 4x4
  const double *PT;          // [B][16] the same matrices transposed ([state j][row i]); nullptr: prune_nuc_kernel only
  const int16_t *codes;      // [L][S_pad]
  int pin_inode, pin_leaf;   // pinned node (see PruneArgs)
  const int16_t *pin;
  const double *ambig;       // [n_ambig][4]
  double *partials;          // [I][4][S_pad]  state-major planes (coalesced per state)
  int32_t *counts;           // [I][S_pad]
  const double *pi;          // [4]
  double *site_lik;
  int32_t *site_cnt;
  const double *freq;
  double *wg_sum;
  long long *wg_cnt;
  int *wg_flag;
  // fused final combine (see PruneArgs): nullptr = off
  double *red_out = nullptr;
  double *red_rec = nullptr;
  const int *red_status = nullptr;
  double red_seq = 0.;
  int *red_done = nullptr;   // arrivals of workgroups (zero between launches)
  // subtree repeats (repeats.hip; prune_nuc_kernel<PIN, REP>): the schedule's leaves are those of the trunk view
  const int2 *leaf_tab = nullptr;   // [L] (first row of the leaf's class table or -1: ordinary leaf, matrix slot of an ordinary leaf)
  const double *gtab = nullptr;     // [rows][4] class tables
  const int32_t *gcnt = nullptr;    // [rows] their 2^64 exponents
};

// The argument block of a run-time generated 4-state kernel (nucgen.hip; the generated source carries the same declaration)
struct NucGenArgs {
  const double *ambig;
  double *partials;
  int32_t *counts;
  const double *pi;
  double *site_lik;
  int32_t *site_cnt;
  const double *freq;
  double *wg_sum;
  long long *wg_cnt;
  int *wg_flag;
  long long S_pad;
  double *red_out;           // fused final combine (small-shard form; see PruneArgs), nullptr: off
  double *red_rec;
  const int *red_status;
  double red_seq;
  int *red_done;
};
struct NucGenExpm {          // what the small-shard form reads of ExpmArgs (declared as hyhip::ExpmArgs in the generated source)
  const double *Q;
  const int32_t *slots;
  int n;
  int is_prob;
  double *Prow;
  double *PTrow;
  int32_t *status;
  const double *templates;
  const double *coeffs;
  int K;
  int coef_inline;
};
int nucgen_mode();                                     // HYPHY_HIP_NUCGEN: 0 off, 1 background compilation (default), 2 synchronous
int nucgen_after();                                    // evaluations under one schedule before its kernel is requested
uint64_t nucgen_key(const int4 *ops, int n_ops, int L, bool lazy, bool small, int n_branches);
void nucgen_request(uint64_t key, const int4 *ops, int n_ops, int L, bool lazy, bool small, int n_branches, bool sync);
bool nucgen_ready(uint64_t key);
bool nucgen_launch(uint64_t key, const NucArgs &a, hipStream_t stream, bool small, int n_branches, const ExpmArgs *ex);
std::string nucgen_source(const int4 *ops, int n_ops, int L, bool lazy, bool small, int n_branches);

constexpr int kSiteFitParkSlots = 1;  // wave-private LDS parking slots of the per-site fit kernel (8 KiB each at D = 61); nodes
                                      // beyond that go through a scratch copy in HBM (cheap next to the series of an edge)

constexpr double kSiteFitMaxRate = 4096.0;  // largest uniformisation rate hyphy_hip_site_fits_evaluate accepts (64 sub-series)

// Per-site batched fits (sitefit.hip): one wave per (16-site tile, parameter set)
struct SiteFitArgs {
  const int4 *ops;           // full post-order schedule, compiled for 2 + kSiteFitParkSlots slots, lazy persistence
  int n_ops;
  int NW, L, I, ntiles, S_pad;
  int K, G, n_sets;          // templates, branch groups, parameter sets (grid.y)
  double dmax[4];            // max_i |T_k[i][i]|: the uniformisation rate of site s on branch b is sum_k x_k dmax_k
  const double *Timg;        // [K][NW][NKK*64] A-operand images of the templates, diagonal = -(row sum)
  const double *bcoef;       // [B][K] branch coefficients
  const int *bgroup;         // [B]    multiplier group of each branch
  const double *smult;       // [n_sets][S_pad][n_mix][G][K] site multipliers (0 for padding sites)
  int n_mix;                 // mixture components per site (1: plain per-site fits)
  const double *smix;        // [n_sets][S_pad][n_mix] mixture weights (n_mix > 1)
  const int16_t *codes_tile; // [ntiles][L][16]
  const double *ambig;       // [n_ambig][DP]
  const double *pi;          // [DP]
  const double *freq;        // [S_pad]
  double *scratch;           // [n_sets][I][ntiles][NKK*64] nodes that found no parking slot (may be null if none)
  int32_t *scratch_cnt;      // [n_sets][I][S_pad]
  double *site_logl;         // [n_sets][S_pad]
  int32_t *status;
};

constexpr int kMaxTwin = 32;  // longest path (in edges) between the given root and the root a re-rooted schedule computes at

struct ExpmArgs {
  const double *Q;           // [n][D*D] row-major (rate matrices, or probabilities if is_prob)
  const int32_t *slots;      // [n] destination branch slot (node code) or nullptr -> identity
  int n;
  int D;
  int is_prob;
  double *Prow;              // optional [.][D*D] row-major output (slot-indexed)
  double *PTrow = nullptr;   // optional (4 states): the transposed matrices [.][16], read by prune_nuc2_kernel
  double *Pfrag;             // optional [.][NW][NKK*64]
  double *PTg;               // optional [.][DP][NW][16]  column-gather image (leaf edges): [code][wb][g][r] = P[16wb + 4r + g][code]
  int32_t *status;           // [1] set to nonzero if any matrix failed (NaN / ill-conditioned)
  // optional fused rate-matrix construction: Q_m = sum_k coeffs[m][k] * templates[k]
  // off-diagonal, diagonal = -(row sum); when templates != nullptr, Q is ignored
  const double *templates;   // [K][D*D]
  const double *coeffs;      // [n][K]
  int K;
  int prof;                  // diagnostic: workgroup 0 stamps its phases (HYPHY_HIP_EXPM_PROF)
  const unsigned char *need = nullptr;  // (expm64_kernel, r06) per branch [need_B]: bit 0 its consumers read the A-operand image, bit 1 the
  int need_B = 0;                       // column-gather image; nullptr: both are written.  Slot s belongs to branch s % need_B
  int fixed_degree = 0;      // expm64_kernel: 1 = always degree 12 (HYPHY_HIP_EXPM_DEGREE=12); 0 = degree from the scaled norm
  // re-rooted schedules (api.hip: reroot_path): matrix j of twin_src (a slot number) also leaves the TRANSPOSED image
  // M[r][c] = P[c][r] (times twin_pi[c] for j == 0, the edge that leaves the old root) in slot twin_dst0 + j of Pfrag
  int n_twin = 0;
  int twin_src[kMaxTwin] = {};
  int twin_dst0 = 0;
  const double *twin_pi = nullptr;
  // expm64_kernel (49..64 states): the templates zero-padded to [K][64*64] with zero diagonals (aligned loads, no masks);
  // the host's view of `coeffs` when it is host-mapped memory (launch_expm may copy it into the kernel-argument block)
  const double *templates_pad = nullptr;
  const double *coeffs_host = nullptr;
  int coef_inline = 0;       // (set by launch_expm) the coefficients are in the kernel's second argument
};
void expm_read_profile(long long out[8]);

// Branch-cache evaluation (hyphy_hip_branch_cache_evaluate): L_s = sum_i A_s[i] (P_c B_s)[i]
struct BcArgs {
  int NW, S_pad, ntiles, L;
  int node_A;                // (virtual) node slot holding the outside vector A, relative to `partials`
  int child_internal;        // internal index of the cached branch's child node, or -1: it is leaf `child_leaf`
  int child_leaf;
  int use_pi;                // the branch hangs off the root: A does not contain the root frequencies yet
  const double *Pfrag;       // A-operand image of the branch's transition matrix
  const double *PTg;         // its column-gather image (leaf child)
  const int16_t *codes_tile;
  const double *ambig;
  const double *partials;
  const int32_t *counts;
  const double *pi;
  const double *freq;
  double *site_lik;
  int32_t *site_cnt;
  double *wg_sum;
  long long *wg_cnt;
  int *wg_flag;
};

// launchers (defined in the .hip files)
void launch_transpose_frag(const double *src_image, double *dst_image, const double *row_scale, int NW, hipStream_t stream);
void launch_bc_eval(const BcArgs &a, hipStream_t stream);
bool launch_expm(const ExpmArgs &a, hipStream_t stream);  // true: fused-construction coefficients were copied at launch
void launch_mix_images(const double *P, const int *off, const double *w, const int32_t *slots, int n, int D, double *Pfrag,
                       double *PTg, double *Prow, hipStream_t stream, double *PTrow = nullptr);
void launch_site_fit(const SiteFitArgs &a, hipStream_t stream);
void launch_prune_mfma(const PruneArgs &a, hipStream_t stream);
void launch_prune_nuc(const NucArgs &a, hipStream_t stream, const ExpmArgs *ex = nullptr);  // ex: matrix exponentials folded into the launch
bool prune_nuc_folds_expm(int L, int S_pad, int n_ops);
bool prune_nuc_fuses_reduce(const NucArgs &a, bool folded);  // launch_prune_nuc will run the instantiation that carries the fused final combine
void launch_site_export(const double *lik, const int32_t *cnt, const int32_t *inv, int S, double *out_lik, long long *out_cnt,
                        hipStream_t stream);
void launch_site_reduce(const double *site_lik, const int32_t *site_cnt, const double *freq, int S_pad, int floor_log,
                        double *out_logl, double *out_cnt, const int *status, hipStream_t stream, double seq = 0.);
void launch_wg_reduce(const double *wg_sum, const long long *wg_cnt, const int *wg_flag, int n, double *out_logl,
                      double *out_cnt, const int *status, hipStream_t stream, double seq = 0.);
int prune_mfma_grid(const PruneArgs &a);
bool prune_fuses_reduce(const PruneArgs &a);  // launch_prune_mfma has a fused-combine instantiation for this launch form
int prune_nuc_grid(const NucArgs &a);
bool prune_nuc_takes_leaf_pairs(int L);  // the kernel launch_prune_nuc picks for L leaves reads leaf groups of two
void launch_mix_categories(const double *site_lik /*[C][S_pad]*/, const int32_t *site_cnt, const double *weights_dev,
                           int C, int S_pad, double *mixed_lik, int32_t *mixed_cnt, hipStream_t stream);
void launch_build_q(const double *templates, const double *coeffs, int n, int K, int D, double *Q, hipStream_t stream);
void launch_unpack_partials_mfma(const double *partials, int I, int ntiles, int NW, int D, int S, double *out,
                                 hipStream_t stream);

}  // namespace hyhip
