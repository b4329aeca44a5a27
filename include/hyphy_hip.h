/*
 * hyphy_hip.h — C-ABI of the MI355X-native phylogenetic likelihood core.
 *
 * This is the drop-in boundary for the ONE hot path of veg/hyphy that this repository
 * replaces:
 *
 *   _LikelihoodFunction::ComputeBlock            src/core/likefunc.cpp:10783-11289
 *     -> _TheTree::ExponentiateMatrices          src/core/tree.cpp:2932-3113
 *          -> _Matrix::Exponentiate              src/core/matrix.cpp:5537-5951
 *     -> _TheTree::ComputeTreeBlockByBranch      src/core/tree_evaluator.cpp:3556-4171
 *     -> Neumaier combine, - logU * scalers      src/core/likefunc.cpp:11046-11123
 *
 * The reference has no plug-in/FFI interface; its (dead) OpenCL hook shows the seam:
 * construct per partition in SetupLFCaches (likefunc.cpp:4182-4183, 4313-4316), destroy in
 * DeleteCaches/Cleanup (likefunc.cpp:10546-10551, 10594-10599) and one call from ComputeBlock
 * (historically launchmdsocl(...), likefuncocl.cpp:1043-1053).  The entry points below are
 * exactly what a HyPhy host adapter binds at those three touch-points (INTEGRATION.md shows
 * the patch).  Plain C: opaque handle, pointers and sizes only — no C++ or torch types.
 *
 * Conventions
 *   - all sizes int64_t, all reals double (hyFloat, include/hy_types.h:57)
 *   - return 0 = ok; > 0 = "unsupported here, use the CPU path" (no state change);
 *     < 0 = hard error, text in hyphy_hip_last_error() (adapter -> HandleApplicationError,
 *     likefunc.cpp:11284)
 *   - host arrays are borrowed for the duration of the call only
 *   - node codes: n < L is leaf n, n >= L is internal node n-L; both in post-order, root
 *     last (tree.cpp:722-766).  P[i*D+j] = Pr(parent state i -> child state j)
 *   - never called concurrently for one partition (ComputeBlock is entered from the HBL
 *     interpreter's single thread; the library replaces the OpenMP region inside it)
 *   - there is NO CPU fallback inside the library: without a usable MI355X every compute
 *     entry point returns < 0.
 */
#ifndef HYPHY_HIP_H
#define HYPHY_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hyphy_hip_partition hyphy_hip_partition; /* device-side state of ONE (filter, tree) partition */

/*
 * Environment switches.  A host integrator needs at most these TEN; the library works without any of them:
 *   HYPHY_HIP_VERBOSE=1           what was decided and why (schedule tuner, subtree repeats on / off, generated kernels) on stderr
 *   HYPHY_HIP_POOL_MB=n           device / pinned blocks of destroyed partitions kept for the next one of the same shape (MiB per kind,
 *                                 default 1024; 0: nothing is kept)
 *   HYPHY_HIP_CACHE=always        every pass stores every node's conditionals (default: lazy persistence, see hyphy_hip_download_partials)
 *   HYPHY_HIP_TUNE=0              no run-time schedule measurement (with HYPHY_HIP_CUT=levels: bit-identical repeats, see below)
 *   HYPHY_HIP_CUT=levels          fixed multiplication order, no arrival-order joins
 *   HYPHY_HIP_REPEATS=0|1         subtree repeats (the reference's `tcc`) never / always, instead of decided by measurement
 *   HYPHY_HIP_NUCGEN=0|2          4 states: never generate straight-line kernels / compile them synchronously at the first full pass
 *   HYPHY_HIP_NUCGEN_AFTER=n      ... evaluations under one schedule before its kernel is compiled in the background (default 8)
 *   HYPHY_HIP_COMBINE=rccl        one process, several devices: sum the shard partials by one RCCL group all-reduce instead of on the host
 *   HYPHY_HIP_EXCHANGE_TIMEOUT_S  one process per GPU, host-side exchange: how long a rank waits for the others (default 120)
 * Everything else that starts with HYPHY_HIP_ (about forty names: HYPHY_HIP_KERNEL, _CHAIN_M, _WAVE_VARIANT, _SLOTS, _REROOT, _FRAGMENT,
 * _TILES, _REP_*, _TRUNK_*, _NUC*, _EXPM*, _FUSED_REDUCE, _SITE_EXPORT, _SPIN, _XCD_PAD, _TIMELINE, _ABLATE, _POISON, _TRACE, ...) is a
 * DIAGNOSTIC of the kernels' authors: it forces one of the forms the tuner chooses between, turns a mechanism off for an A/B run or
 * writes a trace.  The tests use them to reach every form; a host should not.  (The adapter of INTEGRATION.md has a handful of its own:
 * HYPHY_HIP=1 turns it on, HYPHY_HIP_DEVICE(S), HYPHY_HIP_WORLD / _RANK / _LOCAL_RANK / _RUN_ID / _UID_FILE / _COLLECTIVE for one
 * process per GPU, HYPHY_HIP_DEVICE_EXPM, HYPHY_HIP_TEMPLATES, HYPHY_HIP_MIXTURES, HYPHY_HIP_CAT_BATCH.)
 */

/* Number of usable gfx950 devices (0 if none / no HIP runtime). */
int hyphy_hip_device_count(void);

/*
 * Replaces the allocation half of _LikelihoodFunction::SetupLFCaches (likefunc.cpp:4163-4319)
 * for one partition: conditional-likelihood caches (conditionalInternalNodeLikelihoodCaches,
 * :4220-4223), scaling state (siteScalingFactors :4232, siteCorrections :4238-4246), the leaf
 * state table (conditionalTerminalNodeStateFlag :4235-4236, :4305) and the ambiguity vectors
 * (conditionalTerminalNodeLikelihoodCaches :4263-4311) all live on the device afterwards.
 *
 *   D,S,L,I,C       GetDimension, GetPatternCount, GetLeafCount, GetINodeCount, categoryCount
 *   flat_parents    [L+I] parent as internal index, root = -1            (tree.cpp:746-757)
 *   leaf_codes      [L*S] pattern-indexed; >= 0 state, < 0 -> -(k+1) = ambiguity vector k
 *   ambig           [n_ambig*D]
 *   pattern_freq    [S] theFrequencies
 *   device_first, device_count
 *                   patterns are sharded in contiguous ranges over devices
 *                   device_first .. device_first+device_count-1 of THIS process (the
 *                   reference's OpenMP site blocks, likefunc.cpp:10995-11044).  One rank per
 *                   GPU (torch.distributed / HYPHYMPI) passes device_count = 1 and its own
 *                   shard of the patterns.
 * Returns > 0 (unsupported) for D > 64 or D < 2 in this version.
 */
int hyphy_hip_create(hyphy_hip_partition **out, int64_t D, int64_t S, int64_t L, int64_t I, int64_t C,
                     const int64_t *flat_parents, const int64_t *leaf_codes, const double *ambig,
                     int64_t n_ambig, const int64_t *pattern_freq, int device_first, int device_count);

/* Replaces DeleteCaches (likefunc.cpp:10556-10600) for the partition. NULL is a no-op.  The partition's device and pinned host
 * blocks and its stream are kept for the next hyphy_hip_create of the same shape in this process (analyses that build one
 * likelihood function per site: FEL); HYPHY_HIP_POOL_MB caps what is kept (MiB per kind, default 1024, 0: nothing). */
void hyphy_hip_destroy(hyphy_hip_partition *p);

/*
 * One likelihood evaluation of rate class `cat` (-1 == 0 when C == 1): replaces, inside
 * ComputeBlock, ExponentiateMatrices (likefunc.cpp:10978-10980) AND the OpenMP pruning loop
 * + combine (likefunc.cpp:10995-11123).
 *
 *   update_nodes    node codes from _TheTree::DetermineNodesForUpdate (tree.cpp:3117-3331),
 *                   ascending; on the first evaluation after create: all L+I-1 branches
 *                   (likefunc.cpp:10965-10967).  The library recomputes every internal node
 *                   that is the parent of a listed node, from ALL of that node's children.
 *   q_nodes/q_dense node codes whose transition matrix changed and their numeric rate
 *                   matrices [n_q*D*D], already multiplied by the branch length, MultByFreqs
 *                   applied (what _CalcNode::RecomputeMatrix hands to SetCompExp,
 *                   calcnode.cpp:526-735).  q_is_probability = 1: q_dense already holds
 *                   P = exp(Q) (host did the exponential, e.g. explicit-form mixtures
 *                   P = sum_k w_k exp(Q_k), tree.cpp:3047-3090).
 *   root_freqs      [D] theProbs (InitializeTreeFrequencies, tree.cpp:2436-2451)
 *   logl_out        sum_s f_s log L_s - 64 ln2 * scalers — the value ComputeBlock returns
 *                   (likefunc.cpp:11123).  -INFINITY if a pattern has likelihood 0
 *                   (tree_evaluator.cpp:4094-4112); NaN propagates (adapter then calls
 *                   _TerminateAndDump as tree_evaluator.cpp:4142 does).
 *   site_lik_out    optional [S], pattern-indexed == storageVec (tree_evaluator.cpp:4080):
 *                   per-pattern likelihood l_s (NOT log), scaled so that the true value is
 *                   l_s * 2^(-64*c_s)
 *   site_scaler_out optional [S] == the siteCorrections slice: c_s.  Only differences of c_s
 *                   between rate classes and sum_s f_s c_s are observable upstream
 *                   (likefunc2.cpp:736-770, 828-853, 1484-1506).
 *
 * Reproducibility: steady-state full passes run on chain schedules whose joins multiply the children of a node in the
 * order in which their workgroups arrive, and the schedule itself is chosen by timing (HYPHY_HIP_TUNE): the same inputs can
 * give log-likelihoods that differ in the last bits (<= a few 1e-16 relative) between runs and between ranks.  The reference's
 * CPU path is deterministic for a fixed thread count; a host that needs bit-identical repeats sets
 * HYPHY_HIP_TUNE=0 HYPHY_HIP_CUT=levels (fixed order, no joins; ~25 % slower at the headline size on the plain form).  r06: with
 * HYPHY_HIP_TUNE=0 alone a partition that runs class-compressed (subtree repeats; 49-64 states from ~128 tiles of 16 patterns per
 * shard) is bit-identical from run to run too, at the speed of the tuned form — its two walks have no join whose result depends on who
 * arrives first (tests/test_gpu_fullsize.py::test_class_compressed_form_without_tuner_is_bit_reproducible); smaller partitions
 * of that mode still run chain schedules: add HYPHY_HIP_CUT=levels there.
 */
int hyphy_hip_evaluate(hyphy_hip_partition *p, int64_t cat, const int64_t *update_nodes, int64_t n_update,
                       const int64_t *q_nodes, int64_t n_q, const double *q_dense, int q_is_probability,
                       const double *root_freqs, double *logl_out, double *site_lik_out,
                       int64_t *site_scaler_out);

/*
 * hyphy_hip_evaluate for models in the reference's "explicit form" (BUSTED, BS-REL, RELAX — the model is a
 * formula of matrix exponentials, `Model M = ("Exp(Q1)*w1+Exp(Q2)*w2...", freqs, EXPLICIT_FORM_MATRIX_EXPONENTIAL)`,
 * res/TemplateBatchFiles/libv3/models/codon/BS_REL.bf:34-60): the host's ExponentiateMatrices queues the Exp() arguments
 * of every branch (variablecontainer.cpp:208-233), exponentiates each and recombines them through the formula
 * (tree.cpp:3047-3090).  Here branch q_nodes[k] comes with n_components[k] numeric rate matrices (consecutive in
 * q_dense) and their mixture weights; the device exponentiates all of them in one launch and forms
 * P_k = sum_m weights[.] exp(Q[.]) directly in the layouts the pruning kernels read.
 */
int hyphy_hip_evaluate_mixture(hyphy_hip_partition *p, int64_t cat, const int64_t *update_nodes, int64_t n_update,
                               const int64_t *q_nodes, int64_t n_q, const int64_t *n_components /* [n_q] */,
                               const double *q_dense /* [sum n_components][D*D] */, const double *weights /* [sum n_components] */,
                               const double *root_freqs, double *logl_out, double *site_lik_out, int64_t *site_scaler_out);

/*
 * hyphy_hip_evaluate_mixture with the component rate matrices formed ON THE DEVICE (r04): BS-REL / BUSTED / RELAX components
 * differ in GLOBAL parameters only (omega_m: res/TemplateBatchFiles/libv3/models/codon/BS_REL.bf:48), so component m of branch b is
 * Q_(b,m) = sum_k x_(b,m),k T_k over the templates of hyphy_hip_set_q_templates / hyphy_hip_update_q_templates — the host's
 * serial pass that evaluates every component's formulas for every branch (calcnode.cpp:526-704 behind
 * variablecontainer.cpp:208-233) and the dense matrices over PCIe both go away.  Protocol: hyphy_hip_build_q(p, n, coeffs) with
 * n = sum of n_components rows of K coefficients, branch-major (the components of q_nodes[0], then those of q_nodes[1], ...),
 * then this call; the rows are consumed inside the exponential kernel (no rate matrix is ever materialised).  At most 16
 * components per branch.  Everything else as hyphy_hip_evaluate_mixture.
 */
int hyphy_hip_evaluate_mixture_built(hyphy_hip_partition *p, int64_t cat, const int64_t *update_nodes, int64_t n_update,
                                     const int64_t *q_nodes, int64_t n_q, const int64_t *n_components /* [n_q] */,
                                     const double *weights /* [sum n_components] */, const double *root_freqs, double *logl_out,
                                     double *site_lik_out, int64_t *site_scaler_out);

/*
 * The same evaluation split in two, so that the partitions of one likelihood function overlap (different devices, or
 * different streams of one device): enqueue every partition, then collect.  The reference's partition loop in
 * _LikelihoodFunction::Compute (src/core/likefunc.cpp:2524-2589) calls ComputeBlock(partID) serially and its MPI
 * partition mode (:2708-2768) farms the partitions out; the adapter's pre-pass does the former on
 * N devices.  evaluate_async copies the matrices to a pinned staging buffer (host arrays are free on return) and
 * returns as soon as everything is enqueued; collect waits and returns what hyphy_hip_evaluate returns.  Any other
 * evaluation entry point called in between finishes the pending evaluation first.
 */
int hyphy_hip_evaluate_async(hyphy_hip_partition *p, int64_t cat, const int64_t *update_nodes, int64_t n_update,
                             const int64_t *q_nodes, int64_t n_q, const double *q_dense, int q_is_probability,
                             const double *root_freqs);
int hyphy_hip_collect(hyphy_hip_partition *p, double *logl_out, double *site_lik_out, int64_t *site_scaler_out);

/*
 * The all-reduce of the partition log-likelihood over RCCL / xGMI, reachable from a C++ host (north_star: "a single RCCL
 * allreduce of the partition log-likelihood over xGMI per evaluation").  librccl.so is loaded on first use.
 *   one process per GPU:   rank 0: hyphy_hip_comm_unique_id(id) -> the host broadcasts the 128 bytes (MPI_Bcast in a
 *                          HYPHYMPI build, a file ...) -> every rank: hyphy_hip_comm_init_rank(p, id, rank, n) on the
 *                          partition that holds ITS pattern shard -> hyphy_hip_evaluate_allreduce(...) returns the
 *                          log-likelihood of the whole alignment on every rank (evaluate + one ncclAllReduce of one
 *                          double, in the partition's stream).  hyphy_hip_allreduce_device is the bare in-stream sum
 *                          for callers of hyphy_hip_evaluate_device.
 *   one process, N GPUs:   hyphy_hip_create(device_count = N) + hyphy_hip_comm_init_all(p): with HYPHY_HIP_COMBINE=rccl
 *                          hyphy_hip_evaluate / hyphy_hip_evaluate_built sum the shard partials by ONE group all-reduce
 *                          instead of on the host (default: host-side Neumaier combine, likefunc.cpp:11046-11093).
 */
int hyphy_hip_comm_unique_id(void *out128);
int hyphy_hip_comm_init_rank(hyphy_hip_partition *p, const void *unique_id, int rank, int n_ranks);
int hyphy_hip_comm_init_all(hyphy_hip_partition *p);
int hyphy_hip_allreduce_device(hyphy_hip_partition *p, double *d_value);
int hyphy_hip_evaluate_allreduce(hyphy_hip_partition *p, int64_t cat, const int64_t *update_nodes, int64_t n_update,
                                 const int64_t *q_nodes, int64_t n_q, const double *q_dense, int q_is_probability,
                                 const double *root_freqs, double *logl_out);
/* ... behind hyphy_hip_build_q (template models; the step `bench.py --gpus N` times): local construction + exponentials +
 * pruning + reduction, one ncclAllReduce of one double on the partition's stream, the sum back through the host-mapped
 * record.  A rank whose local evaluation fails still joins the collective (with NaN) and then returns its error: no rank
 * is left waiting.  hyphy_hip_last_allreduce_ms: event-timed duration of the last in-stream all-reduce while
 * hyphy_hip_set_timing_detail is on (else 0). */
int hyphy_hip_evaluate_built_allreduce(hyphy_hip_partition *p, int64_t cat, const int64_t *update_nodes, int64_t n_update,
                                       const int64_t *q_nodes, int64_t n_q, const double *root_freqs, double *logl_out);
double hyphy_hip_last_allreduce_ms(const hyphy_hip_partition *p);

/* The collective-free combine of one process per GPU on ONE node (r06): what the single-process form does with its shards — partials
 * back over PCIe into host-mapped records, the reference's Neumaier combine on the host (likefunc.cpp:11046-11093) — between processes,
 * through a POSIX shared-memory segment: each rank finishes its local evaluation like a one-GPU run, posts (value, epoch) into its slot and
 * reads the others' (one release store, N - 1 acquire loads; no device work).  Same bits on every rank (summed in rank order).
 *   hyphy_hip_comm_init_host            `name`: the same string on every rank, unique per run; returns when every rank has attached
 *   hyphy_hip_evaluate(_built)_exchange hyphy_hip_evaluate(_built) + one exchange; a rank whose local evaluation fails still posts (NaN)
 *                                       and then returns its own error: nobody is left waiting (HYPHY_HIP_EXCHANGE_TIMEOUT_S, default 120)
 *   hyphy_hip_xch_open / _sum / _close  the exchange alone (no partition, no device).
 * RCCL (above) stays the C-ABI's and the adapter's default collective; bench.py --gpus N times both and uses the faster (collective_ab). */
int hyphy_hip_comm_init_host(hyphy_hip_partition *p, const char *name, int rank, int n_ranks);
int hyphy_hip_evaluate_exchange(hyphy_hip_partition *p, int64_t cat, const int64_t *update_nodes, int64_t n_update, const int64_t *q_nodes,
                                int64_t n_q, const double *q_dense, int q_is_probability, const double *root_freqs, double *logl_out);
int hyphy_hip_evaluate_built_exchange(hyphy_hip_partition *p, int64_t cat, const int64_t *update_nodes, int64_t n_update,
                                      const int64_t *q_nodes, int64_t n_q, const double *root_freqs, double *logl_out);
int hyphy_hip_xch_open(const char *name, int rank, int n_ranks, void **handle_out);
int hyphy_hip_xch_sum(void *handle, double local, int local_failed, double *sum_out);
void hyphy_hip_xch_close(void *handle);

/*
 * Same evaluation with device-resident inputs/outputs, enqueued asynchronously on the
 * partition's stream (device_count must be 1): d_q is a DEVICE pointer [n_q*D*D];
 * d_logl_out a DEVICE pointer to one double that receives this shard's partial
 * log-likelihood (ready for an RCCL all-reduce across site-sharded ranks).  Call
 * hyphy_hip_synchronize() (or synchronise the stream) before reading it.
 */
int hyphy_hip_evaluate_device(hyphy_hip_partition *p, int64_t cat, const int64_t *update_nodes, int64_t n_update,
                              const int64_t *q_nodes, int64_t n_q, const double *d_q, int q_is_probability,
                              const double *root_freqs, double *d_logl_out);

/* Reads one double that lives in device memory and is produced by work queued on the partition's stream — the log-likelihood
 * summed over ranks by the caller's own collective (RCCL all-reduce behind hyphy_hip_evaluate_device, SURVEY 8e: the
 * reference's MPI ranks) — through the host-mapped result record: a one-thread kernel posts it, the host spins on the record's
 * sequence word.  Replaces a device-to-host copy + stream synchronisation per evaluation.  Single-device partitions. */
int hyphy_hip_fetch_device_scalar(hyphy_hip_partition *p, const double *d_value, double *value_out);

/*
 * Rate-category batch (config "BUSTED 3-rate-class"): evaluates ALL C classes in one
 * schedule (classes are an extra batch dimension on the device) and mixes them exactly as
 * PopulateConditionalProbabilities' weighted-sum mode + SumUpSiteLikelihoods do
 * (likefunc2.cpp:820-853, 1484-1506).  q_dense is [C][n_q][D*D]; weights [C].
 */
int hyphy_hip_evaluate_categories(hyphy_hip_partition *p, const int64_t *update_nodes, int64_t n_update,
                                  const int64_t *q_nodes, int64_t n_q, const double *q_dense,
                                  int q_is_probability, const double *weights, const double *root_freqs,
                                  double *logl_out, double *site_lik_out, int64_t *site_scaler_out);

/* The same from rate matrices staged by hyphy_hip_build_q (n = C*n_q coefficient rows, class-major). */
int hyphy_hip_evaluate_categories_built(hyphy_hip_partition *p, const int64_t *update_nodes, int64_t n_update,
                                        const int64_t *q_nodes, int64_t n_q, const double *weights,
                                        const double *root_freqs, double *logl_out);
/* ... with the mixed per-pattern values: site_lik_out [S] / site_scaler_out [S] as hyphy_hip_evaluate_categories returns them (what
 * PopulateConditionalProbabilities' weighted-sum mode leaves in its buffer and scalers, likefunc2.cpp:772-859); either may be NULL.
 * The adapter's category hook answers the host's whole class loop with this one call ("rate classes"). */
int hyphy_hip_evaluate_categories_built_sites(hyphy_hip_partition *p, const int64_t *update_nodes, int64_t n_update,
                                              const int64_t *q_nodes, int64_t n_q, const double *weights,
                                              const double *root_freqs, double *logl_out, double *site_lik_out,
                                              int64_t *site_scaler_out);

/*
 * Copies device partials back in the reference's host layout for code that reads the caches
 * directly (ReconstructAncestors likefunc2.cpp:416-449, FillInConditionals tree.cpp:3335-3371):
 *   inode_cache [I*S*D]  iNodeCache[(node*S + pattern)*D + state] (tree_evaluator.cpp:3608-3617)
 *   scaler_counts [I*S]  cumulative 2^64-exponent of the subtree below (node, pattern): the
 *                        stored vector times 2^(-64*count) is the unscaled conditional.
 * Cache policy: by default ("lazy"; environment HYPHY_HIP_CACHE=always disables it) a full pass that follows a
 * full pass — a sweep over a global parameter — does not store its conditionals in HBM; nothing can read them
 * before the next full pass overwrites them.  This call, a partial update and hyphy_hip_branch_cache_build
 * first re-run a storing pass when the resident copies are stale, so callers never observe the difference.
 */
int hyphy_hip_download_partials(hyphy_hip_partition *p, int64_t cat, double *inode_cache, int64_t *scaler_counts);

/*
 * Stand-alone batched matrix exponential: drop-in for the OpenMP loop in ExponentiateMatrices
 * (tree.cpp:3011-3037, each iteration = _Matrix::Exponentiate(1., true, storage)).
 * q_dense, p_out: host [n*D*D].  Rows of P sum to 1 (diag_populator, matrix.cpp:5837-5852).
 */
int hyphy_hip_expm_batch(int64_t D, int64_t n, const double *q_dense, double *p_out);

/*
 * Device-side rate-matrix construction for template models: every branch's
 * numeric Q is a linear combination of K fixed D x D templates,
 *       Q_b = sum_k coeff[b][k] * T_k   (off-diagonal),   Q_b[i][i] = -sum_{j != i} Q_b[i][j]
 * e.g. MG94xREV with fixed nucleotide biases: T_0 = synonymous part, T_1 = non-synonymous part,
 * coeff[b] = (t_b, omega * t_b).  Templates are uploaded once; per evaluation only the
 * n*K coefficients cross PCIe.  Replaces the serial RecomputeMatrix + MultByFreqs loop
 * (tree.cpp:2944-2969, matrix.cpp:1546-1677) for such models.  Output goes to the partition's
 * device Q buffer which hyphy_hip_evaluate_device() accepts as d_q
 * (hyphy_hip_q_buffer()).
 */
int hyphy_hip_set_q_templates(hyphy_hip_partition *p, int64_t K, const double *templates /* [K*D*D] */);
int hyphy_hip_build_q(hyphy_hip_partition *p, int64_t n, const double *coeffs /* host [n*K] */);
/* New VALUES for the K templates (K unchanged: no reallocation): for hosts whose templates depend on the global
 * parameters of the current evaluation — the HyPhy adapter derives M_k(globals) from K probe branches per
 * ExponentiateMatrices call and sends Q_b = sum_k x_bk M_k as K coefficients per branch.
 * The caller's array is copied before the call returns (pinned staging ring); the upload itself is queued on the partition's
 * stream ahead of the next exponential launch, the host does not wait for it. */
int hyphy_hip_update_q_templates(hyphy_hip_partition *p, int64_t K, const double *templates /* [K*D*D] */);
double *hyphy_hip_q_buffer(hyphy_hip_partition *p); /* device pointer, capacity (L+I-1)*C*D*D doubles */

/*
 * Per-site batched fits.  The FEL family of analyses gives every alignment site its own rate
 * multipliers — res/TemplateBatchFiles/SelectionAnalyses/FEL.bf:593-605 attaches the globals fel.alpha_scaler,
 * fel.beta_scaler_test and fel.beta_scaler_nuisance to the site model, and fel.handle_a_site (FEL.bf:609+) optimises
 * them for ONE site at a time with a single-site likelihood function: per evaluation (L+I-1) calls of
 * _Matrix::Exponentiate (matrix.cpp:5746+) and a one-pattern pruning pass (tree_evaluator.cpp:3556+), sites fanned
 * out over MPI (libv3/tasks/mpi.bf).  This entry point evaluates ALL S patterns of the partition, each under its own
 * multipliers, for n_sets candidate parameter vectors per pattern (e.g. the 12-point start.grid of FEL.bf:617-680, or
 * the simplex of a batched optimiser) in one launch:
 *
 *       Q_{b,s} = sum_k  site_mult[set][s][branch_group[b]][k] * branch_coeffs[b][k] * T_k       (off-diagonal)
 *       site_logl_out[set][s] = log L(pattern s | tree, {Q_{b,s}}_b, root_freqs)     (pattern frequency NOT applied)
 *
 * with the templates T_k of hyphy_hip_set_q_templates (K <= 4; e.g. MG94xREV: T_0 synonymous, T_1 non-synonymous,
 * branch_coeffs[b] = the branch's synonymous / non-synonymous lengths from the global fit, branch_group = 0 tested,
 * 1 nuisance, site_mult[s][g] = (alpha_s, beta_s^g)).  All multipliers, coefficients and off-diagonal template
 * entries must be >= 0.  The transition matrices are never formed: each pruning step applies exp(Q_{b,s}) to the
 * child's conditional vector by uniformisation on the FP64 matrix cores (sitefit.hip).  The result agrees with
 * "exponentiate, then multiply" to rounding (both are approximations of the same exact value; this one has
 * componentwise relative accuracy).  Cost per edge grows linearly with the site's total rate on that branch
 * (sum_k multiplier * coefficient * max_i |T_k[i][i]|); rates above 4096 are refused.  Returns 1 (unsupported) for
 * 4-state partitions, K > 4 and such rates.
 */
int hyphy_hip_site_fits_evaluate(hyphy_hip_partition *p, int64_t n_sets, int64_t n_groups,
                                 const int64_t *branch_group /* [L+I-1] */, const double *branch_coeffs /* [L+I-1][K] */,
                                 const double *site_mult /* [n_sets][S][n_groups][K] */, const double *root_freqs,
                                 double *site_logl_out /* [n_sets][S] */);
/*
 * The same with a branch-site MIXTURE per site (the "explicit form" models of SURVEY §3.4: MEME's two omega classes per
 * site, `res/TemplateBatchFiles/SelectionAnalyses/MEME.bf`; BS-REL, `libv3/models/codon/BS_REL.bf:34-60`, where
 * `tree.cpp:3047-3090` forms P = sum_m w_m exp(Q_m) on every branch): site s evolves on branch b with
 *       P_{b,s} = sum_m site_weights[set][s][m] * exp(Q^(m)_{b,s}),
 *       Q^(m)_{b,s} = sum_k site_mult[set][s][m][branch_group[b]][k] * branch_coeffs[b][k] * T_k.
 * n_mix <= 8; weights >= 0 (normally summing to 1).
 */
int hyphy_hip_site_fits_evaluate_mixture(hyphy_hip_partition *p, int64_t n_sets, int64_t n_groups, int64_t n_mix,
                                         const int64_t *branch_group, const double *branch_coeffs,
                                         const double *site_mult /* [n_sets][S][n_mix][n_groups][K] */,
                                         const double *site_weights /* [n_sets][S][n_mix] */, const double *root_freqs,
                                         double *site_logl_out /* [n_sets][S] */);
double hyphy_hip_site_fits_kernel_ms(const hyphy_hip_partition *p); /* duration of the last site-fit kernel */
/* Synchronous evaluation from the rate matrices staged by the last hyphy_hip_build_q() (n matrices,
 * in the order of q_nodes): template models never move a dense Q across PCIe — only the n*K
 * coefficients go down and one double comes back.  Same semantics as hyphy_hip_evaluate otherwise. */
int hyphy_hip_evaluate_built(hyphy_hip_partition *p, int64_t cat, const int64_t *update_nodes, int64_t n_update,
                             const int64_t *q_nodes, int64_t n_q, const double *root_freqs, double *logl_out);

/* hyphy_hip_evaluate_built + the per-pattern outputs of hyphy_hip_evaluate (storageVec / siteCorrections): what the reference's
 * category loop (likefunc.cpp:2851-2990) asks for once per rate class.  On a single-device partition the values and exponents are
 * written in the caller's pattern order into host-mapped memory by a kernel in front of the one that publishes the result record
 * (r05; HYPHY_HIP_SITE_EXPORT=0: two device-to-host copies and a host-side scatter, as hyphy_hip_evaluate did until r04). */
int hyphy_hip_evaluate_built_sites(hyphy_hip_partition *p, int64_t cat, const int64_t *update_nodes, int64_t n_update,
                                   const int64_t *q_nodes, int64_t n_q, const double *root_freqs, double *logl_out,
                                   double *site_lik_out, int64_t *site_scaler_out);

/* Blocks until all work enqueued for the partition has finished. */
int hyphy_hip_synchronize(hyphy_hip_partition *p);

/* The HIP stream (hipStream_t) work is enqueued on for shard 0 — for event timing. */
void *hyphy_hip_stream(hyphy_hip_partition *p);

/* Enqueue all further work of a single-device partition on the caller's stream (e.g. the
 * current PyTorch stream, so that an RCCL all-reduce of d_logl_out is ordered in-stream).
 * Any hipStream_t is accepted, including NULL (the legacy default stream);
 * HYPHY_HIP_OWN_STREAM restores the partition's own stream. */
#define HYPHY_HIP_OWN_STREAM ((void *)(intptr_t)-1)
int hyphy_hip_set_stream(hyphy_hip_partition *p, void *stream);

/* Per-call device timers of the last evaluation, milliseconds (SURVEY §5 tracing row):
 * out[0] = expm kernel(s), out[1] = pruning kernel, out[2] = root/site reduction.
 * out[1] is 0 when the last evaluation carried no kernel-duration stamp (one evaluation in HYPHY_HIP_TIMING_EVERY, default 16,
 * does; hyphy_hip_set_timing_detail(p, 1) stamps every evaluation) — it never reports an earlier evaluation's duration.
 * out[0] and out[2] are measured only while timing detail is on. */
int hyphy_hip_last_timings(hyphy_hip_partition *p, double out[3]);
/* out[0] and out[2] are only stamped while the detail switch is on (two more event records per evaluation; the
 * environment variable HYPHY_HIP_ALL_TIMINGS sets its initial state); out[1] comes from the ring below. */
int hyphy_hip_set_timing_detail(hyphy_hip_partition *p, int on);

/* ---- pinned node states ---------------------------------------------------------------------
 * ComputeBlock's branchIndex / branchValues ("setBranch": src/core/likefunc.cpp:10950-10957; leaf case
 * src/core/tree_evaluator.cpp:163-181, internal case :583-594 and :3624): every evaluation that follows sees
 * node `node` (node code: leaf l -> l, internal i -> L + i, the root included) fixed to states[pattern] in
 * [0, D).  node < 0 or states == NULL removes the pin.  The caller lists the node (and, after removing the pin,
 * lists it again) among update_nodes, as RecoverAncestralSequencesMarginal does with
 * AddBranchToForcedRecomputeList (src/core/likefunc2.cpp:932-1040). */
int hyphy_hip_set_pinned_states(hyphy_hip_partition *p, int64_t node, const int64_t *states /* [S] */);

/* ---- branch cache ---------------------------------------------------------------------
 * Replaces _TheTree::ComputeBranchCache (src/core/tree_evaluator.cpp:4286-4845) and
 * _TheTree::ComputeLLWithBranchCache (src/core/tree.cpp:3383-3936), driven by the policy code of
 * _LikelihoodFunction::ComputeBlock (src/core/likefunc.cpp:10886-10948, 11125-11258): while the optimiser
 * varies ONE branch length, every evaluation is a single [D x D] x [D x S] contraction.
 *
 * build: after an ordinary hyphy_hip_evaluate* call (conditionals and transition matrices resident, all
 * parameters at the values the line search keeps fixed), prepare the cache for branch `node` (node code:
 * leaf l -> l, internal i -> L + i; not the root).  Works for any model (the tree is re-rooted with
 * transposed transition matrices, no reversibility needed).  Returns 1 for the 4-state path.
 * evaluate: log-likelihood with this branch's matrix replaced by exp(q) (or q itself when
 * q_is_probability); all other parameters as at build time.  The device keeps the new matrix, as the host
 * tree does; the next ordinary evaluation of the class invalidates its cache.  One cache per rate class.
 * site_lik_out / site_scaler_out: per-pattern (l_s, c_s) as hyphy_hip_evaluate returns them. */
int hyphy_hip_branch_cache_build(hyphy_hip_partition *p, int64_t cat, int64_t node);
int hyphy_hip_branch_cache_evaluate(hyphy_hip_partition *p, int64_t cat, int64_t node, const double *q_dense,
                                    int q_is_probability, double *logl_out, double *site_lik_out /* [S] or NULL */,
                                    int64_t *site_scaler_out /* [S] or NULL */);

/* Durations (ms) of the pruning launches of the last `n` evaluations, oldest first, from a ring of HIP
 * event pairs recorded on the partition's stream (shard 0).  Nothing is queried while evaluations run —
 * call this after the timed region.  Returns the number of entries written (<= n, <= 1024).
 * "Evaluations" here are the STAMPED ones: the library stamps one evaluation in k (environment HYPHY_HIP_TIMING_EVERY=k,
 * default 16; the first evaluation of a partition is always stamped) — an event pair costs ~5 us of stream time. */
int64_t hyphy_hip_prune_timings(hyphy_hip_partition *p, double *out_ms, int64_t n);
/* Pruning-kernel launches per evaluation under the current schedule (forest scheduling cuts a full
 * evaluation into levels of subtree fragments, one launch per level; partial updates use one). */
int hyphy_hip_prune_launches(hyphy_hip_partition *p);
/* Name of the pruning kernel this partition's evaluations launch (chosen by shard size and state count;
 * as it appears in a rocprofv3 kernel trace).  Static string.  Class-compressed partitions (subtree repeats, below) run
 * "class_table_team_kernel" in front of it, and answer "trunk_walk_kernel" once their full passes run the trunk as one
 * row-split walk per tile (r06; first passes, partial updates and pinned states keep the pruning kernels). */
const char *hyphy_hip_prune_kernel_name(const hyphy_hip_partition *p);

/* What the schedule tuner measured for this partition ("" before it ran): on the first steady-state full pass the
 * library times the (idempotent) pruning pass under each candidate cut of the tree — level-peeled fragments, or chains
 * with source subtrees of at most m internal nodes — and keeps the fastest.  HYPHY_HIP_TUNE=0 disables it. */
/* Host-only planning helpers (no device is touched): decisions hyphy_hip_create takes from the topology / the leaf table alone,
 * exposed for inspection and for tests that run without a GPU.
 *   hyphy_hip_plan_reroot       the path (internal indices, given root first) to the `candidate`-th (0, 1) height-minimising
 *                               node the steady-state passes may be rooted at (re-rooted schedules); returns the
 *                               number of nodes on the path, 0 if there is no such candidate, < 0 on bad arguments.
 *   hyphy_hip_plan_pattern_order  the device-side pattern order (order_out[j] = caller's pattern stored j-th).
 *   hyphy_hip_plan_schedule     compiles the steady-state full-pass schedule of a 61-state partition with `ntiles` 16-pattern
 *                               tiles for `kernel` (0 workgroup per tile, 1 wave per tile, 2 row-split workgroups on chain
 *                               schedules) and cut `chain_m` (> 0: chain schedule with sources of at most chain_m nodes, -1:
 *                               level-peeled fragments, 0: the library's heuristic), optionally re-rooted, and decodes its join
 *                               table the way the kernels do.  info_out[8] = {chain schedule (0/1), programs, schedule entries,
 *                               largest matrix-image slot in the join table, most arrivals a node waits for, records that do not
 *                               decode to the topology (must be 0), re-rooted (0/1), trunk nodes}.  Trees the packed join table
 *                               cannot describe (> 65 535 internal nodes, image slots >= 2^15, > 255 internal children of one node)
 *                               get a schedule without chains. */
int64_t hyphy_hip_plan_reroot(int64_t L, int64_t I, const int64_t *flat_parents, int64_t candidate, int64_t *path_out, int64_t cap);
int hyphy_hip_plan_pattern_order(int64_t D, int64_t L, int64_t S, const int64_t *leaf_codes, int64_t *order_out);
int hyphy_hip_plan_schedule(int64_t L, int64_t I, const int64_t *flat_parents, int64_t kernel, int64_t chain_m, int64_t ntiles,
                            int64_t reroot, int64_t *info_out);

const char *hyphy_hip_schedule_info(const hyphy_hip_partition *p);

/* Subtree repeats — the device form of the reference's `tcc` traversal masks (src/core/tree.cpp:2801-2858 populates them,
 * src/core/likefunc.cpp:10854 passes them on every ComputeBlock, src/core/tree_evaluator.cpp:57-76, 240-256 skip a subtree
 * whose leaf states repeat the previous site's).  At hyphy_hip_create the library groups, per internal node, the patterns
 * whose leaves below the node carry the same states into CLASSES; subtrees with few classes are evaluated once per class
 * (class tables, one MFMA product per 16 classes) and enter the rest of the tree as generalised leaves.  Results are the same
 * with and without; partial updates, rate classes, shards and mixtures all run on the compressed form, pinned states and the
 * branch cache on the plain one.  49-64 states: on where a measurement on the first steady-state pass says it pays; 4 states: built
 * and tested but slower than the plain kernel, so only with HYPHY_HIP_REPEATS=1 / 2 in the environment.
 *   hyphy_hip_set_repeats   on = 0: this partition walks every node at every pattern; 1: back on (default where it pays;
 *                           HYPHY_HIP_REPEATS=0 in the environment turns it off for every partition created afterwards).
 *   hyphy_hip_repeat_stats  first shard: out[0] compression available, [1] class tables, [2] table rows (classes padded to
 *                           tiles of 16 = edge products of the lower phase), [3] / [4] internal nodes / leaves of the trunk,
 *                           [5] edge products of one full pass with repeats on, [6] ... off, [7] in use. */
int hyphy_hip_set_repeats(hyphy_hip_partition *p, int on);
/*   hyphy_hip_plan_repeats  host-only (no device): classes of every internal node over the S patterns as given (classes_out[I]), the
 *                           compressed set for `theta` (compressed_out[I]; theta <= 0: the default 0.35); returns the edge products
 *                           of a full pass with one table per compressed node, < 0 on bad arguments. */
int64_t hyphy_hip_plan_repeats(int64_t L, int64_t I, const int64_t *flat_parents, int64_t S, const int64_t *leaf_codes, double theta,
                               int64_t *classes_out, int64_t *compressed_out);
int hyphy_hip_repeat_stats(const hyphy_hip_partition *p, int64_t out[8]);
/* Host-only (no device): the program trunk_walk_kernel would run for a trunk given as a tree over generalised leaves — node codes
 * 0 .. L - 1 leaves, L + i internal node i (children before parents, the root last), parents[c] = node code of c's parent.
 * out: [nodes, inputs, stack depth, two chains?, first / end node of the one-chain form, of chain 0, of chain 1], then per walked node
 * (internal index, or -1: the chain's end at the root; leaf children; first input; flags 1: push the running product and start from
 * ones, 2: multiply the waiting product back in behind the edge product), then the inputs (leaf codes).  Returns the words written
 * (< 0: -words needed).  Restates nothing of the reference: it is the order in which src/core/tree_evaluator.cpp:3556-4171 visits the
 * nodes above the class tables, cut for one or two workgroups per tile. */
int64_t hyphy_hip_plan_trunk_walk(int64_t L, int64_t I, const int64_t *parents, int64_t *out, int64_t cap);

/* 4 states: schedules as run-time generated straight-line kernels (nucgen.hip; replaces the interpretation of the reference's
 * 4-state loop, src/core/tree_evaluator.cpp:2253-2273, 3556-4171, entry by entry).  A 4-state partition whose full-pass schedule
 * keeps coming back (HYPHY_HIP_NUCGEN_AFTER evaluations, default 8) has it compiled by hiprtc on a background thread and runs it
 * from the evaluation that finds it ready; code objects are shared per process by schedule.  HYPHY_HIP_NUCGEN=0: off (always the
 * interpreter); =2: compile synchronously at the first full pass.  hyphy_hip_prune_kernel_name() answers "nucgen_kernel" once the
 * last launch ran one.
 *   hyphy_hip_plan_nucgen   host-only (needs libhiprtc, no device): the source the library would compile for the steady-state full
 *                           pass of a 4-state partition over this tree (leaf_has_ambig[L] or NULL: leaves that carry ambiguity
 *                           codes; small != 0: the form for shards of at most two workgroups per CU — every matrix in LDS, the
 *                           evaluation's exponentials and the final combine inside the launch); returns its length (src_out, if given, receives at most cap - 1 bytes + NUL), 0 when the
 *                           generator does not cover the schedule, < 0 on bad arguments; *compiled_out = 1 when it compiled for gfx950. */
int64_t hyphy_hip_plan_nucgen(int64_t L, int64_t I, const int64_t *flat_parents, const int64_t *leaf_has_ambig, int64_t small, char *src_out,
                              int64_t cap, int64_t *compiled_out);

const char *hyphy_hip_last_error(void);
const char *hyphy_hip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* HYPHY_HIP_H */
