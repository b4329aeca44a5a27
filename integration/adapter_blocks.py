"""The host-adapter code that INTEGRATION.md describes, as text blocks that integration/build.py
splices into a COPY of the reference's src/core/likefunc.cpp (never into /root/reference, and the
patched copy is never committed).  Everything here is our own code; the anchors are short unique
strings of the reference file used only to locate the insertion points."""

# ---- block 1: helpers, after the last project #include ---------------------------------------------
HELPERS = r'''
#ifdef HYPHY_HIP
// ===== MI355X likelihood core: host adapter (see INTEGRATION.md) =====================================
#include "hyphy_hip.h"
#include <algorithm>
#include <map>
#include <unordered_map>
#include <cstdlib>
#include <string>
#include <unistd.h>
#include <sys/stat.h>
#include <time.h>
struct _HyHipPart {
  hyphy_hip_partition *part = nullptr;
  std::unordered_map<const void *, long> code_of;  // _CalcNode* -> node code (flatLeaves, then flatTree)
  std::vector<double> pbuf;
  std::vector<int64_t> qnodes;
  std::vector<char> cat_seen;
  // mode B (device exponentials, INTEGRATION.md): rate matrices the host queued for exponentiation, kept dense
  std::vector<std::vector<double>> qstash;   // per class: [B][D*D]
  std::vector<std::vector<char>> q_pending;  // ... not yet handed to the device
  std::vector<std::vector<char>> host_stale; // ... the node's host-side compExp does not reflect it yet
  std::vector<long> cat_arg;                 // the catID ComputeBlock used for the class (-1 without categories)
  long n_stale = 0;
  // asynchronous pre-pass of _LikelihoodFunction::Compute (all device partitions are enqueued before the first is collected)
  bool pending = false, pre_done = false;
  double pre_value = 0.;
  // template mode (SURVEY 8f-3 through the real host): Q_b = sum_k x_bk M_k(globals), x_b = the branch's independent local
  // parameters.  Per rate class: 0 not analysed yet, 1 enabled, -1 disabled (the model is not linear in its local parameters)
  std::vector<int> tmpl_state;
  // Branch classes (r04): branches that share the model, the template variables of their independent locals and the text of
  // their constraints (foreground / background omega: "fg.nonSynRate := omegaF*fg.synRate" vs "bg.nonSynRate := omegaB*...")
  // form a GROUP with templates of its own; a branch's coefficient row is its K_g locals at the group's columns and zeros
  // elsewhere, so the device sees ONE template model with tmpl_K = sum of the K_g columns.
  struct Group {
    long model = -1, K = 0, col = 0;
    _SimpleList refs;                        // template-variable references of the group's nodes (iVariables odd entries)
    std::string dep_sig;                     // constraints on dependent locals (see _hyhip_dep_signature)
  };
  std::vector<Group> tmpl_groups;
  bool tmpl_groups_open = false;             // the learning call may still add groups
  long tmpl_K = 0;                           // columns of a coefficient row = templates on the device
  std::unordered_map<const void *, int> group_of;   // node -> group (-1: conforms to none), cached
  std::vector<std::vector<double>> tmpl_M;   // per class: [K][D*D] current templates (off-diagonal entries are what counts)
  std::vector<std::vector<double>> tmpl_x;   // per class: [B][K] latest local-parameter rows
  std::vector<char> tmpl_uploaded;           // per class: tmpl_M is what the device holds
  long tmpl_device_cat = -1;                 // class whose templates were uploaded last
  struct Call {                              // one ExponentiateMatrices call in template mode
    bool active = false;
    long cat = 0;
    std::vector<std::vector<long>> probe;    // per group: node codes
    std::vector<long> verify;                // per group: node code or -1
    std::vector<double> verify_dist, skipped_dist;  // per group: relative distance of the verification row / of the farthest skipped row from the probes
    std::vector<long> skipped;               // node codes
  } call;
  long n_template_evals = 0, n_skipped = 0;
  // mixture mode (explicit-form models: P_b = sum_m w_m Exp(Q_bm), weights built from GLOBAL variables only): per rate
  // class 0 not analysed, 2 analysed — waiting for the verification against the host's own matrix, 1 enabled, -1 disabled
  std::vector<int> mix_state;
  long mix_M = 0, mix_probe = -1;
  std::vector<_Formula *> mix_wf;                 // the M weight expressions (the model formula with Exp(.) -> 0 / 1)
  std::vector<double> mix_probe_q;                // [M][D*D] rate matrices of the verification branch
  std::vector<std::vector<double>> mix_q;         // per class: [B][M][D*D]
  std::vector<std::vector<double>> mix_w;         // per class: [M] weights of the matrices stashed last
  long n_mixture_evals = 0;
  // mixture TEMPLATE mode (r04): the components of a BS-REL / BUSTED branch differ in global parameters only, so each is linear
  // in the branch's own locals, Q_(b,m) = sum_k x_bk T^(m)_k: per call the host evaluates the component matrices of K probe
  // branches + one verifier, every other branch reaches the device as its K locals (hyphy_hip_evaluate_mixture_built).
  // Per rate class: 0 not analysed, 1 enabled, -1 disabled.
  std::vector<int> mixT_state;
  Group mixT_group;                               // what every explicit-form node must show (model, template variables, constraints)
  std::unordered_map<const void *, char> mixT_ok;
  std::vector<std::vector<double>> mixT_T;        // per class: [M][K][D*D] component templates of the last call
  std::vector<std::vector<double>> mixT_x;        // per class: [B][K]
  struct MCall {
    bool active = false, broken = false;   // broken: a node outside the group turned up — this call goes the dense way
    bool learning = false;                 // the adapter's learning call: every explicit-form node is recomputed by the host and its
                                           // Exp() arguments are read behind it (`probe` lists them); the analysis runs at the end
    long cat = 0, verify = -1;
    long pending = -1;                     // recomputed node whose Exp() arguments are still to be read (behind its RecomputeMatrix)
    std::vector<long> probe, skipped;
    double verify_dist = 0., skipped_dist = 0.;
  } mcall;
  long n_mixT_evals = 0, n_mix_skipped = 0;
  // SPMD site sharding (one host process per GPU, every process runs the same batch file: HYPHY_HIP_WORLD / HYPHY_HIP_RANK,
  // set by the launcher, one ordinary host process per GPU): this process holds patterns [lo, hi) of the partition
  // and every evaluation ends in ONE ncclAllReduce of the partition log-likelihood (hyphy_hip_evaluate*_allreduce)
  bool spmd = false;
  long spmd_rank = 0, spmd_world = 1;
  bool spmd_host = false;   // HYPHY_HIP_COLLECTIVE=host: the ranks' partials through shared memory (hyphy_hip_comm_init_host), no RCCL
};
static std::map<const void *, std::vector<_HyHipPart>> _hyhip_lfs;
// Rate classes batched through the host's category loop (r06; INTEGRATION.md "rate classes").  PopulateConditionalProbabilities'
// weighted-sum mode (likefunc2.cpp:484-908, :772-859) calls ComputeBlock once per class and mixes after each call; with a batch
// open (_hyphy_hip_cat_begin) those calls only COLLECT what each class would have sent to the device, the host's mixing of a
// collected class is skipped (_hyphy_hip_cat_collected), and _hyphy_hip_cat_end answers the whole loop with ONE
// hyphy_hip_evaluate_categories(_built_sites) call — per-pattern mixed values and exponents into the host's buffer / scalers, or,
// when _LikelihoodFunction::Compute asked (_hyphy_hip_cat_want), just the summed log-likelihood (SumUpSiteLikelihoods is skipped).
struct _HyHipCatBatch {
  bool active = false;
  const void *lf = nullptr;
  long index = -1, C = 0, S = 0;
  _TheTree *tree = nullptr;
  hyFloat *buffer = nullptr;
  long *scalers = nullptr;
  bool mixed_any = false;                 // the adapter has mixed a class into buffer / scalers itself (classes evaluated one by one)
  std::vector<double> weights;
  struct ClassCall {
    bool have = false, taken = false;     // have: inputs stashed; taken: the adapter answers for this class (stashed or mixed already)
    int kind = 0;                         // 1 coefficient rows (template mode), 2 dense rate matrices, 3 transition matrices of the host, 4 nothing changed
    long catID = 0;
    std::vector<int64_t> upd, qn;
    std::vector<double> data;
  };
  std::vector<ClassCall> calls;
  // what Compute asked for / got
  const void *want_lf = nullptr;
  long want_index = -1;
  bool logl_valid = false;
  double logl = 0.;
  long n_batched = 0, n_single = 0;
};
static _HyHipCatBatch _hyhip_cb;
static bool _hyhip_cat_block = false;  // this ComputeBlock call builds / uses a branch cache behind its evaluation: it must really run
static std::map<const void *, std::pair<const void *, long>> _hyhip_tree_owner;  // _TheTree* -> (lf, partition index)
long _hyhip_calls = 0L, _hyhip_cached_calls = 0L, _hyhip_deferred = 0L;
static double _hyhip_compute_seconds = 0.;  // wall time inside _hyphy_hip_compute (coefficients, library call incl. its wait, downloads)
// > 0: ExponentiateMatrices hands the queued rate matrices to the adapter instead of exponentiating them on the host.
// On while Optimize runs (nothing but ComputeBlock reads the transition matrices there; they are brought up to date
// on the host when it returns); HYPHY_HIP_DEVICE_EXPM=0 keeps mode A, =always forces it (LFCompute benchmarks only:
// ancestral reconstruction and simulation read the host matrices between evaluations).
static int _hyhip_defer_depth = 0;
extern bool (*_hyhip_defer_expm_hook)(_TheTree *, long, _List &, _List &, _SimpleList &, _SimpleList &, bool);  // tree.cpp copy
extern bool (*_hyhip_skip_recompute_hook)(_TheTree *, long, _CalcNode *, unsigned long, unsigned long);  // tree.cpp copy
extern bool _hyhip_force_defer_call;  // tree.cpp copy: offer this ExponentiateMatrices call to the adapter whatever its queue holds
static int _hyhip_async_phase = 0;  // > 0: ComputeBlock enqueues the device evaluation and returns (pre-pass of Compute)
static int _hyphy_hip_expm_mode(void) {
  static int mode = -1;
  if (mode < 0) {
    const char *v = getenv("HYPHY_HIP_DEVICE_EXPM");
    mode = !v ? 1 : (!strcmp(v, "0") ? 0 : (!strcmp(v, "always") ? 2 : 1));
  }
  return mode;
}

static bool _hyphy_hip_enabled(void) {
  static int state = -1;
  if (state < 0) {
    const char *v = getenv("HYPHY_HIP");
    state = (v && atoi(v) > 0 && hyphy_hip_device_count() > 0) ? 1 : 0;
  }
  return state == 1;
}

static void _hyphy_hip_teardown(const void *lf) {
  auto it = _hyhip_lfs.find(lf);
  if (it == _hyhip_lfs.end()) return;
  for (auto &hp : it->second) {
    if (hp.part && getenv("HYPHY_HIP_VERBOSE") && hp.n_mixture_evals)
      fprintf(stderr, "[hyphy_hip] mixture mode: %ld evaluations exponentiated and mixed their %ld-component branch-site mixtures on the device\n",
              hp.n_mixture_evals, hp.mix_M);
    if (hp.part && getenv("HYPHY_HIP_VERBOSE") && hp.n_mixT_evals)
      fprintf(stderr, "[hyphy_hip] mixture template mode: %ld evaluations took their component rate matrices as coefficients (M = %ld, K = %ld), %ld RecomputeMatrix calls skipped\n",
              hp.n_mixT_evals, hp.mix_M, hp.mixT_group.K, hp.n_mix_skipped);
    if (hp.part && getenv("HYPHY_HIP_VERBOSE") && hp.n_template_evals)
      fprintf(stderr, "[hyphy_hip] template mode: %ld evaluations took their rate matrices as coefficients (K = %ld), %ld RecomputeMatrix calls skipped\n",
              hp.n_template_evals, hp.tmpl_K, hp.n_skipped);
    hyphy_hip_destroy(hp.part);
  }
  if (_hyhip_cb.lf == lf) {
    if (getenv("HYPHY_HIP_VERBOSE") && (_hyhip_cb.n_batched || _hyhip_cb.n_single))
      fprintf(stderr, "[hyphy_hip] rate classes: %ld category loops answered by ONE device evaluation of all classes, %ld classes evaluated one by one\n",
              _hyhip_cb.n_batched, _hyhip_cb.n_single);
    _hyhip_cb = _HyHipCatBatch();
  }
  _hyhip_lfs.erase(it);
  for (auto o = _hyhip_tree_owner.begin(); o != _hyhip_tree_owner.end();)
    o = o->second.first == lf ? _hyhip_tree_owner.erase(o) : std::next(o);
  if (getenv("HYPHY_HIP_VERBOSE")) fprintf(stderr, "[hyphy_hip] %ld ComputeBlock evaluations ran on the device so far (+ %ld through the branch cache); %ld matrix exponentials moved to the device; adapter mode: %.3f s of wall clock inside the adapter's ComputeBlock calls (coefficients, library call and its wait, per-pattern downloads)\n", _hyhip_calls, _hyhip_cached_calls, _hyhip_deferred, _hyhip_compute_seconds);
}

static bool _hyphy_hip_defer_handler(_TheTree *t, long catID, _List &nodesToDo, _List &matrixQueue, _SimpleList &parallel,
                                     _SimpleList &isExplicitForm, bool hasExpForm);
static bool _hyphy_hip_skip_handler(_TheTree *t, long catID, _CalcNode *node, unsigned long nodeID, unsigned long n_nodes);

static void _hyphy_hip_setup(const void *lf, unsigned long i, unsigned long n_parts, _TheTree *cT,
                             _DataSetFilter const *theFilter, long const *leaf_codes, _Vector *ambigs) {
  if (!_hyphy_hip_enabled()) return;
  auto &v = _hyhip_lfs[lf];
  if (v.size() < n_parts) v.resize(n_parts);
  _HyHipPart &hp = v[i];
  hyphy_hip_destroy(hp.part);
  hp.part = nullptr;
  const long D = theFilter->GetDimension(), S = theFilter->GetPatternCount(), L = cT->GetLeafCount(),
             I = cT->GetINodeCount();
  if (L < 2 || I < 1) return;
  const _SimpleList &fp = cT->flatParents;  // (upstream: a public accessor, see INTEGRATION.md)
  std::vector<int64_t> parents(L + I), freq(S), codes((size_t)L * S);
  for (long k = 0; k < L + I; k++) parents[k] = fp.list_data[k];
  for (long s = 0; s < S; s++) freq[s] = theFilter->theFrequencies.get(s);
  for (size_t k = 0; k < (size_t)L * S; k++) codes[k] = leaf_codes[k];
  const long n_amb = (long)ambigs->get_used() / D;
  // Size policy: a 4-state partition of a few patterns is evaluated faster by the host than a kernel can be launched
  // (61-state partitions are worth it from ONE pattern on: the host spends ~10 ms per evaluation exponentiating the
  // 2L-3 rate matrices of a 64-taxon tree, the device 25 us — profiles/r03_adapter_rate.jsonl).  HYPHY_HIP_MIN_PATTERNS overrides.
  {
    const char *mp = getenv("HYPHY_HIP_MIN_PATTERNS");
    const long min_patterns = mp ? atol(mp) : (D <= 4 ? 64L : 1L);
    if (S < min_patterns) {
      if (getenv("HYPHY_HIP_VERBOSE"))
        fprintf(stderr, "[hyphy_hip] partition %lu: %ld patterns < HYPHY_HIP_MIN_PATTERNS = %ld -> host path\n", i, S, min_patterns);
      return;
    }
  }
  // SPMD site sharding: this process keeps patterns [lo, hi)
  long spmd_world = 1, spmd_rank = 0;
  bool spmd = false;
  if (const char *w = getenv("HYPHY_HIP_WORLD")) {
    spmd_world = atol(w);
    spmd_rank = getenv("HYPHY_HIP_RANK") ? atol(getenv("HYPHY_HIP_RANK")) : 0;
    spmd = spmd_world >= 1 && spmd_rank >= 0 && spmd_rank < spmd_world;
  }
  // (A HYPHYMPI build is NOT a way into this mode: its ranks > 0 sit in the master/slave loop of src/mains/unix.cpp:1037-1042
  //  and never run the batch file, so they would never reach this rendezvous.  SPMD here means: the launcher starts one
  //  ordinary process per GPU with HYPHY_HIP_WORLD / HYPHY_HIP_RANK set.)
  if (spmd && (n_parts > 1 || cT->categoryCount > 1)) spmd = false;  // (one partition, one rate class: the site-sharded config)
  if (spmd && S < spmd_world) {  // fewer patterns than ranks: EVERY rank takes the host path (S and the world are the same everywhere),
    if (getenv("HYPHY_HIP_VERBOSE"))  // before anybody enters a collective
      fprintf(stderr, "[hyphy_hip] partition %lu: %ld patterns < %ld ranks -> host path on every rank\n", i, S, spmd_world);
    return;
  }
  if (spmd) {
    const long lo = S * spmd_rank / spmd_world, hi = S * (spmd_rank + 1) / spmd_world, Sl = hi - lo;
    std::vector<int64_t> c2((size_t)L * Sl), f2(Sl);
    for (long l = 0; l < L; l++)
      for (long k = 0; k < Sl; k++) c2[(size_t)l * Sl + k] = codes[(size_t)l * S + lo + k];
    for (long k = 0; k < Sl; k++) f2[k] = freq[lo + k];
    codes.swap(c2);
    freq.swap(f2);
  }
  const long S_dev = (long)freq.size();
  // Partition -> device(s).  HYPHY_HIP_DEVICES=a-b (or HYPHY_HIP_DEVICE=a, or every visible device): a likelihood
  // function with several partitions puts partition i on device a + i mod n (config "multi-partition -> N GPUs",
  // the evaluations are enqueued together by the pre-pass in Compute); a single partition is site-sharded over
  // the whole range (hyphy_hip_create's device_count, config "site-sharded across N GPUs").
  int dev_first = 0, dev_n = 1;
  {
    const int ndev = hyphy_hip_device_count();
    if (const char *dr = getenv("HYPHY_HIP_DEVICES")) {
      int a = 0, b = -1;
      if (sscanf(dr, "%d-%d", &a, &b) < 2) b = a;
      if (a < 0) a = 0;
      if (b >= ndev) b = ndev - 1;
      if (b < a) b = a;
      dev_first = a;
      dev_n = b - a + 1;
    } else if (const char *dv = getenv("HYPHY_HIP_DEVICE")) {
      dev_first = atoi(dv);
    } else if (n_parts > 1) {
      dev_n = ndev > 0 ? ndev : 1;
    }
  }
  int my_first = n_parts > 1 ? dev_first + (int)(i % (unsigned long)dev_n) : dev_first;
  int my_count = n_parts > 1 ? 1 : dev_n;
  if (spmd) {  // one device per process: local rank -> device (HYPHY_HIP_LOCAL_RANK, else the rank itself), within the device range
    const long local = getenv("HYPHY_HIP_LOCAL_RANK") ? atol(getenv("HYPHY_HIP_LOCAL_RANK")) : spmd_rank;
    const int ndev_all = hyphy_hip_device_count();
    const int span = (getenv("HYPHY_HIP_DEVICES") || getenv("HYPHY_HIP_DEVICE")) ? dev_n : (ndev_all > 0 ? ndev_all : 1);
    my_first = dev_first + (int)(local % span);
    my_count = 1;
  }
  int rc = hyphy_hip_create(&hp.part, D, S_dev, L, I, cT->categoryCount, parents.data(), codes.data(),
                            n_amb ? ambigs->theData : nullptr, n_amb, freq.data(), my_first, my_count);
  const bool spmd_host = spmd && getenv("HYPHY_HIP_COLLECTIVE") && !strcmp(getenv("HYPHY_HIP_COLLECTIVE"), "host");
  if (rc == 0 && spmd_host) {
    // r06: the collective-free combine of one node — every rank attaches to a shared-memory segment named after the run
    // (HYPHY_HIP_RUN_ID, the same on every rank, new for every run) and the n-th SPMD set-up of this process
    static long host_generation = 0;
    const char *run_id = getenv("HYPHY_HIP_RUN_ID");
    const std::string name = std::string(run_id ? run_id : "run") + "_" + std::to_string(host_generation++);
    if (hyphy_hip_comm_init_host(hp.part, name.c_str(), (int)spmd_rank, (int)spmd_world) != 0) {
      ReportWarning(_String("hyphy_hip: SPMD site sharding could not attach to the host exchange (") & hyphy_hip_last_error() & "); host path");
      hyphy_hip_destroy(hp.part);
      hp.part = nullptr;
      return;
    }
  }
  if (rc == 0 && spmd && !spmd_host) {
    // the RCCL unique id: made by rank 0, handed to the other ranks through a file
    // (HYPHY_HIP_UID_FILE) that rank 0 writes and the others wait for
    // One rendezvous PER COMMUNICATOR: every process runs the same batch file, so the n-th SPMD set-up of this process pairs
    // with the n-th of every other rank — the id travels in "<HYPHY_HIP_UID_FILE>.<n>".  Rank 0 removes a left-over of that
    // name before it writes (atomic rename); the others accept only a file that is not older than their own process (a file
    // from an earlier run of the same command is never taken for this run's id).
    char uid[128];
    bool have = false;
    static long spmd_generation = 0;
    static const time_t proc_start = time(nullptr);
    const long gen = spmd_generation++;
    {
      const char *base = getenv("HYPHY_HIP_UID_FILE");
      // HYPHY_HIP_RUN_ID (set by the launcher, the same on every rank, new for every run) goes into the file's name: a rank then
      // cannot take an earlier run's id for this run's, however far apart the ranks start.  Without it the file is only accepted when
      // it is not older than this process (minus 2 s) — ranks that start more than that after rank 0 wrote it would wait in vain.
      const char *run_id = getenv("HYPHY_HIP_RUN_ID");
      const std::string path = base ? std::string(base) + (run_id ? std::string(".") + run_id : std::string()) + "." + std::to_string(gen) : std::string();
      if (spmd_world == 1) {
        have = hyphy_hip_comm_unique_id(uid) == 0;
      } else if (base && spmd_rank == 0) {
        have = hyphy_hip_comm_unique_id(uid) == 0;
        unlink(path.c_str());
        const std::string tmp = path + ".tmp";
        FILE *fh = have ? fopen(tmp.c_str(), "wb") : nullptr;
        if (fh) {
          have = fwrite(uid, 1, 128, fh) == 128;
          fclose(fh);
          if (have) have = rename(tmp.c_str(), path.c_str()) == 0;
        } else have = false;
      } else if (base) {
        for (int tries = 0; tries < 6000 && !have; tries++) {  // (up to ~10 minutes: rank 0 may still be parsing its batch file)
          struct stat st;
          if (stat(path.c_str(), &st) == 0 && (run_id || st.st_mtime + 2 >= proc_start)) {
            FILE *fh = fopen(path.c_str(), "rb");
            if (fh) {
              have = fread(uid, 1, 128, fh) == 128;
              fclose(fh);
            }
          }
          if (!have) usleep(100000);
        }
      }
    }
    if (!have || hyphy_hip_comm_init_rank(hp.part, uid, (int)spmd_rank, (int)spmd_world) != 0) {
      ReportWarning(_String("hyphy_hip: SPMD site sharding could not set up its communicator (") & hyphy_hip_last_error() & "); host path");
      hyphy_hip_destroy(hp.part);
      hp.part = nullptr;
      return;
    }
  }
  if (rc == 0 && getenv("HYPHY_HIP_VERBOSE"))
    fprintf(stderr, "[hyphy_hip] partition %lu of %lu -> device %d%s (%ld states, %ld patterns, %ld leaves)%s\n", i, n_parts, my_first,
            my_count > 1 ? " (+ site shards on the following devices)" : "", D, S_dev, L, spmd ? (spmd_host ? " [SPMD site shard, host-side exchange per evaluation]" : " [SPMD site shard, RCCL all-reduce per evaluation]") : "");
  if (rc != 0) {  // > 0: unsupported here -> the CPU path keeps working; < 0: report and use the CPU path
    hp.part = nullptr;
    ReportWarning(_String("hyphy_hip_create: ") & hyphy_hip_last_error());
    return;
  }
  hp.code_of.clear();
  for (long code = 0; code < L + I; code++) hp.code_of[cT->GetNodeFromFlatIndex(code)] = code;
  const long n_cat = cT->categoryCount > 0 ? cT->categoryCount : 1;
  hp.cat_seen.assign(n_cat, 0);
  hp.qstash.assign(n_cat, std::vector<double>());
  hp.q_pending.assign(n_cat, std::vector<char>(L + I, 0));
  hp.host_stale.assign(n_cat, std::vector<char>(L + I, 0));
  hp.cat_arg.assign(n_cat, -1L);
  hp.n_stale = 0;
  hp.pending = hp.pre_done = false;
  hp.spmd = spmd;
  hp.spmd_host = spmd_host;
  hp.spmd_rank = spmd_rank;
  hp.spmd_world = spmd_world;
  hp.tmpl_state.assign(n_cat, getenv("HYPHY_HIP_TEMPLATES") && !strcmp(getenv("HYPHY_HIP_TEMPLATES"), "0") ? -1 : 0);
  hp.tmpl_K = 0;
  hp.tmpl_groups.clear();
  hp.tmpl_groups_open = false;
  hp.group_of.clear();
  hp.tmpl_M.assign(n_cat, std::vector<double>());
  hp.tmpl_x.assign(n_cat, std::vector<double>());
  hp.tmpl_uploaded.assign(n_cat, 0);
  hp.tmpl_device_cat = -1;
  hp.call = _HyHipPart::Call();
  hp.mix_state.assign(n_cat, (spmd || (getenv("HYPHY_HIP_MIXTURES") && !strcmp(getenv("HYPHY_HIP_MIXTURES"), "0"))) ? -1 : 0);
  hp.mix_M = 0;
  hp.mix_probe = -1;
  for (_Formula *f : hp.mix_wf) delete f;
  hp.mix_wf.clear();
  hp.mix_q.assign(n_cat, std::vector<double>());
  hp.mix_w.assign(n_cat, std::vector<double>());
  hp.mixT_state.assign(n_cat, (getenv("HYPHY_HIP_TEMPLATES") && !strcmp(getenv("HYPHY_HIP_TEMPLATES"), "0")) ? -1 : 0);
  hp.mixT_group = _HyHipPart::Group();
  hp.mixT_ok.clear();
  hp.mixT_T.assign(n_cat, std::vector<double>());
  hp.mixT_x.assign(n_cat, std::vector<double>());
  hp.mcall = _HyHipPart::MCall();
  _hyhip_tree_owner[cT] = std::make_pair(lf, (long)i);
  if (_hyphy_hip_expm_mode() > 0) {
    _hyhip_defer_expm_hook = _hyphy_hip_defer_handler;
    _hyhip_skip_recompute_hook = _hyphy_hip_skip_handler;
    if (_hyphy_hip_expm_mode() == 2 && _hyhip_defer_depth == 0) _hyhip_defer_depth = 1;
  }
}

static bool _hyphy_hip_active(const void *lf, long index) {
  auto it = _hyhip_lfs.find(lf);
  return it != _hyhip_lfs.end() && index < (long)it->second.size() && it->second[index].part != nullptr;
}

// ---- template mode (mode B only): Q_b = sum_k x_bk M_k(globals) ------------------------------------------------
// For the usual models every entry of the rate matrix is (a function of the global parameters) x (ONE independent
// local parameter of the branch) — MG94xREV: t; local MG94: synonymous / non-synonymous rate — i.e. the numeric
// matrix _CalcNode::RecomputeMatrix evaluates entry by entry (calcnode.cpp:526-704, ~526 formulas per codon branch,
// serial loop tree.cpp:2944-2969) is LINEAR and homogeneous in the branch's local parameters x_b.  The adapter checks
// that numerically (never symbolically) and then lets only K + 1 branches per ExponentiateMatrices call go through
// RecomputeMatrix: K probes give the templates M_k of THIS evaluation, one more verifies them (1e-11 relative), all the
// others are skipped and reach the device as K coefficients (hyphy_hip_build_q).  A failed verification recomputes
// the skipped matrices the normal way and switches the mode off for the rate class.
// Constrained (dependent) local parameters — "givenTree.N.nonSynRate := R*givenTree.N.synRate", the usual way a global omega
// enters a codon model — are fine as long as EVERY branch carries the same constraints: the text of each constraint with the
// branch's own name prefix removed, plus the template variable it binds to.  With identical constraints the dependents are the
// same function of (the branch's independent locals, the globals) on every branch, so the numerical linearity test in the
// independent locals covers them; branch-specific constraints (foreground / background omega classes) give different
// signatures and keep the dense path.
static std::string _hyhip_dep_signature(_CalcNode *n) {
  std::string sig;
  if (!n->dVariables || !n->dVariables->lLength) return sig;
  const std::string prefix = std::string(n->GetName()->get_str()) + ".";
  for (unsigned long k = 0; k + 1 < n->dVariables->lLength; k += 2) {
    _Variable *v = LocateVar(n->dVariables->list_data[k]);
    if (!v) return "?";
    _String *fs = v->GetFormulaString(kFormulaStringConversionNormal);
    std::string text(fs->get_str());
    DeleteObject(fs);
    for (size_t pos = text.find(prefix); pos != std::string::npos; pos = text.find(prefix, pos)) text.erase(pos, prefix.size());
    sig += std::to_string(n->dVariables->list_data[k + 1]) + "=" + text + ";";
  }
  return sig;
}
static bool _hyhip_invert(const double *rows, long K, double *inv);
// group of a node (cached): same model, same template variables behind its independent locals, same constraint texts.
// While the learning call is open a node that fits no group founds one (at most 4 groups, 8 columns in all).
static int _hyhip_group(_HyHipPart &hp, _CalcNode *n) {
  auto it = hp.group_of.find(n);
  if (it != hp.group_of.end()) return it->second;
  int g = -1;
  if (!n->HasExplicitFormModel() && n->iVariables && n->iVariables->lLength >= 2 && n->iVariables->lLength <= 6) {
    const long K = (long)n->iVariables->lLength / 2, model = n->GetModelIndex();
    const std::string sig = (n->dVariables && n->dVariables->lLength) ? _hyhip_dep_signature(n) : std::string();
    for (size_t k = 0; k < hp.tmpl_groups.size() && g < 0; k++) {
      const _HyHipPart::Group &G = hp.tmpl_groups[k];
      if (G.model != model || G.K != K || G.dep_sig != sig) continue;
      bool same = true;
      for (long j = 0; j < K && same; j++) same = n->iVariables->list_data[2 * j + 1] == G.refs.list_data[j];
      if (same) g = (int)k;
    }
    if (g < 0 && hp.tmpl_groups_open && sig != "?" && hp.tmpl_groups.size() < 4 && hp.tmpl_K + K <= 8) {
      _HyHipPart::Group G;
      G.model = model;
      G.K = K;
      G.col = hp.tmpl_K;
      G.dep_sig = sig;
      for (long j = 0; j < K; j++) G.refs << n->iVariables->list_data[2 * j + 1];
      hp.tmpl_groups.push_back(G);
      hp.tmpl_K += K;
      g = (int)hp.tmpl_groups.size() - 1;
    }
  }
  if (!hp.tmpl_groups_open) hp.group_of.emplace(n, g);   // (cached once the set of groups is final)
  return g;
}
// the branch's coefficient row: its independent locals at its group's columns, zeros elsewhere
static void _hyhip_local_row(const _HyHipPart &hp, _CalcNode *n, int g, double *x) {
  const _HyHipPart::Group &G = hp.tmpl_groups[g];
  for (long k = 0; k < hp.tmpl_K; k++) x[k] = 0.;
  for (long k = 0; k < G.K; k++) x[G.col + k] = LocateVar(n->iVariables->list_data[2 * k])->Compute()->Value();
}
// are the rows `codes` (+ optionally `extra`) of X independent in the columns of group G?  (Gram determinant, K <= 3)
static bool _hyhip_rows_independent(const std::vector<double> &X, long Ktot, const _HyHipPart::Group &G, const std::vector<long> &codes,
                                    long extra) {
  double rows[16], gram[16], inv[16];
  long n = 0;
  for (long c : codes) {
    for (long k = 0; k < G.K; k++) rows[n * G.K + k] = X[(size_t)c * Ktot + G.col + k];
    n++;
  }
  if (extra >= 0) {
    for (long k = 0; k < G.K; k++) rows[n * G.K + k] = X[(size_t)extra * Ktot + G.col + k];
    n++;
  }
  if (n > G.K) return false;
  for (long i = 0; i < n; i++)
    for (long j = 0; j < n; j++) {
      double gg = 0.;
      for (long k = 0; k < G.K; k++) gg += rows[i * G.K + k] * rows[j * G.K + k];
      gram[i * n + j] = gg;
    }
  return _hyhip_invert(gram, n, inv);
}
// rows [r][K] -> inverse of the K x K matrix (Gauss-Jordan, partial pivoting); false if (numerically) singular
static bool _hyhip_invert(const double *rows, long K, double *inv) {
  double a[16], b[16];
  for (long i = 0; i < K; i++)
    for (long j = 0; j < K; j++) {
      a[i * K + j] = rows[i * K + j];
      b[i * K + j] = i == j ? 1. : 0.;
    }
  double scale = 0.;
  for (long i = 0; i < K * K; i++) scale = fmax(scale, fabs(a[i]));
  if (!(scale > 0.)) return false;
  for (long c = 0; c < K; c++) {
    long piv = c;
    for (long r = c + 1; r < K; r++)
      if (fabs(a[r * K + c]) > fabs(a[piv * K + c])) piv = r;
    if (fabs(a[piv * K + c]) < 1e-8 * scale) return false;
    for (long j = 0; j < K; j++) {
      std::swap(a[c * K + j], a[piv * K + j]);
      std::swap(b[c * K + j], b[piv * K + j]);
    }
    const double d = 1. / a[c * K + c];
    for (long j = 0; j < K; j++) {
      a[c * K + j] *= d;
      b[c * K + j] *= d;
    }
    for (long r = 0; r < K; r++)
      if (r != c) {
        const double f = a[r * K + c];
        for (long j = 0; j < K; j++) {
          a[r * K + j] -= f * a[c * K + j];
          b[r * K + j] -= f * b[c * K + j];
        }
      }
  }
  for (long i = 0; i < K * K; i++) inv[i] = b[i];
  return true;
}
static void _hyhip_dense_copy(_Matrix *m, long DD, double *dst) {
  if (m->is_dense()) {
    memcpy(dst, m->theData, sizeof(double) * DD);
  } else {
    memset(dst, 0, sizeof(double) * DD);
    for (long k = 0; k < m->lDim; k++) {
      const long idx = m->theIndex[k];
      if (idx >= 0 && idx < DD) dst[idx] = m->theData[k];
    }
  }
}
// Q_b = sum_k x_bk M_k with the diagonal set to -(row sum), as the numeric matrix the host would have queued
static void _hyhip_template_dense(const _HyHipPart &hp, long cat, long code, long D, double *dst) {
  const long K = hp.tmpl_K, DD = D * D;
  const double *x = hp.tmpl_x[cat].data() + (size_t)code * K, *M = hp.tmpl_M[cat].data();
  for (long e = 0; e < DD; e++) {
    double v = 0.;
    for (long k = 0; k < K; k++) v += x[k] * M[k * DD + e];
    dst[e] = v;
  }
  for (long i = 0; i < D; i++) {
    double d = 0.;
    for (long j = 0; j < D; j++)
      if (j != i) d -= dst[i * D + j];
    dst[i * D + i] = d;
  }
}
// templates from probes (rows X_p, matrices Q_p): M = X_p^-1 Q_p; returns false when the probes are dependent
static bool _hyhip_solve_templates(long K, long DD, const double *Xp, const std::vector<const double *> &Qp, std::vector<double> &M) {
  double inv[16];
  if (!_hyhip_invert(Xp, K, inv)) return false;
  M.assign((size_t)K * DD, 0.);
  for (long k = 0; k < K; k++)
    for (long i = 0; i < K; i++) {
      const double f = inv[k * K + i];
      if (f == 0.) continue;
      const double *q = Qp[i];
      double *m = M.data() + (size_t)k * DD;
      for (long e = 0; e < DD; e++) m[e] += f * q[e];
    }
  return true;
}
// the same for ONE group: rows [G.col, G.col + G.K) of M (Ktot x DD) from the group's probes
static bool _hyhip_solve_group(const std::vector<double> &X, long Ktot, long DD, const _HyHipPart::Group &G, const std::vector<long> &probe,
                               const double *qstash, std::vector<double> &M) {
  if ((long)probe.size() != G.K) return false;
  double Xp[16], inv[16];
  for (long i = 0; i < G.K; i++)
    for (long k = 0; k < G.K; k++) Xp[i * G.K + k] = X[(size_t)probe[i] * Ktot + G.col + k];
  if (!_hyhip_invert(Xp, G.K, inv)) return false;
  if (M.size() != (size_t)Ktot * DD) M.assign((size_t)Ktot * DD, 0.);
  for (long k = 0; k < G.K; k++) {
    double *m = M.data() + (size_t)(G.col + k) * DD;
    for (long e = 0; e < DD; e++) m[e] = 0.;
    for (long i = 0; i < G.K; i++) {
      const double f = inv[k * G.K + i];
      if (f == 0.) continue;
      const double *q = qstash + (size_t)probe[i] * DD;
      for (long e = 0; e < DD; e++) m[e] += f * q[e];
    }
  }
  return true;
}
static double _hyhip_template_error(long K, long D, const double *x, const std::vector<double> &M, const double *q) {
  const long DD = D * D;
  double err = 0., scale = 0.;
  for (long e = 0; e < DD; e++) {
    if (e / D == e % D) continue;  // (the diagonal follows from the row)
    double v = 0.;
    for (long k = 0; k < K; k++) v += x[k] * M[(size_t)k * DD + e];
    err = fmax(err, fabs(v - q[e]));
    scale = fmax(scale, fabs(q[e]));
  }
  return scale > 0. ? err / scale : (err > 0. ? 1. : 0.);
}

static _HyHipPart *_hyhip_part_of_tree(_TheTree *t) {
  auto own = _hyhip_tree_owner.find(t);
  if (own == _hyhip_tree_owner.end()) return nullptr;
  auto it = _hyhip_lfs.find(own->second.first);
  if (it == _hyhip_lfs.end() || own->second.second >= (long)it->second.size()) return nullptr;
  _HyHipPart &hp = it->second[own->second.second];
  return hp.part ? &hp : nullptr;
}

// explicit-form (mixture) nodes: does the node show the signature of the partition's mixture group?  (cached)
static bool _hyhip_mix_conforms(_HyHipPart &hp, _CalcNode *n) {
  auto it = hp.mixT_ok.find(n);
  if (it != hp.mixT_ok.end()) return it->second != 0;
  const _HyHipPart::Group &G = hp.mixT_group;
  bool ok = n->HasExplicitFormModel() && n->GetModelIndex() == G.model && n->iVariables && (long)n->iVariables->lLength == 2 * G.K;
  for (long k = 0; k < G.K && ok; k++) ok = n->iVariables->list_data[2 * k + 1] == G.refs.list_data[k];
  if (ok) ok = ((n->dVariables && n->dVariables->lLength) ? _hyhip_dep_signature(n) : std::string()) == G.dep_sig;
  hp.mixT_ok.emplace(n, ok ? 1 : 0);
  return ok;
}
static void _hyhip_mix_row(_CalcNode *n, long K, double *x) {
  for (long k = 0; k < K; k++) x[k] = LocateVar(n->iVariables->list_data[2 * k])->Compute()->Value();
}
// rows `codes` (+ `extra`) of X [.][K] independent?
static bool _hyhip_mix_independent(const std::vector<double> &X, long K, const std::vector<long> &codes, long extra) {
  _HyHipPart::Group G;
  G.K = K;
  G.col = 0;
  return _hyhip_rows_independent(X, K, G, codes, extra);
}
// component templates from the probes: T[m][k] = sum_i inv(X_p)[k][i] Q_(probe_i, m); q_of(code, m) -> the stashed matrix
template <typename QOf>
static bool _hyhip_mix_solve(const std::vector<double> &X, long K, long M, long DD, const std::vector<long> &probe, QOf q_of, std::vector<double> &T) {
  if ((long)probe.size() != K) return false;
  double Xp[16], inv[16];
  for (long i = 0; i < K; i++)
    for (long k = 0; k < K; k++) Xp[i * K + k] = X[(size_t)probe[i] * K + k];
  if (!_hyhip_invert(Xp, K, inv)) return false;
  T.assign((size_t)M * K * DD, 0.);
  for (long m = 0; m < M; m++)
    for (long k = 0; k < K; k++) {
      double *dst = T.data() + ((size_t)m * K + k) * DD;
      for (long i = 0; i < K; i++) {
        const double f = inv[k * K + i];
        if (f == 0.) continue;
        const double *q = q_of(probe[i], m);
        for (long e = 0; e < DD; e++) dst[e] += f * q[e];
      }
    }
  return true;
}
// worst relative deviation of one branch's M stashed component matrices from sum_k x_k T[m][k] (off-diagonal entries)
template <typename QOf>
static double _hyhip_mix_error(long K, long M, long D, const double *x, const std::vector<double> &T, long code, QOf q_of) {
  const long DD = D * D;
  double worst = 0.;
  for (long m = 0; m < M; m++) {
    const double *q = q_of(code, m);
    double err = 0., scale = 0.;
    for (long e = 0; e < DD; e++) {
      if (e / D == e % D) continue;
      double v = 0.;
      for (long k = 0; k < K; k++) v += x[k] * T[((size_t)m * K + k) * DD + e];
      err = fmax(err, fabs(v - q[e]));
      scale = fmax(scale, fabs(q[e]));
    }
    worst = fmax(worst, scale > 0. ? err / scale : (err > 0. ? 1. : 0.));
  }
  return worst;
}

// The M component rate matrices of the node whose RecomputeMatrix ran last (its local parameters are still in the model's template
// variables): the arguments of the formula's Exp() terms, evaluated the way _Formula::ExtractMatrixExpArguments does
// (formula.cpp:2153-2214) but WITHOUT its cache comparison — the reference queues only the arguments that changed since the
// formula's last evaluation, the adapter needs all of them for its probes whatever changed.
static bool _hyhip_eval_exp_args(_Formula *f, long M, long D, double *dst) {
  if (!f) return false;
  long count = 0;
  for (unsigned long i = 0UL; i + 1UL < f->theFormula.countitems(); i++) {
    _Operation *this_op = f->GetIthTerm(i), *next_op = f->GetIthTerm(i + 1UL);
    if (!next_op->CanResultsBeCached(this_op, true)) continue;
    _Stack temp;
    this_op->Execute(temp);
    _Matrix *arg = (_Matrix *)temp.Pop(false);
    if (!arg || arg->ObjectClass() != MATRIX || count >= M) return false;
    _Matrix *num = (_Matrix *)arg->ComputeNumeric();
    if (!num || !num->is_numeric() || num->GetHDim() != D || num->GetVDim() != D || !num->theData) return false;
    _hyhip_dense_copy(num, D * D, dst + (size_t)count * D * D);
    count++;
    i++;
  }
  return count == M;
}
static void _hyhip_mix_collect_pending(_HyHipPart &hp, _TheTree *t, long catID) {
  if (hp.mcall.pending < 0) return;
  const long cat = catID < 0 ? 0 : catID, D = t->GetCodeBase(), DD = D * D, M = hp.mix_M, code = hp.mcall.pending;
  hp.mcall.pending = -1;
  _CalcNode *nd = (_CalcNode *)t->GetNodeFromFlatIndex(code);
  if (hp.mix_q[cat].empty()) hp.mix_q[cat].assign(hp.code_of.size() * (size_t)M * DD, 0.);
  if (!_hyhip_eval_exp_args(nd->GetExplicitFormModel(nd->map_global_to_local_category(catID)), M, D,
                            hp.mix_q[cat].data() + (size_t)code * M * DD))
    hp.mcall.broken = true;
}

// called for every node of ExponentiateMatrices' first loop (tree.cpp copy): true = do not call RecomputeMatrix
static bool _hyphy_hip_skip_handler(_TheTree *t, long catID, _CalcNode *node, unsigned long nodeID, unsigned long n_nodes) {
  if (_hyhip_defer_depth <= 0) return false;
  _HyHipPart *php = _hyhip_part_of_tree(t);
  if (!php) return false;
  _HyHipPart &hp = *php;
  const long cat = catID < 0 ? 0 : catID;
  if (nodeID == 0) {
    hp.mcall = _HyHipPart::MCall();
    hp.mcall.cat = cat;
    hp.mcall.active = cat < (long)hp.mix_state.size() && hp.mix_state[cat] == 1 && hp.mixT_state[cat] == 1 &&
                      (long)n_nodes >= hp.mixT_group.K + 6;
    // (the analysis needs every branch's components once; the reference queues only the Exp() arguments that changed since a
    //  formula's last evaluation, so a call that happens to queue all of them may never come: the adapter reads them itself)
    hp.mcall.learning = !hp.mcall.active && cat < (long)hp.mix_state.size() && hp.mix_state[cat] == 1 && hp.mixT_state[cat] == 0 &&
                        hp.mix_M > 0 && (long)n_nodes >= 12;
    _hyhip_force_defer_call = false;
  }
  if ((hp.mcall.active || hp.mcall.learning) && hp.mcall.cat == cat) _hyhip_mix_collect_pending(hp, t, catID);  // (the node recomputed just before this one)
  if (node->HasExplicitFormModel() && hp.mcall.learning && hp.mcall.cat == cat) {
    auto itl = hp.code_of.find(node);
    if (itl == hp.code_of.end()) {
      hp.mcall.broken = true;
      return false;
    }
    hp.mcall.probe.push_back(itl->second);
    hp.mcall.pending = itl->second;
    _hyhip_force_defer_call = true;
    return false;
  }
  if (node->HasExplicitFormModel()) {  // mixture template mode: K probes + one verifier go through RecomputeMatrix
    if (!hp.mcall.active || hp.mcall.cat != cat || hp.mcall.broken) return false;
    auto itm = hp.code_of.find(node);
    if (itm == hp.code_of.end() || !_hyhip_mix_conforms(hp, node)) {
      hp.mcall.broken = true;  // (a branch outside the group: the defer handler recomputes what was skipped so far)
      return false;
    }
    const long code = itm->second, K = hp.mixT_group.K;
    std::vector<double> &X = hp.mixT_x[cat];
    if (X.empty()) X.assign(hp.code_of.size() * K, 0.);
    _hyhip_mix_row(node, K, X.data() + (size_t)code * K);
    _hyhip_force_defer_call = true;
    if ((long)hp.mcall.probe.size() < K) {
      if (_hyhip_mix_independent(X, K, hp.mcall.probe, code)) {
        hp.mcall.probe.push_back(code);
        hp.mcall.pending = code;
      } else {
        hp.mcall.broken = true;  // (dependent on the probes so far and no basis yet: the ordinary way for this call)
      }
      return false;
    }
    double dist = 1e300;
    for (long pc : hp.mcall.probe) {
      double d = 0.;
      for (long k = 0; k < K; k++) {
        const double a = X[(size_t)pc * K + k], b = X[(size_t)code * K + k];
        d = fmax(d, a == b ? 0. : fabs(a - b) / fmax(fabs(a), fabs(b)));
      }
      dist = fmin(dist, d);
    }
    if (dist > 0. && hp.mcall.verify < 0) {
      hp.mcall.verify = code;
      hp.mcall.verify_dist = dist;
      hp.mcall.pending = code;
      return false;
    }
    hp.mcall.skipped_dist = fmax(hp.mcall.skipped_dist, dist);
    hp.mcall.skipped.push_back(code);
    hp.n_mix_skipped++;
    return true;
  }
  if (nodeID == 0) {
    hp.call = _HyHipPart::Call();
    hp.call.cat = cat;
    const size_t ng = hp.tmpl_groups.size();
    hp.call.active = cat < (long)hp.tmpl_state.size() && hp.tmpl_state[cat] == 1 && (long)n_nodes >= hp.tmpl_K + (long)ng + 5;
    hp.call.probe.assign(ng, std::vector<long>());
    hp.call.verify.assign(ng, -1L);
    hp.call.verify_dist.assign(ng, 0.);
    hp.call.skipped_dist.assign(ng, 0.);
  }
  if (!hp.call.active || hp.call.cat != cat) return false;
  auto itc = hp.code_of.find(node);
  if (itc == hp.code_of.end()) return false;
  const int g = _hyhip_group(hp, node);
  if (g < 0) return false;  // (goes the dense way)
  const _HyHipPart::Group &G = hp.tmpl_groups[g];
  const long code = itc->second, K = hp.tmpl_K;
  std::vector<double> &X = hp.tmpl_x[cat];
  if (X.empty()) X.assign(hp.code_of.size() * K, 0.);
  _hyhip_local_row(hp, node, g, X.data() + (size_t)code * K);
  std::vector<long> &probe = hp.call.probe[g];
  if ((long)probe.size() < G.K) {  // a probe of its group, if independent of the group's probes so far
    if (_hyhip_rows_independent(X, K, G, probe, code)) probe.push_back(code);
    return false;  // (dependent and no basis yet: the dense way)
  }
  // A branch whose local parameters EQUAL a probe's has the probe's rate matrix (same group: same model, same constraints,
  // same globals): skipping it assumes nothing about linearity.  The first branch of the group that differs goes through
  // RecomputeMatrix and verifies the group's templates; how far it sits from the probes says how much that verification is worth
  // for the others (r02 ADVICE: with every branch length equal the check used to pass vacuously).
  double dist = 1e300;
  for (long pc : probe) {
    double d = 0.;
    for (long k = 0; k < G.K; k++) {
      const double a = X[(size_t)pc * K + G.col + k], b = X[(size_t)code * K + G.col + k];
      d = fmax(d, a == b ? 0. : fabs(a - b) / fmax(fabs(a), fabs(b)));
    }
    dist = fmin(dist, d);
  }
  if (dist > 0. && hp.call.verify[g] < 0) {
    hp.call.verify[g] = code;
    hp.call.verify_dist[g] = dist;
    return false;
  }
  hp.call.skipped_dist[g] = fmax(hp.call.skipped_dist[g], dist);
  hp.call.skipped.push_back(code);
  hp.n_skipped++;
  return true;
}

// ---- mixture mode (mode B, explicit-form models) -----------------------------------------------------------------------
// The model is a formula of matrix exponentials (BUSTED, BS-REL, RELAX: "Exp(Q1)*w1+Exp(Q2)*w2...").  The host queues the
// Exp() arguments of every branch, exponentiates them and evaluates the formula (tree.cpp:3011-3090).  When the formula is a
// weighted SUM of the exponentials with weights that involve global variables only, the adapter takes the queue instead:
// the weights are the model formula with Exp(.) replaced by 0 / 1 (parsed once by the host's own parser, evaluated per
// call), and hyphy_hip_evaluate_mixture exponentiates and mixes on the device.  Accepted only after ONE branch's matrix
// computed by the host's own formula agrees with sum_m w_m exp(Q_m) to 1e-12 (first evaluation).
static bool _hyhip_build_weight_formulas(_HyHipPart &hp, _Formula *f, long M) {
  if (!f) return false;
  for (unsigned long i = 0; i < f->theFormula.lLength; i++) {  // everything the formula names directly must be global
    _Operation *op = (_Operation *)f->theFormula(i);
    long v = op->theData;
    if (v < -1) v = -v - 2;
    if (v >= 0) {
      _Variable *var = LocateVar(v);
      if (!var) return false;
      if (var->ObjectClass() != MATRIX && !var->IsGlobal()) return false;  // a branch parameter among the weights
    }
  }
  _String *fs = (_String *)f->toStr(kFormulaStringConversionNormal);
  std::string text(fs->get_str());
  DeleteObject(fs);
  std::vector<std::pair<size_t, size_t>> spans;  // [begin, end) of every Exp(...) term
  for (size_t pos = text.find("Exp("); pos != std::string::npos; pos = text.find("Exp(", pos)) {
    if (pos > 0 && (isalnum((unsigned char)text[pos - 1]) || text[pos - 1] == '_' || text[pos - 1] == '.')) {
      pos += 4;
      continue;
    }
    size_t q = pos + 4;
    int depth = 1;
    while (q < text.size() && depth > 0) {
      if (text[q] == '(') depth++;
      else if (text[q] == ')') depth--;
      q++;
    }
    if (depth != 0) return false;
    spans.push_back(std::make_pair(pos, q));
    pos = q;
  }
  if ((long)spans.size() != M) return false;
  for (_Formula *old_f : hp.mix_wf) delete old_f;
  hp.mix_wf.clear();
  for (long m0 = 0; m0 < M; m0++) {
    std::string w;
    size_t at = 0;
    for (long m = 0; m < M; m++) {
      w += text.substr(at, spans[m].first - at);
      w += m == m0 ? "(1)" : "(0)";
      at = spans[m].second;
    }
    w += text.substr(at);
    _String ws(w.c_str());
    _Formula *wf = new _Formula(ws, nil);
    if (wf->IsEmpty()) {
      delete wf;
      return false;
    }
    hp.mix_wf.push_back(wf);
  }
  return true;
}
static bool _hyhip_mixture_weights(_HyHipPart &hp, std::vector<double> &w) {
  w.resize(hp.mix_wf.size());
  for (size_t m = 0; m < hp.mix_wf.size(); m++) {
    HBLObjectRef r = hp.mix_wf[m]->Compute();
    if (!r || r->ObjectClass() != NUMBER) return false;
    w[m] = r->Value();
    if (!(w[m] == w[m])) return false;
  }
  return true;
}
// (HYPHY_HIP_DEBUG=1 says which test made the mixture hand-over decline a call)
#define _HYHIP_DECLINE(n)                                                                                  \
  do {                                                                                                     \
    if (getenv("HYPHY_HIP_DEBUG")) fprintf(stderr, "[hyphy_hip] defer_mixture: declined at #%d\n", (n)); \
    return false;                                                                                          \
  } while (0)
static void _hyhip_mix_learn(_HyHipPart &hp, _TheTree *t, long catID, const std::vector<long> &codes);
static bool _hyhip_defer_mixture(_HyHipPart &hp, _TheTree *t, long catID, _List &nodesToDo, _List &matrixQueue,
                                 _SimpleList &parallel, _SimpleList &isExplicitForm) {
  const long D = t->GetCodeBase(), DD = D * D, cat = catID < 0 ? 0 : catID;
  if (cat >= (long)hp.mix_state.size() || hp.mix_state[cat] < 0 || hp.mix_state[cat] == 2) _HYHIP_DECLINE(1);
  const long M = isExplicitForm.list_data[parallel.get(0)];
  if (M < 1 || M > 16 || parallel.lLength % M || (hp.mix_M && hp.mix_M != M)) _HYHIP_DECLINE(2);
  for (unsigned long g = 0; g < parallel.lLength; g += M) {  // groups of M consecutive queue entries, one node each
    const void *nd = nodesToDo(parallel.get(g));
    if (hp.code_of.find(nd) == hp.code_of.end()) _HYHIP_DECLINE(3);
    for (long m = 0; m < M; m++) {
      const long mid = parallel.get(g + m);
      _Matrix *mx = (_Matrix *)matrixQueue(mid);
      if (nodesToDo(mid) != nd || isExplicitForm.list_data[mid] != M || !mx || !mx->is_numeric() || mx->GetHDim() != D ||
          mx->GetVDim() != D || !mx->theData)
        _HYHIP_DECLINE(4);
    }
  }
  if (hp.mix_state[cat] == 0) {
    // first sight: build the weight expressions, remember one branch for the verification, let the host do this call
    _CalcNode *first = (_CalcNode *)nodesToDo(parallel.get(0));
    hp.mix_state[cat] = -1;
    if (hp.mix_wf.empty() || hp.mix_M != M) {
      if (!_hyhip_build_weight_formulas(hp, first->GetExplicitFormModel(first->map_global_to_local_category(catID)), M)) _HYHIP_DECLINE(5);
      hp.mix_M = M;
    }
    hp.mix_probe = hp.code_of.at(first);
    hp.mix_probe_q.assign((size_t)M * DD, 0.);
    for (long m = 0; m < M; m++) _hyhip_dense_copy((_Matrix *)matrixQueue(parallel.get(m)), DD, hp.mix_probe_q.data() + (size_t)m * DD);
    hp.mix_state[cat] = 2;
    _HYHIP_DECLINE(6);
  }
  // enabled: take the whole queue
  if (!_hyhip_mixture_weights(hp, hp.mix_w[cat])) _HYHIP_DECLINE(7);
  if (hp.mix_q[cat].empty()) hp.mix_q[cat].assign(hp.code_of.size() * (size_t)M * DD, 0.);
  hp.cat_arg[cat] = catID;
  auto mark = [&](long code, char how) {  // how: 3 dense component matrices in mix_q, 4 a row of locals in mixT_x
    hp.q_pending[cat][code] = how;
    if (!hp.host_stale[cat][code]) hp.n_stale++;
    hp.host_stale[cat][code] = 3;         // (either way the host's own formula brings the node up to date: _hyphy_hip_flush_part)
  };
  auto q_of = [&](long code, long m) -> const double * { return hp.mix_q[cat].data() + ((size_t)code * M + m) * DD; };
  std::vector<long> codes;
  for (unsigned long g = 0; g < parallel.lLength; g += M) {
    const long code = hp.code_of.at(nodesToDo(parallel.get(g)));
    for (long m = 0; m < M; m++)
      _hyhip_dense_copy((_Matrix *)matrixQueue(parallel.get(g + m)), DD, hp.mix_q[cat].data() + ((size_t)code * M + m) * DD);
    mark(code, 3);
    codes.push_back(code);
  }
  _hyhip_deferred += parallel.lLength;
  if (hp.mixT_state[cat] == 0 && (long)codes.size() >= 12) _hyhip_mix_learn(hp, t, catID, codes);
  return true;
}
// first evaluation after the analysis: does sum_m w_m exp(Q_m) reproduce the matrix the host's formula produced?
static void _hyhip_verify_mixture(_HyHipPart &hp, _TheTree *t, long catID) {
  const long D = t->GetCodeBase(), DD = D * D, cat = catID < 0 ? 0 : catID, M = hp.mix_M;
  hp.mix_state[cat] = -1;
  _Matrix *P = ((_CalcNode *)t->GetNodeFromFlatIndex(hp.mix_probe))->GetCompExp(catID);
  std::vector<double> w, E((size_t)M * DD), host(DD);
  if (!P || !P->theData || !_hyhip_mixture_weights(hp, w) || hyphy_hip_expm_batch(D, M, hp.mix_probe_q.data(), E.data()) != 0) return;
  _hyhip_dense_copy(P, DD, host.data());
  double err = 0.;
  for (long e = 0; e < DD; e++) {
    double v = 0.;
    for (long m = 0; m < M; m++) v += w[m] * E[(size_t)m * DD + e];
    err = fmax(err, fabs(v - host[e]));
  }
  hp.mix_state[cat] = err < 1e-12 ? 1 : -1;
  if (getenv("HYPHY_HIP_VERBOSE"))
    fprintf(stderr, "[hyphy_hip] explicit-form model, class %ld: %ld components, |host matrix - sum w exp(Q)| = %.2e -> %s\n", cat, M, err,
            hp.mix_state[cat] == 1 ? "mixture mode (device exponentials + mixing)" : "host exponentials");
}

// is every component of the explicit-form model linear in the branches' locals?  `codes`: branches whose M component rate
// matrices are in mix_q (a call in which every one of them was queued, or the adapter's own learning call)
static void _hyhip_mix_learn(_HyHipPart &hp, _TheTree *t, long catID, const std::vector<long> &codes) {
  const long D = t->GetCodeBase(), DD = D * D, cat = catID < 0 ? 0 : catID, M = hp.mix_M;
  auto q_of = [&](long code, long m) -> const double * { return hp.mix_q[cat].data() + ((size_t)code * M + m) * DD; };
    // first large call in mixture mode: is every component linear in the branches' locals?  (all component matrices are here)
    _CalcNode *first = (_CalcNode *)t->GetNodeFromFlatIndex(codes[0]);
    int verdict = -1;
    const long K0 = first->iVariables ? (long)first->iVariables->lLength / 2 : 0;
    if (K0 >= 1 && K0 <= 3 && (hp.mixT_group.K == 0 || hp.mixT_group.K == K0)) {
      if (hp.mixT_group.K == 0) {
        hp.mixT_group.K = K0;
        hp.mixT_group.model = first->GetModelIndex();
        hp.mixT_group.refs.Clear();
        for (long k = 0; k < K0; k++) hp.mixT_group.refs << first->iVariables->list_data[2 * k + 1];
        hp.mixT_group.dep_sig = (first->dVariables && first->dVariables->lLength) ? _hyhip_dep_signature(first) : std::string();
        hp.mixT_ok.clear();
      }
      bool all_conform = hp.mixT_group.dep_sig != "?";
      for (size_t g = 0; g < codes.size() && all_conform; g++) all_conform = _hyhip_mix_conforms(hp, (_CalcNode *)t->GetNodeFromFlatIndex(codes[g]));
      if (all_conform) {
        std::vector<double> &X = hp.mixT_x[cat];
        X.assign(hp.code_of.size() * K0, 0.);
        for (size_t g = 0; g < codes.size(); g++)
          _hyhip_mix_row((_CalcNode *)t->GetNodeFromFlatIndex(codes[g]), K0, X.data() + (size_t)codes[g] * K0);
        std::vector<long> probe;
        for (long code : codes)
          if ((long)probe.size() < K0 && _hyhip_mix_independent(X, K0, probe, code)) probe.push_back(code);
        std::vector<double> T;
        if (_hyhip_mix_solve(X, K0, M, DD, probe, q_of, T)) {
          double worst = 0.;
          for (long code : codes) worst = fmax(worst, _hyhip_mix_error(K0, M, D, X.data() + (size_t)code * K0, T, code, q_of));
          verdict = worst < 1e-11 ? 1 : -1;
          if (getenv("HYPHY_HIP_VERBOSE"))
            fprintf(stderr, "[hyphy_hip] mixture template analysis, class %ld: %ld components x %ld local parameter(s), %ld branches, worst relative deviation from linearity %.2e -> %s\n",
                    cat, M, K0, (long)codes.size(), worst, verdict == 1 ? "mixture template mode" : "dense component matrices");
        } else verdict = 0;  // (dependent probes: try again next time)
      }
    }
    hp.mixT_state[cat] = verdict;
}
// end of an ExponentiateMatrices call in mixture template mode: the K probes and the verifier went through RecomputeMatrix (their
// component matrices were read behind it, _hyhip_mix_collect_pending), every other branch was skipped.  true: every dirty branch
// is now a row of locals over this call's component templates (nothing left for the host); false: the host goes on normally
// (the skipped branches have been recomputed by the host's own formula first).
static bool _hyhip_finish_mixture_call(_HyHipPart &hp, _TheTree *t, long catID) {
  const long D = t->GetCodeBase(), DD = D * D, cat = catID < 0 ? 0 : catID, M = hp.mix_M, K = hp.mixT_group.K;
  _hyhip_mix_collect_pending(hp, t, catID);
  const _HyHipPart::MCall call = hp.mcall;
  hp.mcall = _HyHipPart::MCall();
  if (call.learning) {  // the analysis over every branch of this call; the host goes on with its own queue
    if (!call.broken && call.cat == cat && (long)call.probe.size() >= 12) _hyhip_mix_learn(hp, t, catID, call.probe);
    return false;
  }
  auto q_of = [&](long code, long m) -> const double * { return hp.mix_q[cat].data() + ((size_t)code * M + m) * DD; };
  bool ok = !call.broken && call.cat == cat && (long)call.probe.size() == K;
  const bool weak = call.verify >= 0 && call.skipped_dist > 0. && call.verify_dist < 0.01 && call.skipped_dist > 4. * call.verify_dist;
  std::vector<double> T;
  bool nonlinear = false;
  if (ok) {
    ok = _hyhip_mix_solve(hp.mixT_x[cat], K, M, DD, call.probe, q_of, T);
    if (ok && call.verify >= 0 && _hyhip_mix_error(K, M, D, hp.mixT_x[cat].data() + (size_t)call.verify * K, T, call.verify, q_of) >= 1e-11)
      ok = false, nonlinear = true;
  }
  if (ok) ok = _hyhip_mixture_weights(hp, hp.mix_w[cat]);
  if (ok && !weak) {
    hp.mixT_T[cat].swap(T);
    hp.cat_arg[cat] = catID;
    auto mark = [&](long code) {
      hp.q_pending[cat][code] = 4;
      if (!hp.host_stale[cat][code]) hp.n_stale++;
      hp.host_stale[cat][code] = 3;
    };
    for (long code : call.probe) mark(code);
    if (call.verify >= 0) mark(call.verify);
    for (long code : call.skipped) mark(code);
    _hyhip_deferred += (long)(call.probe.size() + call.skipped.size() + (call.verify >= 0 ? 1 : 0)) * M;
    return true;
  }
  if (nonlinear) {
    hp.mixT_state[cat] = -1;
    ReportWarning("hyphy_hip: the components of the explicit-form model are not linear in the branch parameters; mixture template mode switched off");
  }
  for (long code : call.skipped) {  // the skipped branches by the host's own formula (exponentials and recombination on the host)
    ((_CalcNode *)t->GetNodeFromFlatIndex(code))->RecomputeMatrix(catID, t->categoryCount);
    hp.q_pending[cat][code] = 0;
    if (hp.host_stale[cat][code]) {
      hp.host_stale[cat][code] = 0;
      hp.n_stale--;
    }
  }
  return false;
}

// ---- mode B: the host's ExponentiateMatrices hands its queue over instead of exponentiating (tree.cpp copy) ----
static bool _hyphy_hip_defer_handler(_TheTree *t, long catID, _List &nodesToDo, _List &matrixQueue, _SimpleList &parallel,
                                     _SimpleList &isExplicitForm, bool hasExpForm) {
  if (_hyhip_defer_depth <= 0) return false;
  if (_hyhip_force_defer_call) {  // mixture template mode took decisions in the first loop: it finishes the call
    _hyhip_force_defer_call = false;
    _HyHipPart *mp = _hyhip_part_of_tree(t);
    if (!mp) return false;
    _HyHipPart &fp = *mp;
    const long fcat = catID < 0 ? 0 : catID;
    // A "true" from here makes the host drop its queues.  That is only right when every queued node is one the mixture analysis of
    // THIS call handled (its probes and verifier; the skipped branches never entered the queue): a queue that also holds other
    // recomputed branches — not of the explicit-form group, or left over by the ordinary template mode — is finished by the host
    // (the call counts as broken: the skipped branches are recomputed by the host's own formula and nothing is dropped).
    {
      const _HyHipPart::MCall &mc = fp.mcall;
      bool foreign = false;
      for (unsigned long id = 0; id < nodesToDo.lLength && !foreign; id++) {
        auto c = fp.code_of.find(nodesToDo(id));
        if (c == fp.code_of.end()) {
          foreign = true;
          break;
        }
        const long code = c->second;
        foreign = !(std::find(mc.probe.begin(), mc.probe.end(), code) != mc.probe.end() || code == mc.verify ||
                    std::find(mc.skipped.begin(), mc.skipped.end(), code) != mc.skipped.end());
      }
      if (foreign) fp.mcall.broken = true;
    }
    if (fp.call.active) {
      // the ordinary template mode skipped branches in the same call and will not be finalised for it (this branch returns before
      // that code): take them the normal way — dense, exponentiated on the device — and close the call
      const long D = t->GetCodeBase(), DD = D * D;
      if (fcat < (long)fp.qstash.size() && fp.call.cat == fcat) {
        if (fp.qstash[fcat].empty()) fp.qstash[fcat].assign((size_t)(fp.code_of.size()) * DD, 0.);
        for (long code : fp.call.skipped) {
          _List lq;
          _SimpleList lt;
          _CalcNode *nd = (_CalcNode *)t->GetNodeFromFlatIndex(code);
          nd->RecomputeMatrix(catID, t->categoryCount, nil, &lq, &lt);
          if (lq.lLength != 1) {
            fp.mcall.broken = true;
            continue;
          }
          _hyhip_dense_copy((_Matrix *)lq(0), DD, fp.qstash[fcat].data() + (size_t)code * DD);
          fp.q_pending[fcat][code] = 1;
          if (!fp.host_stale[fcat][code]) fp.n_stale++;
          fp.host_stale[fcat][code] = 1;
        }
        fp.cat_arg[fcat] = catID;
      } else {
        fp.mcall.broken = true;
      }
      fp.call = _HyHipPart::Call();
      fp.mcall.broken = true;  // (a mixed call: the host keeps its queue)
    }
    return _hyhip_finish_mixture_call(fp, t, catID);
  }
  if (hasExpForm) {
    _HyHipPart *mp = _hyhip_part_of_tree(t);
    return mp ? _hyhip_defer_mixture(*mp, t, catID, nodesToDo, matrixQueue, parallel, isExplicitForm) : false;
  }
  auto own = _hyhip_tree_owner.find(t);
  if (own == _hyhip_tree_owner.end()) return false;
  auto it = _hyhip_lfs.find(own->second.first);
  if (it == _hyhip_lfs.end() || own->second.second >= (long)it->second.size()) return false;
  _HyHipPart &hp = it->second[own->second.second];
  if (!hp.part) return false;
  const long D = t->GetCodeBase(), DD = D * D, cat = catID < 0 ? 0 : catID;
  if (cat >= (long)hp.qstash.size()) return false;
  for (unsigned long id = 0; id < parallel.lLength; id++) {  // validate first: all or nothing
    _Matrix *m = (_Matrix *)matrixQueue(parallel.get(id));
    if (!m || !m->is_numeric() || m->GetHDim() != D || m->GetVDim() != D || !m->theData) return false;
    if (hp.code_of.find(nodesToDo(parallel.get(id))) == hp.code_of.end()) return false;
  }
  if (hp.qstash[cat].empty()) hp.qstash[cat].assign((size_t)(hp.code_of.size()) * DD, 0.);
  hp.cat_arg[cat] = catID;
  auto mark = [&](long code, char how) {  // how: 1 dense matrix in qstash, 2 template row in tmpl_x
    hp.q_pending[cat][code] = how;
    if (!hp.host_stale[cat][code]) hp.n_stale++;
    hp.host_stale[cat][code] = how;
  };
  for (unsigned long id = 0; id < parallel.lLength; id++) {  // everything that went through RecomputeMatrix: dense
    const long mid = parallel.get(id);
    const long code = hp.code_of.at(nodesToDo(mid));
    _hyhip_dense_copy((_Matrix *)matrixQueue(mid), DD, hp.qstash[cat].data() + (size_t)code * DD);
    mark(code, 1);
  }
  _hyhip_deferred += parallel.lLength;
  const long K = hp.tmpl_K;
  if (hp.call.active && hp.call.cat == cat) {
    // template call: per group K_g probes + one verification node were recomputed, hp.call.skipped were not.
    // verify < 0: every skipped branch of the group duplicates a probe (nothing to verify, nothing assumed); a group none of
    // whose branches is dirty in this call keeps its templates of the last call (no row of this call refers to them).
    bool ok = true, weak = false;
    std::vector<double> M = hp.tmpl_M[cat];
    std::vector<char> group_has_skipped(hp.tmpl_groups.size(), 0);
    for (long code : hp.call.skipped) group_has_skipped[_hyhip_group(hp, (_CalcNode *)t->GetNodeFromFlatIndex(code))] = 1;
    for (size_t g = 0; g < hp.tmpl_groups.size() && ok; g++) {
      const _HyHipPart::Group &G = hp.tmpl_groups[g];
      if (hp.call.probe[g].empty() && !group_has_skipped[g]) continue;
      // a verification taken closer than 1 % to the probes says little about branches that sit farther away: this call
      // (only) takes the skipped matrices the normal way
      weak = weak || (hp.call.verify[g] >= 0 && hp.call.skipped_dist[g] > 0. && hp.call.verify_dist[g] < 0.01 &&
                      hp.call.skipped_dist[g] > 4. * hp.call.verify_dist[g]);
      ok = _hyhip_solve_group(hp.tmpl_x[cat], K, DD, G, hp.call.probe[g], hp.qstash[cat].data(), M) &&
           (hp.call.verify[g] < 0 || _hyhip_template_error(K, D, hp.tmpl_x[cat].data() + (size_t)hp.call.verify[g] * K, M,
                                                           hp.qstash[cat].data() + (size_t)hp.call.verify[g] * DD) < 1e-11);
    }
    if (ok && !weak) {
      hp.tmpl_M[cat].swap(M);
      hp.tmpl_uploaded[cat] = 0;
      for (size_t g = 0; g < hp.tmpl_groups.size(); g++) {
        for (long code : hp.call.probe[g]) mark(code, 2);
        if (hp.call.verify[g] >= 0) mark(hp.call.verify[g], 2);
      }
      for (long code : hp.call.skipped) mark(code, 2);
      _hyhip_deferred += (long)hp.call.skipped.size();
    } else {
      // not linear after all (or no usable probes): the skipped matrices the normal way, and never again for this class
      // (a merely weak verification keeps the mode for later calls)
      if (!ok) {
        hp.tmpl_state[cat] = -1;
        ReportWarning("hyphy_hip: rate matrices are not linear in the branch parameters; template mode switched off");
      }
      for (long code : hp.call.skipped) {
        _List lq;
        _SimpleList lt;
        _CalcNode *nd = (_CalcNode *)t->GetNodeFromFlatIndex(code);
        nd->RecomputeMatrix(catID, t->categoryCount, nil, &lq, &lt);
        if (lq.lLength != 1) return false;
        _hyhip_dense_copy((_Matrix *)lq(0), DD, hp.qstash[cat].data() + (size_t)code * DD);
        mark(code, 1);
      }
    }
    hp.call = _HyHipPart::Call();
  } else if (cat < (long)hp.tmpl_state.size() && hp.tmpl_state[cat] == 0 && parallel.lLength >= 12) {
    // first large call of this class: is the model linear in the branches' local parameters, group by group?  (all matrices
    // are dense here.)  The groups are founded by the first class that gets here and are final afterwards.
    int verdict = -1;
    const bool founding = hp.tmpl_groups.empty();
    if (founding) {
      hp.tmpl_groups_open = true;
      hp.tmpl_K = 0;
      hp.group_of.clear();
    }
    std::vector<long> codes;
    std::vector<int> groups;
    bool all_conform = true;
    for (unsigned long id = 0; id < parallel.lLength && all_conform; id++) {
      _CalcNode *nd = (_CalcNode *)nodesToDo(parallel.get(id));
      const int g = _hyhip_group(hp, nd);
      all_conform = g >= 0;
      codes.push_back(hp.code_of.at(nd));
      groups.push_back(g);
    }
    hp.tmpl_groups_open = false;
    if (all_conform && hp.tmpl_K >= 1) {
      const long K0 = hp.tmpl_K;
      std::vector<double> &X = hp.tmpl_x[cat];
      X.assign(hp.code_of.size() * K0, 0.);
      for (unsigned long id = 0; id < parallel.lLength; id++)
        _hyhip_local_row(hp, (_CalcNode *)nodesToDo(parallel.get(id)), groups[id], X.data() + (size_t)codes[id] * K0);
      std::vector<double> M((size_t)K0 * DD, 0.);
      bool solved = true;
      for (size_t g = 0; g < hp.tmpl_groups.size() && solved; g++) {  // greedy choice of K_g independent probes per group
        const _HyHipPart::Group &G = hp.tmpl_groups[g];
        std::vector<long> probe;
        for (size_t id = 0; id < codes.size() && (long)probe.size() < G.K; id++)
          if (groups[id] == (int)g && _hyhip_rows_independent(X, K0, G, probe, codes[id])) probe.push_back(codes[id]);
        solved = _hyhip_solve_group(X, K0, DD, G, probe, hp.qstash[cat].data(), M);
      }
      if (solved) {
        double worst = 0.;
        for (long code : codes)
          worst = fmax(worst, _hyhip_template_error(K0, D, X.data() + (size_t)code * K0, M, hp.qstash[cat].data() + (size_t)code * DD));
        verdict = worst < 1e-11 ? 1 : -1;
        if (verdict == 1) hp.tmpl_M[cat] = M;
        if (getenv("HYPHY_HIP_VERBOSE"))
          fprintf(stderr, "[hyphy_hip] template analysis, class %ld: %ld branch class(es), %ld template column(s), %ld branches, worst relative deviation from linearity %.2e -> %s\n",
                  cat, (long)hp.tmpl_groups.size(), K0, (long)codes.size(), worst, verdict == 1 ? "template mode" : "dense matrices");
      } else verdict = 0;  // (dependent probes, e.g. every branch length of a group equal to zero: try again next time)
    }
    if (verdict != 1 && founding) {  // nothing founded that later calls may rely on
      hp.tmpl_groups.clear();
      hp.tmpl_K = 0;
      hp.group_of.clear();
      for (auto &x : hp.tmpl_x) x.clear();
    }
    hp.tmpl_state[cat] = verdict;
  }
  return true;
}

// bring the host-side transition matrices of one partition up to date (host exponentials of the stashed matrices)
static void _hyphy_hip_flush_part(_HyHipPart &hp, _TheTree *t) {
  if (hp.n_stale == 0) return;
  const long D = t->GetCodeBase(), DD = D * D;
  for (size_t cat = 0; cat < hp.host_stale.size(); cat++)
    for (size_t code = 0; code < hp.host_stale[cat].size(); code++)
      if (hp.host_stale[cat][code] == 3) {  // explicit-form mixture: the host's own formula (exponentials + recombination)
        ((_CalcNode *)t->GetNodeFromFlatIndex(code))->RecomputeMatrix(hp.cat_arg[cat], t->categoryCount);
        hp.host_stale[cat][code] = 0;
        hp.q_pending[cat][code] = 0;
      } else if (hp.host_stale[cat][code]) {
        _Matrix q(D, D, false, true);
        if (hp.host_stale[cat][code] == 2) _hyhip_template_dense(hp, (long)cat, (long)code, D, hp.qstash[cat].data() + code * DD);
        memcpy(q.theData, hp.qstash[cat].data() + code * DD, sizeof(double) * DD);
        ((_CalcNode *)t->GetNodeFromFlatIndex(code))->SetCompExp(&q, hp.cat_arg[cat], true);
        hp.host_stale[cat][code] = 0;
        hp.q_pending[cat][code] = 0;  // (the device receives the probabilities if it has not seen this matrix yet:
      }                               //  _hyphy_hip_compute resends every matrix the host lists)
  hp.n_stale = 0;
}

static bool _hyhip_cat_collect(_HyHipPart &hp, _TheTree *t, long cat, long catID, _SimpleList &branches, _List &matrices, long n_q, bool first,
                               long n_pending, long n_template, long n_mixture);
static void _hyhip_cat_flush(_HyHipPart &hp);
// one ComputeBlock evaluation on the device; returns 0 when *result is valid (or, with go_async, when it was enqueued:
// _hyphy_hip_prepass_result collects it)
static int _hyphy_hip_compute_impl(const void *lf, long index, _TheTree *t, long catID, _SimpleList &branches,
                                   _List &matrices, hyFloat *siteRes, long *scc, hyFloat *result, bool go_async);
static int _hyphy_hip_compute(const void *lf, long index, _TheTree *t, long catID, _SimpleList &branches,
                              _List &matrices, hyFloat *siteRes, long *scc, hyFloat *result, bool go_async = false) {
  timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  const int rc = _hyphy_hip_compute_impl(lf, index, t, catID, branches, matrices, siteRes, scc, result, go_async);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  _hyhip_compute_seconds += (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
  return rc;
}
static int _hyphy_hip_compute_impl(const void *lf, long index, _TheTree *t, long catID, _SimpleList &branches,
                                   _List &matrices, hyFloat *siteRes, long *scc, hyFloat *result, bool go_async) {
  _HyHipPart &hp = _hyhip_lfs[lf][index];
  const long D = t->GetCodeBase(), DD = D * D;
  const long B = t->GetLeafCount() + t->GetINodeCount() - 1;
  const long cat = catID < 0 ? 0 : catID;
  long n_q = matrices.lLength;
  const bool first = !hp.cat_seen[cat];
  if (first) n_q = B;  // first evaluation of a rate class: hand over every transition matrix
  hp.qnodes.resize(n_q);
  if (cat < (long)hp.mix_state.size() && hp.mix_state[cat] == 2) _hyhip_verify_mixture(hp, t, catID);
  long n_pending = 0, n_template = 0, n_mixture = 0, n_mixT = 0;
  for (long k = 0; k < n_q; k++) {
    hp.qnodes[k] = first ? k : hp.code_of.at(matrices(k));
    n_pending += hp.q_pending[cat][hp.qnodes[k]] != 0;
    n_template += hp.q_pending[cat][hp.qnodes[k]] == 2;
    n_mixture += hp.q_pending[cat][hp.qnodes[k]] == 3 || hp.q_pending[cat][hp.qnodes[k]] == 4;
    n_mixT += hp.q_pending[cat][hp.qnodes[k]] == 4;
  }
  double ll = 0.;
  int rc = 0;
  if (hp.spmd && (siteRes || scc || go_async)) return 1;  // (per-pattern outputs live on several ranks: this call stays on the host)
  if (_hyhip_cb.active && _hyhip_cb.lf == lf && _hyhip_cb.index == index && !go_async) {
    // the host's category loop is being batched: stash this class's inputs, the device runs once for all classes at the loop's end
    if (!_hyhip_cat_block && _hyhip_cat_collect(hp, t, cat, catID, branches, matrices, n_q, first, n_pending, n_template, n_mixture)) {
      *result = 0.;
      return 0;
    }
    _hyhip_cat_flush(hp);  // (a class that cannot be collected: the classes stashed so far run one by one now, this one the ordinary way)
  }
  if (n_mixture > 0 && (n_mixture < n_q || go_async)) {  // mixed with other kinds (rare): the host's own matrices for everything
    _hyphy_hip_flush_part(hp, t);
    n_pending = n_template = n_mixture = 0;
  }
  if (n_q > 0 && n_mixture == n_q && n_mixT == n_q && !hp.mixT_T[cat].empty()) {
    // mixture template mode: M x K templates (this call's), one row of K locals per (branch, component) at the component's columns
    const long M = hp.mix_M, K = hp.mixT_group.K, KT = M * K;
    rc = hyphy_hip_update_q_templates(hp.part, KT, hp.mixT_T[cat].data());
    for (auto &u : hp.tmpl_uploaded) u = 0;   // (the ordinary template mode's upload is gone)
    hp.pbuf.assign((size_t)n_q * M * KT + (size_t)n_q * M, 0.);
    double *wq = hp.pbuf.data() + (size_t)n_q * M * KT;
    std::vector<int64_t> cnt(n_q, M);
    for (long k = 0; k < n_q; k++)
      for (long m = 0; m < M; m++) {
        for (long j = 0; j < K; j++) hp.pbuf[((size_t)k * M + m) * KT + m * K + j] = hp.mixT_x[cat][(size_t)hp.qnodes[k] * K + j];
        wq[(size_t)k * M + m] = hp.mix_w[cat][m];
      }
    if (rc == 0) rc = hyphy_hip_build_q(hp.part, n_q * M, hp.pbuf.data());
    if (rc == 0)
      rc = hyphy_hip_evaluate_mixture_built(hp.part, catID, (const int64_t *)branches.list_data, branches.lLength, hp.qnodes.data(), n_q,
                                            cnt.data(), wq, t->GetProbs(), &ll, siteRes, (int64_t *)scc);
    if (rc < 0) {
      HandleApplicationError(_String("hyphy_hip_evaluate_mixture_built: ") & hyphy_hip_last_error());
      return rc;
    }
    if (rc == 0) {
      for (long k = 0; k < n_q; k++) hp.q_pending[cat][hp.qnodes[k]] = 0;
      hp.cat_seen[cat] = 1;
      hp.n_mixT_evals++;
      _hyhip_calls++;
      *result = ll;
    }
    return rc;
  }
  if (n_q > 0 && n_mixture == n_q) {
    const long M = hp.mix_M;
    if (n_mixT > 0) {  // rows and dense components mixed: the rows become dense component matrices
      const long K = hp.mixT_group.K;
      for (long k = 0; k < n_q; k++)
        if (hp.q_pending[cat][hp.qnodes[k]] == 4) {
          const long code = hp.qnodes[k];
          for (long m = 0; m < M; m++) {
            double *dst = hp.mix_q[cat].data() + ((size_t)code * M + m) * DD;
            for (long e = 0; e < DD; e++) {
              double v = 0.;
              for (long j = 0; j < K; j++) v += hp.mixT_x[cat][(size_t)code * K + j] * hp.mixT_T[cat][((size_t)m * K + j) * DD + e];
              dst[e] = v;
            }
            for (long i = 0; i < D; i++) {
              double d = 0.;
              for (long j2 = 0; j2 < D; j2++)
                if (j2 != i) d -= dst[i * D + j2];
              dst[i * D + i] = d;
            }
          }
          hp.q_pending[cat][code] = 3;
        }
    }
    hp.pbuf.resize((size_t)n_q * M * DD + (size_t)n_q * M);
    double *wq = hp.pbuf.data() + (size_t)n_q * M * DD;
    std::vector<int64_t> cnt(n_q, M);
    for (long k = 0; k < n_q; k++) {
      memcpy(hp.pbuf.data() + (size_t)k * M * DD, hp.mix_q[cat].data() + (size_t)hp.qnodes[k] * M * DD, sizeof(double) * M * DD);
      for (long m = 0; m < M; m++) wq[(size_t)k * M + m] = hp.mix_w[cat][m];
    }
    rc = hyphy_hip_evaluate_mixture(hp.part, catID, (const int64_t *)branches.list_data, branches.lLength, hp.qnodes.data(), n_q,
                                    cnt.data(), hp.pbuf.data(), wq, t->GetProbs(), &ll, siteRes, (int64_t *)scc);
    if (rc < 0) {
      HandleApplicationError(_String("hyphy_hip_evaluate_mixture: ") & hyphy_hip_last_error());
      return rc;
    }
    if (rc == 0) {
      for (long k = 0; k < n_q; k++) hp.q_pending[cat][hp.qnodes[k]] = 0;
      hp.cat_seen[cat] = 1;
      hp.n_mixture_evals++;
      _hyhip_calls++;
      *result = ll;
    }
    return rc;
  }
  if (n_q > 0 && n_template == n_q && !go_async) {
    // template mode: K coefficients per branch instead of D*D matrix entries; the device builds and exponentiates
    const long K = hp.tmpl_K;
    if (!hp.tmpl_uploaded[cat] || hp.tmpl_device_cat != cat) {
      rc = hyphy_hip_update_q_templates(hp.part, K, hp.tmpl_M[cat].data());
      hp.tmpl_uploaded[cat] = 1;
      hp.tmpl_device_cat = cat;
    }
    hp.pbuf.resize((size_t)n_q * K);
    for (long k = 0; k < n_q; k++)
      for (long j = 0; j < K; j++) hp.pbuf[(size_t)k * K + j] = hp.tmpl_x[cat][(size_t)hp.qnodes[k] * K + j];
    if (rc == 0) rc = hyphy_hip_build_q(hp.part, n_q, hp.pbuf.data());
    if (rc == 0 && hp.spmd && hp.spmd_host)
      rc = hyphy_hip_evaluate_built_exchange(hp.part, catID, (const int64_t *)branches.list_data, branches.lLength, hp.qnodes.data(),
                                             n_q, t->GetProbs(), &ll);
    else if (rc == 0 && hp.spmd)
      rc = hyphy_hip_evaluate_built_allreduce(hp.part, catID, (const int64_t *)branches.list_data, branches.lLength, hp.qnodes.data(),
                                              n_q, t->GetProbs(), &ll);
    else if (rc == 0)
      rc = hyphy_hip_evaluate_built_sites(hp.part, catID, (const int64_t *)branches.list_data, branches.lLength, hp.qnodes.data(),
                                          n_q, t->GetProbs(), &ll, siteRes, (int64_t *)scc);
    if (rc < 0) {
      HandleApplicationError(_String("hyphy_hip (template mode): ") & hyphy_hip_last_error());
      return rc;
    }
    if (rc == 0) {
      for (long k = 0; k < n_q; k++) hp.q_pending[cat][hp.qnodes[k]] = 0;
      hp.cat_seen[cat] = 1;
      hp.n_template_evals++;
      _hyhip_calls++;
      *result = ll;
    }
    return rc;
  }
  if (n_template > 0)  // mixed (or asynchronous): template rows become dense rate matrices
    for (long k = 0; k < n_q; k++)
      if (hp.q_pending[cat][hp.qnodes[k]] == 2) {
        _hyhip_template_dense(hp, cat, hp.qnodes[k], D, hp.qstash[cat].data() + (size_t)hp.qnodes[k] * DD);
        hp.q_pending[cat][hp.qnodes[k]] = 1;
        if (hp.host_stale[cat][hp.qnodes[k]]) hp.host_stale[cat][hp.qnodes[k]] = 1;
      }
  if (n_pending > 0 && n_pending < n_q) {  // mixed (rare): exponentiate the stashed ones on the host, send probabilities
    _hyphy_hip_flush_part(hp, t);
    n_pending = 0;
  }
  const bool rate_matrices = n_pending > 0;
  hp.pbuf.resize((size_t)n_q * DD);
  for (long k = 0; k < n_q; k++) {
    if (rate_matrices) {
      memcpy(hp.pbuf.data() + (size_t)k * DD, hp.qstash[cat].data() + (size_t)hp.qnodes[k] * DD, sizeof(double) * DD);
      continue;
    }
    _CalcNode *n = first ? (_CalcNode *)t->GetNodeFromFlatIndex(k) : (_CalcNode *)matrices(k);
    _Matrix *P = n->GetCompExp(catID);
    if (!P || !P->theData) return 1;
    memcpy(hp.pbuf.data() + (size_t)k * DD, P->theData, sizeof(double) * DD);
  }
  if (go_async) {
    rc = hyphy_hip_evaluate_async(hp.part, catID, (const int64_t *)branches.list_data, branches.lLength, hp.qnodes.data(), n_q,
                                  hp.pbuf.data(), rate_matrices ? 0 : 1, t->GetProbs());
    if (rc == 0) hp.pending = true;
  } else if (hp.spmd && hp.spmd_host) {
    rc = hyphy_hip_evaluate_exchange(hp.part, catID, (const int64_t *)branches.list_data, branches.lLength, hp.qnodes.data(), n_q,
                                     hp.pbuf.data(), /* q_is_probability = */ rate_matrices ? 0 : 1, t->GetProbs(), &ll);
  } else if (hp.spmd) {
    rc = hyphy_hip_evaluate_allreduce(hp.part, catID, (const int64_t *)branches.list_data, branches.lLength, hp.qnodes.data(), n_q,
                                      hp.pbuf.data(), /* q_is_probability = */ rate_matrices ? 0 : 1, t->GetProbs(), &ll);
  } else {
    rc = hyphy_hip_evaluate(hp.part, catID, (const int64_t *)branches.list_data, branches.lLength, hp.qnodes.data(), n_q,
                            hp.pbuf.data(), /* q_is_probability = */ rate_matrices ? 0 : 1, t->GetProbs(), &ll, siteRes,
                            (int64_t *)scc);
  }
  if (rc < 0) {
    HandleApplicationError(_String("hyphy_hip_evaluate: ") & hyphy_hip_last_error());
    return rc;
  }
  if (rc == 0) {
    if (rate_matrices)
      for (long k = 0; k < n_q; k++) hp.q_pending[cat][hp.qnodes[k]] = 0;
    hp.cat_seen[cat] = 1;
    _hyhip_calls++;
    *result = ll;
  }
  return rc;
}

// ---- rate classes batched through the host's category loop ------------------------------------------------------------------------
static bool _hyhip_cat_collect(_HyHipPart &hp, _TheTree *t, long cat, long catID, _SimpleList &branches, _List &matrices, long n_q, bool first,
                               long n_pending, long n_template, long n_mixture) {
  _HyHipCatBatch &cb = _hyhip_cb;
  if (first || n_mixture > 0 || cat < 0 || cat >= cb.C || cb.calls[cat].taken) return false;
  const long D = t->GetCodeBase(), DD = D * D;
  _HyHipCatBatch::ClassCall &cc = cb.calls[cat];
  cc.catID = catID;
  cc.upd.assign((const int64_t *)branches.list_data, (const int64_t *)branches.list_data + branches.lLength);
  cc.qn.assign(hp.qnodes.begin(), hp.qnodes.begin() + n_q);
  if (n_q == 0) {
    cc.kind = 4;
    cc.data.clear();
  } else if (n_template == n_q) {
    const long K = hp.tmpl_K;
    cc.kind = 1;
    cc.data.resize((size_t)n_q * K);
    for (long k = 0; k < n_q; k++)
      for (long j = 0; j < K; j++) cc.data[(size_t)k * K + j] = hp.tmpl_x[cat][(size_t)hp.qnodes[k] * K + j];
  } else {
    if (n_template > 0)  // mixed: template rows become dense rate matrices (as in the ordinary call)
      for (long k = 0; k < n_q; k++)
        if (hp.q_pending[cat][hp.qnodes[k]] == 2) {
          _hyhip_template_dense(hp, cat, hp.qnodes[k], D, hp.qstash[cat].data() + (size_t)hp.qnodes[k] * DD);
          hp.q_pending[cat][hp.qnodes[k]] = 1;
          if (hp.host_stale[cat][hp.qnodes[k]]) hp.host_stale[cat][hp.qnodes[k]] = 1;
        }
    if (n_pending > 0 && n_pending < n_q) {
      _hyphy_hip_flush_part(hp, t);
      n_pending = 0;
    }
    const bool rate_matrices = n_pending > 0;
    cc.kind = rate_matrices ? 2 : 3;
    cc.data.resize((size_t)n_q * DD);
    for (long k = 0; k < n_q; k++) {
      if (rate_matrices) {
        memcpy(cc.data.data() + (size_t)k * DD, hp.qstash[cat].data() + (size_t)hp.qnodes[k] * DD, sizeof(double) * DD);
        continue;
      }
      _Matrix *P = ((_CalcNode *)matrices(k))->GetCompExp(catID);
      if (!P || !P->theData) return false;
      memcpy(cc.data.data() + (size_t)k * DD, P->theData, sizeof(double) * DD);
    }
  }
  cc.have = cc.taken = true;
  return true;
}

// what the device did with class `cat`'s stashed matrices: nothing is pending any more
static void _hyhip_cat_consumed(_HyHipPart &hp, const _HyHipCatBatch::ClassCall &cc, long cat) {
  if (cc.kind == 1 || cc.kind == 2)
    for (int64_t code : cc.qn) hp.q_pending[cat][code] = 0;
  hp.cat_seen[cat] = 1;
}

// one stashed class through its own device call, mixed into the host's buffer / scalers with the reference's rule
// (likefunc2.cpp:826-853: the running sum is kept at the smallest exponent seen so far)
static bool _hyhip_cat_single(_HyHipPart &hp, long cat) {
  _HyHipCatBatch &cb = _hyhip_cb;
  _HyHipCatBatch::ClassCall &cc = cb.calls[cat];
  _TheTree *t = cb.tree;
  std::vector<double> lik(cb.S);
  std::vector<int64_t> cnt(cb.S);
  double ll = 0.;
  int rc = 0;
  if (cc.kind == 1) {
    const long K = hp.tmpl_K;
    rc = hyphy_hip_update_q_templates(hp.part, K, hp.tmpl_M[cat].data());
    for (auto &u : hp.tmpl_uploaded) u = 0;
    hp.tmpl_uploaded[cat] = 1;
    hp.tmpl_device_cat = cat;
    if (rc == 0) rc = hyphy_hip_build_q(hp.part, (int64_t)cc.qn.size(), cc.data.data());
    if (rc == 0)
      rc = hyphy_hip_evaluate_built_sites(hp.part, cc.catID, cc.upd.data(), (int64_t)cc.upd.size(), cc.qn.data(), (int64_t)cc.qn.size(), t->GetProbs(),
                                          &ll, lik.data(), cnt.data());
  } else {
    rc = hyphy_hip_evaluate(hp.part, cc.catID, cc.upd.data(), (int64_t)cc.upd.size(), cc.qn.data(), (int64_t)cc.qn.size(), cc.data.data(),
                            cc.kind == 3 ? 1 : 0, t->GetProbs(), &ll, lik.data(), cnt.data());
  }
  if (rc != 0) {
    HandleApplicationError(_String("hyphy_hip (rate classes, one by one): ") & hyphy_hip_last_error());
    return false;
  }
  _hyhip_cat_consumed(hp, cc, cat);
  _hyhip_calls++;
  cb.n_single++;
  const double w = cb.weights[cat];
  for (long s = 0; s < cb.S; s++) {
    const long scv = (long)cnt[s];
    if (!cb.mixed_any) {
      cb.buffer[s] = w * lik[s];
      cb.scalers[s] = scv;
    } else if (scv < cb.scalers[s]) {
      cb.buffer[s] = w * lik[s] + cb.buffer[s] * acquireScalerMultiplier(cb.scalers[s] - scv);
      cb.scalers[s] = scv;
    } else if (scv > cb.scalers[s]) {
      cb.buffer[s] += w * lik[s] * acquireScalerMultiplier(scv - cb.scalers[s]);
    } else {
      cb.buffer[s] += w * lik[s];
    }
  }
  cb.mixed_any = true;
  cc.have = false;
  return true;
}

// the batch cannot go on (a class that is not collectable turned up): the stashed classes run one by one, in class order
static void _hyhip_cat_flush(_HyHipPart &hp) {
  _HyHipCatBatch &cb = _hyhip_cb;
  cb.active = false;
  for (long c = 0; c < cb.C; c++)
    if (cb.calls[c].have && !_hyhip_cat_single(hp, c)) return;
}

// (extern: called from the likefunc2.cpp copy)
bool _hyphy_hip_cat_begin(const void *lf, long index, long n_classes, const hyFloat *weights, hyFloat *buffer, long *scalers, long block_length) {
  _HyHipCatBatch &cb = _hyhip_cb;
  cb.active = false;
  cb.logl_valid = false;
  static const bool off = getenv("HYPHY_HIP_CAT_BATCH") && !strcmp(getenv("HYPHY_HIP_CAT_BATCH"), "0");
  if (off || !_hyphy_hip_enabled() || n_classes < 2 || !weights) return false;
  auto it = _hyhip_lfs.find(lf);
  if (it == _hyhip_lfs.end() || index < 0 || index >= (long)it->second.size()) return false;
  _HyHipPart &hp = it->second[index];
  if (!hp.part || hp.spmd || hp.pending || (long)hp.cat_seen.size() != n_classes) return false;
  for (long c = 0; c < n_classes; c++) {
    if (!hp.cat_seen[c] || !(weights[c] > 0.)) return false;           // (first evaluations and zero weights: the host's own loop)
    if (c < (long)hp.mix_state.size() && hp.mix_state[c] > 0) return false;  // (explicit-form mixtures have their own entry points)
  }
  _TheTree *t = nullptr;
  for (auto &o : _hyhip_tree_owner)
    if (o.second.first == lf && o.second.second == index) t = (_TheTree *)o.first;
  if (!t) return false;
  cb.active = true;
  cb.lf = lf;
  cb.index = index;
  cb.C = n_classes;
  cb.S = block_length;
  cb.tree = t;
  cb.buffer = buffer;
  cb.scalers = scalers;
  cb.mixed_any = false;
  cb.weights.assign(weights, weights + n_classes);
  cb.calls.assign(n_classes, _HyHipCatBatch::ClassCall());
  return true;
}

// true: the adapter answers for this class — the host must not mix what ComputeBlock left in its buffer
bool _hyphy_hip_cat_collected(const void *lf, long index, long cls) {
  const _HyHipCatBatch &cb = _hyhip_cb;
  return cb.lf == lf && cb.index == index && cls >= 0 && cls < (long)cb.calls.size() && cb.calls[cls].taken;
}

void _hyphy_hip_cat_want(const void *lf, long index) {
  _hyhip_cb.want_lf = lf;
  _hyhip_cb.want_index = index;
  _hyhip_cb.logl_valid = false;
}

bool _hyphy_hip_cat_take_logl(const void *lf, long index, hyFloat *value) {
  _HyHipCatBatch &cb = _hyhip_cb;
  const bool ok = cb.logl_valid && cb.want_lf == lf && cb.want_index == index;
  if (ok) *value = cb.logl;
  cb.logl_valid = false;
  cb.want_lf = nullptr;
  cb.want_index = -1;
  return ok;
}

bool _hyphy_hip_cat_end(const void *lf, long index) {
  _HyHipCatBatch &cb = _hyhip_cb;
  if (cb.lf != lf || cb.index != index) return true;
  const bool was_active = cb.active;
  cb.active = false;
  _HyHipPart &hp = _hyhip_lfs[lf][index];
  long n_have = 0;
  for (long c = 0; c < cb.C; c++) n_have += cb.calls[c].have ? 1 : 0;
  if (n_have == 0) return true;
  _TheTree *t = cb.tree;
  const long D = t->GetCodeBase(), DD = D * D;
  bool batch = was_active && n_have == cb.C && !cb.mixed_any;
  for (long c = 1; c < cb.C && batch; c++)
    batch = cb.calls[c].kind == cb.calls[0].kind && cb.calls[c].upd == cb.calls[0].upd && cb.calls[c].qn == cb.calls[0].qn;
  if (batch) {
    const _HyHipCatBatch::ClassCall &c0 = cb.calls[0];
    const int64_t n_q = (int64_t)c0.qn.size();
    const bool only_logl = cb.want_lf == lf && cb.want_index == index;  // (Compute: the per-pattern values would only be summed up)
    double ll = 0.;
    int rc = 0;
    if (c0.kind == 1) {
      // template mode: the classes' K templates side by side (C * K of them), class c's row of K locals at its own columns
      const long K = hp.tmpl_K, KT = cb.C * K;
      std::vector<double> T((size_t)KT * DD);
      for (long c = 0; c < cb.C; c++) memcpy(T.data() + (size_t)c * K * DD, hp.tmpl_M[c].data(), sizeof(double) * K * DD);
      rc = hyphy_hip_update_q_templates(hp.part, KT, T.data());
      for (auto &u : hp.tmpl_uploaded) u = 0;   // (the one-class template sets are gone from the device)
      hp.tmpl_device_cat = -2;
      hp.pbuf.assign((size_t)cb.C * n_q * KT, 0.);
      for (long c = 0; c < cb.C; c++)
        for (int64_t k = 0; k < n_q; k++)
          memcpy(hp.pbuf.data() + ((size_t)c * n_q + k) * KT + (size_t)c * K, cb.calls[c].data.data() + (size_t)k * K, sizeof(double) * K);
      if (rc == 0) rc = hyphy_hip_build_q(hp.part, cb.C * n_q, hp.pbuf.data());
      if (rc == 0)
        rc = hyphy_hip_evaluate_categories_built_sites(hp.part, c0.upd.data(), (int64_t)c0.upd.size(), c0.qn.data(), n_q, cb.weights.data(),
                                                       t->GetProbs(), &ll, only_logl ? nullptr : cb.buffer, only_logl ? nullptr : (int64_t *)cb.scalers);
    } else {
      hp.pbuf.resize((size_t)cb.C * n_q * DD);
      for (long c = 0; c < cb.C && n_q > 0; c++) memcpy(hp.pbuf.data() + (size_t)c * n_q * DD, cb.calls[c].data.data(), sizeof(double) * n_q * DD);
      rc = hyphy_hip_evaluate_categories(hp.part, c0.upd.data(), (int64_t)c0.upd.size(), c0.qn.data(), n_q, n_q > 0 ? hp.pbuf.data() : nullptr,
                                         c0.kind == 3 ? 1 : 0, cb.weights.data(), t->GetProbs(), &ll, only_logl ? nullptr : cb.buffer,
                                         only_logl ? nullptr : (int64_t *)cb.scalers);
    }
    if (rc < 0) {
      HandleApplicationError(_String("hyphy_hip (rate classes in one evaluation): ") & hyphy_hip_last_error());
      return false;
    }
    if (rc == 0) {
      for (long c = 0; c < cb.C; c++) {
        _hyhip_cat_consumed(hp, cb.calls[c], c);
        cb.calls[c].have = false;
      }
      _hyhip_calls++;
      cb.n_batched++;
      if (only_logl) {
        cb.logl = ll;
        cb.logl_valid = true;
      }
      return true;
    }
    // (rc > 0: this form is not supported by the library — the classes one by one)
  }
  for (long c = 0; c < cb.C; c++)
    if (cb.calls[c].have && !_hyhip_cat_single(hp, c)) return false;
  return true;
}

// pre-pass of _LikelihoodFunction::Compute: every device partition is enqueued before the first result is waited for
static void _hyphy_hip_note_prepass(const void *lf, long index, hyFloat value) {
  _HyHipPart &hp = _hyhip_lfs[lf][index];
  if (!hp.pending) {  // ComputeBlock finished synchronously (branch cache, CPU path ...): keep what it returned
    hp.pre_done = true;
    hp.pre_value = value;
  }
}
static bool _hyphy_hip_prepass_result(const void *lf, long index, hyFloat *value) {
  auto it = _hyhip_lfs.find(lf);
  if (it == _hyhip_lfs.end() || index >= (long)it->second.size()) return false;
  _HyHipPart &hp = it->second[index];
  if (hp.pending) {
    hp.pending = false;
    double ll = 0.;
    if (hyphy_hip_collect(hp.part, &ll, nullptr, nullptr) != 0) {
      HandleApplicationError(_String("hyphy_hip_collect: ") & hyphy_hip_last_error());
      return false;
    }
    *value = ll;
    return true;
  }
  if (hp.pre_done) {
    hp.pre_done = false;
    *value = hp.pre_value;
    return true;
  }
  return false;
}

// pinned node states (branchIndex >= 0: marginal ancestral reconstruction, likefunc2.cpp:932-1040): pin, evaluate, unpin
static int _hyphy_hip_pinned(const void *lf, long index, _TheTree *t, long catID, _SimpleList &branches, _List &matrices,
                             long node_code, long const *states, hyFloat *siteRes, long *scc, hyFloat *result) {
  _HyHipPart &hp = _hyhip_lfs[lf][index];
  if (hp.spmd) return 1;  // (pinned evaluations are per-pattern: host path)
  int rc = hyphy_hip_set_pinned_states(hp.part, node_code, (const int64_t *)states);
  if (rc != 0) return rc > 0 ? rc : 1;
  rc = _hyphy_hip_compute(lf, index, t, catID, branches, matrices, siteRes, scc, result);
  hyphy_hip_set_pinned_states(hp.part, -1, nullptr);
  return rc;
}

// branch cache (SURVEY 8f-1): device counterparts of ComputeBranchCache / ComputeLLWithBranchCache, driven by the
// reference's own policy state machine (computedLocalUpdatePolicy, likefunc.cpp:10886-10948)
static int _hyphy_hip_cache_build(const void *lf, long index, long catID, long node) {
  _HyHipPart &hp = _hyhip_lfs[lf][index];
  if (hp.spmd) return 1;  // (the cached evaluation has no all-reduce entry point: the policy keeps evaluating normally)
  int rc = hyphy_hip_branch_cache_build(hp.part, catID, node);
  if (rc < 0) ReportWarning(_String("hyphy_hip_branch_cache_build: ") & hyphy_hip_last_error());
  return rc;
}
static int _hyphy_hip_cached(const void *lf, long index, _TheTree *t, long catID, long node, hyFloat *siteRes,
                             long *scc, hyFloat *result) {
  _HyHipPart &hp = _hyhip_lfs[lf][index];
  const long cat = catID < 0 ? 0 : catID, DD = t->GetCodeBase() * t->GetCodeBase();
  if (hp.q_pending[cat][node] == 3 || hp.q_pending[cat][node] == 4) {  // (an explicit-form mixture: the host's own matrix for this one branch)
    ((_CalcNode *)t->GetNodeFromFlatIndex(node))->RecomputeMatrix(hp.cat_arg[cat], t->categoryCount);
    hp.q_pending[cat][node] = 0;
    if (hp.host_stale[cat][node]) {
      hp.host_stale[cat][node] = 0;
      hp.n_stale--;
    }
  }
  if (hp.q_pending[cat][node] == 2) {  // (a template row: make it the dense rate matrix)
    _hyhip_template_dense(hp, cat, node, t->GetCodeBase(), hp.qstash[cat].data() + (size_t)node * DD);
    hp.q_pending[cat][node] = 1;
    if (hp.host_stale[cat][node]) hp.host_stale[cat][node] = 1;
  }
  const bool rate_matrix = hp.q_pending[cat][node];  // (mode B: the line search's new matrix was handed over, not exponentiated)
  const double *mx = nullptr;
  if (rate_matrix) {
    mx = hp.qstash[cat].data() + (size_t)node * DD;
  } else {
    _Matrix *P = ((_CalcNode *)t->GetNodeFromFlatIndex(node))->GetCompExp(catID);
    if (!P || !P->theData) return 1;
    mx = P->theData;
  }
  double ll = 0.;
  int rc = hyphy_hip_branch_cache_evaluate(hp.part, catID, node, mx, /* q_is_probability = */ rate_matrix ? 0 : 1, &ll, siteRes,
                                           (int64_t *)scc);
  if (rc == 0 && rate_matrix) hp.q_pending[cat][node] = 0;
  if (rc < 0) {
    HandleApplicationError(_String("hyphy_hip_branch_cache_evaluate: ") & hyphy_hip_last_error());
    return rc;
  }
  if (rc == 0) {
    _hyhip_cached_calls++;
    *result = ll;
  }
  return rc;
}

static void _hyphy_hip_flush(_LikelihoodFunction *lf) {
  auto it = _hyhip_lfs.find(lf);
  if (it == _hyhip_lfs.end()) return;
  for (size_t i = 0; i < it->second.size(); i++)
    if (it->second[i].part) _hyphy_hip_flush_part(it->second[i], lf->GetIthTree(i));
}
struct _HyHipOptimizeScope {  // Optimize: device exponentials inside, host matrices brought up to date on the way out
  _LikelihoodFunction *lf;
  bool on;
  explicit _HyHipOptimizeScope(_LikelihoodFunction *l) : lf(l), on(_hyphy_hip_enabled() && _hyphy_hip_expm_mode() == 1) {
    if (on) _hyhip_defer_depth++;
  }
  ~_HyHipOptimizeScope() {
    if (on && --_hyhip_defer_depth == 0) _hyphy_hip_flush(lf);
  }
};
#endif
'''

# ---- block 2: SetupLFCaches, once the leaf table of partition i is complete ------------------------
SETUP = r'''
#ifdef HYPHY_HIP
    _hyphy_hip_setup(this, i, theTrees.lLength, cT, theFilter, conditionalTerminalNodeStateFlag[i], ambigs);
#endif
'''

# ---- block 3: DeleteCaches ---------------------------------------------------------------------------
TEARDOWN = r'''
#ifdef HYPHY_HIP
  _hyphy_hip_teardown(this);
#endif
'''

# ---- block 4: ComputeBlock, between ExponentiateMatrices and the OpenMP pruning loop -----------------
COMPUTE = r'''
#ifdef HYPHY_HIP
      if (branchIndex >= 0 && branchValues && _hyphy_hip_active(this, index)) {
        // pinned node states: internal node branchIndex, or leaf branchIndex - #internal nodes (:10953-10956)
        hyFloat hip_result = 0.;
        const long n_int = t->GetINodeCount();
        const long code = branchIndex < n_int ? t->GetLeafCount() + branchIndex : branchIndex - n_int;
        if (_hyphy_hip_pinned(this, index, t, catID, *branches, *matrices, code, branchValues->list_data, siteRes, scc,
                              &hip_result) == 0) {
          return hip_result;
        }
      }
      if (branchIndex < 0 && _hyphy_hip_active(this, index)) {
        hyFloat hip_result = 0.;
        if (doCachedComp >= 3) {  // one-branch line search: a single contraction against the device branch cache
          const int crc = _hyphy_hip_cached(this, index, t, catID, doCachedComp - 3, siteRes, scc, &hip_result);
          if (crc == 0) {
            return hip_result;
          }
          if (crc > 0)  // "unsupported here" from the cached entry point: say so (an error, crc < 0, was reported already)
            ReportWarning(_String("hyphy_hip_branch_cache_evaluate declined (") & hyphy_hip_last_error() & "); this evaluation returns -infinity");
          return -INFINITY;  // (the host caches were never filled)
        }
        const bool go_async = _hyhip_async_phase > 0 && !siteRes && doCachedComp == 0;
        _hyhip_cat_block = doCachedComp != 0;
        if (_hyphy_hip_compute(this, index, t, catID, *branches, *matrices, siteRes, scc, &hip_result, go_async) == 0) {
          if (doCachedComp < 0) {  // the policy asked for a cache of this branch after the normal pass
            const long nd = -doCachedComp - 1;
            if (_hyphy_hip_cache_build(this, index, catID, nd) == 0) {
              *cbid = nd;
            } else {  // not available (e.g. 4-state path): keep evaluating normally
              ((_SimpleList *)computedLocalUpdatePolicy(index))->list_data[ciid] = 1;
            }
          }
          return hip_result;  // already  sum_s f_s log L_s - 64 ln2 * scalers  (likefunc.cpp:11123)
        }
      }
      if (_hyphy_hip_active(this, index)) {
        // This call stays on the CPU (pinned node states for marginal ancestral reconstruction, branchIndex >= 0,
        // or an "unsupported" return): the host caches were never filled by the device evaluations before it, so
        // the pass must recompute every node, like the first evaluation after a setup (:10964-10966).
        _hyphy_hip_flush(this);
        branches->Populate(t->GetINodeCount() + t->GetLeafCount() - 1, 0, 1);
        // ... and the per-pattern exponents the device wrote into the siteCorrections slice are ABSOLUTE values, while the
        // reference's pruning code accumulates changes (+= didScale) on top of what it finds and keeps the matching
        // total in overallScalingFactors: start the CPU pass of this partition from a clean slate
        if (scc) {
          const long pc = df->GetPatternCount();
          for (long s_ = 0; s_ < pc; s_++) scc[s_] = 0;
        }
        if (currentRateClass < 1) overallScalingFactors.list_data[index] = 0;
      }
#endif
'''

# ---- block 6: _LikelihoodFunction::Compute, in front of the partition loop (likefunc.cpp:2524) -----------------------
PREPASS = r'''
#ifdef HYPHY_HIP
    if (theTrees.lLength > 1UL && _hyphy_hip_enabled()) {
      // The reference evaluates its partitions one after the other (ComputeBlock is synchronous); on the device the
      // partitions live on different GPUs / streams, so: enqueue every one of them first, collect in the loop below.
      _hyhip_async_phase = 1;
      for (unsigned long partID = 0; partID < theTrees.lLength; partID++)
        if (!blockDependancies.list_data[partID] && _hyphy_hip_active(this, partID)) {
          const hyFloat r_ = ComputeBlock(partID);
          _hyphy_hip_note_prepass(this, partID, r_);
        }
      _hyhip_async_phase = 0;
    }
#endif
'''
# ... and the loop's own call, which must not run ComputeBlock a second time for those partitions
LOOPCALL_OLD = "        hyFloat blockResult = ComputeBlock(partID);\n        if (blockMatrix) {"
LOOPCALL_NEW = r'''#ifdef HYPHY_HIP
        hyFloat blockResult = 0.;
        if (!_hyphy_hip_prepass_result(this, partID, &blockResult)) blockResult = ComputeBlock(partID);
#else
        hyFloat blockResult = ComputeBlock(partID);
#endif
        if (blockMatrix) {'''

# ---- block 4b: BenchmarkThreads (likefunc.cpp:219-420) ---------------------------------------------------------------
# Optimize starts by timing Compute() at 1, 2, 3 ... threads (3-5 trials each, up to the machine's CPU count) to pick the
# OpenMP width of the pruning loop.  With every partition on the device that loop never runs: skip the timing runs
# (dozens of likelihood evaluations) and give the host-side leftovers (formula engine, mode-A exponentials) a fixed width.
BENCHMARK = r'''
#ifdef HYPHY_HIP
  if (_hyphy_hip_enabled()) {
    bool all_on_device = lf->CountObjects(kLFCountPartitions) > 0;
    for (long i_ = 0; i_ < lf->CountObjects(kLFCountPartitions) && all_on_device; i_++) all_on_device = _hyphy_hip_active(lf, i_);
    if (all_on_device) {
      lf->SetThreadCount(MIN(16L, (long)hy_global::system_CPU_count));
      return logL;
    }
  }
#endif
'''

# ---- block 5: Optimize, first statement ----------------------------------------------------------------------------
OPTIMIZE = r'''
#ifdef HYPHY_HIP
  _HyHipOptimizeScope _hyhip_scope(this);
#endif
'''

# ---- tree.cpp copy: the hook and its call site in ExponentiateMatrices (mode B) -----------------------------------
TREE_HOOK_DEF = r'''
#ifdef HYPHY_HIP
// set by the likelihood-function adapter (likefunc.cpp copy); returns true when it took the queued rate matrices
bool (*_hyhip_defer_expm_hook)(_TheTree *, long, _List &, _List &, _SimpleList &, _SimpleList &, bool) = nullptr;
// template mode: asked for every node of ExponentiateMatrices' first loop; true = the adapter derives this node's rate
// matrix from the probes of this call, do not run RecomputeMatrix for it
bool (*_hyhip_skip_recompute_hook)(_TheTree *, long, _CalcNode *, unsigned long, unsigned long) = nullptr;
// mixture template mode: the adapter took decisions in the first loop (branches skipped, probes chosen) and must see the end of
// this call even when nothing is queued for the OpenMP loop (weights-only changes of an explicit-form model queue finished
// matrices, tag -1, on the serial list)
bool _hyhip_force_defer_call = false;
#endif
'''
TREE_SKIP_OLD = "    if (thisNode->RecomputeMatrix(catID, categoryCount, nil, &matrixQueue,\n                                  &isExplicitForm)) {"
TREE_SKIP_NEW = r'''#ifdef HYPHY_HIP
    if (_hyhip_skip_recompute_hook && _hyhip_skip_recompute_hook(this, catID, thisNode, nodeID, expNodes.lLength)) {
      // (template mode: nothing queued for this node)
    } else
#endif
    if (thisNode->RecomputeMatrix(catID, categoryCount, nil, &matrixQueue,
                                  &isExplicitForm)) {'''
TREE_HOOK_CALL = r'''
#ifdef HYPHY_HIP
  if (_hyhip_defer_expm_hook && ((serial.lLength == 0UL && parallel.lLength) || _hyhip_force_defer_call) &&
      _hyhip_defer_expm_hook(this, catID, nodesToDo, matrixQueue, parallel, isExplicitForm, hasExpForm)) {
    parallel.Clear();  // the device exponentiates these; nothing left for the OpenMP loop below ...
    serial.Clear();    // (mixture template mode: finished matrices of its probe branches — the adapter keeps those branches marked stale)
    if (computedExponentials) {  // ... nor (explicit-form models) for the recombination pass behind it
      DeleteObject(computedExponentials);
      computedExponentials = nil;
    }
  }
#endif
'''


# ---- block 7 (r06): rate classes batched through the host's category loop ------------------------------------------------------------
# likefunc.cpp, Compute(): the category branch asks the batch for the summed log-likelihood and skips SumUpSiteLikelihoods when it got it
CATWANT_ANCHOR = "#ifdef __HYPHYMPI__\n          if (hy_mpi_node_rank == 0) {\n            ComputeSiteLikelihoodsForABlock(partID, siteResults->theData,"
CATWANT = r"""#ifdef HYPHY_HIP
          _hyphy_hip_cat_want(this, partID);
#endif
"""
CATSUM_OLD = "          hyFloat blockResult = SumUpSiteLikelihoods(\n              partID, siteResults->theData, siteScalerBuffer);"
CATSUM_NEW = r"""          hyFloat blockResult = 0.;
#ifdef HYPHY_HIP
          if (!_hyphy_hip_cat_take_logl(this, partID, &blockResult))
#endif
          blockResult = SumUpSiteLikelihoods(
              partID, siteResults->theData, siteScalerBuffer);"""
# likefunc2.cpp copy, PopulateConditionalProbabilities (weighted-sum mode, one category variable, no HMM)
CAT_DECL = r"""
#ifdef HYPHY_HIP
bool _hyphy_hip_cat_begin(const void *lf, long index, long n_classes, const hyFloat *weights, hyFloat *buffer, long *scalers, long block_length);
bool _hyphy_hip_cat_collected(const void *lf, long index, long cls);
bool _hyphy_hip_cat_end(const void *lf, long index);
#endif
"""
CAT_BEGIN_ANCHOR = "  scalers.Populate(arrayDim, 0, 0);\n\n#ifdef __HYPHYMPI__\n  _Vector *computedWeights = nil;"
CAT_BEGIN = r"""  scalers.Populate(arrayDim, 0, 0);

#ifdef HYPHY_HIP
  bool hip_cat_batch = false;
  if (runMode == _hyphyLFConditionProbsWeightedSum && catCount == 0 && hmmCatCount == 0 && !isTrivial && branchIndex < 0 && catWeigths &&
      catWeigths->lLength >= 1UL && totalSteps == categoryCounts->list_data[0])
    hip_cat_batch = _hyphy_hip_cat_begin(this, index, totalSteps, ((_Matrix **)catWeigths->list_data)[0]->theData, buffer, scalers.list_data,
                                         blockLength);
#endif
"""
CAT_SKIP_ANCHOR = "          if (runMode == _hyphyLFConditionProbsWeightedSum ||\n              runMode == _hyphyLFConditionMPIIterate) {\n            long lowerBound = hmmCatCount ? blockLength * currentHMMCat : 0,"
CAT_SKIP = r"""          if (runMode == _hyphyLFConditionProbsWeightedSum ||
              runMode == _hyphyLFConditionMPIIterate) {
#ifdef HYPHY_HIP
            if (hip_cat_batch && _hyphy_hip_cat_collected(this, index, useThisPartitonIndex)) continue;  // (mixed on the device / by the adapter)
#endif
            long lowerBound = hmmCatCount ? blockLength * currentHMMCat : 0,"""
CAT_END_ANCHOR = "#ifdef __HYPHYMPI__\n  DeleteObject(computedWeights);\n#endif\n  DeleteObject(catWeigths);\n}"
CAT_END = r"""#ifdef HYPHY_HIP
  if (hip_cat_batch) _hyphy_hip_cat_end(this, index);
#endif
"""
