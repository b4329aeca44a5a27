"""The host-adapter code that INTEGRATION.md describes, as text blocks that integration/build.py
splices into a COPY of the reference's src/core/likefunc.cpp (never into /root/reference, and the
patched copy is never committed).  Everything here is our own code; the anchors are short unique
strings of the reference file used only to locate the insertion points."""

# ---- block 1: helpers, after the last project #include ---------------------------------------------
HELPERS = r'''
#ifdef HYPHY_HIP
// ===== MI355X likelihood core: host adapter (see INTEGRATION.md) =====================================
#include "hyphy_hip.h"
#include <map>
#include <unordered_map>
#include <cstdlib>
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
};
static std::map<const void *, std::vector<_HyHipPart>> _hyhip_lfs;
static std::map<const void *, std::pair<const void *, long>> _hyhip_tree_owner;  // _TheTree* -> (lf, partition index)
long _hyhip_calls = 0L, _hyhip_cached_calls = 0L, _hyhip_deferred = 0L;
// > 0: ExponentiateMatrices hands the queued rate matrices to the adapter instead of exponentiating them on the host.
// On while Optimize runs (nothing but ComputeBlock reads the transition matrices there; they are brought up to date
// on the host when it returns); HYPHY_HIP_DEVICE_EXPM=0 keeps mode A, =always forces it (LFCompute benchmarks only:
// ancestral reconstruction and simulation read the host matrices between evaluations).
static int _hyhip_defer_depth = 0;
extern bool (*_hyhip_defer_expm_hook)(_TheTree *, long, _List &, _List &, _SimpleList &);  // tree.cpp copy
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
  for (auto &hp : it->second) hyphy_hip_destroy(hp.part);
  _hyhip_lfs.erase(it);
  for (auto o = _hyhip_tree_owner.begin(); o != _hyhip_tree_owner.end();)
    o = o->second.first == lf ? _hyhip_tree_owner.erase(o) : std::next(o);
  if (getenv("HYPHY_HIP_VERBOSE")) fprintf(stderr, "[hyphy_hip] %ld ComputeBlock evaluations ran on the device so far (+ %ld through the branch cache); %ld matrix exponentials moved to the device\n", _hyhip_calls, _hyhip_cached_calls, _hyhip_deferred);
}

static bool _hyphy_hip_defer_handler(_TheTree *t, long catID, _List &nodesToDo, _List &matrixQueue, _SimpleList &parallel);

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
  const char *dv = getenv("HYPHY_HIP_DEVICE");
  int rc = hyphy_hip_create(&hp.part, D, S, L, I, cT->categoryCount, parents.data(), codes.data(),
                            n_amb ? ambigs->theData : nullptr, n_amb, freq.data(), dv ? atoi(dv) : 0, 1);
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
  _hyhip_tree_owner[cT] = std::make_pair(lf, (long)i);
  if (_hyphy_hip_expm_mode() > 0) {
    _hyhip_defer_expm_hook = _hyphy_hip_defer_handler;
    if (_hyphy_hip_expm_mode() == 2 && _hyhip_defer_depth == 0) _hyhip_defer_depth = 1;
  }
}

static bool _hyphy_hip_active(const void *lf, long index) {
  auto it = _hyhip_lfs.find(lf);
  return it != _hyhip_lfs.end() && index < (long)it->second.size() && it->second[index].part != nullptr;
}

// ---- mode B: the host's ExponentiateMatrices hands its queue over instead of exponentiating (tree.cpp copy) ----
static bool _hyphy_hip_defer_handler(_TheTree *t, long catID, _List &nodesToDo, _List &matrixQueue, _SimpleList &parallel) {
  if (_hyhip_defer_depth <= 0) return false;
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
  for (unsigned long id = 0; id < parallel.lLength; id++) {
    const long mid = parallel.get(id);
    _Matrix *m = (_Matrix *)matrixQueue(mid);
    const long code = hp.code_of.at(nodesToDo(mid));
    double *dst = hp.qstash[cat].data() + (size_t)code * DD;
    if (m->is_dense()) {
      memcpy(dst, m->theData, sizeof(double) * DD);
    } else {
      memset(dst, 0, sizeof(double) * DD);
      for (long k = 0; k < m->lDim; k++) {
        const long idx = m->theIndex[k];
        if (idx >= 0 && idx < DD) dst[idx] = m->theData[k];
      }
    }
    hp.q_pending[cat][code] = 1;
    if (!hp.host_stale[cat][code]) {
      hp.host_stale[cat][code] = 1;
      hp.n_stale++;
    }
  }
  _hyhip_deferred += parallel.lLength;
  return true;
}

// bring the host-side transition matrices of one partition up to date (host exponentials of the stashed matrices)
static void _hyphy_hip_flush_part(_HyHipPart &hp, _TheTree *t) {
  if (hp.n_stale == 0) return;
  const long D = t->GetCodeBase(), DD = D * D;
  for (size_t cat = 0; cat < hp.host_stale.size(); cat++)
    for (size_t code = 0; code < hp.host_stale[cat].size(); code++)
      if (hp.host_stale[cat][code]) {
        _Matrix q(D, D, false, true);
        memcpy(q.theData, hp.qstash[cat].data() + code * DD, sizeof(double) * DD);
        ((_CalcNode *)t->GetNodeFromFlatIndex(code))->SetCompExp(&q, hp.cat_arg[cat], true);
        hp.host_stale[cat][code] = 0;
        hp.q_pending[cat][code] = 0;  // (the device receives the probabilities if it has not seen this matrix yet:
      }                               //  _hyphy_hip_compute resends every matrix the host lists)
  hp.n_stale = 0;
}

// one ComputeBlock evaluation on the device; returns 0 when *result is valid
static int _hyphy_hip_compute(const void *lf, long index, _TheTree *t, long catID, _SimpleList &branches,
                              _List &matrices, hyFloat *siteRes, long *scc, hyFloat *result) {
  _HyHipPart &hp = _hyhip_lfs[lf][index];
  const long D = t->GetCodeBase();
  const long B = t->GetLeafCount() + t->GetINodeCount() - 1;
  const long cat = catID < 0 ? 0 : catID;
  long n_q = matrices.lLength;
  const bool first = !hp.cat_seen[cat];
  if (first) n_q = B;  // first evaluation of a rate class: hand over every transition matrix
  hp.pbuf.resize((size_t)n_q * D * D);
  hp.qnodes.resize(n_q);
  long n_pending = 0;
  for (long k = 0; k < n_q; k++) {
    hp.qnodes[k] = first ? k : hp.code_of.at(matrices(k));
    n_pending += hp.q_pending[cat][hp.qnodes[k]];
  }
  if (n_pending > 0 && n_pending < n_q) {  // mixed (rare): exponentiate the stashed ones on the host, send probabilities
    _hyphy_hip_flush_part(hp, t);
    n_pending = 0;
  }
  const bool rate_matrices = n_pending > 0;
  for (long k = 0; k < n_q; k++) {
    if (rate_matrices) {
      memcpy(hp.pbuf.data() + (size_t)k * D * D, hp.qstash[cat].data() + (size_t)hp.qnodes[k] * D * D, sizeof(double) * D * D);
      continue;
    }
    _CalcNode *n = first ? (_CalcNode *)t->GetNodeFromFlatIndex(k) : (_CalcNode *)matrices(k);
    _Matrix *P = n->GetCompExp(catID);
    if (!P || !P->theData) return 1;
    memcpy(hp.pbuf.data() + (size_t)k * D * D, P->theData, sizeof(double) * D * D);
  }
  double ll = 0.;
  int rc = hyphy_hip_evaluate(hp.part, catID, (const int64_t *)branches.list_data, branches.lLength,
                              hp.qnodes.data(), n_q, hp.pbuf.data(), /* q_is_probability = */ rate_matrices ? 0 : 1,
                              t->GetProbs(), &ll, siteRes, (int64_t *)scc);
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

// pinned node states (branchIndex >= 0: marginal ancestral reconstruction, likefunc2.cpp:932-1040): pin, evaluate, unpin
static int _hyphy_hip_pinned(const void *lf, long index, _TheTree *t, long catID, _SimpleList &branches, _List &matrices,
                             long node_code, long const *states, hyFloat *siteRes, long *scc, hyFloat *result) {
  _HyHipPart &hp = _hyhip_lfs[lf][index];
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
  int rc = hyphy_hip_branch_cache_build(hp.part, catID, node);
  if (rc < 0) ReportWarning(_String("hyphy_hip_branch_cache_build: ") & hyphy_hip_last_error());
  return rc;
}
static int _hyphy_hip_cached(const void *lf, long index, _TheTree *t, long catID, long node, hyFloat *siteRes,
                             long *scc, hyFloat *result) {
  _HyHipPart &hp = _hyhip_lfs[lf][index];
  const long cat = catID < 0 ? 0 : catID, DD = t->GetCodeBase() * t->GetCodeBase();
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
          if (_hyphy_hip_cached(this, index, t, catID, doCachedComp - 3, siteRes, scc, &hip_result) == 0) {
            return hip_result;
          }
          return -INFINITY;  // (the error was reported; the host caches were never filled)
        }
        if (_hyphy_hip_compute(this, index, t, catID, *branches, *matrices, siteRes, scc, &hip_result) == 0) {
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
bool (*_hyhip_defer_expm_hook)(_TheTree *, long, _List &, _List &, _SimpleList &) = nullptr;
#endif
'''
TREE_HOOK_CALL = r'''
#ifdef HYPHY_HIP
  if (_hyhip_defer_expm_hook && !hasExpForm && serial.lLength == 0UL && parallel.lLength &&
      _hyhip_defer_expm_hook(this, catID, nodesToDo, matrixQueue, parallel)) {
    parallel.Clear();  // the device exponentiates these; nothing left for the OpenMP loop below
  }
#endif
'''
