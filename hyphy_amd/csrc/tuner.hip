// Run-time schedule tuner of libhyphy_hip.so and the launch of the current schedule (shared with the evaluation path).
#include "partition.h"

namespace hyhip {


// Upload the current schedule to one shard and launch its pruning kernel(s) (no expm, no reduction): the body of an
// evaluation's pruning step, shared with the schedule tuner.
int upload_schedule(hyphy_hip_partition *p, Shard &s) {
  HIPCHK(hipSetDevice(s.device));
  HIPCHK(hipStreamSynchronize(s.stream));
  if (p->ops_host.empty()) return 0;
  memcpy(s.h_ops, p->ops_host.data(), p->ops_host.size() * sizeof(int4));
  HIPCHK(hipMemcpyAsync(s.ops, s.h_ops, p->ops_host.size() * sizeof(int4), hipMemcpyHostToDevice, s.stream));
  for (size_t k = 0; k < p->programs.size(); k++)
    s.h_prog[k] = make_int4(p->programs[k].off, p->programs[k].n, p->programs[k].parent, p->programs[k].need);
  HIPCHK(hipMemcpyAsync(s.prog, s.h_prog, p->programs.size() * sizeof(int4), hipMemcpyHostToDevice, s.stream));
  if (p->chain) {
    memcpy(s.h_jn, p->jn_host.data(), p->jn_host.size() * sizeof(int4));
    HIPCHK(hipMemcpyAsync(s.jn, s.h_jn, p->jn_host.size() * sizeof(int4), hipMemcpyHostToDevice, s.stream));
    if (!s.deposits) {
      const size_t bytes = (size_t)p->C * s.partial_stride * sizeof(double);
      HIPCHK(hipMalloc((void **)&s.deposits, bytes));
      if (getenv("HYPHY_HIP_POISON")) {
        HIPCHK(hipMemset(s.deposits, 0xff, bytes));
        HIPCHK(hipDeviceSynchronize());
      }
    }
  }
  return 0;
}

void launch_prune_current(hyphy_hip_partition *p, Shard &s, int cat, int n_cat_batch) {
  if (p->rr_active && p->chain && s.twins_dirty) refresh_twins(p, s);
  PruneArgs pa = base_prune_args(p, s, cat, n_cat_batch);
  int n_ops = 0;
  for (const auto &pr : p->programs) n_ops = std::max(n_ops, pr.n);
  pa.ops = s.ops;
  pa.n_ops = n_ops;
  pa.n_prog_total = p->chain ? (int)p->I : (int)p->programs.size();
  pa.chain = p->chain ? 1 : 0;
  pa.jn = s.jn;
  pa.deposits = s.deposits;
  if (const char *ab = getenv("HYPHY_HIP_ABLATE")) pa.ablate = atoi(ab);
  for (size_t lv = 0; lv < p->levels.size(); lv++) {
    pa.prog = s.prog + p->levels[lv].first;
    pa.n_prog = p->levels[lv].count;
    pa.do_root = (lv + 1 == p->levels.size()) ? 1 : 0;
    launch_prune_mfma(pa, s.stream);
  }
}

// Schedule tuner.  How a full pass is best cut (level-peeled fragments, or chains with sources of at most m nodes)
// depends on the tree's shape, the shard size and the number of classes in the launch; the pruning pass is idempotent,
// so on the first steady-state full pass (lazy persistence: nothing but the root is stored) the library simply runs
// the pass under each candidate cut on the resident transition matrices, times it with an event pair and keeps the
// fastest (a few milliseconds, once per partition and class-batch mode).  HYPHY_HIP_TUNE=0 or any explicit cut
// (HYPHY_HIP_CHAIN_M / HYPHY_HIP_CUT / HYPHY_HIP_FRAGMENT) disables it.
int tune_schedule(hyphy_hip_partition *p, int cat, int n_cat_batch) {
  p->tuned_for = p->batch_classes;
  const int I = (int)p->I;
  Shard &s = p->shards[0];
  std::vector<int> cand = {-1};
  for (int m : {3, 5, 8, 12, 16, 24, 40, 64})
    if (m < I) cand.push_back(m);
  const int T0 = s.T;
  if (!p->kernel_forced && T0 == 1 && s.ntiles < 2 * s.cus) cand.push_back(-2);  // the row-split workgroup kernel (small shards)
  auto set_kernel = [&](int v) {
    p->variant = v;
    p->n_slots = v >= 1 ? p->n_slots_wave : lds_slots(T0);
  };
  double best_ms = 1e30;
  int best = 0;
  char buf[64];
  std::vector<std::pair<double, int>> ranked;  // (time, chain cut) of the first stage
  p->tune_report.clear();
  for (int c : cand) {
    p->chain_m_forced = c == -2 ? 0 : c;
    set_kernel(c == -2 ? 0 : 1);
    build_schedule(p, nullptr, 0, true);
    if (p->ops_host.size() > ops_capacity(p)) continue;
    if (c > 0 && !p->chain) continue;  // (m >= I: the same as no cut)
    if (upload_schedule(p, s)) return -1;
    float ms = 0.f, ms2 = 0.f;
    launch_prune_current(p, s, cat, n_cat_batch);  // warm-up (instruction cache, schedule in L2)
    HIPCHK(hipEventRecord(s.ev[0], s.stream));
    launch_prune_current(p, s, cat, n_cat_batch);
    HIPCHK(hipEventRecord(s.ev[1], s.stream));
    launch_prune_current(p, s, cat, n_cat_batch);
    HIPCHK(hipEventRecord(s.ev[2], s.stream));
    HIPCHK(hipStreamSynchronize(s.stream));
    HIPCHK(hipGetLastError());
    if (hipEventElapsedTime(&ms, s.ev[0], s.ev[1]) != hipSuccess || hipEventElapsedTime(&ms2, s.ev[1], s.ev[2]) != hipSuccess) continue;
    ms = std::min(ms, ms2);
    snprintf(buf, sizeof buf, "%s%s%d:%.1fus", p->tune_report.empty() ? "" : " ", c == -2 ? "wg-kernel" : (c < 0 ? "levels" : "m"), c < 0 ? 0 : c, 1e3 * ms);
    p->tune_report += buf;
    if (c > 0) ranked.push_back(std::make_pair((double)ms, c));
    if (ms < best_ms) {
      best_ms = ms;
      best = c;
    }
  }
  // second stage: the instantiation compiled for 3 waves per SIMD (finalised node in LDS, no parking slot, no register
  // prefetch of deposits) around the best cut — it wins where waves are plentiful (128 taxa x 100k codons: +7 %)
  int best_wv = 0;
  const double stage1_ms = best_ms;
  p->wave_variant = 0;
  if (best > 0 && p->NW == 4 && !getenv("HYPHY_HIP_WAVE_VARIANT") && !getenv("HYPHY_HIP_SLOTS")) {
    std::vector<int> ms;  // the three fastest cuts of the first stage
    std::sort(ranked.begin(), ranked.end());
    for (size_t k = 0; k < ranked.size() && k < 3; k++) ms.push_back(ranked[k].second);
    for (int m : ms) {
      p->chain_m_forced = m;
      p->variant = 1;
      p->n_slots = 2;
      p->wave_variant = 2;
      build_schedule(p, nullptr, 0, true);
      if (p->ops_host.size() > ops_capacity(p) || !p->chain) continue;
      if (upload_schedule(p, s)) return -1;
      float ms1 = 0.f, ms2 = 0.f;
      launch_prune_current(p, s, cat, n_cat_batch);
      HIPCHK(hipEventRecord(s.ev[0], s.stream));
      launch_prune_current(p, s, cat, n_cat_batch);
      HIPCHK(hipEventRecord(s.ev[1], s.stream));
      launch_prune_current(p, s, cat, n_cat_batch);
      HIPCHK(hipEventRecord(s.ev[2], s.stream));
      HIPCHK(hipStreamSynchronize(s.stream));
      HIPCHK(hipGetLastError());
      if (hipEventElapsedTime(&ms1, s.ev[0], s.ev[1]) != hipSuccess || hipEventElapsedTime(&ms2, s.ev[1], s.ev[2]) != hipSuccess) continue;
      ms1 = std::min(ms1, ms2);
      snprintf(buf, sizeof buf, " occ3/m%d:%.1fus", m, 1e3 * ms1);
      p->tune_report += buf;
      // (2 % margin over the first stage: at equal tuner times the production pass of the 2-waves build is the faster one)
      if (ms1 < 0.98 * stage1_ms && ms1 < best_ms) {
        best_ms = ms1;
        best = m;
        best_wv = 2;
      }
    }
  }
  // third stage: the same tree hung from the node that minimises its height (re-rooted schedules, hyphy_hip_partition::rr_path):
  // shorter critical path per tile, the same work — wins on small and medium shards of unbalanced trees
  bool best_rr = false;
  p->rr_use = false;
  if (best > 0 && !p->rr_path.empty() && n_cat_batch <= 1 && !getenv("HYPHY_HIP_REROOT")) {
    std::vector<int> ms;
    std::sort(ranked.begin(), ranked.end());
    for (size_t k = 0; k < ranked.size() && k < 3; k++) ms.push_back(ranked[k].second);
    size_t best_cand = 0;
    for (size_t ci = 0; ci < p->rr_cands.size(); ci++)
    for (int m : ms) {
      if (p->rr_path != p->rr_cands[ci]) {
        p->rr_path = p->rr_cands[ci];
        for (Shard &sh : p->shards) sh.twins_dirty = true;  // (other twins: refreshed by the transpose kernel before the launch)
      }
      p->chain_m_forced = m;
      p->variant = 1;
      p->wave_variant = best_wv;
      p->n_slots = best_wv == 2 ? 2 : p->n_slots_wave;
      p->rr_use = true;
      build_schedule(p, nullptr, 0, true);
      p->rr_use = false;
      if (p->ops_host.size() > ops_capacity(p) || !p->chain || !p->rr_active) continue;
      if (upload_schedule(p, s)) return -1;
      float ms1 = 0.f, ms2 = 0.f;
      launch_prune_current(p, s, cat, n_cat_batch);
      HIPCHK(hipEventRecord(s.ev[0], s.stream));
      launch_prune_current(p, s, cat, n_cat_batch);
      HIPCHK(hipEventRecord(s.ev[1], s.stream));
      launch_prune_current(p, s, cat, n_cat_batch);
      HIPCHK(hipEventRecord(s.ev[2], s.stream));
      HIPCHK(hipStreamSynchronize(s.stream));
      HIPCHK(hipGetLastError());
      if (hipEventElapsedTime(&ms1, s.ev[0], s.ev[1]) != hipSuccess || hipEventElapsedTime(&ms2, s.ev[1], s.ev[2]) != hipSuccess) continue;
      ms1 = std::min(ms1, ms2);
      snprintf(buf, sizeof buf, " rr%zu/m%d:%.1fus", ci, m, 1e3 * ms1);
      p->tune_report += buf;
      // (5 % margin against the given root: the tuner's pass ranked a re-rooted form of the headline tree 4.5 % ahead that was
      // 1 % behind in production)
      if (ms1 < (best_rr ? 1.0 : 0.95) * best_ms) {
        best_ms = ms1;
        best = m;
        best_rr = true;
        best_cand = ci;
      }
    }
    if (p->rr_path != p->rr_cands[best_cand]) {
      p->rr_path = p->rr_cands[best_cand];
      for (Shard &sh : p->shards) sh.twins_dirty = true;
    }
  }
  p->rr_use = best_rr;
  p->wave_variant = best_wv;
  p->chain_m_forced = best == -2 ? 0 : best;
  set_kernel(best == -2 ? 0 : 1);
  if (best_wv == 2) p->n_slots = 2;
  snprintf(buf, sizeof buf, " -> %s%s%s%d", best_rr ? "rr/" : "", best_wv == 2 ? "occ3/" : "", best == -2 ? "wg-kernel" : (best < 0 ? "levels" : "m"), best < 0 ? 0 : best);
  p->tune_report += buf;
  if (getenv("HYPHY_HIP_VERBOSE")) fprintf(stderr, "[hyphy_hip] schedule tuner (%d classes per launch): %s\n", n_cat_batch, p->tune_report.c_str());
  return 0;
}

}  // namespace hyhip
