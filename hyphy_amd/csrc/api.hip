// C-ABI of libhyphy_hip.so (include/hyphy_hip.h): partition state, pattern sharding over devices, evaluation entry
// points, kernel sequencing.  The schedule compiler lives in schedule.hip, the run-time tuner in tuner.hip, RCCL in
// comm.hip; shared host-side types in partition.h.  Host-side bookkeeping only — all arithmetic of the hot path happens
// in expm.hip / prune.hip / sitefit.hip.
#include "partition.h"

using namespace hyhip;

namespace hyhip {

thread_local std::string g_last_error;

int fail(const std::string &msg) {
  g_last_error = msg;
  return -1;
}

const double kOwnQBuffer = 0.;

namespace {

void free_shard(Shard &s) {
  hipSetDevice(s.device);
  if (s.stream) hipStreamSynchronize(s.stream);
  void *dev[] = {s.codes, s.freq,  s.ambig,  s.partials, s.counts, s.site_lik, s.site_cnt, s.mixed_lik, s.mixed_cnt,
                 s.Pfrag, s.PTg,   s.Prow,   s.qbuf,     s.slots,  s.ops,      s.pi,       s.out,       s.status,
                 s.weights, s.templates, s.templates_pad, s.coeffs, s.wg_sum, s.wg_cnt, s.wg_flag, s.prog, s.frag_ctr, s.hand_cnt, s.pi_ones, s.codes_tile,
                 s.bc_ops, s.bc_prog, s.bc_slot, s.bc_q, s.pin, s.jn, s.deposits, s.mix_q, s.mix_p, s.mix_w, s.mix_off, s.ar_buf, s.fit_Timg, s.fit_bcoef, s.fit_smult, s.fit_smix, s.fit_out, s.fit_scratch,
                 s.fit_pi, s.fit_bgroup, s.fit_scratch_cnt, s.fit_ops, s.rep_tab, s.rep_cnt, s.rep_map, s.rep_desc, s.rep_sync,
                 s.rep_codes_tile, s.rep_leaf, s.rep_walk, s.d_inv, s.expm_need};
  for (void *d : dev)
    if (d) pool_free(d);  // (the stream was synchronised above)
  void *host[] = {s.h_ops, s.h_out, s.h_slots, s.h_small, s.h_coeffs, s.h_prog, s.h_jn, s.h_tstage, s.h_site, s.h_export};
  for (void *h : host)
    if (h) pool_host_free(h);
  for (RepPassSlot &ps : s.rep_slot) {  // (item queues of the lower phase, repeats.hip)
    if (ps.items) pool_free(ps.items);
    if (ps.h_items) pool_host_free(ps.h_items);
    if (ps.ev) hipEventDestroy(ps.ev);
  }
  for (auto &e : s.ev)
    if (e) hipEventDestroy(e);
  for (auto &e : s.ev_ar)
    if (e) hipEventDestroy(e);
  for (auto &e : s.ring)
    if (e) hipEventDestroy(e);
  for (auto &e : s.coeff_ev)
    if (e) hipEventDestroy(e);
  for (auto &e : s.tstage_ev)
    if (e) hipEventDestroy(e);
  if (s.comm && g_rccl.CommDestroy) g_rccl.CommDestroy(s.comm);
  if (s.own_stream) pool_stream_put(s.own_stream);
  s = Shard();
}

int upload_small(Shard &s, const double *src, size_t n, double *dst) {
  if (n > s.h_small_cap) return fail("internal: staging buffer too small");
  HIPCHK(hipStreamSynchronize(s.stream));  // staging buffer reuse
  memcpy(s.h_small, src, n * sizeof(double));
  HIPCHK(hipMemcpyAsync(dst, s.h_small, n * sizeof(double), hipMemcpyHostToDevice, s.stream));
  return 0;
}

// Sequence number for the result record of a synchronous evaluation (0: the caller must wait for the stream).
double next_seq(Shard &s, bool host_record) {
  static const bool spin = !(getenv("HYPHY_HIP_SPIN") && atoi(getenv("HYPHY_HIP_SPIN")) == 0);
  if (!spin || !host_record || !s.d_hout) {
    s.seq_wait = 0.;
    return 0.;
  }
  s.seq_wait = s.seq_next;
  s.seq_next += 1.;
  return s.seq_wait;
}

}  // namespace

// Twin images of re-rooted schedules (ExpmArgs::n_twin): slot of twin j relative to the class base, and the refresh by the
// branch cache's transpose kernel for the cases the exponential kernel did not cover (a writer other than the fused path,
// new root frequencies without a new matrix).
void refresh_twins(hyphy_hip_partition *p, Shard &s) {
  if (p->rr_path.empty() || p->nuc) return;
  const size_t DD = (size_t)p->DP * p->DP;
  for (size_t j = 0; j + 1 < p->rr_path.size(); j++)
    launch_transpose_frag(s.Pfrag + (size_t)p->vw().slot[p->vw().L + p->rr_path[j + 1]] * DD, s.Pfrag + (size_t)(twin_slot0(p) + (int)j) * DD,
                          j == 0 ? s.pi : nullptr, p->NW, s.stream);
  s.twins_dirty = false;
}

// Everything in a PruneArgs that does not depend on the schedule being launched.
// Edge products of the non-last arrivers of a chain schedule: one tile per (class, internal node of the view, tile).  Sized for the
// tree the schedules are cut from — the trunk when the partition runs class-compressed (16 of 62 internal nodes at the headline
// workload, 20 of 127 at 128 x 100 k: 1.0 GB instead of 6.5) — and grown when a larger view needs it.
int ensure_deposits(hyphy_hip_partition *p, Shard &s) {
  const size_t node_stride = (size_t)s.ntiles * 16 * p->DP;
  const size_t class_stride = (size_t)p->vw().I * node_stride;
  const size_t need = (size_t)p->C * class_stride;
  if (s.deposits && s.deposits_cap >= need) {
    s.deposits_class_stride = class_stride;
    return 0;
  }
  if (s.deposits) {
    HIPCHK(hipStreamSynchronize(s.stream));
    pool_free_sync(s.deposits);
    s.dev_bytes -= s.deposits_cap * sizeof(double);
    s.deposits = nullptr;
    s.deposits_cap = 0;
  }
  HIPCHK(pool_malloc((void **)&s.deposits, need * sizeof(double)));
  s.deposits_cap = need;
  s.deposits_class_stride = class_stride;
  s.dev_bytes += need * sizeof(double);
  if (getenv("HYPHY_HIP_POISON")) {
    HIPCHK(hipMemset(s.deposits, 0xff, need * sizeof(double)));
    HIPCHK(hipDeviceSynchronize());
  }
  return 0;
}

PruneArgs base_prune_args(hyphy_hip_partition *p, Shard &s, int cat, int n_cat_batch) {
  const int64_t B = p->B;
  const int DP = p->DP;
  PruneArgs pa;
  pa.ops = nullptr;
  pa.n_ops = 0;
  pa.prog = nullptr;
  pa.n_prog = 1;
  pa.do_root = 1;
  pa.NW = p->NW;
  pa.T = s.T;
  pa.S_pad = s.S_pad;
  pa.ntiles = s.ntiles;
  const hyphy_hip_partition::View &v = p->vw();  // (node indices and leaf numbers are the view's; class strides the partition's)
  pa.root_inode = v.I - 1;
  pa.root_slot = p->root_slot;
  pa.L = v.L;
  pa.variant = p->variant;
  pa.n_slots = p->n_slots;
  pa.codes_in_lds = ((size_t)v.L * s.T * 32 + (size_t)(v.L + v.I) * 16 <= 24576) ? 1 : 0;
  pa.Pfrag = s.Pfrag + (size_t)cat * B * DP * DP;
  pa.PTg = s.PTg + (size_t)cat * B * DP * DP;
  pa.codes = s.codes;
  pa.codes_tile = s.codes_tile;
  if (p->mode == 1) {  // the trunk of a class-compressed partition: generalised leaves (repeats.hip)
    pa.codes_tile = s.rep_codes_tile;
    pa.codes = nullptr;  // (row-major leaf table: the kernels of this mode read the tile-major one)
    pa.leaf_tab = s.rep_leaf;
    pa.gtab = s.rep_tab + (size_t)cat * s.rep_rows * DP;
    pa.gcnt = s.rep_cnt + (size_t)cat * s.rep_rows;
    pa.cs_gtab = (size_t)s.rep_rows * DP;
    pa.cs_gcnt = (size_t)s.rep_rows;
  }
  pa.pin = s.pin;
  pa.pin_leaf = (p->pin_node >= 0 && p->pin_node < p->L) ? (int)p->pin_node : -1;
  pa.pin_inode = p->pin_node >= p->L ? (int)(p->pin_node - p->L) : -1;
  pa.ambig = s.ambig;
  pa.partials = s.partials + (size_t)cat * s.partial_stride;
  pa.counts = s.counts + (size_t)cat * p->I * s.S_pad;
  pa.pi = s.pi;
  pa.site_lik = s.site_lik + (size_t)cat * s.S_pad;
  pa.site_cnt = s.site_cnt + (size_t)cat * s.S_pad;
  pa.freq = s.freq;
  pa.wg_sum = s.wg_sum;
  if (p->rr_active && p->chain && !p->nuc) pa.pi = s.pi_ones;  // (re-rooted schedule: pi is folded into the old root's twin image)
  pa.wg_cnt = s.wg_cnt;
  pa.wg_flag = s.wg_flag;
  pa.n_cat = n_cat_batch;
  pa.cs_P = (size_t)B * DP * DP;
  pa.cs_partials = s.partial_stride;
  pa.cs_counts = (size_t)p->I * s.S_pad;
  pa.cs_site = (size_t)s.S_pad;
  pa.cs_wg = (size_t)s.ntiles / s.T;
  pa.timeline = nullptr;
  pa.ablate = 0;
  pa.frag_ctr = s.frag_ctr;
  pa.hand_cnt = s.hand_cnt;
  pa.n_prog_total = 1;
  pa.wave_variant = getenv("HYPHY_HIP_WAVE_VARIANT") ? atoi(getenv("HYPHY_HIP_WAVE_VARIANT")) : p->wave_variant;
  pa.chain = 0;
  pa.jn = nullptr;
  pa.deposits = nullptr;
  pa.red_out = nullptr;  // (fused final combine: enqueue_eval turns it on for the launch that finalises the roots)
  pa.red_rec = nullptr;
  pa.red_status = nullptr;
  pa.red_seq = 0.;
  pa.red_done = nullptr;
  pa.red_n = 0;
  return pa;
}

namespace {

int enqueue_eval(hyphy_hip_partition *p, Shard &s, int cat, int n_cat_batch, bool sched_changed, bool pi_changed,
                 bool slots_changed, const int64_t *q_nodes, int64_t n_q,
                 const double *q, bool q_on_device, int q_is_prob, const double *root_freqs, double *d_logl_out,
                 bool reduce, bool floor_log, const MixSpec *mix = nullptr) {
  Trace tr("enqueue");
  HIPCHK(hipSetDevice(s.device));
  if (q == &kOwnQBuffer) q = s.qbuf;
  const int64_t D = p->D, B = p->B;
  const int DP = p->DP;
  tr.lap("setdevice");
  if (sched_changed && !p->ops_host.empty()) {
    HIPCHK(hipStreamSynchronize(s.stream));
    memcpy(s.h_ops, p->ops_host.data(), p->ops_host.size() * sizeof(int4));
    HIPCHK(hipMemcpyAsync(s.ops, s.h_ops, p->ops_host.size() * sizeof(int4), hipMemcpyHostToDevice, s.stream));
    for (size_t k = 0; k < p->programs.size(); k++)
      s.h_prog[k] = make_int4(p->programs[k].off, p->programs[k].n, p->programs[k].parent, p->programs[k].need);
    HIPCHK(hipMemcpyAsync(s.prog, s.h_prog, p->programs.size() * sizeof(int4), hipMemcpyHostToDevice, s.stream));
    if (p->chain) {
      memcpy(s.h_jn, p->jn_host.data(), p->jn_host.size() * sizeof(int4));
      HIPCHK(hipMemcpyAsync(s.jn, s.h_jn, p->jn_host.size() * sizeof(int4), hipMemcpyHostToDevice, s.stream));
    }
  }
  if (p->chain && ensure_deposits(p, s)) return -1;
  // root frequencies, zero padded (uploaded only when they change)
  if (pi_changed) {
    std::vector<double> pi(p->nuc ? 4 : DP, 0.0);
    for (int64_t k = 0; k < D; k++) pi[k] = root_freqs[k];
    if (upload_small(s, pi.data(), pi.size(), s.pi)) return -1;
    if (!p->rr_path.empty()) s.twins_dirty = true;  // (pi sits in the twin of the old root's edge; an expm launch that covers every twin clears this again)
  }
  tr.lap("ops+pi");
  if (p->all_timings) HIPCHK(hipEventRecord(s.ev[0], s.stream));
  tr.lap("event0");
  int n_ops_planned = 0;  // longest program: > 0 means a pruning launch follows
  for (const auto &pr : p->programs) n_ops_planned = std::max(n_ops_planned, pr.n);
  ExpmArgs folded_expm;
  bool have_folded = false;
  // 4 states: a schedule that keeps coming back runs as straight-line code compiled at run time (nucgen.hip) — requested after
  // nucgen_after() evaluations under it, used from the evaluation that finds it compiled; the interpreter until then and for
  // everything the generator does not cover (pinned states, the trunk of a class-compressed partition, one-leaf entries)
  bool use_gen = false;
  if (p->nuc && p->mode == 0 && p->nucgen_key != 0 && n_ops_planned > 0 && p->pin_node < 0 && p->nuc_leaf_pairs && p->programs.size() == 1 &&
      s.S_pad % 256 == 0) {
    const int gm = nucgen_mode();
    if (gm != 0) {
      use_gen = nucgen_ready(p->nucgen_key);
      if (!use_gen && (gm == 2 || ++p->nucgen_uses >= nucgen_after()) && !p->nucgen_asked) {
        p->nucgen_asked = true;
        nucgen_request(p->nucgen_key, p->ops_host.data() + p->programs[0].off, p->programs[0].n, (int)p->L, !p->cached_persist, p->nucgen_small, (int)p->B, gm == 2);
        use_gen = nucgen_ready(p->nucgen_key);
      }
    }
  }
  if (n_q > 0) {
    // n_cat_batch > 1: the matrices of ALL rate classes in one expm launch, class-major; destination
    // slot of matrix (c, k) is c*B + q_nodes[k] relative to class 0's image arrays
    const int64_t n_mat = n_q * n_cat_batch;
    int32_t *h_slots = s.h_slots + (size_t)cat * B, *d_slots = s.slots + (size_t)cat * B;
    if (slots_changed) {
      HIPCHK(hipStreamSynchronize(s.stream));
      for (int c = 0; c < n_cat_batch; c++)
        for (int64_t k = 0; k < n_q; k++) {
          if (q_nodes[k] < 0 || q_nodes[k] >= B) return fail("q_nodes entry out of range");
          h_slots[c * n_q + k] = (int32_t)(c * B + q_nodes[k]);
        }
      HIPCHK(hipMemcpyAsync(d_slots, h_slots, n_mat * sizeof(int32_t), hipMemcpyHostToDevice, s.stream));
    }
    if (mix) {
      // explicit-form branch-site mixtures: exponentiate every component, then mix into the branch's matrix images
      // component rate matrices: dense from the host, or (q == the partition's own buffer behind hyphy_hip_build_q) formed inside
      // the exponential kernel from the staged coefficient rows, one row per (branch, component)
      const bool mix_built = q_on_device && q == s.qbuf && p->coeffs_pending;
      if ((q_on_device && !mix_built) || q_is_prob || n_cat_batch != 1)
        return fail("mixture evaluation: host rate matrices or hyphy_hip_build_q rows, one class at a time");
      const size_t n_tot = (size_t)mix->n_tot, DD = (size_t)D * D;
      if (mix_built && s.coeff_rows != (int64_t)n_tot)
        return fail("mixture evaluation: hyphy_hip_build_q staged a different number of rows than the components of this evaluation");
      // (rows are staged without saying what they are: the first evaluation that consumes a staging claims it — one row per (branch,
      //  component) here — and an evaluation of the other kind that happens to need the same number of rows is refused)
      if (mix_built && s.coeff_kind == 1) return fail("mixture evaluation: the staged rows were consumed as one row per (class, branch): call hyphy_hip_build_q first");
      if (mix_built) s.coeff_kind = 2;
      if (s.mix_cap < n_tot || s.mix_nq_cap < (size_t)n_q) {
        HIPCHK(hipStreamSynchronize(s.stream));
        for (void *d : {(void *)s.mix_q, (void *)s.mix_p, (void *)s.mix_w, (void *)s.mix_off})
          if (d) pool_free_sync(d);
        s.mix_q = s.mix_p = s.mix_w = nullptr;
        s.mix_off = nullptr;
        s.mix_cap = std::max(n_tot, (size_t)(2 * B));
        s.mix_nq_cap = (size_t)B;
        HIPCHK(pool_malloc((void **)&s.mix_q, s.mix_cap * DD * sizeof(double)));
        HIPCHK(pool_malloc((void **)&s.mix_p, s.mix_cap * DD * sizeof(double)));
        HIPCHK(pool_malloc((void **)&s.mix_w, s.mix_cap * sizeof(double)));
        HIPCHK(pool_malloc((void **)&s.mix_off, (s.mix_nq_cap + 1) * sizeof(int)));
      }
      std::vector<int> off((size_t)n_q + 1, 0);
      for (int64_t k = 0; k < n_q; k++) off[k + 1] = off[k] + (int)mix->count[k];
      if (!mix_built) HIPCHK(hipMemcpyAsync(s.mix_q, q, n_tot * DD * sizeof(double), hipMemcpyHostToDevice, s.stream));
      HIPCHK(hipMemcpyAsync(s.mix_w, mix->weights, n_tot * sizeof(double), hipMemcpyHostToDevice, s.stream));
      HIPCHK(hipMemcpyAsync(s.mix_off, off.data(), off.size() * sizeof(int), hipMemcpyHostToDevice, s.stream));
      HIPCHK(hipStreamSynchronize(s.stream));  // (pageable sources; `off` goes out of scope)
      ExpmArgs ea;
      ea.Q = s.mix_q;
      ea.slots = nullptr;
      ea.n = (int)n_tot;
      ea.D = (int)D;
      ea.is_prob = 0;
      ea.status = s.status;
      ea.templates = nullptr;
      ea.coeffs = nullptr;
      ea.K = 0;
      if (mix_built) {
        ea.templates = s.templates;
        ea.templates_pad = s.templates_pad;
        ea.coeffs = s.coeffs_cur ? s.coeffs_cur : s.coeffs;
        ea.coeffs_host = (s.coeffs_cur && s.d_hcoeffs) ? s.h_coeffs + (s.coeffs_cur - s.d_hcoeffs) : nullptr;
        ea.K = (int)p->K;
      }
      ea.prof = 0;
      ea.Prow = s.mix_p;
      ea.Pfrag = nullptr;
      ea.PTg = nullptr;
      const bool mix_coeffs_consumed = launch_expm(ea, s.stream);
      if (mix_built && d_logl_out && s.coeff_slot >= 0 && !mix_coeffs_consumed) {  // (asynchronous caller: guard the ring slot)
        HIPCHK(hipEventRecord(s.coeff_ev[s.coeff_slot], s.stream));
        s.coeff_busy[s.coeff_slot] = true;
      }
      s.twins_dirty = true;  // (the mixing kernel writes matrix images without their twins)
      launch_mix_images(s.mix_p, s.mix_off, s.mix_w, d_slots, (int)n_q, (int)D,
                        p->nuc ? nullptr : s.Pfrag + (size_t)cat * B * DP * DP, p->nuc ? nullptr : s.PTg + (size_t)cat * B * DP * DP,
                        p->nuc ? s.Prow + (size_t)cat * B * 16 : nullptr, s.stream,
                        p->nuc ? s.Prow + ((size_t)p->C * B + (size_t)cat * B) * 16 : nullptr);
    } else {
    const double *dq = q;
    const bool q_from_templates = q_on_device && q == s.qbuf && p->coeffs_pending && !q_is_prob;
    if (q_on_device && q == s.qbuf && !q_is_prob) {
      // the partition's own Q buffer is only meaningful behind hyphy_hip_build_q: either the staged coefficients
      // (fused construction) or the materialised matrices, with exactly the rows this evaluation consumes
      if (!q_from_templates && !s.qbuf_built) return fail("evaluate from the Q buffer: no rate matrices staged (call hyphy_hip_build_q first)");
      if (s.coeff_rows != n_mat) return fail("evaluate from the Q buffer: hyphy_hip_build_q staged a different number of matrices than this evaluation consumes");
      if (s.coeff_kind == 2) return fail("evaluate from the Q buffer: the staged rows were consumed as mixture components: call hyphy_hip_build_q first");
      s.coeff_kind = 1;
    }
    if (!q_on_device) {
      HIPCHK(hipMemcpyAsync(s.qbuf, q, (size_t)n_mat * D * D * sizeof(double), hipMemcpyHostToDevice, s.stream));
      dq = s.qbuf;
    }
    ExpmArgs ea;
    ea.Q = dq;
    ea.slots = d_slots;
    ea.n = (int)n_mat;
    ea.D = (int)D;
    ea.is_prob = q_is_prob;
    ea.status = s.status;
    ea.templates = nullptr;
    ea.coeffs = nullptr;
    ea.K = 0;
    ea.prof = getenv("HYPHY_HIP_EXPM_PROF") ? 1 : 0;
    if (q_from_templates) {  // fused build: coefficients were staged by hyphy_hip_build_q
      ea.templates = s.templates;
      ea.templates_pad = s.templates_pad;
      ea.coeffs = s.coeffs_cur ? s.coeffs_cur : s.coeffs;
      ea.coeffs_host = (s.coeffs_cur && s.d_hcoeffs) ? s.h_coeffs + (s.coeffs_cur - s.d_hcoeffs) : nullptr;
      ea.K = (int)p->K;
    }
    if (p->nuc) {
      ea.Prow = s.Prow + (size_t)cat * B * 16;
      ea.PTrow = s.Prow + ((size_t)p->C * B + (size_t)cat * B) * 16;  // (transposed copies behind the row-major ones)
      ea.Pfrag = nullptr;
      ea.PTg = nullptr;
    } else {
      ea.Prow = nullptr;
      ea.Pfrag = s.Pfrag + (size_t)cat * B * DP * DP;
      ea.PTg = s.PTg + (size_t)cat * B * DP * DP;
      static const bool mask_on = !(getenv("HYPHY_HIP_EXPM_MASK") && atoi(getenv("HYPHY_HIP_EXPM_MASK")) == 0);
      if (mask_on && s.expm_need) {
        ea.need = s.expm_need;
        ea.need_B = (int)B;
      }
    }
    if (!p->rr_path.empty() && !p->nuc && n_cat_batch <= 1) {  // keep the transposed twins in step with the matrices they mirror
      const size_t k = p->rr_path.size() - 1;
      ea.n_twin = (int)k;
      for (size_t j = 0; j < k; j++) ea.twin_src[j] = p->vw().slot[p->vw().L + p->rr_path[j + 1]];
      ea.twin_dst0 = twin_slot0(p);
      ea.twin_pi = s.pi;
      size_t covered = 0;
      for (int64_t q = 0; q < n_q; q++)
        for (size_t j = 0; j < k; j++)
          if (q_nodes[q] == ea.twin_src[j]) covered++;
      if (covered == k) s.twins_dirty = false;           // every twin rewritten by this launch (with the current pi)
    }
    tr.lap("slots+q");
    bool coeffs_consumed = false;
    if (p->nuc && p->mode == 0 && (!use_gen || p->nucgen_small) && prune_nuc_folds_expm((int)p->L, s.S_pad, n_ops_planned) && !(q_from_templates && !ea.coeffs)) {
      folded_expm = ea;  // (4 states, small shard: the pruning launch computes the exponentials itself)
      have_folded = true;
    } else {
      coeffs_consumed = launch_expm(ea, s.stream);
    }
    if (q_from_templates && d_logl_out && s.coeff_slot >= 0 && !coeffs_consumed && !have_folded) {  // asynchronous caller: guard the ring slot until the kernel has run
      HIPCHK(hipEventRecord(s.coeff_ev[s.coeff_slot], s.stream));
      s.coeff_busy[s.coeff_slot] = true;
    }
    tr.lap("launch_expm");
    }
  }
  // kernel-duration stamps (an event pair around the pruning launches: two barrier packets, ~5 us of stream time at the
  // headline size): one evaluation in 16 by default — an optimiser's sweep should not pay for a profile nobody reads —,
  // HYPHY_HIP_TIMING_EVERY=n keeps one in n (bench.py: 4, the rocprofv3 runs: 1)
  static const int timing_every = getenv("HYPHY_HIP_TIMING_EVERY") ? std::max(1, atoi(getenv("HYPHY_HIP_TIMING_EVERY"))) : 16;
  const bool stamp = timing_every == 1 || p->all_timings || (s.eval_count++ % (uint64_t)timing_every) == 0;
  s.last_stamped = stamp;
  const size_t ring_slot = (size_t)(s.ring_count % kTimingRing) * 2;
  if (stamp && !s.ring[ring_slot]) {  // (the ring's events are made on first use: a short-lived partition never pays for 2 048 of them)
    HIPCHK(hipEventCreate(&s.ring[ring_slot]));
    HIPCHK(hipEventCreate(&s.ring[ring_slot + 1]));
  }
  if (p->all_timings) HIPCHK(hipEventRecord(s.ev[1], s.stream));  // (before the ring's stamp: the exponentials' interval ends here)
  if (stamp) HIPCHK(hipEventRecord(s.ring[ring_slot], s.stream));
  if (p->mode == 1 && rep_launch(p, s, cat)) return -1;  // lower phase: the class tables this pass recomputes
  int n_ops = 0;  // longest program
  for (const auto &pr : p->programs) n_ops = std::max(n_ops, pr.n);
  double *site_lik = s.site_lik + (size_t)cat * s.S_pad;
  int32_t *site_cnt = s.site_cnt + (size_t)cat * s.S_pad;
  int n_wg = 0;
  bool fused_reduce = false;
  if (p->nuc) {
    NucArgs na;
    na.ops = s.ops + (p->programs.empty() ? 0 : p->programs[0].off);
    na.n_ops = n_ops;
    na.S_pad = s.S_pad;
    na.L = (int)p->L;
    na.root_inode = (int)p->I - 1;
    na.P = s.Prow + (size_t)cat * B * 16;
    na.PT = s.Prow + ((size_t)p->C * B + (size_t)cat * B) * 16;
    na.codes = s.codes;
    na.pin = s.pin;
    na.pin_leaf = (p->pin_node >= 0 && p->pin_node < p->L) ? (int)p->pin_node : -1;
    na.pin_inode = p->pin_node >= p->L ? (int)(p->pin_node - p->L) : -1;
    na.ambig = s.ambig;
    na.partials = s.partials + (size_t)cat * s.partial_stride;
    na.counts = s.counts + (size_t)cat * p->I * s.S_pad;
    na.pi = s.pi;
    na.site_lik = site_lik;
    na.site_cnt = site_cnt;
    na.freq = s.freq;
    na.wg_sum = s.wg_sum;
    na.wg_cnt = s.wg_cnt;
    na.wg_flag = s.wg_flag;
    if (p->mode == 1) {  // the trunk of a class-compressed partition (repeats.hip): generalised leaves, one leaf per entry
      const hyphy_hip_partition::View &v = p->vw();
      na.L = v.L;
      na.root_inode = v.I - 1;
      na.codes = s.rep_codes_tile;   // (4 states: row-major [view leaf][pattern])
      na.PT = nullptr;               // (prune_nuc_kernel)
      na.leaf_tab = s.rep_leaf;
      na.gtab = s.rep_tab + (size_t)cat * s.rep_rows * 4;
      na.gcnt = s.rep_cnt + (size_t)cat * s.rep_rows;
    }
    n_wg = prune_nuc_grid(na);
    {  // fused final combine (see the codon branch below): the small-shard instantiation of the 4-state kernel carries it
      const char *fuse_env = getenv("HYPHY_HIP_FUSED_REDUCE");
      if (!(fuse_env && atoi(fuse_env) == 0) && p->mode == 0 && reduce && n_ops > 0 && !floor_log && n_cat_batch <= 1 && !p->export_sites &&
          (use_gen ? p->nucgen_small : prune_nuc_fuses_reduce(na, have_folded))) {
        double *rec = s.d_hout ? s.d_hout : s.out;
        fused_reduce = true;
        na.red_out = d_logl_out ? d_logl_out : rec;
        na.red_rec = d_logl_out ? s.out + 1 : rec + 1;
        na.red_status = d_logl_out ? nullptr : s.status;
        na.red_seq = next_seq(s, !d_logl_out);
        na.red_done = s.wg_flag + (size_t)p->C * s.wg_cap + 3;  // (the spare words behind the flags; zeroed at creation)
      }
    }
    if (!(use_gen && nucgen_launch(p->nucgen_key, na, s.stream, p->nucgen_small, (int)p->B, have_folded ? &folded_expm : nullptr))) {
      if (use_gen) return fail("internal: the generated 4-state kernel could not be launched");
      launch_prune_nuc(na, s.stream, have_folded ? &folded_expm : nullptr);
    }
    s.last_nucgen = use_gen;
    if (have_folded && folded_expm.templates && d_logl_out && s.coeff_slot >= 0) {  // (the ring slot is read by THIS launch)
      HIPCHK(hipEventRecord(s.coeff_ev[s.coeff_slot], s.stream));
      s.coeff_busy[s.coeff_slot] = true;
    }
  } else {
    if (p->rr_active && p->chain && s.twins_dirty) refresh_twins(p, s);
    PruneArgs pa = base_prune_args(p, s, cat, n_cat_batch);
    pa.ops = s.ops;
    pa.n_ops = n_ops;
    pa.prog = s.prog;
    pa.site_lik = site_lik;
    pa.site_cnt = site_cnt;
    pa.timeline = nullptr;
    pa.ablate = 0;
    if (const char *ab = getenv("HYPHY_HIP_ABLATE")) pa.ablate = atoi(ab);
    const char *tl_path = getenv("HYPHY_HIP_TIMELINE");
    const bool tl_wave = p->variant == 1;  // wave-per-tile kernel: one record of 8 words per wave of the grid
    const size_t tl_waves = (size_t)s.ntiles * std::max(1, n_cat_batch) * std::max<size_t>(1, p->programs.size());
    const size_t tl_n = tl_wave ? tl_waves * 24 : (size_t)kTraceWG * p->NW * std::max(1, n_ops) * 4;
    if (tl_path && n_ops > 0 && s.T == 1) {
      HIPCHK(pool_malloc((void **)&pa.timeline, tl_n * sizeof(long long)));
      HIPCHK(hipMemsetAsync(pa.timeline, 0, tl_n * sizeof(long long), s.stream));
    }
    n_wg = prune_mfma_grid(pa);
    pa.frag_ctr = s.frag_ctr;
    pa.hand_cnt = s.hand_cnt;
    pa.n_prog_total = p->chain ? (int)p->I : (int)p->programs.size();
    pa.chain = p->chain ? 1 : 0;
    pa.jn = s.jn;
    pa.deposits = s.deposits;
    pa.cs_deposits = s.deposits_class_stride;
    // Fused final combine: the launch that finalises the roots also sums the per-tile partial sums and publishes the record —
    // the last root-finalising wave does what wg_reduce_kernel would do in a launch of its own (prune.hip: publish_partial).
    // Saves 1.5-3 us per evaluation of a small shard (below two tiles per CU; 64 x 1 250: 69.1 -> 67.7 us, 32 x 5 000: 79.8 ->
    // 77.0); at the headline size the gain shrinks to ~1 us while the pruning kernel's own duration grows by the 3-4 us of the
    // serial tail, so larger shards keep the separate kernel.  HYPHY_HIP_FUSED_REDUCE=0/1 forces either.
    const char *fuse_env = getenv("HYPHY_HIP_FUSED_REDUCE");
    const bool walk = trunk_walk_applies(p, s) && !pa.timeline && n_ops > 0;  // (repeats.hip: the trunk as one row-split walk per tile)
    const bool fuse_on = fuse_env ? atoi(fuse_env) != 0 : s.ntiles <= 2 * s.cus;  // (the walk at the headline: 90.0 / 89.8 us fused, 88.3 / 89.5 not)
    if (fuse_on && reduce && n_ops > 0 && !floor_log && n_cat_batch <= 1 && !pa.timeline && !p->export_sites &&
        (walk ? (trunk_walk_fuses_reduce(p) && !getenv("HYPHY_HIP_WALK_TIMELINE")) : prune_fuses_reduce(pa))) {
      double *rec = s.d_hout ? s.d_hout : s.out;
      fused_reduce = true;
      pa.red_out = d_logl_out ? d_logl_out : rec;
      pa.red_rec = d_logl_out ? s.out + 1 : rec + 1;
      pa.red_status = d_logl_out ? nullptr : s.status;
      pa.red_seq = next_seq(s, !d_logl_out);
      pa.red_done = s.frag_ctr + (size_t)p->C * (p->I + 2) * s.ntiles - 1;  // (a word of the arrival counters no schedule indexes; zero between launches)
      pa.red_n = n_wg;
    }
    double *const red_out = pa.red_out;
    if (walk) {
      // the trunk of a class-compressed partition, lazy full pass: one row-split walk per tile instead of the schedule
      pa.red_out = red_out;
      if (launch_trunk_walk(p, s, cat, n_cat_batch, true, &pa)) return -1;
      s.last_walk = true;
    } else {
    s.last_walk = false;
    for (size_t lv = 0; lv < p->levels.size(); lv++) {  // one launch per level of subtree fragments
      pa.prog = s.prog + p->levels[lv].first;
      pa.n_prog = p->levels[lv].count;
      pa.do_root = (lv + 1 == p->levels.size()) ? 1 : 0;
      pa.red_out = pa.do_root ? red_out : nullptr;
      launch_prune_mfma(pa, s.stream);
    }
    }
    if (pa.timeline) {  // tracing only: synchronous dump of the per-entry s_memtime stamps
      std::vector<long long> h(tl_n);
      HIPCHK(hipStreamSynchronize(s.stream));
      HIPCHK(hipMemcpy(h.data(), pa.timeline, tl_n * sizeof(long long), hipMemcpyDeviceToHost));
      pool_free_sync(pa.timeline);
      if (tl_wave) {
        if (FILE *f = fopen(tl_path, "w")) {
          fprintf(f, "# wave t_start t_prologue t_program t_end levels how hw_id xcc_id   (100 MHz ticks; grid = %s)  then shader cycles: "
                     "16 phase buckets (prune.hip HYPHY_TR)\n",
                  p->chain ? "tiles x classes x sources" : "programs x classes x tiles");
          for (size_t k = 0; k < tl_waves; k++) {
            const long long *r = &h[k * 24];
            fprintf(f, "%zu", k);
            for (int i = 0; i < 24; i++) fprintf(f, " %lld", r[i]);
            fprintf(f, "\n");
          }
          fclose(f);
        }
      } else if (FILE *f = fopen(tl_path, "w")) {
        fprintf(f, "# wg wave entry flags t_start t_compute_done t_after_barrier t_finalised\n");
        for (int b = 0; b < kTraceWG; b++)
          for (int w = 0; w < p->NW; w++)
            for (int o = 0; o < n_ops; o++) {
              const long long *r = &h[(((size_t)b * p->NW + w) * n_ops + o) * 4];
              fprintf(f, "%d %d %d %d %lld %lld %lld %lld\n", b, w, o, o < (int)p->ops_host.size() ? (p->ops_host[o].x & 0xff) : 0, r[0], r[1], r[2], r[3]);
            }
        fclose(f);
      }
    }
  }
  tr.lap("launch_prune");
  if (stamp) {
    HIPCHK(hipEventRecord(s.ring[ring_slot + 1], s.stream));
    s.ring_count++;
  }
  if (p->all_timings) HIPCHK(hipEventRecord(s.ev[2], s.stream));
  s.exported = 0;
  if (p->export_sites && reduce && !floor_log && n_cat_batch <= 1 && s.d_export) {
    // per-pattern results for the caller, in ITS order, into host-mapped memory — in front of the kernel that publishes the result
    // record, so that the record's sequence word says they have landed too
    launch_site_export(site_lik, site_cnt, s.d_inv, (int)s.S, (p->export_sites & 1) ? s.d_export : nullptr,
                       (p->export_sites & 2) ? reinterpret_cast<long long *>(s.d_export + s.S) : nullptr, s.stream);
    s.exported = p->export_sites;
  }
  if (reduce && !fused_reduce) {
    // synchronous entry points: the result record goes straight to host-mapped pinned memory (a
    // posted PCIe write from the kernel) — an SDMA device-to-host copy after the kernels costs far more
    double *rec = s.d_hout ? s.d_hout : s.out;
    double *o0 = d_logl_out ? d_logl_out : rec, *o1 = d_logl_out ? s.out + 1 : rec + 1;
    const int *st = d_logl_out ? nullptr : s.status;
    const double seq = next_seq(s, !d_logl_out);
    if (n_ops > 0 && !floor_log)  // the pruning kernel left per-workgroup partial sums
      launch_wg_reduce(s.wg_sum, s.wg_cnt, s.wg_flag, n_wg, o0, o1, st, s.stream, seq);
    else  // nothing was recomputed (or category mode): reduce the stored per-pattern values
      launch_site_reduce(site_lik, site_cnt, s.freq, s.S_pad, floor_log ? 1 : 0, o0, o1, st, s.stream, seq);
  }
  if (p->all_timings) HIPCHK(hipEventRecord(s.ev[3], s.stream));
  HIPCHK(hipGetLastError());
  tr.lap("reduce+events");
  return 0;
}

bool same_update(const hyphy_hip_partition *p, const int64_t *u, int64_t n, bool full) {
  if (!p->cached_valid) return false;
  if (full && p->cached_full) return true;
  if (full != p->cached_full) return false;
  if ((int64_t)p->cached_update.size() != n) return false;
  return n == 0 || memcmp(p->cached_update.data(), u, n * sizeof(int64_t)) == 0;
}

int prepare_schedule(hyphy_hip_partition *p, int cat, const int64_t *update_nodes, int64_t n_update, bool *changed,
                     bool force_persist, const int64_t *q_nodes, int64_t n_q, int n_classes) {
  bool full = !p->initialized[cat];
  if (!full && n_update >= p->B) full = true;
  const bool requested_full = full;
  bool persist_all = true;
  const std::vector<char> &res = p->mode == 1 ? p->rep_resident : p->resident;  // (each view keeps its own copies)
  if (!full && !res[cat]) full = true;  // a partial update needs current persisted copies: promote to a persisting full pass
  else if (full && p->initialized[cat] && p->cache_policy == 1 && p->last_full[cat] && !force_persist)
    persist_all = false;
  p->sched_full = full;
  p->sched_persist = persist_all;
  p->last_full[cat] = requested_full ? 1 : 0;
  std::vector<int64_t> view_update;
  if (p->mode == 1) {  // class tables to recompute (item queues on the device), and the trunk's own update list
    if (rep_prepare_pass(p, update_nodes, n_update, q_nodes, n_q, full, cat, n_classes, view_update)) return -1;
  }
  if (same_update(p, update_nodes, n_update, full) && p->cached_persist == persist_all) {
    *changed = false;
    return 0;
  }
  if (p->mode == 1) build_schedule(p, view_update.data(), (int64_t)view_update.size(), full);
  else build_schedule(p, update_nodes, n_update, full);
  if (p->ops_host.size() > ops_capacity(p)) return fail("internal: schedule overflow");
  p->cached_update.assign(update_nodes, update_nodes + (full ? 0 : n_update));
  p->cached_full = full;
  p->cached_persist = persist_all;
  p->cached_valid = 1;
  *changed = true;
  // (nucgen.hip) the key of this schedule's generated kernel: full passes of a 4-state partition under its own tree
  // (small-shard form — matrices in LDS, exponentials and final combine inside the launch — for shards of at most two workgroups
  //  per CU: the interpreter's LP rule; HYPHY_HIP_NUCGEN_SMALL=0/1 forces either)
  p->nucgen_small = !p->shards.empty() && p->shards[0].S_pad / 256 <= 2 * p->shards[0].cus && p->B <= 1024;
  if (const char *e = getenv("HYPHY_HIP_NUCGEN_SMALL")) p->nucgen_small = atoi(e) != 0 && p->B <= 1024;
  p->nucgen_key = (p->nuc && p->mode == 0 && full && p->programs.size() == 1 && p->L <= 256)
                      ? nucgen_key(p->ops_host.data() + p->programs[0].off, p->programs[0].n, (int)p->L, !persist_all, p->nucgen_small, (int)p->B) : 0;
  p->nucgen_uses = 0;
  p->nucgen_asked = false;
  return 0;
}

}  // namespace

// Wait for every shard and read its host-mapped result record [log-L, scaler sum, status].
int collect_status(hyphy_hip_partition *p) {
  static const bool expm_prof = getenv("HYPHY_HIP_EXPM_PROF") != nullptr;
  for (Shard &s : p->shards) {
    HIPCHK(hipSetDevice(s.device));
    if (!s.d_hout) HIPCHK(hipMemcpyAsync(s.h_out, s.out, 3 * sizeof(double), hipMemcpyDeviceToHost, s.stream));
    if (s.seq_wait != 0. && s.d_hout && !expm_prof) {
      // the last kernel of the evaluation publishes the sequence number after the record (system-scope fence)
      volatile double *flag = s.h_out + 3;
      bool seen = false;
      for (long spins = 1; !seen; spins++) {
        seen = *flag == s.seq_wait;
        if (!seen && (spins & 0x3fff) == 0) {  // every few tens of microseconds: has the stream died or finished?
          const hipError_t q = hipStreamQuery(s.stream);
          if (q == hipSuccess) {
            seen = *flag == s.seq_wait;
            break;
          }
          if (q != hipErrorNotReady) return fail(std::string("stream failed: ") + hipGetErrorString(q));
        }
      }
      s.seq_wait = 0.;
      if (!seen) HIPCHK(hipStreamSynchronize(s.stream));
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
    } else {
      HIPCHK(hipStreamSynchronize(s.stream));
    }
    if (expm_prof) {
      long long t[8];
      expm_read_profile(t);
      fprintf(stderr, "[hyphy_hip] expm phases (cycles): Q build %lld, norms %lld, Taylor %lld, squarings %lld, store %lld, images %lld\n",
              t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[6] - t[5]);
    }
    if (s.h_out[2] != 0.) {
      hipMemsetAsync(s.status, 0, sizeof(int32_t), s.stream);
      return fail("Failed to compute a valid transition matrix; this is usually caused by ill-conditioned rate "
                  "matrices (e.g. very large rate values)");
    }
  }
  return 0;
}

void record_timings(hyphy_hip_partition *p) {
  Shard &s = p->shards[0];
  float t;
  for (int k = 0; k < 3; k++) {
    t = 0.f;
    if (k == 1) {  // the pruning interval of THIS evaluation, or 0 when it carried no stamp (one evaluation in
                   // HYPHY_HIP_TIMING_EVERY does; hyphy_hip_set_timing_detail stamps every one) — never an older evaluation's
      p->timings[1] = 0.;
      if (s.last_stamped && s.ring_count > 0) {
        const size_t slot = (size_t)((s.ring_count - 1) % kTimingRing) * 2;
        if (hipEventElapsedTime(&t, s.ring[slot], s.ring[slot + 1]) == hipSuccess) p->timings[1] = t;
      }
      continue;
    }
    if (!p->all_timings) continue;
    if (hipEventElapsedTime(&t, s.ev[k], s.ev[k + 1]) == hipSuccess) p->timings[k] = t;
  }
}

// sum of shard partials exactly as ComputeBlock combines its thread blocks (Neumaier,
// likefunc.cpp:11046-11093)
double combine(const std::vector<double> &parts) {
  if (parts.size() == 1) return parts[0];
  double sum = 0., corr = 0.;
  for (double r : parts) {
    if (r != r) return r;
    if (r == -INFINITY) return -INFINITY;
    double t = sum + r;
    if (sum < r) corr += (sum - t) + r;
    else corr += (r - t) + sum;
    sum = t;
  }
  return sum + corr;
}

namespace {

// A synchronous evaluation that returns per-pattern results asks for the export in front of its enqueue (single-device partitions;
// HYPHY_HIP_SITE_EXPORT=0 keeps the copies of gather_sites) ...
int begin_site_export(hyphy_hip_partition *p, bool want_lik, bool want_cnt) {
  if (!p) return 0;
  const char *env = getenv("HYPHY_HIP_SITE_EXPORT");  // (read per call: the tests run both paths in one process)
  const bool on = !(env && atoi(env) == 0);
  p->export_sites = 0;
  if (!on || p->shards.size() != 1 || !(want_lik || want_cnt)) return 0;
  Shard &s = p->shards[0];
  if (s.S <= 0 || s.s0 != 0) return 0;
  HIPCHK(hipSetDevice(s.device));
  if (!s.h_export && !s.export_failed) {
    // (the pattern-order map first — a partition whose patterns are sorted must never export without it, and a failed
    //  allocation must not leave a half-initialised export behind: on any failure this partition keeps the copies of gather_sites)
    int32_t *d_inv = nullptr;
    double *h_export = nullptr, *d_export = nullptr;
    bool ok = true;
    if (!p->perm.empty()) {
      std::vector<int32_t> inv((size_t)s.S, 0);
      for (int64_t j = 0; j < s.S; j++) inv[(size_t)p->perm[(size_t)j]] = (int32_t)j;
      ok = pool_malloc((void **)&d_inv, inv.size() * sizeof(int32_t)) == hipSuccess &&
           hipMemcpy(d_inv, inv.data(), inv.size() * sizeof(int32_t), hipMemcpyHostToDevice) == hipSuccess;
    }
    ok = ok && pool_host_malloc((void **)&h_export, (size_t)s.S * 2 * sizeof(double)) == hipSuccess &&
         hipHostGetDevicePointer((void **)&d_export, h_export, 0) == hipSuccess && d_export;
    if (!ok) {
      (void)hipGetLastError();
      if (d_inv) pool_free_sync(d_inv);
      if (h_export) pool_host_free(h_export);
      s.export_failed = true;
      return 0;
    }
    s.d_inv = d_inv;
    s.h_export = h_export;
    s.d_export = d_export;
  }
  if (!s.d_export || (!p->perm.empty() && !s.d_inv)) return 0;
  p->export_sites = (want_lik ? 1 : 0) | (want_cnt ? 2 : 0);
  return 0;
}
int gather_sites(hyphy_hip_partition *p, int cat, double *site_lik_out, int64_t *site_scaler_out, bool mixed);
// ... and collects them behind the wait for the result record (whatever was not exported comes through gather_sites)
int finish_sites(hyphy_hip_partition *p, int cat, double *site_lik_out, int64_t *site_scaler_out) {
  p->export_sites = 0;
  if (!site_lik_out && !site_scaler_out) return 0;
  Shard &s = p->shards[0];
  const int have = p->shards.size() == 1 ? s.exported : 0;
  s.exported = 0;
  const bool lik_ok = !site_lik_out || (have & 1), cnt_ok = !site_scaler_out || (have & 2);
  if (!(lik_ok && cnt_ok)) return gather_sites(p, cat, site_lik_out, site_scaler_out, false);
  if (site_lik_out) memcpy(site_lik_out, s.h_export, (size_t)s.S * sizeof(double));
  if (site_scaler_out) memcpy(site_scaler_out, s.h_export + s.S, (size_t)s.S * sizeof(int64_t));
  return 0;
}

int gather_sites(hyphy_hip_partition *p, int cat, double *site_lik_out, int64_t *site_scaler_out, bool mixed) {
  for (Shard &s : p->shards) {
    HIPCHK(hipSetDevice(s.device));
    const double *lik = mixed ? s.mixed_lik : s.site_lik + (size_t)cat * s.S_pad;
    const int32_t *cn = mixed ? s.mixed_cnt : s.site_cnt + (size_t)cat * s.S_pad;
    // both arrays through ONE pinned staging block and one wait (r04; was: two pageable temporaries, a wait behind each copy —
    // a host that mixes rate classes itself asks for these once per class and evaluation)
    if (!s.h_site) HIPCHK(pool_host_malloc((void **)&s.h_site, (size_t)s.S_pad * (sizeof(double) + sizeof(int32_t))));
    double *hl = s.h_site;
    int32_t *hc = reinterpret_cast<int32_t *>(s.h_site + s.S_pad);
    if (site_lik_out) HIPCHK(hipMemcpyAsync(hl, lik, s.S * sizeof(double), hipMemcpyDeviceToHost, s.stream));
    if (site_scaler_out) HIPCHK(hipMemcpyAsync(hc, cn, s.S * sizeof(int32_t), hipMemcpyDeviceToHost, s.stream));
    HIPCHK(hipStreamSynchronize(s.stream));
    if (site_lik_out) {
      if (p->perm.empty()) memcpy(site_lik_out + s.s0, hl, s.S * sizeof(double));
      else  // the device keeps its patterns sorted: scatter back into the caller's order
        for (int64_t k = 0; k < s.S; k++) site_lik_out[p->perm[s.s0 + k]] = hl[k];
    }
    if (site_scaler_out)
      for (int64_t k = 0; k < s.S; k++) site_scaler_out[caller_pattern(p, s.s0 + k)] = hc[k];
  }
  return 0;
}

}  // namespace

}  // namespace hyhip

extern "C" {

const char *hyphy_hip_last_error(void) { return g_last_error.c_str(); }

int hyphy_hip_prune_launches(hyphy_hip_partition *p) { return p ? (int)std::max<size_t>(1, p->levels.size()) : 0; }

const char *hyphy_hip_prune_kernel_name(const hyphy_hip_partition *p) {
  if (!p) return "";
  if (p->nuc && !p->shards.empty() && p->shards[0].last_nucgen) return "nucgen_kernel";  // (run-time generated, nucgen.hip)
  if (p->nuc) return (p->nuc_leaf_pairs && p->mode == 0) ? "prune_nuc2_kernel" : "prune_nuc_kernel";
  if (p->mode == 1 && !p->shards.empty() && p->shards[0].last_walk) return "trunk_walk_kernel";  // (repeats.hip)
  return p->variant == 1 ? "prune_wave_kernel" : "prune_mfma_kernel";  // (variant 2: the same kernel on a chain schedule)
}

int64_t hyphy_hip_prune_timings(hyphy_hip_partition *p, double *out_ms, int64_t n) {
  if (!p || !out_ms || n <= 0 || p->shards.empty()) return 0;
  Shard &s = p->shards[0];
  if (hipSetDevice(s.device) != hipSuccess || hipStreamSynchronize(s.stream) != hipSuccess) return 0;
  const int64_t have = (int64_t)std::min<uint64_t>(s.ring_count, (uint64_t)kTimingRing);
  const int64_t m = std::min(n, have);
  for (int64_t k = 0; k < m; k++) {
    const uint64_t idx = s.ring_count - (uint64_t)m + (uint64_t)k;
    const size_t slot = (size_t)(idx % kTimingRing) * 2;
    float t = 0.f;
    if (hipEventElapsedTime(&t, s.ring[slot], s.ring[slot + 1]) != hipSuccess) t = 0.f;
    out_ms[k] = t;
  }
  return m;
}
const char *hyphy_hip_version(void) { return "hyphy_hip 0.1 (gfx950, FP64 MFMA)"; }

int hyphy_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

void hyphy_hip_destroy(hyphy_hip_partition *p) {
  if (!p) return;
  for (Shard &s : p->shards) free_shard(s);
  if (p->h_qstage) { hipDeviceSynchronize(); pool_host_free(p->h_qstage); }
  if (p->xch) hyphy_hip_xch_close(p->xch);
  delete p;
}

int hyphy_hip_create(hyphy_hip_partition **out, int64_t D, int64_t S, int64_t L, int64_t I, int64_t C,
                     const int64_t *flat_parents, const int64_t *leaf_codes, const double *ambig, int64_t n_ambig,
                     const int64_t *pattern_freq, int device_first, int device_count) {
  if (!out) return fail("out == NULL");
  *out = nullptr;
  if (D < 2 || D > 64) {
    g_last_error = "unsupported state count (this version: 2 <= D <= 64)";
    return 1;
  }
  if (S < 1 || L < 2 || I < 1 || C < 1 || !flat_parents || !leaf_codes || !pattern_freq)
    return fail("invalid partition dimensions / null input");
  if (L > 65535) {
    g_last_error = "more than 65535 leaves: packed schedule entries unsupported in this version";
    return 1;
  }
  if (n_ambig > 32767) {
    g_last_error = "too many distinct ambiguity vectors for the packed leaf table";
    return 1;
  }
  int ndev = hyphy_hip_device_count();
  if (ndev <= 0) return fail("no HIP device available (this library has no CPU fallback)");
  int force = 0;
  if (const char *e = getenv("HYPHY_HIP_FORCE_SHARDS")) force = atoi(e);
  if (device_count < 1) device_count = 1;
  if (!force && (device_first < 0 || device_first + device_count > ndev)) return fail("device range out of bounds");
  if (force > 0 && device_first >= ndev) return fail("device out of bounds");
  int nshards = force > 0 ? force : device_count;
  if ((int64_t)nshards > S) nshards = (int)S;

  hyphy_hip_partition *p = new hyphy_hip_partition();
  p->D = D; p->S = S; p->L = L; p->I = I; p->C = C; p->B = L + I - 1;
  p->nuc = (D == 4);
  p->NW = (int)((D + 15) / 16);
  p->DP = 16 * p->NW;
  p->parents.assign(flat_parents, flat_parents + L + I);
  p->children.assign(I, std::vector<int>());
  int roots = 0;
  for (int64_t n = 0; n < L + I; n++) {
    int64_t par = flat_parents[n];
    if (par < 0) { roots++; continue; }
    if (par >= I || (n >= L && par <= n - L)) { delete p; return fail("flat_parents is not a post-order tree"); }
    p->children[par].push_back((int)n);
  }
  if (roots != 1 || flat_parents[L + I - 1] != -1) { delete p; return fail("root must be the last internal node"); }
  for (int64_t k = 0; k < L * S; k++)
    if (leaf_codes[k] >= D || leaf_codes[k] < -n_ambig) { delete p; return fail("leaf code out of range"); }
  p->initialized.assign(C, 0);
  p->bc_node.assign(C, -1);
  p->bc_use_pi.assign(C, 0);
  p->resident.assign(C, 0);
  p->last_full.assign(C, 0);
  if (const char *e = getenv("HYPHY_HIP_CACHE")) p->cache_policy = strcmp(e, "always") == 0 ? 0 : 1;
  p->leaf_has_ambig.assign(L, 0);
  for (int64_t l = 0; l < L; l++)
    for (int64_t k = 0; k < S; k++)
      if (leaf_codes[l * S + k] < 0) {
        p->leaf_has_ambig[l] = 1;
        break;
      }

  const int DP = p->DP;
  const int64_t B = p->B;
  int tiles_override = 0;
  if (const char *e = getenv("HYPHY_HIP_TILES")) tiles_override = atoi(e);
  if (!p->nuc && !(getenv("HYPHY_HIP_SORT_PATTERNS") && atoi(getenv("HYPHY_HIP_SORT_PATTERNS")) == 0))
    sort_patterns(p, leaf_codes, L, S);
  init_plain_view(p);
  if (!p->nuc && C == 1) reroot_path(p);
  auto src_pattern = [&](int64_t j) -> int64_t { return p->perm.empty() ? j : p->perm[j]; };

  int64_t base = S / nshards, rem = S % nshards, s0 = 0;
  for (int k = 0; k < nshards; k++) {
    Shard s;
    s.device = force > 0 ? device_first : device_first + k;
    s.s0 = s0;
    s.S = base + (k < rem ? 1 : 0);
    s0 += s.S;
    p->shards.push_back(s);
  }
  std::vector<std::vector<int16_t>> shard_codes;  // per shard: the leaf table in device pattern order (class computation below)
  for (Shard &s : p->shards) {
    if (hipSetDevice(s.device) != hipSuccess) { hyphy_hip_destroy(p); return fail("hipSetDevice failed"); }
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, s.device);
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
      hyphy_hip_destroy(p);
      return fail(std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    }
    const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const int64_t tiles = (s.S + 15) / 16;
    int T = 1;
    if (!p->nuc) {
      // T = 1 everywhere the wave-per-tile kernel applies (any shard of >= 1.75 tiles per CU: measured 2.3x
      // faster than the T = 2 workgroup kernel at 128 taxa x 100k codons, 2 718 vs 6 190 us); T > 1 only beyond
      // the grid limit of 65 535 tiles
      T = tiles <= 65535 ? 1 : (tiles <= 2 * 65535 ? 2 : 4);
      if (tiles_override >= 1 && tiles_override <= 4) T = tiles_override;
    }
    s.T = T;
    s.cus = cus;
    if (p->nuc) p->n_slots = 2 + kNucParkSlots;  // (prune_nuc_kernel parks pending nodes in LDS)
    if (p->nuc) p->nuc_leaf_pairs = prune_nuc_takes_leaf_pairs((int)L);
    if (!p->nuc) {
      // Kernel choice (measured, tools/sweep_small_shards.sh): the wave-per-tile kernel (no cross-wave
      // exchange, child -> parent through registers) wins once every SIMD holds ~2 waves of it — 160 vs 183 us
      // at 624 tiles; below that its 64-MFMA-per-edge chains are pure latency (167 us at 312 tiles as at
      // 624) and the workgroup-per-tile kernel, which splits a tile's rows over four waves, is faster
      // (123 us at 312 tiles, 63 us at 78).
      // (r02: with chain schedules the wave-per-tile kernel also wins at a rank's share of the headline workload —
      //  89 vs 118 us at 312 tiles, 52 vs 61 us at 78; below ~a quarter tile per CU the row-split workgroup kernel
      //  keeps its shorter critical path.  The schedule tuner re-checks the choice on the first steady-state pass.)
      p->variant = tiles >= (int64_t)cus / 4 ? 1 : 0;
      p->kernel_forced = false;
      if (const char *e = getenv("HYPHY_HIP_KERNEL")) {  // (diagnostic override)
        p->variant = std::max(0, std::min(2, atoi(e)));  // 0: workgroup per tile, 1: wave per tile, 2: workgroup per tile on chain schedules
        p->kernel_forced = true;
      }
      if (T != 1) {
        p->variant = 0;
        p->kernel_forced = true;
      }
      // two "exchange" ids (register hand-over) + wave-private LDS parking slots: two where two tiles + the leaf codes fit an
      // eighth of the CU's 160 KiB (eight waves per CU stay resident: 61 states up to 128 taxa), else one
      {
        const size_t tile_bytes = (size_t)p->NW * 4 * 64 * sizeof(double), codes_bytes = (size_t)p->L * 16 * sizeof(int16_t);
        p->n_slots_wave = 2 * tile_bytes + codes_bytes <= 20480 ? 4 : 3;
      }
      if (const char *e = getenv("HYPHY_HIP_SLOTS")) p->n_slots_wave = std::max(2, std::min(4, atoi(e)));
      p->n_slots = p->variant == 1 ? p->n_slots_wave : lds_slots(T);
    }
    if (p->nuc) {
      s.S_pad = (int)((s.S + 511) / 512 * 512);  // (prune_nuc2_kernel: 256 threads x 2 patterns per workgroup)
      s.ntiles = 0;
      s.partial_stride = (size_t)I * 4 * s.S_pad;
    } else {
      s.ntiles = (int)((tiles + T - 1) / T * T);
      s.S_pad = s.ntiles * 16;
      s.partial_stride = (size_t)I * s.ntiles * 16 * DP;
    }
    // HYPHY_HIP_POISON=1 (tests): fill every fresh allocation with 0xff bytes (NaN / -1), so that anything read before
    // it is written shows up even in a process whose recycled device memory happens to hold zeros
    const bool poison = getenv("HYPHY_HIP_POISON") != nullptr;
#define A_(ptr, n)                                                                                           \
  {                                                                                                          \
    if (pool_malloc((void **)&(ptr), (n)) != hipSuccess) {                                                     \
      hyphy_hip_destroy(p);                                                                                  \
      return fail("hipMalloc failed (" #ptr ")");                                                            \
    }                                                                                                        \
    s.dev_bytes += (size_t)(n);                                                                              \
    if (poison && !getenv("HYPHY_HIP_NOPOISON_" #ptr)) {                                                     \
      hipMemset((ptr), 0xff, (n));                                                                           \
      hipDeviceSynchronize();                                                                                \
    }                                                                                                        \
  }
    pool_stream_get(&s.own_stream);
    s.stream = s.own_stream;
    for (auto &e : s.ev) hipEventCreate(&e);
    s.ring.assign(2 * kTimingRing, nullptr);
    A_(s.codes, (size_t)L * s.S_pad * sizeof(int16_t));
    if (!p->nuc) {
      A_(s.codes_tile, (size_t)L * s.S_pad * sizeof(int16_t));
      {  // which matrix images a branch's consumers read (ExpmArgs::need): leaves are gathered from by state — the column-gather
         // image —, internal branches are A operands of edge products; a leaf with ambiguity codes is both
        std::vector<unsigned char> need((size_t)B, 3);
        for (int64_t c = 0; c < L && c < B; c++) need[(size_t)c] = p->leaf_has_ambig[(size_t)c] ? 3 : 2;
        for (int64_t c = L; c < L + I - 1 && c < B; c++) need[(size_t)c] = 1;
        A_(s.expm_need, (size_t)B);
        hipMemcpy(s.expm_need, need.data(), (size_t)B, hipMemcpyHostToDevice);
      }
      A_(s.bc_ops, ops_capacity(p) * sizeof(int4));
      A_(s.bc_prog, sizeof(int4));
      A_(s.bc_slot, (size_t)C * sizeof(int32_t));
      A_(s.bc_q, (size_t)D * D * sizeof(double));
    }
    A_(s.freq, (size_t)s.S_pad * sizeof(double));
    A_(s.pin, (size_t)s.S_pad * sizeof(int16_t));
    hipMemset(s.pin, 0, (size_t)s.S_pad * sizeof(int16_t));
    A_(s.ambig, (size_t)std::max<int64_t>(1, n_ambig) * DP * sizeof(double));
    // (+2 node slots per class behind the last class: outside vector of the cached branch and a scratch slot, see
    //  hyphy_hip_branch_cache_build)
    const size_t node_stride = p->nuc ? (size_t)4 * s.S_pad : (size_t)s.ntiles * 16 * DP;
    A_(s.partials, ((size_t)C * s.partial_stride + 2 * C * node_stride) * sizeof(double));
    A_(s.counts, ((size_t)C * I + 2 * C) * s.S_pad * sizeof(int32_t));
    A_(s.site_lik, (size_t)C * s.S_pad * sizeof(double));
    A_(s.site_cnt, (size_t)C * s.S_pad * sizeof(int32_t));
    A_(s.mixed_lik, (size_t)s.S_pad * sizeof(double));
    A_(s.mixed_cnt, (size_t)s.S_pad * sizeof(int32_t));
    if (p->nuc) {
      A_(s.Prow, (size_t)2 * C * B * 16 * sizeof(double));  // row-major, then transposed (prune_nuc2_kernel)
    } else {
      A_(s.Pfrag, ((size_t)C * B + (size_t)C * (I + 2) + kMaxTwin) * DP * DP * sizeof(double));  // (+ transposed path matrices of the branch cache, + twins of re-rooted schedules)
      A_(s.pi_ones, (size_t)DP * sizeof(double));
      {
        std::vector<double> ones(DP, 0.);
        for (int64_t k = 0; k < D; k++) ones[k] = 1.;
        hipMemcpy(s.pi_ones, ones.data(), ones.size() * sizeof(double), hipMemcpyHostToDevice);
      }
      A_(s.PTg, (size_t)C * B * DP * DP * sizeof(double));
    }
    A_(s.qbuf, (size_t)C * B * D * D * sizeof(double));
    A_(s.slots, (size_t)C * B * sizeof(int32_t));
    A_(s.ops, ops_capacity(p) * sizeof(int4));
    A_(s.prog, (size_t)(I + 2) * sizeof(int4));
    A_(s.jn, (size_t)(I + 2) * sizeof(int4));
    if (!p->nuc) {
      A_(s.frag_ctr, (size_t)C * (I + 2) * s.ntiles * sizeof(int));
      hipMemset(s.frag_ctr, 0, (size_t)C * (I + 2) * s.ntiles * sizeof(int));
      A_(s.hand_cnt, (size_t)C * I * s.ntiles * 32 * sizeof(int32_t));
    }
    A_(s.pi, (size_t)DP * sizeof(double));
    A_(s.out, 4 * sizeof(double));  // device-side result record [log-L, scaler sum, status copy, pad]
    A_(s.status, sizeof(int32_t));  // set by the expm kernel when a matrix fails
    A_(s.weights, (size_t)C * sizeof(double));
    s.wg_cap = p->nuc ? (s.S_pad + 255) / 256 : s.ntiles;
    A_(s.wg_sum, ((size_t)C * s.wg_cap + 4) * sizeof(double));  // x C: rate-class batching writes one row per class; + 4: the fused
    A_(s.wg_cnt, ((size_t)C * s.wg_cap + 4) * sizeof(long long));  // final combine reads them with 16-byte loads
    A_(s.wg_flag, ((size_t)C * s.wg_cap + 4) * sizeof(int));
    hipMemset(s.wg_flag + (size_t)C * s.wg_cap, 0, 4 * sizeof(int));  // (the last word: arrival counter of the 4-state kernel's fused combine)
#undef A_
    s.h_small_cap = (size_t)std::max<int64_t>(std::max<int64_t>(DP, C), 64);
    if (pool_host_malloc((void **)&s.h_ops, ops_capacity(p) * sizeof(int4)) != hipSuccess ||
        pool_host_malloc((void **)&s.h_prog, (size_t)(I + 2) * sizeof(int4)) != hipSuccess ||
        pool_host_malloc((void **)&s.h_jn, (size_t)(I + 2) * sizeof(int4)) != hipSuccess ||
        pool_host_malloc((void **)&s.h_out, 4 * sizeof(double)) != hipSuccess ||
        pool_host_malloc((void **)&s.h_slots, (size_t)C * B * sizeof(int32_t)) != hipSuccess ||
        pool_host_malloc((void **)&s.h_small, s.h_small_cap * sizeof(double)) != hipSuccess) {
      hyphy_hip_destroy(p);
      return fail("hipHostMalloc failed");
    }
    hipMemsetAsync(s.status, 0, sizeof(int32_t), s.stream);
    if (hipHostGetDevicePointer((void **)&s.d_hout, s.h_out, 0) != hipSuccess) s.d_hout = nullptr;
    for (int k = 0; k < 4; k++) s.h_out[k] = 0.;
    hipMemsetAsync(s.partials, 0, (size_t)C * s.partial_stride * sizeof(double), s.stream);
    hipMemsetAsync(s.counts, 0, (size_t)C * I * s.S_pad * sizeof(int32_t), s.stream);
    hipMemsetAsync(s.site_lik, 0, (size_t)C * s.S_pad * sizeof(double), s.stream);
    hipMemsetAsync(s.site_cnt, 0, (size_t)C * s.S_pad * sizeof(int32_t), s.stream);
    // leaf table: int64 pattern-indexed -> packed int16 [L][S_pad]; padding patterns use state 0, weight 0
    shard_codes.push_back(std::vector<int16_t>((size_t)L * s.S_pad, 0));
    std::vector<int16_t> &codes = shard_codes.back();
    for (int64_t l = 0; l < L; l++)
      for (int64_t k = 0; k < s.S; k++) codes[(size_t)l * s.S_pad + k] = (int16_t)leaf_codes[l * S + src_pattern(s.s0 + k)];
    std::vector<double> fr(s.S_pad, 0.0);
    for (int64_t k = 0; k < s.S; k++) fr[k] = (double)pattern_freq[src_pattern(s.s0 + k)];
    std::vector<double> amb((size_t)std::max<int64_t>(1, n_ambig) * DP, 0.0);
    const int astride = p->nuc ? 4 : DP;
    for (int64_t a = 0; a < n_ambig; a++)
      for (int64_t k = 0; k < D; k++) amb[a * astride + k] = ambig[a * D + k];
    hipMemcpy(s.codes, codes.data(), codes.size() * sizeof(int16_t), hipMemcpyHostToDevice);
    if (!p->nuc) {  // tile-major copy [tile][L][16] for the wave-per-tile kernels: a tile's codes are one 32 L-byte run
      std::vector<int16_t> ct((size_t)L * s.S_pad, 0);
      for (int64_t l = 0; l < L; l++)
        for (int64_t k = 0; k < s.S_pad; k++) ct[((size_t)(k >> 4) * L + l) * 16 + (k & 15)] = codes[(size_t)l * s.S_pad + k];
      hipMemcpy(s.codes_tile, ct.data(), ct.size() * sizeof(int16_t), hipMemcpyHostToDevice);
    }
    hipMemcpy(s.freq, fr.data(), fr.size() * sizeof(double), hipMemcpyHostToDevice);
    hipMemcpy(s.ambig, amb.data(), amb.size() * sizeof(double), hipMemcpyHostToDevice);
    {  // (the host copy is only kept for the class computation of subtree repeats)
      const char *re = getenv("HYPHY_HIP_REPEATS");
      if (re && atoi(re) == 0) std::vector<int16_t>().swap(shard_codes.back());
    }
    if (hipStreamSynchronize(s.stream) != hipSuccess || hipGetLastError() != hipSuccess) {
      hyphy_hip_destroy(p);
      return fail("device initialisation failed");
    }
  }
  // subtree repeats: classes, compressed set, trunk view and tables (repeats.hip); leaves rep_on false when it would not pay
  rep_setup(p, shard_codes);  // (never fails the creation: a partition whose tables cannot be set up evaluates plain)
  *out = p;
  return 0;
}

/* Subtree repeats (repeats.hip) are used wherever they pay; `on` = 0 turns them off for this partition (the pruning kernels
 * then walk every node at every pattern), 1 back on.  Same results either way (HYPHY_HIP_REPEATS=0 does the same for
 * every partition created afterwards). */
int hyphy_hip_set_repeats(hyphy_hip_partition *p, int on) {
  if (!p) return fail("partition == NULL");
  if (finish_pending_async(p)) return -1;
  p->rep_enabled = on != 0;
  p->rep_decided = true;  // (the caller's choice stands: no measurement overrides it)
  return 0;
}

/* What the class compression does on this partition's first shard: out[0] available, [1] class tables, [2] table rows (classes
 * padded to tiles of 16: the edge products of the lower phase), [3] / [4] internal nodes / leaves of the trunk, [5] edge
 * products one full pass executes with repeats on, [6] ... with repeats off (every internal edge at every pattern), [7] in use. */
int hyphy_hip_repeat_stats(const hyphy_hip_partition *p, int64_t out[8]) {
  if (!p || !out) return fail("null argument");
  for (int k = 0; k < 8; k++) out[k] = 0;
  const Shard &s = p->shards[0];
  out[6] = (p->I - 1) * (int64_t)s.S_pad;
  out[5] = out[6];
  if (!p->rep_on) return 0;
  out[0] = 1;
  out[1] = (int64_t)p->rep_nodes.size();
  out[2] = s.rep_rows;
  out[3] = p->views[1].I;
  out[4] = p->views[1].L;
  out[5] = (int64_t)(p->views[1].I - 1) * s.S_pad;
  for (size_t d = 0; d < p->rep_nodes.size(); d++)  // (a path of k nodes is walked once per 16 classes of its top node)
    out[5] += (int64_t)s.rep_tabs[d].rows * (int64_t)std::max<size_t>(1, p->rep_nodes[d].path.size());
  out[7] = p->rep_enabled ? 1 : 0;
  return 0;
}

}  // extern "C"

namespace hyhip {

// An uncollected hyphy_hip_evaluate_async owns the host-mapped result record: every entry point that is about to write it
// (or to wait on it) finishes the pending evaluation first (include/hyphy_hip.h: "any other evaluation entry point ...").
int finish_pending_async(hyphy_hip_partition *p) {
  if (!p->async_pending) return 0;
  for (Shard &s : p->shards) {
    HIPCHK(hipSetDevice(s.device));
    HIPCHK(hipStreamSynchronize(s.stream));
    s.seq_wait = 0.;
  }
  p->async_pending = false;
  return 0;
}

int eval_common(hyphy_hip_partition *p, int64_t cat, const int64_t *update_nodes, int64_t n_update,
                       const int64_t *q_nodes, int64_t n_q, const double *q, bool q_on_device, int q_is_probability,
                       const double *root_freqs, double *d_logl_out, bool reduce, bool floor_log, bool batch,
                       bool force_persist, const MixSpec *mix) {
  if (!p) return fail("partition == NULL");
  if (cat < 0) cat = 0;
  if (cat >= p->C) return fail("rate class out of range");
  if (finish_pending_async(p)) return -1;
  if (batch) {  // all classes in one launch: bookkeeping is shared, keyed on class 0
    if (p->nuc) return fail("internal: class batching is for the MFMA path");
    cat = 0;
    for (int64_t c = 1; c < p->C; c++)
      if (p->initialized[c] != p->initialized[0]) p->initialized[0] = 0;  // mixed state: force a full pass
  }
  // any ordinary evaluation invalidates the branch cache of its class (as the reference resets cachedBranches)
  if (batch) std::fill(p->bc_node.begin(), p->bc_node.end(), -1);
  else if (cat >= 0 && cat < (int64_t)p->bc_node.size()) p->bc_node[cat] = -1;
  if (!root_freqs) return fail("root_freqs == NULL");
  if ((n_update > 0 && !update_nodes) || (n_q > 0 && (!q_nodes || !q))) return fail("null node / matrix list");
  if (n_q > p->B) return fail("more matrices than branches");
  if (!p->initialized[cat] && n_q < p->B)
    return fail("first evaluation of a rate class must supply all L+I-1 transition matrices");
  {  // validate the matrix list BEFORE any host cache is touched (a failed call must not leave a half-committed state)
    std::vector<char> seen(p->B, 0);
    for (int64_t k = 0; k < n_q; k++) {
      if (q_nodes[k] < 0 || q_nodes[k] >= p->B) return fail("q_nodes entry out of range");
      if (seen[q_nodes[k]]) return fail("q_nodes lists a branch twice");
      seen[q_nodes[k]] = 1;
    }
  }
  // the view this evaluation runs under: the class-compressed one wherever it exists, except with pinned states and for the
  // internal passes that restore the per-pattern copies of every node (branch cache, downloads)
  if (p->rep_on && p->rep_enabled && !p->rep_decided && !getenv("HYPHY_HIP_REPEATS") && getenv("HYPHY_HIP_TUNE") && atoi(getenv("HYPHY_HIP_TUNE")) == 0) {
    //  no measurement will ever settle on / off for this partition (the tuner is disabled): the static rule instead
    p->rep_decided = true;
    if (!rep_static_decision(p)) p->rep_enabled = false;
    p->rep_report = std::string("repeats: no measurement (HYPHY_HIP_TUNE=0), static rule -> ") + (p->rep_enabled ? "on" : "off");
    if (getenv("HYPHY_HIP_VERBOSE")) fprintf(stderr, "[hyphy_hip] %s\n", p->rep_report.c_str());
  }
  switch_mode(p, (p->rep_on && p->rep_enabled && p->pin_node < 0 && !force_persist) ? 1 : 0);
  bool changed = false;
  const int64_t bc = batch ? p->C : 1;
  if (bc != p->batch_classes) {
    p->batch_classes = bc;
    p->cached_valid = 0;  // fragment sizing depends on how many classes share the launch
  }
  if (prepare_schedule(p, (int)cat, update_nodes, n_update, &changed, force_persist, q_nodes, n_q, batch ? (int)p->C : 1)) return -1;
  {
    const bool tune_on = p->tuned_for != p->batch_classes && !(getenv("HYPHY_HIP_TUNE") && atoi(getenv("HYPHY_HIP_TUNE")) == 0) && !getenv("HYPHY_HIP_CHAIN_M") &&
                                !getenv("HYPHY_HIP_CUT") && !getenv("HYPHY_HIP_FRAGMENT");
    if (tune_on && !p->nuc && (p->variant >= 1 || (!p->kernel_forced && p->shards[0].ntiles >= 32)) && p->shards[0].T == 1 && p->sched_full && !p->sched_persist &&
        p->tuned_for != p->batch_classes && p->initialized[cat]) {
      if (tune_schedule(p, (int)cat, batch ? (int)p->C : 1)) return -1;
      // class-compressed or plain?  settled once per partition by timing both (repeats.hip), unless the caller or the environment chose
      if (p->mode == 1 && !p->rep_decided && !getenv("HYPHY_HIP_REPEATS") && rep_decide(p, (int)cat, batch ? (int)p->C : 1)) return -1;
      build_schedule(p, nullptr, 0, true);  // the chosen cut (a full pass: no update list)
      if (p->ops_host.size() > ops_capacity(p)) return fail("internal: schedule overflow");
      changed = true;
    }
  }
  bool pi_changed = p->cached_pi.size() != (size_t)p->D || memcmp(p->cached_pi.data(), root_freqs, p->D * sizeof(double));
  if (pi_changed) p->cached_pi.assign(root_freqs, root_freqs + p->D);
  if (p->cached_slots.size() != (size_t)p->C) p->cached_slots.assign(p->C, std::vector<int64_t>());
  // (a batched evaluation writes its C * n_q slot numbers from the start of the device table, across the regions the classes use one at
  //  a time: a switch between the two forms invalidates what is known about EVERY class's region, not only this call's — r06: the
  //  adapter's category hook alternates batched passes with one-class line searches, and class 1 found class 0's batch table in its
  //  region: exponentials written to slots beyond the last class)
  if (p->slots_batch_mode != (batch ? 1 : 0))
    for (auto &v : p->cached_slots) v.assign(1, -1);  // (no list of node codes equals this)
  std::vector<int64_t> &cs = p->cached_slots[cat];
  bool slots_changed = cs.size() != (size_t)n_q || (n_q > 0 && memcmp(cs.data(), q_nodes, n_q * sizeof(int64_t)));
  if (slots_changed) cs.assign(q_nodes, q_nodes + n_q);
  p->slots_batch_mode = batch ? 1 : 0;
  for (Shard &s : p->shards)
    if (enqueue_eval(p, s, (int)cat, batch ? (int)p->C : 1, changed, pi_changed, slots_changed, q_nodes, n_q, q,
                     q_on_device, q_is_probability, root_freqs, d_logl_out, reduce, floor_log, mix)) {
      // some shard may hold a stale schedule / slot table / frequency vector now: rebuild everything next time
      p->cached_valid = 0;
      p->cached_pi.clear();
      for (auto &v : p->cached_slots) v.clear();
      p->slots_batch_mode = -1;
      p->initialized[cat] = 0;
      return -1;
    }
  std::vector<char> &res_here = p->mode == 1 ? p->rep_resident : p->resident, &res_other = p->mode == 1 ? p->resident : p->rep_resident;
  if (batch)
    for (int64_t c = 0; c < p->C; c++) {
      p->initialized[c] = 1;
      if (p->sched_full) res_here[c] = p->sched_persist ? 1 : 0;
      if (!res_other.empty()) res_other[c] = 0;  // (the other view's copies are stale now)
      p->last_full[c] = p->last_full[0];
    }
  p->initialized[cat] = 1;
  if (p->sched_full) res_here[cat] = p->sched_persist ? 1 : 0;
  if (!res_other.empty()) res_other[cat] = 0;
  return 0;  // (coefficients staged by hyphy_hip_build_q stay valid — and pending — until the next hyphy_hip_build_q)
}

// value of a device scalar behind everything queued on the (single) shard's stream -> host, through the host-mapped record
__global__ void publish_scalar_kernel(const double *__restrict__ value, double *__restrict__ rec, const int *__restrict__ status,
                                      double seq) {
  rec[0] = value[0];
  rec[1] = 0.;
  rec[2] = status ? (double)*status : 0.;
  if (seq != 0.) {
    __threadfence_system();
    reinterpret_cast<volatile double *>(rec)[3] = seq;
  }
}
int publish_and_collect(hyphy_hip_partition *p, const double *d_value, double *value_out) {
  if (finish_pending_async(p)) return -1;
  Shard &s = p->shards[0];
  HIPCHK(hipSetDevice(s.device));
  double *rec = s.d_hout ? s.d_hout : s.out;
  hipLaunchKernelGGL(publish_scalar_kernel, dim3(1), dim3(1), 0, s.stream, d_value, rec, (const int *)s.status,
                     next_seq(s, rec == s.d_hout));
  HIPCHK(hipGetLastError());
  if (collect_status(p)) return -1;
  *value_out = s.h_out[0];
  return 0;
}

}  // namespace hyhip

extern "C" {

int hyphy_hip_evaluate(hyphy_hip_partition *p, int64_t cat, const int64_t *update_nodes, int64_t n_update,
                       const int64_t *q_nodes, int64_t n_q, const double *q_dense, int q_is_probability,
                       const double *root_freqs, double *logl_out, double *site_lik_out, int64_t *site_scaler_out) {
  if (p && begin_site_export(p, site_lik_out != nullptr, site_scaler_out != nullptr)) return -1;
  if (eval_common(p, cat, update_nodes, n_update, q_nodes, n_q, q_dense, false, q_is_probability, root_freqs, nullptr,
                  true, false)) {
    if (p) p->export_sites = 0;
    return -1;
  }
  p->export_sites = 0;
  std::vector<double> parts;
  if (collect_status(p)) return -1;
  for (Shard &s : p->shards) parts.push_back(s.h_out[0]);
  record_timings(p);
  if (combine_shards(p, logl_out)) return -1;
  return finish_sites(p, cat < 0 ? 0 : (int)cat, site_lik_out, site_scaler_out);
}

/* Branch-site mixtures on every branch (the reference's "explicit form" models: BUSTED / BS-REL whole-alignment
 * evaluation): P_b = sum_m weights exp(Q_bm), formed on the device. */
int hyphy_hip_evaluate_mixture(hyphy_hip_partition *p, int64_t cat, const int64_t *update_nodes, int64_t n_update,
                               const int64_t *q_nodes, int64_t n_q, const int64_t *n_components, const double *q_dense,
                               const double *weights, const double *root_freqs, double *logl_out, double *site_lik_out,
                               int64_t *site_scaler_out) {
  if (!p) return fail("partition == NULL");
  if (n_q > 0 && (!n_components || !weights || !q_dense)) return fail("mixture evaluation: null argument");
  MixSpec mix{n_components, weights, 0};
  for (int64_t k = 0; k < n_q; k++) {
    if (n_components[k] < 1 || n_components[k] > 16) return fail("mixture evaluation: 1..16 components per branch");
    mix.n_tot += n_components[k];
  }
  if (eval_common(p, cat, update_nodes, n_update, q_nodes, n_q, q_dense, false, 0, root_freqs, nullptr, true, false, false, false,
                  n_q > 0 ? &mix : nullptr))
    return -1;
  if (collect_status(p)) return -1;
  std::vector<double> parts;
  for (Shard &s : p->shards) parts.push_back(s.h_out[0]);
  record_timings(p);
  if (logl_out) *logl_out = combine(parts);
  if (site_lik_out || site_scaler_out) return gather_sites(p, cat < 0 ? 0 : (int)cat, site_lik_out, site_scaler_out, false);
  return 0;
}

/* The same with the component rate matrices formed ON THE DEVICE (r04): Q_(b,m) = sum_k x_(b,m),k T_k over the templates of
 * hyphy_hip_set_q_templates / hyphy_hip_update_q_templates, one coefficient row per (branch of q_nodes, component) staged by
 * hyphy_hip_build_q (n = sum of n_components rows, branch-major) — no dense matrix crosses PCIe.  BS-REL / BUSTED components
 * differ in a global only (omega_m): component m of a branch with locals x_b is x_b at the columns of component m's templates. */
int hyphy_hip_evaluate_mixture_built(hyphy_hip_partition *p, int64_t cat, const int64_t *update_nodes, int64_t n_update,
                                     const int64_t *q_nodes, int64_t n_q, const int64_t *n_components, const double *weights,
                                     const double *root_freqs, double *logl_out, double *site_lik_out, int64_t *site_scaler_out) {
  if (!p) return fail("partition == NULL");
  if (!p->K) return fail("evaluate_mixture_built: templates not set");
  if (n_q > 0 && (!n_components || !weights)) return fail("mixture evaluation: null argument");
  if (n_q <= 0) return fail("evaluate_mixture_built: no matrices (use hyphy_hip_evaluate_built for a pure re-evaluation)");
  MixSpec mix{n_components, weights, 0};
  for (int64_t k = 0; k < n_q; k++) {
    if (n_components[k] < 1 || n_components[k] > kMixRows) return fail("mixture evaluation: 1..16 components per branch");
    mix.n_tot += n_components[k];
  }
  if (eval_common(p, cat, update_nodes, n_update, q_nodes, n_q, &kOwnQBuffer, true, 0, root_freqs, nullptr, true, false, false, false, &mix))
    return -1;
  if (collect_status(p)) return -1;
  std::vector<double> parts;
  for (Shard &s : p->shards) parts.push_back(s.h_out[0]);
  record_timings(p);
  if (logl_out) *logl_out = combine(parts);
  if (site_lik_out || site_scaler_out) return gather_sites(p, cat < 0 ? 0 : (int)cat, site_lik_out, site_scaler_out, false);
  return 0;
}

/* Asynchronous pair (partitions of one likelihood function on different devices / streams overlap: the host enqueues
 * every partition's evaluation before it waits for the first — the reference's partition loop, likefunc.cpp:2524-2589,
 * is serial).  The matrices are copied to a pinned staging buffer, so the caller's array is free on return. */
int hyphy_hip_evaluate_async(hyphy_hip_partition *p, int64_t cat, const int64_t *update_nodes, int64_t n_update,
                             const int64_t *q_nodes, int64_t n_q, const double *q_dense, int q_is_probability,
                             const double *root_freqs) {
  if (!p) return fail("partition == NULL");
  if (p->async_pending && hyphy_hip_synchronize(p)) return -1;
  p->async_pending = false;
  const size_t n = (size_t)std::max<int64_t>(0, n_q) * p->D * p->D;
  if (n > 0) {
    if (!q_dense) return fail("null matrix list");
    if (p->h_qstage_cap < n) {
      if (hyphy_hip_synchronize(p)) return -1;
      if (p->h_qstage) { hipDeviceSynchronize(); pool_host_free(p->h_qstage); }
      p->h_qstage = nullptr;
      p->h_qstage_cap = 0;
      const size_t cap = std::max(n, (size_t)p->B * p->D * p->D);
      HIPCHK(pool_host_malloc((void **)&p->h_qstage, cap * sizeof(double)));
      p->h_qstage_cap = cap;
    }
    // (the previous asynchronous evaluation was collected or synchronised above: the staging buffer is free)
    memcpy(p->h_qstage, q_dense, n * sizeof(double));
  }
  if (eval_common(p, cat, update_nodes, n_update, q_nodes, n_q, n > 0 ? p->h_qstage : nullptr, false, q_is_probability,
                  root_freqs, nullptr, true, false))
    return -1;
  p->async_pending = true;
  p->async_cat = cat < 0 ? 0 : cat;
  return 0;
}

int hyphy_hip_collect(hyphy_hip_partition *p, double *logl_out, double *site_lik_out, int64_t *site_scaler_out) {
  if (!p) return fail("partition == NULL");
  if (!p->async_pending) return fail("collect: no asynchronous evaluation pending");
  p->async_pending = false;
  if (collect_status(p)) return -1;
  std::vector<double> parts;
  for (Shard &s : p->shards) parts.push_back(s.h_out[0]);
  record_timings(p);
  if (logl_out) *logl_out = combine(parts);
  if (site_lik_out || site_scaler_out) return gather_sites(p, (int)p->async_cat, site_lik_out, site_scaler_out, false);
  return 0;
}

int hyphy_hip_evaluate_built_sites(hyphy_hip_partition *p, int64_t cat, const int64_t *update_nodes, int64_t n_update,
                                   const int64_t *q_nodes, int64_t n_q, const double *root_freqs, double *logl_out,
                                   double *site_lik_out, int64_t *site_scaler_out) {
  if (!p) return fail("partition == NULL");
  if (begin_site_export(p, site_lik_out != nullptr, site_scaler_out != nullptr)) return -1;
  const int rc = hyphy_hip_evaluate_built(p, cat, update_nodes, n_update, q_nodes, n_q, root_freqs, logl_out);
  p->export_sites = 0;
  if (rc) return -1;
  return finish_sites(p, cat < 0 ? 0 : (int)cat, site_lik_out, site_scaler_out);
}

int hyphy_hip_evaluate_built(hyphy_hip_partition *p, int64_t cat, const int64_t *update_nodes, int64_t n_update,
                             const int64_t *q_nodes, int64_t n_q, const double *root_freqs, double *logl_out) {
  if (!p) return fail("partition == NULL");
  if (!p->K) return fail("evaluate_built: templates not set");
  // q = each shard's own Q buffer (filled / staged by build_q on every shard)
  if (eval_common(p, cat, update_nodes, n_update, q_nodes, n_q, &kOwnQBuffer, true, 0, root_freqs, nullptr, true, false))
    return -1;
  Trace tr("evaluate_built");
  if (collect_status(p)) return -1;
  tr.lap("wait");
  return combine_shards(p, logl_out);
}

int hyphy_hip_evaluate_device(hyphy_hip_partition *p, int64_t cat, const int64_t *update_nodes, int64_t n_update,
                              const int64_t *q_nodes, int64_t n_q, const double *d_q, int q_is_probability,
                              const double *root_freqs, double *d_logl_out) {
  if (!p) return fail("partition == NULL");
  if (p->shards.size() != 1) return fail("evaluate_device needs a single-device partition");
  Trace tr("evaluate_device");
  struct Fin { Trace &t; ~Fin() { t.lap("total"); } } fin{tr};
  return eval_common(p, cat, update_nodes, n_update, q_nodes, n_q, d_q, true, q_is_probability, root_freqs, d_logl_out,
                     true, false);
}

/* A device scalar (the all-reduced log-likelihood of a multi-rank evaluation: hyphy_hip_evaluate_device + the caller's
 * collective on the same stream) back on the host the way the synchronous entry points do it: a one-thread kernel behind
 * everything queued on the partition's stream posts [value, 0, expm status, sequence word] into the host-mapped result
 * record and the host spins on the sequence word — no device-to-host copy command, no stream synchronisation. */

int hyphy_hip_fetch_device_scalar(hyphy_hip_partition *p, const double *d_value, double *value_out) {
  if (!p || !d_value || !value_out) return fail("fetch_device_scalar: null argument");
  if (p->shards.size() != 1) return fail("fetch_device_scalar needs a single-device partition");
  return publish_and_collect(p, d_value, value_out);
}

int hyphy_hip_evaluate_categories(hyphy_hip_partition *p, const int64_t *update_nodes, int64_t n_update,
                                  const int64_t *q_nodes, int64_t n_q, const double *q_dense, int q_is_probability,
                                  const double *weights, const double *root_freqs, double *logl_out,
                                  double *site_lik_out, int64_t *site_scaler_out) {
  if (!p) return fail("partition == NULL");
  if (!weights) return fail("weights == NULL");
  const int64_t D = p->D;
  if (!p->nuc) {
    // rate-class batching: ONE expm launch over C*n_q matrices and ONE pruning launch with a grid row
    // per class (3x the workgroups of a single pass: the matrix pipe finally has enough waves)
    if (eval_common(p, 0, update_nodes, n_update, q_nodes, n_q, q_dense, false, q_is_probability, root_freqs, nullptr,
                    false, true, /* batch = */ true))
      return -1;
  } else {
    for (int64_t c = 0; c < p->C; c++) {
      if (!p->initialized[c]) p->cached_valid = 0;
      if (eval_common(p, c, update_nodes, n_update, q_nodes, n_q,
                      q_dense ? q_dense + (size_t)c * n_q * D * D : nullptr, false, q_is_probability, root_freqs,
                      nullptr, false, true))
        return -1;
    }
  }
  std::vector<double> parts;
  p->cached_weights.assign(weights, weights + p->C);
  for (Shard &s : p->shards) {
    HIPCHK(hipSetDevice(s.device));
    if (upload_small(s, weights, (size_t)p->C, s.weights)) return -1;
    launch_mix_categories(s.site_lik, s.site_cnt, s.weights, (int)p->C, s.S_pad, s.mixed_lik, s.mixed_cnt, s.stream);
    double *rec = s.d_hout ? s.d_hout : s.out;
    launch_site_reduce(s.mixed_lik, s.mixed_cnt, s.freq, s.S_pad, 1, rec, rec + 1, s.status, s.stream, next_seq(s, rec == s.d_hout));
  }
  if (collect_status(p)) return -1;
  for (Shard &s : p->shards) parts.push_back(s.h_out[0]);
  if (logl_out) *logl_out = combine(parts);
  if (site_lik_out || site_scaler_out) return gather_sites(p, 0, site_lik_out, site_scaler_out, true);
  return 0;
}

int hyphy_hip_evaluate_categories_built_sites(hyphy_hip_partition *p, const int64_t *update_nodes, int64_t n_update,
                                              const int64_t *q_nodes, int64_t n_q, const double *weights,
                                              const double *root_freqs, double *logl_out, double *site_lik_out,
                                              int64_t *site_scaler_out) {
  if (!p) return fail("partition == NULL");
  if (!p->K) return fail("evaluate_categories_built: templates not set");
  if (p->nuc) return fail("evaluate_categories_built: MFMA partitions only (4-state: hyphy_hip_evaluate_categories)");
  if (!weights) return fail("weights == NULL");
  if (eval_common(p, 0, update_nodes, n_update, q_nodes, n_q, &kOwnQBuffer, true, 0, root_freqs, nullptr, false, true, true))
    return -1;
  const bool w_changed = p->cached_weights.size() != (size_t)p->C ||
                         memcmp(p->cached_weights.data(), weights, p->C * sizeof(double));
  if (w_changed) p->cached_weights.assign(weights, weights + p->C);
  for (Shard &s : p->shards) {
    HIPCHK(hipSetDevice(s.device));
    if (w_changed && upload_small(s, weights, (size_t)p->C, s.weights)) {
      p->cached_weights.clear();
      return -1;
    }
    launch_mix_categories(s.site_lik, s.site_cnt, s.weights, (int)p->C, s.S_pad, s.mixed_lik, s.mixed_cnt, s.stream);
    double *rec = s.d_hout ? s.d_hout : s.out;
    launch_site_reduce(s.mixed_lik, s.mixed_cnt, s.freq, s.S_pad, 1, rec, rec + 1, s.status, s.stream, next_seq(s, rec == s.d_hout));
  }
  if (collect_status(p)) return -1;
  if (logl_out) {
    std::vector<double> parts;
    for (Shard &s : p->shards) parts.push_back(s.h_out[0]);
    *logl_out = combine(parts);
  }
  if (site_lik_out || site_scaler_out) return gather_sites(p, 0, site_lik_out, site_scaler_out, true);
  return 0;
}

int hyphy_hip_evaluate_categories_built(hyphy_hip_partition *p, const int64_t *update_nodes, int64_t n_update,
                                        const int64_t *q_nodes, int64_t n_q, const double *weights,
                                        const double *root_freqs, double *logl_out) {
  return hyphy_hip_evaluate_categories_built_sites(p, update_nodes, n_update, q_nodes, n_q, weights, root_freqs, logl_out, nullptr, nullptr);
}

static int ensure_resident(hyphy_hip_partition *p, int64_t cat);

int hyphy_hip_download_partials(hyphy_hip_partition *p, int64_t cat, double *inode_cache, int64_t *scaler_counts) {
  if (!p) return fail("partition == NULL");
  if (cat < 0) cat = 0;
  if (cat >= p->C) return fail("rate class out of range");
  if (ensure_resident(p, cat)) return -1;
  const int64_t D = p->D, I = p->I, S = p->S;
  for (Shard &s : p->shards) {
    HIPCHK(hipSetDevice(s.device));
    HIPCHK(hipStreamSynchronize(s.stream));
    if (inode_cache) {
      if (p->nuc) {
        std::vector<double> tmp((size_t)I * 4 * s.S_pad);
        HIPCHK(hipMemcpy(tmp.data(), s.partials + (size_t)cat * s.partial_stride, tmp.size() * sizeof(double),
                         hipMemcpyDeviceToHost));
        for (int64_t n = 0; n < I; n++)
          for (int64_t k = 0; k < s.S; k++)
            for (int j = 0; j < 4; j++)
              inode_cache[(n * S + caller_pattern(p, s.s0 + k)) * 4 + j] = tmp[((size_t)n * 4 + j) * s.S_pad + k];
      } else {
        double *dtmp = nullptr;
        HIPCHK(pool_malloc((void **)&dtmp, (size_t)I * s.S * D * sizeof(double)));
        launch_unpack_partials_mfma(s.partials + (size_t)cat * s.partial_stride, (int)I, s.ntiles, p->NW, (int)D,
                                    (int)s.S, dtmp, s.stream);
        hipError_t e;
        if (p->perm.empty()) {
          e = hipMemcpy2DAsync(inode_cache + s.s0 * D, (size_t)S * D * sizeof(double), dtmp,
                               (size_t)s.S * D * sizeof(double), (size_t)s.S * D * sizeof(double), (size_t)I,
                               hipMemcpyDeviceToHost, s.stream);
          hipStreamSynchronize(s.stream);
        } else {  // sorted patterns: through a host copy, one node at a time, back into the caller's pattern order
          std::vector<double> tmp((size_t)s.S * D);
          e = hipStreamSynchronize(s.stream);  // (the unpack kernel runs on the shard's stream)
          for (int64_t n = 0; n < I && e == hipSuccess; n++) {
            e = hipMemcpy(tmp.data(), dtmp + (size_t)n * s.S * D, tmp.size() * sizeof(double), hipMemcpyDeviceToHost);
            for (int64_t k = 0; k < s.S; k++)
              memcpy(inode_cache + ((size_t)n * S + p->perm[s.s0 + k]) * D, tmp.data() + (size_t)k * D, (size_t)D * sizeof(double));
          }
        }
        pool_free_sync(dtmp);
        if (e != hipSuccess) return fail(std::string("download_partials: ") + hipGetErrorString(e));
      }
    }
    if (scaler_counts) {
      std::vector<int32_t> tmp((size_t)I * s.S_pad);
      HIPCHK(hipMemcpy(tmp.data(), s.counts + (size_t)cat * I * s.S_pad, tmp.size() * sizeof(int32_t),
                       hipMemcpyDeviceToHost));
      for (int64_t n = 0; n < I; n++)
        for (int64_t k = 0; k < s.S; k++) {
          scaler_counts[n * S + caller_pattern(p, s.s0 + k)] = tmp[(size_t)n * s.S_pad + k];
        }
    }
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// Pinned node states: ComputeBlock's branchIndex / branchValues ("setBranch", likefunc.cpp:10950-10957;
// tree_evaluator.cpp:163-181 for a leaf, :583-594 + :3624 for an internal node) — the evaluations that follow
// see node `node` fixed to states[pattern].  Used by marginal ancestral reconstruction
// (RecoverAncestralSequencesMarginal, likefunc2.cpp:932-1040: one pinned evaluation per node and state).
// ---------------------------------------------------------------------------------------------------
int hyphy_hip_set_pinned_states(hyphy_hip_partition *p, int64_t node, const int64_t *states) {
  if (!p) return fail("partition == NULL");
  if (node < 0 || !states) {
    if (p->pin_node >= 0 && p->rr_use) p->cached_valid = 0;  // (back to the re-rooted form of the steady-state passes)
    p->pin_node = -1;
    return 0;
  }
  if (node >= p->L + p->I) return fail("set_pinned_states: node out of range");
  for (Shard &s : p->shards) {
    HIPCHK(hipSetDevice(s.device));
    std::vector<int16_t> h(s.S_pad, 0);
    for (int64_t k = 0; k < s.S; k++) {
      const int64_t st = states[caller_pattern(p, s.s0 + k)];
      if (st < 0 || st >= p->D) return fail("set_pinned_states: state out of range");
      h[k] = (int16_t)st;
    }
    HIPCHK(hipStreamSynchronize(s.stream));
    HIPCHK(hipMemcpy(s.pin, h.data(), h.size() * sizeof(int16_t), hipMemcpyHostToDevice));
  }
  p->pin_node = node;
  if (p->rr_active) p->cached_valid = 0;  // (re-rooted schedules are built for evaluations without pinned states)
  std::fill(p->bc_node.begin(), p->bc_node.end(), -1);
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// Branch cache: _TheTree::ComputeBranchCache (tree_evaluator.cpp:4286-4845) builds, for one
// branch, everything the likelihood needs except that branch's transition matrix; ComputeLLWithBranchCache
// (tree.cpp:3383-3936) then evaluates L(t) with one [D x D] x [D x S] contraction per call while the
// optimiser's line search varies the branch length (policy code likefunc.cpp:10886-10948, 11125-11258).
// ---------------------------------------------------------------------------------------------------
// The persisted conditionals of class `cat` are stale (the last full pass ran with lazy persistence): re-run
// the pruning pass over the resident transition matrices with every node stored.
static int ensure_resident(hyphy_hip_partition *p, int64_t cat) {
  if (p->resident[cat]) return 0;
  if (!p->initialized[cat] || p->cached_pi.size() != (size_t)p->D) return fail("conditionals not resident: evaluate first");
  std::vector<int64_t> all(p->B);
  for (int64_t k = 0; k < p->B; k++) all[k] = k;
  const std::vector<double> pi = p->cached_pi;
  const std::vector<char> lf = p->last_full;
  if (eval_common(p, cat, all.data(), p->B, nullptr, 0, nullptr, false, 0, pi.data(), nullptr, true, false, false, true))
    return -1;
  p->last_full = lf;  // (an internal pass: the caller's own sequence of evaluations is what the policy looks at)
  return collect_status(p);
}

int hyphy_hip_branch_cache_build(hyphy_hip_partition *p, int64_t cat, int64_t node) {
  if (!p) return fail("partition == NULL");
  if (p->nuc) {
    g_last_error = "branch cache: not available for the 4-state path";
    return 1;
  }
  if (cat < 0) cat = 0;
  if (cat >= p->C) return fail("rate class out of range");
  if (node < 0 || node >= p->B) return fail("branch cache: node out of range (the root has no branch)");
  if (!p->initialized[cat]) return fail("branch cache: evaluate the partition first (conditionals must be resident)");
  if (ensure_resident(p, cat)) return -1;
  switch_mode(p, 0);  // (the outside vector is built over the partition's own tree)
  const int L = (int)p->L, I = (int)p->I, C = (int)p->C;
  const int64_t B = p->B;
  const int DP = p->DP;
  // ancestors of the branch's parent up to the root; a[0] = root ... a[m] = parent of `node`
  std::vector<int> a;
  for (int64_t x = p->parents[node]; x >= 0; x = p->parents[L + x]) a.push_back((int)x);
  std::reverse(a.begin(), a.end());
  const int m = (int)a.size() - 1;
  if (m < 0) return fail("branch cache: node has no parent");
  const int vbase = (C - (int)cat) * I + 2 * (int)cat;        // node slots behind the last class, relative to this class
  const int tbase = (int)((C - cat) * B + cat * (I + 2));     // matrix slots behind the last class, relative to this class
  std::vector<int4> ops;
  for (int k = 0; k <= m; k++) {
    const int vid = k == m ? vbase : vbase + 1;
    const int exclude = k < m ? L + a[k + 1] : (int)node;
    std::vector<int4> entries;
    if (k >= 1) {  // the rest of the tree above: previous chain node through the transposed matrix of a[k]'s branch
      int4 op;
      op.x = OPK_INTERNAL | (((k - 1) & 1) << 24);
      op.y = vid;
      op.z = tbase + k;
      op.w = 0;
      entries.push_back(op);
    }
    const std::vector<int> &ch = p->children[a[k]];
    const int G = p->shards[0].T <= 2 ? 2 : 1;
    std::vector<int> leaves;
    for (int c : ch)
      if (c < L && c != exclude) leaves.push_back(c);
    for (size_t i = 0; i < leaves.size();) {
      int nl = 1;
      const bool amb0 = p->leaf_has_ambig[leaves[i]];
      if (!amb0 && G > 1 && i + 1 < leaves.size() && !p->leaf_has_ambig[leaves[i + 1]]) nl = 2;
      const unsigned l0 = (unsigned)leaves[i], l1 = nl > 1 ? (unsigned)leaves[i + 1] : l0;
      int4 op;
      op.x = OPK_LEAF | (amb0 ? OPF_AMBIG : 0) | (nl << 8) | (0xff << 24);
      op.y = vid;
      op.z = (int)(l0 | (l1 << 16));
      op.w = 0;
      entries.push_back(op);
      i += nl;
    }
    for (int c : ch)
      if (c >= L && c != exclude) {
        int4 op;
        op.x = OPK_INTERNAL_GLOBAL | (0xff << 24);
        op.y = vid;
        op.z = c;
        op.w = c - L;
        entries.push_back(op);
      }
    if (entries.empty()) {  // (a root whose only other child is the excluded one: the outside vector is all ones)
      int4 op;
      op.x = OPK_LEAF | (0xff << 24);
      op.y = vid;
      op.z = 0;
      op.w = 0;
      entries.push_back(op);
    }
    entries.back().x |= OPF_LAST | ((k & 1) ? OPF_PARITY : 0) | ((k & 1) << 16);
    for (const int4 &e : entries) ops.push_back(e);
  }
  int4 nop;
  nop.x = OPK_LEAF | (0xff << 24);
  nop.y = vbase + 1;
  nop.z = 0;
  nop.w = 0;
  if (ops.size() & 1) ops.push_back(nop);
  const int n = (int)ops.size();
  ops.push_back(nop);
  ops.push_back(nop);
  if (ops.size() > ops_capacity(p)) return fail("internal: branch-cache schedule overflow");
  const int4 prog = make_int4(0, n, -1, 0);
  const int32_t slot = (int32_t)node;
  for (Shard &s : p->shards) {
    HIPCHK(hipSetDevice(s.device));
    HIPCHK(hipStreamSynchronize(s.stream));
    HIPCHK(hipMemcpyAsync(s.bc_ops, ops.data(), ops.size() * sizeof(int4), hipMemcpyHostToDevice, s.stream));
    HIPCHK(hipMemcpyAsync(s.bc_prog, &prog, sizeof(int4), hipMemcpyHostToDevice, s.stream));
    HIPCHK(hipMemcpyAsync(s.bc_slot + cat, &slot, sizeof(int32_t), hipMemcpyHostToDevice, s.stream));
    HIPCHK(hipStreamSynchronize(s.stream));  // (the host vectors go out of scope)
    double *Pf = s.Pfrag + (size_t)cat * B * DP * DP;
    for (int k = 1; k <= m; k++)  // M_k[j][i] = P_{a[k]}[i][j], times pi_i on the edge that leaves the old root
      launch_transpose_frag(Pf + (size_t)(L + a[k]) * DP * DP, Pf + (size_t)(tbase + k) * DP * DP, k == 1 ? s.pi : nullptr,
                            p->NW, s.stream);
    PruneArgs pa = base_prune_args(p, s, (int)cat, 1);
    pa.ops = s.bc_ops;
    pa.prog = s.bc_prog;
    pa.n_ops = n;
    pa.n_prog = 1;
    pa.do_root = 0;
    pa.n_prog_total = 1;
    launch_prune_mfma(pa, s.stream);
    HIPCHK(hipGetLastError());
  }
  p->bc_node[cat] = node;
  p->bc_use_pi[cat] = m == 0 ? 1 : 0;
  return 0;
}

int hyphy_hip_branch_cache_evaluate(hyphy_hip_partition *p, int64_t cat, int64_t node, const double *q_dense,
                                    int q_is_probability, double *logl_out, double *site_lik_out,
                                    int64_t *site_scaler_out) {
  if (!p) return fail("partition == NULL");
  if (cat < 0) cat = 0;
  if (cat >= p->C) return fail("rate class out of range");
  if (p->bc_node[cat] < 0 || p->bc_node[cat] != node)
    return fail("branch cache: no cache resident for this branch (call hyphy_hip_branch_cache_build after an evaluation)");
  if (!q_dense) return fail("null matrix pointer");
  if (finish_pending_async(p)) return -1;
  p->rep_stale_branch = node;  // (this branch's matrix image is rewritten below: its class table is stale)
  std::fill(p->rep_resident.begin(), p->rep_resident.end(), 0);
  const int L = (int)p->L, I = (int)p->I, C = (int)p->C;
  const int64_t B = p->B, D = p->D;
  const int DP = p->DP;
  for (Shard &s : p->shards) {
    HIPCHK(hipSetDevice(s.device));
    HIPCHK(hipMemcpyAsync(s.bc_q, q_dense, (size_t)D * D * sizeof(double), hipMemcpyHostToDevice, s.stream));
    ExpmArgs ea;
    ea.Q = s.bc_q;
    ea.slots = s.bc_slot + cat;
    ea.n = 1;
    ea.D = (int)D;
    ea.is_prob = q_is_probability;
    ea.status = s.status;
    ea.templates = nullptr;
    ea.coeffs = nullptr;
    ea.K = 0;
    ea.prof = 0;
    ea.Prow = nullptr;
    ea.Pfrag = s.Pfrag + (size_t)cat * B * DP * DP;
    ea.PTg = s.PTg + (size_t)cat * B * DP * DP;
    launch_expm(ea, s.stream);
    s.twins_dirty = true;  // (this branch's image was rewritten without its twin, if it has one)
    BcArgs ba;
    ba.NW = p->NW;
    ba.S_pad = s.S_pad;
    ba.ntiles = s.ntiles;
    ba.L = L;
    ba.node_A = (C - (int)cat) * I + 2 * (int)cat;
    ba.child_internal = node >= L ? (int)(node - L) : -1;
    ba.child_leaf = node < L ? (int)node : 0;
    ba.use_pi = p->bc_use_pi[cat];
    ba.Pfrag = ea.Pfrag + (size_t)node * DP * DP;
    ba.PTg = ea.PTg + (size_t)node * DP * DP;
    ba.codes_tile = s.codes_tile;
    ba.ambig = s.ambig;
    ba.partials = s.partials + (size_t)cat * s.partial_stride;
    ba.counts = s.counts + (size_t)cat * I * s.S_pad;
    ba.pi = s.pi;
    ba.freq = s.freq;
    ba.site_lik = s.site_lik + (size_t)cat * s.S_pad;
    ba.site_cnt = s.site_cnt + (size_t)cat * s.S_pad;
    ba.wg_sum = s.wg_sum;
    ba.wg_cnt = s.wg_cnt;
    ba.wg_flag = s.wg_flag;
    launch_bc_eval(ba, s.stream);
    double *rec = s.d_hout ? s.d_hout : s.out;
    launch_wg_reduce(s.wg_sum, s.wg_cnt, s.wg_flag, s.ntiles, rec, rec + 1, s.status, s.stream, next_seq(s, rec == s.d_hout));
    HIPCHK(hipGetLastError());
  }
  if (collect_status(p)) return -1;
  std::vector<double> parts;
  for (Shard &s : p->shards) parts.push_back(s.h_out[0]);
  if (logl_out) *logl_out = combine(parts);
  if (site_lik_out || site_scaler_out) return gather_sites(p, (int)cat, site_lik_out, site_scaler_out, false);
  return 0;
}

int hyphy_hip_expm_batch(int64_t D, int64_t n, const double *q_dense, double *p_out) {
  if (D < 2 || D > 64) {
    g_last_error = "unsupported state count (this version: 2 <= D <= 64)";
    return 1;
  }
  if (n <= 0) return 0;
  if (!q_dense || !p_out) return fail("null matrix pointer");
  if (hyphy_hip_device_count() <= 0) return fail("no HIP device available (this library has no CPU fallback)");
  double *dq = nullptr, *dp = nullptr;
  int32_t *st = nullptr;
  const size_t bytes = (size_t)n * D * D * sizeof(double);
  HIPCHK(pool_malloc((void **)&dq, bytes));
  HIPCHK(pool_malloc((void **)&dp, bytes));
  HIPCHK(pool_malloc((void **)&st, sizeof(int32_t)));
  HIPCHK(hipMemset(st, 0, sizeof(int32_t)));
  HIPCHK(hipMemcpy(dq, q_dense, bytes, hipMemcpyHostToDevice));
  ExpmArgs ea;
  ea.Q = dq; ea.slots = nullptr; ea.n = (int)n; ea.D = (int)D; ea.is_prob = 0;
  ea.Prow = dp; ea.Pfrag = nullptr; ea.PTg = nullptr; ea.prof = 0; ea.status = st;
  ea.templates = nullptr; ea.coeffs = nullptr; ea.K = 0;
  launch_expm(ea, nullptr);
  int32_t hst = 0;
  hipError_t e1 = hipMemcpy(p_out, dp, bytes, hipMemcpyDeviceToHost);
  hipError_t e2 = hipMemcpy(&hst, st, sizeof(int32_t), hipMemcpyDeviceToHost);
  pool_free_sync(dq); pool_free_sync(dp); pool_free_sync(st);
  if (e1 != hipSuccess || e2 != hipSuccess) return fail("expm_batch: device copy failed");
  if (hst) return fail("Failed to compute a valid transition matrix; this is usually caused by ill-conditioned rate "
                       "matrices (e.g. very large rate values)");
  return 0;
}

// [K][64*64] copies of the templates for expm64_kernel: zero padding, zero diagonals (the kernel derives Q_ii itself)
static std::vector<double> padded_templates(const double *templates, int64_t K, int64_t D) {
  std::vector<double> pad((size_t)K * 64 * 64, 0.0);
  for (int64_t k = 0; k < K; k++)
    for (int64_t r = 0; r < D; r++)
      for (int64_t c = 0; c < D; c++)
        if (r != c) pad[((size_t)k * 64 + r) * 64 + c] = templates[((size_t)k * D + r) * D + c];
  return pad;
}

int hyphy_hip_set_q_templates(hyphy_hip_partition *p, int64_t K, const double *templates) {
  if (!p || K < 1 || !templates) return fail("invalid templates");
  const int64_t D = p->D;
  for (Shard &s : p->shards) {
    HIPCHK(hipSetDevice(s.device));
    HIPCHK(hipStreamSynchronize(s.stream));
    if (s.templates) pool_free_sync(s.templates);
    if (s.coeffs) pool_free_sync(s.coeffs);
    s.templates = s.coeffs = nullptr;
    HIPCHK(pool_malloc((void **)&s.templates, (size_t)K * D * D * sizeof(double)));
    // rows: one per (class, branch), or — mixtures built on the device — one per (branch, component), up to kMixRows components
    const size_t coeff_rows_cap = (size_t)std::max<int64_t>(p->C, kMixRows) * p->B;
    HIPCHK(pool_malloc((void **)&s.coeffs, coeff_rows_cap * K * sizeof(double)));
    if (s.h_coeffs) { hipDeviceSynchronize(); pool_host_free(s.h_coeffs); }
    HIPCHK(pool_host_malloc((void **)&s.h_coeffs, (size_t)4 * coeff_rows_cap * K * sizeof(double)));
    if (hipHostGetDevicePointer((void **)&s.d_hcoeffs, s.h_coeffs, 0) != hipSuccess) s.d_hcoeffs = nullptr;
    s.coeffs_cur = nullptr;
    s.coeff_rows = 0;
    s.coeff_kind = 0;
    s.coeff_slot = -1;
    s.qbuf_built = false;
    for (int k = 0; k < 4; k++) {
      s.coeff_busy[k] = false;
      if (!s.coeff_ev[k]) HIPCHK(hipEventCreateWithFlags(&s.coeff_ev[k], hipEventDisableTiming));
    }
    HIPCHK(hipMemcpy(s.templates, templates, (size_t)K * D * D * sizeof(double), hipMemcpyHostToDevice));
    if (s.templates_pad) pool_free_sync(s.templates_pad);
    s.templates_pad = nullptr;
    if (p->DP == 64) {
      HIPCHK(pool_malloc((void **)&s.templates_pad, (size_t)K * 64 * 64 * sizeof(double)));
      const std::vector<double> pad = padded_templates(templates, K, D);
      HIPCHK(hipMemcpy(s.templates_pad, pad.data(), pad.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    // pinned ring for hyphy_hip_update_q_templates (two slots: the copy of one update may still be queued when the next arrives)
    if (s.h_tstage) { hipDeviceSynchronize(); pool_host_free(s.h_tstage); }
    s.h_tstage = nullptr;
    s.tstage_slot = (size_t)K * D * D + (p->DP == 64 ? (size_t)K * 64 * 64 : 0);
    HIPCHK(pool_host_malloc((void **)&s.h_tstage, 2 * s.tstage_slot * sizeof(double)));
    for (int k = 0; k < 2; k++) {
      s.tstage_busy[k] = false;
      if (!s.tstage_ev[k]) HIPCHK(hipEventCreateWithFlags(&s.tstage_ev[k], hipEventDisableTiming));
    }
  }
  p->K = K;
  p->templates_host.assign(templates, templates + (size_t)K * D * D);
  for (Shard &s : p->shards) s.fit_static_current = false;
  return 0;
}

/* Replace the VALUES of the K templates (same K as hyphy_hip_set_q_templates): in-stream, no reallocation.  For hosts
 * whose templates depend on the global parameters of the current evaluation (template mode of the
 * adapter: Q_b = sum_k x_bk M_k(globals), M_k re-derived at every evaluation from K probe branches). */
int hyphy_hip_update_q_templates(hyphy_hip_partition *p, int64_t K, const double *templates) {
  if (!p || K < 1 || !templates) return fail("invalid templates");
  if (K != p->K) return hyphy_hip_set_q_templates(p, K, templates);
  const int64_t D = p->D;
  const size_t n = (size_t)K * D * D;
  if (p->templates_host.size() == n && !memcmp(p->templates_host.data(), templates, n * sizeof(double))) return 0;
  p->templates_host.assign(templates, templates + n);
  for (Shard &s : p->shards) {
    HIPCHK(hipSetDevice(s.device));
    // r04: through a pinned ring, in-stream, no wait on the host — the copies run behind whatever is queued on the stream and
    // ahead of the exponential launch that reads them (r03 copied from pageable memory and synchronised the stream twice:
    // ~40 us of every evaluation of a host whose templates follow its global parameters).  A slot is reused
    // two updates later; its event says when its copy has run.
    const int slot = (int)(s.tstage_turn++ & 1);
    if (s.tstage_busy[slot]) {
      HIPCHK(hipEventSynchronize(s.tstage_ev[slot]));
      s.tstage_busy[slot] = false;
    }
    double *st = s.h_tstage + (size_t)slot * s.tstage_slot;
    memcpy(st, templates, n * sizeof(double));
    HIPCHK(hipMemcpyAsync(s.templates, st, n * sizeof(double), hipMemcpyHostToDevice, s.stream));
    if (s.templates_pad) {  // [K][64*64]: zero padding, zero diagonals (padded_templates)
      double *pad = st + n;
      memset(pad, 0, (size_t)K * 64 * 64 * sizeof(double));
      for (int64_t k = 0; k < K; k++)
        for (int64_t r = 0; r < D; r++) {
          memcpy(pad + ((size_t)k * 64 + r) * 64, templates + ((size_t)k * D + r) * D, (size_t)D * sizeof(double));
          pad[((size_t)k * 64 + r) * 64 + r] = 0.;
        }
      HIPCHK(hipMemcpyAsync(s.templates_pad, pad, (size_t)K * 64 * 64 * sizeof(double), hipMemcpyHostToDevice, s.stream));
    }
    HIPCHK(hipEventRecord(s.tstage_ev[slot], s.stream));
    s.tstage_busy[slot] = true;
    s.fit_static_current = false;
  }
  return 0;
}

int hyphy_hip_build_q(hyphy_hip_partition *p, int64_t n, const double *coeffs) {
  if (!p || !p->K) return fail("build_q: templates not set");
  if (n < 0 || n > std::max<int64_t>(p->C, kMixRows) * p->B || !coeffs) return fail("build_q: bad arguments");
  Trace tr("build_q");
  // When the matrices are consumed by the next hyphy_hip_evaluate_device(q_buffer) — the normal use —
  // the construction is fused into the expm kernel (no Q round trip through HBM, one launch less).
  // HYPHY_HIP_MATERIALIZE_Q=1 keeps the stand-alone kernel so that q_buffer can be read back.
  const bool fuse = getenv("HYPHY_HIP_MATERIALIZE_Q") == nullptr;
  for (Shard &s : p->shards) {
    HIPCHK(hipSetDevice(s.device));
    const size_t nbytes = (size_t)n * p->K * sizeof(double);
    const int slot = (int)(s.coeff_turn++ & 3);  // pinned ring of 4
    if (s.coeff_busy[slot]) {  // a queued expm launch may still read this slot
      HIPCHK(hipEventSynchronize(s.coeff_ev[slot]));
      s.coeff_busy[slot] = false;
    }
    double *stage = s.h_coeffs + (size_t)slot * (size_t)std::max<int64_t>(p->C, kMixRows) * p->B * p->K;
    memcpy(stage, coeffs, nbytes);
    s.coeff_slot = slot;
    s.coeff_rows = n;
    s.coeff_kind = 0;  // (unclaimed: the first evaluation that consumes the rows says what they are)
    s.qbuf_built = !fuse;
    if (fuse && s.d_hcoeffs) {
      // the fused expm kernel reads the (few hundred) coefficients straight from the pinned ring slot over
      // PCIe: no copy kernel, no extra dependency in the stream
      s.coeffs_cur = s.d_hcoeffs + (stage - s.h_coeffs);
    } else {
      s.coeffs_cur = nullptr;
      HIPCHK(hipMemcpyAsync(s.coeffs, stage, nbytes, hipMemcpyHostToDevice, s.stream));
    }
    tr.lap("memcpy");
    if (!fuse) {
      if (n > p->C * p->B) return fail("build_q: HYPHY_HIP_MATERIALIZE_Q holds at most one matrix per (class, branch)");
      launch_build_q(s.templates, s.coeffs, (int)n, (int)p->K, (int)p->D, s.qbuf, s.stream);
    }
    tr.lap("launch");
  }
  p->coeffs_pending = fuse;
  return 0;
}

// Per-site batched fits (kernel and method: sitefit.hip).  The FEL family fits site-specific rate
// multipliers (FEL.bf:593-605: fel.alpha_scaler, fel.beta_scaler_test, fel.beta_scaler_nuisance) with one
// single-site likelihood function per site; this evaluates ALL patterns of the partition, each under its own
// multipliers, for n_sets candidate parameter vectors per pattern in one launch.
static int site_fits_common(hyphy_hip_partition *p, int64_t n_sets, int64_t n_groups, int64_t n_mix,
                            const int64_t *branch_group, const double *branch_coeffs, const double *site_mult,
                            const double *site_weights, const double *root_freqs, double *site_logl_out) {
  if (!p) return fail("partition == NULL");
  if (p->nuc) {
    g_last_error = "site fits: not available for the 4-state path";
    return 1;
  }
  if (!p->K || p->templates_host.empty()) return fail("site fits: templates not set (hyphy_hip_set_q_templates)");
  if (p->K > 4) {
    g_last_error = "site fits: at most 4 templates";
    return 1;
  }
  if (n_sets < 1 || n_sets > 65535 || n_groups < 1 || n_groups > 16 || n_mix < 1 || n_mix > 8)
    return fail("site fits: bad set / group / mixture-component count");
  if (!branch_group || !branch_coeffs || !site_mult || !root_freqs || !site_logl_out || (n_mix > 1 && !site_weights))
    return fail("site fits: null argument");
  if (finish_pending_async(p)) return -1;
  switch_mode(p, 0);  // (the per-site kernel walks the partition's own tree)
  const int64_t D = p->D, B = p->B, K = p->K, S = p->S;
  const int L = (int)p->L, I = (int)p->I, DP = p->DP, NW = p->NW;
  const int NKK = 4 * NW, TILE = NKK * 64;
  for (int64_t b = 0; b < B; b++)
    if (branch_group[b] < 0 || branch_group[b] >= n_groups) return fail("site fits: branch group out of range");
  for (int64_t b = 0; b < B * K; b++)
    if (!(branch_coeffs[b] >= 0.)) return fail("site fits: branch coefficients must be non-negative");
  for (int64_t k = 0; k < n_sets * S * n_mix * n_groups * K; k++)
    if (!(site_mult[k] >= 0.)) return fail("site fits: site multipliers must be non-negative");
  if (n_mix > 1)
    for (int64_t k = 0; k < n_sets * S * n_mix; k++)
      if (!(site_weights[k] >= 0.)) return fail("site fits: mixture weights must be non-negative");

  // schedule of a full pass, compiled for this kernel's slot budget (lazy persistence: only nodes that find no
  // parking slot are stored, to the scratch copy)
  if (p->fit_ops_host.empty()) {
    std::vector<int4> keep;
    keep.swap(p->ops_host);
    const int n_slots0 = p->n_slots;
    const bool persist0 = p->sched_persist;
    p->n_slots = 2 + kSiteFitParkSlots;
    p->sched_persist = false;
    std::vector<int> all(I);
    for (int n = 0; n < I; n++) all[n] = n;
    int off = 0, n = 0;
    emit_program(p, all, &off, &n);
    p->fit_ops_host.swap(p->ops_host);
    p->ops_host.swap(keep);
    p->n_slots = n_slots0;
    p->sched_persist = persist0;
    p->fit_n_ops = n;
    p->fit_spills = false;
    for (int e = 0; e < n; e++) {
      const int4 &op = p->fit_ops_host[e];
      if ((op.x & OPF_LAST) && !(op.x & OPF_NOPERSIST) && op.y != I - 1) p->fit_spills = true;
      if ((op.x & OPF_LAST) && op.y == I - 1) p->fit_ops_host[e].x |= OPF_NOPERSIST;  // the root is consumed in registers
    }
  }
  // template images with diagonals, A-operand layout (cf. the image writer in expm.hip); dmax_k = max_i |T_k[i][i]|
  SiteFitArgs fa;
  memset(&fa, 0, sizeof(fa));
  std::vector<double> img;
  {
    std::vector<double> full((size_t)D * D);
    img.assign((size_t)K * DP * DP, 0.0);
    for (int64_t k = 0; k < K; k++) {
      const double *T = p->templates_host.data() + (size_t)k * D * D;
      double dmax = 0.;
      for (int64_t i = 0; i < D; i++) {
        double rs = 0.;
        for (int64_t j = 0; j < D; j++) {
          if (j != i && !(T[i * D + j] >= 0.)) return fail("site fits: templates must have non-negative off-diagonal entries");
          if (j != i) rs += T[i * D + j];
          full[i * D + j] = T[i * D + j];
        }
        full[i * D + i] = -rs;
        dmax = std::max(dmax, rs);
      }
      fa.dmax[k] = dmax;
      double *out = img.data() + (size_t)k * DP * DP;
      for (int idx = 0; idx < DP * DP; idx++) {
        const int wb = idx / TILE, rem = idx - wb * TILE;
        const int kk2 = rem >> 7, l = (rem >> 1) & 63, kb = rem & 1, kk = 2 * kk2 + kb;
        const int rr = 16 * wb + (l & 15), cc = 4 * kk + (l >> 4);
        out[idx] = (rr < D && cc < D) ? full[(size_t)rr * D + cc] : 0.0;
      }
    }
  }
  std::vector<int> grp(B);
  for (int64_t b = 0; b < B; b++) grp[b] = (int)branch_group[b];
  {
    // The series length of an edge grows linearly with its uniformisation rate mu = sum_k x_k dmax_k (sitefit.hip);
    // refuse parameter vectors that would keep a wave busy for seconds (rates this large mean "saturated" anyway).
    std::vector<double> cmax((size_t)n_groups * K, 0.0);  // per group and template: largest branch coefficient x dmax
    for (int64_t b = 0; b < B; b++)
      for (int64_t k = 0; k < K; k++)
        cmax[grp[b] * K + k] = std::max(cmax[grp[b] * K + k], branch_coeffs[b * K + k] * fa.dmax[k]);
    double mu_max = 0.;
    const size_t gk = (size_t)n_groups * K;
    for (int64_t r = 0; r < n_sets * S * n_mix; r++)
      for (int64_t g = 0; g < n_groups; g++) {
        double mu = 0.;
        for (int64_t k = 0; k < K; k++) mu += site_mult[r * gk + g * K + k] * cmax[g * K + k];
        mu_max = std::max(mu_max, mu);
      }
    if (!(mu_max <= kSiteFitMaxRate)) {
      char msg[160];
      snprintf(msg, sizeof msg, "site fits: uniformisation rate %.3g of some site and branch exceeds the limit %.3g "
               "(multipliers too large for this entry point)", mu_max, kSiteFitMaxRate);
      g_last_error = msg;
      return 1;
    }
  }
  std::vector<double> pi(DP, 0.0);
  for (int64_t k = 0; k < D; k++) pi[k] = root_freqs[k];
  const size_t GK = (size_t)n_mix * n_groups * K;  // multipliers per (set, pattern)

  for (Shard &s : p->shards) {
    HIPCHK(hipSetDevice(s.device));
    HIPCHK(hipStreamSynchronize(s.stream));
    if (!s.fit_Timg) {
      HIPCHK(pool_malloc((void **)&s.fit_Timg, (size_t)4 * DP * DP * sizeof(double)));
      HIPCHK(pool_malloc((void **)&s.fit_bcoef, (size_t)B * 4 * sizeof(double)));
      HIPCHK(pool_malloc((void **)&s.fit_bgroup, (size_t)B * sizeof(int)));
      HIPCHK(pool_malloc((void **)&s.fit_pi, (size_t)DP * sizeof(double)));
      HIPCHK(pool_malloc((void **)&s.fit_ops, ops_capacity(p) * sizeof(int4)));
    }
    if (!s.fit_static_current) {
      if (p->fit_ops_host.size() > ops_capacity(p)) return fail("internal: site-fit schedule overflow");
      HIPCHK(hipMemcpy(s.fit_Timg, img.data(), img.size() * sizeof(double), hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(s.fit_ops, p->fit_ops_host.data(), p->fit_ops_host.size() * sizeof(int4), hipMemcpyHostToDevice));
      s.fit_static_current = true;
    }
    const size_t need = (size_t)n_sets * std::max<size_t>(GK, 1);  // (doubles per pattern)
    if (s.fit_sets_cap < need) {
      if (s.fit_smult) pool_free_sync(s.fit_smult);
      if (s.fit_out) pool_free_sync(s.fit_out);
      s.fit_smult = s.fit_out = nullptr;
      s.fit_sets_cap = 0;
      if (s.fit_smix) pool_free_sync(s.fit_smix);
      s.fit_smix = nullptr;
      HIPCHK(pool_malloc((void **)&s.fit_smult, need * s.S_pad * sizeof(double)));
      HIPCHK(pool_malloc((void **)&s.fit_out, need * s.S_pad * sizeof(double)));
      HIPCHK(pool_malloc((void **)&s.fit_smix, need * s.S_pad * sizeof(double)));
      s.fit_sets_cap = need;
    }
    if (p->fit_spills && s.fit_scratch_sets < (size_t)n_sets) {
      if (s.fit_scratch) pool_free_sync(s.fit_scratch);
      if (s.fit_scratch_cnt) pool_free_sync(s.fit_scratch_cnt);
      s.fit_scratch = nullptr;
      s.fit_scratch_cnt = nullptr;
      s.fit_scratch_sets = 0;
      const size_t bytes = (size_t)n_sets * I * s.ntiles * TILE * sizeof(double);
      size_t free_b = 0, total_b = 0;
      hipMemGetInfo(&free_b, &total_b);
      if (bytes > free_b / 2) {
        g_last_error = "site fits: scratch for spilled nodes does not fit; use fewer parameter sets per call";
        return 1;
      }
      HIPCHK(pool_malloc((void **)&s.fit_scratch, bytes));
      HIPCHK(pool_malloc((void **)&s.fit_scratch_cnt, (size_t)n_sets * I * s.S_pad * sizeof(int32_t)));
      s.fit_scratch_sets = (size_t)n_sets;
    }
    // site multipliers of this shard's pattern range, padded with zeros (padding sites: exp(0) = I, weightless)
    std::vector<double> sm((size_t)n_sets * s.S_pad * GK, 0.0);
    for (int64_t st = 0; st < n_sets; st++)
      for (int64_t k = 0; k < s.S; k++)  // (per pattern: the device order is the sorted one)
        memcpy(sm.data() + ((size_t)st * s.S_pad + k) * GK, site_mult + ((size_t)st * S + caller_pattern(p, s.s0 + k)) * GK,
               (size_t)GK * sizeof(double));
    HIPCHK(hipMemcpyAsync(s.fit_smult, sm.data(), sm.size() * sizeof(double), hipMemcpyHostToDevice, s.stream));
    std::vector<double> wm;
    if (n_mix > 1) {  // mixture weights, same padding (they share the multipliers' allocation: need >= n_sets * n_mix)
      wm.assign((size_t)n_sets * s.S_pad * n_mix, 0.0);
      for (int64_t st = 0; st < n_sets; st++)
        for (int64_t k = 0; k < s.S; k++)
          memcpy(wm.data() + ((size_t)st * s.S_pad + k) * n_mix, site_weights + ((size_t)st * S + caller_pattern(p, s.s0 + k)) * n_mix,
                 (size_t)n_mix * sizeof(double));
      HIPCHK(hipMemcpyAsync(s.fit_smix, wm.data(), wm.size() * sizeof(double), hipMemcpyHostToDevice, s.stream));
    }
    HIPCHK(hipMemcpyAsync(s.fit_bcoef, branch_coeffs, (size_t)B * K * sizeof(double), hipMemcpyHostToDevice, s.stream));
    HIPCHK(hipMemcpyAsync(s.fit_bgroup, grp.data(), (size_t)B * sizeof(int), hipMemcpyHostToDevice, s.stream));
    HIPCHK(hipMemcpyAsync(s.fit_pi, pi.data(), (size_t)DP * sizeof(double), hipMemcpyHostToDevice, s.stream));
    HIPCHK(hipMemsetAsync(s.status, 0, sizeof(int32_t), s.stream));
    fa.ops = s.fit_ops;
    fa.n_ops = p->fit_n_ops;
    fa.NW = NW;
    fa.L = L;
    fa.I = I;
    fa.ntiles = s.ntiles;
    fa.S_pad = s.S_pad;
    fa.K = (int)K;
    fa.G = (int)n_groups;
    fa.n_sets = (int)n_sets;
    fa.Timg = s.fit_Timg;
    fa.bcoef = s.fit_bcoef;
    fa.bgroup = s.fit_bgroup;
    fa.smult = s.fit_smult;
    fa.n_mix = (int)n_mix;
    fa.smix = s.fit_smix;
    fa.codes_tile = s.codes_tile;
    fa.ambig = s.ambig;
    fa.pi = s.fit_pi;
    fa.freq = s.freq;
    fa.scratch = s.fit_scratch;
    fa.scratch_cnt = s.fit_scratch_cnt;
    fa.site_logl = s.fit_out;
    fa.status = s.status;
    HIPCHK(hipEventRecord(s.ev[0], s.stream));
    launch_site_fit(fa, s.stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(s.ev[1], s.stream));
    HIPCHK(hipStreamSynchronize(s.stream));  // (the staging vectors go out of scope)
    {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, s.ev[0], s.ev[1]) == hipSuccess) p->fit_kernel_ms = std::max(&s == &p->shards[0] ? 0.0 : p->fit_kernel_ms, (double)ms);
    }
    std::vector<double> out((size_t)n_sets * s.S_pad);
    int32_t st = 0;
    HIPCHK(hipMemcpy(out.data(), s.fit_out, out.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&st, s.status, sizeof(int32_t), hipMemcpyDeviceToHost));
    HIPCHK(hipMemsetAsync(s.status, 0, sizeof(int32_t), s.stream));
    if (st) return fail("site fits: a site likelihood is not a number");
    for (int64_t k = 0; k < n_sets; k++)
      for (int64_t j = 0; j < s.S; j++) site_logl_out[(size_t)k * S + caller_pattern(p, s.s0 + j)] = out[(size_t)k * s.S_pad + j];
  }
  return 0;
}


int hyphy_hip_site_fits_evaluate(hyphy_hip_partition *p, int64_t n_sets, int64_t n_groups, const int64_t *branch_group,
                                 const double *branch_coeffs, const double *site_mult, const double *root_freqs,
                                 double *site_logl_out) {
  return site_fits_common(p, n_sets, n_groups, 1, branch_group, branch_coeffs, site_mult, nullptr, root_freqs, site_logl_out);
}

// Branch-site mixtures per site (MEME / BS-REL style "explicit form" models): on every branch the transition
// matrix of site s is P = sum_m site_weights[s][m] exp(Q^(m)_{b,s}); the series runs once per component (sitefit.hip).
int hyphy_hip_site_fits_evaluate_mixture(hyphy_hip_partition *p, int64_t n_sets, int64_t n_groups, int64_t n_mix,
                                         const int64_t *branch_group, const double *branch_coeffs, const double *site_mult,
                                         const double *site_weights, const double *root_freqs, double *site_logl_out) {
  return site_fits_common(p, n_sets, n_groups, n_mix, branch_group, branch_coeffs, site_mult, site_weights, root_freqs,
                          site_logl_out);
}

double hyphy_hip_site_fits_kernel_ms(const hyphy_hip_partition *p) { return p ? p->fit_kernel_ms : 0.; }

double *hyphy_hip_q_buffer(hyphy_hip_partition *p) { return p && !p->shards.empty() ? p->shards[0].qbuf : nullptr; }

int hyphy_hip_synchronize(hyphy_hip_partition *p) {
  if (!p) return fail("partition == NULL");
  for (Shard &s : p->shards) {
    HIPCHK(hipSetDevice(s.device));
    HIPCHK(hipStreamSynchronize(s.stream));
  }
  return 0;
}

int hyphy_hip_set_stream(hyphy_hip_partition *p, void *stream) {
  if (!p) return fail("partition == NULL");
  if (p->shards.size() != 1) return fail("set_stream needs a single-device partition");
  Shard &s = p->shards[0];
  HIPCHK(hipSetDevice(s.device));
  HIPCHK(hipStreamSynchronize(s.stream));
  s.stream = (stream == HYPHY_HIP_OWN_STREAM) ? s.own_stream : (hipStream_t)stream;
  return 0;
}

void *hyphy_hip_stream(hyphy_hip_partition *p) { return p && !p->shards.empty() ? (void *)p->shards[0].stream : nullptr; }

int hyphy_hip_set_timing_detail(hyphy_hip_partition *p, int on) {
  if (!p) return fail("partition == NULL");
  p->all_timings = on != 0;
  return 0;
}

int hyphy_hip_last_timings(hyphy_hip_partition *p, double out[3]) {
  if (!p || !out) return fail("null argument");
  if (hipSetDevice(p->shards[0].device) != hipSuccess) return fail("hipSetDevice failed");
  hipStreamSynchronize(p->shards[0].stream);
  record_timings(p);
  for (int k = 0; k < 3; k++) out[k] = p->timings[k];
  return 0;
}

}  // extern "C"
