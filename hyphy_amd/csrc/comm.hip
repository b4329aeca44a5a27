// RCCL over xGMI for libhyphy_hip.so: librccl is loaded on first use (a host that never all-reduces does not need it),
// plus the C-ABI entry points that sum the partition log-likelihood over ranks / devices.
#include <dlfcn.h>

#include "partition.h"

namespace hyhip {

Rccl g_rccl;
rccl_init_rank_fn g_rccl_init_rank = nullptr;

int rccl_load() {
  if (g_rccl.lib) return 0;
  // RTLD_LOCAL: a host process may carry a librccl of its own (PyTorch bundles one) — ours must not interpose on it.
  void *h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!h) {
    const char *e = dlerror();  // (ONE call: dlerror() clears the message it returns)
    return fail(std::string("RCCL not available: ") + (e ? e : "librccl.so"));
  }
  g_rccl.GetUniqueId = (int (*)(void *))dlsym(h, "ncclGetUniqueId");
  g_rccl_init_rank = (rccl_init_rank_fn)dlsym(h, "ncclCommInitRank");
  g_rccl.CommInitAll = (int (*)(void **, int, const int *))dlsym(h, "ncclCommInitAll");
  g_rccl.CommDestroy = (int (*)(void *))dlsym(h, "ncclCommDestroy");
  g_rccl.AllReduce = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(h, "ncclAllReduce");
  g_rccl.GroupStart = (int (*)())dlsym(h, "ncclGroupStart");
  g_rccl.GroupEnd = (int (*)())dlsym(h, "ncclGroupEnd");
  g_rccl.GetErrorString = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
  if (!g_rccl.GetUniqueId || !g_rccl_init_rank || !g_rccl.CommInitAll || !g_rccl.CommDestroy || !g_rccl.AllReduce ||
      !g_rccl.GroupStart || !g_rccl.GroupEnd)
    return fail("RCCL: missing symbols in librccl.so");
  g_rccl.lib = h;
  return 0;
}

// Sum of the shard partials of a single-process, multi-device partition (every shard's record has been collected):
// the reference's Neumaier combine on the host (likefunc.cpp:11046-11093), or — HYPHY_HIP_COMBINE=rccl behind
// hyphy_hip_comm_init_all — ONE group all-reduce over xGMI (every shard ends up with the total; shard 0's copy is returned).
int combine_shards(hyphy_hip_partition *p, double *logl_out) {
  if (!logl_out) return 0;
  if (p->shards.size() == 1) {
    *logl_out = p->shards[0].h_out[0];
    return 0;
  }
  std::vector<double> parts;
  for (Shard &s : p->shards) parts.push_back(s.h_out[0]);
  *logl_out = combine(parts);
  const char *mode = getenv("HYPHY_HIP_COMBINE");
  if (!(mode && !strcmp(mode, "rccl") && p->shards[0].comm)) return 0;
  for (Shard &s : p->shards) {
    HIPCHK(hipSetDevice(s.device));
    HIPCHK(hipMemcpyAsync(s.ar_buf, &s.h_out[0], sizeof(double), hipMemcpyHostToDevice, s.stream));
  }
  RCCLCHK(g_rccl.GroupStart());
  for (Shard &s : p->shards) {
    HIPCHK(hipSetDevice(s.device));
    RCCLCHK(g_rccl.AllReduce(s.ar_buf, s.ar_buf, 1, kNcclDouble, kNcclSum, s.comm, s.stream));
  }
  RCCLCHK(g_rccl.GroupEnd());
  Shard &s0 = p->shards[0];
  HIPCHK(hipSetDevice(s0.device));
  double tot = 0.;
  HIPCHK(hipMemcpyAsync(&tot, s0.ar_buf, sizeof(double), hipMemcpyDeviceToHost, s0.stream));
  for (Shard &s : p->shards) {
    HIPCHK(hipSetDevice(s.device));
    HIPCHK(hipStreamSynchronize(s.stream));
  }
  *logl_out = tot;
  return 0;
}

}  // namespace hyhip

using namespace hyhip;

extern "C" {

/* ---- the all-reduce of the partition log-likelihood, over RCCL / xGMI, where a C++ host can reach it -------------------
 * One process per GPU (HYPHYMPI-style hosts, `torchrun`-style launchers): rank 0 makes a 128-byte id
 * (hyphy_hip_comm_unique_id), every rank receives it by whatever channel the host has (MPI_Bcast, a file) and calls
 * hyphy_hip_comm_init_rank on its partition (which holds ITS shard of the patterns); hyphy_hip_evaluate_allreduce is then
 * hyphy_hip_evaluate + ONE ncclAllReduce of one double per evaluation, enqueued on the partition's stream between the
 * reduction kernel and the read-back: every rank returns the log-likelihood of the whole alignment. */
int hyphy_hip_comm_unique_id(void *out128) {
  if (!out128) return fail("null id buffer");
  if (rccl_load()) return -1;
  RCCLCHK(g_rccl.GetUniqueId(out128));
  return 0;
}

int hyphy_hip_comm_init_rank(hyphy_hip_partition *p, const void *unique_id, int rank, int n_ranks) {
  if (!p || !unique_id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail("comm_init_rank: bad arguments");
  if (p->shards.size() != 1) return fail("comm_init_rank: one device per rank (device_count = 1)");
  if (rccl_load()) return -1;
  Shard &s = p->shards[0];
  HIPCHK(hipSetDevice(s.device));
  if (s.comm) {
    g_rccl.CommDestroy(s.comm);
    s.comm = nullptr;
  }
  RcclUniqueId id;
  memcpy(id.internal, unique_id, sizeof id.internal);
  RCCLCHK(g_rccl_init_rank(&s.comm, n_ranks, id, rank));
  if (!s.ar_buf) HIPCHK(pool_malloc((void **)&s.ar_buf, 2 * sizeof(double)));
  return 0;
}

/* In-place sum of one device double over the partition's communicator, on the partition's stream (asynchronous). */
int hyphy_hip_allreduce_device(hyphy_hip_partition *p, double *d_value) {
  if (!p || !d_value) return fail("allreduce: null argument");
  if (p->shards.size() != 1 || !p->shards[0].comm) return fail("allreduce: hyphy_hip_comm_init_rank first");
  Shard &s = p->shards[0];
  HIPCHK(hipSetDevice(s.device));
  RCCLCHK(g_rccl.AllReduce(d_value, d_value, 1, kNcclDouble, kNcclSum, s.comm, s.stream));
  return 0;
}

// The collective + read-back behind a local evaluation that left this rank's partial in s.ar_buf.  A rank whose local
// evaluation FAILED (validation, a HIP error) still joins the collective — with NaN — so that the other ranks are not left
// waiting in it; every rank then sees NaN and the failing one returns its own error.
static int allreduce_and_fetch(hyphy_hip_partition *p, int local_rc, double *logl_out) {
  Shard &s = p->shards[0];
  std::string local_error = g_last_error;
  // Nothing returns before the all-reduce has been enqueued: the peers read their result by spinning on a host-mapped record
  // and would never leave the collective.  Failures on the way (device selection, the NaN upload, timing events) only mark
  // this rank as failed; the timing stamps are optional.
  if (hipSetDevice(s.device) != hipSuccess && !local_rc) {
    local_rc = fail("hipSetDevice failed");
    local_error = g_last_error;
  }
  if (local_rc) {
    static const double kNaN = NAN;
    (void)hipMemcpyAsync(s.ar_buf, &kNaN, sizeof(double), hipMemcpyHostToDevice, s.stream);
  }
  bool stamp = p->all_timings;
  if (stamp) {
    for (auto &e : s.ev_ar)
      if (!e && hipEventCreate(&e) != hipSuccess) stamp = false;
    if (stamp && hipEventRecord(s.ev_ar[0], s.stream) != hipSuccess) stamp = false;
  }
  {
    const int arc = g_rccl.AllReduce(s.ar_buf, s.ar_buf, 1, kNcclDouble, kNcclSum, s.comm, s.stream);
    if (arc != 0) {  // could not even enqueue: the peers cannot be helped from here
      const std::string why = std::string("ncclAllReduce: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(arc) : "error");
      if (local_rc) {
        g_last_error = local_error + "; " + why;
        return -1;
      }
      return fail(why);
    }
  }
  if (stamp && hipEventRecord(s.ev_ar[1], s.stream) != hipSuccess) stamp = false;
  double v = 0.;
  const int rc = publish_and_collect(p, s.ar_buf, &v);  // (host-mapped record: no copy command, no stream synchronisation)
  if (stamp) {
    float ms = 0.f;
    if (hipEventSynchronize(s.ev_ar[1]) == hipSuccess && hipEventElapsedTime(&ms, s.ev_ar[0], s.ev_ar[1]) == hipSuccess) p->allreduce_ms = ms;
  }
  if (local_rc) {
    g_last_error = local_error;
    return -1;
  }
  if (rc) return -1;
  if (logl_out) *logl_out = v;
  return 0;
}

int hyphy_hip_evaluate_allreduce(hyphy_hip_partition *p, int64_t cat, const int64_t *update_nodes, int64_t n_update,
                                 const int64_t *q_nodes, int64_t n_q, const double *q_dense, int q_is_probability,
                                 const double *root_freqs, double *logl_out) {
  if (!p) return fail("partition == NULL");
  if (p->shards.size() != 1 || !p->shards[0].comm) return fail("evaluate_allreduce: hyphy_hip_comm_init_rank first");
  Shard &s = p->shards[0];
  // partial log-L of this rank's patterns into a device scalar, summed over the ranks in-stream, one double back
  const int rc = eval_common(p, cat, update_nodes, n_update, q_nodes, n_q, q_dense, false, q_is_probability, root_freqs, s.ar_buf, true, false);
  return allreduce_and_fetch(p, rc, logl_out);
}

/* The same behind hyphy_hip_build_q (template models: coefficients, not matrices, cross PCIe): the step a site-sharded
 * likelihood function takes per evaluation — local rate-matrix construction + exponentials + pruning + reduction, ONE
 * ncclAllReduce of one double on the partition's stream, the reduced value back through the host-mapped record. */
int hyphy_hip_evaluate_built_allreduce(hyphy_hip_partition *p, int64_t cat, const int64_t *update_nodes, int64_t n_update,
                                       const int64_t *q_nodes, int64_t n_q, const double *root_freqs, double *logl_out) {
  if (!p) return fail("partition == NULL");
  if (!p->K) return fail("evaluate_built_allreduce: templates not set");
  if (p->shards.size() != 1 || !p->shards[0].comm) return fail("evaluate_built_allreduce: hyphy_hip_comm_init_rank first");
  Shard &s = p->shards[0];
  const int rc = eval_common(p, cat, update_nodes, n_update, q_nodes, n_q, &kOwnQBuffer, true, 0, root_freqs, s.ar_buf, true, false);
  return allreduce_and_fetch(p, rc, logl_out);
}

double hyphy_hip_last_allreduce_ms(const hyphy_hip_partition *p) { return p ? p->allreduce_ms : 0.; }

/* Single-process hosts with device_count > 1 (HyPhy proper): by default the shard partials come back over PCIe and are
 * summed on the host with the reference's Neumaier combine; HYPHY_HIP_COMBINE=rccl (or this call) makes one RCCL group
 * all-reduce of it instead — SURVEY 5 asks for both to be measurable. */
int hyphy_hip_comm_init_all(hyphy_hip_partition *p) {
  if (!p) return fail("partition == NULL");
  if (rccl_load()) return -1;
  const int n = (int)p->shards.size();
  std::vector<int> devs(n);
  std::vector<void *> comms(n, nullptr);
  for (int k = 0; k < n; k++) devs[k] = p->shards[k].device;
  for (int k = 0; k < n; k++)
    for (int j = 0; j < k; j++)
      if (devs[k] == devs[j]) return fail("comm_init_all: RCCL needs one distinct device per shard");
  RCCLCHK(g_rccl.CommInitAll(comms.data(), n, devs.data()));
  for (int k = 0; k < n; k++) {
    Shard &s = p->shards[k];
    HIPCHK(hipSetDevice(s.device));
    if (s.comm) g_rccl.CommDestroy(s.comm);
    s.comm = comms[k];
    if (!s.ar_buf) HIPCHK(pool_malloc((void **)&s.ar_buf, 2 * sizeof(double)));
  }
  return 0;
}

}  // extern "C"
