// RCCL over xGMI for libhyphy_hip.so: librccl is loaded on first use (a host that never all-reduces does not need it),
// plus the C-ABI entry points that sum the partition log-likelihood over ranks / devices.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

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

// ---- the collective-free combine of one-process-per-GPU runs (r06) ------------------------------------------------------------------
// What the single-process form has had since r02 — shard partials back over PCIe into host-mapped records, Neumaier on the host
// (likefunc.cpp:11046-11093) — for ranks that are separate PROCESSES of one node: every rank finishes its local evaluation exactly as a
// one-GPU run does (the reduction kernel posts the record to host-mapped memory, the host spins on its sequence word), then posts
// (value, epoch) into its slot of a POSIX shared-memory segment and reads everybody else's — a release store and N - 1 acquire loads
// between cores of one host, no device work, no launch, no stream dependency.  The in-stream ncclAllReduce costs a collective launch
// + the ring's latency (~25 us measured on one device with a one-rank communicator + the fetch kernel behind it); this costs the skew
// between the ranks' kernels and a cache-line transfer.  RCCL stays the default collective of the C-ABI (north_star) and of the
// adapter; bench.py --gpus N times both behind its timed region (collective_ab) and says which one the timed steps used.
// Slots are double-buffered by epoch parity: a rank can only be one exchange ahead of the slowest reader of its previous value.
struct XchSlot {
  double value;
  uint64_t epoch;
  uint64_t ready;
  char pad[40];
};
static_assert(sizeof(XchSlot) == 64, "one cache line per slot");
struct HostExchange {
  XchSlot *slots = nullptr;  // [2][n]
  size_t bytes = 0;
  int rank = 0, n = 1;
  uint64_t epoch = 0;
  std::string name;
};

static double xch_now() {
  timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}
static double xch_timeout_s() {
  const char *e = getenv("HYPHY_HIP_EXCHANGE_TIMEOUT_S");
  return e ? atof(e) : 120.;
}

HostExchange *xch_open(const char *name, int rank, int n_ranks) {
  if (!name || !*name || n_ranks < 1 || rank < 0 || rank >= n_ranks) {
    fail("host exchange: bad arguments");
    return nullptr;
  }
  std::string nm = std::string("/hyphy_hip_") + name;
  for (char &c : nm)
    if (&c != &nm[0] && c == '/') c = '_';
  const int fd = shm_open(nm.c_str(), O_CREAT | O_RDWR, 0600);
  if (fd < 0) {
    fail("host exchange: shm_open(" + nm + ") failed");
    return nullptr;
  }
  const size_t bytes = (size_t)2 * n_ranks * sizeof(XchSlot);
  if (ftruncate(fd, (off_t)bytes) != 0) {  // (idempotent: every rank sets the same size; a new segment reads as zeros)
    close(fd);
    fail("host exchange: ftruncate failed");
    return nullptr;
  }
  void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) {
    fail("host exchange: mmap failed");
    return nullptr;
  }
  HostExchange *x = new HostExchange;
  x->slots = static_cast<XchSlot *>(m);
  x->bytes = bytes;
  x->rank = rank;
  x->n = n_ranks;
  x->name = nm;
  // handshake: every rank marks its slots, then waits for everybody's mark (the name is unique per run: no stale marks)
  uint64_t nonce = 1469598103934665603ull;
  for (unsigned char c : nm) nonce = (nonce ^ c) * 1099511628211ull;
  nonce ^= (uint64_t)n_ranks << 48;
  if (nonce == 0) nonce = 1;
  for (int b = 0; b < 2; b++) {
    XchSlot &mine = x->slots[(size_t)b * n_ranks + rank];
    mine.value = 0.;
    __atomic_store_n(&mine.epoch, (uint64_t)0, __ATOMIC_RELAXED);
    __atomic_store_n(&mine.ready, nonce, __ATOMIC_RELEASE);
  }
  const double t0 = xch_now(), limit = xch_timeout_s();
  for (int r = 0; r < n_ranks; r++)
    for (int b = 0; b < 2; b++)
      while (__atomic_load_n(&x->slots[(size_t)b * n_ranks + r].ready, __ATOMIC_ACQUIRE) != nonce) {
        if (xch_now() - t0 > limit) {
          munmap(m, bytes);
          delete x;
          fail("host exchange: the other ranks did not attach to " + nm);
          return nullptr;
        }
        usleep(200);
      }
  return x;
}

void xch_close(HostExchange *x) {
  if (!x) return;
  if (x->rank == 0) shm_unlink(x->name.c_str());  // (the mappings keep the segment alive until the last rank has gone)
  munmap(x->slots, x->bytes);
  delete x;
}

// one exchange: post this rank's value (NaN when its local evaluation failed), wait for everybody's, sum in rank order with the
// reference's compensated combine — every rank computes the same bits
int xch_sum(HostExchange *x, double local, bool local_failed, double *sum_out) {
  if (!x) return fail("host exchange: not initialised");
  const uint64_t e = ++x->epoch;
  XchSlot *row = x->slots + (size_t)(e & 1) * x->n;
  row[x->rank].value = local_failed ? NAN : local;
  __atomic_store_n(&row[x->rank].epoch, e, __ATOMIC_RELEASE);
  std::vector<double> parts((size_t)x->n);
  const double limit = xch_timeout_s();
  double t0 = 0.;
  for (int r = 0; r < x->n; r++) {
    for (long spins = 1; __atomic_load_n(&row[r].epoch, __ATOMIC_ACQUIRE) != e; spins++) {
      if ((spins & 0xfffff) == 0) {
        if (t0 == 0.) t0 = xch_now();
        else if (xch_now() - t0 > limit) return fail("host exchange: a rank did not arrive (HYPHY_HIP_EXCHANGE_TIMEOUT_S)");
      }
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    }
    parts[(size_t)r] = row[r].value;
  }
  if (sum_out) *sum_out = combine(parts);
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

/* ---- collective-free combine for one process per GPU on ONE node (see HostExchange above) --------------------------------------
 *   hyphy_hip_comm_init_host   attach this rank's partition to the run's shared-memory exchange (`name`: the same string on every rank,
 *                              unique per run); returns when every rank has attached.
 *   hyphy_hip_evaluate(_built)_exchange   the local evaluation of hyphy_hip_evaluate(_built) + ONE host-side exchange: every rank
 *                              returns the log-likelihood of the whole alignment (same bits on every rank).  A rank whose local
 *                              evaluation fails still posts (NaN) and then returns its error: nobody is left waiting.
 *   hyphy_hip_xch_open / _sum / _close   the exchange alone, without a partition or a device (CPU tests; hosts that sum something else). */
int hyphy_hip_comm_init_host(hyphy_hip_partition *p, const char *name, int rank, int n_ranks) {
  if (!p) return fail("partition == NULL");
  if (p->shards.size() != 1) return fail("comm_init_host: one device per rank (device_count = 1)");
  HostExchange *x = xch_open(name, rank, n_ranks);
  if (!x) return -1;
  if (p->xch) xch_close(static_cast<HostExchange *>(p->xch));
  p->xch = x;
  return 0;
}

static int exchange_after(hyphy_hip_partition *p, int local_rc, double *logl_out) {
  const std::string local_error = g_last_error;
  double local = 0., total = 0.;
  bool failed = local_rc != 0;
  if (!failed) {
    if (collect_status(p)) failed = true;
    else local = p->shards[0].h_out[0];
  }
  const std::string err2 = g_last_error;
  const int xrc = xch_sum(static_cast<HostExchange *>(p->xch), local, failed, &total);
  if (local_rc) {
    g_last_error = local_error;
    return -1;
  }
  if (failed) {
    g_last_error = err2;
    return -1;
  }
  if (xrc) return -1;
  record_timings(p);
  if (logl_out) *logl_out = total;
  return 0;
}

int hyphy_hip_evaluate_exchange(hyphy_hip_partition *p, int64_t cat, const int64_t *update_nodes, int64_t n_update, const int64_t *q_nodes,
                                int64_t n_q, const double *q_dense, int q_is_probability, const double *root_freqs, double *logl_out) {
  if (!p) return fail("partition == NULL");
  if (p->shards.size() != 1 || !p->xch) return fail("evaluate_exchange: hyphy_hip_comm_init_host first");
  const int rc = eval_common(p, cat, update_nodes, n_update, q_nodes, n_q, q_dense, false, q_is_probability, root_freqs, nullptr, true, false);
  return exchange_after(p, rc, logl_out);
}

int hyphy_hip_evaluate_built_exchange(hyphy_hip_partition *p, int64_t cat, const int64_t *update_nodes, int64_t n_update,
                                      const int64_t *q_nodes, int64_t n_q, const double *root_freqs, double *logl_out) {
  if (!p) return fail("partition == NULL");
  if (!p->K) return fail("evaluate_built_exchange: templates not set");
  if (p->shards.size() != 1 || !p->xch) return fail("evaluate_built_exchange: hyphy_hip_comm_init_host first");
  const int rc = eval_common(p, cat, update_nodes, n_update, q_nodes, n_q, &kOwnQBuffer, true, 0, root_freqs, nullptr, true, false);
  return exchange_after(p, rc, logl_out);
}

int hyphy_hip_xch_open(const char *name, int rank, int n_ranks, void **handle_out) {
  if (!handle_out) return fail("host exchange: null handle pointer");
  HostExchange *x = xch_open(name, rank, n_ranks);
  if (!x) return -1;
  *handle_out = x;
  return 0;
}
int hyphy_hip_xch_sum(void *handle, double local, int local_failed, double *sum_out) {
  return xch_sum(static_cast<HostExchange *>(handle), local, local_failed != 0, sum_out);
}
void hyphy_hip_xch_close(void *handle) { xch_close(static_cast<HostExchange *>(handle)); }

}  // extern "C"
