// 4 states: the schedule of a partition's tree as STRAIGHT-LINE code, compiled at run time (hiprtc) for gfx950.
//
// prune_nuc2_kernel (prune.hip) INTERPRETS the schedule: per entry it fetches a schedule word, decodes kind / flags / slots, loads
// the branch's matrix through scalar loads it could only request one entry ahead, parks and un-parks pending nodes in LDS — 43
// vector + 36 scalar instructions per entry for the ~20 FP64 operations of the reference's 4-state step
// (_handle4x4_pruning_case_direct, src/core/tree_evaluator.cpp:2253-2273; the loop it sits in, :3556-4171).  Here the schedule
// compiler's entries (common.h: entry format; schedule.hip: emit_program) are turned into the source of ONE kernel per (topology,
// update set, persistence flags): node order fixed, every leaf number, branch slot and plane offset an immediate, a pending node a
// named value the register allocator places (no parking slots), a branch's matrix twelve scalar-load constants, the first factor of
// a node an assignment instead of a multiplication by one, all leaf codes of a pattern requested up front.  What stays exactly as
// in the interpreter: the order of the factors inside a node, the row-stochastic 12-multiply-add form of an internal edge, the
// leaf lookups in the transposed leaf matrices (LDS), the per-node 2^64 rescale test, the epilogue — the two forms are held to each
// other bit for bit in tests/test_gpu_nucgen.py.
//
// Life cycle.  A kernel is worth its compilation (1-2 s of one host core) only for a schedule that keeps coming back: a partition
// asks for one after `kNucGenAfter` evaluations under the same full-pass schedule (HYPHY_HIP_NUCGEN_AFTER), the source is compiled
// by a background thread (the evaluations go on under the interpreter), the code object is kept per process and keyed by the
// source's hash (every partition over the same topology shares it), loaded per device on first use.  HYPHY_HIP_NUCGEN=0: off;
// =2: compile synchronously on the first request (tests, benchmarks that must not depend on timing).  No device, no hiprtc, a
// compilation error: the interpreter stays (HYPHY_HIP_VERBOSE says why).
#include <dlfcn.h>
#include <hip/hiprtc.h>

#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>

#include <stdarg.h>

#include "partition.h"
#include "nucgen_embed.h"  // kEmbedExpm4, kEmbedCombine: expm4.h / combine.h as string constants (Makefile)

namespace hyhip {

namespace {

struct Rtc {
  void *lib = nullptr;
  hiprtcResult (*create)(hiprtcProgram *, const char *, const char *, int, const char **, const char **) = nullptr;
  hiprtcResult (*compile)(hiprtcProgram, int, const char **) = nullptr;
  hiprtcResult (*code_size)(hiprtcProgram, size_t *) = nullptr;
  hiprtcResult (*code)(hiprtcProgram, char *) = nullptr;
  hiprtcResult (*log_size)(hiprtcProgram, size_t *) = nullptr;
  hiprtcResult (*log)(hiprtcProgram, char *) = nullptr;
  hiprtcResult (*destroy)(hiprtcProgram *) = nullptr;
  bool ok = false;
};

Rtc &rtc() {
  static Rtc r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char *name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.lib) break;
    }
    if (!r.lib) return;
    r.create = reinterpret_cast<decltype(r.create)>(dlsym(r.lib, "hiprtcCreateProgram"));
    r.compile = reinterpret_cast<decltype(r.compile)>(dlsym(r.lib, "hiprtcCompileProgram"));
    r.code_size = reinterpret_cast<decltype(r.code_size)>(dlsym(r.lib, "hiprtcGetCodeSize"));
    r.code = reinterpret_cast<decltype(r.code)>(dlsym(r.lib, "hiprtcGetCode"));
    r.log_size = reinterpret_cast<decltype(r.log_size)>(dlsym(r.lib, "hiprtcGetProgramLogSize"));
    r.log = reinterpret_cast<decltype(r.log)>(dlsym(r.lib, "hiprtcGetProgramLog"));
    r.destroy = reinterpret_cast<decltype(r.destroy)>(dlsym(r.lib, "hiprtcDestroyProgram"));
    r.ok = r.create && r.compile && r.code_size && r.code && r.log_size && r.log && r.destroy;
  });
  return r;
}

bool verbose() { return getenv("HYPHY_HIP_VERBOSE") != nullptr; }

// ---- the generator ------------------------------------------------------------------------------------------------------------
const char *kPrologue = R"SRC(
typedef double f64x2 __attribute__((ext_vector_type(2)));
struct GenArgs {   // = hyhip::NucGenArgs (common.h)
  const double *ambig; double *partials; int *counts; const double *pi; double *site_lik; int *site_cnt; const double *freq;
  double *wg_sum; long long *wg_cnt; int *wg_flag; long long S_pad;
  double *red_out; double *red_rec; const int *red_status; double red_seq; int *red_done;   /* fused final combine (small-shard form) */
};
#define T_ 5.42101086242752217e-20      /* 2^-64  _lfScalingFactorThreshold (src/core/tree.cpp:126-129) */
#define U_ 18446744073709551616.0       /* 2^64   _lfScalerUpwards */
static __device__ __forceinline__ int rescale_(double tot, double &sc) {   /* = rescale_decision (devutil.h) */
  int m = 0;
  sc = 1.0;
  if (tot < T_ && tot > 0.0) {
    do { tot *= U_; sc *= U_; m++; } while (tot < T_ && m < 15);
  } else if (tot > U_ && tot < __builtin_huge_val()) {
    do { tot *= T_; sc *= T_; m--; } while (tot > U_ && m > -15);
  }
  return m;
}
/* a leaf that may carry an ambiguity code: M = its transposed matrix in LDS ([state j][row i]) */
static __device__ __forceinline__ void leaf_general_(const double *M, int code, const double *ambig, double &f0, double &f1, double &f2, double &f3) {
  double cv[4];
  if (code >= 0) {
    for (int j = 0; j < 4; j++) cv[j] = (j == code) ? 1. : 0.;
  } else {
    for (int j = 0; j < 4; j++) cv[j] = ambig[(size_t)(-code - 1) * 4 + j];
  }
  double m[4];
  for (int i = 0; i < 4; i++) {
    double x = M[i] * cv[0];
    x = fma(M[4 + i], cv[1], x);
    x = fma(M[8 + i], cv[2], x);
    x = fma(M[12 + i], cv[3], x);
    m[i] = x;
  }
  f0 = m[0], f1 = m[1], f2 = m[2], f3 = m[3];
}
)SRC";

struct Src {
  std::string s;
  void operator()(const char *fmt, ...) __attribute__((format(printf, 2, 3))) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    s += buf;
  }
};

}  // namespace

// The source of the kernel that runs program `ops[0 .. n_ops)` (one program: a 4-state partition's schedule is never cut).
// lazy: the pass persists nothing for later passes (lazy persistence) — the only stores its schedule asks for are of nodes
// a later entry of the SAME pass re-reads for want of a parking slot, and here such a node is a named value: no store, no re-read.
// Returns an empty string when the program has a form the generator does not cover.
// small (the form for shards of at most two workgroups per CU, where a wave is alone on its SIMD and every latency is exposed):
//   * EVERY branch's matrix sits in LDS (n_branches of them) and an internal edge reads its twelve entries from there — LDS reads
//     return in order after ~100 cycles where a scalar load that misses the scalar cache takes several hundred and returns out of order;
//   * this evaluation's matrix exponentials are computed inside the launch by the first threads of every workgroup, straight into
//     the LDS copy (expm4.h, the library's own source), workgroup 0 leaves the global copies for later partial updates;
//   * the last workgroup to arrive sums the per-workgroup partial results and publishes the record (combine.h) —
//   the interpreter's FOLD / LP forms (prune.hip: prune_nuc2_kernel), so that the step is ONE launch.
std::string nucgen_source(const int4 *ops, int n_ops, int L, bool lazy, bool small, int n_branches) {
  if (n_ops <= 0 || L < 1 || L > 256 || (small && (n_branches < L || n_branches > 1024))) return std::string();
  Src o;
  if (small) {
    o.s = "#define HYPHY_NUCGEN_EMBEDDED 1\n#ifndef NAN\n#define NAN __builtin_nan(\"\")\n#endif\n#ifndef INFINITY\n#define INFINITY __builtin_huge_val()\n#endif\nnamespace hyhip {\ntypedef double f64x2 __attribute__((ext_vector_type(2)));\n"
          "typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));\nconstexpr double kLogScaler = 64.0 * 0.69314718055994530942;\n"
          "struct ExpmArgs {   /* = hyhip::NucGenExpm (common.h): what expm4_one and the folded prologue read of the library's ExpmArgs */\n"
          "  const double *Q; const int *slots; int n; int is_prob; double *Prow; double *PTrow; int *status; const double *templates;\n"
          "  const double *coeffs; int K; int coef_inline;\n};\nstruct CoefInline { double c[400]; };\n}\n";
    static_assert(kCoefInline == 400, "the generated source declares CoefInline with 400 entries");
    o.s += kEmbedExpm4;
    o.s += kEmbedCombine;
  }
  o.s += kPrologue;
  if (small) o("extern \"C\" __global__ __launch_bounds__(256) void nucgen_kernel(const double *__restrict__ PT, const short *__restrict__ codes, GenArgs a, hyhip::ExpmArgs ex, hyhip::CoefInline ci) {\n");
  else o("extern \"C\" __global__ __launch_bounds__(256) void nucgen_kernel(const double *__restrict__ PT, const short *__restrict__ codes, GenArgs a) {\n");
  o("  extern __shared__ __align__(16) double L_[];   /* transposed matrices [branch][state j][row i]: %s */\n", small ? "every branch" : "the leaves (branches 0 .. L-1)");
  o("  const int tid = threadIdx.x;\n");
  o("  const size_t SP = (size_t)a.S_pad, s = (size_t)blockIdx.x * 256 + tid;\n");
  // every leaf code of this pattern, requested before anything waits (in the small form: under the exponentials)
  std::vector<char> leaf_used(L, 0);
  for (int k = 0; k < n_ops; k++) {
    const int4 op = ops[k];
    if ((op.x & 3) != OPK_LEAF) continue;
    const int nl = (op.x >> 8) & 0x7f;
    for (int i = 0; i < nl && i < 2; i++) {
      const int lf = (op.z >> (16 * i)) & 0xffff;
      if (lf >= L) return std::string();
      leaf_used[lf] = 1;
    }
    if (nl > 2) return std::string();
  }
  for (int lf = 0; lf < L; lf++)
    if (leaf_used[lf]) o("  const int k%d = codes[(size_t)%d * SP + s];\n", lf, lf);
  if (small) {
    // (a full pass re-exponentiates every branch: nothing of the resident copies survives, so they are not fetched at all)
    o("  if (ex.n < %d) {\n    for (int i = tid; i < %d; i += 256) L_[i] = PT[i];\n  }\n", n_branches, n_branches * 16);
    o("  if (ex.n > 0) {   /* this evaluation's exponentials, over the copy of the resident matrices */\n    __syncthreads();\n");
    o("    for (int m = tid; m < ex.n; m += 256) {\n      double R[16];\n      hyhip::expm4_one(ex, m, R, ex.coef_inline ? ci.c : ex.coeffs);\n");
    o("      const int slot = ex.slots ? ex.slots[m] : m;\n");
    o("      for (int k = 0; k < 16; k++) L_[slot * 16 + k] = R[4 * (k & 3) + (k >> 2)];\n");
    o("      if (blockIdx.x == 0) {\n        for (int k = 0; k < 16; k++) {\n          if (ex.Prow) ex.Prow[(size_t)slot * 16 + k] = R[k];\n");
    o("          ex.PTrow[(size_t)slot * 16 + k] = R[4 * (k & 3) + (k >> 2)];\n        }\n      }\n    }\n  }\n");
  } else {
    o("  for (int i = tid; i < %d; i += 256) L_[i] = PT[i];\n", L * 16);
  }
  o("  __syncthreads();\n");
  std::vector<char> have;  // node finalised earlier in this program: its values are n<node>_0..3, c<node>
  // Rescale tests only where underflow is possible (the rule of the codon kernels, schedule.hip: thin_rescale_tests): a node whose
  // internal children were all TESTED in this pass (their per-pattern sums are >= 2^-64 behind the test) and that has at most four
  // factors cannot fall below 2^-256 times the spread of a conditional vector — hundreds of binary orders above the denormals; its
  // parent tests again, the last node of the program always does.  A rescale is an exact power of 2^64 and conditionals only shrink
  // on the way up, so the (value, exponent) pair the root ends with is the same wherever the steps were taken.
  static const bool thin = !(getenv("HYPHY_HIP_SCALE_THIN") && atoi(getenv("HYPHY_HIP_SCALE_THIN")) == 0);
  std::vector<char> tested;
  int last_closing = -1;
  for (int q = 0; q < n_ops; q++)
    if (ops[q].x & OPF_LAST) last_closing = q;
  int last_node = -1;
  int e = 0;  // factor counter
  int k = 0;
  while (k < n_ops) {
    // the entries of one parent: up to and including the OPF_LAST one
    std::vector<std::string> f[4];
    std::vector<std::string> cterms;
    int par = -1;
    bool closed = false, kids_tested = true;
    for (; k < n_ops && !closed; k++) {
      const int4 op = ops[k];
      const int kind = op.x & 3;
      if (kind == OPK_LEAF) {
        const int nl = (op.x >> 8) & 0x7f;
        if (nl == 0) continue;  // padding
        par = op.y;
        for (int i = 0; i < nl; i++) {
          const int lf = (op.z >> (16 * i)) & 0xffff;
          if (op.x & OPF_AMBIG) {
            o("  double f%d_0, f%d_1, f%d_2, f%d_3;\n", e, e, e, e);
            o("  if (!__any(k%d < 0)) {\n", lf);
            o("    const f64x2 *q = reinterpret_cast<const f64x2 *>(L_ + %d + k%d * 4);\n", lf * 16, lf);
            o("    const f64x2 u = q[0], w = q[1];\n    f%d_0 = u[0], f%d_1 = u[1], f%d_2 = w[0], f%d_3 = w[1];\n", e, e, e, e);
            o("  } else {\n    leaf_general_(L_ + %d, k%d, a.ambig, f%d_0, f%d_1, f%d_2, f%d_3);\n  }\n", lf * 16, lf, e, e, e, e);
          } else {
            o("  const f64x2 *q%d = reinterpret_cast<const f64x2 *>(L_ + %d + k%d * 4);\n", e, lf * 16, lf);
            o("  const f64x2 u%d = q%d[0], w%d = q%d[1];\n", e, e, e, e);
            o("  const double f%d_0 = u%d[0], f%d_1 = u%d[1], f%d_2 = w%d[0], f%d_3 = w%d[1];\n", e, e, e, e, e, e, e, e);
          }
          for (int i2 = 0; i2 < 4; i2++) f[i2].push_back("f" + std::to_string(e) + "_" + std::to_string(i2));
          e++;
        }
      } else {
        if (kind == OPK_DEP) return std::string();
        par = op.y;
        const int c = op.w, br = op.z;
        if (c < 0) return std::string();
        if ((size_t)c >= have.size()) have.resize((size_t)c + 1, 0);
        if ((size_t)c >= tested.size()) tested.resize((size_t)c + 1, 0);
        if (!(have[c] && tested[c])) kids_tested = false;  // (a child read back from its persisted copy counts as untested)
        std::string v[4], vc;
        if (have[c]) {
          for (int j = 0; j < 4; j++) v[j] = "n" + std::to_string(c) + "_" + std::to_string(j);
          vc = "c" + std::to_string(c);
        } else {  // the persisted copy (a partial update: the child was not touched)
          o("  const size_t b%d = (size_t)%d * 4 * SP + s;\n", e, c);
          o("  const double g%d_0 = a.partials[b%d], g%d_1 = a.partials[b%d + SP], g%d_2 = a.partials[b%d + 2 * SP], g%d_3 = a.partials[b%d + 3 * SP];\n", e, e,
            e, e, e, e, e, e);
          o("  const int gc%d = a.counts[(size_t)%d * SP + s];\n", e, c);
          for (int j = 0; j < 4; j++) v[j] = "g" + std::to_string(e) + "_" + std::to_string(j);
          vc = "gc" + std::to_string(e);
        }
        // (P v)_i = (d0 P_i0 + d1 P_i1) + (v3 + d2 P_i2), d_j = v_j - v3: rows of P sum to one.  PT = [state j][row i]
        if (small && br >= n_branches) return std::string();
        o("  const double *P%d = %s + %d;\n", e, small ? "L_" : "PT", br * 16);
        o("  const double d%d_0 = %s - %s, d%d_1 = %s - %s, d%d_2 = %s - %s;\n", e, v[0].c_str(), v[3].c_str(), e, v[1].c_str(), v[3].c_str(), e, v[2].c_str(),
          v[3].c_str());
        for (int i = 0; i < 4; i++)
          o("  const double f%d_%d = fma(d%d_1, P%d[%d], d%d_0 * P%d[%d]) + fma(d%d_2, P%d[%d], %s);\n", e, i, e, e, 4 + i, e, e, i, e, e, 8 + i, v[3].c_str());
        for (int i2 = 0; i2 < 4; i2++) f[i2].push_back("f" + std::to_string(e) + "_" + std::to_string(i2));
        cterms.push_back(vc);
        e++;
      }
      if (op.x & OPF_LAST) {
        if (par < 0 || f[0].empty()) return std::string();
        if ((size_t)par >= have.size()) have.resize((size_t)par + 1, 0);
        for (int i = 0; i < 4; i++) {
          std::string prod = f[i][0];
          for (size_t t = 1; t < f[i].size(); t++) prod = "(" + prod + ") * " + f[i][t];  // (the interpreter's order: left to right)
          o("  double n%d_%d = %s;\n", par, i, prod.c_str());
        }
        std::string csum = "0";
        for (const std::string &t : cterms) csum += " + " + t;
        o("  int c%d = %s;\n", par, csum.c_str());
        if ((size_t)par >= tested.size()) tested.resize((size_t)par + 1, 0);
        const bool test_here = !thin || k == last_closing || !kids_tested || f[0].size() > 4;
        tested[par] = test_here ? 1 : 0;
        if (test_here) {
          o("  {\n    const double tot = (n%d_0 + n%d_1) + (n%d_2 + n%d_3);\n", par, par, par, par);
          o("    if (__any(!(tot >= T_ && tot <= U_))) {   /* rare: some pattern of the wave needs (or cannot have) a rescale */\n");
          o("      double sc;\n      const int m = rescale_(tot, sc);\n");
          o("      if (m != 0) { n%d_0 *= sc; n%d_1 *= sc; n%d_2 *= sc; n%d_3 *= sc; c%d += m; }\n    }\n  }\n", par, par, par, par, par);
        } else {
          // (no test: keep the node's products apart from what reads them — under -ffp-contract=fast the differences d = v - v3 of the
          //  parent's edge would fuse with them into multiply-adds the interpreter, where the test's branch sits in between, never forms)
          o("  asm(\"\" : \"+v\"(n%d_0), \"+v\"(n%d_1), \"+v\"(n%d_2), \"+v\"(n%d_3));\n", par, par, par, par);
        }
        if (!(op.x & OPF_NOPERSIST_NUC) && !lazy) {
          o("  {\n    const size_t b = (size_t)%d * 4 * SP + s;\n", par);
          o("    a.partials[b] = n%d_0; a.partials[b + SP] = n%d_1; a.partials[b + 2 * SP] = n%d_2; a.partials[b + 3 * SP] = n%d_3;\n", par, par, par, par);
          o("    a.counts[(size_t)%d * SP + s] = c%d;\n  }\n", par, par);
        }
        have[par] = 1;
        last_node = par;
        closed = true;
      }
    }
    if (!closed) {
      if (!f[0].empty()) return std::string();  // entries of a parent without a closing one
      break;
    }
  }
  if (last_node < 0) return std::string();
  // root: L_s = sum_k root[s][k] pi[k]; this workgroup's share of sum_s f_s log L_s and of the integer scaler sum — the
  // interpreter's epilogue (prune.hip: prune_nuc2_kernel)
  o("  double Lk = n%d_0 * a.pi[0];\n  Lk = fma(n%d_1, a.pi[1], Lk);\n  Lk = fma(n%d_2, a.pi[2], Lk);\n  Lk = fma(n%d_3, a.pi[3], Lk);\n", last_node, last_node,
    last_node, last_node);
  o("  a.site_lik[s] = Lk;\n  a.site_cnt[s] = c%d;\n", last_node);
  o("  double term = 0.;\n  long long tc = 0;\n  int fl = 0;\n  const double fr = a.freq[s];\n");
  o("  if (fr != 0.) {\n    if (Lk != Lk || isinf(Lk)) fl |= 2;\n    else if (Lk <= 0.) fl |= 1;\n");
  o("    else { term += log(Lk) * fr; tc += (long long)c%d * (long long)fr; }\n  }\n", last_node);
  o("  for (int off = 32; off > 0; off >>= 1) { term += __shfl_xor(term, off); tc += __shfl_xor(tc, off); fl |= __shfl_xor(fl, off); }\n");
  o("  __shared__ double rs[4];\n  __shared__ long long rc[4];\n  __shared__ int rf[4];\n");
  o("  if ((tid & 63) == 0) { rs[tid >> 6] = term; rc[tid >> 6] = tc; rf[tid >> 6] = fl; }\n  __syncthreads();\n");
  if (small) {  // fused final combine: the protocol of the interpreter's LP form (prune.hip) and of the codon kernels' publish_partial
    o("  if (a.red_out) {\n    if (tid < 64) {\n      if (tid == 0) {\n");
    o("        __hip_atomic_store(a.wg_sum + blockIdx.x, (rs[0] + rs[1]) + (rs[2] + rs[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\n");
    o("        __hip_atomic_store(a.wg_cnt + blockIdx.x, (rc[0] + rc[1]) + (rc[2] + rc[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\n");
    o("        __hip_atomic_store(a.wg_flag + blockIdx.x, rf[0] | rf[1] | rf[2] | rf[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\n      }\n");
    o("      asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n      int old = 0;\n");
    o("      if (tid == 0) old = __hip_atomic_fetch_add(a.red_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\n");
    o("      old = __builtin_amdgcn_readfirstlane(old);\n      asm volatile(\"\" ::: \"memory\");\n");
    o("      if (old + 1 == (int)gridDim.x) {\n        if (tid == 0) __hip_atomic_store(a.red_done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\n");
    o("        hyhip::combine_partials(a.wg_sum, a.wg_cnt, a.wg_flag, (int)gridDim.x, a.red_out, a.red_rec, a.red_status, a.red_seq, tid);\n      }\n    }\n    return;\n  }\n");
  }
  o("  if (tid == 0) {\n    a.wg_sum[blockIdx.x] = (rs[0] + rs[1]) + (rs[2] + rs[3]);\n    a.wg_cnt[blockIdx.x] = (rc[0] + rc[1]) + (rc[2] + rc[3]);\n");
  o("    a.wg_flag[blockIdx.x] = rf[0] | rf[1] | rf[2] | rf[3];\n  }\n}\n");
  return o.s;
}

namespace {

uint64_t fnv1a(const std::string &s) {
  uint64_t h = 1469598103934665603ull;
  for (unsigned char c : s) h = (h ^ c) * 1099511628211ull;
  return h ? h : 1;
}

struct GenEntry {
  int state = 0;  // 0 compiling, 1 code object ready, -1 failed
  std::vector<char> code;
  std::string error;
  std::map<int, std::pair<hipModule_t, hipFunction_t>> loaded;  // per device
};

std::mutex g_mu;
std::condition_variable g_cv;
std::unordered_map<uint64_t, std::shared_ptr<GenEntry>> g_cache;

void compile_entry(std::shared_ptr<GenEntry> en, std::string src, uint64_t key) {
  Rtc &r = rtc();
  std::vector<char> code;
  std::string err;
  if (!r.ok) {
    err = "libhiprtc.so not available";
  } else {
    hiprtcProgram prog = nullptr;
    if (r.create(&prog, src.c_str(), "nucgen.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
      err = "hiprtcCreateProgram failed";
    } else {
      const char *opts[] = {"--offload-arch=gfx950", "-O3", "-ffp-contract=fast"};  // (the flags the interpreter is built with: same contractions)
      const hiprtcResult rc = r.compile(prog, 3, opts);
      if (rc != HIPRTC_SUCCESS) {
        size_t n = 0;
        r.log_size(prog, &n);
        std::string lg(n, '\0');
        if (n) r.log(prog, &lg[0]);
        err = "hiprtcCompileProgram failed: " + lg.substr(0, 2000);
      } else {
        size_t n = 0;
        if (r.code_size(prog, &n) == HIPRTC_SUCCESS && n > 0) {
          code.resize(n);
          if (r.code(prog, code.data()) != HIPRTC_SUCCESS) {
            code.clear();
            err = "hiprtcGetCode failed";
          }
        } else {
          err = "hiprtcGetCodeSize failed";
        }
      }
      r.destroy(&prog);
    }
  }
  if (const char *dump = getenv("HYPHY_HIP_NUCGEN_DUMP")) {  // diagnostic: the generated source (and, beside it, the code object)
    char path[512];
    snprintf(path, sizeof path, "%s/nucgen_%016llx.hip", dump, (unsigned long long)key);
    if (FILE *f = fopen(path, "w")) {
      fwrite(src.data(), 1, src.size(), f);
      fclose(f);
    }
    if (!code.empty()) {
      snprintf(path, sizeof path, "%s/nucgen_%016llx.co", dump, (unsigned long long)key);
      if (FILE *f = fopen(path, "wb")) {
        fwrite(code.data(), 1, code.size(), f);
        fclose(f);
      }
    }
  }
  {
    std::lock_guard<std::mutex> lk(g_mu);
    en->code.swap(code);
    en->error = err;
    en->state = en->code.empty() ? -1 : 1;
  }
  g_cv.notify_all();
  if (verbose()) {
    if (en->state == 1) fprintf(stderr, "[hyphy_hip] nucgen: kernel %016llx compiled (%zu bytes of source, %zu of code object)\n", (unsigned long long)key, src.size(), en->code.size());
    else fprintf(stderr, "[hyphy_hip] nucgen: kernel %016llx NOT available (%s): the interpreter stays\n", (unsigned long long)key, err.c_str());
  }
}

}  // namespace

int nucgen_mode() {  // 0 off, 1 background compilation (default), 2 synchronous
  const char *e = getenv("HYPHY_HIP_NUCGEN");
  return e ? atoi(e) : 1;
}
int nucgen_after() {
  const char *e = getenv("HYPHY_HIP_NUCGEN_AFTER");
  return e ? std::max(1, atoi(e)) : 8;
}

uint64_t nucgen_key(const int4 *ops, int n_ops, int L, bool lazy, bool small, int n_branches) {
  // (the source is a function of the entries and L alone: hash those, generate only when a kernel is actually requested)
  uint64_t h = 1469598103934665603ull;
  auto mix = [&](uint32_t v) {
    for (int b = 0; b < 4; b++) h = (h ^ ((v >> (8 * b)) & 0xff)) * 1099511628211ull;
  };
  mix((uint32_t)L);
  mix((uint32_t)n_ops);
  mix((lazy ? 1u : 0u) | (small ? 2u : 0u));
  mix(small ? (uint32_t)n_branches : 0u);
  for (int k = 0; k < n_ops; k++) mix((uint32_t)ops[k].x), mix((uint32_t)ops[k].y), mix((uint32_t)ops[k].z), mix((uint32_t)ops[k].w);
  return h ? h : 1;
}

// Ask for the kernel of a program (no-op when it is cached, being compiled, or known to have failed).
void nucgen_request(uint64_t key, const int4 *ops, int n_ops, int L, bool lazy, bool small, int n_branches, bool sync) {
  std::shared_ptr<GenEntry> en;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_cache.count(key)) {
      en = g_cache[key];
      if (!sync || en->state != 0) return;
    }
  }
  if (!en) {
    std::string src = nucgen_source(ops, n_ops, L, lazy, small, n_branches);
    en = std::make_shared<GenEntry>();
    {
      std::lock_guard<std::mutex> lk(g_mu);
      if (g_cache.count(key)) return;
      g_cache[key] = en;
      if (src.empty()) {
        en->state = -1;
        en->error = "schedule form not covered by the generator";
        return;
      }
    }
    if (sync) {
      compile_entry(en, src, key);
      return;
    }
    std::thread(compile_entry, en, std::move(src), key).detach();
    return;
  }
  std::unique_lock<std::mutex> lk(g_mu);  // (sync request for a kernel a background thread is compiling: wait for it)
  g_cv.wait(lk, [&] { return en->state != 0; });
}

// The loaded kernel for `key` on the current device, or nullptr (not requested / still compiling / failed).
static hipFunction_t nucgen_function(uint64_t key) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_cache.find(key);
  if (it == g_cache.end() || it->second->state != 1) return nullptr;
  GenEntry &en = *it->second;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  auto f = en.loaded.find(dev);
  if (f != en.loaded.end()) return f->second.second;
  hipModule_t mod = nullptr;
  hipFunction_t fn = nullptr;
  if (hipModuleLoadData(&mod, en.code.data()) != hipSuccess || hipModuleGetFunction(&fn, mod, "nucgen_kernel") != hipSuccess) {
    (void)hipGetLastError();
    en.state = -1;
    en.error = "hipModuleLoadData failed";
    if (verbose()) fprintf(stderr, "[hyphy_hip] nucgen: kernel %016llx could not be loaded: the interpreter stays\n", (unsigned long long)key);
    return nullptr;
  }
  en.loaded[dev] = std::make_pair(mod, fn);
  return fn;
}

bool nucgen_ready(uint64_t key) { return key != 0 && nucgen_function(key) != nullptr; }

// Launch the generated kernel in place of prune_nuc2_kernel (same grid: S_pad / 256 workgroups of 256 patterns, same outputs).
// small: the kernel was generated in its small-shard form (every matrix in LDS; takes the evaluation's exponentials along when
// ex != nullptr and the fused final combine when a.red_out is set).
bool nucgen_launch(uint64_t key, const NucArgs &a, hipStream_t stream, bool small, int n_branches, const ExpmArgs *ex) {
  hipFunction_t fn = nucgen_function(key);
  if (!fn || a.S_pad % 256 != 0 || !a.PT) return false;
  if (!small && ((ex && ex->n > 0) || a.red_out)) return false;
  NucGenArgs g;
  g.ambig = a.ambig;
  g.partials = a.partials;
  g.counts = a.counts;
  g.pi = a.pi;
  g.site_lik = a.site_lik;
  g.site_cnt = a.site_cnt;
  g.freq = a.freq;
  g.wg_sum = a.wg_sum;
  g.wg_cnt = a.wg_cnt;
  g.wg_flag = a.wg_flag;
  g.S_pad = a.S_pad;
  g.red_out = a.red_out;
  g.red_rec = a.red_rec;
  g.red_status = a.red_status;
  g.red_seq = a.red_seq;
  g.red_done = a.red_done;
  const double *PT = a.PT;
  const int16_t *codes = a.codes;
  NucGenExpm ge = {};
  CoefInline ci;  // (contents only matter to a folded launch, which fills it below)
  if (small && ex && ex->n > 0) {
    ExpmArgs exb = *ex;
    fill_coef_inline(exb, ci);
    ge.Q = exb.Q, ge.slots = exb.slots, ge.n = exb.n, ge.is_prob = exb.is_prob, ge.Prow = exb.Prow, ge.PTrow = exb.PTrow, ge.status = exb.status;
    ge.templates = exb.templates, ge.coeffs = exb.coeffs, ge.K = exb.K, ge.coef_inline = exb.coef_inline;
  }
  void *params[] = {(void *)&PT, (void *)&codes, (void *)&g, (void *)&ge, (void *)&ci};
  const size_t lds = (size_t)(small ? n_branches : a.L) * 16 * sizeof(double);
  const hipError_t rc = hipModuleLaunchKernel(fn, (unsigned)(a.S_pad / 256), 1, 1, 256, 1, 1, (unsigned)lds, stream, params, nullptr);
  if (rc != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return true;
}

}  // namespace hyhip

using namespace hyhip;

extern "C" {

/* Host-only (no device needed; needs libhiprtc): the source the library would compile for a 4-state partition's FULL pass in
 * steady state (lazy persistence) over the given tree, and whether it compiles for gfx950.  Returns the number of bytes of source
 * written to `src_out` (at most `cap`, NUL-terminated), 0 when the generator does not cover the tree, -1 on bad arguments;
 * small != 0: the small-shard form (matrices in LDS, exponentials and final combine inside the launch);
 * *compiled_out: 1 compiled, 0 compilation failed / hiprtc missing. */
int64_t hyphy_hip_plan_nucgen(int64_t L, int64_t I, const int64_t *flat_parents, const int64_t *leaf_has_ambig, int64_t small, char *src_out,
                              int64_t cap, int64_t *compiled_out) {
  if (L < 2 || I < 1 || !flat_parents) return -1;
  hyphy_hip_partition tmp;
  tmp.D = 4; tmp.L = L; tmp.I = I; tmp.C = 1; tmp.B = L + I - 1; tmp.NW = 1; tmp.DP = 16;
  tmp.nuc = true;
  tmp.nuc_leaf_pairs = true;
  tmp.parents.assign(flat_parents, flat_parents + L + I);
  tmp.children.assign(I, std::vector<int>());
  for (int64_t n = 0; n < L + I - 1; n++) {
    const int64_t par = flat_parents[n];
    if (par < 0 || par >= I || (n >= L && par <= n - L)) return -1;
    tmp.children[par].push_back((int)n);
  }
  tmp.leaf_has_ambig.assign(L, 0);
  if (leaf_has_ambig)
    for (int64_t l = 0; l < L; l++) tmp.leaf_has_ambig[l] = leaf_has_ambig[l] ? 1 : 0;
  init_plain_view(&tmp);
  tmp.n_slots = 2 + kNucParkSlots;
  tmp.sched_persist = false;  // a steady-state (lazy) full pass
  build_schedule(&tmp, nullptr, 0, true);
  if (tmp.programs.size() != 1) return 0;
  const int4 *ops = tmp.ops_host.data() + tmp.programs[0].off;
  const int n_ops = tmp.programs[0].n;
  const std::string src = nucgen_source(ops, n_ops, (int)L, true, small != 0, (int)(L + I - 1));
  if (src.empty()) return 0;
  if (src_out && cap > 0) {
    const size_t n = std::min<size_t>(src.size(), (size_t)cap - 1);
    memcpy(src_out, src.data(), n);
    src_out[n] = '\0';
  }
  if (compiled_out) {
    const uint64_t key = nucgen_key(ops, n_ops, (int)L, true, small != 0, (int)(L + I - 1));
    nucgen_request(key, ops, n_ops, (int)L, true, small != 0, (int)(L + I - 1), true);
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_cache.find(key);
    *compiled_out = (it != g_cache.end() && it->second->state == 1) ? 1 : 0;
  }
  return (int64_t)src.size();
}
}
