// Felsenstein pruning on gfx950 — device counterpart of _TheTree::ComputeTreeBlockByBranch
// (src/core/tree_evaluator.cpp:3556-4171) and of the combine step of ComputeBlock
// (src/core/likefunc.cpp:11046-11123).
//
// Design (MI355X-first, not a translation of the per-site CPU loop):
//  * A workgroup owns T tiles of 16 site patterns and walks the WHOLE post-order schedule for
//    them, so conditionals flow child -> parent through registers/LDS; HBM sees each finished
//    node once (persist, for later partial updates) instead of a write + read per tree level.
//  * Per child edge the product  [DP x DP] x [DP x 16 sites]  runs on the FP64 matrix cores
//    (v_mfma_f64_16x16x4_f64).  Wave w of the workgroup owns parent-state rows 16w..16w+15; the
//    MFMA C/D register image equals the B-operand image (common.h), so a node's result feeds its
//    parent's product with no shuffle.  The four row blocks are exchanged through LDS once per
//    node, together with the per-site sums that drive the 2^64 underflow rescaling.
//  * Leaf edges are a column gather from a transposed image of P (K4 in SURVEY §2.1); leaves
//    with ambiguity codes take the MFMA path with their resolution vector as the B operand.
//  * Rescaling is stateless per evaluation: integer exponents per (node, pattern) are carried up
//    the tree; observable contract of SURVEY A.5 (l_s, c_s with L_s = l_s 2^(-64 c_s)).
#include "common.h"

namespace hyhip {

namespace {

__device__ __forceinline__ f64x4 mfma(double a, double b, f64x4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory
// counter (vmcnt(0)), i.e. it would stall every node on the operand prefetch issued for the next
// schedule entry and on the fire-and-forget persist stores; nothing exchanged between the waves of
// a workgroup inside this kernel goes through global memory (except OP_GSYNC entries).
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// Decide the power-of-2^64 rescale for a site whose conditional vector sums to `tot`
// (__ll_loop_handle_scaling tree_evaluator.cpp:410-525, _computeBoostScaler /
// _computeReductionScaler tree.cpp:160-202).  Returns the exponent change m (true value =
// stored * 2^(-64 m)) and the multiplier in `sc`.
__device__ __forceinline__ int rescale_decision(double tot, double &sc) {
  int m = 0;
  sc = 1.0;
  if (tot < kScalerThreshold && tot > 0.0) {
    do {
      tot *= kScalerUp;
      sc *= kScalerUp;
      m++;
    } while (tot < kScalerThreshold && m < 15);
  } else if (tot > kScalerUp && tot < HUGE_VAL) {
    do {
      tot *= kScalerThreshold;
      sc *= kScalerThreshold;
      m--;
    } while (tot > kScalerUp && m > -15);
  }
  return m;
}

// Operand bundle fetched one schedule entry ahead: 16 doubles per lane, either the A-operand image
// of the next internal edge's transition matrix or the gathered columns of the next leaf group.
struct Payload {
  f64x2 v[8];
};

// CLDS: leaf codes of the workgroup's tiles and the schedule are staged in LDS (the common case);
// the !CLDS variant (thousands of taxa) reads both from global memory.
template <int NW, int T, bool CLDS>
__global__ __launch_bounds__(64 * NW, (T == 1 ? 3 : 1)) void prune_mfma_kernel(PruneArgs a) {
  constexpr int NKK = 4 * NW, DP = 16 * NW, TILE = NKK * 64;
  constexpr int G = (T == 1) ? 4 : (T == 2 ? 2 : 1);  // leaves per leaf-group entry (T*G*4 doubles <= 16)
  static_assert(NW == 4 || NKK <= 16, "payload sized for DP <= 64");
  // LDS: NS slots of T tiles (exchange buffers that double as a cache for finished nodes whose parent
  // is not the next schedule entry — host-allocated, see build_schedule), per-site sums, slot exponents,
  // then (dynamic) the leaf codes of this workgroup's tiles.
  constexpr int NS = lds_slots(T);
  __shared__ __align__(16) double xbuf[NS * T * TILE + T * NW * 16];
  __shared__ int slot_cnt[NS * T * 16];
  extern __shared__ __align__(16) int4 dyn_lds[];  // CLDS: [n_ops] schedule, then [L][T*16] int16 leaf codes
  int4 *ops_lds = dyn_lds;
  int16_t *codes_lds = reinterpret_cast<int16_t *>(dyn_lds + a.n_ops);
  double *sums = xbuf + NS * T * TILE;

  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, g = lane >> 4, sl = lane & 15;
  const int tile0 = blockIdx.x * T;
  const int S_pad = a.S_pad;

  if (CLDS) {
    for (int i = threadIdx.x; i < a.n_ops; i += 64 * NW) ops_lds[i] = a.ops[i];
    const int n = a.L * T * 16;
    for (int i = threadIdx.x; i < n; i += 64 * NW) {
      const int leaf = i / (T * 16), off = i - leaf * (T * 16);
      codes_lds[i] = a.codes[(size_t)leaf * S_pad + tile0 * 16 + off];
    }
    __syncthreads();
  }
  auto leaf_code = [&](int leaf, int t) -> int {
    if (CLDS) return (int)codes_lds[leaf * (T * 16) + t * 16 + sl];
    return (int)a.codes[(size_t)leaf * S_pad + (tile0 + t) * 16 + sl];
  };
  // schedule entry -> SGPRs (wave-uniform control flow, no vector-memory wait in the way)
  auto load_op = [&](int i) -> int4 {
    const int4 v = CLDS ? ops_lds[i] : a.ops[i];
    int4 r;
    r.x = __builtin_amdgcn_readfirstlane(v.x);
    r.y = __builtin_amdgcn_readfirstlane(v.y);
    r.z = __builtin_amdgcn_readfirstlane(v.z);
    r.w = __builtin_amdgcn_readfirstlane(v.w);
    return r;
  };
  auto leaf_of = [](const int4 &op, int i) -> int {
    const unsigned packed = (i < 2) ? (unsigned)op.z : (unsigned)op.w;
    return (int)((packed >> ((i & 1) * 16)) & 0xffffu);
  };
  // issue the global loads for a schedule entry (they complete while the previous entry computes)
  auto prefetch = [&](const int4 &op, Payload &pay) {
    const int flags = op.x & 0xff;
    if (flags & OP_LEAF) {
      const int nl = (op.x >> 8) & 0xff;
#pragma unroll
      for (int i = 0; i < G; i++) {
        if (i < nl) {
          const int leaf = leaf_of(op, i);
#pragma unroll
          for (int t = 0; t < T; t++) {
            int code = leaf_code(leaf, t);
            code = code < 0 ? 0 : code;  // ambiguous sites: value unused (MFMA path below)
            const f64x2 *src =
                reinterpret_cast<const f64x2 *>(a.PTg + (((size_t)leaf * DP + code) * NW + w) * 16 + g * 4);
            pay.v[(i * T + t) * 2] = src[0];
            pay.v[(i * T + t) * 2 + 1] = src[1];
          }
        }
      }
    } else {
      const double *Af = a.Pfrag + ((size_t)op.z * NW + w) * TILE;
#pragma unroll
      for (int k2 = 0; k2 < NKK / 2; k2++) pay.v[k2 % 8] = *reinterpret_cast<const f64x2 *>(Af + (k2 * 64 + lane) * 2);
    }
  };

  double slot_keep_scale[T];
  double B[T][NKK];   // child conditionals, B-operand image (also: the node finalised last)
  f64x4 acc[T];       // this wave's 16 parent states x 16 sites running product
  int cnt[T], bcnt[T];
#pragma unroll
  for (int t = 0; t < T; t++) {
    acc[t] = (f64x4){1., 1., 1., 1.};
    cnt[t] = 0;
    bcnt[t] = 0;
#pragma unroll
    for (int kk = 0; kk < NKK; kk++) B[t][kk] = 0.;
  }

  // schedule entries are fetched two ahead so the scalar load never sits in the prefetch address chain
  const int last_op = a.n_ops - 1;
  int4 nxt = load_op(0);
  int4 nxt2 = load_op(last_op < 1 ? last_op : 1);
  Payload pnext;
  prefetch(nxt, pnext);

  for (int oi = 0; oi < a.n_ops; oi++) {
    const int4 op = nxt;
    const Payload pay = pnext;
    nxt = nxt2;
    nxt2 = load_op(oi + 2 < last_op ? oi + 2 : last_op);
    if (oi + 1 < a.n_ops) prefetch(nxt, pnext);
    const int flags = op.x & 0xff, parent = op.y;
    const int dst_slot = (op.x >> 16) & 0xff, src_slot = (op.x >> 24) & 0xff;
    if (flags & OP_FIRST) {
#pragma unroll
      for (int t = 0; t < T; t++) {
        acc[t] = (f64x4){1., 1., 1., 1.};
        cnt[t] = 0;
      }
    }

    if (flags & OP_LEAF) {
      const int nl = (op.x >> 8) & 0xff;
#pragma unroll
      for (int i = 0; i < G; i++) {
        if (i < nl) {
          const int leaf = leaf_of(op, i);
          int code[T];
          bool amb = false;
#pragma unroll
          for (int t = 0; t < T; t++) {
            code[t] = leaf_code(leaf, t);
            amb |= code[t] < 0;
          }
          if (!__any(amb)) {
            // K4: parent[k] *= P[k][state] — columns were gathered one entry ahead
#pragma unroll
            for (int t = 0; t < T; t++) {
              const f64x2 lo = pay.v[(i * T + t) * 2], hi = pay.v[(i * T + t) * 2 + 1];
              acc[t] *= (f64x4){lo[0], lo[1], hi[0], hi[1]};
            }
          } else {
            // ambiguity codes in this tile: full product with the resolution vector as B operand
            // (rare path: operands are streamed, not staged in registers, to keep the hot path lean)
            const double *Af = a.Pfrag + ((size_t)leaf * NW + w) * TILE;
#pragma unroll 1
            for (int t = 0; t < T; t++) {
              f64x4 d = (f64x4){0., 0., 0., 0.};
              const int c = code[t];
              const double *av = a.ambig + (size_t)(c < 0 ? -c - 1 : 0) * DP;
#pragma unroll 2
              for (int kk = 0; kk < NKK; kk++) {
                const double bv = (c >= 0) ? ((4 * kk + g == c) ? 1.0 : 0.0) : av[4 * kk + g];
                d = mfma(Af[frag_index(kk, lane)], bv, d);
              }
              acc[t] *= d;
            }
          }
        }
      }
    } else {
      const int cinode = op.w;  // internal index of the child
      if (!(flags & OP_INREGS)) {
        if (flags & OP_GSYNC) __syncthreads();
        if (src_slot != 0xff) {  // still cached in LDS
#pragma unroll
          for (int t = 0; t < T; t++) {
            const double *src = xbuf + (src_slot * T + t) * TILE;
#pragma unroll
            for (int k2 = 0; k2 < NKK / 2; k2++) {
              const f64x2 v = *reinterpret_cast<const f64x2 *>(src + (k2 * 64 + lane) * 2);
              B[t][2 * k2] = v[0];
              B[t][2 * k2 + 1] = v[1];
            }
            bcnt[t] = slot_cnt[(src_slot * T + t) * 16 + sl];
          }
        } else {  // persisted copy (node not recomputed in this call, or the LDS slots ran out)
#pragma unroll
          for (int t = 0; t < T; t++) {
            const double *src = a.partials + ((size_t)cinode * a.ntiles + tile0 + t) * TILE;
#pragma unroll
            for (int k2 = 0; k2 < NKK / 2; k2++) {
              const f64x2 v = *reinterpret_cast<const f64x2 *>(src + (k2 * 64 + lane) * 2);
              B[t][2 * k2] = v[0];
              B[t][2 * k2 + 1] = v[1];
            }
            bcnt[t] = a.counts[(size_t)cinode * S_pad + (tile0 + t) * 16 + sl];
          }
        }
      }
      // two accumulator chains per tile: the f64 MFMA's dependent-issue latency (~200 cycles) exceeds
      // its independent issue interval (~143), tools/ubench_mfma_f64
      f64x4 d0[T], d1[T];
#pragma unroll
      for (int t = 0; t < T; t++) d0[t] = d1[t] = (f64x4){0., 0., 0., 0.};
#pragma unroll
      for (int kk = 0; kk < NKK; kk += 2)
#pragma unroll
        for (int t = 0; t < T; t++) {
          d0[t] = mfma(pay.v[(kk >> 1) % 8][0], B[t][kk], d0[t]);
          d1[t] = mfma(pay.v[(kk >> 1) % 8][1], B[t][kk + 1], d1[t]);
        }
#pragma unroll
      for (int t = 0; t < T; t++) {
        acc[t] *= (d0[t] + d1[t]);
        cnt[t] += bcnt[t];
      }
    }

    if (flags & OP_LAST) {
      // exchange the row blocks + per-site sums through LDS
#pragma unroll
      for (int t = 0; t < T; t++) {
        double s = (acc[t][0] + acc[t][1]) + (acc[t][2] + acc[t][3]);
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (g == 0) sums[(t * NW + w) * 16 + sl] = s;
        double *dst = xbuf + (dst_slot * T + t) * TILE;
        // kk = 4w + r  ->  frag_index(kk, lane): two 16-byte stores
        *reinterpret_cast<f64x2 *>(dst + ((2 * w) * 64 + lane) * 2) = (f64x2){acc[t][0], acc[t][1]};
        *reinterpret_cast<f64x2 *>(dst + ((2 * w + 1) * 64 + lane) * 2) = (f64x2){acc[t][2], acc[t][3]};
      }
      lds_barrier();
#pragma unroll
      for (int t = 0; t < T; t++) {
        const double *src = xbuf + (dst_slot * T + t) * TILE;
#pragma unroll
        for (int k2 = 0; k2 < NKK / 2; k2++) {
          const f64x2 v = *reinterpret_cast<const f64x2 *>(src + (k2 * 64 + lane) * 2);
          B[t][2 * k2] = v[0];
          B[t][2 * k2 + 1] = v[1];
        }
        double tot = sums[(t * NW) * 16 + sl];
#pragma unroll
        for (int ww = 1; ww < NW; ww++) tot += sums[(t * NW + ww) * 16 + sl];
        double sc;
        const int m = rescale_decision(tot, sc);
        if (m != 0) {
#pragma unroll
          for (int kk = 0; kk < NKK; kk++) B[t][kk] *= sc;
        }
        cnt[t] += m;
        bcnt[t] = cnt[t];
        // persist this wave's quarter (k-steps 4w .. 4w+3 == its own accumulator, times the exact
        // power-of-two scale) and the exponent
        double *out = a.partials + ((size_t)parent * a.ntiles + tile0 + t) * TILE;
        const f64x4 q = acc[t] * sc;
        *reinterpret_cast<f64x2 *>(out + ((2 * w) * 64 + lane) * 2) = (f64x2){q[0], q[1]};
        *reinterpret_cast<f64x2 *>(out + ((2 * w + 1) * 64 + lane) * 2) = (f64x2){q[2], q[3]};
        if (w == 0 && g == 0) a.counts[(size_t)parent * S_pad + (tile0 + t) * 16 + sl] = cnt[t];
        if (flags & OP_KEEP) {  // the slot will be read again later: it must hold the rescaled vector
          slot_keep_scale[t] = sc;
          if (w == 0 && g == 0) slot_cnt[(dst_slot * T + t) * 16 + sl] = cnt[t];
        }
      }
      lds_barrier();  // every wave has read the slot and the sums
      if (flags & OP_KEEP) {
#pragma unroll
        for (int t = 0; t < T; t++) {
          if (__any(slot_keep_scale[t] != 1.0)) {
            double *dst = xbuf + (dst_slot * T + t) * TILE;
            const f64x4 q = acc[t] * slot_keep_scale[t];
            *reinterpret_cast<f64x2 *>(dst + ((2 * w) * 64 + lane) * 2) = (f64x2){q[0], q[1]};
            *reinterpret_cast<f64x2 *>(dst + ((2 * w + 1) * 64 + lane) * 2) = (f64x2){q[2], q[3]};
          }
        }
      }
    }
  }

  // root: L_s = sum_k root[s][k] pi[k]; this workgroup's share of sum_s f_s log L_s
  // (tree_evaluator.cpp:4046-4128) and of the integer scaler sum (likefunc.cpp:11123)
  if (a.n_ops > 0) {
    double pk[NKK];
#pragma unroll
    for (int kk = 0; kk < NKK; kk++) pk[kk] = a.pi[4 * kk + g];
    double wsum = 0.;
    long long wcnt = 0;
    int wflag = 0;
#pragma unroll
    for (int t = 0; t < T; t++) {
      double s = 0.;
#pragma unroll
      for (int kk = 0; kk < NKK; kk++) s = fma(B[t][kk], pk[kk], s);
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      if (w == 0 && g == 0) {
        const int site = (tile0 + t) * 16 + sl;
        a.site_lik[site] = s;
        a.site_cnt[site] = bcnt[t];
        const double f = a.freq[site];
        if (f != 0.) {
          if (s != s || isinf(s)) wflag |= 2;
          else if (s <= 0.) wflag |= 1;
          else {
            wsum += log(s) * f;
            wcnt += (long long)bcnt[t] * (long long)f;
          }
        }
      }
    }
    if (w == 0) {  // fixed-order butterfly over the 16 site lanes (lanes >= 16 hold zeros)
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) {
        wsum += __shfl_xor(wsum, off);
        wcnt += __shfl_xor(wcnt, off);
        wflag |= __shfl_xor(wflag, off);
      }
      if (lane == 0) {
        a.wg_sum[blockIdx.x] = wsum;
        a.wg_cnt[blockIdx.x] = wcnt;
        a.wg_flag[blockIdx.x] = wflag;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// 4-state (nucleotide) kernel: one thread per site pattern walks the whole schedule; P matrices
// are wave-uniform (scalar loads), conditionals live in registers and are persisted as
// state-major planes so every global access is a coalesced 512-byte line per wave.
// HBM-bound: per node 32 B/site written (+ re-read of children that are not in registers).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prune_nuc_kernel(NucArgs a) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;  // S_pad is a multiple of the block size
  const size_t S_pad = a.S_pad;
  double acc[4] = {1., 1., 1., 1.}, b[4] = {0., 0., 0., 0.};
  int cnt = 0, bcnt = 0;
  for (int oi = 0; oi < a.n_ops; oi++) {
    const int4 op = a.ops[oi];
    const int flags = op.x & 0xff, parent = op.y;
    const int child = (flags & OP_LEAF) ? (op.z & 0xffff) : op.z;  // nucleotide schedules use 1 leaf per entry
    if (flags & OP_FIRST) {
      acc[0] = acc[1] = acc[2] = acc[3] = 1.;
      cnt = 0;
    }
    const double *__restrict__ P = a.P + (size_t)child * 16;
    double cv[4];
    bool matvec = true;
    if (flags & OP_LEAF) {
      const int code = a.codes[(size_t)child * S_pad + s];
      if (code >= 0) {
        matvec = false;
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const double p0 = P[4 * i], p1 = P[4 * i + 1], p2 = P[4 * i + 2], p3 = P[4 * i + 3];
          acc[i] *= (code == 0) ? p0 : (code == 1) ? p1 : (code == 2) ? p2 : p3;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++) cv[j] = a.ambig[(size_t)(-code - 1) * 4 + j];
      }
    } else {
      if (!(flags & OP_INREGS)) {
        const size_t base = (size_t)op.w * 4 * S_pad + s;
#pragma unroll
        for (int j = 0; j < 4; j++) b[j] = a.partials[base + j * S_pad];
        bcnt = a.counts[(size_t)op.w * S_pad + s];
      }
#pragma unroll
      for (int j = 0; j < 4; j++) cv[j] = b[j];
      cnt += bcnt;
    }
    if (matvec) {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        double m = P[4 * i] * cv[0];
        m = fma(P[4 * i + 1], cv[1], m);
        m = fma(P[4 * i + 2], cv[2], m);
        m = fma(P[4 * i + 3], cv[3], m);
        acc[i] *= m;
      }
    }
    if (flags & OP_LAST) {
      const double tot = (acc[0] + acc[1]) + (acc[2] + acc[3]);
      double sc;
      const int m = rescale_decision(tot, sc);
#pragma unroll
      for (int j = 0; j < 4; j++) b[j] = (m != 0) ? acc[j] * sc : acc[j];
      cnt += m;
      bcnt = cnt;
      const size_t base = (size_t)parent * 4 * S_pad + s;
#pragma unroll
      for (int j = 0; j < 4; j++) a.partials[base + j * S_pad] = b[j];
      a.counts[(size_t)parent * S_pad + s] = cnt;
    }
  }
  __shared__ double rs[256];
  __shared__ long long rc[256];
  __shared__ int rf;
  if (threadIdx.x == 0) rf = 0;
  __syncthreads();
  double term = 0.;
  long long tc = 0;
  if (a.n_ops > 0 && s < a.S_pad) {
    double L = b[0] * a.pi[0];
    L = fma(b[1], a.pi[1], L);
    L = fma(b[2], a.pi[2], L);
    L = fma(b[3], a.pi[3], L);
    a.site_lik[s] = L;
    a.site_cnt[s] = bcnt;
    const double f = a.freq[s];
    if (f != 0.) {
      if (L != L || isinf(L)) atomicOr(&rf, 2);
      else if (L <= 0.) atomicOr(&rf, 1);
      else {
        term = log(L) * f;
        tc = (long long)bcnt * (long long)f;
      }
    }
  }
  rs[threadIdx.x] = term;
  rc[threadIdx.x] = tc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      rs[threadIdx.x] += rs[threadIdx.x + off];
      rc[threadIdx.x] += rc[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0 && a.n_ops > 0) {
    a.wg_sum[blockIdx.x] = rs[0];
    a.wg_cnt[blockIdx.x] = rc[0];
    a.wg_flag[blockIdx.x] = rf;
  }
}

// ---------------------------------------------------------------------------------------------
// logL = sum_s f_s log L_s  -  64 ln2 * sum_s f_s c_s      (tree_evaluator.cpp:4114-4128 Kahan sum,
// likefunc.cpp:11123 scaler correction).  One workgroup; per-thread Kahan accumulation over a
// fixed stride, then a fixed-order tree: deterministic run to run.  The scaler part is summed in
// exact integer arithmetic like the reference's `long overallScaler`.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void site_reduce_kernel(const double *__restrict__ site_lik,
                                                           const int32_t *__restrict__ site_cnt,
                                                           const double *__restrict__ freq, int S_pad,
                                                           int floor_log, double *__restrict__ out,
                                                           double *__restrict__ out_cnt) {
  __shared__ double ssum[1024];
  __shared__ double scomp[1024];
  __shared__ long long scnt[1024];
  __shared__ int sflags;
  const int tid = threadIdx.x;
  if (tid == 0) sflags = 0;
  __syncthreads();
  double sum = 0., comp = 0.;
  long long c = 0;
  int fl = 0;  // 1: a pattern with zero likelihood (-> -inf, tree_evaluator.cpp:4094-4112), 2: NaN
  for (int s = tid; s < S_pad; s += 1024) {
    const double f = freq[s];
    if (f == 0.) continue;
    const double L = site_lik[s];
    if (L != L) { fl |= 2; continue; }
    if (L <= 0. || isinf(L)) {
      if (floor_log && L <= 0.) {  // myLog floor, no scaler (likefunc.cpp:644-661) in category mode
        const double y0 = -1000000. * f - comp;
        const double t0 = sum + y0;
        comp = (t0 - sum) - y0;
        sum = t0;
      } else {
        fl |= (L <= 0.) ? 1 : 2;
      }
      continue;
    }
    const double y = log(L) * f - comp;  // Kahan
    const double t = sum + y;
    comp = (t - sum) - y;
    sum = t;
    c += (long long)site_cnt[s] * (long long)f;
  }
  if (fl) atomicOr(&sflags, fl);
  ssum[tid] = sum;
  scomp[tid] = comp;
  scnt[tid] = c;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if (tid < off) {
      const double a0 = ssum[tid], b0 = ssum[tid + off];
      const double t = a0 + b0;
      const double e = (fabs(a0) >= fabs(b0)) ? (a0 - t) + b0 : (b0 - t) + a0;  // a0 + b0 = t + e exactly
      ssum[tid] = t;
      scomp[tid] = scomp[tid] + scomp[tid + off] - e;
      scnt[tid] += scnt[tid + off];
    }
    __syncthreads();
  }
  if (tid == 0) {
    double r = (ssum[0] - scomp[0]) - kLogScaler * (double)scnt[0];
    if (sflags & 2) r = NAN;
    else if (sflags & 1) r = -INFINITY;
    out[0] = r;
    out_cnt[0] = (double)scnt[0];
  }
}

// Final combine of the per-workgroup partial sums: Neumaier-compensated, fixed order (the device
// analogue of ComputeBlock's combine of its thread blocks, likefunc.cpp:11046-11123).
__global__ __launch_bounds__(256) void wg_reduce_kernel(const double *__restrict__ wg_sum,
                                                        const long long *__restrict__ wg_cnt,
                                                        const int *__restrict__ wg_flag, int n,
                                                        double *__restrict__ out, double *__restrict__ out_cnt) {
  __shared__ double ssum[256];
  __shared__ double scomp[256];
  __shared__ long long scnt[256];
  __shared__ int sflags;
  const int tid = threadIdx.x;
  if (tid == 0) sflags = 0;
  __syncthreads();
  double sum = 0., comp = 0.;
  long long c = 0;
  int fl = 0;
  for (int k = tid; k < n; k += 256) {
    const double y = wg_sum[k] - comp;
    const double t = sum + y;
    comp = (t - sum) - y;
    sum = t;
    c += wg_cnt[k];
    fl |= wg_flag[k];
  }
  if (fl) atomicOr(&sflags, fl);
  ssum[tid] = sum;
  scomp[tid] = comp;
  scnt[tid] = c;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) {
      const double a0 = ssum[tid], b0 = ssum[tid + off];
      const double t = a0 + b0;
      const double e = (fabs(a0) >= fabs(b0)) ? (a0 - t) + b0 : (b0 - t) + a0;
      ssum[tid] = t;
      scomp[tid] = scomp[tid] + scomp[tid + off] - e;
      scnt[tid] += scnt[tid + off];
    }
    __syncthreads();
  }
  if (tid == 0) {
    double r = (ssum[0] - scomp[0]) - kLogScaler * (double)scnt[0];
    if (sflags & 2) r = NAN;
    else if (sflags & 1) r = -INFINITY;
    out[0] = r;
    out_cnt[0] = (double)scnt[0];
  }
}

// Category mixing on the device: PopulateConditionalProbabilities weighted-sum mode
// (likefunc2.cpp:820-853): buf[s] = sum_c w_c L_c[s] 2^(-64 (c_c[s] - min_c c_c[s])).
__global__ void mix_categories_kernel(const double *__restrict__ site_lik, const int32_t *__restrict__ site_cnt,
                                      const double *__restrict__ w, int C, int S_pad, double *__restrict__ mixed,
                                      int32_t *__restrict__ mixed_cnt) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S_pad) return;
  double buf = 0.;
  int sc = 0;
  for (int c = 0; c < C; c++) {
    const double v = site_lik[(size_t)c * S_pad + s] * w[c];
    const int scv = site_cnt[(size_t)c * S_pad + s];
    if (c == 0) {
      buf = v;
      sc = scv;
    } else if (scv < sc) {
      buf = v + buf * exp(-kLogScaler * (double)(sc - scv));
      sc = scv;
    } else if (scv > sc) {
      buf += v * exp(-kLogScaler * (double)(scv - sc));
    } else {
      buf += v;
    }
  }
  mixed[s] = buf;
  mixed_cnt[s] = sc;
}

// fragment layout -> reference iNodeCache layout [(node*S + pattern)*D + state]
__global__ void unpack_partials_kernel(const double *__restrict__ partials, int I, int ntiles, int NW, int D, int S,
                                       double *__restrict__ out) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)I * S * D;
  if (idx >= total) return;
  const int state = idx % D;
  const size_t rest = idx / D;
  const int pat = rest % S;
  const int node = rest / S;
  const int tile = pat >> 4, sl = pat & 15;
  const int kk = state >> 2, lane = (state & 3) * 16 + sl;
  const int TILE = NW * 4 * 64;
  out[idx] = partials[((size_t)node * ntiles + tile) * TILE + frag_index(kk, lane)];
}

template <int NW, bool CLDS>
void launch_prune_T(const PruneArgs &a, hipStream_t stream) {
  const dim3 grid(a.ntiles / a.T), block(64 * NW);
  const size_t lds = CLDS ? (size_t)a.n_ops * sizeof(int4) + (size_t)a.L * a.T * 16 * sizeof(int16_t) : 0;
  switch (a.T) {
    case 1:
      hipLaunchKernelGGL((prune_mfma_kernel<NW, 1, CLDS>), grid, block, lds, stream, a);
      break;
    case 2:
      hipLaunchKernelGGL((prune_mfma_kernel<NW, 2, CLDS>), grid, block, lds, stream, a);
      break;
    case 3:
      hipLaunchKernelGGL((prune_mfma_kernel<NW, 3, CLDS>), grid, block, lds, stream, a);
      break;
    default:
      hipLaunchKernelGGL((prune_mfma_kernel<NW, 4, CLDS>), grid, block, lds, stream, a);
      break;
  }
}

template <int NW>
void launch_prune_NW(const PruneArgs &a, hipStream_t stream) {
  if (a.codes_in_lds) launch_prune_T<NW, true>(a, stream);
  else launch_prune_T<NW, false>(a, stream);
}

}  // namespace

void launch_prune_mfma(const PruneArgs &a, hipStream_t stream) {
  if (a.n_ops <= 0) return;
  switch (a.NW) {
    case 1:
      launch_prune_NW<1>(a, stream);
      break;
    case 2:
      launch_prune_NW<2>(a, stream);
      break;
    case 3:
      launch_prune_NW<3>(a, stream);
      break;
    default:
      launch_prune_NW<4>(a, stream);
      break;
  }
}

void launch_prune_nuc(const NucArgs &a, hipStream_t stream) {
  if (a.n_ops <= 0) return;
  hipLaunchKernelGGL(prune_nuc_kernel, dim3((a.S_pad + 255) / 256), dim3(256), 0, stream, a);
}

void launch_site_reduce(const double *site_lik, const int32_t *site_cnt, const double *freq, int S_pad, int floor_log,
                        double *out, double *out_cnt, hipStream_t stream) {
  hipLaunchKernelGGL(site_reduce_kernel, dim3(1), dim3(1024), 0, stream, site_lik, site_cnt, freq, S_pad, floor_log,
                     out, out_cnt);
}

void launch_wg_reduce(const double *wg_sum, const long long *wg_cnt, const int *wg_flag, int n, double *out_logl,
                      double *out_cnt, hipStream_t stream) {
  hipLaunchKernelGGL(wg_reduce_kernel, dim3(1), dim3(256), 0, stream, wg_sum, wg_cnt, wg_flag, n, out_logl, out_cnt);
}

int prune_mfma_grid(const PruneArgs &a) { return a.ntiles / a.T; }
int prune_nuc_grid(const NucArgs &a) { return (a.S_pad + 255) / 256; }

void launch_mix_categories(const double *site_lik, const int32_t *site_cnt, const double *weights_dev, int C,
                           int S_pad, double *mixed_lik, int32_t *mixed_cnt, hipStream_t stream) {
  hipLaunchKernelGGL(mix_categories_kernel, dim3((S_pad + 255) / 256), dim3(256), 0, stream, site_lik, site_cnt,
                     weights_dev, C, S_pad, mixed_lik, mixed_cnt);
}

void launch_unpack_partials_mfma(const double *partials, int I, int ntiles, int NW, int D, int S, int64_t, int64_t,
                                 double *out, hipStream_t stream) {
  const size_t total = (size_t)I * S * D;
  hipLaunchKernelGGL(unpack_partials_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, partials, I,
                     ntiles, NW, D, S, out);
}

}  // namespace hyhip
