// Per-site batched fits on gfx950 — the device primitive behind FEL-style analyses
// (res/TemplateBatchFiles/SelectionAnalyses/FEL.bf:593-605, 609+): every alignment site s carries its OWN
// rate multipliers (alpha_s, beta_s, ...), so each (site, branch) pair has its own rate matrix
//       Q_{b,s} = sum_k  m[s][group(b)][k] * c[b][k] * T_k          (T_k: fixed D x D templates with their diagonals)
// and the reference evaluates one single-site likelihood function per site: sites x branches matrix exponentials
// (_Matrix::Exponentiate, matrix.cpp:5746+) followed by a one-pattern pruning pass (tree_evaluator.cpp:3556+).
//
// MI355X-first design — the transition matrices are never formed.  A pruning step only needs the ACTION of
// exp(Q_{b,s}) on one vector per site (the child's conditionals; a unit vector for an observed leaf state), and
// uniformisation turns that action into products with the SHARED templates:
//       exp(Q) v = sum_j  Pois(j; mu) R^j v,     R = I + Q / mu,   mu >= max_i |Q_ii|
//       R^j v    = t_j,   t_j = t_{j-1} + (sum_k T_k (x_k o t_{j-1})) / mu      (x_k = per-site coefficient)
// For a tile of 16 sites `T_k (x_k o t)` is a [DP x DP] x [DP x 16] product on the FP64 matrix cores with the
// template image as the A operand — the same MFMA shape, register images and L2-resident operand stream as the
// pruning kernel (prune.hip), only the B operand is scaled per site.  All terms are non-negative (R is a
// stochastic matrix): no cancellation, componentwise relative accuracy — better conditioned than Taylor +
// squaring of the full matrix, and ~D/(2 terms) times less arithmetic than sites x branches exponentials.
// One wave owns a 16-site tile of one parameter set and walks the whole post-order schedule; nodes flow child ->
// parent through registers / wave-private LDS parking slots exactly as in prune_wave_kernel.
#include "common.h"

namespace hyhip {

namespace {

__device__ __forceinline__ f64x4 mfma(double a, double b, f64x4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f64x2 ld16(const double *ubase, unsigned byte_off) {
  return *reinterpret_cast<const f64x2 *>(reinterpret_cast<const char *>(ubase) + byte_off);
}
__device__ __forceinline__ void st16(double *ubase, unsigned byte_off, f64x2 v) {
  *reinterpret_cast<f64x2 *>(reinterpret_cast<char *>(ubase) + byte_off) = v;
}
__device__ __forceinline__ double row_sum4(double x) {
  x += __shfl_xor(x, 16);
  x += __shfl_xor(x, 32);
  return x;
}
__device__ __forceinline__ double wave_max(double x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x = fmax(x, __shfl_xor(x, off));
  return x;
}
// (same contract as prune.hip: __ll_loop_handle_scaling tree_evaluator.cpp:410-525)
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

constexpr double kMuStep = 64.0;      // uniformisation rate handled by one Poisson series (e^-64 is a normal double)
constexpr double kTailEps = 1e-18;    // neglected Poisson mass (relative to total mass 1)

template <int NW, int NP, bool MIX>
__global__ __launch_bounds__(64, 2) void site_fit_kernel(const int4 *__restrict__ ops, const double *__restrict__ Timg,
                                                         const double *__restrict__ bcoef, const int *__restrict__ bgroup,
                                                         SiteFitArgs a) {
  constexpr int NKK = 4 * NW, DP = 16 * NW, TILE = NKK * 64;
  __shared__ __align__(16) double park[NP * TILE];
  __shared__ __align__(16) double accl[TILE];  // running product of the current parent (registers are needed for the series)
  // MIX (branch-site mixtures: P_b = sum_m w_m exp(Q_b^(m)) per site, MEME / BS-REL style): the edge's operand vector,
  // kept while the series runs once per mixture component
  __shared__ __align__(16) double vsave[MIX ? TILE : 2];
  __shared__ int park_cnt[NP][16];
  extern __shared__ __align__(16) int16_t codes_lds[];  // [L][16]

  const int lane = threadIdx.x, g = lane >> 4, sl = lane & 15;
  const int tile0 = blockIdx.x, set = blockIdx.y;
  const int K = a.K;
  {
    const int4 *src = reinterpret_cast<const int4 *>(a.codes_tile + (size_t)tile0 * a.L * 16);
    int4 *dst = reinterpret_cast<int4 *>(codes_lds);
    for (int i = lane; i < a.L * 2; i += 64) dst[i] = src[i];
    __syncthreads();
  }
  // this lane's site: multipliers [G][K] of the set
  const int n_mix = MIX ? a.n_mix : 1;
  const double *sm0 = a.smult + ((size_t)set * a.S_pad + (size_t)tile0 * 16 + sl) * (size_t)(n_mix * a.G * K);
  const double *wmix = MIX ? a.smix + ((size_t)set * a.S_pad + (size_t)tile0 * 16 + sl) * (size_t)n_mix : nullptr;

  const f64x4 ones = (f64x4){1., 1., 1., 1.}, zeros = (f64x4){0., 0., 0., 0.};
  f64x4 bch[NW];  // the node finalised last (scaled)
  int cnt = 0, bcnt = 0;
  bool first_edge = true;  // (uniform) the running product is still all ones
#pragma unroll
  for (int w = 0; w < NW; w++) bch[w] = zeros;
  auto acc_multiply = [&](const f64x4 *t) {
#pragma unroll
    for (int w = 0; w < NW; w++) {
      f64x2 *lo = reinterpret_cast<f64x2 *>(accl + ((2 * w) * 64 + lane) * 2), *hi = reinterpret_cast<f64x2 *>(accl + ((2 * w + 1) * 64 + lane) * 2);
      if (first_edge) {
        *lo = (f64x2){t[w][0], t[w][1]};
        *hi = (f64x2){t[w][2], t[w][3]};
      } else {
        const f64x2 a0 = *lo, a1 = *hi;
        *lo = (f64x2){a0[0] * t[w][0], a0[1] * t[w][1]};
        *hi = (f64x2){a1[0] * t[w][2], a1[1] * t[w][3]};
      }
    }
    first_edge = false;
  };

  // acc *= exp(Q_{branch, site}) v   for the 16 sites of the tile (v: [NW] C/D-image registers = B-operand image)
  f64x4 term[NW];  // the edge's operand vector on entry (filled by the schedule entry), the series' running term inside
  auto series = [&](int branch, const double *sm) {  // term <- exp(Q_{branch, site}) term
    const int grp = bgroup[branch];
    // per-site coefficients of the (<= 4) templates; named scalars: a runtime-indexed array would live in scratch
    const double x0 = sm[grp * K] * bcoef[branch * K];
    const double x1 = K > 1 ? sm[grp * K + 1] * bcoef[branch * K + 1] : 0.;
    const double x2 = K > 2 ? sm[grp * K + 2] * bcoef[branch * K + 2] : 0.;
    const double x3 = K > 3 ? sm[grp * K + 3] * bcoef[branch * K + 3] : 0.;
    const double mu = fma(x0, a.dmax[0], fma(x1, a.dmax[1], fma(x2, a.dmax[2], x3 * a.dmax[3])));
    const double mu_max = wave_max(mu);
    if (!(mu_max > 0.)) return;  // zero-length branch for every site of the tile: exp(Q) = I
    const int n_sub = (int)ceil(mu_max / kMuStep);
    const double mu_sub = mu / (double)n_sub, mu_sub_max = mu_max / (double)n_sub;
    const double inv_mu = mu > 0. ? 1.0 / mu : 0.;
    const double w0 = exp(-mu_sub);
    f64x4 sum[NW];
    for (int sub = 0; sub < n_sub; sub++) {
      double wgt = w0;
#pragma unroll
      for (int w = 0; w < NW; w++) sum[w] = term[w] * wgt;
      for (int j = 1;; j++) {
        // D = sum_k T_k (x_k o term): K passes over the template images accumulate into the same NW chains.
        // Two-stage software pipeline as in prune_wave_kernel::edge_product.
        f64x4 D[NW];
#pragma unroll
        for (int w = 0; w < NW; w++) D[w] = zeros;
        for (int k = 0; k < K; k++) {
          const double *pf = Timg + (size_t)k * NW * TILE;  // uniform
          const double xk = k == 0 ? x0 : (k == 1 ? x1 : (k == 2 ? x2 : x3));
          f64x2 Ac[NW], An[NW];
#pragma unroll
          for (int w = 0; w < NW; w++) Ac[w] = ld16(pf, (unsigned)((w * TILE + lane * 2) * 8));
#pragma unroll
          for (int k2 = 0; k2 < NKK / 2; k2++) {
            if (k2 + 1 < NKK / 2) {
#pragma unroll
              for (int w = 0; w < NW; w++) An[w] = ld16(pf, (unsigned)((w * TILE + ((k2 + 1) * 64 + lane) * 2) * 8));
            }
            const double b0 = term[k2 >> 1][(k2 & 1) * 2] * xk, b1 = term[k2 >> 1][(k2 & 1) * 2 + 1] * xk;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int w = 0; w < NW; w++) D[w] = mfma(Ac[w][0], b0, D[w]);
#pragma unroll
            for (int w = 0; w < NW; w++) D[w] = mfma(Ac[w][1], b1, D[w]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int w = 0; w < NW; w++) Ac[w] = An[w];
          }
        }
        wgt *= mu_sub / (double)j;
#pragma unroll
        for (int w = 0; w < NW; w++) {
          term[w] += D[w] * inv_mu;  // R t = t + Q t / mu
          sum[w] += term[w] * wgt;
        }
        // remaining Poisson mass <= wgt * r / (1 - r), r = mu_sub / (j + 1) (geometric bound past the mode);
        // uniform decision on the tile's largest rate (its weights dominate the others' from the mode on)
        const double r = mu_sub_max / (double)(j + 1);
        double wmax = wave_max(wgt);
        if (r < 0.5 && wmax * r / (1.0 - r) < kTailEps) break;
        if (j > 4096) break;  // (cannot happen: mu_sub_max <= kMuStep)
      }
#pragma unroll
      for (int w = 0; w < NW; w++) term[w] = sum[w];
    }
  };
  auto apply_edge = [&](int branch) {
    if (!MIX) {
      series(branch, sm0);
      acc_multiply(term);
      return;
    }
    // mixture: the node finalised last (bch) is dead from here to the next finalisation — the schedule reads it only
    // in the FIRST entry of the next parent, which has copied it into `term` already — so its registers accumulate
    // sum_m w_m exp(Q^(m)) v
#pragma unroll
    for (int w = 0; w < NW; w++) {
      *reinterpret_cast<f64x2 *>(vsave + ((2 * w) * 64 + lane) * 2) = (f64x2){term[w][0], term[w][1]};
      *reinterpret_cast<f64x2 *>(vsave + ((2 * w + 1) * 64 + lane) * 2) = (f64x2){term[w][2], term[w][3]};
      bch[w] = zeros;
    }
    for (int m = 0; m < n_mix; m++) {
      if (m > 0) {
#pragma unroll
        for (int w = 0; w < NW; w++) {
          const f64x2 lo = *reinterpret_cast<const f64x2 *>(vsave + ((2 * w) * 64 + lane) * 2);
          const f64x2 hi = *reinterpret_cast<const f64x2 *>(vsave + ((2 * w + 1) * 64 + lane) * 2);
          term[w] = (f64x4){lo[0], lo[1], hi[0], hi[1]};
        }
      }
      series(branch, sm0 + (size_t)m * a.G * K);
      const double wm = wmix[m];
#pragma unroll
      for (int w = 0; w < NW; w++) bch[w] += term[w] * wm;
    }
    acc_multiply(bch);
  };

  int4 op = ops[0];
  for (int oi = 0; oi < a.n_ops; oi++) {
    const int4 nxt = ops[oi + 1];
    const int kind = op.x & 3;
    if (kind == OPK_LEAF) {
      const int nl = (op.x >> 8) & 0x7f;
      for (int i = 0; i < nl; i++) {
        const int lf = (op.z >> (16 * i)) & 0xffff;
        const int c = (int)codes_lds[lf * 16 + sl];
#pragma unroll
        for (int w = 0; w < NW; w++)
#pragma unroll
          for (int r = 0; r < 4; r++) term[w][r] = (16 * w + 4 * r + g == c) ? 1.0 : 0.0;
        if ((op.x & OPF_AMBIG) && __any(c < 0)) {  // ambiguity codes in this tile: resolution vectors
          const double *av = a.ambig + (size_t)(c < 0 ? -c - 1 : 0) * DP;
#pragma unroll
          for (int w = 0; w < NW; w++)
#pragma unroll
            for (int r = 0; r < 4; r++)
              if (c < 0) term[w][r] = av[16 * w + 4 * r + g];
        }
        apply_edge(lf);
      }
    } else if (kind == OPK_INTERNAL) {
      const int slot = (op.x >> 24) & 0xff;
      if (slot < 2) {
#pragma unroll
        for (int w = 0; w < NW; w++) term[w] = bch[w];
        cnt += bcnt;
      } else {
        const double *src = park + (slot - 2) * TILE;
#pragma unroll
        for (int w = 0; w < NW; w++) {
          const f64x2 lo = *reinterpret_cast<const f64x2 *>(src + ((2 * w) * 64 + lane) * 2);
          const f64x2 hi = *reinterpret_cast<const f64x2 *>(src + ((2 * w + 1) * 64 + lane) * 2);
          term[w] = (f64x4){lo[0], lo[1], hi[0], hi[1]};
        }
        cnt += park_cnt[slot - 2][sl];
      }
      apply_edge(op.z);
    } else {
      // a node that found no parking slot: spilled to the scratch copy by this wave earlier in the pass
      const double *src = a.scratch + (((size_t)set * a.I + op.w) * a.ntiles + tile0) * TILE;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int w = 0; w < NW; w++) {
        const f64x2 lo = ld16(src, (unsigned)((2 * w) * 64 + lane) * 16u), hi = ld16(src, (unsigned)((2 * w + 1) * 64 + lane) * 16u);
        term[w] = (f64x4){lo[0], lo[1], hi[0], hi[1]};
      }
      cnt += a.scratch_cnt[((size_t)set * a.I + op.w) * a.S_pad + tile0 * 16 + sl];
      apply_edge(op.z);
    }

    if (op.x & OPF_LAST) {
      const int slot = (op.x >> 16) & 0xff;
      f64x4 acc[NW];
      double s = 0.;
#pragma unroll
      for (int w = 0; w < NW; w++) {
        const f64x2 lo = *reinterpret_cast<const f64x2 *>(accl + ((2 * w) * 64 + lane) * 2);
        const f64x2 hi = *reinterpret_cast<const f64x2 *>(accl + ((2 * w + 1) * 64 + lane) * 2);
        acc[w] = first_edge ? ones : (f64x4){lo[0], lo[1], hi[0], hi[1]};
        s += (acc[w][0] + acc[w][1]) + (acc[w][2] + acc[w][3]);
      }
      first_edge = true;
      const double tot = row_sum4(s);
      double sc = 1.0;
      int m = 0;
      if (__any(!(tot >= kScalerThreshold && tot <= kScalerUp))) m = rescale_decision(tot, sc);
      cnt += m;
#pragma unroll
      for (int w = 0; w < NW; w++) bch[w] = acc[w] * sc;
      if (!(op.x & OPF_NOPERSIST)) {  // re-read later through the scratch copy
        double *out = a.scratch + (((size_t)set * a.I + op.y) * a.ntiles + tile0) * TILE;
#pragma unroll
        for (int w = 0; w < NW; w++) {
          st16(out, (unsigned)((2 * w) * 64 + lane) * 16u, (f64x2){bch[w][0], bch[w][1]});
          st16(out, (unsigned)((2 * w + 1) * 64 + lane) * 16u, (f64x2){bch[w][2], bch[w][3]});
        }
        if (g == 0) a.scratch_cnt[((size_t)set * a.I + op.y) * a.S_pad + tile0 * 16 + sl] = cnt;
      }
      if (slot >= 2) {
        double *dst = park + (slot - 2) * TILE;
#pragma unroll
        for (int w = 0; w < NW; w++) {
          *reinterpret_cast<f64x2 *>(dst + ((2 * w) * 64 + lane) * 2) = (f64x2){bch[w][0], bch[w][1]};
          *reinterpret_cast<f64x2 *>(dst + ((2 * w + 1) * 64 + lane) * 2) = (f64x2){bch[w][2], bch[w][3]};
        }
        park_cnt[slot - 2][sl] = cnt;
      }
      bcnt = cnt;
      cnt = 0;
    }
    op = nxt;
  }

  // root: log L_s = log(sum_k root[s][k] pi[k]) - 64 ln2 * exponent   (likefunc.cpp:11123 for one pattern)
  double s = 0.;
#pragma unroll
  for (int kk = 0; kk < NKK; kk++) s = fma(bch[kk >> 2][kk & 3], a.pi[4 * kk + g], s);
  s = row_sum4(s);
  if (g == 0) {
    const int site = tile0 * 16 + sl;
    double ll;
    if (s != s || isinf(s)) {
      ll = s;
      if (a.freq[site] != 0.) atomicOr(a.status, 2);
    } else if (s <= 0.) {
      ll = -HUGE_VAL;
    } else {
      ll = log(s) - kLogScaler * (double)bcnt;
    }
    a.site_logl[(size_t)set * a.S_pad + site] = ll;
  }
}

template <int NW>
void launch_site_fit_NW(const SiteFitArgs &a, hipStream_t stream) {
  const dim3 grid(a.ntiles, a.n_sets), block(64);
  const size_t lds = (size_t)a.L * 16 * sizeof(int16_t);
  if (a.n_mix > 1) hipLaunchKernelGGL((site_fit_kernel<NW, kSiteFitParkSlots, true>), grid, block, lds, stream, a.ops, a.Timg, a.bcoef, a.bgroup, a);
  else hipLaunchKernelGGL((site_fit_kernel<NW, kSiteFitParkSlots, false>), grid, block, lds, stream, a.ops, a.Timg, a.bcoef, a.bgroup, a);
}

}  // namespace

void launch_site_fit(const SiteFitArgs &a, hipStream_t stream) {
  switch (a.NW) {
    case 1: launch_site_fit_NW<1>(a, stream); break;
    case 2: launch_site_fit_NW<2>(a, stream); break;
    case 3: launch_site_fit_NW<3>(a, stream); break;
    default: launch_site_fit_NW<4>(a, stream); break;
  }
}

}  // namespace hyhip
