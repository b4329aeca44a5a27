// Batched matrix exponential P = exp(Q) on gfx950 — the device counterpart of the OpenMP loop
// in _TheTree::ExponentiateMatrices (src/core/tree.cpp:3011-3037), each iteration of which is
// _Matrix::Exponentiate(1., true, storage) (src/core/matrix.cpp:5537-5951).
//
// Same numerical contract as the reference (scaling by a power of two, Taylor series,
// diag_populator forcing row sums to 1 before AND after the squarings, restart with a 100x
// larger scale when a diagonal exceeds 1, early exit from the squarings), re-designed for the
// matrix cores: one workgroup per matrix, everything resident in LDS/registers, every product a
// DPxDP FP64 MFMA GEMM (v_mfma_f64_16x16x4_f64).  The degree-12 Taylor polynomial is evaluated
// by Paterson-Stockmeyer in 5 products instead of the reference's one sparse product per term
// (matrix.cpp:5703-5721): on the GPU a dense 64^3 MFMA product is cheaper than CSR bookkeeping
// (SURVEY §2.1 K10).  With ||Q/2^p|| <= 1/4 the truncation error is < 2.5e-18.
#include "common.h"
#include "expm4.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>

namespace hyhip {

namespace {

// phase stamps of workgroup 0 (diagnostic, HYPHY_HIP_EXPM_PROF): start, Q built, norms, Taylor polynomial,
// squarings done, result in LDS, images written
__device__ long long g_expm_prof[8];

__device__ __forceinline__ f64x4 mfma(double a, double b, f64x4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// 1/k!
__constant__ double kInvFact[13] = {1.0,
                                    1.0,
                                    0.5,
                                    1.0 / 6.0,
                                    1.0 / 24.0,
                                    1.0 / 120.0,
                                    1.0 / 720.0,
                                    1.0 / 5040.0,
                                    1.0 / 40320.0,
                                    1.0 / 362880.0,
                                    1.0 / 3628800.0,
                                    1.0 / 39916800.0,
                                    1.0 / 479001600.0};

// NT = DP/16 row blocks; CS = column splits: the workgroup has NT*CS waves, wave (w, h) owns row block
// w and the NTW = NT/CS column tiles [h*NTW, (h+1)*NTW).  CS = 2 for DP = 64 puts two waves on every
// SIMD: one wave alone cannot issue f64 MFMAs fast enough to keep the pipe busy (tools/ubench_mfma_f64).
template <int NTW>
struct Frag {
  f64x4 t[NTW];  // C/D image: t[c][r] = M[16w + 4r + g][16(c0 + c) + sl]
};

template <int NT, int CS>
__device__ __forceinline__ Frag<NT / CS> mm(const double *__restrict__ Lm, const double *__restrict__ Rm, int w, int c0,
                                            int g, int sl) {
  constexpr int DP = 16 * NT, LD = DP + 2, NKK = DP / 4, NTW = NT / CS;
  Frag<NTW> d;
#pragma unroll
  for (int c = 0; c < NTW; c++) d.t[c] = (f64x4){0., 0., 0., 0.};
#pragma unroll 4
  for (int kk = 0; kk < NKK; kk++) {
    const double av = Lm[(16 * w + sl) * LD + 4 * kk + g];
#pragma unroll
    for (int c = 0; c < NTW; c++) {
      const double bv = Rm[(4 * kk + g) * LD + 16 * (c0 + c) + sl];
      d.t[c] = mfma(av, bv, d.t[c]);
    }
  }
  return d;
}

template <int NT, int CS>
__device__ __forceinline__ void store_frag(double *__restrict__ M, const Frag<NT / CS> &f, int w, int c0, int g,
                                           int sl) {
  constexpr int DP = 16 * NT, LD = DP + 2, NTW = NT / CS;
#pragma unroll
  for (int c = 0; c < NTW; c++)
#pragma unroll
    for (int r = 0; r < 4; r++) M[(16 * w + 4 * r + g) * LD + 16 * (c0 + c) + sl] = f.t[c][r];
}

// Row sums -> R_ii += 1 - sum (diag_populator, matrix.cpp:5837-5852); reports a diagonal > 1
// (transition_verifier, :5820-5835) or a NaN (workgroup-wide OR via LDS).  With CS > 1 the row sums
// are completed across the column-split waves through `rowpart` ([DP][CS] doubles in LDS).
template <int NT, int CS>
__device__ __forceinline__ bool diag_fix(Frag<NT / CS> &R, int w, int h, int c0, int g, int sl, int *flag,
                                         double *rowpart) {
  constexpr int NTW = NT / CS;
  bool bad = false;
  double part[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    double s = 0.;
#pragma unroll
    for (int c = 0; c < NTW; c++) s += R.t[c][r];
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 4);
    s += __shfl_xor(s, 8);
    part[r] = s;
    if (CS > 1 && sl == 0) rowpart[(16 * w + 4 * r + g) * CS + h] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) *flag = 0;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    double s = part[r];
    if (CS > 1) {
      s = 0.;
#pragma unroll
      for (int hh = 0; hh < CS; hh++) s += rowpart[(16 * w + 4 * r + g) * CS + hh];
    }
    if (s != s) bad = true;
    if (sl == 4 * r + g) {  // this lane holds the diagonal of row 16w + 4r + g if column tile w is ours
#pragma unroll
      for (int c = 0; c < NTW; c++)
        if (c0 + c == w) {
          if (R.t[c][r] > 1.) bad = true;
          R.t[c][r] += 1. - s;
        }
    }
  }
  __syncthreads();
  if (bad) atomicOr(flag, 1);
  __syncthreads();
  return *flag == 0;
}

template <int NT, int CS>
__global__ __launch_bounds__(64 * NT * CS) void expm_mfma_kernel(ExpmArgs a) {
  constexpr int DP = 16 * NT, LD = DP + 2, NKK = DP / 4, MS = DP * LD, NWV = NT * CS, NTW = NT / CS,
                NTHR = 64 * NWV;
  extern __shared__ __align__(16) double sm[];
  double *Xs = sm, *Ys = sm + MS, *Zs = sm + 2 * MS, *red = sm + 3 * MS;  // red: 2*DP + 8 doubles
  int *flag = reinterpret_cast<int *>(red + 2 * DP + 4);
  double *rowpart = red + 2 * DP + 8;  // [DP][CS]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g = lane >> 4, sl = lane & 15;
  const int w = wv % NT, h = wv / NT, c0 = h * NTW;
  const int D = a.D;
  const int m = blockIdx.x;
  const bool prof = a.prof && blockIdx.x == 0 && tid == 0;
  if (prof) g_expm_prof[0] = clock64();
  const int slot = a.slots ? a.slots[m] : m;
  const double *Q = a.Q + (size_t)m * D * D;

  bool diag_pending = false;  // (uniform) fused construction: the diagonal of Q is still zero in LDS
  if (a.templates) {
    // fused device-side rate-matrix construction: Q = sum_k c_k T_k; the diagonal = -(row sum) comes out of the segmented norm
    // pass below (partial sums combined by shuffles: it can differ from the serial column order of _Matrix::MultByFreqs,
    // matrix.cpp:1664-1674, in the last bit).  Every load of a thread is
    // independent of the others (clamped index + mask instead of a branch), so the whole build costs one
    // memory round trip instead of one per element.
    constexpr int EPT = DP * DP / NTHR;  // elements per thread (DP*DP is a multiple of the block size)
    const int K = a.K;
    const double *ck = a.coeffs + (size_t)m * K;
    double v[EPT];
#pragma unroll
    for (int e = 0; e < EPT; e++) v[e] = 0.;
    // (specialised on K <= 4 so that the loads of ALL templates are in flight together: one memory round trip, not K)
    auto accumulate = [&](auto kc) {
      constexpr int KC = decltype(kc)::value;
      double cf[KC], t[KC][EPT];
#pragma unroll
      for (int k = 0; k < KC; k++) cf[k] = ck[k];
#pragma unroll
      for (int k = 0; k < KC; k++) {
        const double *Tk = a.templates + (size_t)k * D * D;
#pragma unroll
        for (int e = 0; e < EPT; e++) {
          const int idx = tid + e * NTHR, r = idx / DP, c = idx - r * DP;
          const bool in = r < D && c < D && r != c;
          t[k][e] = Tk[in ? r * D + c : 0];
        }
      }
#pragma unroll
      for (int k = 0; k < KC; k++)
#pragma unroll
        for (int e = 0; e < EPT; e++) {
          const int idx = tid + e * NTHR, r = idx / DP, c = idx - r * DP;
          const bool in = r < D && c < D && r != c;
          v[e] += in ? cf[k] * t[k][e] : 0.;
        }
    };
    switch (K) {
      case 1: accumulate(std::integral_constant<int, 1>()); break;
      case 2: accumulate(std::integral_constant<int, 2>()); break;
      case 3: accumulate(std::integral_constant<int, 3>()); break;
      case 4: accumulate(std::integral_constant<int, 4>()); break;
      default:
        for (int k = 0; k < K; k++) {
          const double cf = ck[k];
          const double *Tk = a.templates + (size_t)k * D * D;
#pragma unroll
          for (int e = 0; e < EPT; e++) {
            const int idx = tid + e * NTHR, r = idx / DP, c = idx - r * DP;
            const bool in = r < D && c < D && r != c;
            const double t = Tk[in ? r * D + c : 0];
            v[e] += in ? cf * t : 0.;
          }
        }
    }
#pragma unroll
    for (int e = 0; e < EPT; e++) {
      const int idx = tid + e * NTHR, r = idx / DP, c = idx - r * DP;
      Xs[r * LD + c] = v[e];
    }
    diag_pending = true;  // (the diagonal = minus the row sum comes out of the norm pass below: one pass over the matrix, not two)
  } else {
    for (int idx = tid; idx < DP * DP; idx += NTHR) {
      const int r = idx / DP, c = idx - r * DP;
      Xs[r * LD + c] = (r < D && c < D) ? Q[r * D + c] : 0.0;
    }
  }
  __syncthreads();
  if (prof) g_expm_prof[1] = clock64();

  if (!a.is_prob) {
    // max row / column absolute sums (RowAndColumnMax, matrix.cpp:4901): the block's threads split every
    // row / column into NTHR/DP contiguous parts, combined in fixed order
    constexpr int PARTS = NTHR / DP >= 1 ? NTHR / DP : 1, SEG = DP / PARTS;
    {
      const int line = tid / PARTS, part = tid - line * PARTS;
      double rs = 0., cs = 0., rsum = 0.;
      if (line < DP) {
#pragma unroll
        for (int k = 0; k < SEG; k++) {
          const double x = Xs[line * LD + part * SEG + k];
          rsum += x;
          rs += fabs(x);
          cs += fabs(Xs[(part * SEG + k) * LD + line]);
        }
      }
#pragma unroll
      for (int off = 1; off < PARTS; off <<= 1) {  // PARTS is a power of two <= 8: partners sit in the same wave
        rs += __shfl_xor(rs, off);
        cs += __shfl_xor(cs, off);
        rsum += __shfl_xor(rsum, off);
      }
      if (line < DP && part == 0) {
        if (diag_pending && line < D) {  // fused construction: Q_ii = -(row sum); it enters both absolute sums
          Xs[line * LD + line] = -rsum;
          rs += fabs(rsum);
          cs += fabs(rsum);
        }
        red[line] = rs;
        red[DP + line] = cs;
      }
    }
    __syncthreads();
    // every wave forms the two maxima for itself (DP <= 64: one value per lane, then a butterfly)
    double rmax = lane < DP ? red[lane] : 0., cmax = lane < DP ? red[DP + lane] : 0.;
    const bool nan_in = __any(rmax != rmax || cmax != cmax);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      rmax = fmax(rmax, __shfl_xor(rmax, off));
      cmax = fmax(cmax, __shfl_xor(cmax, off));
    }
    const double mnorm = rmax * cmax;
    int p = 0;
    if (mnorm > 0.) {
      const double s = 4. * sqrt(mnorm);  // ||Q / 2^p|| <= 1/4  (||.||_2 <= sqrt(||.||_1 ||.||_inf))
      if (s > 1.) p = ilogb(s) + 1;
    }
    // original Q in registers (C/D image) so that a restart can rescale it
    if (prof) g_expm_prof[2] = clock64();
    Frag<NTW> Qr;
#pragma unroll
    for (int c = 0; c < NTW; c++)
#pragma unroll
      for (int r = 0; r < 4; r++) Qr.t[c][r] = Xs[(16 * w + 4 * r + g) * LD + 16 * (c0 + c) + sl];
    __syncthreads();

    auto add_diag = [&](Frag<NTW> &f, double v) {
#pragma unroll
      for (int c = 0; c < NTW; c++)
#pragma unroll
        for (int r = 0; r < 4; r++)
          if (c0 + c == w && sl == 4 * r + g) f.t[c][r] += v;
    };

    Frag<NTW> R;
    bool done = false, failed = nan_in || !(mnorm < 1e300);
    for (int attempt = 0; attempt < 48 && !done && !failed; attempt++) {
      const double scale = ldexp(1.0, -p);
      Frag<NTW> Xr;
#pragma unroll
      for (int c = 0; c < NTW; c++) Xr.t[c] = Qr.t[c] * scale;
      store_frag<NT, CS>(Xs, Xr, w, c0, g, sl);
      __syncthreads();
      Frag<NTW> X2 = mm<NT, CS>(Xs, Xs, w, c0, g, sl);
      store_frag<NT, CS>(Ys, X2, w, c0, g, sl);
      __syncthreads();
      Frag<NTW> X3 = mm<NT, CS>(Xs, Ys, w, c0, g, sl);
      store_frag<NT, CS>(Zs, X3, w, c0, g, sl);
      // Paterson-Stockmeyer, s = 3:  p(X) = B0 + X3 (B1 + X3 (B2 + X3 (B3 + c12 X3)))
      Frag<NTW> acc;
#pragma unroll
      for (int c = 0; c < NTW; c++)
        acc.t[c] = kInvFact[10] * Xr.t[c] + kInvFact[11] * X2.t[c] + kInvFact[12] * X3.t[c];
      add_diag(acc, kInvFact[9]);
#pragma unroll 1
      for (int blk = 2; blk >= 0; blk--) {
        __syncthreads();  // previous readers of Ys are done (and Zs is complete on the first pass)
        store_frag<NT, CS>(Ys, acc, w, c0, g, sl);
        __syncthreads();
        acc = mm<NT, CS>(Zs, Ys, w, c0, g, sl);
        const double k0 = kInvFact[3 * blk], k1 = kInvFact[3 * blk + 1], k2 = kInvFact[3 * blk + 2];
#pragma unroll
        for (int c = 0; c < NTW; c++) acc.t[c] += k1 * Xr.t[c] + k2 * X2.t[c];
        add_diag(acc, k0);
      }
      R = acc;
      if (prof) g_expm_prof[3] = clock64();
      if (!diag_fix<NT, CS>(R, w, h, c0, g, sl, flag, rowpart)) {  // matrix.cpp:5854-5864: restart, scale_to *= 100
        p += 7;
        if (p > 900) failed = true;
        continue;
      }
      double last_diff = 0.;
      for (int s = 0; s < p; s++) {  // matrix.cpp:5873-5920
        __syncthreads();
        store_frag<NT, CS>(Xs, R, w, c0, g, sl);
        __syncthreads();
        Frag<NTW> Rn = mm<NT, CS>(Xs, Xs, w, c0, g, sl);
        double diff = 0.;
#pragma unroll
        for (int c = 0; c < NTW; c++)
#pragma unroll
          for (int r = 0; r < 4; r++) diff = fmax(diff, fabs(Rn.t[c][r] - R.t[c][r]));
        R = Rn;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) diff = fmax(diff, __shfl_xor(diff, off));
        if (lane == 0) red[wv] = diff;
        __syncthreads();
        diff = red[0];
#pragma unroll
        for (int k = 1; k < NWV; k++) diff = fmax(diff, red[k]);
        if (diff < 2.220446049250313e-16 * 1.e3 || (s >= 10 && diff > last_diff * 100.)) break;
        last_diff = diff;
      }
      if (p > 0 && !diag_fix<NT, CS>(R, w, h, c0, g, sl, flag, rowpart)) {
        p += 7;
        if (p > 900) failed = true;
        continue;
      }
      done = true;
    }
    if (!done) {
      // No valid transition matrix (the reference raises "Failed to compute a valid transition matrix" here).  Besides the
      // sticky status word, the matrix images are poisoned with NaN: the log-likelihood of THIS evaluation becomes NaN
      // — also on the asynchronous device path, where nobody reads the status word before an all-reduce — instead of
      // being computed with the previous, stale matrix of the slot.
      if (tid == 0) atomicOr(a.status, 1);
#pragma unroll
      for (int c = 0; c < NTW; c++) R.t[c] = (f64x4){NAN, NAN, NAN, NAN};
    }
    if (prof) g_expm_prof[4] = clock64();
    __syncthreads();
    store_frag<NT, CS>(Xs, R, w, c0, g, sl);
    __syncthreads();
  }
  if (prof) g_expm_prof[5] = clock64();

  // ---- outputs: row-major, A-operand image, column-gather image ----
  if (a.Prow) {
    double *out = a.Prow + (size_t)slot * D * D;
    for (int idx = tid; idx < D * D; idx += NTHR) {
      const int r = idx / D, c = idx - r * D;
      out[idx] = Xs[r * LD + c];
    }
  }
  if (a.Pfrag) {
    // wave wb, k-step kk, lane l  <-  P[16 wb + (l & 15)][4 kk + (l >> 4)]
    double *out = a.Pfrag + (size_t)slot * DP * DP;
    for (int idx = tid; idx < DP * DP; idx += NTHR) {
      const int wb = idx / (NKK * 64), rem = idx - wb * (NKK * 64);
      const int kk2 = rem >> 7, l = (rem >> 1) & 63, kb = rem & 1, kk = 2 * kk2 + kb;
      const int rr = 16 * wb + (l & 15), cc = 4 * kk + (l >> 4);
      out[idx] = (rr < D && cc < D) ? Xs[rr * LD + cc] : 0.0;  // padded states carry exact zeros
    }
  }
  if (a.Pfrag && a.n_twin > 0) {
    for (int j = 0; j < a.n_twin; j++)
      if (slot == a.twin_src[j]) {  // (uniform) transposed twin for re-rooted schedules, pi-scaled on the old root's edge
        double *out = a.Pfrag + (size_t)(a.twin_dst0 + j) * DP * DP;
        for (int idx = tid; idx < DP * DP; idx += NTHR) {
          const int wb = idx / (NKK * 64), rem = idx - wb * (NKK * 64);
          const int kk2 = rem >> 7, l = (rem >> 1) & 63, kb = rem & 1, kk = 2 * kk2 + kb;
          const int rr = 16 * wb + (l & 15), cc = 4 * kk + (l >> 4);
          double v = (rr < D && cc < D) ? Xs[cc * LD + rr] : 0.0;
          if (j == 0) v *= a.twin_pi[cc];
          out[idx] = v;
        }
      }
  }
  if (a.PTg) {
    // [code][wb][gg][r]  <-  P[16 wb + 4 r + gg][code]
    double *out = a.PTg + (size_t)slot * DP * DP;
    for (int idx = tid; idx < DP * DP; idx += NTHR) {
      const int code = idx / (NT * 16), rem = idx - code * (NT * 16);
      const int wb = rem >> 4, gg = (rem >> 2) & 3, r = rem & 3;
      const int rr = 16 * wb + 4 * r + gg;
      out[idx] = (rr < D && code < D) ? Xs[rr * LD + code] : 0.0;
    }
  }
  if (prof) g_expm_prof[6] = clock64();
}

// ---------------------------------------------------------------------------------------------
// DP = 64 (49..64 states; the codon case), r03: `expm64_kernel<H>`.
//
// Why another kernel: at the headline workload (125 branches) expm_mfma_kernel<4, 2> keeps 125 of the 256 CUs busy for
// ~21 us of which 12.7 us are five 64^3 products at the instruction's issue rate on ONE CU — the other
// 131 CUs idle, and a product cannot be split over CUs without an exchange per product.  What CAN be split without any
// exchange is the Horner part of Paterson-Stockmeyer: p(X) = B0 + (B1 + (B2 + (B3 + c12 X3) X3) X3) X3 only ever multiplies
// a ROW PANEL of the running value into the full X3 from the right, so H workgroups per matrix each form X2 and X3 in full
// (duplicated) and then carry 64/H rows of the polynomial: 2 + 3/H products per CU instead of 5, row sums (the
// diag_populator fix-up) are panel-local, and each workgroup writes its rows of the three matrix images.
// The squarings need the whole matrix, so panels are only used for matrices that need none (p = 0: ||Q||_inf <= 1/4,
// every workgroup of a matrix derives the same p from the same Q); for p > 0 panel 0 does all the work and the others
// retire.  Further differences to expm_mfma_kernel:
//  * scaling exponent from the infinity norm alone (max absolute row sum): the Taylor remainder bound ||X||^13/13! holds in
//    any submultiplicative norm, the row sums fall out of the rate-matrix construction, and the column pass is gone;
//  * the rate-matrix construction reads 64x64-padded templates (aligned 16-byte loads, no masks) with 8 threads per row,
//    so diagonal and norm are three shuffles away; coefficients may arrive in the kernel-argument block (no PCIe reads
//    from 125 workgroups at once);
//  * LDS row stride 65 doubles: the A-operand reads (16 rows x 4 columns per wave access) are conflict-free as well;
//  * the column-gather image is written straight from the C/D registers (a lane holds 4 consecutive entries of it).
// In panel mode a transition_verifier failure (a diagonal > 1 after the Taylor polynomial of a matrix with ||X||_inf <= 1/4:
// the input is not a rate matrix) is reported as failure right away; the full-mode restart with a 2^7 larger scale would
// square its way back to the same matrix and fail after the retries (matrix.cpp:5854-5864).
// ---------------------------------------------------------------------------------------------
// Degree of the Taylor polynomial from the norm of the (scaled) matrix: 3 nb with nb blocks of Paterson-Stockmeyer in X^3.
// Remainder ||X||^(m+1) / (m+1)!: 12 for ||X|| <= 1/4 (2.4e-18), 9 for <= 0.11 (7e-17), 6 for <= 1/64 (4.5e-17) — each block
// less is one product less behind X^2 and X^3 (a third of the Horner part at the headline's ||Q t|| = 0.10).
__device__ __forceinline__ int taylor_blocks(double scaled_norm, int fixed_degree) {
  if (fixed_degree) return 4;
  return scaled_norm <= 0.015625 ? 2 : (scaled_norm <= 0.11 ? 3 : 4);
}

template <int NTW>
__device__ __forceinline__ void load_tiles(const double *__restrict__ M, f64x4 (&f)[NTW], int rb, int ct0, int g, int sl) {
  constexpr int LD = 65;
#pragma unroll
  for (int c = 0; c < NTW; c++)
#pragma unroll
    for (int r = 0; r < 4; r++) f[c][r] = M[(16 * rb + 4 * r + g) * LD + 16 * (ct0 + c) + sl];
}
template <int NTW>
__device__ __forceinline__ void store_tiles(double *__restrict__ M, const f64x4 (&f)[NTW], int rb, int ct0, int g, int sl) {
  constexpr int LD = 65;
#pragma unroll
  for (int c = 0; c < NTW; c++)
#pragma unroll
    for (int r = 0; r < 4; r++) M[(16 * rb + 4 * r + g) * LD + 16 * (ct0 + c) + sl] = f[c][r];
}
// d[c] = (rows 16 rb .. of Lm) x (column tiles ct0 + c of Rm); two accumulator chains per tile (even / odd k-steps)
template <int NTW>
__device__ __forceinline__ void mm64(const double *__restrict__ Lm, const double *__restrict__ Rm, f64x4 (&d)[NTW], int rb,
                                     int ct0, int g, int sl) {
  constexpr int LD = 65;
  f64x4 e[NTW];
#pragma unroll
  for (int c = 0; c < NTW; c++) d[c] = e[c] = (f64x4){0., 0., 0., 0.};
#pragma unroll 4
  for (int kk = 0; kk < 16; kk += 2) {
    // k-slot g of MFMA step kk carries k = 16 g + kk (any bijection does): with the row stride of 65 doubles the 32 lanes of a
    // ds_read_b64 lane group (sl = 0..15, g = 0, 1 / 2, 3) then fall on 32 different bank pairs for the A operand (sl + 16 g)
    // AND for the B operand (16 g 65 + sl = 16 g + sl mod 32).  k = 4 kk + g (r03) put both on sl + g: two-way conflicts on
    // every operand read (SQ_LDS_BANK_CONFLICT of the headline launch: 382 620 -> 254 620; the rest are the C-layout tile stores /
    // loads and the image passes, profiles/r04_pmc_final_build_lds.json).
    const double a0 = Lm[(16 * rb + sl) * LD + 16 * g + kk], a1 = Lm[(16 * rb + sl) * LD + 16 * g + kk + 1];
#pragma unroll
    for (int c = 0; c < NTW; c++) {
      const double b0 = Rm[(16 * g + kk) * LD + 16 * (ct0 + c) + sl], b1 = Rm[(16 * g + kk + 1) * LD + 16 * (ct0 + c) + sl];
      d[c] = mfma(a0, b0, d[c]);
      e[c] = mfma(a1, b1, e[c]);
    }
  }
#pragma unroll
  for (int c = 0; c < NTW; c++) d[c] += e[c];
}


template <int H>
__global__ __launch_bounds__(512) void expm64_kernel(ExpmArgs a, CoefInline ci) {
  constexpr int DP = 64, LD = 65, MS = DP * LD, NTHR = 512;
  constexpr int NTW = (H == 1) ? 2 : 1;  // result tiles per active wave
  extern __shared__ __align__(16) double sm[];
  double *Xs = sm, *Ys = sm + MS, *Zs = sm + 2 * MS, *red = sm + 3 * MS;  // red: 16 doubles
  double *rowpart = red + 16;                                            // [DP][4]
  int *flag = reinterpret_cast<int *>(rowpart + 4 * DP);
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, sl = lane & 15;
  const int D = a.D;
  const int m = blockIdx.x / H, panel = blockIdx.x - m * H;
  const bool prof = a.prof && blockIdx.x == 0 && tid == 0;
  if (prof) g_expm_prof[0] = clock64();
  const int slot = a.slots ? a.slots[m] : m;
  // full products (X^2, X^3): wave (fw, fh) owns row block fw and column tiles 2 fh, 2 fh + 1
  const int fw = wv & 3, fc0 = (wv >> 2) * 2;

  // ---- the matrix, 8 threads per row ----
  const int br = tid >> 3, bcs = tid & 7;
  double v[8];
  double rabs = 0.;
  if (a.templates_pad) {
    const int K = a.K;
    const double *ck = a.coeffs + (size_t)m * K;
    const bool inl = a.coef_inline != 0;  // (uniform) coefficients in the kernel-argument block
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = 0.;
    auto accumulate = [&](auto kc) {  // (specialised on K <= 4: the loads of ALL templates are in flight together)
      constexpr int KC = decltype(kc)::value;
      double cf[KC];
      f64x2 t[KC][4];
      if (inl) {
#pragma unroll
        for (int k = 0; k < KC; k++) cf[k] = ci.c[m * KC + k];
      } else {
#pragma unroll
        for (int k = 0; k < KC; k++) cf[k] = ck[k];
      }
#pragma unroll
      for (int k = 0; k < KC; k++) {
        const f64x2 *Tk = reinterpret_cast<const f64x2 *>(a.templates_pad + (size_t)k * DP * DP + br * DP + 8 * bcs);
#pragma unroll
        for (int q = 0; q < 4; q++) t[k][q] = Tk[q];
      }
#pragma unroll
      for (int k = 0; k < KC; k++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
          v[2 * q] += cf[k] * t[k][q][0];
          v[2 * q + 1] += cf[k] * t[k][q][1];
        }
    };
    switch (K) {
      case 1: accumulate(std::integral_constant<int, 1>()); break;
      case 2: accumulate(std::integral_constant<int, 2>()); break;
      case 3: accumulate(std::integral_constant<int, 3>()); break;
      case 4: accumulate(std::integral_constant<int, 4>()); break;
      default:
        for (int k = 0; k < K; k++) {
          const double cf = inl ? ci.c[m * K + k] : ck[k];
          const double *Tk = a.templates_pad + (size_t)k * DP * DP + br * DP + 8 * bcs;
#pragma unroll
          for (int e = 0; e < 8; e++) v[e] += cf * Tk[e];
        }
    }
    // (padded templates carry zeros on the diagonal and in the padding: the row sum is the off-diagonal sum)
    double rsum = 0.;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      rsum += v[e];
      rabs += fabs(v[e]);
    }
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) {
      rsum += __shfl_xor(rsum, off);
      rabs += __shfl_xor(rabs, off);
    }
    rabs += fabs(rsum);
#pragma unroll
    for (int e = 0; e < 8; e++)
      if (bcs == (br >> 3) && e == (br & 7)) v[e] = -rsum;  // Q_ii = -(row sum)
  } else {
    const double *Q = a.Q + (size_t)m * D * D;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const int c = 8 * bcs + e;
      const bool in = br < D && c < D;
      const double x = Q[in ? br * D + c : 0];
      v[e] = in ? x : 0.;
      rabs += fabs(v[e]);
    }
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) rabs += __shfl_xor(rabs, off);
  }
#pragma unroll
  for (int e = 0; e < 8; e++) Xs[br * LD + 8 * bcs + e] = v[e];
  {
    const bool nan_here = __any(rabs != rabs);
    double wmax = rabs;
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) wmax = fmax(wmax, __shfl_xor(wmax, off));
    if (lane == 0) {
      red[wv] = wmax;
      red[8 + wv] = nan_here ? 1. : 0.;
    }
  }
  __syncthreads();
  if (prof) g_expm_prof[1] = clock64();

  int rb = fw, ct0 = fc0;  // result tiles of this wave (H = 1: the full-product mapping)
  bool active = true;
  f64x4 R[NTW];
  if (!a.is_prob) {
    double norm = 0.;
    bool nan_in = false;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      norm = fmax(norm, red[k]);
      nan_in = nan_in || red[8 + k] != 0.;
    }
    int p = 0;
    if (norm > 0.) {
      const double s = 4. * norm;  // ||Q / 2^p||_inf <= 1/4
      if (s > 1.) p = ilogb(s) + 1;
    }
    bool failed = nan_in || !(norm < 1e300);
    const bool panels = H > 1 && p == 0 && !failed;  // (uniform over the H workgroups of the matrix)
    if (H > 1 && !panels && panel != 0) return;      // panel 0 does this matrix alone
    if (prof) g_expm_prof[2] = clock64();
    f64x4 Qf[2];  // the unscaled matrix, full-product mapping (restarts rescale it; H = 1: also the Horner terms)
    load_tiles<2>(Xs, Qf, fw, fc0, g, sl);
    if (H == 1 || !panels) {
      // ================= one workgroup, whole matrix: scaling, Taylor, squarings, restarts =================
      bool done = false;
      f64x4 Rf[2];
      for (int attempt = 0; attempt < 48 && !done && !failed; attempt++) {
        const double scale = ldexp(1.0, -p);
        f64x4 Xr[2], X2[2], X3[2], acc[2];
#pragma unroll
        for (int c = 0; c < 2; c++) Xr[c] = Qf[c] * scale;
        if (p > 0) {
          __syncthreads();
          store_tiles<2>(Xs, Xr, fw, fc0, g, sl);
          __syncthreads();
        }
        mm64<2>(Xs, Xs, X2, fw, fc0, g, sl);
        store_tiles<2>(Ys, X2, fw, fc0, g, sl);
        __syncthreads();
        mm64<2>(Xs, Ys, X3, fw, fc0, g, sl);
        store_tiles<2>(Zs, X3, fw, fc0, g, sl);
        auto add_diag = [&](f64x4 (&f)[2], double dv) {
#pragma unroll
          for (int c = 0; c < 2; c++)
#pragma unroll
            for (int r = 0; r < 4; r++)
              if (fc0 + c == fw && sl == 4 * r + g) f[c][r] += dv;
        };
        const int nb = taylor_blocks(norm * scale, a.fixed_degree);  // degree 3 nb from the scaled norm (uniform)
#pragma unroll
        for (int c = 0; c < 2; c++) acc[c] = kInvFact[3 * nb - 2] * Xr[c] + kInvFact[3 * nb - 1] * X2[c] + kInvFact[3 * nb] * X3[c];
        add_diag(acc, kInvFact[3 * nb - 3]);
#pragma unroll 1
        for (int blk = nb - 2; blk >= 0; blk--) {
          __syncthreads();  // previous readers of Ys are done (and Zs is complete on the first pass)
          store_tiles<2>(Ys, acc, fw, fc0, g, sl);
          __syncthreads();
          mm64<2>(Ys, Zs, acc, fw, fc0, g, sl);
          const double k0 = kInvFact[3 * blk], k1 = kInvFact[3 * blk + 1], k2 = kInvFact[3 * blk + 2];
#pragma unroll
          for (int c = 0; c < 2; c++) acc[c] += k1 * Xr[c] + k2 * X2[c];
          add_diag(acc, k0);
        }
#pragma unroll
        for (int c = 0; c < 2; c++) Rf[c] = acc[c];
        if (prof) g_expm_prof[3] = clock64();
        auto fix = [&]() -> bool {  // diag_populator + transition_verifier over the whole matrix
          bool bad = false;
          double part[4];
#pragma unroll
          for (int r = 0; r < 4; r++) {
            double s = Rf[0][r] + Rf[1][r];
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            s += __shfl_xor(s, 4);
            s += __shfl_xor(s, 8);
            part[r] = s;
            if (sl == 0) rowpart[(16 * fw + 4 * r + g) * 4 + (wv >> 2)] = s;
          }
          __syncthreads();
          if (tid == 0) *flag = 0;
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const double s = rowpart[(16 * fw + 4 * r + g) * 4] + rowpart[(16 * fw + 4 * r + g) * 4 + 1];
            if (s != s) bad = true;
            if (sl == 4 * r + g) {
#pragma unroll
              for (int c = 0; c < 2; c++)
                if (fc0 + c == fw) {
                  if (Rf[c][r] > 1.) bad = true;
                  Rf[c][r] += 1. - s;
                }
            }
          }
          __syncthreads();
          if (bad) atomicOr(flag, 1);
          __syncthreads();
          return *flag == 0;
        };
        if (!fix()) {  // matrix.cpp:5854-5864: restart, scale_to *= 100
          p += 7;
          if (p > 900) failed = true;
          continue;
        }
        double last_diff = 0.;
        for (int s = 0; s < p; s++) {  // matrix.cpp:5873-5920
          __syncthreads();
          store_tiles<2>(Xs, Rf, fw, fc0, g, sl);
          __syncthreads();
          f64x4 Rn[2];
          mm64<2>(Xs, Xs, Rn, fw, fc0, g, sl);
          double diff = 0.;
#pragma unroll
          for (int c = 0; c < 2; c++)
#pragma unroll
            for (int r = 0; r < 4; r++) diff = fmax(diff, fabs(Rn[c][r] - Rf[c][r]));
#pragma unroll
          for (int c = 0; c < 2; c++) Rf[c] = Rn[c];
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) diff = fmax(diff, __shfl_xor(diff, off));
          if (lane == 0) red[wv] = diff;
          __syncthreads();
          diff = red[0];
#pragma unroll
          for (int k = 1; k < 8; k++) diff = fmax(diff, red[k]);
          if (diff < 2.220446049250313e-16 * 1.e3 || (s >= 10 && diff > last_diff * 100.)) break;
          last_diff = diff;
        }
        if (p > 0 && !fix()) {
          p += 7;
          if (p > 900) failed = true;
          continue;
        }
        done = true;
      }
      if (!done) {  // sticky status + NaN images: this evaluation's log-L becomes NaN (see expm_mfma_kernel)
        if (tid == 0) atomicOr(a.status, 1);
#pragma unroll
        for (int c = 0; c < 2; c++) Rf[c] = (f64x4){NAN, NAN, NAN, NAN};
      }
      if (prof) g_expm_prof[4] = clock64();
      __syncthreads();
      store_tiles<2>(Xs, Rf, fw, fc0, g, sl);
      if (H == 1) {
#pragma unroll
        for (int c = 0; c < NTW; c++) R[c] = Rf[c];
      }
      __syncthreads();  // (H > 1: a matrix that left panel mode — this workgroup writes every row below)
    } else {
      // ================= panel mode (p = 0): rows [64/H * panel, ...) of the polynomial =================
      if (H == 2) {
        rb = 2 * panel + (wv & 1);
        ct0 = wv >> 1;
      } else {
        rb = panel;
        ct0 = wv & 3;
        active = wv < 4;
      }
      const int nb = taylor_blocks(norm, a.fixed_degree);  // (p = 0: the matrix is unscaled; uniform over the workgroup and its siblings)
      f64x4 X2f[2], X3f[2];
      mm64<2>(Xs, Xs, X2f, fw, fc0, g, sl);
      store_tiles<2>(Ys, X2f, fw, fc0, g, sl);
      __syncthreads();
      mm64<2>(Xs, Ys, X3f, fw, fc0, g, sl);
      store_tiles<2>(Zs, X3f, fw, fc0, g, sl);
      __syncthreads();
      f64x4 Xp[NTW], X2p[NTW], acc[NTW];
      auto add_diag = [&](f64x4 (&f)[NTW], double dv) {
#pragma unroll
        for (int c = 0; c < NTW; c++)
#pragma unroll
          for (int r = 0; r < 4; r++)
            if (ct0 + c == rb && sl == 4 * r + g) f[c][r] += dv;
      };
      if (active) {
        f64x4 X3p[NTW];
        load_tiles<NTW>(Xs, Xp, rb, ct0, g, sl);
        load_tiles<NTW>(Ys, X2p, rb, ct0, g, sl);
        load_tiles<NTW>(Zs, X3p, rb, ct0, g, sl);
#pragma unroll
        for (int c = 0; c < NTW; c++) acc[c] = kInvFact[3 * nb - 2] * Xp[c] + kInvFact[3 * nb - 1] * X2p[c] + kInvFact[3 * nb] * X3p[c];
        add_diag(acc, kInvFact[3 * nb - 3]);
      }
#pragma unroll 1
      for (int blk = nb - 2; blk >= 0; blk--) {
        __syncthreads();  // every wave has its tiles of X^2 (first pass) / has read the previous panel
        if (active) store_tiles<NTW>(Ys, acc, rb, ct0, g, sl);
        __syncthreads();
        if (active) {
          mm64<NTW>(Ys, Zs, acc, rb, ct0, g, sl);
          const double k0 = kInvFact[3 * blk], k1 = kInvFact[3 * blk + 1], k2 = kInvFact[3 * blk + 2];
#pragma unroll
          for (int c = 0; c < NTW; c++) acc[c] += k1 * Xp[c] + k2 * X2p[c];
          add_diag(acc, k0);
        }
      }
      if (prof) g_expm_prof[3] = clock64();
      // diag_populator + transition_verifier over the panel's rows (4 column tiles, one per wave of a row block)
      bool bad = false;
      if (active) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          double s = acc[0][r];
          s += __shfl_xor(s, 1);
          s += __shfl_xor(s, 2);
          s += __shfl_xor(s, 4);
          s += __shfl_xor(s, 8);
          if (sl == 0) rowpart[(16 * rb + 4 * r + g) * 4 + ct0] = s;
        }
      }
      if (tid == 0) *flag = 0;
      __syncthreads();
      if (active) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const double *rp = rowpart + (16 * rb + 4 * r + g) * 4;
          const double s = (rp[0] + rp[1]) + (rp[2] + rp[3]);
          if (s != s) bad = true;
          if (sl == 4 * r + g && ct0 == rb) {
            if (acc[0][r] > 1.) bad = true;
            acc[0][r] += 1. - s;
          }
        }
      }
      if (bad) atomicOr(flag, 1);
      __syncthreads();
      if (*flag) {
        if (tid == 0) atomicOr(a.status, 1);
#pragma unroll
        for (int c = 0; c < NTW; c++) acc[c] = (f64x4){NAN, NAN, NAN, NAN};
      }
      if (prof) g_expm_prof[4] = clock64();
#pragma unroll
      for (int c = 0; c < NTW; c++) R[c] = acc[c];
      if (active) store_tiles<NTW>(Xs, R, rb, ct0, g, sl);
      __syncthreads();
    }
    // rows this workgroup writes: [row0, row0 + nrows)
    const bool whole = (H == 1) || !(H > 1 && p == 0 && !failed);
    if (prof) g_expm_prof[5] = clock64();
    const int row0 = whole ? 0 : (DP / H) * panel, nrows = whole ? DP : DP / H;
    // ---- outputs for rows [row0, row0 + nrows): row-major, A-operand image, column-gather image, twins ----
    if (a.Prow) {
      double *out = a.Prow + (size_t)slot * D * D;
      for (int idx = tid; idx < nrows * DP; idx += NTHR) {
        const int r = row0 + (idx >> 6), c = idx & 63;
        if (r < D && c < D) out[r * D + c] = Xs[r * LD + c];
      }
    }
    // (r06) only the images this branch's consumers read: internal branches the A-operand image, leaf branches the column-gather
    // image (+ the A-operand image when the leaf carries ambiguity codes) — half of the launch's 8-16 MB of stores
    const int need = a.need ? (int)a.need[slot % a.need_B] : 3;
    if (a.Pfrag && (need & 1)) {
      // wave wb, k-step kk, lane l  <-  P[16 wb + (l & 15)][4 kk + (l >> 4)]; a row block is one contiguous 8 KiB run
      double *out = a.Pfrag + (size_t)slot * DP * DP + (size_t)row0 * DP;
      for (int idx = tid; idx < nrows * DP; idx += NTHR) {
        const int wb = idx >> 10, rem = idx & 1023;
        const int kk2 = rem >> 7, l = (rem >> 1) & 63, kb = rem & 1, kk = 2 * kk2 + kb;
        const int rr = row0 + 16 * wb + (l & 15), cc = 4 * kk + (l >> 4);
        out[idx] = (rr < D && cc < D) ? Xs[rr * LD + cc] : 0.0;  // padded states carry exact zeros
      }
    }
    if (a.Pfrag && a.n_twin > 0) {
      for (int j = 0; j < a.n_twin; j++)
        if (slot == a.twin_src[j]) {  // (uniform) transposed twin M[r][c] = P[c][r]: this workgroup owns the source rows c
          double *out = a.Pfrag + (size_t)(a.twin_dst0 + j) * DP * DP;
          for (int q = tid; q < nrows * DP; q += NTHR) {
            // columns cc in [row0, row0 + nrows): k-steps kk in [row0/4, ...), every row block wb
            const int wb = q / (nrows * 16), rem = q - wb * (nrows * 16);   // nrows * 16 elements per row block
            const int kkl = rem >> 6, l = rem & 63, kk = (row0 >> 2) + kkl;
            const int rr = 16 * wb + (l & 15), cc = 4 * kk + (l >> 4);
            double val = (rr < D && cc < D) ? Xs[cc * LD + rr] : 0.0;
            if (j == 0) val *= a.twin_pi[cc];
            out[wb * 1024 + (kk >> 1) * 128 + l * 2 + (kk & 1)] = val;
          }
        }
    }
    if (a.PTg && (need & 2)) {
      // [code][wb][gg][r]  <-  P[16 wb + 4 r + gg][code]: lane (g, sl) of the wave that owns tile (rb, ct) holds the four
      // consecutive entries r = 0..3 of code = 16 ct + sl  ->  32 contiguous bytes per lane, straight from the registers
      double *out = a.PTg + (size_t)slot * DP * DP;
      if (whole && H > 1) {  // (a matrix that left panel mode: R sits in the full-product mapping in LDS)
        for (int idx = tid; idx < DP * DP; idx += NTHR) {
          const int code = idx >> 6, rem = idx & 63;
          const int wb = rem >> 4, gg = (rem >> 2) & 3, r = rem & 3;
          const int rr = 16 * wb + 4 * r + gg;
          out[idx] = (rr < D && code < D) ? Xs[rr * LD + code] : 0.0;
        }
      } else if (active) {
#pragma unroll
        for (int c = 0; c < NTW; c++) {
          const int code = 16 * (ct0 + c) + sl;
          f64x4 o;
#pragma unroll
          for (int r = 0; r < 4; r++) o[r] = (16 * rb + 4 * r + g < D && code < D) ? R[c][r] : 0.0;
          f64x2 *dst = reinterpret_cast<f64x2 *>(out + ((size_t)code * 4 + rb) * 16 + g * 4);
          dst[0] = (f64x2){o[0], o[1]};
          dst[1] = (f64x2){o[2], o[3]};
        }
      }
    }
    if (prof) g_expm_prof[6] = clock64();
    return;
  }
  // ---- is_prob: the caller's matrices are transition probabilities already (layout conversion only; H = 1) ----
  if (prof) g_expm_prof[5] = clock64();
  if (a.Prow) {
    double *out = a.Prow + (size_t)slot * D * D;
    for (int idx = tid; idx < D * D; idx += NTHR) {
      const int r = idx / D, c = idx - r * D;
      out[idx] = Xs[r * LD + c];
    }
  }
  if (a.Pfrag) {
    double *out = a.Pfrag + (size_t)slot * DP * DP;
    for (int idx = tid; idx < DP * DP; idx += NTHR) {
      const int wb = idx >> 10, rem = idx & 1023;
      const int kk2 = rem >> 7, l = (rem >> 1) & 63, kb = rem & 1, kk = 2 * kk2 + kb;
      const int rr = 16 * wb + (l & 15), cc = 4 * kk + (l >> 4);
      out[idx] = (rr < D && cc < D) ? Xs[rr * LD + cc] : 0.0;
    }
  }
  if (a.Pfrag && a.n_twin > 0) {
    for (int j = 0; j < a.n_twin; j++)
      if (slot == a.twin_src[j]) {
        double *out = a.Pfrag + (size_t)(a.twin_dst0 + j) * DP * DP;
        for (int idx = tid; idx < DP * DP; idx += NTHR) {
          const int wb = idx >> 10, rem = idx & 1023;
          const int kk2 = rem >> 7, l = (rem >> 1) & 63, kb = rem & 1, kk = 2 * kk2 + kb;
          const int rr = 16 * wb + (l & 15), cc = 4 * kk + (l >> 4);
          double val = (rr < D && cc < D) ? Xs[cc * LD + rr] : 0.0;
          if (j == 0) val *= a.twin_pi[cc];
          out[idx] = val;
        }
      }
  }
  if (a.PTg) {
    double *out = a.PTg + (size_t)slot * DP * DP;
    for (int idx = tid; idx < DP * DP; idx += NTHR) {
      const int code = idx >> 6, rem = idx & 63;
      const int wb = rem >> 4, gg = (rem >> 2) & 3, r = rem & 3;
      const int rr = 16 * wb + 4 * r + gg;
      out[idx] = (rr < D && code < D) ? Xs[rr * LD + code] : 0.0;
    }
  }
  if (prof) g_expm_prof[6] = clock64();
}

// ---------------------------------------------------------------------------------------------
// 4-state specialisation: one thread per matrix, everything in registers.
// ---------------------------------------------------------------------------------------------
// (coefficients from the ring slot: a 3.2 KB kernel-argument block costs this 61-thread launch more than the PCIe read —
//  measured 11-15 against 9.8 us event-timed)
__global__ __launch_bounds__(64) void expm_nuc_kernel(ExpmArgs a) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= a.n) return;
  const int slot = a.slots ? a.slots[m] : m;
  double R[16];
  expm4_one(a, m, R, a.coeffs);
  if (a.Prow) {
#pragma unroll
    for (int k = 0; k < 16; k++) a.Prow[(size_t)slot * 16 + k] = R[k];
  }
  if (a.PTrow) {
#pragma unroll
    for (int k = 0; k < 16; k++) a.PTrow[(size_t)slot * 16 + k] = R[4 * (k & 3) + (k >> 2)];
  }
}

// Q_b = sum_k coeff[b][k] T_k off-diagonal, diagonal = -(row sum)   
__global__ __launch_bounds__(256) void build_q_kernel(const double *__restrict__ templates,
                                                      const double *__restrict__ coeffs, int K, int D,
                                                      double *__restrict__ Q) {
  const int b = blockIdx.x;
  extern __shared__ double qs[];  // [D*D]
  double *out = Q + (size_t)b * D * D;
  const int N = D * D;
  for (int idx = threadIdx.x; idx < N; idx += 256) {
    double v = 0.;
    for (int k = 0; k < K; k++) v += coeffs[(size_t)b * K + k] * templates[(size_t)k * N + idx];
    qs[idx] = v;
  }
  __syncthreads();
  // one thread per row keeps the column-order subtraction of MultByFreqs (matrix.cpp:1664-1674)
  for (int r = threadIdx.x; r < D; r += 256) {
    double d = 0.;
    for (int c = 0; c < D; c++)
      if (c != r) d -= qs[r * D + c];
    qs[r * D + r] = d;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < N; idx += 256) out[idx] = qs[idx];
}

// Branch-site mixtures (the reference's "explicit form" models, tree.cpp:3047-3090): P_b = sum_m w_bm exp(Q_bm).  The
// exponentials arrive row-major from the expm kernel; this writes the mixed matrix of every branch in the layouts the
// pruning kernels read (A-operand image + column-gather image, or row-major for the 4-state path).
__global__ __launch_bounds__(256) void mix_images_kernel(const double *__restrict__ P, const int *__restrict__ off,
                                                         const double *__restrict__ w, const int32_t *__restrict__ slots,
                                                         int D, int NT, double *__restrict__ Pfrag, double *__restrict__ PTg,
                                                         double *__restrict__ Prow, double *__restrict__ PTrow) {
  const int b = blockIdx.x, m0 = off[b], m1 = off[b + 1], slot = slots ? slots[b] : b;
  const int DP = 16 * NT, NKK = DP / 4, DD = D * D;
  auto mixed = [&](int rr, int cc) -> double {
    if (rr >= D || cc >= D) return 0.0;  // padded states carry exact zeros
    double v = 0.;
    for (int m = m0; m < m1; m++) v += w[m] * P[(size_t)m * DD + rr * D + cc];
    return v;
  };
  if (Pfrag) {  // wave wb, k-step kk, lane l  <-  P[16 wb + (l & 15)][4 kk + (l >> 4)]
    double *out = Pfrag + (size_t)slot * DP * DP;
    for (int idx = threadIdx.x; idx < DP * DP; idx += blockDim.x) {
      const int wb = idx / (NKK * 64), rem = idx - wb * (NKK * 64);
      const int kk2 = rem >> 7, l = (rem >> 1) & 63, kb = rem & 1, kk = 2 * kk2 + kb;
      out[idx] = mixed(16 * wb + (l & 15), 4 * kk + (l >> 4));
    }
  }
  if (PTg) {  // [code][wb][gg][r]  <-  P[16 wb + 4 r + gg][code]
    double *out = PTg + (size_t)slot * DP * DP;
    for (int idx = threadIdx.x; idx < DP * DP; idx += blockDim.x) {
      const int code = idx / (NT * 16), rem = idx - code * (NT * 16);
      const int wb = rem >> 4, gg = (rem >> 2) & 3, r = rem & 3;
      out[idx] = mixed(16 * wb + 4 * r + gg, code);
    }
  }
  if (Prow) {
    double *out = Prow + (size_t)slot * DD;
    for (int idx = threadIdx.x; idx < DD; idx += blockDim.x) out[idx] = mixed(idx / D, idx % D);
  }
  if (PTrow) {
    double *out = PTrow + (size_t)slot * DD;
    for (int idx = threadIdx.x; idx < DD; idx += blockDim.x) out[idx] = mixed(idx % D, idx / D);
  }
}

}  // namespace

void launch_mix_images(const double *P, const int *off, const double *w, const int32_t *slots, int n, int D, double *Pfrag,
                       double *PTg, double *Prow, hipStream_t stream, double *PTrow) {
  if (n <= 0) return;
  hipLaunchKernelGGL(mix_images_kernel, dim3(n), dim3(256), 0, stream, P, off, w, slots, D, (D + 15) / 16, Pfrag, PTg, Prow, PTrow);
}

void expm_read_profile(long long out[8]) { hipMemcpyFromSymbol(out, HIP_SYMBOL(g_expm_prof), 8 * sizeof(long long)); }

// returns true when the coefficients of a fused construction travelled in the kernel-argument block (the caller's staging
// buffer is free again on return), false when the kernel will read them from `coeffs` when it executes
// Coefficients of the fused rate-matrix construction into the kernel-argument block when they fit: `coeffs_host` is the host's
// view of the (host-mapped) ring slot and is dereferenced HERE, by the caller's thread — staged before this call
// (hyphy_hip_build_q).  true: the launch no longer reads the ring slot.
bool fill_coef_inline(ExpmArgs &b, CoefInline &ci) {
  static const int coef_mode = getenv("HYPHY_HIP_COEF_INLINE") ? atoi(getenv("HYPHY_HIP_COEF_INLINE")) : 1;
  b.coef_inline = 0;
  if (b.templates && b.coeffs_host && coef_mode && (size_t)b.n * b.K <= (size_t)kCoefInline) {
    memcpy(ci.c, b.coeffs_host, (size_t)b.n * b.K * sizeof(double));
    b.coef_inline = 1;
  }
  return b.coef_inline != 0;
}

bool launch_expm(const ExpmArgs &a, hipStream_t stream) {
  if (a.n <= 0) return false;
  if (a.D == 4 && !a.Pfrag && !a.PTg) {
    hipLaunchKernelGGL(expm_nuc_kernel, dim3((a.n + 63) / 64), dim3(64), 0, stream, a);
    return false;
  }
  // hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of the function: the single-process
  // multi-device path (hyphy_hip_create with device_count > 1) launches on every device of the node
  static bool attr_done[64][5];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  const int NT = (a.D + 15) / 16;
  const int DP = 16 * NT, LD = DP + 2;
  const size_t lds = (size_t)(3 * DP * LD + 2 * DP + 8 + 2 * DP) * sizeof(double);
  switch (NT) {
    case 1:
      hipLaunchKernelGGL((expm_mfma_kernel<1, 1>), dim3(a.n), dim3(64), lds, stream, a);
      break;
    case 2:
      hipLaunchKernelGGL((expm_mfma_kernel<2, 2>), dim3(a.n), dim3(256), lds, stream, a);
      break;
    case 3:
      if (!attr_done[dev][3]) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(expm_mfma_kernel<3, 1>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done[dev][3] = true;
      }
      hipLaunchKernelGGL((expm_mfma_kernel<3, 1>), dim3(a.n), dim3(192), lds, stream, a);
      break;
    default: {
      static const int mode = getenv("HYPHY_HIP_EXPM") ? atoi(getenv("HYPHY_HIP_EXPM")) : -1;  // 0: r02 kernel; 1/2/4: panels forced
      if (mode == 0) {
        if (!attr_done[dev][4]) {
          hipFuncSetAttribute(reinterpret_cast<const void *>(expm_mfma_kernel<4, 2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
          attr_done[dev][4] = true;
        }
        hipLaunchKernelGGL((expm_mfma_kernel<4, 2>), dim3(a.n), dim3(512), lds, stream, a);
        break;
      }
      // expm64_kernel<H>: H workgroups per matrix while they all fit the chip at once (one workgroup per CU: 100 KiB of LDS)
      static int cus[64];
      if (!cus[dev]) {
        hipDeviceProp_t pr;
        cus[dev] = (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
      }
      int H = 1;
      if (!a.is_prob) {
        if (4 * a.n <= cus[dev]) H = 4;
        else if (2 * a.n <= cus[dev]) H = 2;
      }
      if (mode == 1 || mode == 2 || mode == 4) H = a.is_prob ? 1 : mode;
      const size_t lds64 = (size_t)(3 * 64 * 65 + 16 + 4 * 64 + 2) * sizeof(double);
      static bool attr64[64];
      if (!attr64[dev]) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(expm64_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds64);
        hipFuncSetAttribute(reinterpret_cast<const void *>(expm64_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds64);
        hipFuncSetAttribute(reinterpret_cast<const void *>(expm64_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds64);
        attr64[dev] = true;
      }
      // Coefficients of the fused construction in the kernel-argument block when they fit: `coeffs` is host-mapped
      // memory and is dereferenced HERE, on the host — the caller staged it before this call (hyphy_hip_build_q).
      ExpmArgs b = a;
      static const bool fixed12 = getenv("HYPHY_HIP_EXPM_DEGREE") && atoi(getenv("HYPHY_HIP_EXPM_DEGREE")) == 12;
      b.fixed_degree = fixed12 ? 1 : 0;
      CoefInline ci;
      if (b.templates_pad) fill_coef_inline(b, ci);
      switch (H) {
        case 4: hipLaunchKernelGGL((expm64_kernel<4>), dim3(4 * b.n), dim3(512), lds64, stream, b, ci); break;
        case 2: hipLaunchKernelGGL((expm64_kernel<2>), dim3(2 * b.n), dim3(512), lds64, stream, b, ci); break;
        default: hipLaunchKernelGGL((expm64_kernel<1>), dim3(b.n), dim3(512), lds64, stream, b, ci); break;
      }
      return b.coef_inline != 0;
    }
  }
  return false;
}

void launch_build_q(const double *templates, const double *coeffs, int n, int K, int D, double *Q, hipStream_t stream) {
  if (n <= 0) return;
  hipLaunchKernelGGL(build_q_kernel, dim3(n), dim3(256), (size_t)D * D * sizeof(double), stream, templates, coeffs, K,
                     D, Q);
}

}  // namespace hyhip
