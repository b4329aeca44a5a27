// 4-state matrix exponential: one thread, everything in registers (shared by expm_nuc_kernel and — folded into the pruning
// launch of small shards — prune_nuc2_kernel).  Same contract as the MFMA kernels: scaling by a power of two, Taylor polynomial of
// degree 12 / 9 / 6 by the scaled norm (Paterson-Stockmeyer), diag_populator before and after the squarings, restart with a 2^7 larger scale when a diagonal exceeds 1, early
// exit from the squarings, sticky status + NaN matrix on failure (matrix.cpp:5537-5951).
// (also compiled as embedded source by the run-time generated kernels of nucgen.hip — the Makefile turns this file into a string
//  constant: nothing in it may need more than the names hyhip::ExpmArgs gives it there)
#ifndef HYPHY_NUCGEN_EMBEDDED
#pragma once
#include "common.h"
#endif

namespace hyhip {

static __constant__ double kInvFact4[13] = {1.0, 1.0, 1.0 / 2, 1.0 / 6, 1.0 / 24, 1.0 / 120, 1.0 / 720, 1.0 / 5040, 1.0 / 40320, 1.0 / 362880,
                                            1.0 / 3628800, 1.0 / 39916800, 1.0 / 479001600};

__device__ __forceinline__ void mm4(const double *A, const double *B, double *C) {
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      double s = 0.;
#pragma unroll
      for (int k = 0; k < 4; k++) s = fma(A[4 * i + k], B[4 * k + j], s);
      C[4 * i + j] = s;
    }
}

__device__ __forceinline__ bool diag_fix4(double *R) {
  bool ok = true;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const double s = (R[4 * i] + R[4 * i + 1]) + (R[4 * i + 2] + R[4 * i + 3]);
    if (s != s || R[5 * i] > 1.) ok = false;
    R[5 * i] += 1. - s;
  }
  return ok;
}

// exp of matrix m of the batch `a` (rate matrix given, built from templates, or — is_prob — a transition matrix passed through)
// `coef`: the [n][K] coefficients of the fused construction (the ring slot, or the copy in the kernel-argument block)
__device__ __forceinline__ void expm4_one(const ExpmArgs &a, int m, double (&R)[16], const double *coef) {
  double Q[16];
  if (a.templates) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      double d = 0.;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (j == i) continue;
        double v = 0.;
        for (int k = 0; k < a.K; k++) v += coef[(size_t)m * a.K + k] * a.templates[(size_t)k * 16 + 4 * i + j];
        Q[4 * i + j] = v;
        d -= v;
      }
      Q[5 * i] = d;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 16; k++) Q[k] = a.Q[(size_t)m * 16 + k];
  }
  if (a.is_prob) {
#pragma unroll
    for (int k = 0; k < 16; k++) R[k] = Q[k];
  } else {
    double rmax = 0., cmax = 0.;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      double rs = 0., cs = 0.;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        rs += fabs(Q[4 * i + j]);
        cs += fabs(Q[4 * j + i]);
      }
      rmax = fmax(rmax, rs);
      cmax = fmax(cmax, cs);
    }
    const double mnorm = rmax * cmax;
    int p = 0;
    if (mnorm > 0.) {
      const double s = 4. * sqrt(mnorm);
      if (s > 1.) p = ilogb(s) + 1;
    }
    bool done = false, failed = !(mnorm < 1e300);
    for (int attempt = 0; attempt < 48 && !done && !failed; attempt++) {
      const double scale = ldexp(1.0, -p);
      double X[16], T[16], X2[16], X3[16];
#pragma unroll
      for (int k = 0; k < 16; k++) X[k] = Q[k] * scale;
      // Paterson-Stockmeyer in X^3, degree 3 nb chosen from the scaled norm like the 64-state kernel (expm.hip: taylor_blocks;
      // sqrt(||X||_1 ||X||_inf) bounds the 2-norm, so the same remainder bound holds): 5 / 4 / 3 products of 4 x 4 instead of the
      // 11 of a plain Horner scheme (late r03: one thread's 704 dependent multiply-adds were 2.3 us of a 7.8 us launch).
      const double sn = sqrt(mnorm) * scale;
      const int nb = sn <= 0.015625 ? 2 : (sn <= 0.11 ? 3 : 4);
      mm4(X, X, X2);
      mm4(X2, X, X3);
      const double *kF = kInvFact4;  // (a __constant__ table: a local array indexed by nb would live in scratch memory)
#pragma unroll
      for (int e = 0; e < 16; e++) R[e] = kF[3 * nb - 2] * X[e] + kF[3 * nb - 1] * X2[e] + kF[3 * nb] * X3[e];
#pragma unroll
      for (int d = 0; d < 4; d++) R[5 * d] += kF[3 * nb - 3];
      for (int blk = nb - 2; blk >= 0; blk--) {
        mm4(R, X3, T);
#pragma unroll
        for (int e = 0; e < 16; e++) R[e] = T[e] + kF[3 * blk + 1] * X[e] + kF[3 * blk + 2] * X2[e];
#pragma unroll
        for (int d = 0; d < 4; d++) R[5 * d] += kF[3 * blk];
      }
      if (!diag_fix4(R)) {
        p += 7;
        if (p > 900) failed = true;
        continue;
      }
      double last_diff = 0.;
      for (int s = 0; s < p; s++) {
        mm4(R, R, T);
        double diff = 0.;
#pragma unroll
        for (int k = 0; k < 16; k++) {
          diff = fmax(diff, fabs(T[k] - R[k]));
          R[k] = T[k];
        }
        if (diff < 2.220446049250313e-16 * 1.e3 || (s >= 10 && diff > last_diff * 100.)) break;
        last_diff = diff;
      }
      if (p > 0 && !diag_fix4(R)) {
        p += 7;
        if (p > 900) failed = true;
        continue;
      }
      done = true;
    }
    if (!done) {  // (as in the MFMA kernel: sticky status + NaN matrix, so that this evaluation's log-L is NaN)
      atomicOr(a.status, 1);
#pragma unroll
      for (int k = 0; k < 16; k++) R[k] = NAN;
    }
  }
}

}  // namespace hyhip
