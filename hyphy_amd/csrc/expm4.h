// 4-state matrix exponential: one thread, everything in registers (shared by expm_nuc_kernel and — folded into the pruning
// launch of small shards — prune_nuc2_kernel).  Same contract as the MFMA kernels: scaling by a power of two, degree-12 Taylor
// (Horner), diag_populator before and after the squarings, restart with a 2^7 larger scale when a diagonal exceeds 1, early
// exit from the squarings, sticky status + NaN matrix on failure (matrix.cpp:5537-5951).
#pragma once
#include "common.h"

namespace hyhip {

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
__device__ __forceinline__ void expm4_one(const ExpmArgs &a, int m, double (&R)[16]) {
  double Q[16];
  if (a.templates) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      double d = 0.;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (j == i) continue;
        double v = 0.;
        for (int k = 0; k < a.K; k++) v += a.coeffs[(size_t)m * a.K + k] * a.templates[(size_t)k * 16 + 4 * i + j];
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
      double X[16], T[16], T2[16];
#pragma unroll
      for (int k = 0; k < 16; k++) X[k] = Q[k] * scale;
      // Horner: R = I + X (I + X/2 (I + X/3 (... (I + X/12))))
#pragma unroll
      for (int k = 0; k < 16; k++) T[k] = X[k] * (1.0 / 12.0);
#pragma unroll
      for (int d = 0; d < 4; d++) T[5 * d] += 1.0;
      for (int k = 11; k >= 1; k--) {
        mm4(X, T, T2);
        const double f = 1.0 / (double)k;
#pragma unroll
        for (int e = 0; e < 16; e++) T[e] = T2[e] * f;
#pragma unroll
        for (int d = 0; d < 4; d++) T[5 * d] += 1.0;
      }
#pragma unroll
      for (int k = 0; k < 16; k++) R[k] = T[k];
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
