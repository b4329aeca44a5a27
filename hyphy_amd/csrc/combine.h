// The last arriver's part of a fused final combine — shared by the pruning kernels (prune.hip) and, as embedded source, by the run-time
// generated 4-state kernels (nucgen.hip: the Makefile turns this file into a string constant, so it must stay free of anything a
// bare hiprtc compilation does not have: no includes besides the guard below, only the names declared in front of it).
// Needs: hyhip::f64x2, hyhip::u32x4_t, hyhip::kLogScaler.
#ifndef HYPHY_NUCGEN_EMBEDDED
#pragma once
#include "devutil.h"
#endif

namespace hyhip {
namespace {

// One wave's fixed-order compensated sum over entries [first, last) of the partial sums (all n of them behind the buffer bounds; `first`
// a multiple of 512): every lane a Kahan sum over its entries, then a compensated shuffle tree — lane 0 ends with (sum, comp, c, fl).
__device__ __forceinline__ void combine_range(double *wg_sum, long long *wg_cnt, int *wg_flag, int n, int first, int last, int lane,
                                              double &sum, double &comp, long long &c, int &fl) {
  // (bounds rounded up to whole 16-byte accesses — the arrays are allocated 4 entries longer than any n —, entries >= last masked below)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(wg_sum, 0, ((n + 1) & ~1) * 8, 0x00020000);
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(wg_cnt, 0, ((n + 1) & ~1) * 8, 0x00020000);
  const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(wg_flag, 0, ((n + 3) & ~3) * 4, 0x00020000);
  sum = 0., comp = 0.;
  c = 0;
  fl = 0;
  constexpr int U = 4;
  for (int base = first; base < last; base += 128 * U) {
    u32x4_t vs[U], vc[U];
    u32x4_t vf[U / 2];
#pragma unroll
    for (int j = 0; j < U; j++) {
      vs[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(base + 128 * j + 2 * lane) * 8u, 0, 16);
      vc[j] = __builtin_amdgcn_raw_buffer_load_b128(rc, (unsigned)(base + 128 * j + 2 * lane) * 8u, 0, 16);
    }
#pragma unroll
    for (int j = 0; j < U / 2; j++) vf[j] = __builtin_amdgcn_raw_buffer_load_b128(rf, (unsigned)(base + 256 * j + 4 * lane) * 4u, 0, 16);
#pragma unroll
    for (int j = 0; j < U; j++) {
      f64x2 x;
      long long cc[2];
      __builtin_memcpy(&x, &vs[j], 16);
      __builtin_memcpy(cc, &vc[j], 16);
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const bool in = base + 128 * j + 2 * lane + h < last;
        const double y = (in ? x[h] : 0.) - comp;  // Kahan
        const double t = sum + y;
        comp = (t - sum) - y;
        sum = t;
        c += in ? cc[h] : 0ll;
      }
    }
#pragma unroll
    for (int j = 0; j < U / 2; j++)
#pragma unroll
      for (int h = 0; h < 4; h++)
        if (base + 256 * j + 4 * lane + h < last) fl |= (int)vf[j][h];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double b0 = __shfl_down(sum, off), bc = __shfl_down(comp, off);
    const long long cc = __shfl_down(c, off);
    fl |= __shfl_down(fl, off);
    const double t = sum + b0;
    const double e = (fabs(sum) >= fabs(b0)) ? (sum - t) + b0 : (b0 - t) + sum;  // sum + b0 = t + e exactly
    comp = comp + bc - e;
    sum = t;
    c += cc;
  }
}

// The result record from a finished sum (one lane): log-L, scaler sum, status, then the sequence word the host spins on
__device__ __forceinline__ void combine_publish(double sum, double comp, long long c, int fl, double *red_out, double *red_rec,
                                                const int *red_status, double red_seq) {
  double r = (sum - comp) - kLogScaler * (double)c;
  if (fl & 2) r = NAN;
  else if (fl & 1) r = -INFINITY;
  red_out[0] = r;
  red_rec[0] = (double)c;
  red_rec[1] = red_status ? (double)*red_status : 0.;
  if (red_seq != 0.) {  // host spins on this word instead of waiting for the stream (record complete before it)
    __threadfence_system();
    reinterpret_cast<volatile double *>(red_rec)[2] = red_seq;
  }
}

// The last arriver's part of the fused final combine (one full wave): n partial sums read back with sc1 loads, fixed-order
// compensated sum, result record published like wg_reduce_kernel's.  Shared by the codon kernels' publish_partial and the
// 4-state kernel's epilogue.
__device__ __forceinline__ void combine_partials(double *wg_sum, long long *wg_cnt, int *wg_flag, int n, double *red_out,
                                                 double *red_rec, const int *red_status, double red_seq, int lane) {
  double sum, comp;
  long long c;
  int fl;
  combine_range(wg_sum, wg_cnt, wg_flag, n, 0, n, lane, sum, comp, c, fl);
  if (lane == 0) combine_publish(sum, comp, c, fl, red_out, red_rec, red_status, red_seq);
}

}  // namespace
}  // namespace hyhip
