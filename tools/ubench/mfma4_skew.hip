// Microbenchmark + correctness probe for the edge product  D[64 x 16 sites] = P[64 x 64] V[64 x 16]
// on v_mfma_f64_4x4x4_4b_f64 in the "skew layout":
//   conditionals: register q = 4Q + c, lane l = j + 4b + 16i holds V[16Q + 4((c + b) & 3) + i][site 4b + j]
//   (block b of the instruction = site quad b; each block walks the k-groups in its own rotated order, so the
//   C/D image of a node IS the B image of the next product and the A operand of instruction (q', q) is
//   block b <- P-subblock((c' + b) & 3, (c + b) & 3) of the 16 x 16 block (Q', Q): a DPP row rotation by 4c' lanes of
//   the loaded register L[m = (c - c') & 3], block b <- subblock(b, (b + m) & 3).  No LDS, no replication in memory.)
// Modes: 0 full stream (loads + DPP + MFMA, results checked), 1 no loads after the first edge, 2 MFMA only,
//        3 the 16x16x4 stream of the production kernel (loads + MFMA), 4 16x16x4 MFMA only.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f64x4 mfma16(double a, double b, f64x4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f64x2 ld16(const double *ubase, unsigned byte_off) {
  return *reinterpret_cast<const f64x2 *>(reinterpret_cast<const char *>(ubase) + byte_off);
}
// block b <- block (b + C) & 3 inside every row of 16 lanes
template <int C>
__device__ __forceinline__ double rotq(double x) {
  if constexpr (C == 0) return x;
  constexpr int ctrl = 0x120 + ((16 - 4 * C) & 15);  // row_ror:n  (dst lane p <- src lane (p - n) & 15)
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_mov_dpp(lo, ctrl, 0xf, 0xf, false);
  hi = __builtin_amdgcn_mov_dpp(hi, ctrl, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

#ifndef PFD
#define PFD 3  // groups of 16 MFMAs the loads run ahead
#endif

// image of one matrix: [G = 4Q + Q' (16)][mp (2)][lane (64)][2] doubles
template <int MODE>
__device__ __forceinline__ void edge4(const double *img, int lane, const double (&B)[16], double (&D)[16]) {
#pragma unroll
  for (int q = 0; q < 16; q++) D[q] = 0.;
  f64x2 L[16][2];
  if (MODE == 0) {
#pragma unroll
    for (int G = 0; G < PFD; G++)
#pragma unroll
      for (int mp = 0; mp < 2; mp++) L[G][mp] = ld16(img, (unsigned)(((G * 2 + mp) * 64 + lane) * 16));
  } else {
#pragma unroll
    for (int G = 0; G < 16; G++)
#pragma unroll
      for (int mp = 0; mp < 2; mp++) L[G][mp] = ld16(img, (unsigned)(((G * 2 + mp) * 64 + lane) * 16));
  }
#pragma unroll
  for (int G = 0; G < 16; G++) {
    const int Q = G >> 2, Qp = G & 3;
    if (MODE == 0 && G + PFD < 16) {
#pragma unroll
      for (int mp = 0; mp < 2; mp++) L[G + PFD][mp] = ld16(img, (unsigned)((((G + PFD) * 2 + mp) * 64 + lane) * 16));
    }
    double R[4][4];  // [c'][m]
#pragma unroll
    for (int m = 0; m < 4; m++) {
      const double x = L[G][m >> 1][m & 1];
      R[0][m] = x;
      if (MODE != 2) {
        R[1][m] = rotq<1>(x);
        R[2][m] = rotq<2>(x);
        R[3][m] = rotq<3>(x);
      } else {  // (distinct operand registers, no DPP: the instruction's own issue rate; results meaningless)
        R[1][m] = L[(G + 1) & 15][m >> 1][m & 1], R[2][m] = L[(G + 2) & 15][m >> 1][m & 1], R[3][m] = L[(G + 3) & 15][m >> 1][m & 1];
      }
    }
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
      for (int cp = 0; cp < 4; cp++) D[4 * Qp + cp] = mfma4(R[cp][(c - cp) & 3], B[4 * Q + c], D[4 * Qp + cp]);
  }
}

// MODE 5: unskewed layout (register q, lane j + 4b + 16i: state 4q + i, site 4b + j), A operand of (q', q) = P[4q' + i][4q + k]
// replicated over the four blocks: a broadcast ds_read_b128 from an LDS image [q'][q / 2][k][i][q & 1] (two q per read)
__device__ __forceinline__ void edge4_lds(const double *lds_img, int lane, const double (&B)[16], double (&D)[16]) {
#pragma unroll
  for (int q = 0; q < 16; q++) D[q] = 0.;
  const unsigned loff = (unsigned)(((lane >> 4) * 4 + (lane & 3)) * 16);
  auto rd = [&](int q2, int qp) -> f64x2 {
    return *reinterpret_cast<const f64x2 *>(reinterpret_cast<const char *>(lds_img) + (qp * 8 + q2) * 256 + loff);
  };
  f64x2 Ac[8], An[8];  // half a q2 step (8 reads, 16 MFMAs) ahead
#pragma unroll
  for (int i = 0; i < 8; i++) Ac[i] = rd(0, i);
#pragma unroll
  for (int s2 = 0; s2 < 16; s2++) {
    const int q2 = s2 >> 1, h = s2 & 1;
    if (s2 + 1 < 16) {
#pragma unroll
      for (int i = 0; i < 8; i++) An[i] = rd((s2 + 1) >> 1, ((s2 + 1) & 1) * 8 + i);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; i++) D[8 * h + i] = mfma4(Ac[i][0], B[2 * q2], D[8 * h + i]);
#pragma unroll
    for (int i = 0; i < 8; i++) D[8 * h + i] = mfma4(Ac[i][1], B[2 * q2 + 1], D[8 * h + i]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; i++) Ac[i] = An[i];
  }
}
template <int OCC>
__global__ __launch_bounds__(256, OCC) void k5(const double *__restrict__ img5, const double *__restrict__ v0,
                                               double *__restrict__ out, long long *__restrict__ cyc, int edges) {
  __shared__ __align__(16) double limg[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) limg[i] = img5[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  double B[16], D[16];
#pragma unroll
  for (int q = 0; q < 16; q++) B[q] = v0[q * 64 + lane];
  const long long t0 = clock64();
  for (int e = 0; e < edges; e++) {
    edge4_lds(limg, lane, B, D);
#pragma unroll
    for (int q = 0; q < 16; q++) B[q] = D[q];
  }
  const long long t1 = clock64();
#pragma unroll
  for (int q = 0; q < 16; q++) out[((size_t)wave * 16 + q) * 64 + lane] = B[q];
  if (lane == 0) cyc[wave] = t1 - t0;
}

// the production kernel's 16x16x4 stream: image [w (4)][k2 (8)][lane][2]
template <int MODE>
__device__ __forceinline__ void edge16(const double *img, int lane, const f64x4 (&B)[4], f64x4 (&D)[4]) {
  constexpr int TILE = 16 * 64;
#pragma unroll
  for (int w = 0; w < 4; w++) D[w] = (f64x4){0., 0., 0., 0.};
  f64x2 Ac[4], An[4];
#pragma unroll
  for (int w = 0; w < 4; w++) Ac[w] = ld16(img, (unsigned)((w * TILE + lane * 2) * 8));
#pragma unroll
  for (int k2 = 0; k2 < 8; k2++) {
    if (k2 + 1 < 8) {
#pragma unroll
      for (int w = 0; w < 4; w++) An[w] = (MODE == 4) ? Ac[w] : ld16(img, (unsigned)((w * TILE + ((k2 + 1) * 64 + lane) * 2) * 8));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int w = 0; w < 4; w++) D[w] = mfma16(Ac[w][0], B[k2 >> 1][(k2 & 1) * 2], D[w]);
#pragma unroll
    for (int w = 0; w < 4; w++) D[w] = mfma16(Ac[w][1], B[k2 >> 1][(k2 & 1) * 2 + 1], D[w]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int w = 0; w < 4; w++) Ac[w] = An[w];
  }
}

template <int MODE, int OCC>
__global__ __launch_bounds__(64, OCC) void k(const double *__restrict__ imgs, int n_img, const double *__restrict__ v0,
                                            double *__restrict__ out, long long *__restrict__ cyc, int edges) {
  const int lane = threadIdx.x;
  const int wave = blockIdx.x;
  long long t0 = 0;
  if (MODE < 3) {
    double B[16], D[16];
#pragma unroll
    for (int q = 0; q < 16; q++) B[q] = v0[q * 64 + lane];
    t0 = clock64();
    for (int e = 0; e < edges; e++) {
      const int br = (int)(((unsigned)e * 7u + (unsigned)wave * 13u) % (unsigned)n_img);
      edge4<MODE>(imgs + (size_t)br * 4096, lane, B, D);
#pragma unroll
      for (int q = 0; q < 16; q++) B[q] = D[q];
    }
    const long long t1 = clock64();
#pragma unroll
    for (int q = 0; q < 16; q++) out[((size_t)wave * 16 + q) * 64 + lane] = B[q];
    if (lane == 0) cyc[wave] = t1 - t0;
  } else {
    f64x4 B[4], D[4];
#pragma unroll
    for (int w = 0; w < 4; w++)
#pragma unroll
      for (int r = 0; r < 4; r++) B[w][r] = v0[(4 * w + r) * 64 + lane];
    t0 = clock64();
    for (int e = 0; e < edges; e++) {
      const int br = (int)(((unsigned)e * 7u + (unsigned)wave * 13u) % (unsigned)n_img);
      edge16<MODE>(imgs + (size_t)br * 4096, lane, B, D);
#pragma unroll
      for (int w = 0; w < 4; w++) B[w] = D[w];
    }
    const long long t1 = clock64();
#pragma unroll
    for (int w = 0; w < 4; w++)
#pragma unroll
      for (int r = 0; r < 4; r++) out[((size_t)wave * 16 + 4 * w + r) * 64 + lane] = B[w][r];
    if (lane == 0) cyc[wave] = t1 - t0;
  }
}

// r01's "instruction ceiling" kernel (tools/ubench_mfma_f64.hip: one operand pair, NACC in-place accumulators), compiled once
// as it was (__launch_bounds__(256): the compiler puts the accumulators in AGPRs) and once with two waves per SIMD declared
// (VGPR accumulators, the form every pruning kernel uses)
template <int NACC, bool AG>
__device__ __forceinline__ void kold_body(double *out, int iters, double a0, double b0) {
  f64x4 acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = (f64x4){0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = mfma16(a, b, acc[i]);
  }
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678) out[0] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void kold_agpr(double *out, int iters, double a0, double b0) { kold_body<NACC, true>(out, iters, a0, b0); }
template <int NACC>
__global__ __launch_bounds__(256, 2) void kold_vgpr(double *out, int iters, double a0, double b0) { kold_body<NACC, false>(out, iters, a0, b0); }
template <int NACC, bool AG>
static void run_old(int wg_per_cu) {
  double *d;
  hipMalloc(&d, 8);
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  dim3 grid(256 * wg_per_cu), block(256);
  auto go = [&](int it) {
    if (AG) hipLaunchKernelGGL((kold_agpr<NACC>), grid, block, 0, 0, d, it, 1.0, 1e-3);
    else hipLaunchKernelGGL((kold_vgpr<NACC>), grid, block, 0, 0, d, it, 1.0, 1e-3);
  };
  go(100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  go(iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)grid.x * 4 * iters * NACC;
  printf("r01 kernel, %s accumulators, %dacc, waves/SIMD=%d: %.3f ms  %.1f TF\n", AG ? "AGPR" : "VGPR", NACC, wg_per_cu, ms, n * 2048.0 / ms / 1e9);
  hipFree(d);
}

static int skew_state(int q, int lane) {
  const int b = (lane >> 2) & 3, i = lane >> 4;
  return 16 * (q >> 2) + 4 * (((q & 3) + b) & 3) + i;
}

template <int MODE, int OCC>
static void run(const char *name, int waves, int edges, const double *dimg, int n_img, const double *dv0, const std::vector<double> &P,
                const std::vector<double> &v0h) {
  double *dout;
  long long *dcyc;
  hipMalloc(&dout, (size_t)waves * 1024 * 8);
  hipMalloc(&dcyc, (size_t)waves * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto launch = [&](int ed) {
    if constexpr (MODE == 5) hipLaunchKernelGGL((k5<OCC>), dim3(waves / 4), dim3(256), 0, 0, dimg, dv0, dout, dcyc, ed);
    else hipLaunchKernelGGL((k<MODE, OCC>), dim3(waves), dim3(64), 0, 0, dimg, n_img, dv0, dout, dcyc, ed);
  };
  launch(4);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < (edges > 1000 ? 2 : 3); rep++) {
    hipEventRecord(e0);
    launch(edges);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  std::vector<long long> cyc(waves);
  hipMemcpy(cyc.data(), dcyc, (size_t)waves * 8, hipMemcpyDeviceToHost);
  double mean = 0;
  long long mx = 0;
  for (auto c : cyc) mean += (double)c, mx = c > mx ? c : mx;
  mean /= waves;
  double err = -1;
  if (MODE == 0 || MODE == 3 || MODE == 5) {  // check wave 0 and the last wave against a host product chain
    err = 0;
    std::vector<double> out(1024);
    for (int wv : {0, waves - 1}) {
      hipMemcpy(out.data(), dout + (size_t)wv * 1024, 8192, hipMemcpyDeviceToHost);
      std::vector<double> V(64 * 16), W(64 * 16);
      for (int q = 0; q < 16; q++)
        for (int l = 0; l < 64; l++) {
          const int st = MODE == 0 ? skew_state(q, l) : 4 * q + (l >> 4);  // (modes 3 and 5: the same plain layout)
          V[st * 16 + (l & 15)] = v0h[q * 64 + l];
        }
      for (int e = 0; e < edges; e++) {
        const int br = MODE == 5 ? 0 : (int)(((unsigned)e * 7u + (unsigned)wv * 13u) % (unsigned)n_img);
        const double *Pm = P.data() + (size_t)br * 4096;
        for (int i = 0; i < 64; i++)
          for (int s = 0; s < 16; s++) {
            double a = 0;
            for (int kk = 0; kk < 64; kk++) a += Pm[i * 64 + kk] * V[kk * 16 + s];
            W[i * 16 + s] = a;
          }
        V.swap(W);
      }
      for (int q = 0; q < 16; q++)
        for (int l = 0; l < 64; l++) {
          const int st = MODE == 0 ? skew_state(q, l) : 4 * q + (l >> 4);  // (modes 3 and 5: the same plain layout)
          const double ref = V[st * 16 + (l & 15)], got = out[q * 64 + l];
          const double e = fabs(got - ref) / (fabs(ref) + 1e-300);
          if (e > err) err = e;
        }
    }
  }
  const double flops = (double)waves * edges * 64. * 64 * 16 * 2;
  printf("%-34s waves=%5d edges=%3d: %8.3f ms  %6.2f TF  cycles/edge mean %7.0f max %7.0f  (per MFMA-equivalent-of-512-flop %5.1f)%s",
         name, waves, edges, best, flops / best / 1e9, mean / edges, (double)mx / edges, mean / edges / 256.,
         err >= 0 ? "" : "\n");
  if (err >= 0) printf("  max rel err %.2e\n", err);
  hipFree(dout);
  hipFree(dcyc);
}

int main() {
  const int n_img = 125;
  std::vector<double> P((size_t)n_img * 4096), img4((size_t)n_img * 4096), img16((size_t)n_img * 4096), v0(1024);
  srand(7);
  for (int b = 0; b < n_img; b++) {
    double *Pm = P.data() + (size_t)b * 4096;
    for (int i = 0; i < 64; i++) {
      double s = 0;
      for (int j = 0; j < 64; j++) s += (Pm[i * 64 + j] = (i == j ? 20.0 : 0.0) + rand() / (double)RAND_MAX);
      for (int j = 0; j < 64; j++) Pm[i * 64 + j] /= s;
    }
    // skew image: [G = 4Q + Q'][mp][lane][2]: m = 2mp + h, lane (i, b, k): P[16Q' + 4b + i][16Q + 4((b + m) & 3) + k]
    double *I4 = img4.data() + (size_t)b * 4096;
    for (int G = 0; G < 16; G++)
      for (int mp = 0; mp < 2; mp++)
        for (int l = 0; l < 64; l++)
          for (int h = 0; h < 2; h++) {
            const int Q = G >> 2, Qp = G & 3, m = 2 * mp + h, i = l & 3, bb = (l >> 2) & 3, kk = l >> 4;
            I4[((G * 2 + mp) * 64 + l) * 2 + h] = Pm[(16 * Qp + 4 * bb + i) * 64 + 16 * Q + 4 * ((bb + m) & 3) + kk];
          }
    // 16x16x4 image: [w][k2][lane][2]: row 16w + (lane & 15), col 4(2 k2 + h) + (lane >> 4)
    double *I16 = img16.data() + (size_t)b * 4096;
    for (int w = 0; w < 4; w++)
      for (int k2 = 0; k2 < 8; k2++)
        for (int l = 0; l < 64; l++)
          for (int h = 0; h < 2; h++) I16[((w * 8 + k2) * 64 + l) * 2 + h] = Pm[(16 * w + (l & 15)) * 64 + 4 * (2 * k2 + h) + (l >> 4)];
  }
  for (auto &x : v0) x = rand() / (double)RAND_MAX;
  double *d4, *d16, *dv;
  hipMalloc(&d4, img4.size() * 8);
  hipMalloc(&d16, img16.size() * 8);
  hipMalloc(&dv, 8192);
  hipMemcpy(d4, img4.data(), img4.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(d16, img16.data(), img16.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dv, v0.data(), 8192, hipMemcpyHostToDevice);
  // mode 5 image (branch 0): [q'][q / 2][k][i][q & 1] = P[4q' + i][4q + k]
  std::vector<double> img5(4096);
  for (int qp = 0; qp < 16; qp++)
    for (int q = 0; q < 16; q++)
      for (int kk = 0; kk < 4; kk++)
        for (int i = 0; i < 4; i++) img5[(((qp * 8 + (q >> 1)) * 4 + kk) * 4 + i) * 2 + (q & 1)] = P[(4 * qp + i) * 64 + 4 * q + kk];
  double *d5;
  hipMalloc(&d5, 4096 * 8);
  hipMemcpy(d5, img5.data(), 4096 * 8, hipMemcpyHostToDevice);
  run_old<4, true>(1), run_old<4, false>(1), run_old<4, true>(2), run_old<4, false>(2), run_old<4, true>(4);
  if (getenv("OLD_ONLY")) return 0;
  for (int E : {60, 600, 6000}) {
    for (int waves : {1024, 2048}) {
      run<3, 2>("16x16x4 full (loads+mfma)", waves, E, d16, n_img, dv, P, v0);
      run<4, 2>("16x16x4 mfma only", waves, E, d16, n_img, dv, P, v0);
      run<0, 2>("4x4x4 skew full (loads+dpp+mfma)", waves, E, d4, n_img, dv, P, v0);
      run<2, 2>("4x4x4 mfma only distinct regs", waves, E, d4, n_img, dv, P, v0);
      run<5, 2>("4x4x4 LDS-broadcast A (no stream)", waves, E, d5, n_img, dv, P, v0);
    }
  }
  run<3, 2>("16x16x4 full", 3072, 600, d16, n_img, dv, P, v0);
  run<3, 2>("16x16x4 full", 624, 600, d16, n_img, dv, P, v0);
  return 0;
}
