// Microbenchmark: achievable rate of v_mfma_f64_16x16x4_f64 on gfx950 vs waves per SIMD, with the
// real shader clock (s_memtime cycles / wall time), alone and next to FP64 VALU FMAs.
// Pins the "peak" of the FP64 roofline in DESIGN.md — the microarch guide has no f64 row.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int NACC, int VALU>
__global__ __launch_bounds__(256) void k(double *out, long long *cyc, int iters, double a0, double b0) {
  f64x4 acc[NACC > 0 ? NACC : 1];
  for (int i = 0; i < (NACC > 0 ? NACC : 1); i++) acc[i] = (f64x4){0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  double v0 = a, v1 = b, v2 = a * b, v3 = a + b;
  long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < VALU; j++) { v0 = fma(v0, a, b); v1 = fma(v1, a, b); v2 = fma(v2, a, b); v3 = fma(v3, a, b); }
  }
  long long t1 = clock64();
  double s = v0 + v1 + v2 + v3;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678) out[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

// v_mfma_f64_4x4x4_4b_f64: four independent 4x4x4 blocks per instruction (512 flops)
template <int NACC>
__global__ __launch_bounds__(256) void k4(double *out, int iters, double a0, double b0) {
  double acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = 0;
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i];
  if (s == 12345.678) out[0] = s;
}
template <int NACC>
void run4(int wg_per_cu) {
  double *d; hipMalloc(&d, 8);
  int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 grid(256 * wg_per_cu), block(256);
  hipLaunchKernelGGL((k4<NACC>), grid, block, 0, 0, d, 100, 1.0, 1e-3);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k4<NACC>), grid, block, 0, 0, d, iters, 1.0, 1e-3);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double n = (double)grid.x * 4 * iters * NACC;
  printf("mfma_f64_4x4x4_4b %dacc waves/SIMD=%d: %.3f ms  %.1f TF  (%.1f cycles per instruction per SIMD at 2.39 GHz)\n", NACC,
         wg_per_cu, ms, n * 512.0 / ms / 1e9, ms * 1e-3 * 2.39e9 / (n / 1024.0));
  hipFree(d);
}

// the 16x16x4 form with DISTINCT A/B operand registers per instruction (a real kernel's pattern)
template <int NACC>
__global__ __launch_bounds__(256) void kd(double *out, int iters, double a0, double b0) {
  f64x4 acc[NACC];
  double av[NACC], bv[NACC];
  for (int i = 0; i < NACC; i++) {
    acc[i] = (f64x4){0, 0, 0, 0};
    av[i] = a0 + threadIdx.x * 1e-9 + i;
    bv[i] = b0 + i * 1e-3;
  }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[i], acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NACC; i++) { av[i] += 1e-12; bv[i] -= 1e-12; }  // (cheap VALU so the operands really change)
  }
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678) out[0] = s;
}
template <int NACC>
void rund(int wg_per_cu) {
  double *d; hipMalloc(&d, 8);
  int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 grid(256 * wg_per_cu), block(256);
  hipLaunchKernelGGL((kd<NACC>), grid, block, 0, 0, d, 100, 1.0, 1e-3);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((kd<NACC>), grid, block, 0, 0, d, iters, 1.0, 1e-3);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double n = (double)grid.x * 4 * iters * NACC;
  printf("mfma_f64_16x16x4 distinct A/B regs %dacc waves/SIMD=%d: %.3f ms  %.1f TF  (%.1f cycles per instruction per SIMD at 2.39 GHz)\n",
         NACC, wg_per_cu, ms, n * 2048.0 / ms / 1e9, ms * 1e-3 * 2.39e9 / (n / 1024.0));
  hipFree(d);
}

// the 4x4x4 form with DISTINCT A/B operand registers per instruction: NA x NB operand grid, NA*NB accumulators
template <int NA, int NB>
__global__ __launch_bounds__(256) void kd4(double *out, int iters, double a0, double b0) {
  double acc[NA][NB], av[NA], bv[NB];
  for (int i = 0; i < NA; i++) av[i] = a0 + threadIdx.x * 1e-9 + i;
  for (int j = 0; j < NB; j++) bv[j] = b0 + j * 1e-3;
  for (int i = 0; i < NA; i++) for (int j = 0; j < NB; j++) acc[i][j] = 0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NA; i++)
#pragma unroll
      for (int j = 0; j < NB; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[i], bv[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NA; i++) av[i] += 1e-12;
#pragma unroll
    for (int j = 0; j < NB; j++) bv[j] -= 1e-12;
  }
  double s = 0;
  for (int i = 0; i < NA; i++) for (int j = 0; j < NB; j++) s += acc[i][j];
  if (s == 12345.678) out[0] = s;
}
template <int NA, int NB>
void rund4(int wg_per_cu) {
  double *d; hipMalloc(&d, 8);
  int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 grid(256 * wg_per_cu), block(256);
  hipLaunchKernelGGL((kd4<NA, NB>), grid, block, 0, 0, d, 100, 1.0, 1e-3);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((kd4<NA, NB>), grid, block, 0, 0, d, iters, 1.0, 1e-3);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double n = (double)grid.x * 4 * iters * NA * NB;
  printf("mfma_f64_4x4x4_4b %d x %d operand grid waves/SIMD=%d: %.3f ms  %.1f TF  (%.1f cycles per instruction per SIMD at 2.39 GHz)\n",
         NA, NB, wg_per_cu, ms, n * 512.0 / ms / 1e9, ms * 1e-3 * 2.39e9 / (n / 1024.0));
  hipFree(d);
}

template <int NACC, int VALU>
void run(const char *name, int wg_per_cu, int waves) {
  double *d; long long *c; hipMalloc(&d, 8); hipMalloc(&c, 8);
  int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 grid(256 * wg_per_cu), block(64 * waves);
  hipLaunchKernelGGL((k<NACC, VALU>), grid, block, 0, 0, d, c, 100, 1.0, 1e-3);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, VALU>), grid, block, 0, 0, d, c, iters, 1.0, 1e-3);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long hc; hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
  double nwaves = (double)grid.x * waves;
  double mf = nwaves * iters * NACC * 2048.0, vf = nwaves * iters * VALU * 4 * 128.0;
  double ghz = hc / (ms * 1e-3) / 1e9;
  double wps = wg_per_cu * waves / 4.0;
  printf("%-22s waves/SIMD=%.0f: %.3f ms  MFMA %.1f TF  VALU %.1f TF  clock %.2f GHz  real cycles per MFMA per SIMD %.1f\n", name,
         wps, ms, mf / ms / 1e9, vf / ms / 1e9, ghz, (double)hc / (iters * (NACC > 0 ? NACC : 1) * wps));
  hipFree(d); hipFree(c);
}
int main() {
  rund4<2, 4>(1); rund4<2, 4>(2); rund4<2, 4>(4); rund4<4, 4>(1); rund4<4, 4>(2); rund4<1, 8>(1); rund4<8, 1>(1);
  rund<4>(1); rund<4>(2); rund<4>(4); rund<8>(2);
  run4<4>(1); run4<8>(1); run4<8>(2); run4<8>(4); run4<8>(8);
  run<4, 0>("mfma only 4acc", 1, 4);
  run<4, 0>("mfma only 4acc", 2, 4);
  run<4, 0>("mfma only 4acc", 3, 4);
  run<4, 0>("mfma only 4acc", 4, 4);
  run<4, 0>("mfma only 4acc", 6, 4);
  run<4, 0>("mfma only 4acc", 8, 4);
  run<1, 0>("mfma 1acc dependent", 1, 4);
  run<2, 0>("mfma only 2acc", 1, 4);
  run<8, 0>("mfma only 8acc", 1, 4);
  run<8, 0>("mfma only 8acc", 2, 4);
  run<1, 0>("mfma 1acc dependent", 4, 4);
  run<1, 0>("mfma 1acc dependent", 8, 4);
  run<0, 8>("valu fma only", 1, 4);
  run<0, 8>("valu fma only", 4, 4);
  run<4, 8>("mfma+32 vfma", 1, 4);
  run<4, 8>("mfma+32 vfma", 2, 4);
  run<4, 8>("mfma+32 vfma", 4, 4);
  run<4, 16>("mfma+64 vfma", 4, 4);
  return 0;
}
