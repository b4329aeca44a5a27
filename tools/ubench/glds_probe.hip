// Semantics + latency probe for per-lane gathers straight into LDS (global_load_lds_dwordx4, gfx950).
// build: hipcc --offload-arch=gfx950 -O3 -o glds_probe glds_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double f64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// table: [64 codes][4 w][4 g][4 r] doubles (one leaf's gather table); codes: per site lane
__global__ __launch_bounds__(64) void probe(const double *table, const int *codes, double *out, long long *cycles, int mode, int reps) {
  __shared__ __align__(16) double stage[1024];
  const int lane = threadIdx.x, g = lane >> 4, sl = lane & 15;
  const unsigned lds0 = (unsigned)(size_t)stage;
  double acc[16];
  for (int i = 0; i < 16; i++) acc[i] = 1.0;
  long long t0 = wall_clock64();
  long long c0 = clock64();
  for (int rep = 0; rep < reps; rep++) {
    const int c = codes[(rep * 16 + sl) & 1023];
    const double *bl = table + (size_t)(rep & 63) * 4096;
    if (mode == 0) {
#pragma unroll
      for (int w = 0; w < 4; w++) {
        const f64x2 v0 = *reinterpret_cast<const f64x2 *>(bl + (c * 4 + w) * 16 + g * 4);
        const f64x2 v1 = *reinterpret_cast<const f64x2 *>(bl + (c * 4 + w) * 16 + g * 4 + 2);
        acc[4 * w] *= v0[0], acc[4 * w + 1] *= v0[1], acc[4 * w + 2] *= v1[0], acc[4 * w + 3] *= v1[1];
      }
    } else {
#pragma unroll
      for (int w = 0; w < 4; w++) {
        glds16(bl + (c * 4 + w) * 16 + g * 4, __builtin_amdgcn_readfirstlane(lds0 + (2 * w) * 1024));
        glds16(bl + (c * 4 + w) * 16 + g * 4 + 2, __builtin_amdgcn_readfirstlane(lds0 + (2 * w + 1) * 1024));
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int w = 0; w < 4; w++) {
        const f64x2 v0 = *reinterpret_cast<const f64x2 *>(stage + ((2 * w) * 64 + lane) * 2);
        const f64x2 v1 = *reinterpret_cast<const f64x2 *>(stage + ((2 * w + 1) * 64 + lane) * 2);
        acc[4 * w] *= v0[0], acc[4 * w + 1] *= v0[1], acc[4 * w + 2] *= v1[0], acc[4 * w + 3] *= v1[1];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  long long c1 = clock64();
  long long t1 = wall_clock64();
  for (int i = 0; i < 16; i++) out[(size_t)blockIdx.x * 1024 + i * 64 + lane] = acc[i];
  if (lane == 0 && blockIdx.x == 0) cycles[0] = c1 - c0, cycles[1] = t1 - t0;
}

int main(int argc, char **argv) {
  const int runlen = argc > 1 ? atoi(argv[1]) : 1;  // consecutive sites sharing a code (sorted patterns: ~4)
  const int NL = 64;
  std::vector<double> tab((size_t)NL * 4096);
  for (size_t i = 0; i < tab.size(); i++) tab[i] = 1.0 + (double)(i % 977) / 977.0 * 0.001;
  std::vector<int> codes(1024);
  for (int i = 0; i < 1024; i++) codes[i] = ((i / runlen) * 37 + 11) % 61;
  printf("run length %d\n", runlen);
  double *dt, *dout;
  int *dc;
  long long *dcy;
  hipMalloc(&dt, tab.size() * 8);
  hipMalloc(&dc, 4096);
  hipMalloc(&dout, 2048 * 1024 * 8);
  hipMalloc(&dcy, 16);
  hipMemcpy(dt, tab.data(), tab.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dc, codes.data(), 4096, hipMemcpyHostToDevice);
  std::vector<double> r0(1024), r1(1024);
  for (int blocks : {1, 2048}) {
    for (int mode = 0; mode < 2; mode++) {
      for (int it = 0; it < 3; it++) hipLaunchKernelGGL(probe, dim3(blocks), dim3(64), 0, 0, dt, dc, dout, dcy, mode, 64);
      hipDeviceSynchronize();
      long long cy[2];
      hipMemcpy(cy, dcy, 16, hipMemcpyDeviceToHost);
      hipMemcpy(mode ? r1.data() : r0.data(), dout, 8192, hipMemcpyDeviceToHost);
      printf("blocks %4d mode %d (%s): %lld cycles per gather of 8 KB (wall %lld x10ns total)\n", blocks, mode,
             mode ? "global_load_lds" : "registers", cy[0] / 64, cy[1]);
    }
    double md = 0;
    for (int i = 0; i < 1024; i++) md = fmax(md, fabs(r0[i] - r1[i]));
    printf("blocks %4d: max |registers - lds| = %g %s\n", blocks, md, md == 0. ? "OK" : "MISMATCH");
  }
  return 0;
}
