// How long after a kernel's end does the host see "done"?  Three ways of publishing a sequence word into host-mapped pinned memory
// behind a ~40 us kernel: a one-thread kernel (what the library's reduction kernel does besides summing), hipStreamWriteValue64
// (a stream memory operation: no kernel), and the busy kernel's own last workgroup (atomic arrival counter).
//   hipcc --offload-arch=gfx950 -O3 -o stream_write_probe stream_write_probe.hip && ./stream_write_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHK(e)                                                                          \
  do {                                                                                  \
    hipError_t r_ = (e);                                                                \
    if (r_ != hipSuccess) {                                                             \
      fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_));                           \
      exit(1);                                                                          \
    }                                                                                   \
  } while (0)

__global__ void busy(double *out, int iters, unsigned long long *fuse_flag, unsigned long long seq, int *ctr, int n_wg, double *hpart) {
  double x = threadIdx.x * 1e-3;
  for (int i = 0; i < iters; i++) x = fma(x, 1.0000001, 1e-9);
  if (threadIdx.x == 0) {
    out[blockIdx.x] = x;
    if (hpart) hpart[blockIdx.x] = x;  // (partial sums straight into host memory)
    if (fuse_flag) {
      __threadfence_system();
      if (atomicAdd(ctr, 1) == n_wg - 1) {
        *ctr = 0;
        __hip_atomic_store(fuse_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}
__global__ void publish(unsigned long long *flag, unsigned long long seq) { __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

int main() {
  const int n_wg = 624, iters = 14000, reps = 400;
  double *d_out, *h_part, *d_hpart;
  int *d_ctr;
  unsigned long long *h_flag, *d_flag;
  CHK(hipMalloc(&d_out, n_wg * sizeof(double)));
  CHK(hipMalloc(&d_ctr, sizeof(int)));
  CHK(hipMemset(d_ctr, 0, sizeof(int)));
  CHK(hipHostMalloc(&h_flag, 64));
  CHK(hipHostMalloc(&h_part, n_wg * sizeof(double)));
  CHK(hipHostGetDevicePointer((void **)&d_flag, h_flag, 0));
  CHK(hipHostGetDevicePointer((void **)&d_hpart, h_part, 0));
  hipStream_t st;
  CHK(hipStreamCreate(&st));
  volatile unsigned long long *vf = h_flag;
  const char *names[] = {"publish kernel", "hipStreamWriteValue64", "last workgroup of the busy kernel", "busy kernel + hipStreamSynchronize",
                         "hipStreamWriteValue64, partials in host memory"};
  for (int mode = 0; mode < 5; mode++) {
    std::vector<double> t;
    unsigned long long seq = 1;
    for (int r = 0; r < reps + 20; r++, seq++) {
      auto t0 = std::chrono::steady_clock::now();
      if (mode == 0) {
        hipLaunchKernelGGL(busy, dim3(n_wg), dim3(64), 0, st, d_out, iters, nullptr, seq, d_ctr, n_wg, nullptr);
        hipLaunchKernelGGL(publish, dim3(1), dim3(1), 0, st, d_flag, seq);
      } else if (mode == 1 || mode == 4) {
        hipLaunchKernelGGL(busy, dim3(n_wg), dim3(64), 0, st, d_out, iters, nullptr, seq, d_ctr, n_wg, mode == 4 ? d_hpart : nullptr);
        CHK(hipStreamWriteValue64(st, d_flag, seq, 0));
      } else if (mode == 2) {
        hipLaunchKernelGGL(busy, dim3(n_wg), dim3(64), 0, st, d_out, iters, d_flag, seq, d_ctr, n_wg, nullptr);
      } else {
        hipLaunchKernelGGL(busy, dim3(n_wg), dim3(64), 0, st, d_out, iters, nullptr, seq, d_ctr, n_wg, nullptr);
        CHK(hipStreamSynchronize(st));
      }
      if (mode != 3)
        while (*vf != seq) {}
      auto t1 = std::chrono::steady_clock::now();
      if (r >= 20) t.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
    }
    std::sort(t.begin(), t.end());
    printf("%-52s median %.2f us  p10 %.2f  p90 %.2f (launch -> host sees completion, busy kernel included)\n", names[mode], t[t.size() / 2], t[t.size() / 10],
           t[t.size() * 9 / 10]);
  }
  return 0;
}
