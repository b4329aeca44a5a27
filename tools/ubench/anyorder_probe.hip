// Probe (r03): does a kernel launched with hipExtAnyOrderLaunch start while the previous kernel of the SAME stream is
// still running on gfx950?  (hip_ext.h says the flag is "not supported on AMD GFX9xx boards" for the module-launch form.)
// A: one wave spinning for ~50 us; B: one wave stamping its start.  Control: B launched normally; B on a second stream.
// build: hipcc --offload-arch=gfx950 -O2 -o anyorder_probe anyorder_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>

__global__ void spin_kernel(long long *out, long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
  if (threadIdx.x == 0) {
    out[0] = t0;
    out[1] = wall_clock64();
  }
}
__global__ void stamp_kernel(long long *out) {
  if (threadIdx.x == 0) out[0] = wall_clock64();
}

#define CK(x)                                                              \
  do {                                                                     \
    hipError_t e_ = (x);                                                   \
    if (e_ != hipSuccess) {                                                \
      printf("%s -> %s\n", #x, hipGetErrorString(e_));                    \
      return 1;                                                            \
    }                                                                      \
  } while (0)

int main() {
  long long *d;
  CK(hipMalloc((void **)&d, 64));
  hipStream_t s, s2;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  long long h[4];
  for (int mode = 0; mode < 3; mode++)
    for (int rep = 0; rep < 3; rep++) {
      CK(hipMemset(d, 0, 64));
      CK(hipDeviceSynchronize());
      hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, d, 5000ll);  // 50 us at 100 MHz
      if (mode == 0) hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s, d + 2);
      else if (mode == 1) hipExtLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, d + 2);
      else hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s2, d + 2);
      CK(hipDeviceSynchronize());
      CK(hipGetLastError());
      CK(hipMemcpy(h, d, 32, hipMemcpyDeviceToHost));
      printf("%s: A ran %.1f us; B started %.1f us after A's start (%s)\n",
             mode == 0 ? "same stream, ordinary launch " : (mode == 1 ? "same stream, hipExtAnyOrderLaunch" : "second stream, ordinary launch"),
             (h[1] - h[0]) * 0.01, (h[2] - h[0]) * 0.01, h[2] < h[1] ? "OVERLAPPED" : "after A's end");
    }
  return 0;
}
