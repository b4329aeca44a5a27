// Empirical operand layout of v_mfma_f64_4x4x4_4b_f64 on gfx950 (the microarch guide has no table):
// one-hot A (lane la) x one-hot B (lane lb) -> which D lanes light up; also CBSZ/ABID broadcast.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CBSZ, int ABID, int BLGP>
__global__ void probe(int *out) {  // out[la][lb] = bitmask lanes (as 64-bit split in two ints) -> store first lane + count
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; la++)
    for (int lb = 0; lb < 64; lb++) {
      double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
      double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, CBSZ, ABID, BLGP);
      unsigned long long m = __ballot(d != 0.0);
      if (lane == 0) { out[(la * 64 + lb) * 2] = (int)(m & 0xffffffffu); out[(la * 64 + lb) * 2 + 1] = (int)(m >> 32); }
    }
}
template <int CBSZ, int ABID, int BLGP>
void run() {
  int *d; hipMalloc(&d, 64 * 64 * 2 * 4);
  hipLaunchKernelGGL((probe<CBSZ, ABID, BLGP>), dim3(1), dim3(64), 0, 0, d);
  static int h[64 * 64 * 2];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("# cbsz=%d abid=%d blgp=%d : la lb -> D lanes\n", CBSZ, ABID, BLGP);
  for (int la = 0; la < 64; la++)
    for (int lb = 0; lb < 64; lb++) {
      unsigned long long m = (unsigned)h[(la * 64 + lb) * 2] | ((unsigned long long)(unsigned)h[(la * 64 + lb) * 2 + 1] << 32);
      if (!m) continue;
      printf("%d %d :", la, lb);
      for (int l = 0; l < 64; l++) if (m >> l & 1) printf(" %d", l);
      printf("\n");
    }
  hipFree(d);
}
int main() {
  run<0, 0, 0>();
  run<2, 0, 0>();
  run<2, 1, 0>();
  run<1, 1, 0>();
  run<0, 0, 1>();
  run<0, 0, 4>();
  return 0;
}
