// Device helpers shared by the pruning kernels (prune.hip) and the class-table kernel (repeats.hip): the FP64 MFMA, the
// 2^64 rescale decision, 16-byte accesses on a wave-uniform base (plain and agent scope), the row sum over a wave's four
// 16-lane rows.  gfx950 only.
#pragma once
#include "common.h"

namespace hyhip {
namespace {

__device__ __forceinline__ f64x4 mfma(double a, double b, f64x4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory
// counter (vmcnt(0)), i.e. it would stall every node on the operand prefetch issued for the next
// schedule entry and on the fire-and-forget persist stores; nothing exchanged between the waves of
// a workgroup inside the walks goes through global memory (except OP_GSYNC entries of the pruning kernels).
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// Decide the power-of-2^64 rescale for a site whose conditional vector sums to `tot`
// (__ll_loop_handle_scaling tree_evaluator.cpp:410-525, _computeBoostScaler /
// _computeReductionScaler tree.cpp:160-202).  Returns the exponent change m (true value =
// stored * 2^(-64 m)) and the multiplier in `sc`.
__device__ __forceinline__ int rescale_decision(double tot, double &sc) {
  int m = 0;
  sc = 1.0;
  if (tot < kScalerThreshold && tot > 0.0) {
    do {
      tot *= kScalerUp;
      sc *= kScalerUp;
      m++;
    } while (tot < kScalerThreshold && m < 15);
  } else if (tot > kScalerUp && tot < HUGE_VAL) {
    do {
      tot *= kScalerThreshold;
      sc *= kScalerThreshold;
      m--;
    } while (tot > kScalerUp && m > -15);
  }
  return m;
}

// 16-byte access at (wave-uniform base) + (32-bit per-lane byte offset): lets the compiler keep the
// base in SGPRs and a single VGPR offset instead of a 64-bit per-lane pointer per stream.
__device__ __forceinline__ f64x2 ld16(const double *ubase, unsigned byte_off) {
  return *reinterpret_cast<const f64x2 *>(reinterpret_cast<const char *>(ubase) + byte_off);
}
__device__ __forceinline__ void st16(double *ubase, unsigned byte_off, f64x2 v) {
  *reinterpret_cast<f64x2 *>(reinterpret_cast<char *>(ubase) + byte_off) = v;
}

// The same at AGENT scope (sc1: write-through store / L1-bypassing load) for data handed between
// workgroups inside a launch — valid under any workgroup -> XCD placement (microarch guide, inter-
// workgroup visibility).  Raw buffer instructions carry the cache-policy bits and are tracked by the
// compiler's s_waitcnt insertion (inline asm would not be).
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t agent_rsrc(const double *ubase) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(ubase), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ f64x2 ld16_agent(const double *ubase, unsigned byte_off) {
  const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(agent_rsrc(ubase), byte_off, 0, 16);
  f64x2 r;
  __builtin_memcpy(&r, &v, 16);
  return r;
}
// The A-operand stream of the wave kernel: buffer loads (plain cache policy) so that the address is (4 SGPRs of resource) +
// (ONE per-lane VGPR offset, lane * 16) + (a scalar / immediate offset per chunk).  With global_load the compiler materialised a
// 64-bit per-lane address per row block and advanced it with v_add_co / v_addc per chunk: ~10 VGPRs and ~8 VALU instructions
// (plus their s_nop hazards) per k-step that the matrix pipe's shadow had to absorb.
__device__ __forceinline__ f64x2 ld16_buf(__amdgpu_buffer_rsrc_t rsrc, unsigned lane_off, unsigned uniform_off) {
  const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, uniform_off, 0);
  f64x2 r;
  __builtin_memcpy(&r, &v, 16);
  return r;
}
__device__ __forceinline__ void st16_agent(double *ubase, unsigned byte_off, f64x2 x) {
  u32x4_t v;
  __builtin_memcpy(&v, &x, 16);
  __builtin_amdgcn_raw_buffer_store_b128(v, agent_rsrc(ubase), byte_off, 0, 16);
}

// sum over the four 16-lane rows of a wave (every lane ends up with the same value, same order): x[i] + x[i ^ 16], then
// + x[i ^ 32], on gfx950's row / half swaps (v_permlane16_swap, v_permlane32_swap: pure VALU, no LDS round trip as a
// ds_bpermute shuffle has; bit-identical to the shuffle form, a + b == b + a)
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double row_sum4(double x) {
  unsigned lo = __double2loint(x), hi = __double2hiint(x);
  u32x2_t a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  x = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
  lo = __double2loint(x), hi = __double2hiint(x);
  a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}


}  // namespace
}  // namespace hyhip
