// Device helpers shared by the split-bf16 ("x6") matrix kernels: gemm_x6.hip, ffn_x6.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace x6 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) char lds_char;

// one LDS-DMA wave-instruction: 64 lanes x 16 bytes from per-lane global addresses to LDS [lds_dst, lds_dst + 1 KiB)
// (lds_dst wave-uniform; M0 is compiler-reserved: saved and restored inside the statement)
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}
// The same with a wave-uniform 64-bit base in SGPRs and a 32-bit per-lane byte offset: the per-instruction address
// arithmetic is scalar (the base) and the VGPR offset can be one loop-invariant register (16 * lane).
__device__ __forceinline__ void dma16s(const void* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}
// lgkmcnt(0) as the BUILTIN (vmcnt 63, expcnt 7 untouched): the compiler's own wait-count bookkeeping sees it, so it does
// not re-wait for the operand reads of the previous step in front of the MFMAs
__device__ __forceinline__ void wait_lgkm0() { __builtin_amdgcn_s_waitcnt(0xc07f); }

template <int N, int I = 0, class Fn>
__device__ __forceinline__ void static_for(Fn&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, I + 1>(f);
  }
}

__device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// The six leading cross terms of (w_hi + w_mid + w_lo) (a_hi + a_mid + a_lo), small terms first.  MFMA roles: A operand =
// weight piece (i = n), B operand = activation piece (j = m): a lane ends up with one row m and 4 CONSECUTIVE columns n per
// accumulator quad.
__device__ __forceinline__ f32x16 mfma6(const bf16x8 (&w)[3], const bf16x8 (&a)[3], f32x16 c) {
  c = mfma(w[2], a[0], c);
  c = mfma(w[0], a[2], c);
  c = mfma(w[1], a[1], c);
  c = mfma(w[1], a[0], c);
  c = mfma(w[0], a[1], c);
  c = mfma(w[0], a[0], c);
  return c;
}

}  // namespace x6
