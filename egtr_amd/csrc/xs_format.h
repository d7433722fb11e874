// The "XS" (split, tiled) operand format of the bf16x6 matrix kernels (gemm_x6.hip) and the helpers every PRODUCER of such
// an operand shares (LayerNorm, the MSDA epilogue, the GEMM epilogue, the stand-alone split pass).
//
// A logical fp32 matrix X[rows][K] (K % 16 == 0) is stored as its exact three-way bf16 split x = hi + mid + lo, cut into
// FRAGMENTS of 32 rows x 16 k of ONE piece = 1 KiB:
//     fragment (rb = row / 32, ks = k / 16, piece p) at byte ((rb * (K / 16) + ks) * 3 + p) * 1024
//     element (r = row % 32, kk = k % 16) inside it at byte (kk / 8) * 512 + r * 16 + (kk % 8) * 2
// i.e. lane l of a wave owns bytes [16 l, 16 l + 16) of a fragment = row l & 31, k-group l >> 5 -- exactly the A / B
// operand of v_mfma_f32_32x32x16_bf16 (8 consecutive k of one row per lane).  One fragment is therefore
//   * what ONE global_load_lds_dwordx4 wave-instruction moves (LDS destination = base + 16 * lane: lane-linear), and
//   * what ONE ds_read_b128 wave-instruction reads back, conflict-free, with no swizzle and no address arithmetic.
// Rows beyond the matrix in the last block are never read into stored results (an output row depends on its own operand
// row only), so producers need not clear them.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace xs {

constexpr int kFragBytes = 1024;

__host__ __device__ inline long long buffer_bytes(long long rows, int K) {
  return ((rows + 31) / 32) * (long long)(K / 16) * 3 * kFragBytes;
}

// byte offset of the 8-byte group holding elements k .. k + 3 (k % 4 == 0) of `row`, piece 0; pieces are 1 KiB apart
__device__ __forceinline__ size_t group_offset(int row, int k, int KS) {
  const int rb = row >> 5, r = row & 31, ks = k >> 4, kk = k & 15;
  return ((size_t)rb * KS + ks) * (3 * kFragBytes) + (kk >> 3) * 512 + r * 16 + (kk & 7) * 2;
}

struct Split3 {
  unsigned hi, mid, lo;   // fp32 bit patterns with zero low halves (the upper 16 bits are the bf16 piece)
};

// Exact truncation split of an ACTIVATION: hi = upper 16 bits of x, mid = upper 16 bits of the exact residual x - hi,
// lo = (x - hi) - mid, which has <= 8 significant bits and is a bf16 as it stands.  Non-finite x: hi carries the inf /
// a quiet NaN alone and mid = lo = 0 (x - hi would be inf - inf = NaN, and a NaN with payload in the low half would
// truncate to inf).
__device__ __forceinline__ Split3 split3(float x) {
  Split3 s;
  const unsigned u = __float_as_uint(x);
  const bool fin = (u & 0x7f800000u) != 0x7f800000u;
  s.hi = u & 0xffff0000u;
  if ((u & 0x7fffffffu) > 0x7f800000u) s.hi = 0x7fc00000u;
  const float r = fin ? x - __uint_as_float(s.hi) : 0.f;
  s.mid = __float_as_uint(r) & 0xffff0000u;
  s.lo = __float_as_uint(r - __uint_as_float(s.mid));
  return s;
}

// The same without the non-finite special cases (4 VALU operations instead of 10): for a non-finite x the lower pieces come
// out NaN (inf - inf), so the products of its row are NaN instead of +-inf / NaN -- still non-finite, still only that row.
// Used where the split sits on a kernel's critical path (the hidden activations of the fused FFN).
__device__ __forceinline__ Split3 split3_fast(float x) {
  Split3 s;
  s.hi = __float_as_uint(x) & 0xffff0000u;
  const float r = x - __uint_as_float(s.hi);
  s.mid = __float_as_uint(r) & 0xffff0000u;
  s.lo = __float_as_uint(r - __uint_as_float(s.mid));
  return s;
}

// round-to-nearest-even split (weights: prepared once, the residuals stay exact in fp32)
__device__ __forceinline__ unsigned bf16_rne_bits(float x) {   // result in the UPPER half, low half zero
  const unsigned u = __float_as_uint(x);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc00000u;
  if ((u & 0x7f800000u) == 0x7f800000u) return u;
  return (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
}
__device__ __forceinline__ Split3 split3_rne(float x) {
  Split3 s;
  const bool fin = (__float_as_uint(x) & 0x7f800000u) != 0x7f800000u;
  s.hi = bf16_rne_bits(x);
  const float r1 = fin ? x - __uint_as_float(s.hi) : 0.f;
  s.mid = bf16_rne_bits(r1);
  s.lo = bf16_rne_bits(r1 - __uint_as_float(s.mid));
  return s;
}

// (a, b) -> one dword holding the upper halves: a in the low 16 bits (the element at the lower address)
__device__ __forceinline__ unsigned pack_hi16(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// Store 4 consecutive elements (k % 4 == 0) of one row into the three pieces: 3 x 8 bytes.
template <bool RNE = false>
__device__ __forceinline__ void store4(char* base, size_t off, float v0, float v1, float v2, float v3) {
  const Split3 s0 = RNE ? split3_rne(v0) : split3(v0), s1 = RNE ? split3_rne(v1) : split3(v1);
  const Split3 s2 = RNE ? split3_rne(v2) : split3(v2), s3 = RNE ? split3_rne(v3) : split3(v3);
  *reinterpret_cast<uint2*>(base + off) = make_uint2(pack_hi16(s0.hi, s1.hi), pack_hi16(s2.hi, s3.hi));
  *reinterpret_cast<uint2*>(base + off + kFragBytes) = make_uint2(pack_hi16(s0.mid, s1.mid), pack_hi16(s2.mid, s3.mid));
  *reinterpret_cast<uint2*>(base + off + 2 * kFragBytes) = make_uint2(pack_hi16(s0.lo, s1.lo), pack_hi16(s2.lo, s3.lo));
}

}  // namespace xs
