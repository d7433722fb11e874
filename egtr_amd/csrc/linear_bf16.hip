// Object-query-sized nn.Linear of a bf16 model: y[M, N] = act(alpha (x[M, K] . W[N, K]^T + bias)), bf16 in / out, fp32 accumulation
// on v_mfma_f32_32x32x16_bf16.  (The decoder of the bf16 stress configuration runs 9 such layers per decoder layer on
// 16 x 300 = 4800 rows; the vendor library's pick for these shapes -- a 224 x 192 tile: 44 workgroups -- takes 20 us each,
// profiles/r05_stress_forward_breakdown.txt.)
//
// Both operands of the 32x32x16 product are "8 consecutive k per lane": lane (i = l & 31, half = l >> 5) holds
// A[i][8 half + 0..7] and B[8 half + 0..7][i].  With A = rows of x and B^T = rows of W (nn.Linear layout) each operand of a
// K = 16 step is ONE 16-byte load per lane straight from the row-major tensors -- no staging, no re-layout.  One wave per
// workgroup owns 32 rows x 64 columns (two accumulators): M / 32 x N / 64 workgroups (600 at M = 4800, N = 256), each reading
// its 32 x K slice of x once and a 64 x K slice of W (L2-resident: W is at most 512 KB).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float bf2f(unsigned short u) { return __uint_as_float(((unsigned)u) << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {  // round to nearest even; NaN stays NaN
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

__global__ __launch_bounds__(64) void linear_bf16_rows32(const unsigned short* __restrict__ x, int ldx,
                                                         const unsigned short* __restrict__ w,
                                                         const unsigned short* __restrict__ bias,
                                                         unsigned short* __restrict__ y, int ldy, int M, int N, int K,
                                                         int relu, float alpha) {
  const int lane = threadIdx.x, li = lane & 31, hf = lane >> 5;
  const int r0 = blockIdx.x * 32, n0 = blockIdx.y * 64;
  const bool two = n0 + 32 < N;
  const uint4* xa = reinterpret_cast<const uint4*>(x + (size_t)min(r0 + li, M - 1) * ldx + 8 * hf);
  const uint4* wb0 = reinterpret_cast<const uint4*>(w + (size_t)(n0 + li) * K + 8 * hf);
  const uint4* wb1 = reinterpret_cast<const uint4*>(w + (size_t)(two ? n0 + 32 + li : n0 + li) * K + 8 * hf);
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    acc0[r] = 0.f;
    acc1[r] = 0.f;
  }
  const int steps = K >> 4;   // 16 k per step = two uint4 per row
  // chunks of 4 steps, the operands of chunk c + 1 requested before the products of chunk c (two register sets)
  const int nchunks = steps >> 2;
  uint4 a[2][4], b0[2][4], b1[2][4];
  auto fetch = [&](int set, int c) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[set][u] = xa[2 * (4 * c + u)];
      b0[set][u] = wb0[2 * (4 * c + u)];
      b1[set][u] = wb1[2 * (4 * c + u)];
    }
  };
  auto multiply = [&](int set) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[set][u]), __builtin_bit_cast(bf16x8, b0[set][u]), acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[set][u]), __builtin_bit_cast(bf16x8, b1[set][u]), acc1, 0, 0, 0);
    }
  };
  if (nchunks > 0) fetch(0, 0);
  int c = 0;
  for (; c + 2 <= nchunks; c += 2) {
    fetch(1, c + 1);
    multiply(0);
    if (c + 2 < nchunks) fetch(0, c + 2);
    multiply(1);
  }
  if (c < nchunks) multiply(0);
  for (int s = 4 * nchunks; s < steps; ++s) {
    const uint4 av = xa[2 * s], bv0 = wb0[2 * s], bv1 = wb1[2 * s];
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv0), acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv1), acc1, 0, 0, 0);
  }
  // D[row][col]: lane = column li, register r = row (r & 3) + 8 (r >> 2) + 4 half
  const float bb0 = bias != nullptr ? bf2f(bias[n0 + li]) : 0.f;
  const float bb1 = bias != nullptr && two ? bf2f(bias[n0 + 32 + li]) : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = r0 + (r & 3) + 8 * (r >> 2) + 4 * hf;
    if (row < M) {
      float v0 = (acc0[r] + bb0) * alpha, v1 = (acc1[r] + bb1) * alpha;
      if (relu) {
        v0 = egtr_relu(v0);
        v1 = egtr_relu(v1);
      }
      unsigned short* yr = y + (size_t)row * ldy + n0 + li;
      yr[0] = f2bf(v0);
      if (two) yr[32] = f2bf(v1);
    }
  }
}

}  // namespace

extern "C" int egtr_linear_bf16(egtr_stream_t stream, const uint16_t* x, int ldx, const uint16_t* weight,
                                const uint16_t* bias, uint16_t* y, int ldy, int M, int N, int K, int relu,
                                float alpha) {
  if (!x || !weight || !y) return EGTR_E_ARG;
  if (M <= 0 || N <= 0 || K <= 0 || ldx < K || ldy < N) return EGTR_E_ARG;
  if ((K & 15) || (N & 31) || (ldx & 7)) return EGTR_E_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(weight)) & 15) return EGTR_E_UNSUPPORTED;
  if (M > 65535 * 32) return EGTR_E_UNSUPPORTED;
  const dim3 grid((unsigned)((M + 31) / 32), (unsigned)((N + 63) / 64));
  hipLaunchKernelGGL(linear_bf16_rows32, grid, dim3(64), 0, static_cast<hipStream_t>(stream), x, ldx, weight, bias, y, ldy,
                     M, N, K, relu, alpha);
  return egtr_check_launch();
}
