// The tail of a ResNet bottleneck on channels-last bf16 data as ONE kernel (the bf16 model of the stress configuration; frozen
// batch norm folded into the weights):
//     z = relu( bf16( relu(a + shift2) . W3^T ) + shift3 + shortcut )          (bf16 storage, fp32 arithmetic)
// a = the raw output of the 3x3 convolution [M = B H W, K = planes], W3 the 1x1 convolution [N = 4 planes, K], shortcut the block
// input or the downsample branch [M, N]  (reference: model/deformable_detr.py:735-760 -- the timm ResNet-50 backbone with frozen
// batch norm; a bottleneck ends conv3 -> bn3 -> += shortcut -> relu).  The fp32 twin is conv_tail_x6.hip.  Before: an in-place
// shift + ReLU pass over a, a vendor bf16 GEMM, a shift + shortcut + ReLU pass over z: at bs 16 and 800 x 1333 the layer-1 block
// moved 2.6 GB for 1.23 GB of operands and results (33 `bias_act_nhwc_flat8_bf16` launches = 2.3 of the 24.7 ms forward).
// The rounding points are the ones of that composition: relu(a + shift2) rounded to bf16 (the pass wrote it), the product
// rounded to bf16 (the GEMM wrote it), the sum rounded once more.
//
// HBM-bound (layer 1: 1152 bytes per pixel row for 32 K flop): a workgroup of 4 waves owns 64 rows x 256 columns;
//   * the 64 x K panel is shifted, rectified, rounded and parked in LDS as [row][K + 8] bf16 (K = 512: in two halves);
//   * a wave owns 64 columns; its weight fragments come straight from global memory in MFMA operand order (a derived constant
//     of the weights: egtr_amd/ops.py::conv_tail_pack_bf16), three k-steps ahead;
//   * MFMA roles as in the other bf16 kernels (A = weights, i = column; B = activations, j = row): a lane ends up with one row and
//     4 consecutive columns per accumulator quad.  The wave rounds its 32 x 64 tile to bf16 into a private LDS patch and reads
//     it back row-major, 16 bytes = 8 columns per lane: shortcut loads (requested at kernel start) and stores are whole 128-byte
//     lines, eight rows per instruction.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"
#include "x6_common.h"

namespace {
using x6::bf16x8;
using x6::f32x16;
using x6::static_for;

constexpr int kBM = 64, kBN = 256, kNTW = 2, kMT = 2;
constexpr int kPatchPitch = 64 + 8;   // bf16 elements per row of a wave's 32 x 64 output patch (144 bytes: 16-byte aligned rows)

struct TailArgs {
  const unsigned short* a;         // [M, lda] bf16
  const float* a_shift;            // [K] or null
  const unsigned short* w;         // packed fragments [N / 32][K / 16][64 lanes][8] bf16
  const float* bias;               // [N] or null
  const unsigned short* shortcut;  // [M, ldsc] bf16 or null
  unsigned short* y;               // [M, ldy] bf16
  int M, N, lda, ldsc, ldy;
  int relu_in, relu_out;
  int row_major;   // tile order: 1 = the column blocks of a row panel are neighbours, 0 = the row panels of a column block
};

__device__ __forceinline__ float lo_bf(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hi_bf(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {   // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 v;
  v[0] = (__bf16)a;
  v[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ int xcd_tile(int bid, int total) {
  const int q = total >> 3, r = total & 7, x = bid & 7;
  return x * q + min(x, r) + (bid >> 3);
}

template <int KS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KS > 16 ? 2 : 3))) void conv_tail_bf16_kernel(TailArgs A) {
  constexpr int PH = KS > 16 ? 2 : 1;    // K = 512: the panel is built in two halves (34 instead of 67 KiB: three workgroups
  constexpr int KSP = KS / PH;           // per CU instead of one); the accumulators carry over
  constexpr int KP = 16 * KSP;           // panel columns per phase
  constexpr int kPitch = KP + 8;         // bf16 elements per panel row
  constexpr int C8 = KP / 8;             // 16-byte chunks per panel row
  constexpr int NQ = kBM * C8 / 256;     // chunks per thread
  constexpr int PF = KS == 16 ? 2 : 3;   // weight fragments are requested PF k-steps ahead (K = 256: 2, or 8 registers spill)
  constexpr int CH = NQ < 4 ? NQ : 4;    // panel chunks in flight per thread (<= 170 registers: three waves per SIMD)
  static_assert(256 % C8 == 0 && NQ >= 1, "a thread keeps one column group of the panel");
  extern __shared__ __attribute__((aligned(16))) char s_raw[];
  unsigned short* const sA = reinterpret_cast<unsigned short*>(s_raw);                      // [64][kPitch]
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, hf = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned short* const patch = sA + kBM * kPitch + wave * (32 * kPatchPitch);                // [32][kPatchPitch], wave-private

  const int mblocks = (A.M + kBM - 1) / kBM;
  const int tile = xcd_tile(blockIdx.x, gridDim.x);
  // neighbours in the tile order run side by side on one XCD and share its L2: whichever operand is larger is the one
  // that must not be fetched once per partner (the host decides: panel bytes M K against weight bytes N K)
  constexpr int nblocks_unit = kBN;
  const int nblocks = A.N / nblocks_unit;
  const int nb = A.row_major ? tile % nblocks : tile / mblocks;
  const int m0 = (A.row_major ? tile / nblocks : tile - nb * mblocks) * kBM;
  const int nt0 = nb * (kBN / 32) + wave * kNTW;   // first 32-column tile of this wave
  const int col0 = nt0 * 32;

  // Requests in the order the data is needed (a wave's loads return in order): the first chunk of the panel (thread t owns chunk
  // c8 = t % C8 of rows t / C8 + (256 / C8) q), the weight fragments of the first k-steps, then shift3 and the shortcut values.
  const int c8 = tid % C8, r0 = tid / C8;
  uint4 v[CH];
  auto request_panel = [&](int ph, int q0) {
#pragma unroll
    for (int q = 0; q < CH; ++q) {
      const int row = min(m0 + r0 + (256 / C8) * (q0 + q), A.M - 1);
      v[q] = *reinterpret_cast<const uint4*>(A.a + (size_t)row * A.lda + ph * KP + 8 * c8);
    }
  };
  request_panel(0, 0);
  __builtin_amdgcn_sched_barrier(0);

  const unsigned short* const wlane = A.w + ((size_t)nt0 * KS * 64 + lane) * 8;
  bf16x8 w[PF + 1][kNTW];
  auto load_w = [&](int ks, bf16x8 (&dst)[kNTW]) {
#pragma unroll
    for (int t = 0; t < kNTW; ++t)
      dst[t] = *reinterpret_cast<const bf16x8*>(wlane + (size_t)(t * KS + ks) * (64 * 8));
  };
  static_for<PF>([&](auto i_) {
    constexpr int i = decltype(i_)::value;
    load_w(i, w[i]);
  });

  // the shortcut values of the rows this lane will store (row-major read-back of the patch: lane -> row lane / 8 + 8 i of the
  // 32-row tile, columns 8 (lane % 8) .. + 7 of the wave's 64), requested now
  const int prow = lane >> 3, pcg = lane & 7;
  uint4 sc[kMT][4];
  auto request_shortcut = [&]() {
#pragma unroll
    for (int m = 0; m < kMT; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = min(m0 + m * 32 + prow + 8 * i, A.M - 1);
        sc[m][i] = A.shortcut != nullptr
                       ? *reinterpret_cast<const uint4*>(A.shortcut + (size_t)row * A.ldsc + col0 + 8 * pcg)
                       : make_uint4(0u, 0u, 0u, 0u);
      }
  };
  // K <= 64: at kernel start (the product loop is too short to cover them); longer K: behind the first panel chunk, whose
  // registers they take over (<= 170 registers = three waves per SIMD)
  constexpr bool kShortcutFirst = KS <= 4;
  if constexpr (kShortcutFirst) request_shortcut();
  __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise sinks the requests to their first use)

  // the panel: shift, ReLU, round, park
  auto build_panel = [&](int ph) {
    float sh[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (A.a_shift != nullptr) {
      const float4 s0 = *reinterpret_cast<const float4*>(A.a_shift + ph * KP + 8 * c8);
      const float4 s1 = *reinterpret_cast<const float4*>(A.a_shift + ph * KP + 8 * c8 + 4);
      sh[0] = s0.x; sh[1] = s0.y; sh[2] = s0.z; sh[3] = s0.w;
      sh[4] = s1.x; sh[5] = s1.y; sh[6] = s1.z; sh[7] = s1.w;
    }
    const bool relu_in = A.relu_in != 0;
#pragma unroll
    for (int q0 = 0; q0 < NQ; q0 += CH) {
      if (ph > 0 || q0 > 0) request_panel(ph, q0);
#pragma unroll
      for (int q = 0; q < CH; ++q) {
        const unsigned u[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
        unsigned o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float lo = lo_bf(u[k]) + sh[2 * k], hi = hi_bf(u[k]) + sh[2 * k + 1];
          if (relu_in) {
            lo = egtr_relu(lo);
            hi = egtr_relu(hi);
          }
          o[k] = pk_bf16(lo, hi);
        }
        *reinterpret_cast<uint4*>(sA + (r0 + (256 / C8) * (q0 + q)) * kPitch + 8 * c8) = make_uint4(o[0], o[1], o[2], o[3]);
      }
    }
  };

  f32x16 acc[kMT][kNTW];
#pragma unroll
  for (int m = 0; m < kMT; ++m)
#pragma unroll
    for (int t = 0; t < kNTW; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;

  const unsigned short* const pa0 = sA + li * kPitch + 8 * hf;
  bf16x8 a[2][kMT];
  auto read_a = [&](int j, bf16x8 (&dst)[kMT]) {
#pragma unroll
    for (int m = 0; m < kMT; ++m) dst[m] = *reinterpret_cast<const bf16x8*>(pa0 + m * 32 * kPitch + 16 * j);
  };
  float bz[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto load_bias = [&]() {
    if (A.bias != nullptr) {
      const float4 b0 = *reinterpret_cast<const float4*>(A.bias + col0 + 8 * pcg);
      const float4 b1 = *reinterpret_cast<const float4*>(A.bias + col0 + 8 * pcg + 4);
      bz[0] = b0.x; bz[1] = b0.y; bz[2] = b0.z; bz[3] = b0.w;
      bz[4] = b1.x; bz[5] = b1.y; bz[6] = b1.z; bz[7] = b1.w;
    }
  };
  static_for<PH>([&](auto ph_) {
    constexpr int ph = decltype(ph_)::value;
    if constexpr (ph > 0) __syncthreads();   // every wave has read the previous half out of LDS
    build_panel(ph);
    if constexpr (ph == 0 && !kShortcutFirst) request_shortcut();
    if constexpr (ph == PH - 1) load_bias();   // (requested here, not at kernel start: the panel chunks need those registers)
    __syncthreads();
    read_a(0, a[0]);
    static_for<KSP>([&](auto j_) {
      constexpr int j = decltype(j_)::value, ks = ph * KSP + j;
      if constexpr (ks + PF < KS) load_w(ks + PF, w[(ks + PF) % (PF + 1)]);
      if constexpr (j + 1 < KSP) read_a(j + 1, a[(j + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < kMT; ++m)
#pragma unroll
        for (int t = 0; t < kNTW; ++t)
          acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[ks % (PF + 1)][t], a[j & 1][m], acc[m][t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    });
  });

  // epilogue, one 32 x 64 tile at a time: D[i = n][j = m] -- lane l holds row l & 31, accumulator quad q columns 8 q + 4 (l >> 5)
  // .. + 3 of a 32-wide tile.  Rounded to bf16 (the product as the GEMM stored it) into the wave's patch, read back row-major.
  const bool relu_out = A.relu_out != 0;
#pragma unroll
  for (int m = 0; m < kMT; ++m) {
#pragma unroll
    for (int t = 0; t < kNTW; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<uint2*>(patch + li * kPatchPitch + t * 32 + 8 * q + 4 * hf) =
            make_uint2(pk_bf16(acc[m][t][4 * q], acc[m][t][4 * q + 1]), pk_bf16(acc[m][t][4 * q + 2], acc[m][t][4 * q + 3]));
    // (wave-private patch: the LDS unit serves a wave's requests in order; the compiler's lgkmcnt wait orders the read-back)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = m0 + m * 32 + prow + 8 * i;
      const uint4 z = *reinterpret_cast<const uint4*>(patch + (prow + 8 * i) * kPatchPitch + 8 * pcg);
      const unsigned zu[4] = {z.x, z.y, z.z, z.w}, su[4] = {sc[m][i].x, sc[m][i].y, sc[m][i].z, sc[m][i].w};
      unsigned o[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float lo = lo_bf(zu[k]) + bz[2 * k] + lo_bf(su[k]), hi = hi_bf(zu[k]) + bz[2 * k + 1] + hi_bf(su[k]);
        if (relu_out) {
          lo = egtr_relu(lo);
          hi = egtr_relu(hi);
        }
        o[k] = pk_bf16(lo, hi);
      }
      if (row < A.M) *reinterpret_cast<uint4*>(A.y + (size_t)row * A.ldy + col0 + 8 * pcg) = make_uint4(o[0], o[1], o[2], o[3]);
    }
  }
}

template <int KS>
int launch(hipStream_t st, const TailArgs& A) {
  static unsigned long long raised = 0;
  constexpr int lds = (kBM * (16 * (KS > 16 ? KS / 2 : KS) + 8) + 4 * 32 * kPatchPitch) * 2;
  auto kern = conv_tail_bf16_kernel<KS>;
  if (lds > 64 * 1024) {
    const int rc = egtr_raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds, &raised);
    if (rc != EGTR_OK) return rc;
  }
  const long long tiles = (long long)((A.M + kBM - 1) / kBM) * (A.N / kBN);
  if (tiles >= (1ll << 31)) return EGTR_E_UNSUPPORTED;
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), lds, st, A);
  return egtr_check_launch();
}

// W [N, K] bf16 -> MFMA operand fragments [N / 32][K / 16][64 lanes][8]: lane l of fragment (nt, ks) holds
// W[32 nt + (l & 31)][16 ks + 8 (l >> 5) .. + 7]
__global__ __launch_bounds__(256) void conv_tail_pack_bf16(const unsigned short* __restrict__ w, int ldw, int N, int K,
                                                           uint4* __restrict__ packed) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= N * K / 8) return;
  const int l = idx & 63, frag = idx >> 6, KS = K >> 4;
  const int ks = frag % KS, nt = frag / KS;
  packed[idx] = *reinterpret_cast<const uint4*>(w + (size_t)(32 * nt + (l & 31)) * ldw + 16 * ks + 8 * (l >> 5));
}

}  // namespace

extern "C" int egtr_conv1x1_tail_pack_weights_bf16(egtr_stream_t stream, const uint16_t* w, int ldw, int N, int K,
                                                   uint16_t* w_packed) {
  if (!w || !w_packed || N <= 0 || K <= 0 || ldw < K) return EGTR_E_ARG;
  if (N % 32 || K % 16 || (ldw & 7) || (reinterpret_cast<uintptr_t>(w) & 15) || (reinterpret_cast<uintptr_t>(w_packed) & 15) ||
      (long long)N * K > (1ll << 30))
    return EGTR_E_UNSUPPORTED;
  hipLaunchKernelGGL(conv_tail_pack_bf16, dim3((N * K / 8 + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), w, ldw,
                     N, K, reinterpret_cast<uint4*>(w_packed));
  return egtr_check_launch();
}

extern "C" int egtr_conv1x1_tail_bf16(egtr_stream_t stream, const uint16_t* a, int lda, const float* a_shift, int relu_in,
                                      const uint16_t* w_packed, const float* bias, const uint16_t* shortcut, int ld_shortcut,
                                      int relu_out, uint16_t* y, int ldy, int M, int K, int N) {
  if (!a || !w_packed || !y || M <= 0 || K <= 0 || N <= 0 || lda < K || ldy < N || (shortcut && ld_shortcut < N)) return EGTR_E_ARG;
  if ((K != 64 && K != 128 && K != 256 && K != 512) || N % kBN || (lda & 7) || (ldy & 7) || (shortcut && (ld_shortcut & 7)) ||
      (reinterpret_cast<uintptr_t>(a) & 15) || (reinterpret_cast<uintptr_t>(y) & 15) ||
      (reinterpret_cast<uintptr_t>(w_packed) & 15) || (reinterpret_cast<uintptr_t>(a_shift) & 15) ||
      (reinterpret_cast<uintptr_t>(bias) & 15) || (reinterpret_cast<uintptr_t>(shortcut) & 15))
    return EGTR_E_UNSUPPORTED;
  TailArgs A{a, a_shift, w_packed, bias, shortcut, y, M, N, lda, ld_shortcut, ldy, relu_in, relu_out,
             (long long)M * K * 2 >= (long long)N * K * 2 ? 1 : 0};
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (K) {
    case 64: return launch<4>(st, A);
    case 128: return launch<8>(st, A);
    case 256: return launch<16>(st, A);
    default: return launch<32>(st, A);
  }
}
