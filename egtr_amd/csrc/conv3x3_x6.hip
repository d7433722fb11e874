// 3x3 convolution (stride 1, padding 1, no bias) on channels-last fp32 data with fp32-level accuracy on the bf16 matrix cores:
// the middle convolution of a ResNet bottleneck in inference (reference: model/deformable_detr.py:735-760, the timm ResNet-50
// backbone; frozen batch norm folded into the weights, its shift + ReLU applied by the consumer, conv_tail_x6.hip).
//     y[b, h, w, n] = sum_{dy, dx, c} x[b, h + dy - 1, w + dx - 1, c] * W[n, c, dy, dx]
// MIOpen serves these with fp32-MFMA implicit-GEMM kernels at 65-75 TFLOP/s (37-42 us per convolution at 600 x 1000, bs 1).
// Here: the six-term split-bf16 product of the x6 kernels (x6_common.h) as an implicit GEMM whose activation operand never leaves
// the CU once loaded:
//   * a workgroup (4 waves) owns TH x TW output pixels x all BN output channels of its column block.  The (TH + 2) x (TW + 2)
//     input halo tile is loaded once, split into its three bf16 pieces and parked in LDS as [piece][halo pixel][C + 8]; pixels
//     outside the image are zeros (the padding);
//   * the product loop walks the 9 taps x C / 16 k-steps: the activation fragment of a tap is the SAME LDS tile read at a shifted
//     pixel -- per lane one base address, tap / k-step / piece are immediate offsets; the weight fragments stream from global
//     memory in the XS operand format of the [N, 9 C] matrix W[n][dy][dx][c] (egtr_xs_split_f32), straight into registers,
//     PF k-steps ahead;
//   * MFMA roles as in conv_tail_x6 (A = activation pixels, B = weight columns): a lane holds one output channel and a half wave
//     128 consecutive bytes of a pixel: whole-line stores.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"
#include "x6_common.h"
#include "xs_format.h"

namespace {
using namespace x6;
typedef float f32x4v __attribute__((ext_vector_type(4)));

struct ConvArgs {
  const float* x;   // [B, H, W, C] channels-last
  const char* w;    // XS(Wm [N, 9 C]), Wm[n][(dy * 3 + dx) * C + c]
  float* y;         // [B, H, W, N]
  int B, H, W, N;
  int tiles_x, tiles_y;
};

#ifdef EGTR_CONV_TIMING
// debugging aid (tools/conv3x3_timing.py): per workgroup of the LAST launch, the real-time counter (100 MHz) at entry and exit and
// the shader-clock stamps of wave 0 between the phases
constexpr int kRecWg = 4096;
__device__ unsigned long long g_conv_rec[kRecWg][8];
#define CONV_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#else
#define CONV_T(v) do { } while (0)
#endif

// TH x TW output pixels per workgroup; an MFMA row tile is 4 rows x 8 pixels (TW == 8), MT = TH / 4 of them, split over WM
// wave rows; WN = 4 / WM waves side by side over the 32-column tiles.
template <int C, int TH, int WM, int NTW, int NWV = 4>
__global__ __launch_bounds__(64 * NWV) void conv3x3_x6_kernel(ConvArgs A) {
  constexpr int TW = 8;
  constexpr int HW_ = TW + 2, HH = TH + 2, HP = HH * HW_;   // halo tile
  constexpr int KC = C / 16;                                // k-steps per tap
  constexpr int KS = 9 * KC;
  constexpr int kPitch = C + 8;                             // bf16 elements per halo pixel in LDS
  constexpr int MTW = (TH / 4) / WM;                        // row tiles per wave
  constexpr int WN = NWV / WM;
  constexpr int NT = 64 * NWV;                              // threads
  constexpr int BN = 32 * NTW * WN;                         // output channels per workgroup
#ifndef EGTR_CONV_PF
#define EGTR_CONV_PF 3
#endif
  constexpr int PF = EGTR_CONV_PF;
  constexpr int C4 = C / 4;
  static_assert(TH % (4 * WM) == 0 && NWV % WM == 0, "tile shape");
  extern __shared__ __attribute__((aligned(16))) char s_raw[];
  __bf16* const sA = reinterpret_cast<__bf16*>(s_raw);   // [3][HP][kPitch]

  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, hf = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
#ifdef EGTR_CONV_TIMING
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  CONV_T(t0);
  const int nblocks = A.N / BN;
  int t = blockIdx.x;
  const int nb = t % nblocks;
  t /= nblocks;
  const int tx = t % A.tiles_x;
  t /= A.tiles_x;
  const int ty = t % A.tiles_y, b = t / A.tiles_y;
  const int y0 = ty * TH, x0 = tx * TW;
  const int nt0 = nb * (BN / 32) + wn * NTW;

  // weight fragments of the first k-steps
  const char* const wlane = A.w + (size_t)nt0 * KS * (3 * xs::kFragBytes) + lane * 16;
  bf16x8 w[PF + 1][NTW][3];
  auto load_w = [&](int ks, bf16x8 (&dst)[NTW][3]) {
#pragma unroll
    for (int tt = 0; tt < NTW; ++tt)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        dst[tt][p] = *reinterpret_cast<const bf16x8*>(wlane + ((size_t)(tt * KS + ks) * 3 + p) * xs::kFragBytes);
  };

  // the halo tile: chunk idx = halo pixel * C4 + c4; pixels outside the image are zeros
  {
    constexpr int CHUNKS = HP * C4;
    constexpr int NQ = (CHUNKS + NT - 1) / NT;
    constexpr int CH = 6;
    const float* const xb = A.x + (size_t)b * A.H * A.W * C;
#pragma unroll
    for (int q0 = 0; q0 < NQ; q0 += CH) {
      f32x4v v[CH];
#pragma unroll
      for (int q = 0; q < CH; ++q) {
        const int idx = tid + NT * (q0 + q);
        const int hp = idx / C4, c4 = idx % C4;
        const int gy = y0 - 1 + hp / HW_, gx = x0 - 1 + hp % HW_;
        const bool in = (q0 + q < NQ) && idx < CHUNKS && gy >= 0 && gy < A.H && gx >= 0 && gx < A.W;
        v[q] = f32x4v{0.f, 0.f, 0.f, 0.f};
        if (in) v[q] = *reinterpret_cast<const f32x4v*>(xb + ((size_t)gy * A.W + gx) * C + 4 * c4);
      }
      if (q0 == 0) {
        static_for<PF>([&](auto i_) {
          constexpr int i = decltype(i_)::value;
          load_w(i, w[i]);
        });
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int q = 0; q < CH; ++q) {
        const int idx = tid + NT * (q0 + q);
        if (q0 + q < NQ && idx < CHUNKS) {
          const int hp = idx / C4, c4 = idx % C4;
          const xs::Split3 s0 = xs::split3_fast(v[q].x), s1 = xs::split3_fast(v[q].y), s2 = xs::split3_fast(v[q].z),
                           s3 = xs::split3_fast(v[q].w);
          __bf16* p = sA + hp * kPitch + 4 * c4;
          *reinterpret_cast<uint2*>(p) = make_uint2(xs::pack_hi16(s0.hi, s1.hi), xs::pack_hi16(s2.hi, s3.hi));
          *reinterpret_cast<uint2*>(p + HP * kPitch) = make_uint2(xs::pack_hi16(s0.mid, s1.mid), xs::pack_hi16(s2.mid, s3.mid));
          *reinterpret_cast<uint2*>(p + 2 * HP * kPitch) = make_uint2(xs::pack_hi16(s0.lo, s1.lo), xs::pack_hi16(s2.lo, s3.lo));
        }
      }
    }
  }
  CONV_T(t1);
  __syncthreads();
  CONV_T(t2);

  f32x16 acc[MTW][NTW];
#pragma unroll
  for (int m = 0; m < MTW; ++m)
#pragma unroll
    for (int tt = 0; tt < NTW; ++tt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][tt][r] = 0.f;

  // lane -> pixel (li / 8, li % 8) of a 4 x 8 row tile; row tile m of this wave starts at tile row 4 (wm MTW + m)
  const __bf16* pa[MTW];
#pragma unroll
  for (int m = 0; m < MTW; ++m)
    pa[m] = sA + ((4 * (wm * MTW + m) + (li >> 3)) * HW_ + (li & 7)) * kPitch + 8 * hf;
  bf16x8 a[2][MTW][3];
  auto read_a = [&](auto ks_, bf16x8 (&dst)[MTW][3]) {
    constexpr int ks = decltype(ks_)::value;
    constexpr int tap = ks / KC, kc = ks % KC;
    constexpr int off = ((tap / 3) * HW_ + (tap % 3)) * kPitch + 16 * kc;   // halo origin is (-1, -1): tap (dy, dx) reads (py + dy, px + dx)
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
      for (int p = 0; p < 3; ++p) dst[m][p] = *reinterpret_cast<const bf16x8*>(pa[m] + p * HP * kPitch + off);
  };
  read_a(std::integral_constant<int, 0>{}, a[0]);
  static_for<KS>([&](auto ks_) {
    constexpr int ks = decltype(ks_)::value;
    if constexpr (ks + PF < KS) load_w(ks + PF, w[(ks + PF) % (PF + 1)]);
    if constexpr (ks + 1 < KS) read_a(std::integral_constant<int, ks + 1>{}, a[(ks + 1) & 1]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
      for (int tt = 0; tt < NTW; ++tt) acc[m][tt] = mfma6(a[ks & 1][m], w[ks % (PF + 1)][tt], acc[m][tt]);
    __builtin_amdgcn_sched_barrier(0);
  });

  CONV_T(t3);
  // epilogue: D[i = pixel][j = channel]: lane l holds channel l & 31 of a 32-wide tile, accumulator r pixel (r & 3) + 8 (r >> 2)
  // + 4 (l >> 5) of the 4 x 8 row tile
  float* const yb = A.y + (size_t)b * A.H * A.W * A.N;
#pragma unroll
  for (int m = 0; m < MTW; ++m)
#pragma unroll
    for (int tt = 0; tt < NTW; ++tt) {
      const int col = (nt0 + tt) * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int p = (r & 3) + 8 * (r >> 2) + 4 * hf;
        const int gy = y0 + 4 * (wm * MTW + m) + (p >> 3), gx = x0 + (p & 7);
        if (gy < A.H && gx < A.W) yb[((size_t)gy * A.W + gx) * A.N + col] = acc[m][tt][r];
      }
    }
#ifdef EGTR_CONV_TIMING
  CONV_T(t4);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  CONV_T(t5);
  if (tid == 0 && blockIdx.x < kRecWg) {
    unsigned long long* r = g_conv_rec[blockIdx.x];
    r[0] = rt0;
    r[1] = __builtin_amdgcn_s_memrealtime();
    r[2] = t1 - t0;   // requests + halo tile
    r[3] = t2 - t1;   // barrier
    r[4] = t3 - t2;   // products
    r[5] = t4 - t3;   // epilogue
    r[6] = t5 - t4;   // store drain
    r[7] = 1;
  }
#endif
}

// The same with the channels walked in PHASES of CP (the halo tile of all C channels does not fit the LDS at C >= 256, nor at
// stride 2) and with stride 1 or 2.  A workgroup owns 4 x 8 output pixels x 128 output channels (4 waves x 32); per phase the halo
// tile of CP channels is built, the 9 taps x CP / 16 k-steps of that phase are multiplied, the accumulators carry over.  The
// weight stream is ordered to match: Wm[n][((ph * 3 + dy) * 3 + dx) * CP + c'] = W[n][ph * CP + c'][dy][dx]
// (egtr_amd/ops.py::conv3x3_weights with `phase`).  One phase of k-steps is unrolled; the phases are a loop.
// TAPS == 1: the same machinery as a 1x1 convolution with stride (a bottleneck's shortcut projection: no padding, the tile is
// the 4 x 8 input pixels the outputs read, gathered with the stride; weights XS(W [N, C]) as they are).
template <int C, int CP, int STRIDE, int TAPS = 9>
__global__ __launch_bounds__(256) void conv3x3_x6_phased_kernel(ConvArgs A) {
  constexpr int TH = 4, TW = 8;
  static_assert(TAPS == 9 || TAPS == 1, "3x3 or 1x1");
  constexpr int HH = TAPS == 9 ? STRIDE * (TH - 1) + 3 : TH, HW_ = TAPS == 9 ? STRIDE * (TW - 1) + 3 : TW, HP = HH * HW_;
  constexpr int PH = C / CP;
  constexpr int KCP = CP / 16;            // k-steps per tap and phase
  constexpr int KSP = TAPS * KCP;         // k-steps per phase
  constexpr int KS = PH * KSP;
  constexpr int kPitch = CP + 8;
  constexpr int PF = EGTR_CONV_PF;
  constexpr int C4 = CP / 4;
  static_assert(KSP % (PF + 1) == 0, "the fragment ring must line up at the phase boundary");
  extern __shared__ __attribute__((aligned(16))) char s_raw[];
  __bf16* const sA = reinterpret_cast<__bf16*>(s_raw);   // [3][HP][kPitch]

  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, hf = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nblocks = A.N / 128;
  int t = blockIdx.x;
  const int nb = t % nblocks;
  t /= nblocks;
  const int tx = t % A.tiles_x;
  t /= A.tiles_x;
  const int ty = t % A.tiles_y, b = t / A.tiles_y;
  const int y0 = ty * TH, x0 = tx * TW;                       // output pixels
  const int iy0 = STRIDE * y0 - 1, ix0 = STRIDE * x0 - 1;     // halo origin in the input
  const int nt = nb * 4 + wave;

  const char* const wlane = A.w + (size_t)nt * KS * (3 * xs::kFragBytes) + lane * 16;
  bf16x8 w[PF + 1][3];
  auto load_w = [&](int ks, bf16x8 (&dst)[3]) {
#pragma unroll
    for (int p = 0; p < 3; ++p) dst[p] = *reinterpret_cast<const bf16x8*>(wlane + ((size_t)ks * 3 + p) * xs::kFragBytes);
  };
  static_for<PF>([&](auto i_) {
    constexpr int i = decltype(i_)::value;
    load_w(i, w[i]);
  });
  __builtin_amdgcn_sched_barrier(0);

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const float* const xb = A.x + (size_t)b * A.H * A.W * C;
  constexpr int PS = TAPS == 9 ? STRIDE : 1;   // pixel spacing of the outputs inside the tile
  const __bf16* const pa = sA + (PS * (li >> 3) * HW_ + PS * (li & 7)) * kPitch + 8 * hf;

#pragma unroll 1
  for (int ph = 0; ph < PH; ++ph) {
    if (ph > 0) __syncthreads();   // every wave has read the previous phase's tile
    {
      constexpr int CHUNKS = HP * C4;
      constexpr int NQ = (CHUNKS + 255) / 256;
      constexpr int CH = NQ < 8 ? NQ : 8;
#pragma unroll
      for (int q0 = 0; q0 < NQ; q0 += CH) {
        f32x4v v[CH];
#pragma unroll
        for (int q = 0; q < CH; ++q) {
          const int idx = tid + 256 * (q0 + q);
          const int hp = idx / C4, c4 = idx % C4;
          const int gy = TAPS == 9 ? iy0 + hp / HW_ : STRIDE * (y0 + hp / HW_), gx = TAPS == 9 ? ix0 + hp % HW_ : STRIDE * (x0 + hp % HW_);
          const bool in = (q0 + q < NQ) && idx < CHUNKS && gy >= 0 && gy < A.H && gx >= 0 && gx < A.W;
          v[q] = f32x4v{0.f, 0.f, 0.f, 0.f};
          if (in) v[q] = *reinterpret_cast<const f32x4v*>(xb + ((size_t)gy * A.W + gx) * C + ph * CP + 4 * c4);
        }
#pragma unroll
        for (int q = 0; q < CH; ++q) {
          const int idx = tid + 256 * (q0 + q);
          if (q0 + q < NQ && idx < CHUNKS) {
            const int hp = idx / C4, c4 = idx % C4;
            const xs::Split3 s0 = xs::split3_fast(v[q].x), s1 = xs::split3_fast(v[q].y), s2 = xs::split3_fast(v[q].z),
                             s3 = xs::split3_fast(v[q].w);
            __bf16* p = sA + hp * kPitch + 4 * c4;
            *reinterpret_cast<uint2*>(p) = make_uint2(xs::pack_hi16(s0.hi, s1.hi), xs::pack_hi16(s2.hi, s3.hi));
            *reinterpret_cast<uint2*>(p + HP * kPitch) = make_uint2(xs::pack_hi16(s0.mid, s1.mid), xs::pack_hi16(s2.mid, s3.mid));
            *reinterpret_cast<uint2*>(p + 2 * HP * kPitch) = make_uint2(xs::pack_hi16(s0.lo, s1.lo), xs::pack_hi16(s2.lo, s3.lo));
          }
        }
      }
    }
    __syncthreads();

    const int ks0 = ph * KSP;
    bf16x8 a[2][3];
    auto read_a = [&](auto j_, bf16x8 (&dst)[3]) {
      constexpr int j = decltype(j_)::value;
      constexpr int tap = j / KCP, kc = j % KCP;
      constexpr int off = ((tap / 3) * HW_ + (tap % 3)) * kPitch + 16 * kc;
#pragma unroll
      for (int p = 0; p < 3; ++p) dst[p] = *reinterpret_cast<const bf16x8*>(pa + p * HP * kPitch + off);
    };
    read_a(std::integral_constant<int, 0>{}, a[0]);
    static_for<KSP>([&](auto j_) {
      constexpr int j = decltype(j_)::value;
      if (ks0 + j + PF < KS) load_w(ks0 + j + PF, w[(j + PF) % (PF + 1)]);   // (uniform; false only in the last phase's tail)
      if constexpr (j + 1 < KSP) read_a(std::integral_constant<int, j + 1>{}, a[(j + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      acc = mfma6(a[j & 1], w[j % (PF + 1)], acc);
      __builtin_amdgcn_sched_barrier(0);
    });
  }

  const int Ho = (A.H - 1) / STRIDE + 1, Wo = (A.W - 1) / STRIDE + 1;
  float* const yb = A.y + (size_t)b * Ho * Wo * A.N;
  const int col = nt * 32 + li;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int p = (r & 3) + 8 * (r >> 2) + 4 * hf;
    const int gy = y0 + (p >> 3), gx = x0 + (p & 7);
    if (gy < Ho && gx < Wo) yb[((size_t)gy * Wo + gx) * A.N + col] = acc[r];
  }
}

// The phased kernel with K split over the waves (C = 512: one wave walking K = 4608 alone takes 24 us of matrix time).  A workgroup
// owns 4 x 8 output pixels x ONE 32-channel tile; within every phase wave q multiplies the q-th quarter of the phase's channels
// (all 9 taps), so a wave walks K / 4; the four partial tiles meet in LDS at the end (fixed order: wave 0 + 1 + 2 + 3), each wave
// finishing and storing a quarter of the rows.
template <int C, int CP, int STRIDE>
__global__ __launch_bounds__(256) void conv3x3_x6_ksplit_kernel(ConvArgs A) {
  constexpr int TH = 4, TW = 8;
  constexpr int HH = STRIDE * (TH - 1) + 3, HW_ = STRIDE * (TW - 1) + 3, HP = HH * HW_;
  constexpr int PH = C / CP;
  constexpr int KCP = CP / 16;            // k-steps per tap and phase
  constexpr int KSP = 9 * KCP;            // k-steps per phase (all waves)
  constexpr int KCW = KCP / 4;            // k-steps per tap, phase and wave
  constexpr int JS = 9 * KCW;             // k-steps per phase and wave
  constexpr int kPitch = CP + 8;
  constexpr int PF = (JS % 6 == 0) ? 5 : 2;
  constexpr int C4 = CP / 4;
  static_assert(KCP % 4 == 0 && JS % (PF + 1) == 0, "phase shape");
  extern __shared__ __attribute__((aligned(16))) char s_raw[];
  __bf16* const sA = reinterpret_cast<__bf16*>(s_raw);   // [3][HP][kPitch]; afterwards the partial tiles [4][16][64] floats

  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, hf = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntiles = A.N / 32;
  int t = blockIdx.x;
  const int nt = t % ntiles;
  t /= ntiles;
  const int tx = t % A.tiles_x;
  t /= A.tiles_x;
  const int ty = t % A.tiles_y, b = t / A.tiles_y;
  const int y0 = ty * TH, x0 = tx * TW;
  const int iy0 = STRIDE * y0 - 1, ix0 = STRIDE * x0 - 1;

  // k-step (phase ph, local step j) of this wave: tap j / KCW, channel step wave * KCW + j % KCW of the phase
  const char* const wlane = A.w + (size_t)nt * (PH * KSP) * (3 * xs::kFragBytes) + lane * 16;
  auto wptr = [&](int ph, int j) {
    const int ks = ph * KSP + (j / KCW) * KCP + wave * KCW + (j % KCW);
    return wlane + (size_t)ks * 3 * xs::kFragBytes;
  };
  bf16x8 w[PF + 1][3];
  auto load_w = [&](int ph, int j, bf16x8 (&dst)[3]) {
    const char* p0 = wptr(ph, j);
#pragma unroll
    for (int p = 0; p < 3; ++p) dst[p] = *reinterpret_cast<const bf16x8*>(p0 + p * xs::kFragBytes);
  };
  static_for<PF>([&](auto i_) {
    constexpr int i = decltype(i_)::value;
    load_w(0, i, w[i]);
  });
  __builtin_amdgcn_sched_barrier(0);

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const float* const xb = A.x + (size_t)b * A.H * A.W * C;
  const __bf16* const pa = sA + (STRIDE * (li >> 3) * HW_ + STRIDE * (li & 7)) * kPitch + 8 * hf + 16 * KCW * wave;

#pragma unroll 1
  for (int ph = 0; ph < PH; ++ph) {
    if (ph > 0) __syncthreads();
    {
      constexpr int CHUNKS = HP * C4;
      constexpr int NQ = (CHUNKS + 255) / 256;
      constexpr int CH = NQ < 8 ? NQ : 8;
#pragma unroll
      for (int q0 = 0; q0 < NQ; q0 += CH) {
        f32x4v v[CH];
#pragma unroll
        for (int q = 0; q < CH; ++q) {
          const int idx = tid + 256 * (q0 + q);
          const int hp = idx / C4, c4 = idx % C4;
          const int gy = iy0 + hp / HW_, gx = ix0 + hp % HW_;
          const bool in = (q0 + q < NQ) && idx < CHUNKS && gy >= 0 && gy < A.H && gx >= 0 && gx < A.W;
          v[q] = f32x4v{0.f, 0.f, 0.f, 0.f};
          if (in) v[q] = *reinterpret_cast<const f32x4v*>(xb + ((size_t)gy * A.W + gx) * C + ph * CP + 4 * c4);
        }
#pragma unroll
        for (int q = 0; q < CH; ++q) {
          const int idx = tid + 256 * (q0 + q);
          if (q0 + q < NQ && idx < CHUNKS) {
            const int hp = idx / C4, c4 = idx % C4;
            const xs::Split3 s0 = xs::split3_fast(v[q].x), s1 = xs::split3_fast(v[q].y), s2 = xs::split3_fast(v[q].z),
                             s3 = xs::split3_fast(v[q].w);
            __bf16* p = sA + hp * kPitch + 4 * c4;
            *reinterpret_cast<uint2*>(p) = make_uint2(xs::pack_hi16(s0.hi, s1.hi), xs::pack_hi16(s2.hi, s3.hi));
            *reinterpret_cast<uint2*>(p + HP * kPitch) = make_uint2(xs::pack_hi16(s0.mid, s1.mid), xs::pack_hi16(s2.mid, s3.mid));
            *reinterpret_cast<uint2*>(p + 2 * HP * kPitch) = make_uint2(xs::pack_hi16(s0.lo, s1.lo), xs::pack_hi16(s2.lo, s3.lo));
          }
        }
      }
    }
    __syncthreads();

    bf16x8 a[2][3];
    auto read_a = [&](auto j_, bf16x8 (&dst)[3]) {
      constexpr int j = decltype(j_)::value;
      constexpr int tap = j / KCW, kcl = j % KCW;
      constexpr int off = ((tap / 3) * HW_ + (tap % 3)) * kPitch + 16 * kcl;
#pragma unroll
      for (int p = 0; p < 3; ++p) dst[p] = *reinterpret_cast<const bf16x8*>(pa + p * HP * kPitch + off);
    };
    read_a(std::integral_constant<int, 0>{}, a[0]);
    static_for<JS>([&](auto j_) {
      constexpr int j = decltype(j_)::value;
      if constexpr (j + PF < JS) {
        load_w(ph, j + PF, w[(j + PF) % (PF + 1)]);
      } else {
        if (ph + 1 < PH) load_w(ph + 1, j + PF - JS, w[(j + PF) % (PF + 1)]);   // (uniform)
      }
      if constexpr (j + 1 < JS) read_a(std::integral_constant<int, j + 1>{}, a[(j + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      acc = mfma6(a[j & 1], w[j % (PF + 1)], acc);
      __builtin_amdgcn_sched_barrier(0);
    });
  }

  // the four partial tiles: [wave][accumulator r][lane] floats in LDS, then wave q sums and stores rows r = 4 q .. 4 q + 3
  __syncthreads();
  float* const part = reinterpret_cast<float*>(s_raw);
#pragma unroll
  for (int r = 0; r < 16; ++r) part[(wave * 16 + r) * 64 + lane] = acc[r];
  __syncthreads();
  const int Ho = (A.H - 1) / STRIDE + 1, Wo = (A.W - 1) / STRIDE + 1;
  float* const yb = A.y + (size_t)b * Ho * Wo * A.N;
  const int col = nt * 32 + li;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int r = 4 * wave + rr;
    const float v = ((part[(0 * 16 + r) * 64 + lane] + part[(1 * 16 + r) * 64 + lane]) + part[(2 * 16 + r) * 64 + lane]) +
                    part[(3 * 16 + r) * 64 + lane];
    const int p = (r & 3) + 8 * (r >> 2) + 4 * hf;
    const int gy = y0 + (p >> 3), gx = x0 + (p & 7);
    if (gy < Ho && gx < Wo) yb[((size_t)gy * Wo + gx) * A.N + col] = v;
  }
}

template <int C, int CP, int STRIDE>
int launch_ksplit(hipStream_t st, ConvArgs A) {
  static unsigned long long raised = 0;
  constexpr int HP = (STRIDE * 3 + 3) * (STRIDE * 7 + 3);
  constexpr int lds = 3 * HP * (CP + 8) * 2;
  static_assert(lds >= 4 * 16 * 64 * 4, "the partial tiles reuse the halo tile's LDS");
  auto kern = conv3x3_x6_ksplit_kernel<C, CP, STRIDE>;
  if (lds > 64 * 1024) {
    const int rc = egtr_raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds, &raised);
    if (rc != EGTR_OK) return rc;
  }
  const int Ho = (A.H - 1) / STRIDE + 1, Wo = (A.W - 1) / STRIDE + 1;
  A.tiles_x = (Wo + 7) / 8;
  A.tiles_y = (Ho + 3) / 4;
  const long long wgs = (long long)A.B * A.tiles_x * A.tiles_y * (A.N / 32);
  if (wgs >= (1ll << 31)) return EGTR_E_UNSUPPORTED;
  hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(256), lds, st, A);
  return egtr_check_launch();
}

template <int C, int CP, int STRIDE, int TAPS = 9>
int launch_phased(hipStream_t st, ConvArgs A) {
  static unsigned long long raised = 0;
  constexpr int HP = TAPS == 9 ? (STRIDE * 3 + 3) * (STRIDE * 7 + 3) : 32;
  constexpr int lds = 3 * HP * (CP + 8) * 2;
  auto kern = conv3x3_x6_phased_kernel<C, CP, STRIDE, TAPS>;
  if (lds > 64 * 1024) {
    const int rc = egtr_raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds, &raised);
    if (rc != EGTR_OK) return rc;
  }
  const int Ho = (A.H - 1) / STRIDE + 1, Wo = (A.W - 1) / STRIDE + 1;
  A.tiles_x = (Wo + 7) / 8;
  A.tiles_y = (Ho + 3) / 4;
  const long long wgs = (long long)A.B * A.tiles_x * A.tiles_y * (A.N / 128);
  if (wgs >= (1ll << 31)) return EGTR_E_UNSUPPORTED;
  hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(256), lds, st, A);
  return egtr_check_launch();
}

template <int C, int TH, int WM, int NTW, int NWV = 4>
int launch(hipStream_t st, ConvArgs A) {
  static unsigned long long raised = 0;
  // (fewer resident workgroups -- LDS padded to 53 / 80 / 160 KB so that later workgroups' halo loads overlap earlier ones'
  // products -- measured slower: 23.1 -> 27.7 / 31.1 / 36.6 us at C = 64)
  constexpr int lds = 3 * (TH + 2) * 10 * (C + 8) * 2;
  constexpr int BN = 32 * NTW * (NWV / WM);
  auto kern = conv3x3_x6_kernel<C, TH, WM, NTW, NWV>;
  if (lds > 64 * 1024) {
    const int rc = egtr_raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds, &raised);
    if (rc != EGTR_OK) return rc;
  }
  A.tiles_x = (A.W + 7) / 8;
  A.tiles_y = (A.H + TH - 1) / TH;
  const long long wgs = (long long)A.B * A.tiles_x * A.tiles_y * (A.N / BN);
  if (wgs >= (1ll << 31)) return EGTR_E_UNSUPPORTED;
  hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(64 * NWV), lds, st, A);
  return egtr_check_launch();
}

}  // namespace

#ifdef EGTR_CONV_TIMING
extern "C" int egtr_conv3x3_stamps(unsigned long long* host_out, int reset) {   // host_out: kRecWg x 8
  if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_conv_rec), sizeof(unsigned long long) * kRecWg * 8) != hipSuccess)
    return EGTR_E_LAUNCH;
  if (reset) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_conv_rec)) != hipSuccess ||
        hipMemset(p, 0, sizeof(unsigned long long) * kRecWg * 8) != hipSuccess)
      return EGTR_E_LAUNCH;
  }
  return EGTR_OK;
}
#endif

extern "C" int egtr_conv3x3_phase_channels(int C, int N, int stride, int variant) {
  // channels per phase of the weight stream the kernel that serves (C, N, stride, variant) expects; 0: not served
  if (C != N) return 0;
  if (stride == 1) {
    if (C == 64 || C == 128) return C;   // (every variant)
    if (C == 256) return variant == 3 ? 256 : 128;
    if (C == 512) return 128;
    return 0;
  }
  if (stride == 2) return (C == 128 || C == 256 || C == 512) ? 64 : 0;
  return 0;
}

extern "C" int egtr_conv3x3_x6_f32(egtr_stream_t stream, const float* x, const void* w_xs, float* y, int B, int H, int W, int C,
                                   int N, int stride, int variant) {
  if (!x || !w_xs || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || N <= 0) return EGTR_E_ARG;
  if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(y) & 15) || (reinterpret_cast<uintptr_t>(w_xs) & 15))
    return EGTR_E_UNSUPPORTED;
  if (egtr_conv3x3_phase_channels(C, N, stride, variant) == 0) return EGTR_E_UNSUPPORTED;
  ConvArgs A{x, static_cast<const char*>(w_xs), y, B, H, W, N, 0, 0};
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (stride == 2) {
    // (inside the forward: C = 128 phased 27.1 us, K split 33.8; C = 256 phased 37.4, K split 39.6; C = 512 phased 49, K split 37)
    if (C == 128) return variant == 1 ? launch_ksplit<128, 64, 2>(st, A) : launch_phased<128, 64, 2>(st, A);
    if (C == 256) return variant == 1 ? launch_ksplit<256, 64, 2>(st, A) : launch_phased<256, 64, 2>(st, A);
    if (variant == 1) return launch_phased<512, 64, 2>(st, A);
    return launch_ksplit<512, 64, 2>(st, A);
  }
  // variant 0 = the library's choice, by the kernels' times INSIDE the forward (tools/conv2_ab.sh; stand-alone the variants are
  // within 10 % of each other): the kernels are bound by the matrix pipes during their product phase and by the halo-tile
  // latency before it (tools/conv3x3_timing.sh) -- many small workgroups balance the CUs best, and from C = 128 on splitting K
  // over the waves of a workgroup shortens the longest wave.  The other variants pin a kernel for the tests and the sweeps.
  if (C == 64) {
    if (variant == 1) return launch<64, 8, 2, 1>(st, A);     // 8 x 8 pixels x 64 channels: 2 (pixel halves) x 2 (channel tiles)
    if (variant == 2) return launch<64, 16, 2, 1>(st, A);    // 16 x 8 pixels
    if (variant == 3) return launch<64, 16, 4, 1>(st, A);    // 16 x 8 pixels x 32 channels: the four waves share a weight stream
    if (variant == 4) return launch_ksplit<64, 64, 1>(st, A);   // K split over the waves: 24.7 us inside the forward
    return launch<64, 4, 1, 1, 2>(st, A);                       // 4 x 8 pixels x 64 channels, two waves: 20.4 us
  }
  if (C == 128) {
    if (variant == 1) return launch<128, 8, 2, 1>(st, A);    // 8 x 8 pixels x 64 channels
    if (variant == 3) return launch<128, 16, 4, 1>(st, A);   // 16 x 8 pixels x 32 channels
    if (variant == 4) return launch<128, 4, 1, 1, 2>(st, A); // 4 x 8 pixels x 64 channels, two waves
    if (variant == 2) return launch<128, 4, 1, 1>(st, A);    // 4 x 8 pixels x 128 channels (4 waves x 32)
    return launch_ksplit<128, 128, 1>(st, A);
  }
  if (C == 256) {
    if (variant == 1) return launch_phased<256, 128, 1>(st, A);   // two phases of 128 channels, three workgroups per CU: 31.4 us
    if (variant == 3) return launch<256, 4, 1, 1>(st, A);
    return launch_ksplit<256, 128, 1>(st, A);   // K split over the waves                           // inside the forward, the whole halo tile resident (95 KiB): 29.6
  }
  if (variant == 1) return launch_phased<512, 128, 1>(st, A);
  return launch_ksplit<512, 128, 1>(st, A);
}

extern "C" int egtr_conv1x1_strided_x6_f32(egtr_stream_t stream, const float* x, const void* w_xs, float* y, int B, int H, int W,
                                           int C, int N, int stride) {
  if (!x || !w_xs || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || N <= 0) return EGTR_E_ARG;
  if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(y) & 15) || (reinterpret_cast<uintptr_t>(w_xs) & 15) ||
      N % 128 || (stride != 1 && stride != 2))
    return EGTR_E_UNSUPPORTED;
  ConvArgs A{x, static_cast<const char*>(w_xs), y, B, H, W, N, 0, 0};
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (stride == 2) {
    if (C == 256) return launch_phased<256, 256, 2, 1>(st, A);
    if (C == 512) return launch_phased<512, 256, 2, 1>(st, A);
    if (C == 1024) return launch_phased<1024, 256, 2, 1>(st, A);
    return EGTR_E_UNSUPPORTED;
  }
  if (C == 256) return launch_phased<256, 256, 1, 1>(st, A);
  if (C == 512) return launch_phased<512, 256, 1, 1>(st, A);
  if (C == 1024) return launch_phased<1024, 256, 1, 1>(st, A);
  return EGTR_E_UNSUPPORTED;
}
