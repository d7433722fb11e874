// The ResNet stem in inference as ONE kernel: 7x7 convolution (stride 2, padding 3, 3 -> 64 channels) + folded batch-norm shift
// + ReLU + 3x3 max-pool (stride 2, padding 1), NCHW fp32 pixels in, channels-last fp32 out
// (reference: model/deformable_detr.py:735-760 -- the timm ResNet-50 backbone: conv1 -> bn1 -> act1 -> maxpool).
// Before: MIOpen's Winograd convolution (73 us at 600 x 1000) writing the 300 x 500 x 64 map, a shift + ReLU + pool kernel reading
// it back (19 us), a layout change of the pooled map (12 us).  Here the convolution output never leaves the CU.
//
// Arithmetic: the six-term split-bf16 product of the x6 kernels (fp32 operands, fp32 accumulation, error of an fp32 convolution).
// K is tiny and awkward (7 x 7 x 3 = 147), so it is laid out for the matrix cores: per kernel row ky the 7 taps x 3 channels are
// padded to 8 taps x 4 channels = 32 (two k-steps), K = 7 x 32 = 224; the padded tap / channel carry zero weights.  With the input
// tile parked in LDS as [row][column][4 channels] bf16, the 8 consecutive k of an MFMA operand are two neighbouring input pixels:
// one aligned 16-byte read at a per-lane base plus an immediate.
//
// A workgroup (4 waves) owns 3 x 8 POOLED pixels x 64 channels: the 7 x 17 convolution outputs they cover (128 slots = 4 row
// tiles of 32 x 2 channel tiles: two per wave), from a 19 x 40 input tile (4 pooled rows = 5 row tiles deal 3 : 2 over the waves
// and measure 47 us against 44.5; tools/stem_variants.sh).  Convolution outputs + shift, ReLU -> LDS (fp32,
// zeros outside the image: a pool window always holds a valid value >= 0), then the pool reads 3 x 3 of them per output.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"
#include "x6_common.h"
#include "xs_format.h"

namespace {
using namespace x6;

struct StemArgs {
  const float* x;     // [B, 3, H, W]
  const char* w;      // XS(Wm [64, 224]), Wm[n][ky * 32 + kx * 4 + c] (kx == 7 and c == 3: zeros)
  const float* bias;  // [64] folded batch-norm shift
  float* y;           // [B, Hp, Wp, 64] channels-last
  int B, H, W, Hc, Wc, Hp, Wp, tiles_x, tiles_y;
};

#ifndef EGTR_STEM_PH
#define EGTR_STEM_PH 3
#endif
constexpr int kPH = EGTR_STEM_PH, kPW = 8;      // pooled pixels per workgroup
constexpr int kCH = 2 * kPH + 1, kCW = 2 * kPW + 1;   // convolution outputs per workgroup: 7 x 17
constexpr int kCP = kCH * kCW;                  // 119
constexpr int kMT = (kCP + 31) / 32;            // row tiles of 32 convolution pixels
constexpr int kIH = 2 * kCH + 5, kIW = 40;      // input tile: 19 rows x 40 columns (2 * 17 + 5 = 39, + the padded tap)
constexpr int kKS = 14;                         // k-steps: 7 kernel rows x 2
constexpr int kPieceBytes = kIH * kIW * 4 * 2;  // one bf16 piece of the input tile: 7360 bytes
constexpr int kConvPitch = 64;                  // floats per convolution pixel in LDS

#ifdef EGTR_STEM_TIMING
// debugging aid (tools/stem_timing.sh): per workgroup of the last launch, real-time counter at entry / exit and shader-clock stamps
constexpr int kRecWg = 4096;
__device__ unsigned long long g_stem_rec[kRecWg][8];
#define STEM_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#else
#define STEM_T(v) do { } while (0)
#endif

__global__ __launch_bounds__(256) void stem_x6_kernel(StemArgs A) {
  __shared__ __attribute__((aligned(16))) char s_in[3 * kPieceBytes];           // [piece][row][col][4] bf16
  __shared__ __attribute__((aligned(16))) float s_conv[32 * kMT * kConvPitch];  // [conv pixel slot][64]
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, hf = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef EGTR_STEM_TIMING
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  STEM_T(t0);
  int t = blockIdx.x;
  const int tx = t % A.tiles_x;
  t /= A.tiles_x;
  const int ty = t % A.tiles_y, b = t / A.tiles_y;
  const int py0 = ty * kPH, px0 = tx * kPW;          // pooled origin
  const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;    // convolution origin (pool padding 1)
  const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;    // input origin (convolution padding 3)

  // the weight fragments of this wave's channel tile: 14 k-steps x 3 pieces = 42 KiB, re-read per row tile (the whole grid reads
  // the same 84 KiB: L1 / L2 hits)
  const int nt = wave & 1;
  const char* const wlane = A.w + (size_t)nt * kKS * (3 * xs::kFragBytes) + lane * 16;

  // input tile: item = (channel c, row r, column quad q): 3 x 23 x 10 float4-sized groups of 4 columns
  for (int it = tid; it < 3 * kIH * (kIW / 4); it += 256) {
    const int q = it % (kIW / 4), r = (it / (kIW / 4)) % kIH, c = it / ((kIW / 4) * kIH);
    const int gy = iy0 + r;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gx = ix0 + 4 * q + j;
      v[j] = (gy >= 0 && gy < A.H && gx >= 0 && gx < A.W) ? A.x[((size_t)(b * 3 + c) * A.H + gy) * A.W + gx] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const xs::Split3 s = xs::split3_fast(v[j]);
      unsigned short* p = reinterpret_cast<unsigned short*>(s_in) + ((r * kIW + 4 * q + j) * 4 + c);
      p[0] = (unsigned short)(s.hi >> 16);
      p[kPieceBytes / 2] = (unsigned short)(s.mid >> 16);
      p[kPieceBytes] = (unsigned short)(s.lo >> 16);
    }
  }
  // the fourth channel of every pixel is a zero (its weights are zeros too, but 0 x garbage could be NaN)
  for (int it = tid; it < 3 * kIH * kIW; it += 256) {
    const int piece = it / (kIH * kIW), px = it % (kIH * kIW);
    reinterpret_cast<unsigned short*>(s_in)[piece * (kPieceBytes / 2) + px * 4 + 3] = 0;
  }
  STEM_T(t1);
  __syncthreads();
  STEM_T(t2);

  // k-steps outside, this wave's row tiles (m = wave >> 1, + 2, + 4: waves 0, 1 three of the five, waves 2, 3 two) inside: a
  // weight fragment is loaded once per wave and k-step (not once per row tile), PF k-steps ahead; the row tiles' accumulators are
  // independent MFMA chains.  (Row tiles outside, fragments loaded where used: 11.7 of a workgroup's 17.3 us were this loop.)
  const float bz = A.bias[nt * 32 + li];
  constexpr int MW = (kMT + 1) / 2, PF = 3;
  bf16x8 w[PF + 1][3];
  auto load_w = [&](int ks, bf16x8 (&dst)[3]) {
#pragma unroll
    for (int p = 0; p < 3; ++p) dst[p] = *reinterpret_cast<const bf16x8*>(wlane + ((size_t)ks * 3 + p) * xs::kFragBytes);
  };
  static_for<PF>([&](auto i_) {
    constexpr int i = decltype(i_)::value;
    load_w(i, w[i]);
  });
  const char* pa[MW];
  f32x16 acc[MW];
#pragma unroll
  for (int i = 0; i < MW; ++i) {
    const int slot = min(32 * ((wave >> 1) + 2 * i) + li, kCP - 1);   // (slots 153 .. 159 repeat the last pixel)
    pa[i] = s_in + ((2 * (slot / kCW)) * kIW + 2 * (slot % kCW)) * 8 + hf * 16;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  }
  static_for<kKS>([&](auto ks_) {
    constexpr int ks = decltype(ks_)::value;
    constexpr int ky = ks >> 1, h = ks & 1;
    if constexpr (ks + PF < kKS) {
      load_w(ks + PF, w[(ks + PF) % (PF + 1)]);
      __builtin_amdgcn_sched_barrier(0);
    }
    // k = ky * 32 + 16 h + 8 hf + (0 .. 7) = taps kx = 4 h + 2 hf, + 1 (4 channels each) of kernel row ky
#pragma unroll
    for (int i = 0; i < MW; ++i) {
      if ((wave >> 1) + 2 * i < kMT) {                // (wave-uniform)
        bf16x8 a[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) a[p] = *reinterpret_cast<const bf16x8*>(pa[i] + p * kPieceBytes + (ky * kIW + 4 * h) * 8);
        if constexpr (h == 1) {
          // lanes of the upper k-group hold taps 6 and 7: the padded tap's weights are zeros, but 0 x (a non-finite neighbour
          // pixel) would be NaN -- a pixel must reach exactly the windows that contain it: blank those four elements
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            u32x4 u = __builtin_bit_cast(u32x4, a[p]);
            u.z = hf ? 0u : u.z;
            u.w = hf ? 0u : u.w;
            a[p] = __builtin_bit_cast(bf16x8, u);
          }
        }
        acc[i] = mfma6(a, w[ks % (PF + 1)], acc[i]);
      }
    }
  });
  // D[i = pixel slot][j = channel]: lane l holds channel l & 31, accumulator r slot (r & 3) + 8 (r >> 2) + 4 (l >> 5)
#pragma unroll
  for (int i = 0; i < MW; ++i) {
    const int m = (wave >> 1) + 2 * i;
    if (m < kMT) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int s2 = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * hf;
        const int cy = cy0 + s2 / kCW, cx = cx0 + s2 % kCW;
        const bool valid = s2 < kCP && cy >= 0 && cy < A.Hc && cx >= 0 && cx < A.Wc;
        s_conv[s2 * kConvPitch + nt * 32 + li] = valid ? egtr_relu(acc[i][r] + bz) : 0.f;
      }
    }
  }
  STEM_T(t3);
  __syncthreads();
  STEM_T(t4);

  // pool: item = (pooled pixel, channel quad): 32 x 16
  for (int it = tid; it < kPH * kPW * 16; it += 256) {
    const int c4 = it & 15, pp = it >> 4;
    const int pyl = pp / kPW, pxl = pp % kPW;
    const int py = py0 + pyl, px = px0 + pxl;
    if (py >= A.Hp || px >= A.Wp) continue;
    float4 mx = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const float4 v = *reinterpret_cast<const float4*>(&s_conv[((2 * pyl + dy) * kCW + 2 * pxl + dx) * kConvPitch + 4 * c4]);
        // (NaN propagates like torch's max-pool: a comparison with NaN is false, so test for it)
        mx.x = (v.x > mx.x || v.x != v.x) ? v.x : mx.x;
        mx.y = (v.y > mx.y || v.y != v.y) ? v.y : mx.y;
        mx.z = (v.z > mx.z || v.z != v.z) ? v.z : mx.z;
        mx.w = (v.w > mx.w || v.w != v.w) ? v.w : mx.w;
      }
    *reinterpret_cast<float4*>(A.y + (((size_t)b * A.Hp + py) * A.Wp + px) * 64 + 4 * c4) = mx;
  }
#ifdef EGTR_STEM_TIMING
  STEM_T(t5);
  if (tid == 0 && blockIdx.x < kRecWg) {
    unsigned long long* r = g_stem_rec[blockIdx.x];
    r[0] = rt0;
    r[1] = __builtin_amdgcn_s_memrealtime();
    r[2] = t1 - t0;   // input tile
    r[3] = t2 - t1;   // barrier
    r[4] = t3 - t2;   // products + outputs to LDS
    r[5] = t4 - t3;   // barrier
    r[6] = t5 - t4;   // pool + stores
    r[7] = 1;
  }
#endif
}

}  // namespace

#ifdef EGTR_STEM_TIMING
extern "C" int egtr_stem_stamps(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stem_rec), sizeof(unsigned long long) * kRecWg * 8) == hipSuccess ? EGTR_OK
                                                                                                                  : EGTR_E_LAUNCH;
}
#endif

extern "C" int egtr_stem_conv7x7_pool_x6_f32(egtr_stream_t stream, const float* x, const void* w_xs, const float* bias, float* y,
                                             int B, int H, int W) {
  if (!x || !w_xs || !bias || !y || B <= 0 || H <= 0 || W <= 0) return EGTR_E_ARG;
  if ((reinterpret_cast<uintptr_t>(y) & 15) || (reinterpret_cast<uintptr_t>(w_xs) & 15)) return EGTR_E_UNSUPPORTED;
  StemArgs A{x, static_cast<const char*>(w_xs), bias, y, B, H, W, 0, 0, 0, 0, 0, 0};
  A.Hc = (H - 1) / 2 + 1;
  A.Wc = (W - 1) / 2 + 1;
  A.Hp = (A.Hc - 1) / 2 + 1;
  A.Wp = (A.Wc - 1) / 2 + 1;
  A.tiles_x = (A.Wp + kPW - 1) / kPW;
  A.tiles_y = (A.Hp + kPH - 1) / kPH;
  const long long wgs = (long long)B * A.tiles_x * A.tiles_y;
  if (wgs >= (1ll << 31)) return EGTR_E_UNSUPPORTED;
  hipLaunchKernelGGL(stem_x6_kernel, dim3((unsigned)wgs), dim3(256), 0, static_cast<hipStream_t>(stream), A);
  return egtr_check_launch();
}
