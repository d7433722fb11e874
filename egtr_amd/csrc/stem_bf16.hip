// The bf16 twin of stem_x6.hip (the bf16 model of the stress configuration): the ResNet stem in inference as ONE kernel -- 7x7
// convolution (stride 2, padding 3, 3 -> 64 channels) + folded batch-norm shift + ReLU + 3x3 max-pool (stride 2, padding 1), NCHW
// bf16 pixels in, channels-last bf16 out, fp32 accumulation (reference: model/deformable_detr.py:735-760 -- the timm ResNet-50
// backbone: conv1 -> bn1 -> act1 -> maxpool).  Before: a layout change of the image, MIOpen's bf16 convolution (0.46 ms at bs 16,
// 800 x 1333) writing the 400 x 667 x 64 maps, torch's channels-last max-pool reading them back (0.33 ms), the shift + ReLU pass.
// Rounding points as in that composition: the convolution output is rounded to bf16, shift + ReLU in fp32, rounded once more (max
// commutes with the monotone shift / ReLU / rounding, so pooling behind them is the same function).
//
// The layout of stem_x6.hip with one MFMA per product: per kernel row ky the 7 taps x 3 channels are padded to 8 taps x 4 channels
// (two k-steps), K = 7 x 32 = 224, padded tap / channel with zero weights; the input tile sits in LDS as [row][column][4 channels]
// bf16, so the 8 consecutive k of an MFMA operand are two neighbouring input pixels: one aligned 16-byte read at a per-lane base
// plus an immediate.  A workgroup (4 waves) owns 4 x 8 POOLED pixels x 64 channels: the 9 x 17 convolution outputs they cover
// (5 row tiles of 32 x 2 channel tiles dealt to the waves) from a 23 x 40 input tile; outputs -> LDS (fp32, zeros outside the
// image), the pool reads 3 x 3 of them per output and stores 8 channels = 16 bytes per thread.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"
#include "x6_common.h"

namespace {
using x6::bf16x8;
using x6::f32x16;
using x6::static_for;

struct StemArgs {
  const unsigned short* x;   // [B, 3, H, W] bf16
  const unsigned short* w;   // packed fragments [2 channel tiles][14 k-steps][64 lanes][8] bf16 (egtr_conv1x1_tail_pack_weights_bf16 of
                             // Wm [64, 224], Wm[n][ky * 32 + kx * 4 + c], zeros at kx == 7 / c == 3)
  const float* bias;         // [64] folded batch-norm shift
  unsigned short* y;         // [B, Hp, Wp, 64] channels-last bf16
  int B, H, W, Hc, Wc, Hp, Wp, tiles_x, tiles_y;
};

#ifndef EGTR_STEM_BF16_PH
#define EGTR_STEM_BF16_PH 4
#endif
constexpr int kPH = EGTR_STEM_BF16_PH, kPW = 8;       // pooled pixels per workgroup
constexpr int kCH = 2 * kPH + 1, kCW = 2 * kPW + 1;   // convolution outputs per workgroup: 9 x 17
constexpr int kCP = kCH * kCW;                        // 153
constexpr int kMT = (kCP + 31) / 32;                  // row tiles of 32 convolution pixels
constexpr int kIH = 2 * kCH + 5, kIW = 40;            // input tile: 23 rows x 40 columns (2 * 17 + 5 = 39, + the padded tap)
constexpr int kKS = 14;                               // k-steps: 7 kernel rows x 2
constexpr int kConvPitch = 64;                        // bf16 elements per convolution pixel in LDS (rounding is monotone: the
                                                      // pool of the rounded values is the rounded pool)

__device__ __forceinline__ float bf2f(unsigned short u) { return __uint_as_float(((unsigned)u) << 16); }
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {   // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 v;
  v[0] = (__bf16)a;
  v[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float rbf(float a) { return __uint_as_float(pk_bf16(a, 0.f) << 16); }

__global__ __launch_bounds__(256) void stem_bf16_kernel(StemArgs A) {
  __shared__ __attribute__((aligned(16))) unsigned short s_in[kIH * kIW * 4];       // [row][col][4] bf16
  __shared__ __attribute__((aligned(16))) unsigned short s_conv[32 * kMT * kConvPitch];   // [conv pixel slot][64] bf16
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, hf = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int t = blockIdx.x;
  const int tx = t % A.tiles_x;
  t /= A.tiles_x;
  const int ty = t % A.tiles_y, b = t / A.tiles_y;
  const int py0 = ty * kPH, px0 = tx * kPW;          // pooled origin
  const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;    // convolution origin (pool padding 1)
  const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;    // input origin (convolution padding 3)
  const int nt = wave & 1;
  const unsigned short* const wlane = A.w + ((size_t)nt * kKS * 64 + lane) * 8;

  // input tile: item = one element (channel c, row r, column x), consecutive threads on consecutive columns of a row (a wave's
  // load covers 1.6 rows of 80 bytes); all requests of a thread first, then the LDS stores; the fourth channel is a zero
  {
    constexpr int NE = 3 * kIH * kIW, NQ = (NE + 255) / 256;
    unsigned short v[NQ];
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
      const int it = tid + 256 * k;
      const int xx = it % kIW, r = (it / kIW) % kIH, c = it / (kIW * kIH);
      const int gy = iy0 + r, gx = ix0 + xx;
      v[k] = (it < NE && gy >= 0 && gy < A.H && gx >= 0 && gx < A.W) ? A.x[((size_t)(b * 3 + c) * A.H + gy) * A.W + gx]
                                                                     : (unsigned short)0;
    }
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
      const int it = tid + 256 * k;
      if (it < NE) {
        const int xx = it % kIW, r = (it / kIW) % kIH, c = it / (kIW * kIH);
        s_in[(r * kIW + xx) * 4 + c] = v[k];
      }
    }
  }
  for (int it = tid; it < kIH * kIW; it += 256) s_in[it * 4 + 3] = 0;
  __syncthreads();

  // k-steps outside, this wave's row tiles (m = wave >> 1, + 2, ...) inside: a weight fragment is loaded ONCE per wave (all 14
  // up front: 56 registers) and feeds every row tile; the row tiles' accumulators are independent MFMA chains
  const float bz = A.bias[nt * 32 + li];
  constexpr int MW = (kMT + 1) / 2;                   // row tiles per wave (the odd group may have one less)
  bf16x8 w[kKS];
#pragma unroll
  for (int ks = 0; ks < kKS; ++ks) w[ks] = *reinterpret_cast<const bf16x8*>(wlane + (size_t)ks * (64 * 8));
  const char* pa[MW];
  f32x16 acc[MW];
#pragma unroll
  for (int i = 0; i < MW; ++i) {
    const int slot = min(32 * ((wave >> 1) + 2 * i) + li, kCP - 1);
    pa[i] = reinterpret_cast<const char*>(s_in) + ((2 * (slot / kCW)) * kIW + 2 * (slot % kCW)) * 8 + hf * 16;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  }
  static_for<kKS>([&](auto ks_) {
    constexpr int ks = decltype(ks_)::value;
    constexpr int ky = ks >> 1, h = ks & 1;
    // k = ky * 32 + 16 h + 8 hf + (0 .. 7) = taps kx = 4 h + 2 hf, + 1 (4 channels each) of kernel row ky
#pragma unroll
    for (int i = 0; i < MW; ++i) {
      if ((wave >> 1) + 2 * i < kMT) {                // (wave-uniform)
        bf16x8 a = *reinterpret_cast<const bf16x8*>(pa[i] + (ky * kIW + 4 * h) * 8);
        if constexpr (h == 1) {
          // lanes of the upper k-group hold taps 6 and 7: the padded tap's weights are zeros, but 0 x (a non-finite neighbour
          // pixel) would be NaN -- a pixel must reach exactly the windows that contain it: blank those four elements
          typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
          u32x4 u = __builtin_bit_cast(u32x4, a);
          u.z = hf ? 0u : u.z;
          u.w = hf ? 0u : u.w;
          a = __builtin_bit_cast(bf16x8, u);
        }
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, w[ks], acc[i], 0, 0, 0);
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
        // (the convolution as bf16 stored it, then shift + ReLU in fp32, rounded)
        s_conv[s2 * kConvPitch + nt * 32 + li] =
            valid ? (unsigned short)(pk_bf16(egtr_relu(rbf(acc[i][r]) + bz), 0.f) & 0xffffu) : (unsigned short)0;
      }
    }
  }
  __syncthreads();

  // pool: item = (pooled pixel, channel octet): 32 x 8 = one per thread
  for (int it = tid; it < kPH * kPW * 8; it += 256) {
    const int c8 = it & 7, pp = it >> 3;
    const int pyl = pp / kPW, pxl = pp % kPW;
    const int py = py0 + pyl, px = px0 + pxl;
    if (py < A.Hp && px < A.Wp) {
      float mx[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const uint4 u = *reinterpret_cast<const uint4*>(&s_conv[((2 * pyl + dy) * kCW + 2 * pxl + dx) * kConvPitch + 8 * c8]);
          const float v[8] = {__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                              __uint_as_float(u.y & 0xffff0000u), __uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u),
                              __uint_as_float(u.w << 16), __uint_as_float(u.w & 0xffff0000u)};
#pragma unroll
          for (int k = 0; k < 8; ++k) mx[k] = (v[k] > mx[k] || v[k] != v[k]) ? v[k] : mx[k];   // NaN propagates as in torch
        }
      *reinterpret_cast<uint4*>(A.y + (((size_t)b * A.Hp + py) * A.Wp + px) * 64 + 8 * c8) =
          make_uint4(pk_bf16(mx[0], mx[1]), pk_bf16(mx[2], mx[3]), pk_bf16(mx[4], mx[5]), pk_bf16(mx[6], mx[7]));
    }
  }
}

}  // namespace

extern "C" int egtr_stem_conv7x7_pool_bf16(egtr_stream_t stream, const uint16_t* x, const uint16_t* w_packed, const float* bias,
                                           uint16_t* y, int B, int H, int W) {
  if (!x || !w_packed || !bias || !y || B <= 0 || H <= 0 || W <= 0) return EGTR_E_ARG;
  if ((reinterpret_cast<uintptr_t>(y) & 15) || (reinterpret_cast<uintptr_t>(w_packed) & 15)) return EGTR_E_UNSUPPORTED;
  StemArgs A{x, w_packed, bias, y, B, H, W, 0, 0, 0, 0, 0, 0};
  A.Hc = (H - 1) / 2 + 1;
  A.Wc = (W - 1) / 2 + 1;
  A.Hp = (A.Hc - 1) / 2 + 1;
  A.Wp = (A.Wc - 1) / 2 + 1;
  A.tiles_x = (A.Wp + kPW - 1) / kPW;
  A.tiles_y = (A.Hp + kPH - 1) / kPH;
  const long long wgs = (long long)B * A.tiles_x * A.tiles_y;
  if (wgs >= (1ll << 31)) return EGTR_E_UNSUPPORTED;
  hipLaunchKernelGGL(stem_bf16_kernel, dim3((unsigned)wgs), dim3(256), 0, static_cast<hipStream_t>(stream), A);
  return egtr_check_launch();
}
