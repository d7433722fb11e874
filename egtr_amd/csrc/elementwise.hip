// Fused memory-bound epilogues for gfx950 (HBM-bound: one read + one write of the activation, 16 B per lane).
//
//  egtr_bias_act_nchw_f32 : y = act(x + bias[c] (+ residual)) on NCHW activations -- the per-channel shift of a folded
//                           frozen BatchNorm, the bottleneck's residual add and the ReLU in ONE pass (PyTorch issues a
//                           broadcast add, an add and a clamp kernel: 3 reads + 3 writes).
//  egtr_add_layernorm_f32 : y = LayerNorm(x + residual) * gamma + beta over rows of 256 channels (d_model) -- the
//                           "residual + dropout(identity) + LayerNorm" of every encoder / decoder sub-layer
//                           (model/deformable_detr.py:1329-1330, 1343-1344, 1443-1444, 1465-1467, 1479-1480).
//                           One wavefront per row: 4 channels per lane, statistics by DPP/xor-shuffle reduction.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

namespace {

__global__ __launch_bounds__(256) void bias_act_nchw_vec4(const float* __restrict__ x, const float* __restrict__ bias,
                                                          const float* __restrict__ res, float* __restrict__ y,
                                                          long long n4, int C, int HW4, int relu) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)((i / HW4) % C);
    const float b = bias[c];
    float4 v = reinterpret_cast<const float4*>(x)[i];
    v.x += b; v.y += b; v.z += b; v.w += b;
    if (res != nullptr) {
      const float4 r = reinterpret_cast<const float4*>(res)[i];
      v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    if (relu) { v.x = egtr_relu(v.x); v.y = egtr_relu(v.y); v.z = egtr_relu(v.z); v.w = egtr_relu(v.w); }
    reinterpret_cast<float4*>(y)[i] = v;
  }
}

// HW not a multiple of 4 (e.g. 75 x 125): still 16 B per lane over the FLAT tensor; the 4 elements of a float4 may
// straddle a channel boundary, so the channel is resolved per element (exact float reciprocal division, n < 2^24*4).
__global__ __launch_bounds__(256) void bias_act_nchw_flat4(const float* __restrict__ x, const float* __restrict__ bias,
                                                           const float* __restrict__ res, float* __restrict__ y,
                                                           long long n4, int C, int HW, int relu) {
  const float inv = 1.0f / (float)HW;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const long long e0 = i * 4;
    long long p0 = (long long)((float)e0 * inv);          // approximate plane index, fix up exactly below
    while ((p0 + 1) * HW <= e0) ++p0;
    while (p0 * HW > e0) --p0;
    const int left = (int)((p0 + 1) * HW - e0);          // elements of this float4 still in plane p0
    const float b0 = bias[(int)(p0 % C)], b1 = bias[(int)((p0 + 1) % C)];
    float4 v = reinterpret_cast<const float4*>(x)[i];
    v.x += b0;
    v.y += (left > 1) ? b0 : b1;
    v.z += (left > 2) ? b0 : b1;
    v.w += (left > 3) ? b0 : b1;
    if (res != nullptr) {
      const float4 r = reinterpret_cast<const float4*>(res)[i];
      v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    if (relu) { v.x = egtr_relu(v.x); v.y = egtr_relu(v.y); v.z = egtr_relu(v.z); v.w = egtr_relu(v.w); }
    reinterpret_cast<float4*>(y)[i] = v;
  }
}

__global__ __launch_bounds__(256) void bias_act_nchw_scalar(const float* __restrict__ x, const float* __restrict__ bias,
                                                            const float* __restrict__ res, float* __restrict__ y,
                                                            long long n, int C, int HW, int relu) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)((i / HW) % C);
    float v = x[i] + bias[c];
    if (res != nullptr) v += res[i];
    if (relu) v = egtr_relu(v);
    y[i] = v;
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
  return v;
}

// rows x 256, one wave per row, 4 rows per workgroup (two rows per wave with both rows' loads requested up front was
// measured on the 12 537-row encoder calls: 9.5 vs 9.3 us per launch, no gain)
__global__ __launch_bounds__(256) void add_layernorm_256(const float* __restrict__ x, const float* __restrict__ res,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ y,
                                                         int rows, float eps, const float* __restrict__ pos,
                                                         int pos_rows, float* __restrict__ y_pos) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float4 v = reinterpret_cast<const float4*>(x + (size_t)row * 256)[lane];
  if (res != nullptr) {
    const float4 r = reinterpret_cast<const float4*>(res + (size_t)row * 256)[lane];
    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
  }
  const float mean = wave_sum(v.x + v.y + v.z + v.w) * (1.f / 256.f);
  const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
  const float var = wave_sum(dx * dx + dy * dy + dz * dz + dw * dw) * (1.f / 256.f);  // biased, as nn.LayerNorm
  const float rstd = rsqrtf(var + eps);
  const float4 g = reinterpret_cast<const float4*>(gamma)[lane], b = reinterpret_cast<const float4*>(beta)[lane];
  const float4 o = make_float4(dx * rstd * g.x + b.x, dy * rstd * g.y + b.y, dz * rstd * g.z + b.z, dw * rstd * g.w + b.w);
  reinterpret_cast<float4*>(y + (size_t)row * 256)[lane] = o;
  if (y_pos != nullptr) {  // the next sub-layer's "with_pos_embed" input: y + pos (pos rows repeat every pos_rows)
    const float4 pe = reinterpret_cast<const float4*>(pos + (size_t)(row % pos_rows) * 256)[lane];
    reinterpret_cast<float4*>(y_pos + (size_t)row * 256)[lane] = make_float4(o.x + pe.x, o.y + pe.y, o.z + pe.z, o.w + pe.w);
  }
}

// Backward of y = LayerNorm(x + res) * gamma + beta over rows of 256 channels: the statistics are recomputed from x + res
// (nothing but the inputs is kept from the forward), gs = d loss / d (x + res) is written, and the wave's share of
// d gamma = sum_rows gy * xhat, d beta = sum_rows gy stays in registers across its `rpw` consecutive rows; the four waves
// of a workgroup fold through LDS into partial[blockIdx][512] (fixed order; egtr_add_layernorm_backward_f32 sums the
// workgroup partials with colsum_final_f32).
__global__ __launch_bounds__(256) void add_layernorm_256_bwd(const float* __restrict__ x, const float* __restrict__ res,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ gy, float* __restrict__ gs,
                                                             float* __restrict__ partial, int rows, int rpw, float eps) {
  __shared__ float4 sm[2][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float4 gm = reinterpret_cast<const float4*>(gamma)[lane];
  float4 dg = make_float4(0.f, 0.f, 0.f, 0.f), db = dg;
  const int r0 = (blockIdx.x * 4 + wave) * rpw, r1 = min(r0 + rpw, rows);
#pragma unroll 2
  for (int row = r0; row < r1; ++row) {
    float4 v = reinterpret_cast<const float4*>(x + (size_t)row * 256)[lane];
    const float4 g = reinterpret_cast<const float4*>(gy + (size_t)row * 256)[lane];
    if (res != nullptr) {
      const float4 r = reinterpret_cast<const float4*>(res + (size_t)row * 256)[lane];
      v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    const float mean = wave_sum(v.x + v.y + v.z + v.w) * (1.f / 256.f);
    const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
    const float var = wave_sum(dx * dx + dy * dy + dz * dz + dw * dw) * (1.f / 256.f);
    const float rstd = rsqrtf(var + eps);
    const float hx = dx * rstd, hy = dy * rstd, hz = dz * rstd, hw = dw * rstd;      // xhat
    const float ax = g.x * gm.x, ay = g.y * gm.y, az = g.z * gm.z, aw = g.w * gm.w;  // gy * gamma
    const float c1 = wave_sum(ax + ay + az + aw) * (1.f / 256.f);
    const float c2 = wave_sum(ax * hx + ay * hy + az * hz + aw * hw) * (1.f / 256.f);
    reinterpret_cast<float4*>(gs + (size_t)row * 256)[lane] =
        make_float4(rstd * (ax - c1 - hx * c2), rstd * (ay - c1 - hy * c2), rstd * (az - c1 - hz * c2),
                    rstd * (aw - c1 - hw * c2));
    dg.x += g.x * hx; dg.y += g.y * hy; dg.z += g.z * hz; dg.w += g.w * hw;
    db.x += g.x; db.y += g.y; db.z += g.z; db.w += g.w;
  }
  sm[0][threadIdx.x] = dg;
  sm[1][threadIdx.x] = db;
  __syncthreads();
  if (threadIdx.x < 128) {   // 64 lanes x {d gamma, d beta}
    const int which = threadIdx.x >> 6;
    float4 a = sm[which][lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float4 o = sm[which][w * 64 + lane];
      a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
    }
    reinterpret_cast<float4*>(partial + (size_t)blockIdx.x * 512 + which * 256)[lane] = a;
  }
}

// ---- bf16 storage, fp32 arithmetic (the bf16 stress configuration; raw bfloat16 bits as uint16_t) ---------------------
__device__ __forceinline__ float bf2f(unsigned short u) { return __uint_as_float(((unsigned)u) << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {  // round to nearest even; NaN stays NaN
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
struct bf16x8 { uint4 v; };
__device__ __forceinline__ void unpack8(const uint4 v, float (&f)[8]) {
  const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(w[i] << 16);
    f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  unsigned w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = (unsigned)f2bf(f[2 * i]) | ((unsigned)f2bf(f[2 * i + 1]) << 16);
  return make_uint4(w[0], w[1], w[2], w[3]);
}

// rows x 256 bf16, one HALF-wave per row: a lane holds 8 channels (16 B), lanes 0..31 one row, lanes 32..63 the next (round 5:
// one row per wave with the upper half idle moved 3.3 TB/s; the statistics are 32-lane butterflies); gamma / beta bf16 (a
// model cast with .to(bfloat16) carries them in bf16), statistics in fp32.  Optionally y_pos = bf16(y + pos[row % pos_rows])
// (the next encoder layer's `hidden + pos`, dd:1041, rounded like the reference's bf16 add of the rounded y).
__device__ __forceinline__ float half_wave_sum(float v) {
#pragma unroll
  for (int o = 1; o < 32; o <<= 1) v += __shfl_xor(v, o);
  return v;
}

__global__ __launch_bounds__(256) void add_layernorm_256_bf16(const unsigned short* __restrict__ x,
                                                              const unsigned short* __restrict__ res,
                                                              const unsigned short* __restrict__ gamma,
                                                              const unsigned short* __restrict__ beta,
                                                              unsigned short* __restrict__ y, int rows, float eps,
                                                              const unsigned short* __restrict__ pos, int pos_rows,
                                                              unsigned short* __restrict__ y_pos) {
  const int lane = threadIdx.x & 31;
  const int row = blockIdx.x * 8 + (threadIdx.x >> 5);
  const bool act = row < rows;
  const int r_ = act ? row : rows - 1;
  float v[8];
  unpack8(reinterpret_cast<const uint4*>(x + (size_t)r_ * 256)[lane], v);
  if (res != nullptr) {
    float r[8];
    unpack8(reinterpret_cast<const uint4*>(res + (size_t)r_ * 256)[lane], r);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = bf2f(f2bf(v[i] + r[i]));  // the reference rounds the residual sum to bf16
  }
  float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (pos != nullptr) unpack8(reinterpret_cast<const uint4*>(pos + (size_t)(r_ % pos_rows) * 256)[lane], p);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) sum += v[i];
  const float mean = half_wave_sum(sum) * (1.f / 256.f);
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) sq += (v[i] - mean) * (v[i] - mean);
  const float rstd = rsqrtf(half_wave_sum(sq) * (1.f / 256.f) + eps);
  if (act) {
    float g[8], b[8], o[8];
    unpack8(reinterpret_cast<const uint4*>(gamma)[lane], g);
    unpack8(reinterpret_cast<const uint4*>(beta)[lane], b);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (v[i] - mean) * rstd * g[i] + b[i];
    const uint4 packed = pack8(o);
    reinterpret_cast<uint4*>(y + (size_t)row * 256)[lane] = packed;
    if (y_pos != nullptr) {
      float yr[8];
      unpack8(packed, yr);
#pragma unroll
      for (int i = 0; i < 8; ++i) yr[i] += p[i];
      reinterpret_cast<uint4*>(y_pos + (size_t)row * 256)[lane] = pack8(yr);
    }
  }
}

// y = act(x + bias[c] (+ residual)) on a bf16 NCHW activation, fp32 bias; 8 elements (16 B) per lane over the FLAT
// tensor, the channel resolved per element (a group of 8 may straddle one plane boundary; HW >= 8)
__global__ __launch_bounds__(256) void bias_act_nchw_flat8_bf16(const unsigned short* __restrict__ x,
                                                                const float* __restrict__ bias,
                                                                const unsigned short* __restrict__ res,
                                                                unsigned short* __restrict__ y, long long n8, int C,
                                                                int HW, int relu) {
  const double inv = 1.0 / (double)HW;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    const long long e0 = i * 8;
    long long p0 = (long long)((double)e0 * inv);
    while ((p0 + 1) * HW <= e0) ++p0;
    while (p0 * HW > e0) --p0;
    const int left = (int)((p0 + 1) * HW - e0);
    const float b0 = bias[(int)(p0 % C)], b1 = bias[(int)((p0 + 1) % C)];
    float v[8];
    unpack8(reinterpret_cast<const uint4*>(x)[i], v);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] += (k < left) ? b0 : b1;
    if (res != nullptr) {
      float r[8];
      unpack8(reinterpret_cast<const uint4*>(res)[i], r);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] += r[k];
    }
    if (relu) {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = egtr_relu(v[k]);
    }
    reinterpret_cast<uint4*>(y)[i] = pack8(v);
  }
}

// The same epilogue on a channels-last (NHWC) bf16 activation, i.e. a [rows, C] matrix: the 8 elements of a lane are 8
// consecutive channels (C % 8 == 0), their biases two 16-byte loads.  Hardware bf16 conversion.
__global__ __launch_bounds__(256) void bias_act_nhwc_flat8_bf16(const unsigned short* __restrict__ x,
                                                                const float* __restrict__ bias,
                                                                const unsigned short* __restrict__ res,
                                                                unsigned short* __restrict__ y, long long n8, int C, int relu) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    const int c0 = (int)((i * 8) % C);
    const float4 ba = *reinterpret_cast<const float4*>(bias + c0), bb = *reinterpret_cast<const float4*>(bias + c0 + 4);
    float v[8];
    unpack8(reinterpret_cast<const uint4*>(x)[i], v);
    v[0] += ba.x; v[1] += ba.y; v[2] += ba.z; v[3] += ba.w;
    v[4] += bb.x; v[5] += bb.y; v[6] += bb.z; v[7] += bb.w;
    if (res != nullptr) {
      float r[8];
      unpack8(reinterpret_cast<const uint4*>(res)[i], r);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] += r[k];
    }
    if (relu) {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = egtr_relu(v[k]);
    }
    unsigned w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      bf16x2 pr;
      pr[0] = (__bf16)v[2 * k];
      pr[1] = (__bf16)v[2 * k + 1];
      w[k] = __builtin_bit_cast(unsigned, pr);
    }
    reinterpret_cast<uint4*>(y)[i] = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

__global__ __launch_bounds__(256) void bias_act_nchw_scalar_bf16(const unsigned short* __restrict__ x,
                                                                 const float* __restrict__ bias,
                                                                 const unsigned short* __restrict__ res,
                                                                 unsigned short* __restrict__ y, long long n, int C,
                                                                 int HW, int relu) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)((i / HW) % C);
    float v = bf2f(x[i]) + bias[c];
    if (res != nullptr) v += bf2f(res[i]);
    if (relu) v = egtr_relu(v);
    y[i] = f2bf(v);
  }
}

// y[g, r, :] = keep[r] ? y[g, r, :] + bias[g, :] : 0   (the value projections of all decoder layers at once: bias add
// and the padding-mask select of deformable_detr.py:1050-1052 in one pass over G x R x C)
__global__ __launch_bounds__(256) void bias_mask_rows(float* __restrict__ y, const float* __restrict__ bias,
                                                      const unsigned char* __restrict__ keep, long long n4, int R,
                                                      int C4) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    const long long gr = i / C4;
    const int r = (int)(gr % R);
    const int g = (int)(gr / R);
    float4 v = reinterpret_cast<float4*>(y)[i];
    const float4 b = reinterpret_cast<const float4*>(bias)[(size_t)g * C4 + c4];
    const bool k = keep == nullptr || keep[r] != 0;
    v = k ? make_float4(v.x + b.x, v.y + b.y, v.z + b.z, v.w + b.w) : make_float4(0.f, 0.f, 0.f, 0.f);
    reinterpret_cast<float4*>(y)[i] = v;
  }
}

// ResNet stem epilogue: y = relu(maxpool3x3/s2/p1(x) + shift[c])  ==  maxpool(relu(x + shift[c]))  bit for bit (a
// per-channel constant and monotone rounding commute with max; padding is -inf).  x is the bias-free 7x7 convolution
// output [N, C, H, W] (77 MB at 600 x 1000), y [N, C, Ho, Wo]: one pass over x instead of PyTorch's max-pool kernel (which
// alone takes 24 us) plus an epilogue pass.  One thread per output pixel; a wave covers 64 consecutive output columns
// of one row, so its 9 loads walk three input rows with stride-2 lanes (every line is fetched once per wave).
__global__ __launch_bounds__(256) void bias_relu_maxpool3x3s2(const float* __restrict__ x, const float* __restrict__ bias,
                                                              float* __restrict__ y, int C, int H, int W, int Ho,
                                                              int Wo, long long planes) {
  const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
  const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
  const long long plane = blockIdx.z;
  if (ox >= Wo || oy >= Ho) return;
  const float* xp = x + plane * H * W;
  const int iy0 = 2 * oy - 1, ix0 = 2 * ox - 1;
  float m = -INFINITY;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int iy = iy0 + dy;
    if (iy < 0 || iy >= H) continue;
    const float* row = xp + (size_t)iy * W;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int ix = ix0 + dx;
      if (ix >= 0 && ix < W) {
        const float v = row[ix];
        m = (v > m || v != v) ? v : m;  // NaN propagates, as in torch max-pool
      }
    }
  }
  const float r = m + bias[(int)(plane % C)];
  y[(plane * Ho + oy) * Wo + ox] = (r != r) ? r : fmaxf(r, 0.f);  // relu(NaN) = NaN, as torch
}

// Box decoding of the detection head for every decoder level at once (model/egtr.py:286-305, with_box_refine = False):
//   reference_l = init_reference (l == 0) or inter_references[:, l-1];  r = inverse_sigmoid(reference_l)
//   (deformable_detr.py:658-662: x = clamp(x, 0, 1); log(max(x, eps) / max(1 - x, eps)));
//   box = sigmoid(delta + [r, 0, 0]) for 2-d references, sigmoid(delta + r) for 4-d ones.
// One thread per (b, l, n) row; replaces cat + 3 clamp + sub + div + log + add + cat + sigmoid launches on ~5 KB tensors.
// inter_ref == nullptr: every level uses init_ref (no iterative box refinement: the decoder hands the same reference
// points to every layer, dd:1903-1918 inactive).  logits_all != nullptr ([B, Ld, N, C]): the thread of the LAST level also
// writes node_cls[b, n] = argmax_c logits_all[b, Ld - 1, n, c] (first maximum on ties, NaN counts as the maximum -- as
// torch.argmax; egtr:405-413: the frequency-bias lookup of the relation head).
__global__ __launch_bounds__(256) void box_decode(const float* __restrict__ delta, const float* __restrict__ init_ref,
                                                  const float* __restrict__ inter_ref, int B, int Ld, int N, int RD,
                                                  float eps, float* __restrict__ out,
                                                  const float* __restrict__ logits_all, int C,
                                                  long long* __restrict__ node_cls) {
  const int nb_decode = (B * Ld * N + 255) / 256;
  if ((int)blockIdx.x >= nb_decode) {
    // torch.argmax over the last decoder layer's class logits (model/egtr.py:401-403), one WAVE per query row: lanes stride
    // over the classes, then a butterfly; the first maximal index wins, a NaN beats every number (torch semantics).
    const int row = ((int)blockIdx.x - nb_decode) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B * N) return;
    const int b = row / N, n = row - b * N;
    const float* lg = logits_all + (((size_t)b * Ld + (Ld - 1)) * N + n) * C;
    float best = lane < C ? lg[lane] : 0.f;
    int arg = lane < C ? lane : 0x7fffffff;
    for (int c = lane + 64; c < C; c += 64) {
      const float v = lg[c];
      if (v > best || (v != v && best == best)) { best = v; arg = c; }
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const float v = __shfl_xor(best, o);
      const int a = __shfl_xor(arg, o);
      const bool vn = v != v, bn = best != best;
      const bool take = a != 0x7fffffff && (arg == 0x7fffffff || (vn && !bn) || (!bn && v > best) || ((vn == bn) && (vn || v == best) && a < arg));
      if (take) { best = v; arg = a; }
    }
    if (lane == 0) node_cls[row] = arg;
    return;
  }
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * Ld * N) return;
  const int n = i % N, l = (i / N) % Ld, b = i / (N * Ld);
  const float* rp = (l == 0 || inter_ref == nullptr) ? init_ref + ((size_t)b * N + n) * RD
                                                      : inter_ref + (((size_t)b * Ld + (l - 1)) * N + n) * RD;
  const float4 d = reinterpret_cast<const float4*>(delta)[i];
  float v[4] = {d.x, d.y, d.z, d.w};
  for (int k = 0; k < RD; ++k) {
    float x = rp[k];
    x = fminf(fmaxf(x, 0.f), 1.f);
    // torch.clamp propagates NaN; fmaxf / fminf do not
    if (rp[k] != rp[k]) x = rp[k];
    const float x1 = (x != x) ? x : fmaxf(x, eps), x2 = (x != x) ? x : fmaxf(1.f - x, eps);
    v[k] += logf(x1 / x2);
  }
  float4 o;
  o.x = 1.f / (1.f + expf(-v[0]));
  o.y = 1.f / (1.f + expf(-v[1]));
  o.z = 1.f / (1.f + expf(-v[2]));
  o.w = 1.f / (1.f + expf(-v[3]));
  reinterpret_cast<float4*>(out)[i] = o;
}

// Sine position embedding (DeformableDetrSinePositionEmbedding, normalize=True; model/deformable_detr.py:850-876) from
// the two cumulative sums of the mask: out[b, c, y, x], c < E: sin/cos((y_embed-0.5)/(y_last+eps)*scale / dim_t[c]),
// c >= E: the same with x_embed; even channel index -> sin, odd -> cos.  One thread per (b, y, x, channel pair).
__global__ __launch_bounds__(256) void sine_pos_embed(const float* __restrict__ y_embed, const float* __restrict__ x_embed,
                                                      const float* __restrict__ dim_t, float* __restrict__ out, int B,
                                                      int H, int Wd, int E, float scale, float eps) {
  const long long n = (long long)B * E * H * Wd;  // (b, channel-pair-or-axis slot, y, x): E/2 pairs x 2 axes = E slots
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % Wd);
    long long t = i / Wd;
    const int y = (int)(t % H);
    t /= H;
    const int slot = (int)(t % E);
    const int b = (int)(t / E);
    const int axis = slot >= E / 2;       // 0: y half (channels 0..E-1), 1: x half (channels E..2E-1)
    const int pair = axis ? slot - E / 2 : slot;
    const size_t pix = ((size_t)b * H + y) * Wd + x;
    float e, last;
    if (axis == 0) {
      e = y_embed[pix];
      last = y_embed[((size_t)b * H + (H - 1)) * Wd + x];
    } else {
      e = x_embed[pix];
      last = x_embed[((size_t)b * H + y) * Wd + (Wd - 1)];
    }
    const float v = (e - 0.5f) / (last + eps) * scale;
    const float a0 = v / dim_t[2 * pair], a1 = v / dim_t[2 * pair + 1];
    const size_t plane = (size_t)H * Wd;
    float* o = out + ((size_t)b * 2 * E + (size_t)axis * E + 2 * pair) * plane + (size_t)y * Wd + x;
    o[0] = sinf(a0);
    o[plane] = cosf(a1);
  }
}


// ---- level geometry: everything DeformableDetrModel.forward derives from pixel_mask alone ----------------------------
// (model/deformable_detr.py:2195-2278 + 1616-1648 + 850-876).  For every feature level l (H_l x W_l) and pixel (y, x):
//   mask_l[y, x]  = nearest-neighbour resize of pixel_mask (F.interpolate(mask.float(), size).bool(): source index
//                   min(floor(dst * in/out), in - 1) with a float scale)                                     -> mask_flat
//   y_embed, x_embed = cumulative sums of mask_l along y / x; normalised sine embedding of 2E channels
//                   (+ level_embed[l])                                                                        -> pos_flat
//   valid_ratios[b, l] = (sum_x mask_l[0, x] / W_l, sum_y mask_l[y, 0] / H_l)
//   reference_points[b, s, l', :] = ((x + 0.5) / (vr[b,l,0] W_l), (y + 0.5) / (vr[b,l,1] H_l)) * vr[b, l', :]
// replacing ~100 tiny PyTorch kernels per forward.  Two launches: (1) one thread per token resizes the mask (the only
// reads of the full-resolution int64 mask: S scattered 8-byte reads) into mask_flat bytes + the bit mask; (2) one
// workgroup = 8 consecutive tokens of the flattened level list of one image: 32 threads count the mask column / row of a
// token in mask_flat (dense bytes, L1-resident), then thread c writes channel c of the 8 tokens.
struct LevelDims {
  int H[4], W[4], start[4];
};

template <typename MaskT>
__global__ __launch_bounds__(256) void level_mask_resize(const MaskT* __restrict__ pixel_mask, LevelDims ld, int L, int S,
                                                         int Hin, int Win, unsigned char* __restrict__ mask_flat,
                                                         unsigned* __restrict__ mask_bits) {
  const int b = blockIdx.y;
  const int s = blockIdx.x * 256 + threadIdx.x;
  int m = 0;
  if (s < S) {
    int l = 0;
    while (l + 1 < L && s >= ld.start[l + 1]) ++l;
    const int rel = s - ld.start[l];
    const int y = rel / ld.W[l], x = rel - y * ld.W[l];
    const float sh = (float)Hin / (float)ld.H[l], sw = (float)Win / (float)ld.W[l];
    const int sy = min((int)floorf((float)y * sh), Hin - 1), sx = min((int)floorf((float)x * sw), Win - 1);
    m = pixel_mask[((size_t)b * Hin + sy) * Win + sx] != 0 ? 1 : 0;
    mask_flat[(size_t)b * S + s] = (unsigned char)m;
  }
  if (mask_bits != nullptr) {
    const unsigned long long bal = __ballot(m != 0);
    const int lane = threadIdx.x & 63, s64 = s - lane;  // first token of this wave
    const int nwords = (S + 31) >> 5;
    if (lane < 2 && s64 + 32 * lane < S && (s64 >> 5) + lane < nwords)
      mask_bits[(size_t)b * nwords + (s64 >> 5) + lane] = (unsigned)(bal >> (32 * lane));
  }
}

// PosT = float, or unsigned short: bf16 position rows for a bf16 model -- bf16(bf16(sine) + level_embed), the two roundings
// of the reference's `position_embedding(..).to(dtype)` followed by its bf16 `+ level_embed[level]` (dd:2224, 2259);
// level_embed arrives widened to fp32 (exact).
template <typename PosT>
__global__ __launch_bounds__(256) void level_geometry(const unsigned char* __restrict__ mask_all,
                                                      const float* __restrict__ dim_t,
                                                      const float* __restrict__ level_embed, LevelDims ld, int L, int S,
                                                      int E, float scale, float eps, PosT* __restrict__ pos_flat,
                                                      float* __restrict__ valid_ratios, float* __restrict__ ref_points) {
  constexpr int TP = 8;
  __shared__ int s_cnt[8];          // [level][row-0 count, column-0 count]
  __shared__ float s_vr[8];         // valid ratios of this image [level][x, y]
  __shared__ int s_pix[TP][5];      // level, cy, toty, cx, totx
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int s0 = blockIdx.x * TP;
  const unsigned char* mk = mask_all + (size_t)b * S;
  if (tid < 8) s_cnt[tid] = 0;
  __syncthreads();
  // valid ratios: row 0 and column 0 of every level
  for (int l = 0; l < L; ++l) {
    const unsigned char* ml = mk + ld.start[l];
    int cw = 0, ch = 0;
    for (int x = tid; x < ld.W[l]; x += 256) cw += ml[x];
    for (int y = tid; y < ld.H[l]; y += 256) ch += ml[(size_t)y * ld.W[l]];
    // one LDS atomic per wave, not per lane (same-address atomics serialise)
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      cw += __shfl_xor(cw, o);
      ch += __shfl_xor(ch, o);
    }
    if ((tid & 63) == 0) {
      if (cw) atomicAdd(&s_cnt[2 * l], cw);
      if (ch) atomicAdd(&s_cnt[2 * l + 1], ch);
    }
  }
  // per-token column / row counts: 32 threads per token
  {
    const int p = tid >> 5, part = tid & 31;
    const int s = s0 + p;
    int l = 0, cy = 0, toty = 0, cx = 0, totx = 0;
    if (s < S) {
      while (l + 1 < L && s >= ld.start[l + 1]) ++l;
      const unsigned char* ml = mk + ld.start[l];
      const int rel = s - ld.start[l];
      const int Wl = ld.W[l], Hl = ld.H[l];
      const int y = rel / Wl, x = rel - y * Wl;
      for (int yy = part; yy < Hl; yy += 32) {
        const int v = ml[yy * Wl + x];
        toty += v;
        cy += (yy <= y) ? v : 0;
      }
      for (int xx = part; xx < Wl; xx += 32) {
        const int v = ml[y * Wl + xx];
        totx += v;
        cx += (xx <= x) ? v : 0;
      }
    }
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
      cy += __shfl_xor(cy, o);
      toty += __shfl_xor(toty, o);
      cx += __shfl_xor(cx, o);
      totx += __shfl_xor(totx, o);
    }
    if (part == 0) {
      s_pix[p][0] = l;
      s_pix[p][1] = cy;
      s_pix[p][2] = toty;
      s_pix[p][3] = cx;
      s_pix[p][4] = totx;
    }
  }
  __syncthreads();
  if (tid < 2 * L) {
    const int l = tid >> 1;
    const float vr = (float)s_cnt[tid] / (float)((tid & 1) ? ld.H[l] : ld.W[l]);
    s_vr[tid] = vr;
    if (blockIdx.x == 0) valid_ratios[(size_t)b * L * 2 + tid] = vr;
  }
  __syncthreads();
  // reference points: thread t < 8 TP -> token t / 8, (target level, xy) = t % 8
  if (tid < 8 * TP) {
    const int p = tid >> 3, k = tid & 7, s = s0 + p;
    if (s < S && k < 2 * L) {
      const int l = s_pix[p][0];
      const int rel = s - ld.start[l];
      const int y = rel / ld.W[l], x = rel - y * ld.W[l];
      const int lt = k >> 1, ax = k & 1;  // ax 0: x, 1: y
      const float base = ax ? ((float)y + 0.5f) / (s_vr[2 * l + 1] * (float)ld.H[l])
                            : ((float)x + 0.5f) / (s_vr[2 * l] * (float)ld.W[l]);
      ref_points[(((size_t)b * S + s) * L + lt) * 2 + ax] = base * s_vr[2 * lt + ax];
    }
  }
  // position embedding: channels (2k, 2k+1) of a half are sin / cos of the SAME argument (dim_t[2k] == dim_t[2k+1],
  // dd:864-865), so one thread evaluates sincosf once per (token, channel pair) and writes 8 bytes; E pairs per token
  // (E/2 of the y half, E/2 of the x half), tokens strided over the remaining threads.
  {
    const int npair = E;  // 2E channels = E pairs
    const int per = 256 / npair > 0 ? 256 / npair : 1;  // tokens handled side by side (E = 128: 2)
    for (int k = tid % npair; k < npair; k += 256) {
      const int c = 2 * k;  // first channel of the pair within the 2E-channel row
      const int axis = c >= E, i = axis ? c - E : c;
      const float dt = dim_t[i];
      for (int p = (npair <= 256 ? tid / npair : 0); p < TP; p += per) {
        const int s = s0 + p;
        if (s >= S) break;
        const int l = s_pix[p][0];
        const float e = (float)(axis ? s_pix[p][3] : s_pix[p][1]), last = (float)(axis ? s_pix[p][4] : s_pix[p][2]);
        const float v = (e - 0.5f) / (last + eps) * scale;
        const float a = v / dt;
        float sn, cs;
        sincosf(a, &sn, &cs);
        const float2 le = *reinterpret_cast<const float2*>(level_embed + l * 2 * E + c);
        if constexpr (sizeof(PosT) == 4) {
          *reinterpret_cast<float2*>(pos_flat + ((size_t)b * S + s) * (2 * E) + c) = make_float2(sn + le.x, cs + le.y);
        } else {
          const unsigned lo = f2bf(bf2f(f2bf(sn)) + le.x), hi = f2bf(bf2f(f2bf(cs)) + le.y);
          *reinterpret_cast<unsigned*>(pos_flat + ((size_t)b * S + s) * (2 * E) + c) = lo | (hi << 16);
        }
      }
    }
  }
}


// ---- input projection epilogue: conv bias + GroupNorm(32) + flatten(2).transpose(1, 2) + cat over the levels -----------
// (model/deformable_detr.py:2209-2262: nn.Sequential(Conv2d, GroupNorm) per level, then source.flatten(2).transpose(1, 2)
// and torch.cat).  x_l is the bias-free convolution output [B, C, H_l, W_l]; out is [B, S, C].
struct GnLevels {
  const void* x[4];      // fp32 or bf16 (raw bits) convolution outputs
  const float* conv_bias[4];
  const float* gamma[4];
  const float* beta[4];
  int hw[4], start[4], tile0[4];
};

// stats[(l * B + b) * G + g] = (mean, rstd) of (x + conv_bias) over the C/G channels x H_l W_l pixels of one group.
// One pass: per-thread fp32 partial sums of <= ~100 elements, combined in double (E[x^2] - mean^2 in double).
__device__ __forceinline__ float gn_load(const float* p) { return *p; }
__device__ __forceinline__ float gn_load(const unsigned short* p) { return bf2f(*p); }

template <typename T>
__global__ __launch_bounds__(1024) void gn_stats_levels(GnLevels P, int C, int G, float eps, float2* __restrict__ stats) {
  __shared__ double s_red[2][16];
  const int g = blockIdx.x, b = blockIdx.y, l = blockIdx.z;
  const int cpg = C / G, hw = P.hw[l];
  const T* x = static_cast<const T*>(P.x[l]) + ((size_t)b * C + (size_t)g * cpg) * hw;  // the group's channels are contiguous in NCHW
  float s1 = 0.f, s2 = 0.f;
  for (int c = 0; c < cpg; ++c) {
    const float cb = P.conv_bias[l][g * cpg + c];
    const T* xc = x + (size_t)c * hw;
    // 8 loads in flight per thread, accumulated in the original order (the loop was one dependent load -> add per
    // ~300 ns: 73 round trips for the 75 000 elements of a level-0 group)
    for (int i0 = threadIdx.x; i0 < hw; i0 += 8 * 1024) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + 1024 * u;
        v[u] = i < hw ? gn_load(xc + i) : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (i0 + 1024 * u < hw) {
          const float t = v[u] + cb;
          s1 += t;
          s2 += t * t;
        }
      }
    }
  }
  double d1 = (double)s1, d2 = (double)s2;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    d1 += __shfl_xor(d1, o);
    d2 += __shfl_xor(d2, o);
  }
  if ((threadIdx.x & 63) == 0) {
    s_red[0][threadIdx.x >> 6] = d1;
    s_red[1][threadIdx.x >> 6] = d2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t1 = 0.0, t2 = 0.0;
    for (int w = 0; w < 16; ++w) {
      t1 += s_red[0][w];
      t2 += s_red[1][w];
    }
    const double n = (double)cpg * hw, mean = t1 / n;
    const double var = fmax(t2 / n - mean * mean, 0.0);  // biased, as nn.GroupNorm
    stats[((size_t)l * gridDim.y + b) * G + g] = make_float2((float)mean, (float)(1.0 / sqrt(var + (double)eps)));
  }
}

// one workgroup = 32 pixels of one level x all C = 256 channels: coalesced NCHW reads (lanes along the pixels), LDS
// transpose, coalesced [.., C] writes (lanes along the channels)
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_flatten(GnLevels P, int L, int C, int G, int S, const float2* __restrict__ stats,
                                                        T* __restrict__ out) {
  __shared__ float s_t[32][257];
  const int b = blockIdx.y;
  int l = 0;
  while (l + 1 < L && (int)blockIdx.x >= P.tile0[l + 1]) ++l;
  const int p0 = ((int)blockIdx.x - P.tile0[l]) * 32, hw = P.hw[l];
  const int cpg = C / G;
  const int px = threadIdx.x & 31, cs = threadIdx.x >> 5;  // 8 channel sub-lanes
  const T* x = static_cast<const T*>(P.x[l]) + (size_t)b * C * hw;
  // C == 256 (checked by the launcher): 32 channels per thread, all 32 activation loads requested before the first use
  float xv[32];
  const bool inside = p0 + px < hw;
#pragma unroll
  for (int k = 0; k < 32; ++k) xv[k] = inside ? gn_load(x + (size_t)(cs + 8 * k) * hw + p0 + px) : 0.f;
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    const int c = cs + 8 * k;
    const float2 st = stats[((size_t)l * gridDim.y + b) * G + c / cpg];
    const float v = (xv[k] + P.conv_bias[l][c] - st.x) * st.y * P.gamma[l][c] + P.beta[l][c];
    s_t[px][c] = inside ? v : 0.f;
  }
  __syncthreads();
  for (int p = 0; p < 32; ++p) {
    if (p0 + p >= hw) break;
    if constexpr (sizeof(T) == 4) out[((size_t)b * S + P.start[l] + p0 + p) * C + threadIdx.x] = s_t[p][threadIdx.x];
    else out[((size_t)b * S + P.start[l] + p0 + p) * C + threadIdx.x] = f2bf(s_t[p][threadIdx.x]);
  }
}

}  // namespace

extern "C" int egtr_sine_pos_embed_f32(egtr_stream_t stream, const float* y_embed, const float* x_embed,
                                       const float* dim_t, float* out, int B, int H, int W, int E, float scale,
                                       float eps) {
  if (!y_embed || !x_embed || !dim_t || !out) return EGTR_E_ARG;
  if (B <= 0 || H <= 0 || W <= 0 || E <= 0 || (E & 1)) return EGTR_E_ARG;
  const long long n = (long long)B * E * H * W;
  const int blocks = (int)std::min<long long>((n + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(sine_pos_embed, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), y_embed, x_embed,
                     dim_t, out, B, H, W, E, scale, eps);
  return egtr_check_launch();
}

static int level_geometry_launch(egtr_stream_t stream, const void* pixel_mask, int mask_elem_size, const float* dim_t,
                                 const float* level_embed, const int* level_hw, int num_levels, int batch, int height,
                                 int width, int embed_dim, float scale, float eps, unsigned char* mask_flat,
                                 void* pos_flat, bool pos_bf16, float* valid_ratios, float* ref_points,
                                 unsigned* mask_bits) {
  if (!pixel_mask || !dim_t || !level_embed || !level_hw || !mask_flat || !pos_flat || !valid_ratios || !ref_points)
    return EGTR_E_ARG;
  if (num_levels < 1 || num_levels > 4 || batch <= 0 || height <= 0 || width <= 0 || embed_dim <= 0)
    return EGTR_E_ARG;
  if (mask_elem_size != 1 && mask_elem_size != 8) return EGTR_E_UNSUPPORTED;
  if (embed_dim & 1) return EGTR_E_UNSUPPORTED;  // channels (2k, 2k+1) share dim_t[2k] (sin / cos of one argument)
  LevelDims ld;
  int S = 0;
  for (int l = 0; l < 4; ++l) {
    const bool on = l < num_levels;
    ld.H[l] = on ? level_hw[2 * l] : 1;
    ld.W[l] = on ? level_hw[2 * l + 1] : 1;
    ld.start[l] = S;
    if (on) {
      if (ld.H[l] <= 0 || ld.W[l] <= 0) return EGTR_E_ARG;
      S += ld.H[l] * ld.W[l];
    }
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid1((unsigned)((S + 255) / 256), (unsigned)batch);
  if (mask_elem_size == 8)
    hipLaunchKernelGGL(level_mask_resize<long long>, grid1, dim3(256), 0, st, static_cast<const long long*>(pixel_mask),
                       ld, num_levels, S, height, width, mask_flat, mask_bits);
  else
    hipLaunchKernelGGL(level_mask_resize<unsigned char>, grid1, dim3(256), 0, st,
                       static_cast<const unsigned char*>(pixel_mask), ld, num_levels, S, height, width, mask_flat,
                       mask_bits);
  const dim3 grid((unsigned)((S + 7) / 8), (unsigned)batch);
  if (pos_bf16)
    hipLaunchKernelGGL(level_geometry<unsigned short>, grid, dim3(256), 0, st, mask_flat, dim_t, level_embed, ld,
                       num_levels, S, embed_dim, scale, eps, static_cast<unsigned short*>(pos_flat), valid_ratios,
                       ref_points);
  else
    hipLaunchKernelGGL(level_geometry<float>, grid, dim3(256), 0, st, mask_flat, dim_t, level_embed, ld, num_levels, S,
                       embed_dim, scale, eps, static_cast<float*>(pos_flat), valid_ratios, ref_points);
  return egtr_check_launch();
}

extern "C" int egtr_level_geometry_f32(egtr_stream_t stream, const void* pixel_mask, int mask_elem_size,
                                       const float* dim_t, const float* level_embed, const int* level_hw,
                                       int num_levels, int batch, int height, int width, int embed_dim, float scale,
                                       float eps, unsigned char* mask_flat, float* pos_flat, float* valid_ratios,
                                       float* ref_points, unsigned* mask_bits) {
  return level_geometry_launch(stream, pixel_mask, mask_elem_size, dim_t, level_embed, level_hw, num_levels, batch, height,
                               width, embed_dim, scale, eps, mask_flat, pos_flat, false, valid_ratios, ref_points,
                               mask_bits);
}

extern "C" int egtr_level_geometry_bf16(egtr_stream_t stream, const void* pixel_mask, int mask_elem_size,
                                        const float* dim_t, const float* level_embed, const int* level_hw,
                                        int num_levels, int batch, int height, int width, int embed_dim, float scale,
                                        float eps, unsigned char* mask_flat, uint16_t* pos_flat, float* valid_ratios,
                                        float* ref_points, unsigned* mask_bits) {
  return level_geometry_launch(stream, pixel_mask, mask_elem_size, dim_t, level_embed, level_hw, num_levels, batch, height,
                               width, embed_dim, scale, eps, mask_flat, pos_flat, true, valid_ratios, ref_points,
                               mask_bits);
}

static int groupnorm_flatten_launch(egtr_stream_t stream, int num_levels, const void* const* x,
                                    const float* const* conv_bias, const float* const* gamma, const float* const* beta,
                                    const int* level_hw, int batch, int channels, int num_groups, float eps, float* stats,
                                    void* out, bool bf16) {
  if (!x || !conv_bias || !gamma || !beta || !level_hw || !stats || !out) return EGTR_E_ARG;
  if (num_levels < 1 || num_levels > 4 || batch <= 0 || num_groups <= 0) return EGTR_E_ARG;
  if (channels != 256 || channels % num_groups != 0) return EGTR_E_UNSUPPORTED;
  GnLevels P;
  int S = 0, tiles = 0;
  for (int l = 0; l < 4; ++l) {
    const int s = l < num_levels ? l : 0;
    if (!x[s] || !conv_bias[s] || !gamma[s] || !beta[s] || level_hw[2 * s] <= 0 || level_hw[2 * s + 1] <= 0)
      return EGTR_E_ARG;
    P.x[l] = x[s];
    P.conv_bias[l] = conv_bias[s];
    P.gamma[l] = gamma[s];
    P.beta[l] = beta[s];
    P.hw[l] = level_hw[2 * s] * level_hw[2 * s + 1];
    P.start[l] = S;
    P.tile0[l] = tiles;
    if (l < num_levels) {
      S += P.hw[l];
      tiles += (P.hw[l] + 31) / 32;
    }
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (bf16) {
    hipLaunchKernelGGL(gn_stats_levels<unsigned short>, dim3(num_groups, batch, num_levels), dim3(1024), 0, st, P, channels,
                       num_groups, eps, reinterpret_cast<float2*>(stats));
    hipLaunchKernelGGL(gn_apply_flatten<unsigned short>, dim3(tiles, batch), dim3(256), 0, st, P, num_levels, channels,
                       num_groups, S, reinterpret_cast<const float2*>(stats), static_cast<unsigned short*>(out));
  } else {
    hipLaunchKernelGGL(gn_stats_levels<float>, dim3(num_groups, batch, num_levels), dim3(1024), 0, st, P, channels,
                       num_groups, eps, reinterpret_cast<float2*>(stats));
    hipLaunchKernelGGL(gn_apply_flatten<float>, dim3(tiles, batch), dim3(256), 0, st, P, num_levels, channels, num_groups,
                       S, reinterpret_cast<const float2*>(stats), static_cast<float*>(out));
  }
  return egtr_check_launch();
}

// ---- the same epilogue for TOKEN-MAJOR bf16 projections (channels-last backbone): x_l is [B, H_l*W_l, 256], the bias-free
// output of the level's 1x1 convolution run as a plain GEMM.  With 32 groups of 8 channels a 16-byte chunk is exactly one
// (token, group): no transpose anywhere.  Two launches for all levels.
struct GnTokLevels {
  const void* x[4];        // bf16 bits or fp32
  const float* conv_bias[4];
  const float* gamma[4];
  const float* beta[4];
  int hw[4], start[4];
  long long chunk0[5];   // first (token, group) chunk of a level in the flat apply grid (per image)
  int cchunk0[5];        // first 256-token statistics chunk of a level
};

namespace {
// the 8 channels of (token, group) number `idx` of a [tokens, 256] matrix
__device__ __forceinline__ void load_group8(const unsigned short* x, size_t idx, float (&f)[8]) {
  unpack8(reinterpret_cast<const uint4*>(x)[idx], f);
}
__device__ __forceinline__ void load_group8(const float* x, size_t idx, float (&f)[8]) {
  const float4 a = reinterpret_cast<const float4*>(x)[2 * idx], b = reinterpret_cast<const float4*>(x)[2 * idx + 1];
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
// Statistics in two small launches (one workgroup per (image, level) took 103 us at bs 1: 293 dependent loads deep):
// (1) a workgroup sums 256 tokens of one level: thread = (group t & 31, token phase t >> 5), partial (sum, sum of squares) per
//     group in double; (2) one wave per (image, level) adds the level's partials in order and writes (mean, rstd).
constexpr int kGnTokChunk = 256;
template <typename T>
__global__ __launch_bounds__(256) void gn_partial_tokens(GnTokLevels P, int L, int chunks_total, double2* __restrict__ partial) {
  __shared__ double s_red[2][32][9];
  const int b = blockIdx.y;
  int l = 0;
  while (l + 1 < L && (int)blockIdx.x >= P.cchunk0[l + 1]) ++l;
  const int t_begin = ((int)blockIdx.x - P.cchunk0[l]) * kGnTokChunk, hw = P.hw[l];
  const int g = threadIdx.x & 31, ph = threadIdx.x >> 5;
  const T* x = static_cast<const T*>(P.x[l]) + (size_t)b * hw * 256;
  const float4 ba = *reinterpret_cast<const float4*>(P.conv_bias[l] + 8 * g), bb = *reinterpret_cast<const float4*>(P.conv_bias[l] + 8 * g + 4);
  const float cb[8] = {ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, bb.z, bb.w};
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int u0 = 0; u0 < kGnTokChunk / 8; u0 += 8) {
    float f[8][8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int t = t_begin + ph + 8 * (u0 + u);
      if (t < hw) load_group8(x, (size_t)t * 32 + g, f[u]);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int t = t_begin + ph + 8 * (u0 + u);
      if (t < hw) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float v = f[u][k] + cb[k];
          s1 += v;
          s2 += v * v;
        }
      }
    }
  }
  s_red[0][g][ph] = (double)s1;
  s_red[1][g][ph] = (double)s2;
  __syncthreads();
  if (threadIdx.x < 32) {
    double t1 = 0.0, t2 = 0.0;
    for (int w = 0; w < 8; ++w) {
      t1 += s_red[0][threadIdx.x][w];
      t2 += s_red[1][threadIdx.x][w];
    }
    partial[((size_t)b * chunks_total + blockIdx.x) * 32 + threadIdx.x] = make_double2(t1, t2);
  }
}

// 256 threads per (image, level): group g = thread & 31 is summed by the 8 threads thread >> 5 = 0 .. 7, each over every 8th
// chunk, then over the 8 in a fixed order through LDS (a single thread per group walked up to 37 dependent 16-byte loads: 10 us
// of pure latency for the bench's level 0).
__global__ __launch_bounds__(256) void gn_finalize_tokens(GnTokLevels P, int chunks_total, float eps,
                                                          const double2* __restrict__ partial, float2* __restrict__ stats) {
  __shared__ double2 s_part[8][32];
  const int b = blockIdx.x, l = blockIdx.y, B = gridDim.x, g = threadIdx.x & 31, part = threadIdx.x >> 5;
  double t1 = 0.0, t2 = 0.0;
  for (int c = P.cchunk0[l] + part; c < P.cchunk0[l + 1]; c += 8) {
    const double2 p = partial[((size_t)b * chunks_total + c) * 32 + g];
    t1 += p.x;
    t2 += p.y;
  }
  s_part[part][g] = make_double2(t1, t2);
  __syncthreads();
  if (part != 0) return;
  t1 = 0.0;
  t2 = 0.0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    t1 += s_part[k][g].x;
    t2 += s_part[k][g].y;
  }
  const double n = 8.0 * P.hw[l], mean = t1 / n;
  const double var = fmax(t2 / n - mean * mean, 0.0);  // biased, as nn.GroupNorm
  stats[((size_t)l * B + b) * 32 + g] = make_float2((float)mean, (float)(1.0 / sqrt(var + (double)eps)));
}

template <typename T>
__global__ __launch_bounds__(256) void gn_apply_tokens(GnTokLevels P, int L, int S, const float2* __restrict__ stats,
                                                       T* __restrict__ out) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  const int b = blockIdx.y, B = gridDim.y;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;      // 16-byte chunk of this image's [S, 256] output
  if (i >= (long long)S * 32) return;
  int l = 0;
  while (l + 1 < L && i >= P.chunk0[l + 1]) ++l;
  const long long rel = i - P.chunk0[l];
  const int g = (int)(rel & 31);
  const float2 st = stats[((size_t)l * B + b) * 32 + g];
  float f[8];
  load_group8(static_cast<const T*>(P.x[l]) + (size_t)b * P.hw[l] * 256, (size_t)rel, f);
  const float4 ca = *reinterpret_cast<const float4*>(P.conv_bias[l] + 8 * g), cb = *reinterpret_cast<const float4*>(P.conv_bias[l] + 8 * g + 4);
  const float4 ga = *reinterpret_cast<const float4*>(P.gamma[l] + 8 * g), gb = *reinterpret_cast<const float4*>(P.gamma[l] + 8 * g + 4);
  const float4 ea = *reinterpret_cast<const float4*>(P.beta[l] + 8 * g), eb = *reinterpret_cast<const float4*>(P.beta[l] + 8 * g + 4);
  const float c8[8] = {ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w};
  const float g8[8] = {ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, gb.z, gb.w};
  const float e8[8] = {ea.x, ea.y, ea.z, ea.w, eb.x, eb.y, eb.z, eb.w};
  float o[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = (f[k] + c8[k] - st.x) * st.y * g8[k] + e8[k];
  T* dst = out + ((size_t)b * S + P.start[l]) * 256;
  if constexpr (sizeof(T) == 2) {
    unsigned w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      bf16x2 pr;
      pr[0] = (__bf16)o[2 * k];
      pr[1] = (__bf16)o[2 * k + 1];
      w[k] = __builtin_bit_cast(unsigned, pr);
    }
    reinterpret_cast<uint4*>(dst)[rel] = make_uint4(w[0], w[1], w[2], w[3]);
  } else {
    reinterpret_cast<float4*>(dst)[2 * rel] = make_float4(o[0], o[1], o[2], o[3]);
    reinterpret_cast<float4*>(dst)[2 * rel + 1] = make_float4(o[4], o[5], o[6], o[7]);
  }
}
}  // namespace

static int gn_tokens_chunks(int num_levels, const int* level_tokens) {
  int c = 0;
  for (int l = 0; l < num_levels; ++l) c += (level_tokens[l] + kGnTokChunk - 1) / kGnTokChunk;
  return c;
}

extern "C" long long egtr_input_proj_groupnorm_tokens_workspace_floats(int num_levels, const int* level_tokens, int batch) {
  if (!level_tokens || num_levels < 1 || num_levels > 4 || batch <= 0) return 0;
  // (mean, rstd) per (level, image, group) + the double2 partials per (image, chunk, group), 16-byte aligned behind them
  return (long long)num_levels * batch * 64 + (long long)batch * gn_tokens_chunks(num_levels, level_tokens) * 128;
}

static int groupnorm_tokens_launch(egtr_stream_t stream, int num_levels, const void* const* x, const float* const* conv_bias,
                                   const float* const* gamma, const float* const* beta, const int* level_tokens, int batch,
                                   int channels, int num_groups, float eps, float* stats, void* out, bool bf16) {
  if (!x || !conv_bias || !gamma || !beta || !level_tokens || !stats || !out) return EGTR_E_ARG;
  if (num_levels < 1 || num_levels > 4 || batch <= 0) return EGTR_E_ARG;
  if (channels != 256 || num_groups != 32) return EGTR_E_UNSUPPORTED;
  GnTokLevels P;
  int S = 0, cc = 0;
  for (int l = 0; l < 4; ++l) {
    const int s = l < num_levels ? l : 0;
    if (!x[s] || !conv_bias[s] || !gamma[s] || !beta[s] || level_tokens[s] <= 0) return EGTR_E_ARG;
    if (reinterpret_cast<uintptr_t>(x[s]) & 15) return EGTR_E_UNSUPPORTED;
    P.x[l] = x[s];
    P.conv_bias[l] = conv_bias[s];
    P.gamma[l] = gamma[s];
    P.beta[l] = beta[s];
    P.hw[l] = level_tokens[s];
    P.start[l] = S;
    P.chunk0[l] = (long long)S * 32;
    P.cchunk0[l] = cc;
    if (l < num_levels) {
      S += P.hw[l];
      cc += (P.hw[l] + kGnTokChunk - 1) / kGnTokChunk;
    }
  }
  P.chunk0[4] = (long long)S * 32;
  P.cchunk0[4] = cc;
  for (int l = num_levels; l < 4; ++l) {
    P.chunk0[l] = (long long)S * 32;
    P.cchunk0[l] = cc;
  }
  if ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(stats)) & 15) return EGTR_E_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  float2* st2 = reinterpret_cast<float2*>(stats);
  double2* partial = reinterpret_cast<double2*>(stats + (size_t)num_levels * batch * 64);
  const long long chunks = (long long)S * 32;
  const dim3 agrid((unsigned)((chunks + 255) / 256), batch);
  if (bf16) {
    hipLaunchKernelGGL(gn_partial_tokens<unsigned short>, dim3(cc, batch), dim3(256), 0, st, P, num_levels, cc, partial);
    hipLaunchKernelGGL(gn_finalize_tokens, dim3(batch, num_levels), dim3(256), 0, st, P, cc, eps, partial, st2);
    hipLaunchKernelGGL(gn_apply_tokens<unsigned short>, agrid, dim3(256), 0, st, P, num_levels, S, st2,
                       static_cast<unsigned short*>(out));
  } else {
    hipLaunchKernelGGL(gn_partial_tokens<float>, dim3(cc, batch), dim3(256), 0, st, P, num_levels, cc, partial);
    hipLaunchKernelGGL(gn_finalize_tokens, dim3(batch, num_levels), dim3(256), 0, st, P, cc, eps, partial, st2);
    hipLaunchKernelGGL(gn_apply_tokens<float>, agrid, dim3(256), 0, st, P, num_levels, S, st2, static_cast<float*>(out));
  }
  return egtr_check_launch();
}

extern "C" int egtr_input_proj_groupnorm_tokens_bf16(egtr_stream_t stream, int num_levels, const uint16_t* const* x,
                                                     const float* const* conv_bias, const float* const* gamma,
                                                     const float* const* beta, const int* level_tokens, int batch,
                                                     int channels, int num_groups, float eps, float* stats, uint16_t* out) {
  return groupnorm_tokens_launch(stream, num_levels, reinterpret_cast<const void* const*>(x), conv_bias, gamma, beta,
                                 level_tokens, batch, channels, num_groups, eps, stats, out, true);
}

extern "C" int egtr_input_proj_groupnorm_tokens_f32(egtr_stream_t stream, int num_levels, const float* const* x,
                                                    const float* const* conv_bias, const float* const* gamma,
                                                    const float* const* beta, const int* level_tokens, int batch,
                                                    int channels, int num_groups, float eps, float* stats, float* out) {
  return groupnorm_tokens_launch(stream, num_levels, reinterpret_cast<const void* const*>(x), conv_bias, gamma, beta,
                                 level_tokens, batch, channels, num_groups, eps, stats, out, false);
}

extern "C" int egtr_input_proj_groupnorm_flatten_f32(egtr_stream_t stream, int num_levels, const float* const* x,
                                                     const float* const* conv_bias, const float* const* gamma,
                                                     const float* const* beta, const int* level_hw, int batch,
                                                     int channels, int num_groups, float eps, float* stats,
                                                     float* out) {
  return groupnorm_flatten_launch(stream, num_levels, reinterpret_cast<const void* const*>(x), conv_bias, gamma, beta,
                                  level_hw, batch, channels, num_groups, eps, stats, out, false);
}

extern "C" int egtr_input_proj_groupnorm_flatten_bf16(egtr_stream_t stream, int num_levels, const uint16_t* const* x,
                                                      const float* const* conv_bias, const float* const* gamma,
                                                      const float* const* beta, const int* level_hw, int batch,
                                                      int channels, int num_groups, float eps, float* stats,
                                                      uint16_t* out) {
  return groupnorm_flatten_launch(stream, num_levels, reinterpret_cast<const void* const*>(x), conv_bias, gamma, beta,
                                  level_hw, batch, channels, num_groups, eps, stats, out, true);
}

// ---- many weight tensors x per-row scales in ONE launch ------------------------------------------------------------------
// out_t[r, :] = w_t[r, :] * scale_t[r] for up to 64 tensors (the frozen-BN scale riding on every trainable convolution weight
// of the backbone, and the same product on the weight gradients: 42 + 42 aten::mul launches per train step otherwise).
struct ScaleRowsArgs {
  const float* w[64];
  const float* scale[64];
  float* out[64];
  int rows[64], cols[64], blk0[65];
  int n;
};

namespace {
__global__ __launch_bounds__(256) void scale_rows_multi(ScaleRowsArgs A) {
  int t = 0;
  while (t + 1 < A.n && (int)blockIdx.x >= A.blk0[t + 1]) ++t;
  const int cols = A.cols[t];
  const long long n4 = (long long)A.rows[t] * cols / 4;
  const float4* w = reinterpret_cast<const float4*>(A.w[t]);
  float4* o = reinterpret_cast<float4*>(A.out[t]);
  const float* sc = A.scale[t];
  const long long base = (long long)((int)blockIdx.x - A.blk0[t]) * 1024;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long long i = base + u * 256 + threadIdx.x;
    if (i < n4) {
      const float s = sc[(i * 4) / cols];     // cols % 4 == 0: the four elements share a row
      const float4 v = w[i];
      o[i] = make_float4(v.x * s, v.y * s, v.z * s, v.w * s);
    }
  }
}
}  // namespace

extern "C" int egtr_scale_rows_multi_f32(egtr_stream_t stream, int n, const float* const* w, const float* const* scale,
                                         float* const* out, const int* rows, const int* cols) {
  if (!w || !scale || !out || !rows || !cols) return EGTR_E_ARG;
  if (n <= 0 || n > 64) return EGTR_E_UNSUPPORTED;
  ScaleRowsArgs A;
  int blocks = 0;
  for (int t = 0; t < n; ++t) {
    if (!w[t] || !scale[t] || !out[t] || rows[t] <= 0 || cols[t] <= 0) return EGTR_E_ARG;
    if (cols[t] % 4 != 0) return EGTR_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(w[t]) | reinterpret_cast<uintptr_t>(out[t])) & 15) return EGTR_E_UNSUPPORTED;
    A.w[t] = w[t];
    A.scale[t] = scale[t];
    A.out[t] = out[t];
    A.rows[t] = rows[t];
    A.cols[t] = cols[t];
    A.blk0[t] = blocks;
    const long long n4 = (long long)rows[t] * cols[t] / 4;
    blocks += (int)((n4 + 1023) / 1024);
  }
  A.blk0[n] = blocks;
  A.n = n;
  hipLaunchKernelGGL(scale_rows_multi, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), A);
  return egtr_check_launch();
}

extern "C" int egtr_bias_act_nchw_f32(egtr_stream_t stream, const float* x, const float* bias, const float* residual,
                                      float* y, int N, int C, int HW, int relu) {
  if (!x || !bias || !y) return EGTR_E_ARG;
  if (N <= 0 || C <= 0 || HW <= 0) return EGTR_E_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long n = (long long)N * C * HW;
  const bool aligned = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) |
                         reinterpret_cast<uintptr_t>(residual)) & 15) == 0;
  // grid-stride over at most 8 workgroups per CU (measured: 2048 / 4096 / 8192 / one element per thread within 4 %, 2048
  // marginally best; a one-workgroup-per-plane kernel without the index division was 5-13 % slower on every ResNet shape)
  const long long cap = 256 * 8;
  if (HW % 4 == 0 && aligned) {
    const long long n4 = n / 4;
    const int blocks = (int)std::min<long long>((n4 + 255) / 256, cap);
    hipLaunchKernelGGL(bias_act_nchw_vec4, dim3(blocks), dim3(256), 0, st, x, bias, residual, y, n4, C, HW / 4, relu);
  } else if (n % 4 == 0 && aligned && HW >= 4 && n < (1ll << 25)) {
    const long long n4 = n / 4;
    const int blocks = (int)std::min<long long>((n4 + 255) / 256, cap);
    hipLaunchKernelGGL(bias_act_nchw_flat4, dim3(blocks), dim3(256), 0, st, x, bias, residual, y, n4, C, HW, relu);
  } else {
    const int blocks = (int)std::min<long long>((n + 255) / 256, cap);
    hipLaunchKernelGGL(bias_act_nchw_scalar, dim3(blocks), dim3(256), 0, st, x, bias, residual, y, n, C, HW, relu);
  }
  return egtr_check_launch();
}

extern "C" int egtr_add_layernorm_f32(egtr_stream_t stream, const float* x, const float* residual, const float* gamma,
                                      const float* beta, float* y, int rows, int dim, float eps) {
  if (!x || !gamma || !beta || !y) return EGTR_E_ARG;
  if (rows <= 0) return EGTR_E_ARG;
  if (dim != 256) return EGTR_E_UNSUPPORTED;
  hipLaunchKernelGGL(add_layernorm_256, dim3((rows + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                     residual, gamma, beta, y, rows, eps, (const float*)nullptr, 1, (float*)nullptr);
  return egtr_check_launch();
}

extern "C" int egtr_add_layernorm_pos_f32(egtr_stream_t stream, const float* x, const float* residual,
                                          const float* gamma, const float* beta, float* y, int rows, int dim,
                                          float eps, const float* pos, int pos_rows, float* y_plus_pos) {
  if (!x || !gamma || !beta || !y || !pos || !y_plus_pos) return EGTR_E_ARG;
  if (rows <= 0 || pos_rows <= 0) return EGTR_E_ARG;
  if (dim != 256) return EGTR_E_UNSUPPORTED;
  hipLaunchKernelGGL(add_layernorm_256, dim3((rows + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                     residual, gamma, beta, y, rows, eps, pos, pos_rows, y_plus_pos);
  return egtr_check_launch();
}

extern "C" int egtr_bias_relu_maxpool3x3s2_f32(egtr_stream_t stream, const float* x, const float* bias, float* y, int N,
                                               int C, int H, int W) {
  if (!x || !bias || !y) return EGTR_E_ARG;
  if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return EGTR_E_ARG;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;  // floor((H + 2 - 3) / 2) + 1
  const long long planes = (long long)N * C;
  if (planes > 65535 || (Ho + 3) / 4 > 65535) return EGTR_E_UNSUPPORTED;
  const dim3 grid((unsigned)((Wo + 63) / 64), (unsigned)((Ho + 3) / 4), (unsigned)planes);
  hipLaunchKernelGGL(bias_relu_maxpool3x3s2, grid, dim3(256), 0, static_cast<hipStream_t>(stream), x, bias, y, C, H, W, Ho,
                     Wo, planes);
  return egtr_check_launch();
}

extern "C" int egtr_box_decode_argmax_f32(egtr_stream_t stream, const float* delta, const float* init_reference,
                                          const float* inter_references, int batch, int num_levels, int num_query,
                                          int ref_dim, float eps, float* boxes, const float* logits_all,
                                          int num_classes, int64_t* node_cls) {
  if (!delta || !init_reference || !boxes) return EGTR_E_ARG;
  if (batch <= 0 || num_levels <= 0 || num_query <= 0) return EGTR_E_ARG;
  if (logits_all && (!node_cls || num_classes <= 0)) return EGTR_E_ARG;
  if (ref_dim != 2 && ref_dim != 4) return EGTR_E_UNSUPPORTED;
  const long long n = (long long)batch * num_levels * num_query;
  if (n >= (1ll << 31)) return EGTR_E_UNSUPPORTED;
  // the decode blocks, then (with logits) one wave per query row for the class argmax
  const long long blocks = (n + 255) / 256 + (logits_all ? ((long long)batch * num_query + 3) / 4 : 0);
  hipLaunchKernelGGL(box_decode, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), delta,
                     init_reference, inter_references, batch, num_levels, num_query, ref_dim, eps, boxes, logits_all,
                     num_classes, reinterpret_cast<long long*>(node_cls));
  return egtr_check_launch();
}

extern "C" int egtr_bias_mask_rows_f32(egtr_stream_t stream, float* y, const float* bias, const unsigned char* keep,
                                       int groups, int rows, int cols) {
  if (!y || !bias) return EGTR_E_ARG;
  if (groups <= 0 || rows <= 0 || cols <= 0) return EGTR_E_ARG;
  if (cols % 4 != 0) return EGTR_E_UNSUPPORTED;
  const long long n4 = (long long)groups * rows * (cols / 4);
  const int blocks = (int)std::min<long long>((n4 + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(bias_mask_rows, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), y, bias, keep, n4,
                     rows, cols / 4);
  return egtr_check_launch();
}

extern "C" int egtr_bias_act_nchw_bf16(egtr_stream_t stream, const uint16_t* x, const float* bias,
                                       const uint16_t* residual, uint16_t* y, int N, int C, int HW, int relu) {
  if (!x || !bias || !y) return EGTR_E_ARG;
  if (N <= 0 || C <= 0 || HW <= 0) return EGTR_E_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long n = (long long)N * C * HW;
  const bool aligned = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) |
                         reinterpret_cast<uintptr_t>(residual)) & 15) == 0;
  if (n % 8 == 0 && aligned && HW >= 8) {
    const long long n8 = n / 8;
    const int blocks = (int)std::min<long long>((n8 + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(bias_act_nchw_flat8_bf16, dim3(blocks), dim3(256), 0, st, x, bias, residual, y, n8, C, HW, relu);
  } else {
    const int blocks = (int)std::min<long long>((n + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(bias_act_nchw_scalar_bf16, dim3(blocks), dim3(256), 0, st, x, bias, residual, y, n, C, HW, relu);
  }
  return egtr_check_launch();
}

namespace {
// fp32 twin: 4 consecutive channels per lane (C % 4 == 0)
__global__ __launch_bounds__(256) void bias_act_nhwc_flat4_f32(const float* __restrict__ x, const float* __restrict__ bias,
                                                               const float* __restrict__ res, float* __restrict__ y,
                                                               long long n4, int C, int relu) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const float4 b = *reinterpret_cast<const float4*>(bias + (int)((i * 4) % C));
    float4 v = reinterpret_cast<const float4*>(x)[i];
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    if (res != nullptr) {
      const float4 r = reinterpret_cast<const float4*>(res)[i];
      v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    if (relu) v = make_float4(egtr_relu(v.x), egtr_relu(v.y), egtr_relu(v.z), egtr_relu(v.w));
    reinterpret_cast<float4*>(y)[i] = v;
  }
}
}  // namespace

extern "C" int egtr_bias_act_nhwc_f32(egtr_stream_t stream, const float* x, const float* bias, const float* residual, float* y,
                                      long long rows, int C, int relu) {
  if (!x || !bias || !y) return EGTR_E_ARG;
  if (rows <= 0 || C <= 0) return EGTR_E_ARG;
  if (C % 4 != 0 || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual) |
                      reinterpret_cast<uintptr_t>(bias)) & 15))
    return EGTR_E_UNSUPPORTED;
  const long long n4 = rows * C / 4;
  const int blocks = (int)std::min<long long>((n4 + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(bias_act_nhwc_flat4_f32, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, bias, residual, y,
                     n4, C, relu);
  return egtr_check_launch();
}

extern "C" int egtr_bias_act_nhwc_bf16(egtr_stream_t stream, const uint16_t* x, const float* bias, const uint16_t* residual,
                                       uint16_t* y, long long rows, int C, int relu) {
  if (!x || !bias || !y) return EGTR_E_ARG;
  if (rows <= 0 || C <= 0) return EGTR_E_ARG;
  if (C % 8 != 0 || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual) |
                      reinterpret_cast<uintptr_t>(bias)) & 15))
    return EGTR_E_UNSUPPORTED;
  const long long n8 = rows * C / 8;
  const int blocks = (int)std::min<long long>((n8 + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(bias_act_nhwc_flat8_bf16, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, bias, residual,
                     y, n8, C, relu);
  return egtr_check_launch();
}

extern "C" int egtr_add_layernorm_bf16(egtr_stream_t stream, const uint16_t* x, const uint16_t* residual,
                                       const uint16_t* gamma, const uint16_t* beta, uint16_t* y, int rows, int dim,
                                       float eps) {
  if (!x || !gamma || !beta || !y) return EGTR_E_ARG;
  if (rows <= 0) return EGTR_E_ARG;
  if (dim != 256) return EGTR_E_UNSUPPORTED;
  hipLaunchKernelGGL(add_layernorm_256_bf16, dim3((rows + 7) / 8), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                     residual, gamma, beta, y, rows, eps, static_cast<const uint16_t*>(nullptr), 1,
                     static_cast<uint16_t*>(nullptr));
  return egtr_check_launch();
}

extern "C" int egtr_add_layernorm_pos_bf16(egtr_stream_t stream, const uint16_t* x, const uint16_t* residual,
                                           const uint16_t* gamma, const uint16_t* beta, uint16_t* y, int rows, int dim,
                                           float eps, const uint16_t* pos, int pos_rows, uint16_t* y_pos) {
  if (!x || !gamma || !beta || !y || !pos || !y_pos) return EGTR_E_ARG;
  if (rows <= 0 || pos_rows <= 0 || rows % pos_rows != 0) return EGTR_E_ARG;
  if (dim != 256) return EGTR_E_UNSUPPORTED;
  hipLaunchKernelGGL(add_layernorm_256_bf16, dim3((rows + 7) / 8), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                     residual, gamma, beta, y, rows, eps, pos, pos_rows, y_pos);
  return egtr_check_launch();
}


// ---- "clamp iff any element is inf / nan" of the encoder layers in training (model/deformable_detr.py:1346-1351) ------
// The reference branches on the host (two synchronisations per layer).  Here: one pass raises a device flag if any element
// is non-finite; the clamp pass and the masking pass of the backward return at once while the flag is clear -- which is
// every step of a healthy run -- so the states are neither copied nor re-read.
namespace {

__global__ __launch_bounds__(256) void any_nonfinite_f32(const float* __restrict__ x, long long n4, long long n,
                                                         int* __restrict__ flag) {
  bool bad = false;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    // finite <=> exponent field not all ones
    bad |= (__float_as_uint(v.x) & 0x7f800000u) == 0x7f800000u || (__float_as_uint(v.y) & 0x7f800000u) == 0x7f800000u ||
           (__float_as_uint(v.z) & 0x7f800000u) == 0x7f800000u || (__float_as_uint(v.w) & 0x7f800000u) == 0x7f800000u;
  }
  if (blockIdx.x == 0)
    for (long long i = 4 * n4 + threadIdx.x; i < n; i += blockDim.x)
      bad |= (__float_as_uint(x[i]) & 0x7f800000u) == 0x7f800000u;
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// MASK = false: x <- clamp(x, -c, c) in place (NaN stays NaN, as torch.clamp); MASK = true: g <- g * [|x| < c] (the
// gradient of that clamp).  Both are no-ops unless *flag != 0.
template <bool MASK>
__global__ __launch_bounds__(256) void clamp_if_flag_f32(float* __restrict__ t, const float* __restrict__ x, long long n,
                                                         const int* __restrict__ flag, float c) {
  if (*flag == 0) return;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    if (MASK) {
      const float v = x[i];
      if (!(fabsf(v) < c)) t[i] = 0.f;   // clamped (and NaN) elements pass no gradient, as torch's clamp backward
    } else {
      const float v = t[i];
      t[i] = v != v ? v : fminf(fmaxf(v, -c), c);
    }
  }
}

}  // namespace

extern "C" int egtr_any_nonfinite_f32(egtr_stream_t stream, const float* x, long long n, int* flag) {
  if (!x || !flag || n <= 0) return EGTR_E_ARG;
  if (reinterpret_cast<uintptr_t>(x) & 15) return EGTR_E_UNSUPPORTED;
  const long long n4 = n / 4;
  const int blocks = (int)std::min<long long>((n4 + 255) / 256 + 1, 2048);
  hipLaunchKernelGGL(any_nonfinite_f32, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, n4, n, flag);
  return egtr_check_launch();
}

extern "C" int egtr_clamp_if_flag_f32(egtr_stream_t stream, float* t, const float* x, long long n, const int* flag,
                                      float clamp_value, int mask_gradient) {
  if (!t || !flag || n <= 0 || (mask_gradient && !x)) return EGTR_E_ARG;
  const int blocks = (int)std::min<long long>((n + 255) / 256, 4096);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (mask_gradient)
    hipLaunchKernelGGL(clamp_if_flag_f32<true>, dim3(blocks), dim3(256), 0, st, t, x, n, flag, clamp_value);
  else
    hipLaunchKernelGGL(clamp_if_flag_f32<false>, dim3(blocks), dim3(256), 0, st, t, x, n, flag, clamp_value);
  return egtr_check_launch();
}

// ---- bias gradient of a token-sized linear layer: column sums of g [M, N] (optionally of g masked by y > 0, the ReLU
// backward, written out on the way) -- two passes, fixed summation order (bit-reproducible) --------------------------------
namespace {

constexpr int kCsRows = 64;

// One workgroup per 64-row slab.  LPR lanes cover a row in float4 columns (256 / LPR rows in flight per pass); the row
// lanes of a column are folded through LDS in a fixed order.  g_masked may alias g (an element is read and written by the
// same thread).
template <int LPR, bool MASK>
__global__ __launch_bounds__(256) void colsum_partial_v4_f32(const float* g, const float* __restrict__ y,
                                                             float* g_masked, float* __restrict__ partial,
                                                             int M, int N, const float* __restrict__ row_weight) {
  constexpr int RP = 256 / LPR;
  __shared__ float4 sm[256];
  const int lane = threadIdx.x % LPR, rl = threadIdx.x / LPR;
  const int r0 = blockIdx.x * kCsRows, r1 = min(r0 + kCsRows, M);
  const int n4 = N >> 2;
  for (int cb = 0; cb < n4; cb += LPR) {
    const int c4 = cb + lane;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 < n4) {
#pragma unroll 8
      for (int r = r0 + rl; r < r1; r += RP) {
        const size_t i = (size_t)r * n4 + c4;
        float4 v = reinterpret_cast<const float4*>(g)[i];
        if (MASK) {
          const float4 t = reinterpret_cast<const float4*>(y)[i];
          v.x = t.x > 0.f ? v.x : 0.f;
          v.y = t.y > 0.f ? v.y : 0.f;
          v.z = t.z > 0.f ? v.z : 0.f;
          v.w = t.w > 0.f ? v.w : 0.f;
          reinterpret_cast<float4*>(g_masked)[i] = v;
        }
        if (!MASK && row_weight != nullptr) {   // weighted column sum: sum_r w[r] g[r][c]
          const float w = row_weight[r];
          v.x *= w; v.y *= w; v.z *= w; v.w *= w;
        }
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
    }
    if (RP > 1) {
      sm[threadIdx.x] = s;
      __syncthreads();
      if (rl == 0) {
#pragma unroll
        for (int k = 1; k < RP; ++k) {
          const float4 o = sm[k * LPR + lane];
          s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
        }
      }
      __syncthreads();
    }
    if (rl == 0 && c4 < n4) reinterpret_cast<float4*>(partial)[(size_t)blockIdx.x * n4 + c4] = s;
  }
}

// N % 4 != 0: scalar columns
__global__ __launch_bounds__(256) void colsum_partial_f32(const float* g, const float* __restrict__ y,
                                                          float* g_masked, float* __restrict__ partial,
                                                          int M, int N) {
  const int r0 = blockIdx.x * kCsRows, r1 = min(r0 + kCsRows, M);
  for (int c = threadIdx.x; c < N; c += 256) {
    float s = 0.f;
    for (int r = r0; r < r1; ++r) {
      const size_t i = (size_t)r * N + c;
      float v = g[i];
      if (y != nullptr) {
        v = y[i] > 0.f ? v : 0.f;
        g_masked[i] = v;
      }
      s += v;
    }
    partial[(size_t)blockIdx.x * N + c] = s;
  }
}

// 16 columns per workgroup, 16 slab lanes per column (fixed order: lane-strided sums, then the 16 lanes in sequence)
__global__ __launch_bounds__(256) void colsum_final_f32(const float* __restrict__ partial, int chunks, int N,
                                                        float* __restrict__ out) {
  __shared__ float sm[256];
  const int cl = threadIdx.x & 15, kl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float s = 0.f;
  if (c < N) {
#pragma unroll 8
    for (int k = kl; k < chunks; k += 16) s += partial[(size_t)k * N + c];
  }
  sm[threadIdx.x] = s;
  __syncthreads();
  if (kl == 0 && c < N) {
#pragma unroll
    for (int k = 1; k < 16; ++k) s += sm[k * 16 + cl];
    out[c] = s;
  }
}

template <int LPR>
void launch_colsum_v4(hipStream_t st, int chunks, const float* g, const float* y, float* g_masked, float* partial, int M,
                      int N, const float* row_weight = nullptr) {
  if (y)
    hipLaunchKernelGGL((colsum_partial_v4_f32<LPR, true>), dim3(chunks), dim3(256), 0, st, g, y, g_masked, partial, M, N,
                       (const float*)nullptr);
  else
    hipLaunchKernelGGL((colsum_partial_v4_f32<LPR, false>), dim3(chunks), dim3(256), 0, st, g, y, g_masked, partial, M, N,
                       row_weight);
}

}  // namespace

extern "C" long long egtr_column_sum_workspace_floats(int M, int N) {
  if (M <= 0 || N <= 0) return 0;
  return (long long)((M + kCsRows - 1) / kCsRows) * N;
}

extern "C" int egtr_column_sum_f32(egtr_stream_t stream, const float* g, const float* relu_output, float* g_masked,
                                   float* workspace, float* out, int M, int N) {
  if (!g || !workspace || !out || M <= 0 || N <= 0) return EGTR_E_ARG;
  if ((relu_output != nullptr) != (g_masked != nullptr)) return EGTR_E_ARG;
  const int chunks = (M + kCsRows - 1) / kCsRows;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (!relu_output && M <= 2048) {   // object-query-sized inputs: the final pass alone, over the rows of g
    hipLaunchKernelGGL(colsum_final_f32, dim3((N + 15) / 16), dim3(256), 0, st, g, M, N, out);
    return egtr_check_launch();
  }
  const bool v4 = N % 4 == 0 && ((uintptr_t)g % 16) == 0 && (!relu_output || (((uintptr_t)relu_output | (uintptr_t)g_masked) % 16) == 0);
  const int n4 = N / 4;
  if (!v4)
    hipLaunchKernelGGL(colsum_partial_f32, dim3(chunks), dim3(256), 0, st, g, relu_output, g_masked, workspace, M, N);
  else if (n4 <= 32)
    launch_colsum_v4<32>(st, chunks, g, relu_output, g_masked, workspace, M, N);
  else if (n4 <= 64)
    launch_colsum_v4<64>(st, chunks, g, relu_output, g_masked, workspace, M, N);
  else if (n4 <= 128)
    launch_colsum_v4<128>(st, chunks, g, relu_output, g_masked, workspace, M, N);
  else
    launch_colsum_v4<256>(st, chunks, g, relu_output, g_masked, workspace, M, N);
  int rc = egtr_check_launch();
  if (rc != EGTR_OK) return rc;
  hipLaunchKernelGGL(colsum_final_f32, dim3((N + 15) / 16), dim3(256), 0, st, workspace, chunks, N, out);
  return egtr_check_launch();
}

// ---- backward of egtr_add_layernorm_f32 ------------------------------------------------------------------------------
namespace {
inline int ln_bwd_rows_per_wave(int rows) { return rows >= 16384 ? 8 : 1; }
}  // namespace

extern "C" long long egtr_add_layernorm_backward_workspace_floats(int rows) {
  if (rows <= 0) return 0;
  const int per_wg = 4 * ln_bwd_rows_per_wave(rows);
  return (long long)((rows + per_wg - 1) / per_wg) * 512;
}

extern "C" int egtr_add_layernorm_backward_f32(egtr_stream_t stream, const float* x, const float* residual,
                                               const float* gamma, const float* grad_y, float* grad_sum,
                                               float* workspace, float* grad_gamma_beta, int rows, int dim, float eps) {
  if (!x || !gamma || !grad_y || !grad_sum || !workspace || !grad_gamma_beta || rows <= 0) return EGTR_E_ARG;
  if (dim != 256) return EGTR_E_UNSUPPORTED;
  const int rpw = ln_bwd_rows_per_wave(rows), per_wg = 4 * rpw, wgs = (rows + per_wg - 1) / per_wg;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(add_layernorm_256_bwd, dim3(wgs), dim3(256), 0, st, x, residual, gamma, grad_y, grad_sum, workspace,
                     rows, rpw, eps);
  int rc = egtr_check_launch();
  if (rc != EGTR_OK) return rc;
  hipLaunchKernelGGL(colsum_final_f32, dim3(512 / 16), dim3(256), 0, st, workspace, wgs, 512, grad_gamma_beta);
  return egtr_check_launch();
}

// out [N] = sum_r row_weight[r] * g[r][:] -- the [1, M] x [M, N] product behind the weight gradient of a one-output linear
// layer (the relation head's connectivity output), which the vendor library serves as a 220 us GEMV at M = 160 000
extern "C" int egtr_weighted_column_sum_f32(egtr_stream_t stream, const float* g, const float* row_weight,
                                            float* workspace, float* out, int M, int N) {
  if (!g || !row_weight || !workspace || !out || M <= 0 || N <= 0) return EGTR_E_ARG;
  if (N % 4 != 0 || ((uintptr_t)g % 16) != 0) return EGTR_E_UNSUPPORTED;
  const int chunks = (M + kCsRows - 1) / kCsRows, n4 = N / 4;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n4 <= 32)
    launch_colsum_v4<32>(st, chunks, g, nullptr, nullptr, workspace, M, N, row_weight);
  else if (n4 <= 64)
    launch_colsum_v4<64>(st, chunks, g, nullptr, nullptr, workspace, M, N, row_weight);
  else if (n4 <= 128)
    launch_colsum_v4<128>(st, chunks, g, nullptr, nullptr, workspace, M, N, row_weight);
  else
    launch_colsum_v4<256>(st, chunks, g, nullptr, nullptr, workspace, M, N, row_weight);
  int rc = egtr_check_launch();
  if (rc != EGTR_OK) return rc;
  hipLaunchKernelGGL(colsum_final_f32, dim3((N + 15) / 16), dim3(256), 0, st, workspace, chunks, N, out);
  return egtr_check_launch();
}

// ---- pad_and_create_pixel_mask on the device (DeformableDetrFeatureExtractor, reference preprocessing at
// model/deformable_detr.py:270-385 via HF's DetrFeatureExtractor.pad_and_create_pixel_mask): B images [C, h_b, w_b] ->
// zero-padded batch [B, C, H, W] (top-left aligned) + int64 mask [B, H, W] (1 = real pixel), ONE launch.
namespace {
__global__ __launch_bounds__(256) void pad_batch_f32(const float* const* __restrict__ imgs, const int* __restrict__ hw,
                                                     int B, int C, int H, int W, float* __restrict__ out,
                                                     long long* __restrict__ mask) {
  const long long row = blockIdx.x;                 // (b, c, y), plus one extra "channel" per image for the mask rows
  const int y = (int)(row % H);
  const int c = (int)((row / H) % (C + 1));
  const int b = (int)(row / ((long long)H * (C + 1)));
  const int h = hw[2 * b], w = hw[2 * b + 1];
  if (c == C) {
    long long* m = mask + ((size_t)b * H + y) * W;
    for (int x = threadIdx.x; x < W; x += 256) m[x] = (y < h && x < w) ? 1 : 0;
    return;
  }
  float* o = out + (((size_t)b * C + c) * H + y) * W;
  const float* src = imgs[b] + ((size_t)c * h + y) * w;
  for (int x = threadIdx.x; x < W; x += 256) o[x] = (y < h && x < w) ? src[x] : 0.f;
}
}  // namespace

extern "C" int egtr_pad_batch_f32(egtr_stream_t stream, const float* const* images, const int* heights_widths, int batch,
                                  int channels, int H, int W, float* pixel_values, int64_t* pixel_mask) {
  if (!images || !heights_widths || !pixel_values || !pixel_mask || batch <= 0 || channels <= 0 || H <= 0 || W <= 0)
    return EGTR_E_ARG;
  const long long rows = (long long)batch * (channels + 1) * H;
  if (rows >= (1ll << 31)) return EGTR_E_UNSUPPORTED;
  hipLaunchKernelGGL(pad_batch_f32, dim3((unsigned)rows), dim3(256), 0, static_cast<hipStream_t>(stream), images,
                     heights_widths, batch, channels, H, W, pixel_values, reinterpret_cast<long long*>(pixel_mask));
  return egtr_check_launch();
}
