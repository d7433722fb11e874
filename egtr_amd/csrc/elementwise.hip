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
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
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
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
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
    if (relu) v = fmaxf(v, 0.f);
    y[i] = v;
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
  return v;
}

// rows x 256, one wave per row, 4 rows per workgroup
__global__ __launch_bounds__(256) void add_layernorm_256(const float* __restrict__ x, const float* __restrict__ res,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ y,
                                                         int rows, float eps) {
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
  reinterpret_cast<float4*>(y + (size_t)row * 256)[lane] =
      make_float4(dx * rstd * g.x + b.x, dy * rstd * g.y + b.y, dz * rstd * g.z + b.z, dw * rstd * g.w + b.w);
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

extern "C" int egtr_bias_act_nchw_f32(egtr_stream_t stream, const float* x, const float* bias, const float* residual,
                                      float* y, int N, int C, int HW, int relu) {
  if (!x || !bias || !y) return EGTR_E_ARG;
  if (N <= 0 || C <= 0 || HW <= 0) return EGTR_E_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long n = (long long)N * C * HW;
  const bool aligned = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) |
                         reinterpret_cast<uintptr_t>(residual)) & 15) == 0;
  if (HW % 4 == 0 && aligned) {
    const long long n4 = n / 4;
    const int blocks = (int)std::min<long long>((n4 + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(bias_act_nchw_vec4, dim3(blocks), dim3(256), 0, st, x, bias, residual, y, n4, C, HW / 4, relu);
  } else if (n % 4 == 0 && aligned && HW >= 4 && n < (1ll << 25)) {
    const long long n4 = n / 4;
    const int blocks = (int)std::min<long long>((n4 + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(bias_act_nchw_flat4, dim3(blocks), dim3(256), 0, st, x, bias, residual, y, n4, C, HW, relu);
  } else {
    const int blocks = (int)std::min<long long>((n + 255) / 256, 256 * 16);
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
                     residual, gamma, beta, y, rows, eps);
  return egtr_check_launch();
}
