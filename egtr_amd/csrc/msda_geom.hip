// libegtr_hip.so -- the elementwise prologue of multi-scale deformable attention under autograd (training):
//   sampling_locations = reference_points + sampling_offsets / (W_l, H_l)                       (2-d reference points)
//                      = reference_xy + sampling_offsets / P * reference_wh * 0.5               (4-d reference boxes)
//   attention_weights  = softmax over the L * P samples of a head
// (model/deformable_detr.py:1055-1073) and its backward.  The reference issues softmax, a broadcast division, one or
// three broadcast multiplications and an addition (and their backward kernels) over [B, Lq, M, L, P, 2] tensors; here each
// direction is ONE pass: a thread owns one (query, head) -- 2 L P offsets and L P logits, contiguous in memory.
// The reference-point gradient (decoder: reference points are a learned function of the queries) is the sum over heads
// and points, folded across the M threads of a query with xor shuffles (M a power of two <= 64).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

namespace {

template <int L, int P, bool BOX>
__global__ __launch_bounds__(256) void msda_geometry_fwd(const float* __restrict__ off, long long ld_off,
                                                         const float* __restrict__ logits, long long ld_logits,
                                                         const float* __restrict__ ref,
                                                         const int64_t* __restrict__ shapes, float* __restrict__ loc,
                                                         float* __restrict__ probs, long long rows, int M) {
  constexpr int LP = L * P;
  const long long idx = blockIdx.x * 256ll + threadIdx.x;
  if (idx >= rows * M) return;
  const long long row = idx / M;
  const int m = (int)(idx - row * M);
  const float4* o4 = reinterpret_cast<const float4*>(off + row * ld_off + (size_t)m * LP * 2);
  const float4* l4 = reinterpret_cast<const float4*>(logits + row * ld_logits + (size_t)m * LP);
  float4* loc4 = reinterpret_cast<float4*>(loc + (size_t)idx * LP * 2);
  float4* p4 = reinterpret_cast<float4*>(probs + (size_t)idx * LP);
  float x[LP];
#pragma unroll
  for (int i = 0; i < LP / 4; ++i) {
    const float4 v = l4[i];
    x[4 * i] = v.x; x[4 * i + 1] = v.y; x[4 * i + 2] = v.z; x[4 * i + 3] = v.w;
  }
  float mx = x[0];
#pragma unroll
  for (int i = 1; i < LP; ++i) mx = fmaxf(mx, x[i]);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LP; ++i) {
    x[i] = expf(x[i] - mx);
    s += x[i];
  }
#pragma unroll
  for (int i = 0; i < LP / 4; ++i) p4[i] = make_float4(x[4 * i] / s, x[4 * i + 1] / s, x[4 * i + 2] / s, x[4 * i + 3] / s);
  const float* r = ref + row * (size_t)(L * (BOX ? 4 : 2));
#pragma unroll
  for (int l = 0; l < L; ++l) {
    float rx, ry, sx, sy;
    if (BOX) {
      const float4 b = reinterpret_cast<const float4*>(r)[l];
      rx = b.x; ry = b.y; sx = b.z; sy = b.w;
    } else {
      rx = r[2 * l]; ry = r[2 * l + 1];
      sx = (float)shapes[2 * l + 1]; sy = (float)shapes[2 * l];   // (W, H)
    }
#pragma unroll
    for (int q = 0; q < P / 2; ++q) {   // two points per 16-byte access
      const float4 o = o4[l * (P / 2) + q];
      float4 w;
      if (BOX) {
        w.x = rx + o.x / (float)P * sx * 0.5f; w.y = ry + o.y / (float)P * sy * 0.5f;
        w.z = rx + o.z / (float)P * sx * 0.5f; w.w = ry + o.w / (float)P * sy * 0.5f;
      } else {
        w.x = rx + o.x / sx; w.y = ry + o.y / sy; w.z = rx + o.z / sx; w.w = ry + o.w / sy;
      }
      loc4[l * (P / 2) + q] = w;
    }
  }
}

template <int L, int P, bool BOX>
__global__ __launch_bounds__(256) void msda_geometry_bwd(const float* __restrict__ g_loc, const float* __restrict__ g_prob,
                                                         const float* __restrict__ probs, const float* __restrict__ off,
                                                         long long ld_off, const float* __restrict__ ref,
                                                         const int64_t* __restrict__ shapes, float* __restrict__ g_off,
                                                         long long ld_goff, float* __restrict__ g_logits,
                                                         long long ld_glogits, float* __restrict__ g_ref,
                                                         long long rows, int M) {
  constexpr int LP = L * P;
  const long long idx = blockIdx.x * 256ll + threadIdx.x;
  const bool live = idx < rows * M;          // dead lanes still take part in the shuffles below
  const long long row = live ? idx / M : 0;
  const int m = live ? (int)(idx - row * M) : 0;
  const size_t e = (size_t)(live ? idx : 0);
  const float4* gl4 = reinterpret_cast<const float4*>(g_loc + e * LP * 2);
  const float4* gp4 = reinterpret_cast<const float4*>(g_prob + e * LP);
  const float4* p4 = reinterpret_cast<const float4*>(probs + e * LP);
  float p[LP], g[LP];
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < LP / 4; ++i) {
    const float4 a = p4[i], b = gp4[i];
    p[4 * i] = a.x; p[4 * i + 1] = a.y; p[4 * i + 2] = a.z; p[4 * i + 3] = a.w;
    g[4 * i] = b.x; g[4 * i + 1] = b.y; g[4 * i + 2] = b.z; g[4 * i + 3] = b.w;
    dot += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
  }
  if (live) {
    float4* o = reinterpret_cast<float4*>(g_logits + row * ld_glogits + (size_t)m * LP);
#pragma unroll
    for (int i = 0; i < LP / 4; ++i)
      o[i] = make_float4(p[4 * i] * (g[4 * i] - dot), p[4 * i + 1] * (g[4 * i + 1] - dot),
                         p[4 * i + 2] * (g[4 * i + 2] - dot), p[4 * i + 3] * (g[4 * i + 3] - dot));
  }
  const float* r = ref + row * (size_t)(L * (BOX ? 4 : 2));
  const float4* o4 = reinterpret_cast<const float4*>(off + row * ld_off + (size_t)m * LP * 2);
  float4* go4 = reinterpret_cast<float4*>(g_off + row * ld_goff + (size_t)m * LP * 2);
#pragma unroll
  for (int l = 0; l < L; ++l) {
    float sx, sy;
    if (BOX) {
      const float4 b = reinterpret_cast<const float4*>(r)[l];
      sx = b.z; sy = b.w;
    } else {
      sx = (float)shapes[2 * l + 1]; sy = (float)shapes[2 * l];
    }
    float ax = 0.f, ay = 0.f, aw = 0.f, ah = 0.f;   // this head's share of d ref (x, y, w, h) at level l
#pragma unroll
    for (int q = 0; q < P / 2; ++q) {
      const float4 gl = gl4[l * (P / 2) + q];
      ax += gl.x + gl.z;
      ay += gl.y + gl.w;
      float4 w;
      if (BOX) {
        w.x = gl.x * 0.5f * sx / (float)P; w.y = gl.y * 0.5f * sy / (float)P;
        w.z = gl.z * 0.5f * sx / (float)P; w.w = gl.w * 0.5f * sy / (float)P;
        if (g_ref != nullptr) {
          const float4 o = o4[l * (P / 2) + q];
          aw += (gl.x * 0.5f) * (o.x / (float)P) + (gl.z * 0.5f) * (o.z / (float)P);
          ah += (gl.y * 0.5f) * (o.y / (float)P) + (gl.w * 0.5f) * (o.w / (float)P);
        }
      } else {
        w.x = gl.x / sx; w.y = gl.y / sy; w.z = gl.z / sx; w.w = gl.w / sy;
      }
      if (live) go4[l * (P / 2) + q] = w;
    }
    if (g_ref != nullptr) {
      if (!live) ax = ay = aw = ah = 0.f;
      for (int d = 1; d < M; d <<= 1) {
        ax += __shfl_xor(ax, d);
        ay += __shfl_xor(ay, d);
        if (BOX) {
          aw += __shfl_xor(aw, d);
          ah += __shfl_xor(ah, d);
        }
      }
      if (live && m == 0) {
        if (BOX)
          reinterpret_cast<float4*>(g_ref + row * (size_t)(L * 4))[l] = make_float4(ax, ay, aw, ah);
        else
          reinterpret_cast<float2*>(g_ref + row * (size_t)(L * 2))[l] = make_float2(ax, ay);
      }
    }
  }
}

bool geometry_ok(int M, int L, int P, int ref_dim) {
  return L == 4 && P == 4 && (ref_dim == 2 || ref_dim == 4) && M >= 1 && M <= 64 && (M & (M - 1)) == 0;
}

}  // namespace

extern "C" int egtr_msda_geometry_forward_f32(egtr_stream_t stream, const float* sampling_offsets, long long ld_offsets,
                                              const float* attention_logits, long long ld_logits,
                                              const float* reference_points, int ref_dim,
                                              const int64_t* spatial_shapes, float* sampling_locations,
                                              float* attention_weights, long long rows, int num_heads, int num_levels,
                                              int num_points) {
  if (!sampling_offsets || !attention_logits || !reference_points || !spatial_shapes || !sampling_locations ||
      !attention_weights || rows <= 0)
    return EGTR_E_ARG;
  if (!geometry_ok(num_heads, num_levels, num_points, ref_dim)) return EGTR_E_UNSUPPORTED;
  if (ld_offsets % 4 || ld_logits % 4 || ((uintptr_t)sampling_offsets | (uintptr_t)attention_logits |
                                          (uintptr_t)reference_points) % 16)
    return EGTR_E_UNSUPPORTED;
  const long long n = rows * num_heads;
  const dim3 grid((unsigned)((n + 255) / 256));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (ref_dim == 4)
    hipLaunchKernelGGL((msda_geometry_fwd<4, 4, true>), grid, dim3(256), 0, st, sampling_offsets, ld_offsets,
                       attention_logits, ld_logits, reference_points, spatial_shapes, sampling_locations,
                       attention_weights, rows, num_heads);
  else
    hipLaunchKernelGGL((msda_geometry_fwd<4, 4, false>), grid, dim3(256), 0, st, sampling_offsets, ld_offsets,
                       attention_logits, ld_logits, reference_points, spatial_shapes, sampling_locations,
                       attention_weights, rows, num_heads);
  return egtr_check_launch();
}

extern "C" int egtr_msda_geometry_backward_f32(egtr_stream_t stream, const float* grad_locations,
                                               const float* grad_weights, const float* attention_weights,
                                               const float* sampling_offsets, long long ld_offsets,
                                               const float* reference_points, int ref_dim,
                                               const int64_t* spatial_shapes, float* grad_offsets,
                                               long long ld_grad_offsets, float* grad_logits,
                                               long long ld_grad_logits, float* grad_reference, long long rows,
                                               int num_heads, int num_levels, int num_points) {
  if (!grad_locations || !grad_weights || !attention_weights || !sampling_offsets || !reference_points ||
      !spatial_shapes || !grad_offsets || !grad_logits || rows <= 0)
    return EGTR_E_ARG;
  if (!geometry_ok(num_heads, num_levels, num_points, ref_dim)) return EGTR_E_UNSUPPORTED;
  if (ld_offsets % 4 || ld_grad_offsets % 4 || ld_grad_logits % 4 ||
      ((uintptr_t)sampling_offsets | (uintptr_t)reference_points | (uintptr_t)grad_offsets | (uintptr_t)grad_logits) % 16)
    return EGTR_E_UNSUPPORTED;
  const long long n = rows * num_heads;
  const dim3 grid((unsigned)((n + 255) / 256));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (ref_dim == 4)
    hipLaunchKernelGGL((msda_geometry_bwd<4, 4, true>), grid, dim3(256), 0, st, grad_locations, grad_weights,
                       attention_weights, sampling_offsets, ld_offsets, reference_points, spatial_shapes, grad_offsets,
                       ld_grad_offsets, grad_logits, ld_grad_logits, grad_reference, rows, num_heads);
  else
    hipLaunchKernelGGL((msda_geometry_bwd<4, 4, false>), grid, dim3(256), 0, st, grad_locations, grad_weights,
                       attention_weights, sampling_offsets, ld_offsets, reference_points, spatial_shapes, grad_offsets,
                       ld_grad_offsets, grad_logits, ld_grad_logits, grad_reference, rows, num_heads);
  return egtr_check_launch();
}
