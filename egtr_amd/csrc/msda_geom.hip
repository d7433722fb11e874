// libegtr_hip.so -- the elementwise prologue of multi-scale deformable attention under autograd (training):
//   sampling_locations = reference_points + sampling_offsets / (W_l, H_l)                       (2-d reference points)
//                      = reference_xy + sampling_offsets / P * reference_wh * 0.5               (4-d reference boxes)
//   attention_weights  = softmax over the L * P samples of a head
// (model/deformable_detr.py:1055-1073) and its backward.  The reference issues softmax, a broadcast division, one or
// three broadcast multiplications and an addition (and their backward kernels) over [B, Lq, M, L, P, 2] tensors; here each
// direction is ONE pass, one wavefront per (batch, query) row.  The reference-point gradient (decoder: reference points
// are a learned function of the queries) is the sum over heads and points, folded across the lanes with xor shuffles.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

namespace {

// One wavefront per (batch, query) row, M = 8 heads, L = 4 levels, P = 4 points: the 256 offsets of a row are 64 float4
// -- lane = 8 m + 2 l + q holds points 2 q, 2 q + 1 of (head m, level l) -- and its 128 logits are 32 float4 -- lane
// (< 32) = 4 m + j, the softmax of a head runs over 4 neighbouring lanes.  Every access of a wave is one contiguous
// 1 KiB / 512 B row segment.
constexpr int kM = 8, kL = 4, kP = 4;

__device__ __forceinline__ float quad_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 1));
  return fmaxf(v, __shfl_xor(v, 2));
}
__device__ __forceinline__ float quad_sum(float v) {
  v += __shfl_xor(v, 1);
  return v + __shfl_xor(v, 2);
}

template <bool BOX>
__global__ __launch_bounds__(256) void msda_geometry_fwd(const float* __restrict__ off, long long ld_off,
                                                         const float* __restrict__ logits, long long ld_logits,
                                                         const float* __restrict__ ref,
                                                         const int64_t* __restrict__ shapes, float* __restrict__ loc,
                                                         float* __restrict__ probs, long long rows) {
  const int lane = threadIdx.x & 63;
  const long long row = blockIdx.x * 4ll + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int l = (lane >> 1) & 3;
  const float4 o = reinterpret_cast<const float4*>(off + row * ld_off)[lane];
  float rx, ry, sx, sy;
  if (BOX) {
    const float4 b = reinterpret_cast<const float4*>(ref + row * (kL * 4))[l];
    rx = b.x; ry = b.y; sx = b.z; sy = b.w;
  } else {
    const float2 b = reinterpret_cast<const float2*>(ref + row * (kL * 2))[l];
    rx = b.x; ry = b.y;
    sx = (float)shapes[2 * l + 1]; sy = (float)shapes[2 * l];   // (W, H)
  }
  float4 w;
  if (BOX) {
    w.x = rx + o.x / (float)kP * sx * 0.5f; w.y = ry + o.y / (float)kP * sy * 0.5f;
    w.z = rx + o.z / (float)kP * sx * 0.5f; w.w = ry + o.w / (float)kP * sy * 0.5f;
  } else {
    w.x = rx + o.x / sx; w.y = ry + o.y / sy; w.z = rx + o.z / sx; w.w = ry + o.w / sy;
  }
  reinterpret_cast<float4*>(loc + row * 256)[lane] = w;
  // softmax: lanes 0..31 hold the row's logits, the other half works on a copy (all lanes take part in the shuffles)
  float4 x = reinterpret_cast<const float4*>(logits + row * ld_logits)[lane & 31];
  const float mx = quad_max(fmaxf(fmaxf(x.x, x.y), fmaxf(x.z, x.w)));
  x.x = expf(x.x - mx); x.y = expf(x.y - mx); x.z = expf(x.z - mx); x.w = expf(x.w - mx);
  const float s = quad_sum((x.x + x.y) + (x.z + x.w));
  if (lane < 32) reinterpret_cast<float4*>(probs + row * 128)[lane] = make_float4(x.x / s, x.y / s, x.z / s, x.w / s);
}

template <bool BOX>
__global__ __launch_bounds__(256) void msda_geometry_bwd(const float* __restrict__ g_loc, const float* __restrict__ g_prob,
                                                         const float* __restrict__ probs, const float* __restrict__ off,
                                                         long long ld_off, const float* __restrict__ ref,
                                                         const int64_t* __restrict__ shapes, float* __restrict__ g_off,
                                                         long long ld_goff, float* __restrict__ g_logits,
                                                         long long ld_glogits, float* __restrict__ g_ref,
                                                         long long rows) {
  const int lane = threadIdx.x & 63;
  const long long row = blockIdx.x * 4ll + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int l = (lane >> 1) & 3;
  // softmax backward: g_logit = p (g - sum_j p_j g_j) over the 16 samples of a head (4 lanes)
  const float4 p = reinterpret_cast<const float4*>(probs + row * 128)[lane & 31];
  const float4 g = reinterpret_cast<const float4*>(g_prob + row * 128)[lane & 31];
  const float dot = quad_sum((p.x * g.x + p.y * g.y) + (p.z * g.z + p.w * g.w));
  if (lane < 32)
    reinterpret_cast<float4*>(g_logits + row * ld_glogits)[lane] =
        make_float4(p.x * (g.x - dot), p.y * (g.y - dot), p.z * (g.z - dot), p.w * (g.w - dot));
  const float4 gl = reinterpret_cast<const float4*>(g_loc + row * 256)[lane];
  float sx, sy;
  if (BOX) {
    const float4 b = reinterpret_cast<const float4*>(ref + row * (kL * 4))[l];
    sx = b.z; sy = b.w;
  } else {
    sx = (float)shapes[2 * l + 1]; sy = (float)shapes[2 * l];
  }
  float4 w;
  if (BOX) {
    w.x = gl.x * 0.5f * sx / (float)kP; w.y = gl.y * 0.5f * sy / (float)kP;
    w.z = gl.z * 0.5f * sx / (float)kP; w.w = gl.w * 0.5f * sy / (float)kP;
  } else {
    w.x = gl.x / sx; w.y = gl.y / sy; w.z = gl.z / sx; w.w = gl.w / sy;
  }
  reinterpret_cast<float4*>(g_off + row * ld_goff)[lane] = w;
  if (g_ref != nullptr) {   // d ref[l] = sum over heads (lane bits 3..5) and point pairs (bit 0)
    float ax = gl.x + gl.z, ay = gl.y + gl.w, aw = 0.f, ah = 0.f;
    if (BOX) {
      const float4 o = reinterpret_cast<const float4*>(off + row * ld_off)[lane];
      aw = (gl.x * 0.5f) * (o.x / (float)kP) + (gl.z * 0.5f) * (o.z / (float)kP);
      ah = (gl.y * 0.5f) * (o.y / (float)kP) + (gl.w * 0.5f) * (o.w / (float)kP);
    }
#pragma unroll
    for (int d = 1; d < 64; d = (d == 1 ? 8 : d << 1)) {
      ax += __shfl_xor(ax, d);
      ay += __shfl_xor(ay, d);
      if (BOX) {
        aw += __shfl_xor(aw, d);
        ah += __shfl_xor(ah, d);
      }
    }
    if ((lane & ~6) == 0) {   // lanes 0, 2, 4, 6: level l
      if (BOX)
        reinterpret_cast<float4*>(g_ref + row * (kL * 4))[l] = make_float4(ax, ay, aw, ah);
      else
        reinterpret_cast<float2*>(g_ref + row * (kL * 2))[l] = make_float2(ax, ay);
    }
  }
}

bool geometry_ok(int M, int L, int P, int ref_dim) {
  return M == kM && L == kL && P == kP && (ref_dim == 2 || ref_dim == 4);
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
  const dim3 grid((unsigned)((rows + 3) / 4));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (ref_dim == 4)
    hipLaunchKernelGGL((msda_geometry_fwd<true>), grid, dim3(256), 0, st, sampling_offsets, ld_offsets,
                       attention_logits, ld_logits, reference_points, spatial_shapes, sampling_locations,
                       attention_weights, rows);
  else
    hipLaunchKernelGGL((msda_geometry_fwd<false>), grid, dim3(256), 0, st, sampling_offsets, ld_offsets,
                       attention_logits, ld_logits, reference_points, spatial_shapes, sampling_locations,
                       attention_weights, rows);
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
  const dim3 grid((unsigned)((rows + 3) / 4));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (ref_dim == 4)
    hipLaunchKernelGGL((msda_geometry_bwd<true>), grid, dim3(256), 0, st, grad_locations, grad_weights,
                       attention_weights, sampling_offsets, ld_offsets, reference_points, spatial_shapes, grad_offsets,
                       ld_grad_offsets, grad_logits, ld_grad_logits, grad_reference, rows);
  else
    hipLaunchKernelGGL((msda_geometry_bwd<false>), grid, dim3(256), 0, st, grad_locations, grad_weights,
                       attention_weights, sampling_offsets, ld_offsets, reference_points, spatial_shapes, grad_offsets,
                       ld_grad_offsets, grad_logits, ld_grad_logits, grad_reference, rows);
  return egtr_check_launch();
}
