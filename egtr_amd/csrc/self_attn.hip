// Decoder multi-head self-attention core for gfx950: softmax(q k^T) v per head, fp32, exact-f32 MFMA.
//
// Replaces the bmm -> softmax -> bmm chain (and the head transposes around it) of
// DeformableDetrMultiheadAttention.forward, model/deformable_detr.py:1170-1253, and emits the retained
// per-layer maps "scaled q" and "k" in [B, M, N, D] layout (dd:1179-1185) that the EGTR relation head consumes.
//
// Shape regime: N = 100..300 object queries, D = 32, M = 8: the whole score row fits in registers, so no
// online-softmax rescaling is needed inside a wave.  One workgroup owns (batch b, head h, a tile of 16 query rows); its
// four waves split the keys and merge their partial (max, sum, output) through LDS.
//
// MFMA use (v_mfma_f32_16x16x4_f32: exact f32, A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D[row=(l>>4)*4+r][col=l&15]):
//   * scores are computed TRANSPOSED, S^T = K Q^T, so that D[row = key][col = query]: a lane (c = l&15, g = l>>4)
//     then holds, for query row c, the keys {k0 + 4g + r}: the softmax row statistics are an in-lane reduce plus
//     two cross-lane steps (xor 16, xor 32), and
//   * the probabilities are ALREADY in the B-operand layout of the second contraction O^T = V^T P^T (k-slot g at
//     step t <-> key k0 + 4g + t), so P never leaves registers (no LDS round trip, no transpose).
//   * the contraction index of the first product is assigned d = 8g + kk so each lane fetches its K / Q row
//     fragment as two 16-byte loads.
//
// Backward (training): one wave per (b, h, 16-row tile) plays two roles without atomics:
//   role A: dQ for its 16 query rows (loops over key tiles), role B: dK, dV for its 16 key rows (loops over
//   query tiles); P is recomputed from the saved log-sum-exp.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float xor16_32_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16));
  v = fmaxf(v, __shfl_xor(v, 32));
  return v;
}
__device__ __forceinline__ float xor16_32_sum(float v) {
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

// Load 8 contiguous floats of row `row` (clamped to N-1) of a [B, N, M*32] tensor, head h, columns 8g..8g+7.
__device__ __forceinline__ void load_frag8(const float* __restrict__ base, int N, int MD, int b, int row, int h,
                                           int g, float (&f)[8]) {
  const int r = min(row, N - 1);
  const float4* p = reinterpret_cast<const float4*>(base + ((size_t)b * N + r) * MD + h * 32 + g * 8);
  const float4 a = p[0], c = p[1];
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w;
  f[4] = c.x; f[5] = c.y; f[6] = c.z; f[7] = c.w;
}

// bf16 storage (the bf16 model's decoder, inference): the same kernel reads / writes raw bfloat16 and computes in fp32 -- widening
// is exact, the output is rounded to nearest even once, the retained maps are bit copies of the inputs.  (Round 5 ran the fp32
// kernel between three widening and three narrowing cast launches per layer: ~0.5 ms of a 24.8 ms stress forward.)
__device__ __forceinline__ float bf16_up(uint16_t u) { return __uint_as_float((unsigned)u << 16); }
__device__ __forceinline__ uint16_t bf16_rne(float x) {
  unsigned u = __float_as_uint(x);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0;
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ void load_frag8(const uint16_t* __restrict__ base, int N, int MD, int b, int row, int h, int g,
                                           float (&f)[8]) {
  const int r = min(row, N - 1);
  const uint4 a = *reinterpret_cast<const uint4*>(base + ((size_t)b * N + r) * MD + h * 32 + g * 8);
  f[0] = __uint_as_float(a.x << 16); f[1] = __uint_as_float(a.x & 0xffff0000u);
  f[2] = __uint_as_float(a.y << 16); f[3] = __uint_as_float(a.y & 0xffff0000u);
  f[4] = __uint_as_float(a.z << 16); f[5] = __uint_as_float(a.z & 0xffff0000u);
  f[6] = __uint_as_float(a.w << 16); f[7] = __uint_as_float(a.w & 0xffff0000u);
}
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const uint16_t* p) { return bf16_up(*p); }
__device__ __forceinline__ void store8(float* p, const float (&f)[8]) {
  reinterpret_cast<float4*>(p)[0] = make_float4(f[0], f[1], f[2], f[3]);
  reinterpret_cast<float4*>(p)[1] = make_float4(f[4], f[5], f[6], f[7]);
}
__device__ __forceinline__ void store8(uint16_t* p, const float (&f)[8]) {   // (values that came from bf16: exact)
  uint4 o;
  o.x = (__float_as_uint(f[0]) >> 16) | (__float_as_uint(f[1]) & 0xffff0000u);
  o.y = (__float_as_uint(f[2]) >> 16) | (__float_as_uint(f[3]) & 0xffff0000u);
  o.z = (__float_as_uint(f[4]) >> 16) | (__float_as_uint(f[5]) & 0xffff0000u);
  o.w = (__float_as_uint(f[6]) >> 16) | (__float_as_uint(f[7]) & 0xffff0000u);
  *reinterpret_cast<uint4*>(p) = o;
}
__device__ __forceinline__ void store4(float* p, float a, float b, float c, float d) {
  *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}
__device__ __forceinline__ void store4(uint16_t* p, float a, float b, float c, float d) {
  *reinterpret_cast<uint2*>(p) = make_uint2((unsigned)bf16_rne(a) | ((unsigned)bf16_rne(b) << 16),
                                            (unsigned)bf16_rne(c) | ((unsigned)bf16_rne(d) << 16));
}

// Forward.  The keys are split over KS waves of one workgroup (wave w takes key tiles w, w + KS, ...; NTW = tiles per
// wave held in registers): with ONE wave per (b, head, 16-query tile) the kernel was a single dependent chain of
// 13 x (K-tile load -> 8 MFMAs) + 104 x (V load -> MFMA) -- 15.4 us per launch for 41 MFLOP with only 104 waves on the
// chip; split four ways 11 us (226.9 vs 225.6 images/s end to end).  Each wave keeps its own running
// maximum / sum / unnormalised output and the partial results are merged lane by lane through LDS
// (O = sum_w O_w e^{m_w - M} / sum_w s_w e^{m_w - M}); the softmax is exact as before, only the summation order differs.
template <int NTW, int KS, typename T = float>
__global__ __launch_bounds__(64 * KS) void self_attn_fwd_split_f32(
    const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, T* __restrict__ out,
    T* __restrict__ q_heads, T* __restrict__ k_heads, float* __restrict__ lse, int B, int N, int M) {
  __shared__ float s_part[KS][10][64];  // [wave][o0[0..3], o1[0..3], max, sum][lane]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int ntile = (N + 15) >> 4;
  int wid = blockIdx.x;
  const int qt = wid % ntile;
  wid /= ntile;
  const int h = wid % M, b = wid / M;
  const int MD = M * 32;
  const int q0 = qt * 16;

  float qf[8];
  load_frag8(q, N, MD, b, q0 + c, h, g, qf);
  if (wave == 0 && (q_heads != nullptr || k_heads != nullptr) && q0 + c < N) {
    const size_t o = (((size_t)b * M + h) * N + q0 + c) * 32 + g * 8;
    if (q_heads) store8(q_heads + o, qf);
    if (k_heads) {
      float kf0[8];
      load_frag8(k, N, MD, b, q0 + c, h, g, kf0);
      store8(k_heads + o, kf0);
    }
  }

  f32x4 s[NTW];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < NTW; ++i) {
    const int k0 = (wave + i * KS) * 16;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (k0 < N) {
      float kf[8];
      load_frag8(k, N, MD, b, k0 + c, h, g, kf);
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) acc = mfma16(kf[kk], qf[kk], acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = (k0 + g * 4 + r) < N;
      acc[r] = ok ? acc[r] : -INFINITY;
      mx = fmaxf(mx, acc[r]);
    }
    s[i] = acc;
  }
  mx = xor16_32_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NTW; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float p = (mx == -INFINITY) ? 0.f : __expf(s[i][r] - mx);  // a wave without keys contributes nothing
      s[i][r] = p;
      sum += p;
    }
  }
  sum = xor16_32_sum(sum);

  f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NTW; ++i) {
    const int k0 = (wave + i * KS) * 16;
    if (k0 < N) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int key = min(k0 + g * 4 + t, N - 1);
        const T* vp = v + ((size_t)b * N + key) * MD + h * 32 + c;
        o0 = mfma16(ld1(vp), s[i][t], o0);
        o1 = mfma16(ld1(vp + 16), s[i][t], o1);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    s_part[wave][r][lane] = o0[r];
    s_part[wave][4 + r][lane] = o1[r];
  }
  s_part[wave][8][lane] = mx;
  s_part[wave][9][lane] = sum;
  __syncthreads();
  if (wave != 0 || q0 + c >= N) return;
  float mall = -INFINITY;
#pragma unroll
  for (int w = 0; w < KS; ++w) mall = fmaxf(mall, s_part[w][8][lane]);
  float tot = 0.f, o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int w = 0; w < KS; ++w) {
    const float mw = s_part[w][8][lane];
    const float f = (mw == -INFINITY) ? 0.f : __expf(mw - mall);
    tot += s_part[w][9][lane] * f;
#pragma unroll
    for (int r = 0; r < 8; ++r) o[r] += s_part[w][r][lane] * f;
  }
  const float inv = 1.f / tot;
  T* op = out + ((size_t)b * N + q0 + c) * MD + h * 32 + g * 4;
  store4(op, o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
  store4(op + 16, o[4] * inv, o[5] * inv, o[6] * inv, o[7] * inv);
  if (lse != nullptr && g == 0) lse[((size_t)b * M + h) * N + q0 + c] = mall + __logf(tot);
}

// Backward.  See file header.  grad wrt the kernel's inputs q (already scaled), k, v.
constexpr int kBwdSplit = 4;   // waves per workgroup: each takes every fourth tile of the reduction, partials summed in LDS
__global__ __launch_bounds__(64 * kBwdSplit) void self_attn_bwd_f32(const float* __restrict__ q, const float* __restrict__ k,
                                                        const float* __restrict__ v, const float* __restrict__ out,
                                                        const float* __restrict__ lse,
                                                        const float* __restrict__ grad_out,
                                                        float* __restrict__ grad_q, float* __restrict__ grad_k,
                                                        float* __restrict__ grad_v, int B, int N, int M,
                                                        const float* __restrict__ add_q, const float* __restrict__ add_k) {
  __shared__ f32x4 s_acc[kBwdSplit][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int ntile = (N + 15) >> 4;
  int wid = blockIdx.x;
  const int tile = wid % ntile;
  wid /= ntile;
  const int h = wid % M, b = wid / M;
  const int MD = M * 32;
  const int r0 = tile * 16;
  const float* lse_bh = lse + ((size_t)b * M + h) * N;

  // The two roles are independent and run as separate workgroups (blockIdx.y), and the reduction loop of a role is dealt to
  // the four waves of its workgroup, whose partial sums meet in LDS in wave order (round 4: one wave did both roles in turn,
  // 416 waves for B = 4, N = 200 on 1 024 SIMDs: 44 us per call).
  // ---------------- role A: dQ for query rows r0..r0+15 (layout: scores transposed, query row = c) ------------
  if (blockIdx.y == 0) {
    float qf[8], dof[8], of[8];
    load_frag8(q, N, MD, b, r0 + c, h, g, qf);
    load_frag8(grad_out, N, MD, b, r0 + c, h, g, dof);
    load_frag8(out, N, MD, b, r0 + c, h, g, of);
    float delta = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) delta += dof[i] * of[i];
    delta = xor16_32_sum(delta);
    const float l_row = lse_bh[min(r0 + c, N - 1)];
    f32x4 dq0 = {0.f, 0.f, 0.f, 0.f}, dq1 = {0.f, 0.f, 0.f, 0.f};
    for (int kt = wave; kt < ntile; kt += kBwdSplit) {
      const int k0 = kt * 16;
      float kf[8], vf[8];
      load_frag8(k, N, MD, b, k0 + c, h, g, kf);
      load_frag8(v, N, MD, b, k0 + c, h, g, vf);
      f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        st = mfma16(kf[kk], qf[kk], st);    // S^T[key][qrow]
        dp = mfma16(vf[kk], dof[kk], dp);   // dP^T[key][qrow] = V dO^T
      }
      f32x4 ds;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool ok = (k0 + g * 4 + r) < N;
        const float p = ok ? __expf(st[r] - l_row) : 0.f;
        ds[r] = p * (dp[r] - delta);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int key = min(k0 + g * 4 + t, N - 1);
        const float* kp = k + ((size_t)b * N + key) * MD + h * 32 + c;
        dq0 = mfma16(kp[0], ds[t], dq0);   // dQ^T[d][qrow] += K^T[d][key] dS^T[key][qrow]
        dq1 = mfma16(kp[16], ds[t], dq1);
      }
    }
    s_acc[wave][0][lane] = dq0;
    s_acc[wave][1][lane] = dq1;
    __syncthreads();
    if (wave == 0 && r0 + c < N) {
#pragma unroll
      for (int w = 1; w < kBwdSplit; ++w) {
        dq0 += s_acc[w][0][lane];
        dq1 += s_acc[w][1][lane];
      }
      const size_t at = ((size_t)b * N + r0 + c) * MD + h * 32 + g * 4;
      float* p = grad_q + at;
      if (add_q != nullptr) {   // gradient that reaches q by another route (the retained maps): added here, not in a launch
        const float4 a0 = *reinterpret_cast<const float4*>(add_q + at), a1 = *reinterpret_cast<const float4*>(add_q + at + 16);
        dq0 += f32x4{a0.x, a0.y, a0.z, a0.w};
        dq1 += f32x4{a1.x, a1.y, a1.z, a1.w};
      }
      *reinterpret_cast<float4*>(p) = make_float4(dq0[0], dq0[1], dq0[2], dq0[3]);
      *reinterpret_cast<float4*>(p + 16) = make_float4(dq1[0], dq1[1], dq1[2], dq1[3]);
    }
  }
  // ---------------- role B: dK, dV for key rows r0..r0+15 (scores NOT transposed: D[row = qrow][col = key]) ----
  else {
    float kf[8], vf[8];
    load_frag8(k, N, MD, b, r0 + c, h, g, kf);
    load_frag8(v, N, MD, b, r0 + c, h, g, vf);
    const bool key_ok = (r0 + c) < N;
    f32x4 dk0 = {0.f, 0.f, 0.f, 0.f}, dk1 = {0.f, 0.f, 0.f, 0.f}, dv0 = {0.f, 0.f, 0.f, 0.f},
          dv1 = {0.f, 0.f, 0.f, 0.f};
    for (int qt = wave; qt < ntile; qt += kBwdSplit) {
      const int q0 = qt * 16;
      float qf[8], dof[8], of[8];
      load_frag8(q, N, MD, b, q0 + c, h, g, qf);
      load_frag8(grad_out, N, MD, b, q0 + c, h, g, dof);
      load_frag8(out, N, MD, b, q0 + c, h, g, of);
      float delta_c = 0.f;  // delta of query row q0 + c
#pragma unroll
      for (int i = 0; i < 8; ++i) delta_c += dof[i] * of[i];
      delta_c = xor16_32_sum(delta_c);
      f32x4 sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        sc = mfma16(qf[kk], kf[kk], sc);    // S[qrow][key]
        dp = mfma16(dof[kk], vf[kk], dp);   // dP[qrow][key] = dO V^T
      }
      f32x4 p, ds;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qrow = q0 + g * 4 + r;
        const float l_row = lse_bh[min(qrow, N - 1)];
        const float dl = __shfl(delta_c, g * 4 + r);  // lane (c' = 4g+r, g' = 0) holds delta of row q0+4g+r
        const bool ok = key_ok && qrow < N;
        p[r] = ok ? __expf(sc[r] - l_row) : 0.f;
        ds[r] = p[r] * (dp[r] - dl);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int qrow = min(q0 + g * 4 + t, N - 1);
        const float* dop = grad_out + ((size_t)b * N + qrow) * MD + h * 32 + c;
        const float* qp = q + ((size_t)b * N + qrow) * MD + h * 32 + c;
        dv0 = mfma16(dop[0], p[t], dv0);    // dV^T[d][key] += dO^T[d][qrow] P[qrow][key]
        dv1 = mfma16(dop[16], p[t], dv1);
        dk0 = mfma16(qp[0], ds[t], dk0);    // dK^T[d][key] += Q^T[d][qrow] dS[qrow][key]
        dk1 = mfma16(qp[16], ds[t], dk1);
      }
    }
    s_acc[wave][0][lane] = dk0;
    s_acc[wave][1][lane] = dk1;
    s_acc[wave][2][lane] = dv0;
    s_acc[wave][3][lane] = dv1;
    __syncthreads();
    if (wave == 0 && key_ok) {
#pragma unroll
      for (int w = 1; w < kBwdSplit; ++w) {
        dk0 += s_acc[w][0][lane];
        dk1 += s_acc[w][1][lane];
        dv0 += s_acc[w][2][lane];
        dv1 += s_acc[w][3][lane];
      }
      const size_t at = ((size_t)b * N + r0 + c) * MD + h * 32 + g * 4;
      float* pk = grad_k + at;
      float* pv = grad_v + at;
      if (add_k != nullptr) {
        const float4 a0 = *reinterpret_cast<const float4*>(add_k + at), a1 = *reinterpret_cast<const float4*>(add_k + at + 16);
        dk0 += f32x4{a0.x, a0.y, a0.z, a0.w};
        dk1 += f32x4{a1.x, a1.y, a1.z, a1.w};
      }
      *reinterpret_cast<float4*>(pk) = make_float4(dk0[0], dk0[1], dk0[2], dk0[3]);
      *reinterpret_cast<float4*>(pk + 16) = make_float4(dk1[0], dk1[1], dk1[2], dk1[3]);
      *reinterpret_cast<float4*>(pv) = make_float4(dv0[0], dv0[1], dv0[2], dv0[3]);
      *reinterpret_cast<float4*>(pv + 16) = make_float4(dv1[0], dv1[1], dv1[2], dv1[3]);
    }
  }
}

}  // namespace

extern "C" int egtr_self_attn_forward_f32(egtr_stream_t stream, const float* q, const float* k, const float* v,
                                          int batch, int num_query, int num_heads, int head_dim, float* out,
                                          float* q_heads, float* k_heads, float* lse) {
  if (!q || !k || !v || !out) return EGTR_E_ARG;
  if (batch <= 0 || num_query <= 0 || num_heads <= 0) return EGTR_E_ARG;
  if (head_dim != 32 || num_query > 16 * 40) return EGTR_E_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nt = (num_query + 15) / 16;
  // keys split over 4 waves per workgroup (shorter dependent chains, 4x the waves on the chip)
  const dim3 grid(batch * num_heads * nt), block(256);
  if (nt <= 16)
    hipLaunchKernelGGL((self_attn_fwd_split_f32<4, 4>), grid, block, 0, st, q, k, v, out, q_heads, k_heads, lse, batch,
                       num_query, num_heads);
  else
    hipLaunchKernelGGL((self_attn_fwd_split_f32<10, 4>), grid, block, 0, st, q, k, v, out, q_heads, k_heads, lse, batch,
                       num_query, num_heads);
  return egtr_check_launch();
}

extern "C" int egtr_self_attn_forward_bf16(egtr_stream_t stream, const uint16_t* q, const uint16_t* k, const uint16_t* v,
                                           int batch, int num_query, int num_heads, int head_dim, uint16_t* out,
                                           uint16_t* q_heads, uint16_t* k_heads) {
  if (!q || !k || !v || !out) return EGTR_E_ARG;
  if (batch <= 0 || num_query <= 0 || num_heads <= 0) return EGTR_E_ARG;
  if (head_dim != 32 || num_query > 16 * 40) return EGTR_E_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v) |
       reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(q_heads) | reinterpret_cast<uintptr_t>(k_heads)) & 15)
    return EGTR_E_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nt = (num_query + 15) / 16;
  const dim3 grid(batch * num_heads * nt), block(256);
  if (nt <= 16)
    hipLaunchKernelGGL((self_attn_fwd_split_f32<4, 4, uint16_t>), grid, block, 0, st, q, k, v, out, q_heads, k_heads,
                       (float*)nullptr, batch, num_query, num_heads);
  else
    hipLaunchKernelGGL((self_attn_fwd_split_f32<10, 4, uint16_t>), grid, block, 0, st, q, k, v, out, q_heads, k_heads,
                       (float*)nullptr, batch, num_query, num_heads);
  return egtr_check_launch();
}

extern "C" int egtr_self_attn_backward_acc_f32(egtr_stream_t stream, const float* q, const float* k, const float* v,
                                               const float* out, const float* lse, const float* grad_out, int batch,
                                               int num_query, int num_heads, int head_dim, float* grad_q, float* grad_k,
                                               float* grad_v, const float* grad_q_add, const float* grad_k_add) {
  if (!q || !k || !v || !out || !lse || !grad_out || !grad_q || !grad_k || !grad_v) return EGTR_E_ARG;
  if ((reinterpret_cast<uintptr_t>(grad_q_add) | reinterpret_cast<uintptr_t>(grad_k_add)) & 15) return EGTR_E_UNSUPPORTED;
  if (batch <= 0 || num_query <= 0 || num_heads <= 0) return EGTR_E_ARG;
  if (head_dim != 32) return EGTR_E_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int ntile = (num_query + 15) / 16;
  hipLaunchKernelGGL(self_attn_bwd_f32, dim3(batch * num_heads * ntile, 2), dim3(64 * kBwdSplit), 0, st, q, k, v, out, lse, grad_out,
                     grad_q, grad_k, grad_v, batch, num_query, num_heads, grad_q_add, grad_k_add);
  return egtr_check_launch();
}

extern "C" int egtr_self_attn_backward_f32(egtr_stream_t stream, const float* q, const float* k, const float* v,
                                           const float* out, const float* lse, const float* grad_out, int batch,
                                           int num_query, int num_heads, int head_dim, float* grad_q, float* grad_k,
                                           float* grad_v) {
  return egtr_self_attn_backward_acc_f32(stream, q, k, v, out, lse, grad_out, batch, num_query, num_heads, head_dim, grad_q,
                                         grad_k, grad_v, nullptr, nullptr);
}
