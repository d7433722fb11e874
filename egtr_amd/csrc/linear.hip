// Skinny fp32 linear layer for gfx950: Y[M,N] = act((X[M,K] . W[N,K]^T + b[N]) * alpha), M small (object queries).
//
// Why: the decoder, the detection heads and the relation-head projections apply ~60 nn.Linear layers per image to
// M = 200 query rows.  hipBLASLt serves those from a 256x208 macro-tile kernel -- ONE workgroup, 54 us per call
// (rocprofv3, profiles/r01_bench_kernel_stats.txt: 62 calls = 3.3 ms of a 12 ms forward).  Here a workgroup owns a
// 16 x 32 output tile; its 4 waves split K four ways (and the 4 lane groups of a wave split it again), so a K = 256
// product is 32 exact-f32 MFMAs (v_mfma_f32_16x16x4_f32) per wave, followed by an LDS reduction and a fused
// bias / scale / ReLU epilogue.  104 workgroups for a 200 x 256 output.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// hipcc sinks ordinary global loads next to their first use (20 VGPRs, one L2 round trip per K step): the operand loads
// of a whole batch of K steps are therefore issued up front as inline asm (kept in program order) and consumed behind
// hand-counted s_waitcnt vmcnt -- loads return in order, so waiting until only the 3 (NB - 1 - i) newer loads are
// outstanding guarantees the three operands of step i have landed (same technique as rel_head.hip).
template <int OFF>
__device__ __forceinline__ f32x4 gload16(const float4* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=&v"(v) : "v"(p), "n"(OFF));
  return v;
}
template <int N>
__device__ __forceinline__ void vmwait(f32x4& v) {
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v) : "n"(N));
}

template <int N, int I = 0, class Fn>
__device__ __forceinline__ void static_for_lin(Fn&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for_lin<N, I + 1>(f);
  }
}

// one batch of NB K-steps (4 floats each) starting at float4 index T0 of the three operand rows
template <int NB, int T0>
struct SkinnyBatch {
  template <int I>
  static __device__ __forceinline__ void issue(f32x4 (&a)[NB], f32x4 (&b0)[NB], f32x4 (&b1)[NB], const float4* xp,
                                               const float4* w0, const float4* w1) {
    a[I] = gload16<(T0 + I) * 16>(xp);
    b0[I] = gload16<(T0 + I) * 16>(w0);
    b1[I] = gload16<(T0 + I) * 16>(w1);
    if constexpr (I + 1 < NB) issue<I + 1>(a, b0, b1, xp, w0, w1);
  }
  template <int I>
  static __device__ __forceinline__ void consume(f32x4 (&a)[NB], f32x4 (&b0)[NB], f32x4 (&b1)[NB], f32x4& acc0,
                                                 f32x4& acc1) {
    vmwait<3 * (NB - 1 - I)>(a[I]);
    vmwait<3 * (NB - 1 - I)>(b0[I]);
    vmwait<3 * (NB - 1 - I)>(b1[I]);
    // D[row i = X row][col j = W row]: A[i][kk] = X[m0+i][k], B[kk][j] = W[n0+j][k]
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[I].x, b0[I].x, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[I].x, b1[I].x, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[I].y, b0[I].y, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[I].y, b1[I].y, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[I].z, b0[I].z, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[I].z, b1[I].z, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[I].w, b0[I].w, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[I].w, b1[I].w, acc1, 0, 0, 0);
    if constexpr (I + 1 < NB) consume<I + 1>(a, b0, b1, acc0, acc1);
  }
};

// the K loop of one (wave, lane group): SPAN floats in batches of up to 8 steps
template <int SPAN>
__device__ __forceinline__ void skinny_kloop(const float4* xp, const float4* w0, const float4* w1, f32x4& acc0,
                                             f32x4& acc1) {
  constexpr int STEPS = SPAN / 4, NB = STEPS < 8 ? STEPS : 8;
  static_assert(STEPS % NB == 0, "SPAN / 4 must be a multiple of the batch");
#pragma unroll
  for (int t0 = 0; t0 < STEPS; t0 += NB) {
    f32x4 a[NB], b0[NB], b1[NB];
    SkinnyBatch<NB, 0>::template issue<0>(a, b0, b1, xp + t0, w0 + t0, w1 + t0);
    SkinnyBatch<NB, 0>::template consume<0>(a, b0, b1, acc0, acc1);
  }
}

// SPAN = K / 16 floats handled by one (wave, lane-group); compile-time for full unrolling: 16 (K=256), 64 (K=1024),
// 32 (K=512); generic runtime variant below for other K % 64 == 0.
template <int SPAN>
__global__ __launch_bounds__(256) void linear_skinny_f32(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y,
                                                         int M, int K, int N, float alpha, int relu, int span_rt) {
  __shared__ float s_red[4 * 2 * 64 * 4];
  const int span = SPAN > 0 ? SPAN : span_rt;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 32;
  const int kb = (wave * 4 + g) * span;
  const int row = min(m0 + c, M - 1);
  const int wr0 = min(n0 + c, N - 1), wr1 = min(n0 + 16 + c, N - 1);
  const float4* xp = reinterpret_cast<const float4*>(x + (size_t)row * K + kb);
  const float4* w0 = reinterpret_cast<const float4*>(w + (size_t)wr0 * K + kb);
  const float4* w1 = reinterpret_cast<const float4*>(w + (size_t)wr1 * K + kb);
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  if constexpr (SPAN > 0) {
    skinny_kloop<SPAN>(xp, w0, w1, acc0, acc1);
  } else
#pragma unroll 4
  for (int t4 = 0; t4 < span / 4; ++t4) {
    const float4 a = xp[t4], b0 = w0[t4], b1 = w1[t4];
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b0.x, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b1.x, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b0.y, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b1.y, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b0.z, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b1.z, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b0.w, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b1.w, acc1, 0, 0, 0);
  }
  // cross-wave reduction through LDS: s_red[wave][sub][lane][r]
  reinterpret_cast<f32x4*>(s_red)[(wave * 2 + 0) * 64 + lane] = acc0;
  reinterpret_cast<f32x4*>(s_red)[(wave * 2 + 1) * 64 + lane] = acc1;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int e = tid + 256 * j;  // element of the 16 x 32 tile in accumulator order
    const int sub = e >> 8, ln = (e >> 2) & 63, r = e & 3;
    float v = s_red[((0 * 2 + sub) * 64 + ln) * 4 + r] + s_red[((1 * 2 + sub) * 64 + ln) * 4 + r] +
              s_red[((2 * 2 + sub) * 64 + ln) * 4 + r] + s_red[((3 * 2 + sub) * 64 + ln) * 4 + r];
    const int orow = m0 + (ln >> 4) * 4 + r, ocol = n0 + sub * 16 + (ln & 15);
    if (orow < M && ocol < N) {
      if (bias != nullptr) v += bias[ocol];
      v *= alpha;
      if (relu) v = egtr_relu(v);
      y[(size_t)orow * N + ocol] = v;
    }
  }
}

// Up to 16 independent skinny linears in one launch (blockIdx.z = group): the q / k / v projections of a decoder layer,
// the sampling_offsets + attention_weights pair, the 14 slot projections of the relation head, ...  Every kernel
// launch costs ~5 us in a graph-replayed forward, whatever its size; these layers are all launch-bound.
//   y_g[M_g, N_g] (row stride ldy_g) = act((alpha_x_g * X_g W_g^T + b_g) * alpha_g)
struct LinGroup {
  const float* x;
  const float* w;
  const float* b;
  float* y;
  int M, N, ldy, relu;
  float alpha_x, alpha;
  // LayerNorm prologue (K == 256 only; gamma == nullptr: none): the layer's input is LayerNorm(x + res) * gamma + beta
  // [+ pos[row % pos_rows]]; the workgroups of output-column tile 0 also store the LayerNorm result (without pos) to ln_out
  const float* res;
  const float* gamma;
  const float* beta;
  const float* pos;
  float* ln_out;
  int pos_rows;
  float eps;
};
struct LinGroups {
  LinGroup g[16];
};

template <int SPAN>
__global__ __launch_bounds__(256) void linear_skinny_grouped_f32(LinGroups P, int K, int span_rt) {
  __shared__ float s_red[4 * 2 * 64 * 4];
  const LinGroup& G = P.g[blockIdx.z];
  const int M = G.M, N = G.N;
  const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 32;
  if (m0 >= M || n0 >= N) return;  // uniform per workgroup
  const int span = SPAN > 0 ? SPAN : span_rt;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int kb = (wave * 4 + g) * span;
  const int row = min(m0 + c, M - 1);
  const int wr0 = min(n0 + c, N - 1), wr1 = min(n0 + 16 + c, N - 1);
  const float4* xp = reinterpret_cast<const float4*>(G.x + (size_t)row * K + kb);
  const float4* w0 = reinterpret_cast<const float4*>(G.w + (size_t)wr0 * K + kb);
  const float4* w1 = reinterpret_cast<const float4*>(G.w + (size_t)wr1 * K + kb);
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  if constexpr (SPAN == 16) {
    if (G.gamma != nullptr) {   // uniform per workgroup
      // ---- LayerNorm prologue (K = 256): the 16 (wave, lane group) parts of a row hold 16 channels each.  The decoder's
      // residual add + LayerNorm launches (4.8 us apiece at 200 rows: pure launch floor) disappear into their consumers;
      // every column tile recomputes the statistics of its 16 rows (8 KiB of input), tile 0 stores the result.
      __shared__ float s_stat[2][4][16];
      const float4* rp = reinterpret_cast<const float4*>(G.res + (size_t)row * K + kb);
      const float4* gp = reinterpret_cast<const float4*>(G.gamma + kb);
      const float4* bp = reinterpret_cast<const float4*>(G.beta + kb);
      const float4* pp = G.pos != nullptr ? reinterpret_cast<const float4*>(G.pos + (size_t)(row % G.pos_rows) * K + kb) : gp;
      f32x4 a[4], rr[4], b0[4], b1[4], ga[4], be[4], po[4];
      static_for_lin<4>([&](auto I) {
        constexpr int i = decltype(I)::value;
        a[i] = gload16<i * 16>(xp);
        rr[i] = gload16<i * 16>(rp);
      });
      static_for_lin<4>([&](auto I) {
        constexpr int i = decltype(I)::value;
        ga[i] = gload16<i * 16>(gp);
        be[i] = gload16<i * 16>(bp);
        po[i] = gload16<i * 16>(pp);
        b0[i] = gload16<i * 16>(w0);
        b1[i] = gload16<i * 16>(w1);
      });
      // the statistics start as soon as x and the residual (the 8 oldest of the 28 loads) have landed; the LayerNorm
      // parameters, the position rows and the weights are waited for where they are used
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        vmwait<20>(a[i]);
        vmwait<20>(rr[i]);
      }
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[i] += rr[i];
        sum += (a[i].x + a[i].y) + (a[i].z + a[i].w);
      }
      sum += __shfl_xor(sum, 16);
      sum += __shfl_xor(sum, 32);
      if (g == 0) s_stat[0][wave][c] = sum;
      __syncthreads();
      const float mean = ((s_stat[0][0][c] + s_stat[0][1][c]) + (s_stat[0][2][c] + s_stat[0][3][c])) * (1.f / 256.f);
      float sq = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[i] -= mean;
        sq += (a[i].x * a[i].x + a[i].y * a[i].y) + (a[i].z * a[i].z + a[i].w * a[i].w);
      }
      sq += __shfl_xor(sq, 16);
      sq += __shfl_xor(sq, 32);
      if (g == 0) s_stat[1][wave][c] = sq;
      __syncthreads();
      const float rstd =
          rsqrtf(((s_stat[1][0][c] + s_stat[1][1][c]) + (s_stat[1][2][c] + s_stat[1][3][c])) * (1.f / 256.f) + G.eps);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        vmwait<0>(ga[i]); vmwait<0>(be[i]); vmwait<0>(po[i]); vmwait<0>(b0[i]); vmwait<0>(b1[i]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = a[i] * rstd * ga[i] + be[i];
      if (blockIdx.x == 0 && G.ln_out != nullptr && m0 + c < M) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          *reinterpret_cast<float4*>(G.ln_out + (size_t)row * K + kb + 4 * i) = make_float4(a[i].x, a[i].y, a[i].z, a[i].w);
      }
      if (G.pos != nullptr) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] += po[i];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b0[i].x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b1[i].x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b0[i].y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b1[i].y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b0[i].z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b1[i].z, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b0[i].w, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b1[i].w, acc1, 0, 0, 0);
      }
    } else {
      skinny_kloop<SPAN>(xp, w0, w1, acc0, acc1);
    }
  } else if constexpr (SPAN > 0) {
    skinny_kloop<SPAN>(xp, w0, w1, acc0, acc1);
  } else
#pragma unroll 4
  for (int t4 = 0; t4 < span / 4; ++t4) {
    const float4 a = xp[t4], b0 = w0[t4], b1 = w1[t4];
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b0.x, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b1.x, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b0.y, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b1.y, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b0.z, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b1.z, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b0.w, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b1.w, acc1, 0, 0, 0);
  }
  reinterpret_cast<f32x4*>(s_red)[(wave * 2 + 0) * 64 + lane] = acc0;
  reinterpret_cast<f32x4*>(s_red)[(wave * 2 + 1) * 64 + lane] = acc1;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int e = tid + 256 * j;
    const int sub = e >> 8, ln = (e >> 2) & 63, r = e & 3;
    float v = s_red[((0 * 2 + sub) * 64 + ln) * 4 + r] + s_red[((1 * 2 + sub) * 64 + ln) * 4 + r] +
              s_red[((2 * 2 + sub) * 64 + ln) * 4 + r] + s_red[((3 * 2 + sub) * 64 + ln) * 4 + r];
    const int orow = m0 + (ln >> 4) * 4 + r, ocol = n0 + sub * 16 + (ln & 15);
    if (orow < M && ocol < N) {
      if (G.alpha_x != 1.f) v *= G.alpha_x;  // (alpha_x X) W^T: the product is scaled before the bias is added
      if (G.b != nullptr) v += G.b[ocol];
      v *= G.alpha;
      if (G.relu) v = egtr_relu(v);
      G.y[(size_t)orow * G.ldy + ocol] = v;
    }
  }
}

// ---- backward of the skinny linear layer in ONE launch -------------------------------------------------------------------
// With g' = alpha * g * [y > 0] (y: the layer's post-ReLU output, optional):
//   gx[M, K] = g' . W        (reduction over the N output features)
//   gw[N, K] = g'^T . x      (reduction over the M rows)
//   gb[N]    = column sums of g'
// autograd issues two vendor GEMMs (9 us each at M = 800: one or two workgroups' worth of work), a mask pass and a
// reduction per layer; ~70 such layers per train step are launch-bound.  Here the first `x_tiles` workgroups own
// 16 x 64 tiles of gx and the rest 16 x 16 tiles of gw; in both the reduction is split 16 ways over (wave, lane group) as
// in the forward kernel -- a lane group's "k" slot of v_mfma_f32_16x16x4_f32 carries its own span of the reduction -- and
// the operands whose reduction index is the SLOW one in memory (W for gx, g and x for gw) are read as float4 / float2
// along the OUTPUT index instead: the 4 (2) values feed 4 (2) MFMAs whose output columns interleave.
struct SkinnyBwdArgs {
  const float* g;
  const float* y;   // post-ReLU output or nullptr
  const float* x;
  const float* w;
  float* gx;        // nullable
  float* gw;        // nullable
  float* gb;        // nullable (needs gw tiles: computed by the k0 == 0 column of gw workgroups)
  int M, N, K, x_tiles;
  float alpha;
  const float* add1;   // nullable, [M, K]: added to gx in the epilogue (gradient branches that meet at x: no add launches)
  const float* add2;   // nullable, [M, K]
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
// loads of a whole batch are issued up front (asm keeps them in program order and out of the compiler's reach: it
// otherwise sinks every load next to its first use, one memory round trip per reduction step) and consumed behind one
// s_waitcnt tied to each destination register
__device__ __forceinline__ f32x4 gload_x4(const float* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v) : "v"(p));
  return v;
}
__device__ __forceinline__ f32x2 gload_x2(const float* p) {
  f32x2 v;
  asm volatile("global_load_dwordx2 %0, %1, off" : "=&v"(v) : "v"(p));
  return v;
}
__device__ __forceinline__ float gload_x1(const float* p) {
  float v;
  asm volatile("global_load_dword %0, %1, off" : "=&v"(v) : "v"(p));
  return v;
}
__device__ __forceinline__ void vm_wait0(float& v) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(v)); }
__device__ __forceinline__ void vm_wait0(f32x4& v) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(v)); }
__device__ __forceinline__ void vm_wait0(f32x2& v) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(v)); }

// MASK: ReLU mask from y; B4: float4 steps of the gx reduction per batch (divides N / 64)
// Workgroups of EIGHT waves: a gw tile's reduction over the M rows is split 32 ways (round 4: 16 ways over four waves left every
// lane group a chain of M / 16 rows = two dependent load batches at M = 800 -- the launch's critical path, 10.4 us against 4.2 us
// for the gx tiles; 25 rows are one batch).  gx tiles keep their four-wave form: waves 4..7 of those workgroups leave at once.
constexpr int kBwdWaves = 8;
template <bool MASK, int B4, bool GW32>
__global__ __launch_bounds__(64 * kBwdWaves) void linear_skinny_bwd_f32(SkinnyBwdArgs P) {
  __shared__ float s_red[kBwdWaves * 4 * 64 * 4];
  __shared__ float s_gb[4 * kBwdWaves * 32];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, gq = lane >> 4;
  const int grp = wave * 4 + gq;   // its share of the reduction: 0..15 (gx tiles), 0..31 (gw tiles)
  const int M = P.M, N = P.N, K = P.K;
  f32x4 acc[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) acc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
  if ((int)blockIdx.x < P.x_tiles) {
    if (wave >= 4) return;   // (a finished wave no longer counts at the workgroup's barriers)
    // ---- gx tile: rows m0..m0+15, columns k0..k0+63; reduction n in [grp * span, (grp + 1) * span)
    const int ktiles = K / 64;
    const int m0 = ((int)blockIdx.x / ktiles) * 16, k0 = ((int)blockIdx.x % ktiles) * 64;
    const int span = N / 16;
    const int row = min(m0 + c, M - 1);
    const float* gp = P.g + (size_t)row * N + grp * span;
    const float* yp = P.y + (size_t)row * N + grp * span;
    const float* wp = P.w + (size_t)(grp * span) * K + k0 + 4 * c;
#pragma unroll 1
    for (int t0 = 0; t0 < span; t0 += 4 * B4) {
      f32x4 a[B4], mk[B4], b[B4][4];
#pragma unroll
      for (int q = 0; q < B4; ++q) {
        a[q] = gload_x4(gp + t0 + 4 * q);
        if (MASK) mk[q] = gload_x4(yp + t0 + 4 * q);
#pragma unroll
        for (int t = 0; t < 4; ++t) b[q][t] = gload_x4(wp + (size_t)(t0 + 4 * q + t) * K);
      }
#pragma unroll
      for (int q = 0; q < B4; ++q) {
        vm_wait0(a[q]);
        if (MASK) vm_wait0(mk[q]);
#pragma unroll
        for (int t = 0; t < 4; ++t) vm_wait0(b[q][t]);
      }
#pragma unroll
      for (int q = 0; q < B4; ++q)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float av = (MASK && !(mk[q][t] > 0.f)) ? 0.f : a[q][t];
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[q][t].x, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[q][t].y, acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[q][t].z, acc[2], 0, 0, 0);
          acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[q][t].w, acc[3], 0, 0, 0);
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) reinterpret_cast<f32x4*>(s_red)[(wave * 4 + e) * 64 + lane] = acc[e];
    __syncthreads();
    // thread (ln, r): row m0 + 4 (ln >> 4) + r, columns k0 + 4 (ln & 15) + e
    const int ln = tid >> 2, r = tid & 3;
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
      o[e] = s_red[((0 * 4 + e) * 64 + ln) * 4 + r] + s_red[((1 * 4 + e) * 64 + ln) * 4 + r] +
             s_red[((2 * 4 + e) * 64 + ln) * 4 + r] + s_red[((3 * 4 + e) * 64 + ln) * 4 + r];
    const int orow = m0 + (ln >> 4) * 4 + r;
    if (orow < M) {
      const size_t at = (size_t)orow * K + k0 + 4 * (ln & 15);
      float4 r = make_float4(o[0] * P.alpha, o[1] * P.alpha, o[2] * P.alpha, o[3] * P.alpha);
      if (P.add1 != nullptr) {
        const float4 a = *reinterpret_cast<const float4*>(P.add1 + at);
        r = make_float4(r.x + a.x, r.y + a.y, r.z + a.z, r.w + a.w);
      }
      if (P.add2 != nullptr) {
        const float4 a = *reinterpret_cast<const float4*>(P.add2 + at);
        r = make_float4(r.x + a.x, r.y + a.y, r.z + a.z, r.w + a.w);
      }
      *reinterpret_cast<float4*>(P.gx + at) = r;
    }
    return;
  }
  if (GW32) {
  // ---- gw tile, 32 x 32 form (layers with more than 512 16 x 16 tiles: fc1 / fc2): rows n0..n0+31, columns k0..k0+31; four
  // accumulators per wave; reduction m in [grp * rpg, (grp + 1) * rpg)
  const int bt = (int)blockIdx.x - P.x_tiles;
  const int ktiles = K / 32;
  const int n0 = (bt / ktiles) * 32, k0 = (bt % ktiles) * 32;
  constexpr int kGroups = 4 * kBwdWaves;
  const int rpg = (M + kGroups - 1) / kGroups;
  const int mb = grp * rpg;
  const float* gp = P.g + n0 + 2 * c;
  const float* yp = P.y + n0 + 2 * c;
  const float* xp = P.x + k0 + 2 * c;
  float sb0 = 0.f, sb1 = 0.f;
  constexpr int TB = 32;   // rows per batch
  // the same trip count in every lane group (the matrix instruction runs on all 64 lanes): rows past the group's share
  // or past M are read from a valid row and contribute zeros
#pragma unroll 1
  for (int t0 = 0; t0 < rpg; t0 += TB) {
    f32x2 a[TB], mk[TB], b[TB];
#pragma unroll
    for (int t = 0; t < TB; ++t) {
      const bool ok = t0 + t < rpg && mb + t0 + t < M;
      const size_t m = (size_t)(ok ? mb + t0 + t : M - 1);
      a[t] = gload_x2(gp + m * N);
      if (MASK) mk[t] = gload_x2(yp + m * N);
      b[t] = gload_x2(xp + m * K);
    }
#pragma unroll
    for (int t = 0; t < TB; ++t) {
      vm_wait0(a[t]);
      if (MASK) vm_wait0(mk[t]);
      vm_wait0(b[t]);
    }
#pragma unroll
    for (int t = 0; t < TB; ++t) {
      const bool ok = t0 + t < rpg && mb + t0 + t < M;
      const float ax = (!ok || (MASK && !(mk[t].x > 0.f))) ? 0.f : a[t].x;
      const float ay = (!ok || (MASK && !(mk[t].y > 0.f))) ? 0.f : a[t].y;
      sb0 += ax;
      sb1 += ay;
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax, b[t].x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax, b[t].y, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(ay, b[t].x, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(ay, b[t].y, acc[3], 0, 0, 0);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) reinterpret_cast<f32x4*>(s_red)[(wave * 4 + e) * 64 + lane] = acc[e];
  s_gb[grp * 32 + 2 * c] = sb0;
  s_gb[grp * 32 + 2 * c + 1] = sb1;
  __syncthreads();
  if (P.gw != nullptr && tid < 256) {
    // accumulator e = 2 ei + ej of lane (c', gq') register r: row n0 + 2 (4 gq' + r) + ei, column k0 + 2 c' + ej
    const int ln = tid >> 2, r = tid & 3;
#pragma unroll
    for (int ei = 0; ei < 2; ++ei) {
      float o[2];
#pragma unroll
      for (int ej = 0; ej < 2; ++ej) {
        const int e = 2 * ei + ej;
        float acc_o = 0.f;
#pragma unroll
        for (int wq = 0; wq < kBwdWaves; ++wq) acc_o += s_red[((wq * 4 + e) * 64 + ln) * 4 + r];   // fixed order
        o[ej] = acc_o;
      }
      const int orow = n0 + 2 * (4 * (ln >> 4) + r) + ei;
      *reinterpret_cast<float2*>(P.gw + (size_t)orow * K + k0 + 2 * (ln & 15)) = make_float2(o[0] * P.alpha, o[1] * P.alpha);
    }
  }
  if (P.gb != nullptr && k0 == 0 && tid < 32) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < kGroups; ++q) s += s_gb[q * 32 + tid];
    P.gb[n0 + tid] = s * P.alpha;
  }
  return;
  }
  // ---- gw tile: rows n0..n0+15, columns k0..k0+15 -- ONE 16 x 16 accumulator per wave; reduction m in [grp * rpg, (grp + 1) * rpg).
  // (Round 4.  32 x 32 tiles were (N / 32)(K / 32) = 64 workgroups for a 256 x 256 layer -- a quarter of the chip -- each with
  // 800 matrix instructions to issue: the launch's critical path at 10 us.  16 x 16 tiles are 256 workgroups of 200.)
  const int bt = (int)blockIdx.x - P.x_tiles;
  const int ktiles = K / 16;
  const int n0 = (bt / ktiles) * 16, k0 = (bt % ktiles) * 16;
  constexpr int kGroups = 4 * kBwdWaves;
  const int rpg = (M + kGroups - 1) / kGroups;
  const int mb = grp * rpg;
  const float* gp = P.g + n0 + c;
  const float* yp = P.y + n0 + c;
  const float* xp = P.x + k0 + c;
  float sb = 0.f;
  f32x4 accw = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int TB = 32;   // rows per batch (800 rows = 25 per lane group: one memory round trip)
  // the same trip count in every lane group (the matrix instruction runs on all 64 lanes): rows past the group's share
  // or past M are read from a valid row and contribute zeros
#pragma unroll 1
  for (int t0 = 0; t0 < rpg; t0 += TB) {
    float a[TB], mk[TB], b[TB];
#pragma unroll
    for (int t = 0; t < TB; ++t) {
      const bool ok = t0 + t < rpg && mb + t0 + t < M;
      const size_t m = (size_t)(ok ? mb + t0 + t : M - 1);
      a[t] = gload_x1(gp + m * N);
      if (MASK) mk[t] = gload_x1(yp + m * N);
      b[t] = gload_x1(xp + m * K);
    }
#pragma unroll
    for (int t = 0; t < TB; ++t) {
      vm_wait0(a[t]);
      if (MASK) vm_wait0(mk[t]);
      vm_wait0(b[t]);
    }
#pragma unroll
    for (int t = 0; t < TB; ++t) {
      const bool ok = t0 + t < rpg && mb + t0 + t < M;
      const float av = (!ok || (MASK && !(mk[t] > 0.f))) ? 0.f : a[t];
      sb += av;
      accw = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[t], accw, 0, 0, 0);
    }
  }
  reinterpret_cast<f32x4*>(s_red)[wave * 64 + lane] = accw;
  s_gb[grp * 16 + c] = sb;
  __syncthreads();
  if (P.gw != nullptr && tid < 256) {
    // lane (c', gq') register r of a wave's accumulator: row n0 + 4 gq' + r, column k0 + c'
    const int ln = tid >> 2, r = tid & 3;
    float o = 0.f;
#pragma unroll
    for (int wq = 0; wq < kBwdWaves; ++wq) o += s_red[(wq * 64 + ln) * 4 + r];   // fixed order
    P.gw[(size_t)(n0 + 4 * (ln >> 4) + r) * K + k0 + (ln & 15)] = o * P.alpha;
  }
  if (P.gb != nullptr && k0 == 0 && tid < 16) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < kGroups; ++q) s += s_gb[q * 16 + tid];
    P.gb[n0 + tid] = s * P.alpha;
  }
}

template <bool MASK, bool GW32>
void launch_skinny_bwd2(hipStream_t st, const SkinnyBwdArgs& P, int grid) {
  const int steps = P.N / 64;   // float4 steps of the gx reduction per lane group
  if (steps % 4 == 0)
    hipLaunchKernelGGL((linear_skinny_bwd_f32<MASK, 4, GW32>), dim3(grid), dim3(64 * kBwdWaves), 0, st, P);
  else if (steps % 2 == 0)
    hipLaunchKernelGGL((linear_skinny_bwd_f32<MASK, 2, GW32>), dim3(grid), dim3(64 * kBwdWaves), 0, st, P);
  else
    hipLaunchKernelGGL((linear_skinny_bwd_f32<MASK, 1, GW32>), dim3(grid), dim3(64 * kBwdWaves), 0, st, P);
}

template <bool MASK>
void launch_skinny_bwd(hipStream_t st, const SkinnyBwdArgs& P, int grid, bool gw32) {
  if (gw32)
    launch_skinny_bwd2<MASK, true>(st, P, grid);
  else
    launch_skinny_bwd2<MASK, false>(st, P, grid);
}

}  // namespace

extern "C" int egtr_linear_grouped_ln_f32(egtr_stream_t stream, int num_groups, const float* const* x,
                                          const float* const* w, const float* const* bias, float* const* y, const int* M,
                                          const int* N, const int* ldy, const float* alpha_x, const float* alpha,
                                          const int* relu, int K, const float* const* ln_residual,
                                          const float* const* ln_gamma, const float* const* ln_beta, const float* ln_eps,
                                          const float* const* pos, const int* pos_rows, float* const* ln_out) {
  if (!x || !w || !bias || !y || !M || !N || !ldy || !alpha_x || !alpha || !relu) return EGTR_E_ARG;
  if (num_groups <= 0 || num_groups > 16 || K <= 0) return EGTR_E_ARG;
  if (K % 64 != 0) return EGTR_E_UNSUPPORTED;
  const bool any_ln = ln_gamma != nullptr;
  if (any_ln && (!ln_residual || !ln_beta || !ln_eps || !pos || !pos_rows || !ln_out)) return EGTR_E_ARG;
  LinGroups P;
  int maxM = 0, maxN = 0;
  for (int i = 0; i < 16; ++i) {
    const int s = i < num_groups ? i : 0;
    if (!x[s] || !w[s] || !y[s] || M[s] <= 0 || N[s] <= 0 || ldy[s] < N[s]) return EGTR_E_ARG;
    P.g[i] = LinGroup{x[s], w[s], bias[s], y[s], M[s], N[s], ldy[s], relu[s], alpha_x[s], alpha[s],
                      nullptr, nullptr, nullptr, nullptr, nullptr, 1, 0.f};
    if (any_ln && ln_gamma[s] != nullptr) {
      if (K != 256) return EGTR_E_UNSUPPORTED;
      if (!ln_residual[s] || !ln_beta[s] || (pos[s] != nullptr && pos_rows[s] <= 0)) return EGTR_E_ARG;
      LinGroup& g = P.g[i];
      g.res = ln_residual[s]; g.gamma = ln_gamma[s]; g.beta = ln_beta[s]; g.pos = pos[s]; g.ln_out = ln_out[s];
      g.pos_rows = pos[s] != nullptr ? pos_rows[s] : 1;
      g.eps = ln_eps[s];
    }
    if (i < num_groups) {
      maxM = M[s] > maxM ? M[s] : maxM;
      maxN = N[s] > maxN ? N[s] : maxN;
    }
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((maxN + 31) / 32, (maxM + 15) / 16, num_groups);
  const int span = K / 16;
  if (span == 16)
    hipLaunchKernelGGL(linear_skinny_grouped_f32<16>, grid, dim3(256), 0, st, P, K, span);
  else if (span == 64)
    hipLaunchKernelGGL(linear_skinny_grouped_f32<64>, grid, dim3(256), 0, st, P, K, span);
  else
    hipLaunchKernelGGL(linear_skinny_grouped_f32<0>, grid, dim3(256), 0, st, P, K, span);
  return egtr_check_launch();
}

extern "C" int egtr_linear_grouped_f32(egtr_stream_t stream, int num_groups, const float* const* x,
                                       const float* const* w, const float* const* bias, float* const* y, const int* M,
                                       const int* N, const int* ldy, const float* alpha_x, const float* alpha,
                                       const int* relu, int K) {
  return egtr_linear_grouped_ln_f32(stream, num_groups, x, w, bias, y, M, N, ldy, alpha_x, alpha, relu, K, nullptr, nullptr,
                                    nullptr, nullptr, nullptr, nullptr, nullptr);
}

extern "C" int egtr_linear_f32(egtr_stream_t stream, const float* x, const float* w, const float* bias, float* y,
                               int M, int K, int N, float alpha, int relu) {
  if (!x || !w || !y) return EGTR_E_ARG;
  if (M <= 0 || K <= 0 || N <= 0) return EGTR_E_ARG;
  if (K % 64 != 0) return EGTR_E_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((N + 31) / 32, (M + 15) / 16);
  const int span = K / 16;
  if (span == 16)
    hipLaunchKernelGGL(linear_skinny_f32<16>, grid, dim3(256), 0, st, x, w, bias, y, M, K, N, alpha, relu, span);
  else if (span == 32)
    hipLaunchKernelGGL(linear_skinny_f32<32>, grid, dim3(256), 0, st, x, w, bias, y, M, K, N, alpha, relu, span);
  else if (span == 64)
    hipLaunchKernelGGL(linear_skinny_f32<64>, grid, dim3(256), 0, st, x, w, bias, y, M, K, N, alpha, relu, span);
  else
    hipLaunchKernelGGL(linear_skinny_f32<0>, grid, dim3(256), 0, st, x, w, bias, y, M, K, N, alpha, relu, span);
  return egtr_check_launch();
}

extern "C" int egtr_linear_backward_acc_f32(egtr_stream_t stream, const float* grad_y, const float* relu_output,
                                            const float* x, const float* w, float alpha, float* grad_x, float* grad_w,
                                            float* grad_bias, int M, int K, int N, const float* grad_x_add1,
                                            const float* grad_x_add2) {
  if (!grad_y || !x || !w || M <= 0 || K <= 0 || N <= 0) return EGTR_E_ARG;
  if (!grad_x && !grad_w && !grad_bias) return EGTR_E_ARG;
  if ((grad_x_add1 || grad_x_add2) && !grad_x) return EGTR_E_ARG;
  if (K % 64 != 0 || N % 64 != 0) return EGTR_E_UNSUPPORTED;
  const uintptr_t al = reinterpret_cast<uintptr_t>(grad_y) | reinterpret_cast<uintptr_t>(relu_output) |
                       reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) |
                       reinterpret_cast<uintptr_t>(grad_x) | reinterpret_cast<uintptr_t>(grad_w) |
                       reinterpret_cast<uintptr_t>(grad_x_add1) | reinterpret_cast<uintptr_t>(grad_x_add2);
  if (al & 15) return EGTR_E_UNSUPPORTED;
  SkinnyBwdArgs P;
  P.g = grad_y; P.y = relu_output; P.x = x; P.w = w; P.gx = grad_x; P.gw = grad_w; P.gb = grad_bias;
  P.M = M; P.N = N; P.K = K; P.alpha = alpha; P.add1 = grad_x_add1; P.add2 = grad_x_add2;
  P.x_tiles = grad_x ? ((M + 15) / 16) * (K / 64) : 0;
  // gw tiles: 16 x 16 while that is at most two workgroups per CU, 32 x 32 (four accumulators per wave) for the wide layers
  const bool gw32 = (N / 16) * (K / 16) > 512;
  const int w_tiles = (grad_w || grad_bias) ? (gw32 ? (N / 32) * (K / 32) : (N / 16) * (K / 16)) : 0;
  if (relu_output)
    launch_skinny_bwd<true>(static_cast<hipStream_t>(stream), P, P.x_tiles + w_tiles, gw32);
  else
    launch_skinny_bwd<false>(static_cast<hipStream_t>(stream), P, P.x_tiles + w_tiles, gw32);
  return egtr_check_launch();
}

extern "C" int egtr_linear_backward_f32(egtr_stream_t stream, const float* grad_y, const float* relu_output,
                                        const float* x, const float* w, float alpha, float* grad_x, float* grad_w,
                                        float* grad_bias, int M, int K, int N) {
  return egtr_linear_backward_acc_f32(stream, grad_y, relu_output, x, w, alpha, grad_x, grad_w, grad_bias, M, K, N, nullptr,
                                      nullptr);
}
