// EGTR relation head for gfx950: pairwise gate -> gated sum over slots -> two 3-layer MLPs, fused.
//
// Replaces model/egtr.py:366-416 (plain PyTorch in the reference), which materialises
// relation_source [B, N, N, Ld+1, 2d] (573 MB fp32 at N = 200, Ld = 6) plus same-sized temporaries.
//
// Algebra (exact identities; only fp32 re-association differs -- DESIGN.md "relation head"):
//   src[i,j,t]  = [q^[i,t] ; k^[j,t]]                      (t = decoder layer slots + the final-hidden slot)
//   g[i,j,t]    = sigmoid(w_g . src + b_g) = sigmoid(gate_q[i,t] + gate_k[j,t])            (separable logit)
//   W1 . sum_t g src = sum_t g[i,j,t] (uq[i,t,:] + uk[j,t,:])   with uq = W1[:, :d] q^, uk = W1[:, d:] k^
// so the N^2 x (Ld+1) x 2d tensor never exists: live inputs are O(N (Ld+1) d), outputs O(N^2 R).
//
// Kernel shape: a workgroup of TWO wavefronts owns a tile of 32 pairs -- 8 subjects i x 4 objects j -- of ONE of the two
// MLPs (blockIdx.y): layer 1 then needs only 8 + 4 per-query row sets instead of the 1 + 32 of 32 consecutive pairs.
// The two waves split layer 1 by subjects (4 each) and layers 2 / 3 by the hidden-2 dimension (n tiles 0-3 / 4-7), the
// partial layer-3 sums meet in LDS: 5000 half-size work units on 1024 SIMDs (4.9 rounds of 5) instead of 2500 whole
// ones (2.4 rounds of 3).
//   layer 1 (VALU): lane (pair = l&31, half = l>>5) builds the 128 hidden-1 channels {128*half + s} of its pair
//                   directly in the register layout the MFMA wants -- no LDS round trip.
//   layer 2 (MFMA, v_mfma_f32_32x32x2_f32, exact f32): computed transposed, h2^T = W2 h1^T, so D[row = n][col =
//                   pair]; A = W2[n0 + (l&31)][128*half + s], B = the lane's own h1 value; 128 steps per 32-wide
//                   n tile.  Bias + ReLU on the accumulator.
//   layer 3: relation MLP: rel^T = W3 h2^T again on MFMA, consuming the accumulator registers of layer 2 as the
//            B operand in place (k-slot `half` at step r <-> n = n0 + (r&3) + 8(r>>2) + 4 half);
//            connectivity MLP (1 output): VALU dot product + one cross-half shuffle.
//   epilogue: relation tile staged through LDS and written as one contiguous run of 32*R floats, with the
//            Neural-Motifs frequency bias triplet_dist[cls_i, cls_j, :] gathered and added on the way out
//            (egtr.py:405-413).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"
#include "xs_format.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

constexpr int kHd = 256;  // hidden width of both MLPs (= d_model)
constexpr int kH1Stride = 260;  // LDS row pitch (floats) of the layer-1 transpose: 256 + 4

typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int OFF>
__device__ __forceinline__ f32x4v gload_b128(const float4* p) {
  f32x4v v;
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=&v"(v) : "v"(p), "n"(OFF));
  return v;
}
template <int N>
__device__ __forceinline__ void vm_wait(f32x4v& v) {
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v) : "n"(N));
}

// wait for load t of a set whose load 0 has N_T0 newer own loads (one fewer per later t); t is a compile-time constant
// after unrolling, so the switch folds to a single s_waitcnt
template <int N_T0>
__device__ __forceinline__ void vm_wait1(f32x4v& c, int t) {
#define EGTR_VMW(K) case K: asm volatile("s_waitcnt vmcnt(%1)" : "+v"(c) : "n"(N_T0 - K > 0 ? N_T0 - K : 0)); break;
  switch (t) {
    EGTR_VMW(0) EGTR_VMW(1) EGTR_VMW(2) EGTR_VMW(3) EGTR_VMW(4) EGTR_VMW(5) EGTR_VMW(6) EGTR_VMW(7) EGTR_VMW(8)
    EGTR_VMW(9)
    default: asm volatile("s_waitcnt vmcnt(0)" : "+v"(c)); break;
  }
#undef EGTR_VMW
}

// Layer-2 inner product of one 32-wide n tile: 32 float4 of the lane's W2 row, each feeding 4 MFMAs, with the row
// loads running PRE float4 ahead (template recursion = full unroll with compile-time offsets / wait counts).
template <int PRE>
struct PrefetchRow {
  template <int I>
  static __device__ __forceinline__ void issue(f32x4v (&wq)[PRE], const float4* wrow) {
    wq[I] = gload_b128<I * 16>(wrow);
    if constexpr (I + 1 < PRE) issue<I + 1>(wq, wrow);
  }
  template <int S4>
  static __device__ __forceinline__ void run(f32x4v (&wq)[PRE], const float4* wrow, const float (&h1)[128], f32x16& acc) {
    constexpr int newer = (31 - S4) < (PRE - 1) ? (31 - S4) : (PRE - 1);  // own loads issued after this one
    vm_wait<newer>(wq[S4 % PRE]);
    const f32x4v w = wq[S4 % PRE];
    acc = mfma32(w.x, h1[4 * S4 + 0], acc);
    acc = mfma32(w.y, h1[4 * S4 + 1], acc);
    acc = mfma32(w.z, h1[4 * S4 + 2], acc);
    acc = mfma32(w.w, h1[4 * S4 + 3], acc);
    if constexpr (S4 + PRE < 32) wq[S4 % PRE] = gload_b128<(S4 + PRE) * 16>(wrow);
    if constexpr (S4 + 1 < 32) run<S4 + 1>(wq, wrow, h1, acc);
  }
};

// T = number of slots (decoder layers + 1), OT = number of 32-wide relation-output tiles (R <= 32*OT).
constexpr int kRhWaves = 2;

template <int T, int OT>
__global__ __launch_bounds__(64 * kRhWaves) void rel_head_fwd_f32(
    const float* __restrict__ gate_q, const float* __restrict__ gate_k, const float* __restrict__ uq,
    const float* __restrict__ uk, const float* __restrict__ b1, const float* __restrict__ w2r,
    const float* __restrict__ b2r, const float* __restrict__ w3r, const float* __restrict__ b3r,
    const float* __restrict__ w2c, const float* __restrict__ b2c, const float* __restrict__ w3c,
    const float* __restrict__ b3c, const float* __restrict__ triplet, const int64_t* __restrict__ node_cls, int B,
    int N, int R, int C1, float* __restrict__ rel_logits, float* __restrict__ conn_logits,
    float* __restrict__ gate_mean, float* __restrict__ h1_save, float* __restrict__ h2_save) {
  // h1_save / h2_save (training only, may be null): post-ReLU hidden activations of both layers, [2 (mlp)][B*N*N][256],
  // so that the backward needs no recomputation (egtr_rel_head_backward_pairs_f32 + rocBLAS GEMMs, egtr_amd/ops.py)
  // one buffer, two lives: the layer-1 transpose (read back into registers before layer 2), then the output tile
  constexpr int kBuf = 32 * kH1Stride > 32 * (32 * OT + 1) ? 32 * kH1Stride : 32 * (32 * OT + 1);
  __shared__ __attribute__((aligned(16))) float s_buf[kBuf];
  float* const s_out = s_buf;
  float* const s_h1 = s_buf;
  __shared__ int s_tb[32];
  __shared__ long long s_pp[32];
  __shared__ float s_cacc[kRhWaves][32];
  const int lane = threadIdx.x & 63, pi = lane & 31, hf = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave of the workgroup
  const int mlp = blockIdx.y;  // 0 = relation, 1 = connectivity (wave-uniform)
  const long long total = (long long)B * N * N;
  // tile = (image, 8 subjects i0.., 4 objects j0..); pair pi of the tile = (i0 + (pi >> 2), j0 + (pi & 3))
  const int tj = (N + 3) >> 2, ti = (N + 7) >> 3;
  const int b = blockIdx.x / (ti * tj);
  const int trem = blockIdx.x - b * ti * tj;
  const int i0 = (trem / tj) * 8, j0 = (trem - (trem / tj) * tj) * 4;
  const int i_raw = i0 + (pi >> 2), j_raw = j0 + (pi & 3);
  const bool valid = i_raw < N && j_raw < N;
  const int i = i_raw < N ? i_raw : N - 1, j = j_raw < N ? j_raw : N - 1;   // clamped: loads stay in range
  const long long p = ((long long)b * N + i) * N + j;
  const size_t qi = (size_t)b * N + i, kj = (size_t)b * N + j;

  float g[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const float x = gate_q[qi * T + t] + gate_k[kj * T + t];
    g[t] = 1.f / (1.f + expf(-x));
  }
  if (gate_mean != nullptr && mlp == 0 && wv == 0) {  // rel_gate_{t} logging (egtr.py:496-505)
#pragma unroll
    for (int t = 0; t < T; ++t) {
      float v = (valid && hf == 0) ? g[t] : 0.f;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
      if (lane == 0) unsafeAtomicAdd(gate_mean + t, v / (float)total);
    }
  }

  // ---- layer 1: h1[pair][ch] = relu(b1 + sum_t g[pair,t] (uq[i,t,ch] + uk[j,t,ch])), ch in this MLP's 256 ------------
  // The wave walks its 32 pairs one at a time with lane = channel quad (4 l .. 4 l + 3), so every uq / uk row read is
  // one contiguous 1 KiB per wave, and the 2T row loads of pair p+1 are in flight while pair p is accumulated (inline
  // asm: hipcc would sink them back next to their uses).  The result goes through LDS (rows padded to 260 floats:
  // conflict-free ds_read_b128) into the layout the MFMA wants: lane (pair, half) holds the 128 channels
  // {128 half + s}.  (First version: lane = pair reading its own rows, 16 B from 64 different lines per instruction,
  // each used at once -- 45 % of the kernel's time went into waiting for those loads.)
  float h1[128];
  {
    // The tile's 32 pairs share 8 uq row sets (one per subject) and 4 uk row sets (one per object): the 4 uk sets are
    // loaded once up front (lane = channel quad: every row read is one contiguous 1 KiB per wave), the uq set of the
    // next subject is in flight while the current subject's 4 pairs are accumulated.  The result goes through LDS
    // (32 rows padded to 260 floats: conflict-free ds_read_b128) into the layout the MFMA wants: lane (pair, half)
    // holds the 128 channels {128 half + s} of its pair.
    const int krow_l = (int)kj;
    const float4 bias4 = reinterpret_cast<const float4*>(b1 + mlp * kHd)[lane];
    const float4* uq4 = reinterpret_cast<const float4*>(uq) + mlp * (kHd / 4) + lane;
    const float4* uk4 = reinterpret_cast<const float4*>(uk) + mlp * (kHd / 4) + lane;
    constexpr int ROW4 = 2 * kHd / 4;  // float4 per (row, slot)
    float4 kk[4][T];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int kr = __builtin_amdgcn_readlane(krow_l, jj);       // pair jj = (i0, j0 + jj)
      const float4* pk = uk4 + (size_t)kr * T * ROW4;
#pragma unroll
      for (int t = 0; t < T; ++t) kk[jj][t] = pk[t * ROW4];
    }
    // subject row of tile slot ii, by wave-uniform arithmetic (clamped like the lanes' own rows)
    auto qrow_of = [&](int ii) { return b * N + (i0 + ii < N ? i0 + ii : N - 1); };
    const int ii0 = wv * (8 / kRhWaves);   // this wave's subjects
    float4 ua[2][T];
    {
      const float4* pq = uq4 + (size_t)qrow_of(ii0) * T * ROW4;
#pragma unroll
      for (int t = 0; t < T; ++t) ua[0][t] = pq[t * ROW4];
    }
#pragma unroll
    for (int iu = 0; iu < 8 / kRhWaves; ++iu) {
      const int ii = ii0 + iu;
      if (iu + 1 < 8 / kRhWaves) {
        const float4* pq = uq4 + (size_t)qrow_of(ii + 1) * T * ROW4;
#pragma unroll
        for (int t = 0; t < T; ++t) ua[(iu + 1) & 1][t] = pq[t * ROW4];
      }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int pp = ii * 4 + jj;
        float4 acc = bias4;
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const float4 a = ua[iu & 1][t];
          const float4 c = kk[jj][t];
          const float gt = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g[t]), pp));
          acc.x += gt * (a.x + c.x);
          acc.y += gt * (a.y + c.y);
          acc.z += gt * (a.z + c.z);
          acc.w += gt * (a.w + c.w);
        }
        const float4 hv = make_float4(egtr_relu(acc.x), egtr_relu(acc.y), egtr_relu(acc.z), egtr_relu(acc.w));
        *reinterpret_cast<float4*>(&s_h1[pp * kH1Stride + 4 * lane]) = hv;
        if (h1_save != nullptr && i0 + ii < N && j0 + jj < N)  // one contiguous 1 KiB row per wave
          reinterpret_cast<float4*>(h1_save + ((size_t)mlp * total +
                                               (size_t)(((long long)b * N + i0 + ii) * N + j0 + jj)) * kHd)[lane] = hv;
      }
    }
    __syncthreads();   // both waves' rows are in LDS
    {
      const float4* hp = reinterpret_cast<const float4*>(&s_h1[pi * kH1Stride + hf * 128]);
#pragma unroll
      for (int s4 = 0; s4 < 32; ++s4) {
        const float4 v = hp[s4];
        h1[4 * s4 + 0] = v.x;
        h1[4 * s4 + 1] = v.y;
        h1[4 * s4 + 2] = v.z;
        h1[4 * s4 + 3] = v.w;
      }
    }
    __syncthreads();   // the buffer is reused for the output tiles
  }

  const float* w2 = mlp ? w2c : w2r;
  const float* b2 = mlp ? b2c : b2r;
  f32x16 racc[OT];
#pragma unroll
  for (int ot = 0; ot < OT; ++ot)
#pragma unroll
    for (int r = 0; r < 16; ++r) racc[ot][r] = 0.f;
  float cacc = 0.f;

  constexpr int kNtPerWave = kHd / 32 / kRhWaves;
#pragma unroll 1
  for (int nt = wv * kNtPerWave; nt < (wv + 1) * kNtPerWave; ++nt) {
    // ---- layer 2: h2^T[n][pair], n = 32 nt + (r&3) + 8 (r>>2) + 4 hf ---------------------------------------
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // W2 row of this lane, software-pipelined kPre float4 ahead of the MFMAs that consume it: with one wave per SIMD
    // nothing else hides the L2 latency, and hipcc sinks ordinary loads back to 1-2 in flight (the matrix pipe idled 2/3
    // of the time).  The loads are inline asm (kept in program order) with hand-counted s_waitcnt; loads return in
    // order, so compiler-issued loads in between only make these waits more conservative.
    const float4* wrow = reinterpret_cast<const float4*>(w2 + (size_t)(nt * 32 + pi) * kHd + hf * 128);
    constexpr int kPre = 8;
    f32x4v wq[kPre];
    PrefetchRow<kPre>::template issue<0>(wq, wrow);
    PrefetchRow<kPre>::template run<0>(wq, wrow, h1, acc);
    // bias + relu; 4 consecutive n per register quad
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const int n0 = nt * 32 + 8 * rq + 4 * hf;
      const float4 bb = *reinterpret_cast<const float4*>(b2 + n0);
      acc[4 * rq + 0] = egtr_relu(acc[4 * rq + 0] + bb.x);
      acc[4 * rq + 1] = egtr_relu(acc[4 * rq + 1] + bb.y);
      acc[4 * rq + 2] = egtr_relu(acc[4 * rq + 2] + bb.z);
      acc[4 * rq + 3] = egtr_relu(acc[4 * rq + 3] + bb.w);
      if (h2_save != nullptr && valid)
        *reinterpret_cast<float4*>(h2_save + ((size_t)mlp * total + (size_t)p) * kHd + n0) =
            make_float4(acc[4 * rq + 0], acc[4 * rq + 1], acc[4 * rq + 2], acc[4 * rq + 3]);
    }
    if (mlp == 0) {
      // ---- layer 3 (relation): rel^T[r_out][pair] += W3[r_out][n] h2^T[n][pair] ----------------------------
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) {
        const int ro = ot * 32 + pi;
        const bool rok = ro < R;
        const float* w3row = w3r + (size_t)(rok ? ro : 0) * kHd + nt * 32 + 4 * hf;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          float4 w = *reinterpret_cast<const float4*>(w3row + 8 * rq);
          if (!rok) w = make_float4(0.f, 0.f, 0.f, 0.f);
          racc[ot] = mfma32(w.x, acc[4 * rq + 0], racc[ot]);
          racc[ot] = mfma32(w.y, acc[4 * rq + 1], racc[ot]);
          racc[ot] = mfma32(w.z, acc[4 * rq + 2], racc[ot]);
          racc[ot] = mfma32(w.w, acc[4 * rq + 3], racc[ot]);
        }
      }
    } else {
      // ---- layer 3 (connectivity, one output): dot with w3c over this lane's 16 n values ---------------------
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const float4 w = *reinterpret_cast<const float4*>(w3c + nt * 32 + 8 * rq + 4 * hf);
        cacc += w.x * acc[4 * rq + 0] + w.y * acc[4 * rq + 1] + w.z * acc[4 * rq + 2] + w.w * acc[4 * rq + 3];
      }
    }
  }

  if (mlp == 1) {
    cacc += __shfl_xor(cacc, 32);
    if (hf == 0) s_cacc[wv][pi] = cacc;
    __syncthreads();
    if (wv == 0 && valid && hf == 0) {
      float v = s_cacc[0][pi];
#pragma unroll
      for (int w = 1; w < kRhWaves; ++w) v += s_cacc[w][pi];
      conn_logits[p] = v + b3c[0];
    }
    return;
  }
  // ---- relation epilogue: each wave stages its partial [pair][r_out] tile in LDS; the sum + b3 + frequency bias goes out
  // as one contiguous run of R floats per pair (4 R per subject) ---------------------------------------------------------
  constexpr int kStride = 32 * OT + 1;
  static_assert(kRhWaves * 32 * kStride <= kBuf, "output tiles fit in the transpose buffer");
  float* const my_out = s_out + wv * 32 * kStride;
#pragma unroll
  for (int ot = 0; ot < OT; ++ot)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ro = ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
      my_out[pi * kStride + ro] = racc[ot][r];
    }
  if (wv == 0 && hf == 0) {
    int tb = -1;
    if (triplet != nullptr) tb = ((int)node_cls[qi] * C1 + (int)node_cls[kj]) * R;
    s_tb[pi] = tb;
    s_pp[pi] = valid ? p : -1;
  }
  __syncthreads();
  for (int pp = wv; pp < 32; pp += kRhWaves) {
    const long long po = s_pp[pp];
    if (po < 0) continue;   // wave-uniform: a slot beyond the N x N grid
    const int tb = s_tb[pp];
    float* dst = rel_logits + (size_t)po * R;
    for (int r = lane; r < R; r += 64) {
      float v = s_out[pp * kStride + r];
#pragma unroll
      for (int w = 1; w < kRhWaves; ++w) v += s_out[(w * 32 + pp) * kStride + r];
      v += b3r[r];
      if (tb >= 0) v += triplet[tb + r];
      dst[r] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------ bf16 matrix cores
// Same kernel with layers 2 and 3 on v_mfma_f32_32x32x16_bf16 (the bf16 stress configuration: the weights W2 / W3 are
// the model's bf16 parameters, h1 / h2 are rounded to bf16 as operands, accumulation in fp32).  Layer 1 (gate, gated
// sum, ReLU) is unchanged fp32 VALU work.  Operand layout of the 32x32x16 MFMA: lane (i = l & 31, hf = l >> 5) holds
// A[i][8 hf + 0..7] and B[8 hf + 0..7][i]; D as for 32x32x2 (row = (r & 3) + 8 (r >> 2) + 4 hf, col = l & 31).
//   layer 2:  h2^T[n][pair] = sum_k W2[n][k] h1[pair][k]: step t covers k = 16 t .. 16 t + 15, so lane (pair, hf) keeps the
//             h1 channels {16 t + 8 hf + e} as 16 packed operands (64 VGPRs instead of 128 fp32 values);
//   layer 3:  the 16 accumulators of a 32-wide n tile are rows {0-3, 8-11, 16-19, 24-27} + 4 hf: registers 0..7 / 8..15
//             are the B operands of two K = 16 steps whose k slot e stands for row (e & 3) + 8 (e >> 2) + 4 hf (+ 16 for
//             the second step); the W3 operand is gathered in the same order (two 8-byte loads per step).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 pack_bf16x8(const float (&f)[8]) {
  bf16x8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (__bf16)f[i];
  return v;
}
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short u) { return __uint_as_float(((unsigned)u) << 16); }

template <int T, int OT>
__global__ __launch_bounds__(64) void rel_head_fwd_bf16w(
    const float* __restrict__ gate_q, const float* __restrict__ gate_k, const float* __restrict__ uq,
    const float* __restrict__ uk, const float* __restrict__ b1, const unsigned short* __restrict__ w2r,
    const float* __restrict__ b2r, const unsigned short* __restrict__ w3r, const float* __restrict__ b3r,
    const unsigned short* __restrict__ w2c, const float* __restrict__ b2c, const unsigned short* __restrict__ w3c,
    const float* __restrict__ b3c, const float* __restrict__ triplet, const int64_t* __restrict__ node_cls, int B,
    int N, int R, int C1, float* __restrict__ rel_logits, float* __restrict__ conn_logits,
    float* __restrict__ gate_mean) {
  constexpr int kBuf = 16 * kH1Stride > 32 * (32 * OT + 1) ? 16 * kH1Stride : 32 * (32 * OT + 1);
  __shared__ __attribute__((aligned(16))) float s_buf[kBuf];
  float* const s_out = s_buf;
  float* const s_h1 = s_buf;
  __shared__ int s_tb[32];
  const int lane = threadIdx.x, pi = lane & 31, hf = lane >> 5;
  const int mlp = blockIdx.y;
  const long long total = (long long)B * N * N;
  const long long p0 = (long long)blockIdx.x * 32;
  const long long p = p0 + pi;
  const bool valid = p < total;
  const long long pc = valid ? p : total - 1;
  const int b = (int)(pc / ((long long)N * N));
  const int rem = (int)(pc - (long long)b * N * N);
  const int i = rem / N, j = rem - i * N;
  const size_t qi = (size_t)b * N + i, kj = (size_t)b * N + j;

  float g[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const float x = gate_q[qi * T + t] + gate_k[kj * T + t];
    g[t] = 1.f / (1.f + expf(-x));
  }
  if (gate_mean != nullptr && mlp == 0) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      float v = (valid && hf == 0) ? g[t] : 0.f;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
      if (lane == 0) unsafeAtomicAdd(gate_mean + t, v / (float)total);
    }
  }

  // ---- layer 1 (fp32, as in rel_head_fwd_f32): pair by pair, lane = channel quad, transposed through LDS ------------
  bf16x8 h1b[16];
  {
    const int qrow_l = (int)qi, krow_l = (int)kj;
    const float4 bias4 = reinterpret_cast<const float4*>(b1 + mlp * kHd)[lane];
    const float4* uq4 = reinterpret_cast<const float4*>(uq) + mlp * (kHd / 4) + lane;
    const float4* uk4 = reinterpret_cast<const float4*>(uk) + mlp * (kHd / 4) + lane;
    constexpr int ROW4 = 2 * kHd / 4;
    float4 ua[T];
    f32x4v rc[2][T];
    int q_cur = -1;
    auto issue = [&](int set, int pp) {
      const int kr = __builtin_amdgcn_readlane(krow_l, pp);
      const float4* pk = uk4 + (size_t)kr * T * ROW4;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        if (set == 0) rc[0][t] = gload_b128<0>(pk + t * ROW4);
        else rc[1][t] = gload_b128<0>(pk + t * ROW4);
      }
    };
    auto consume = [&](int set, int pp, int row, bool more) {
      const int qr = __builtin_amdgcn_readlane(qrow_l, pp);
      if (qr != q_cur) {
        q_cur = qr;
        const float4* pq = uq4 + (size_t)qr * T * ROW4;
#pragma unroll
        for (int t = 0; t < T; ++t) ua[t] = pq[t * ROW4];
      }
      float4 acc = bias4;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        if (set == 0) {
          if (more) { vm_wait1<T + (T - 1)>(rc[0][t], t); } else { vm_wait1<T - 1>(rc[0][t], t); }
        } else {
          if (more) { vm_wait1<T + (T - 1)>(rc[1][t], t); } else { vm_wait1<T - 1>(rc[1][t], t); }
        }
        const float4 a = ua[t];
        const f32x4v c = set ? rc[1][t] : rc[0][t];
        const float gt = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g[t]), pp));
        acc.x += gt * (a.x + c.x);
        acc.y += gt * (a.y + c.y);
        acc.z += gt * (a.z + c.z);
        acc.w += gt * (a.w + c.w);
      }
      *reinterpret_cast<float4*>(&s_h1[row * kH1Stride + 4 * lane]) =
          make_float4(egtr_relu(acc.x), egtr_relu(acc.y), egtr_relu(acc.z), egtr_relu(acc.w));
    };
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
      const int pbase = half * 16;
      issue(0, pbase);
#pragma unroll 1
      for (int pp = 0; pp < 16; pp += 2) {
        issue(1, pbase + pp + 1);
        consume(0, pbase + pp, pp, true);
        if (pp + 2 < 16) issue(0, pbase + pp + 2);
        consume(1, pbase + pp + 1, pp + 1, pp + 2 < 16);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if ((pi >> 4) == half) {
        const float* hp = &s_h1[(pi & 15) * kH1Stride + 8 * hf];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const float4 v0 = *reinterpret_cast<const float4*>(hp + 16 * t);
          const float4 v1 = *reinterpret_cast<const float4*>(hp + 16 * t + 4);
          const float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
          h1b[t] = pack_bf16x8(f);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }

  const unsigned short* w2 = mlp ? w2c : w2r;
  const float* b2 = mlp ? b2c : b2r;
  f32x16 racc[OT];
#pragma unroll
  for (int ot = 0; ot < OT; ++ot)
#pragma unroll
    for (int r = 0; r < 16; ++r) racc[ot][r] = 0.f;
  float cacc = 0.f;

#pragma unroll 1
  for (int nt = 0; nt < kHd / 32; ++nt) {
    // ---- layer 2: 16 steps of K = 16; the lane's W2 row fragment for step t is 16 contiguous bytes ------------------
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float4* wrow = reinterpret_cast<const float4*>(w2 + (size_t)(nt * 32 + pi) * kHd) + hf;
    f32x4v wq[16];
    // all 16 operand loads in flight, consumed in order (inline asm keeps them in program order)
    wq[0] = gload_b128<0 * 32>(wrow);   wq[1] = gload_b128<1 * 32>(wrow);   wq[2] = gload_b128<2 * 32>(wrow);
    wq[3] = gload_b128<3 * 32>(wrow);   wq[4] = gload_b128<4 * 32>(wrow);   wq[5] = gload_b128<5 * 32>(wrow);
    wq[6] = gload_b128<6 * 32>(wrow);   wq[7] = gload_b128<7 * 32>(wrow);   wq[8] = gload_b128<8 * 32>(wrow);
    wq[9] = gload_b128<9 * 32>(wrow);   wq[10] = gload_b128<10 * 32>(wrow); wq[11] = gload_b128<11 * 32>(wrow);
    wq[12] = gload_b128<12 * 32>(wrow); wq[13] = gload_b128<13 * 32>(wrow); wq[14] = gload_b128<14 * 32>(wrow);
    wq[15] = gload_b128<15 * 32>(wrow);
#define EGTR_L2STEP(TT)                                                       \
    vm_wait<15 - TT>(wq[TT]);                                                  \
    acc = mfma_bf16(__builtin_bit_cast(bf16x8, wq[TT]), h1b[TT], acc);
    EGTR_L2STEP(0) EGTR_L2STEP(1) EGTR_L2STEP(2) EGTR_L2STEP(3) EGTR_L2STEP(4) EGTR_L2STEP(5) EGTR_L2STEP(6)
    EGTR_L2STEP(7) EGTR_L2STEP(8) EGTR_L2STEP(9) EGTR_L2STEP(10) EGTR_L2STEP(11) EGTR_L2STEP(12) EGTR_L2STEP(13)
    EGTR_L2STEP(14) EGTR_L2STEP(15)
#undef EGTR_L2STEP
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const int n0 = nt * 32 + 8 * rq + 4 * hf;
      const float4 bb = *reinterpret_cast<const float4*>(b2 + n0);
      acc[4 * rq + 0] = egtr_relu(acc[4 * rq + 0] + bb.x);
      acc[4 * rq + 1] = egtr_relu(acc[4 * rq + 1] + bb.y);
      acc[4 * rq + 2] = egtr_relu(acc[4 * rq + 2] + bb.z);
      acc[4 * rq + 3] = egtr_relu(acc[4 * rq + 3] + bb.w);
    }
    if (mlp == 0) {
      // ---- layer 3 (relation): two K = 16 steps per n tile; k slot e <-> n = nt*32 + 16 kb + (e&3) + 8 (e>>2) + 4 hf ----
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const float hv[8] = {acc[8 * kb + 0], acc[8 * kb + 1], acc[8 * kb + 2], acc[8 * kb + 3],
                             acc[8 * kb + 4], acc[8 * kb + 5], acc[8 * kb + 6], acc[8 * kb + 7]};
        const bf16x8 hb = pack_bf16x8(hv);
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) {
          const int ro = ot * 32 + pi;
          const bool rok = ro < R;
          const unsigned short* w3p = w3r + (size_t)(rok ? ro : 0) * kHd + nt * 32 + 16 * kb + 4 * hf;
          uint2 lo = *reinterpret_cast<const uint2*>(w3p), hi = *reinterpret_cast<const uint2*>(w3p + 8);
          if (!rok) { lo = make_uint2(0u, 0u); hi = make_uint2(0u, 0u); }
          const uint4 wv = make_uint4(lo.x, lo.y, hi.x, hi.y);
          racc[ot] = mfma_bf16(__builtin_bit_cast(bf16x8, wv), hb, racc[ot]);
        }
      }
    } else {
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const uint2 wv = *reinterpret_cast<const uint2*>(w3c + nt * 32 + 8 * rq + 4 * hf);
        cacc += bf16_bits_to_f32((unsigned short)(wv.x & 0xffffu)) * acc[4 * rq + 0] +
                bf16_bits_to_f32((unsigned short)(wv.x >> 16)) * acc[4 * rq + 1] +
                bf16_bits_to_f32((unsigned short)(wv.y & 0xffffu)) * acc[4 * rq + 2] +
                bf16_bits_to_f32((unsigned short)(wv.y >> 16)) * acc[4 * rq + 3];
      }
    }
  }

  if (mlp == 1) {
    cacc += __shfl_xor(cacc, 32);
    if (valid && hf == 0) conn_logits[p] = cacc + b3c[0];
    return;
  }
  constexpr int kStride = 32 * OT + 1;
#pragma unroll
  for (int ot = 0; ot < OT; ++ot)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ro = ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
      s_out[pi * kStride + ro] = racc[ot][r];
    }
  if (hf == 0) {
    int tb = -1;
    if (triplet != nullptr) tb = ((int)node_cls[qi] * C1 + (int)node_cls[kj]) * R;
    s_tb[pi] = tb;
  }
  __syncthreads();
  const int npair = (int)((total - p0) < 32 ? (total - p0) : 32);
  float* dst = rel_logits + (size_t)p0 * R;
  for (int pp = 0; pp < npair; ++pp) {
    const int tb = s_tb[pp];
    for (int r = lane; r < R; r += 64) {
      float v = s_out[pp * kStride + r] + b3r[r];
      if (tb >= 0) v += triplet[tb + r];
      dst[(size_t)pp * R + r] = v;
    }
  }
}

// ------------------------------------------------------------------------------ fp32 through split-bf16 operands
// On gfx950 the fp32 matrix rate (v_mfma_f32_32x32x2_f32: 64 FLOP/clk/SIMD) IS the fp32 vector rate; the bf16 matrix
// rate is 16x that.  This kernel is rel_head_fwd_f32 with layers 2 and 3 evaluated on v_mfma_f32_32x32x16_bf16 from
// THREE-way bf16 splits of both operands:  x = x_hi + x_mid + x_lo  EXACTLY (each piece a bf16 holding 8 of the 24
// mantissa bits of the fp32 value; the residuals x - x_hi and (x - x_hi) - x_mid are exact in fp32), and
//   w . h  ~=  w_hi h_hi + (w_hi h_mid + w_mid h_hi) + (w_hi h_lo + w_mid h_mid + w_lo h_hi)
// -- the six leading cross terms; the three dropped ones are <= 2^-24 |w||h|, the size of ONE fp32 rounding of the
// product.  Every bf16 x bf16 product is exact in fp32 and the accumulation is fp32 (the main term and the five
// correction terms in separate accumulators, added once per 32-wide tile), so the result carries fp32-level error
// (tests/test_gpu_kernels.py::test_relation_head_split_bf16_is_fp32_accurate measures both kernels against float64):
// 6 MFMAs of 32 cycles per K = 16 instead of 8 of 64 cycles -- 2.67x less matrix time at the same accuracy.
// Layer 1 (gates, gated sum, ReLU) and the connectivity output layer stay fp32 VALU work.  Inference only (no saved
// activations).  Operands:
//   h1 pieces: LDS, [3][32 pairs][264] bf16 (pitch 528 B: conflict-free ds_read_b128), written by layer 1;
//   W2 / W3 pieces: pre-split and pre-ordered on the host into the exact operand order (egtr_amd/ops.py), so that a
//   wave-level operand load is one contiguous KiB:
//     w2x[nt 8][t 16][piece 3][lane 64][8]      A[i = n (lane & 31)][k = 16 t + 8 (lane >> 5) + e]
//     w3x[nt 8][kb 2][ot OT][piece 3][lane 64][8]  k slot e <-> n = 32 nt + 16 kb + (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
constexpr int kHp = 264;  // bf16 elements per LDS row of an h1 piece

// x = hi + mid + lo EXACTLY, by truncation: hi = the top 8 mantissa bits of x (its upper 16 bits as they are), the
// residual x - hi is exact in fp32 and holds the remaining <= 16 bits, mid = its top 8, and what is left has <= 8
// significant bits, i.e. already is a bf16.  Pieces are kept as fp32 bit patterns whose low 16 bits are zero; two of
// them are packed into one dword of bf16 pairs with a single v_perm_b32.
// (xs_format.h: non-finite x keeps the inf / a quiet NaN in hi alone, mid = lo = 0 -- `x - hi` would be inf - inf)
typedef float f32x2v __attribute__((ext_vector_type(2)));
using Split3 = xs::Split3;
__device__ __forceinline__ Split3 split3(float x) { return xs::split3(x); }
// {bf16(a) in the low half, bf16(b) in the high half} from two fp32 bit patterns with zero low halves
__device__ __forceinline__ unsigned pack_hi16(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

__device__ __forceinline__ f32x4v gload_b128_s(unsigned voff, const void* sbase) {
  f32x4v v;
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(v) : "v"(voff), "s"(sbase));
  return v;
}
__device__ __forceinline__ f32x4v gload_b128_s1k(unsigned voff, const void* sbase) {
  f32x4v v;
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=&v"(v) : "v"(voff), "s"(sbase));
  return v;
}
__device__ __forceinline__ f32x4v gload_b128_s2k(unsigned voff, const void* sbase) {
  f32x4v v;
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=&v"(v) : "v"(voff), "s"(sbase));
  return v;
}

// layer-2 K loop of one 32-wide n tile: 16 steps, the three W2 pieces of a step are three contiguous KiB; PRE steps of
// loads in flight (counted vmcnt, as in PrefetchRow)
template <int PRE>
struct PrefetchX6 {
  template <int I>
  static __device__ __forceinline__ void issue(f32x4v (&wq)[PRE][3], unsigned voff, const char* wtile) {
    wq[I][0] = gload_b128_s(voff, wtile + I * 3072);
    wq[I][1] = gload_b128_s1k(voff, wtile + I * 3072);
    wq[I][2] = gload_b128_s2k(voff, wtile + I * 3072);
    if constexpr (I + 1 < PRE) issue<I + 1>(wq, voff, wtile);
  }
  template <int S>
  static __device__ __forceinline__ void run(f32x4v (&wq)[PRE][3], unsigned voff, const char* wtile,
                                             const __bf16* hrow, f32x16& acc_m, f32x16& acc_c) {
    constexpr int newer = 3 * ((15 - S) < (PRE - 1) ? (15 - S) : (PRE - 1));  // own loads issued after this step's
    vm_wait<newer>(wq[S % PRE][0]);
    vm_wait<newer>(wq[S % PRE][1]);
    vm_wait<newer>(wq[S % PRE][2]);
    const bf16x8 whi = __builtin_bit_cast(bf16x8, wq[S % PRE][0]);
    const bf16x8 wmid = __builtin_bit_cast(bf16x8, wq[S % PRE][1]);
    const bf16x8 wlo = __builtin_bit_cast(bf16x8, wq[S % PRE][2]);
    const bf16x8 hhi = *reinterpret_cast<const bf16x8*>(hrow + 16 * S);
    const bf16x8 hmid = *reinterpret_cast<const bf16x8*>(hrow + 32 * kHp + 16 * S);
    const bf16x8 hlo = *reinterpret_cast<const bf16x8*>(hrow + 64 * kHp + 16 * S);
    acc_c = mfma_bf16(whi, hlo, acc_c);
    acc_m = mfma_bf16(whi, hhi, acc_m);
    acc_c = mfma_bf16(wlo, hhi, acc_c);
    acc_c = mfma_bf16(wmid, hmid, acc_c);
    acc_c = mfma_bf16(whi, hmid, acc_c);
    acc_c = mfma_bf16(wmid, hhi, acc_c);
    if constexpr (S + PRE < 16) {
      wq[S % PRE][0] = gload_b128_s(voff, wtile + (S + PRE) * 3072);
      wq[S % PRE][1] = gload_b128_s1k(voff, wtile + (S + PRE) * 3072);
      wq[S % PRE][2] = gload_b128_s2k(voff, wtile + (S + PRE) * 3072);
    }
    if constexpr (S + 1 < 16) run<S + 1>(wq, voff, wtile, hrow, acc_m, acc_c);
  }
};

// SAVE (training forward): the post-ReLU activations of layers 1 and 2 go to h1_save / h2_save [2 (mlp)][B N N][256] as in
// rel_head_fwd_f32 -- the backward's GEMMs read them (ops.RelationHeadFunction).
template <int T, int OT, bool SAVE = false>
__global__ __launch_bounds__(64 * kRhWaves) __attribute__((amdgpu_waves_per_eu(2, 2))) void rel_head_fwd_x6(
    const float* __restrict__ gate_q, const float* __restrict__ gate_k, const float* __restrict__ uq,
    const float* __restrict__ uk, const float* __restrict__ b1, const unsigned short* __restrict__ w2xr,
    const float* __restrict__ b2r, const unsigned short* __restrict__ w3xr, const float* __restrict__ b3r,
    const unsigned short* __restrict__ w2xc, const float* __restrict__ b2c, const float* __restrict__ w3c,
    const float* __restrict__ b3c, const float* __restrict__ triplet, const int64_t* __restrict__ node_cls, int B,
    int N, int R, int C1, float* __restrict__ rel_logits, float* __restrict__ conn_logits,
    float* __restrict__ gate_mean, int apply_sigmoid, float* __restrict__ h1_save, float* __restrict__ h2_save) {
  // apply_sigmoid: write sigmoid(logit) (egtr.py:450-454, the model's pred_rel / pred_connectivity) instead of the logit
  // one buffer, two lives: the three h1 pieces (read by every layer-2 step), then the output tiles
  constexpr int kStride = 32 * OT + 1;
  static_assert(kRhWaves * 32 * kStride * 4 <= 3 * 32 * kHp * 2, "output tiles fit in the h1 buffer");
  __shared__ __attribute__((aligned(16))) __bf16 s_h[3 * 32 * kHp];
  float* const s_out = reinterpret_cast<float*>(s_h);
  __shared__ int s_tb[32];
  __shared__ long long s_pp[32];
  __shared__ float s_cacc[kRhWaves][32];
  // SAVE: a wave's 32 x 32 hidden-2 tile goes through this buffer so that it leaves as whole 128-byte row segments (written
  // straight from the accumulator layout, every store instruction touched 32 lines with 32 bytes each)
  // (eight pairs at a time: with a whole tile staged the kernel's LDS would allow two workgroups per CU instead of three, and it
  // is latency-bound -- 750 us against 4 x 98 us for the same work at inference)
  constexpr int kH2Pitch = 36;
  __shared__ __attribute__((aligned(16))) float s_h2[SAVE ? kRhWaves * 8 * kH2Pitch : 4];
  const int lane = threadIdx.x & 63, pi = lane & 31, hf = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int mlp = blockIdx.y;
  const long long total = (long long)B * N * N;
  const int tj = (N + 3) >> 2, ti = (N + 7) >> 3;
  const int b = blockIdx.x / (ti * tj);
  const int trem = blockIdx.x - b * ti * tj;
  const int i0 = (trem / tj) * 8, j0 = (trem - (trem / tj) * tj) * 4;
  const int i_raw = i0 + (pi >> 2), j_raw = j0 + (pi & 3);
  const bool valid = i_raw < N && j_raw < N;
  const int i = i_raw < N ? i_raw : N - 1, j = j_raw < N ? j_raw : N - 1;
  const long long p = ((long long)b * N + i) * N + j;
  const size_t qi = (size_t)b * N + i, kj = (size_t)b * N + j;

  if (SAVE && wv == 0 && hf == 0) s_pp[pi] = valid ? p : -1;   // visible behind the barrier that closes layer 1
  float g[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const float x = gate_q[qi * T + t] + gate_k[kj * T + t];
    g[t] = 1.f / (1.f + expf(-x));
  }
  if (gate_mean != nullptr && mlp == 0 && wv == 0) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      float v = (valid && hf == 0) ? g[t] : 0.f;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
      if (lane == 0) unsafeAtomicAdd(gate_mean + t, v / (float)total);
    }
  }

  // ---- layer 1: as rel_head_fwd_f32 (fp32); the ReLU output is split into its three bf16 pieces on the way to LDS ------
  {
    const int krow_l = (int)kj;
    const float4 bias4 = reinterpret_cast<const float4*>(b1 + mlp * kHd)[lane];
    const float4* uq4 = reinterpret_cast<const float4*>(uq) + mlp * (kHd / 4) + lane;
    const float4* uk4 = reinterpret_cast<const float4*>(uk) + mlp * (kHd / 4) + lane;
    constexpr int ROW4 = 2 * kHd / 4;
    float4 kk[4][T];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int kr = __builtin_amdgcn_readlane(krow_l, jj);
      const float4* pk = uk4 + (size_t)kr * T * ROW4;
#pragma unroll
      for (int t = 0; t < T; ++t) kk[jj][t] = pk[t * ROW4];
    }
    auto qrow_of = [&](int ii) { return b * N + (i0 + ii < N ? i0 + ii : N - 1); };
    const int ii0 = wv * (8 / kRhWaves);
    float4 ua[2][T];
    {
      const float4* pq = uq4 + (size_t)qrow_of(ii0) * T * ROW4;
#pragma unroll
      for (int t = 0; t < T; ++t) ua[0][t] = pq[t * ROW4];
    }
#pragma unroll
    for (int iu = 0; iu < 8 / kRhWaves; ++iu) {
      const int ii = ii0 + iu;
      if (iu + 1 < 8 / kRhWaves) {
        const float4* pq = uq4 + (size_t)qrow_of(ii + 1) * T * ROW4;
#pragma unroll
        for (int t = 0; t < T; ++t) ua[(iu + 1) & 1][t] = pq[t * ROW4];
      }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int pp = ii * 4 + jj;
        // two-wide (v_pk_add_f32 / v_pk_fma_f32) and the split without its non-finite special cases: the kernel issues ~9 VALU
        // instructions per MFMA (profiles/r03_x6_mfma_pmc.txt), this loop and the two splits are most of them
        f32x2v a01 = {bias4.x, bias4.y}, a23 = {bias4.z, bias4.w};
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const float4 a = ua[iu & 1][t];
          const float4 c = kk[jj][t];
          const float gt = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g[t]), pp));
          const f32x2v gg = {gt, gt};
          const f32x2v u01 = {a.x + c.x, a.y + c.y}, u23 = {a.z + c.z, a.w + c.w};
          a01 = gg * u01 + a01;
          a23 = gg * u23 + a23;
        }
        if (SAVE && i0 + ii < N && j0 + jj < N)   // one contiguous 1 KiB row per wave
          reinterpret_cast<float4*>(h1_save + ((size_t)mlp * total +
                                               (size_t)(((long long)b * N + i0 + ii) * N + j0 + jj)) * kHd)[lane] =
              make_float4(egtr_relu(a01.x), egtr_relu(a01.y), egtr_relu(a23.x), egtr_relu(a23.y));
        const Split3 sx = xs::split3_fast(egtr_relu(a01.x)), sy = xs::split3_fast(egtr_relu(a01.y)),
                     sz = xs::split3_fast(egtr_relu(a23.x)), sw = xs::split3_fast(egtr_relu(a23.y));
        __bf16* hp = s_h + pp * kHp + 4 * lane;
        *reinterpret_cast<uint2*>(hp) = make_uint2(pack_hi16(sx.hi, sy.hi), pack_hi16(sz.hi, sw.hi));
        *reinterpret_cast<uint2*>(hp + 32 * kHp) = make_uint2(pack_hi16(sx.mid, sy.mid), pack_hi16(sz.mid, sw.mid));
        *reinterpret_cast<uint2*>(hp + 64 * kHp) = make_uint2(pack_hi16(sx.lo, sy.lo), pack_hi16(sz.lo, sw.lo));
      }
    }
    __syncthreads();   // both waves' rows are in LDS
  }

  const char* w2x = reinterpret_cast<const char*>(mlp ? w2xc : w2xr);
  const float* b2 = mlp ? b2c : b2r;
  f32x16 racc[OT];
#pragma unroll
  for (int ot = 0; ot < OT; ++ot)
#pragma unroll
    for (int r = 0; r < 16; ++r) racc[ot][r] = 0.f;
  float cacc = 0.f;
  const unsigned voff = (unsigned)lane * 16u;
  const __bf16* hrow = s_h + pi * kHp + 8 * hf;

  constexpr int kNtPerWave = kHd / 32 / kRhWaves;
#pragma unroll 1
  for (int nt = wv * kNtPerWave; nt < (wv + 1) * kNtPerWave; ++nt) {
    // ---- layer 2: h2^T[n][pair], n = 32 nt + (r&3) + 8 (r>>2) + 4 hf ---------------------------------------
    f32x16 acc, accc;
#pragma unroll
    for (int r = 0; r < 16; ++r) accc[r] = 0.f;
    // the main accumulator starts from the layer-2 bias (requested before the weight loads, not behind the loop): register
    // 4 rq + j <-> hidden-2 unit 32 nt + 8 rq + 4 hf + j
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const float4 bb = *reinterpret_cast<const float4*>(b2 + nt * 32 + 8 * rq + 4 * hf);
      acc[4 * rq + 0] = bb.x; acc[4 * rq + 1] = bb.y; acc[4 * rq + 2] = bb.z; acc[4 * rq + 3] = bb.w;
    }
    const char* wtile = w2x + (size_t)nt * (16 * 3072);
    constexpr int kPre = 4;
    f32x4v wq[kPre][3];
    // relation MLP: the 12 W3 operand fragments of this n tile are requested BEFORE the layer-2 loop (asm loads, oldest in the
    // in-order queue: the loop's own counted waits cover them).  Loaded at their use they were twelve exposed L2 round trips
    // per n tile -- about as long as the tile's MFMAs.
    f32x4v w3q[2 * OT][3];
    if (mlp == 0) {
      const char* w3t = reinterpret_cast<const char*>(w3xr) + (size_t)nt * (2 * OT * 3072);
#pragma unroll
      for (int f = 0; f < 2 * OT; ++f) {
        w3q[f][0] = gload_b128_s(voff, w3t + f * 3072);
        w3q[f][1] = gload_b128_s1k(voff, w3t + f * 3072);
        w3q[f][2] = gload_b128_s2k(voff, w3t + f * 3072);
      }
    }
    PrefetchX6<kPre>::template issue<0>(wq, voff, wtile);
    PrefetchX6<kPre>::template run<0>(wq, voff, wtile, hrow, acc, accc);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = egtr_relu(acc[r] + accc[r]);
    if (SAVE) {
      float* stg = s_h2 + wv * (8 * kH2Pitch);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if ((pi >> 3) == i) {
#pragma unroll
          for (int rq = 0; rq < 4; ++rq)
            *reinterpret_cast<float4*>(stg + (pi & 7) * kH2Pitch + 8 * rq + 4 * hf) =
                make_float4(acc[4 * rq + 0], acc[4 * rq + 1], acc[4 * rq + 2], acc[4 * rq + 3]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // a wave's LDS operations execute in order: the fences only
        __builtin_amdgcn_wave_barrier();                          // keep the compiler from moving them
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int pr = lane >> 3, ch = lane & 7;                  // 8 lanes = the 128 bytes of one pair's 32 units
        const float4 v = *reinterpret_cast<const float4*>(stg + pr * kH2Pitch + 4 * ch);
        const long long po = s_pp[8 * i + pr];
        if (po >= 0) *reinterpret_cast<float4*>(h2_save + ((size_t)mlp * total + (size_t)po) * kHd + nt * 32 + 4 * ch) = v;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
    if (mlp == 0) {
      // ---- layer 3 (relation): two K = 16 steps per n tile on the split accumulators; W3 pieces pre-ordered ----------
#pragma unroll
      for (int f = 0; f < 2 * OT; ++f) {   // the layer-2 loop ended on vmcnt(0): everything has landed
        asm volatile("" : "+v"(w3q[f][0]), "+v"(w3q[f][1]), "+v"(w3q[f][2]));
      }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        uint4 phi, pmid, plo;
        {
          Split3 sp[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) sp[e] = xs::split3_fast(acc[8 * kb + e]);
          phi = make_uint4(pack_hi16(sp[0].hi, sp[1].hi), pack_hi16(sp[2].hi, sp[3].hi), pack_hi16(sp[4].hi, sp[5].hi),
                           pack_hi16(sp[6].hi, sp[7].hi));
          pmid = make_uint4(pack_hi16(sp[0].mid, sp[1].mid), pack_hi16(sp[2].mid, sp[3].mid),
                            pack_hi16(sp[4].mid, sp[5].mid), pack_hi16(sp[6].mid, sp[7].mid));
          plo = make_uint4(pack_hi16(sp[0].lo, sp[1].lo), pack_hi16(sp[2].lo, sp[3].lo), pack_hi16(sp[4].lo, sp[5].lo),
                           pack_hi16(sp[6].lo, sp[7].lo));
        }
        const bf16x8 hhi = __builtin_bit_cast(bf16x8, phi), hmid = __builtin_bit_cast(bf16x8, pmid),
                     hlo = __builtin_bit_cast(bf16x8, plo);
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) {
          const bf16x8 whi = __builtin_bit_cast(bf16x8, w3q[kb * OT + ot][0]);
          const bf16x8 wmid = __builtin_bit_cast(bf16x8, w3q[kb * OT + ot][1]);
          const bf16x8 wlo = __builtin_bit_cast(bf16x8, w3q[kb * OT + ot][2]);
          racc[ot] = mfma_bf16(whi, hlo, racc[ot]);
          racc[ot] = mfma_bf16(wlo, hhi, racc[ot]);
          racc[ot] = mfma_bf16(wmid, hmid, racc[ot]);
          racc[ot] = mfma_bf16(whi, hmid, racc[ot]);
          racc[ot] = mfma_bf16(wmid, hhi, racc[ot]);
          racc[ot] = mfma_bf16(whi, hhi, racc[ot]);
        }
      }
    } else {
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const float4 w = *reinterpret_cast<const float4*>(w3c + nt * 32 + 8 * rq + 4 * hf);
        cacc += w.x * acc[4 * rq + 0] + w.y * acc[4 * rq + 1] + w.z * acc[4 * rq + 2] + w.w * acc[4 * rq + 3];
      }
    }
  }

  if (mlp == 1) {
    cacc += __shfl_xor(cacc, 32);
    if (hf == 0) s_cacc[wv][pi] = cacc;
    __syncthreads();
    if (wv == 0 && valid && hf == 0) {
      float v = s_cacc[0][pi];
#pragma unroll
      for (int w = 1; w < kRhWaves; ++w) v += s_cacc[w][pi];
      v += b3c[0];
      conn_logits[p] = apply_sigmoid ? 1.f / (1.f + expf(-v)) : v;
    }
    return;
  }
  __syncthreads();   // every wave has finished reading the h1 pieces: the buffer becomes the output staging area
  float* const my_out = s_out + wv * 32 * kStride;
#pragma unroll
  for (int ot = 0; ot < OT; ++ot)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ro = ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
      my_out[pi * kStride + ro] = racc[ot][r];
    }
  if (wv == 0 && hf == 0) {
    int tb = -1;
    if (triplet != nullptr) tb = ((int)node_cls[qi] * C1 + (int)node_cls[kj]) * R;
    s_tb[pi] = tb;
    s_pp[pi] = valid ? p : -1;
  }
  __syncthreads();
  for (int pp = wv; pp < 32; pp += kRhWaves) {
    const long long po = s_pp[pp];
    if (po < 0) continue;
    const int tb = s_tb[pp];
    float* dst = rel_logits + (size_t)po * R;
    for (int r = lane; r < R; r += 64) {
      float v = s_out[pp * kStride + r];
#pragma unroll
      for (int w = 1; w < kRhWaves; ++w) v += s_out[(w * 32 + pp) * kStride + r];
      v += b3r[r];
      if (tb >= 0) v += triplet[tb + r];
      dst[r] = apply_sigmoid ? 1.f / (1.f + expf(-v)) : v;
    }
  }
}

template <int T>
int launch_T(hipStream_t st, int R, dim3 grid, const float* gate_q, const float* gate_k, const float* uq,
             const float* uk, const float* b1, const float* w2r, const float* b2r, const float* w3r,
             const float* b3r, const float* w2c, const float* b2c, const float* w3c, const float* b3c,
             const float* triplet, const int64_t* node_cls, int B, int N, int C1, float* rel, float* conn,
             float* gate_mean, float* h1_save, float* h2_save) {
  if (R <= 32)
    hipLaunchKernelGGL((rel_head_fwd_f32<T, 1>), grid, dim3(64 * kRhWaves), 0, st, gate_q, gate_k, uq, uk, b1, w2r, b2r, w3r,
                       b3r, w2c, b2c, w3c, b3c, triplet, node_cls, B, N, R, C1, rel, conn, gate_mean, h1_save,
                       h2_save);
  else
    hipLaunchKernelGGL((rel_head_fwd_f32<T, 2>), grid, dim3(64 * kRhWaves), 0, st, gate_q, gate_k, uq, uk, b1, w2r, b2r, w3r,
                       b3r, w2c, b2c, w3c, b3c, triplet, node_cls, B, N, R, C1, rel, conn, gate_mean, h1_save,
                       h2_save);
  return 0;
}

// ------------------------------------------------------------------------------------------------------ backward
// The MLP part of the backward is plain GEMMs on [B*N*N, 256] matrices (rocBLAS, driven from egtr_amd/ops.py) using
// the activations saved by the forward.  What is NOT a GEMM is the pairwise part below: from dh1 = dL/d(pre-ReLU
// layer-1 output) [2][B*N*N][256] it produces the gradients of the per-query tables,
//   duq[i,t,:] = sum_j g[i,j,t] dh1[i,j,:]          duk[j,t,:] = sum_i g[i,j,t] dh1[i,j,:]
//   dg[i,j,t]  = dh1[i,j,:] . (uq[i,t,:] + uk[j,t,:])       dz = dg g (1 - g)
//   dgate_q[i,t] = sum_j dz[i,j,t]                  dgate_k[j,t] = sum_i dz[i,j,t]
// in two passes over dh1 (one contiguous 2 x 1 KiB row per wave per pair, each read exactly once per pass):
//   pass Q: workgroup = (b, i), waves split j: duq, and the uq half of dg -> dz buffer
//   pass K: workgroup = (b, j), waves split i: duk, the uk half of dg, dz = (.) g (1 - g) -> dz buffer, dgate_k
//   pass A: dgate_q[i,t] = sum_j dz[i,j,t]
// Lane l of a wave owns channels {4 l .. 4 l + 3} of the relation half and of the connectivity half (8 channels).
template <int T, bool KPASS>
__global__ __launch_bounds__(256) void rel_head_bwd_pairs_f32(
    const float* __restrict__ dh1, const float* __restrict__ gate_q, const float* __restrict__ gate_k,
    const float* __restrict__ utab /* uq (Q pass) or uk (K pass): [B,N,T,512] */, int B, int N,
    float* __restrict__ dutab /* duq or duk [B,N,T,512] */, float* __restrict__ dz /* [B,N,N,T] */,
    float* __restrict__ dgate_k /* K pass: [B,N,T] */) {
  __shared__ float s_g[256 * T];                                  // gate of (fixed, m) for m < N in chunks of 256
  __shared__ __attribute__((aligned(16))) float s_red[4 * T * 512];  // per-wave partial dutab, reduced at the end
  __shared__ float s_dk[4 * T];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / N, f = blockIdx.x - b * N;  // f = the fixed index (i in the Q pass, j in the K pass)
  const size_t total = (size_t)B * N * N;
  const float* gfix = (KPASS ? gate_k : gate_q) + ((size_t)b * N + f) * T;
  const float* gvar = (KPASS ? gate_q : gate_k) + (size_t)b * N * T;
  float gf[T];
#pragma unroll
  for (int t = 0; t < T; ++t) gf[t] = gfix[t];
  // this lane's 8 channels of the fixed row's table, all T slots
  float4 ur[T], uc[T];
  {
    const float4* up = reinterpret_cast<const float4*>(utab + ((size_t)b * N + f) * T * 512);
#pragma unroll
    for (int t = 0; t < T; ++t) {
      ur[t] = up[t * 128 + lane];
      uc[t] = up[t * 128 + 64 + lane];
    }
  }
  float4 ar[T], ac[T];
  float dk[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    ar[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    ac[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    dk[t] = 0.f;
  }
  for (int m0 = 0; m0 < N; m0 += 256) {
    const int mc = min(256, N - m0);
    __syncthreads();
    if (tid < mc) {
#pragma unroll
      for (int t = 0; t < T; ++t) s_g[tid * T + t] = 1.f / (1.f + expf(-(gf[t] + gvar[(size_t)(m0 + tid) * T + t])));
    }
    __syncthreads();
    for (int mm = wave; mm < mc; mm += 4) {
      const int m = m0 + mm;
      const size_t pair = KPASS ? ((size_t)b * N + m) * N + f : ((size_t)b * N + f) * N + m;
      const float4 dr = reinterpret_cast<const float4*>(dh1 + pair * 256)[lane];
      const float4 dc = reinterpret_cast<const float4*>(dh1 + (total + pair) * 256)[lane];
      float dot[T];
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const float g = s_g[mm * T + t];
        ar[t].x += g * dr.x; ar[t].y += g * dr.y; ar[t].z += g * dr.z; ar[t].w += g * dr.w;
        ac[t].x += g * dc.x; ac[t].y += g * dc.y; ac[t].z += g * dc.z; ac[t].w += g * dc.w;
        dot[t] = dr.x * ur[t].x + dr.y * ur[t].y + dr.z * ur[t].z + dr.w * ur[t].w + dc.x * uc[t].x +
                 dc.y * uc[t].y + dc.z * uc[t].z + dc.w * uc[t].w;
      }
      // reduce the T dot products over the 64 lanes
#pragma unroll
      for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) dot[t] += __shfl_xor(dot[t], o);
      }
      if (lane < T) {
        float v = 0.f, g = 0.f;
#pragma unroll
        for (int t = 0; t < T; ++t) {
          v = (lane == t) ? dot[t] : v;
          g = (lane == t) ? s_g[mm * T + t] : g;
        }
        if (KPASS) {
          const float z = (dz[pair * T + lane] + v) * g * (1.f - g);
          dz[pair * T + lane] = z;
          dk[0] += z;  // lane t accumulates dgate_k[f, t]
        } else {
          dz[pair * T + lane] = v;
        }
      }
    }
  }
  // cross-wave reduction of the table gradient
  {
    float4* mine = reinterpret_cast<float4*>(s_red + wave * T * 512);
#pragma unroll
    for (int t = 0; t < T; ++t) {
      mine[t * 128 + lane] = ar[t];
      mine[t * 128 + 64 + lane] = ac[t];
    }
    if (KPASS && lane < T) s_dk[wave * T + lane] = dk[0];
    __syncthreads();
    float4* dst = reinterpret_cast<float4*>(dutab + ((size_t)b * N + f) * T * 512);
    const float4* r0 = reinterpret_cast<const float4*>(s_red);
    for (int e = tid; e < T * 128; e += 256) {
      const float4 a = r0[e], c = r0[T * 128 + e], d = r0[2 * T * 128 + e], g = r0[3 * T * 128 + e];
      dst[e] = make_float4(a.x + c.x + d.x + g.x, a.y + c.y + d.y + g.y, a.z + c.z + d.z + g.z, a.w + c.w + d.w + g.w);
    }
    if (KPASS && tid < T)
      dgate_k[((size_t)b * N + f) * T + tid] = s_dk[tid] + s_dk[T + tid] + s_dk[2 * T + tid] + s_dk[3 * T + tid];
  }
}

// dgate_q[b,i,t] = sum_j dz[b,i,j,t]: one wave per (b, i)
template <int T>
__global__ __launch_bounds__(64) void rel_head_bwd_gate_q_f32(const float* __restrict__ dz, int N,
                                                              float* __restrict__ dgate_q) {
  const size_t row = blockIdx.x;  // b * N + i
  const float* src = dz + row * N * T;
  float acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t) acc[t] = 0.f;
  for (int j = threadIdx.x; j < N; j += 64)
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] += src[(size_t)j * T + t];
#pragma unroll
  for (int t = 0; t < T; ++t) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc[t] += __shfl_xor(acc[t], o);
    if (threadIdx.x == 0) dgate_q[row * T + t] = acc[t];
  }
}

template <int T>
int launch_bwd_T(hipStream_t st, const float* dh1, const float* gate_q, const float* gate_k, const float* uq,
                 const float* uk, int B, int N, float* duq, float* duk, float* dgq, float* dgk, float* dz) {
  hipLaunchKernelGGL((rel_head_bwd_pairs_f32<T, false>), dim3(B * N), dim3(256), 0, st, dh1, gate_q, gate_k, uq, B, N,
                     duq, dz, (float*)nullptr);
  hipLaunchKernelGGL((rel_head_bwd_pairs_f32<T, true>), dim3(B * N), dim3(256), 0, st, dh1, gate_q, gate_k, uk, B, N,
                     duk, dz, dgk);
  hipLaunchKernelGGL((rel_head_bwd_gate_q_f32<T>), dim3(B * N), dim3(64), 0, st, dz, N, dgq);
  return 0;
}

}  // namespace

extern "C" int egtr_rel_head_forward_save_f32(egtr_stream_t stream, const float* gate_q, const float* gate_k,
                                              const float* uq, const float* uk, const float* b1, const float* w2r,
                                              const float* b2r, const float* w3r, const float* b3r, const float* w2c,
                                              const float* b2c, const float* w3c, const float* b3c,
                                              const float* triplet_dist, const int64_t* node_cls, int batch,
                                              int num_query, int num_slots, int hidden, int num_rel,
                                              int num_cls_plus1, float* rel_logits, float* conn_logits,
                                              float* gate_mean, float* h1_save, float* h2_save) {
  if (!gate_q || !gate_k || !uq || !uk || !b1 || !w2r || !b2r || !w3r || !b3r || !w2c || !b2c || !w3c || !b3c ||
      !rel_logits || !conn_logits)
    return EGTR_E_ARG;
  if (triplet_dist != nullptr && node_cls == nullptr) return EGTR_E_ARG;
  if (batch <= 0 || num_query <= 0 || num_slots <= 0 || num_rel <= 0) return EGTR_E_ARG;
  if (hidden != kHd || num_rel > 64 || num_slots > 10) return EGTR_E_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // one wave per (image, 8 x 4 pair tile, MLP)
  const long long tiles = (long long)batch * ((num_query + 7) / 8) * ((num_query + 3) / 4);
  if (tiles >= (1ll << 31)) return EGTR_E_UNSUPPORTED;
  const dim3 grid((unsigned)tiles, 2);
#define EGTR_T(TT)                                                                                                  \
  case TT:                                                                                                          \
    launch_T<TT>(st, num_rel, grid, gate_q, gate_k, uq, uk, b1, w2r, b2r, w3r, b3r, w2c, b2c, w3c, b3c,             \
                 triplet_dist, node_cls, batch, num_query, num_cls_plus1, rel_logits, conn_logits, gate_mean,       \
                 h1_save, h2_save);                                                                                 \
    break;
  switch (num_slots) {
    EGTR_T(1) EGTR_T(2) EGTR_T(3) EGTR_T(4) EGTR_T(5) EGTR_T(6) EGTR_T(7) EGTR_T(8) EGTR_T(9) EGTR_T(10)
    default: return EGTR_E_UNSUPPORTED;
  }
#undef EGTR_T
  return egtr_check_launch();
}

extern "C" int egtr_rel_head_backward_pairs_f32(egtr_stream_t stream, const float* dh1, const float* gate_q,
                                                const float* gate_k, const float* uq, const float* uk, int batch,
                                                int num_query, int num_slots, int hidden, float* grad_uq,
                                                float* grad_uk, float* grad_gate_q, float* grad_gate_k,
                                                float* dz_workspace) {
  if (!dh1 || !gate_q || !gate_k || !uq || !uk || !grad_uq || !grad_uk || !grad_gate_q || !grad_gate_k ||
      !dz_workspace)
    return EGTR_E_ARG;
  if (batch <= 0 || num_query <= 0 || num_slots <= 0) return EGTR_E_ARG;
  if (hidden != kHd || num_slots > 10) return EGTR_E_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
#define EGTR_T(TT)                                                                                                  \
  case TT:                                                                                                          \
    launch_bwd_T<TT>(st, dh1, gate_q, gate_k, uq, uk, batch, num_query, grad_uq, grad_uk, grad_gate_q, grad_gate_k, \
                     dz_workspace);                                                                                 \
    break;
  switch (num_slots) {
    EGTR_T(1) EGTR_T(2) EGTR_T(3) EGTR_T(4) EGTR_T(5) EGTR_T(6) EGTR_T(7) EGTR_T(8) EGTR_T(9) EGTR_T(10)
    default: return EGTR_E_UNSUPPORTED;
  }
#undef EGTR_T
  return egtr_check_launch();
}

extern "C" int egtr_rel_head_forward_bf16w(egtr_stream_t stream, const float* gate_q, const float* gate_k,
                                           const float* uq, const float* uk, const float* b1, const uint16_t* w2r,
                                           const float* b2r, const uint16_t* w3r, const float* b3r,
                                           const uint16_t* w2c, const float* b2c, const uint16_t* w3c,
                                           const float* b3c, const float* triplet_dist, const int64_t* node_cls,
                                           int batch, int num_query, int num_slots, int hidden, int num_rel,
                                           int num_cls_plus1, float* rel_logits, float* conn_logits,
                                           float* gate_mean) {
  if (!gate_q || !gate_k || !uq || !uk || !b1 || !w2r || !b2r || !w3r || !b3r || !w2c || !b2c || !w3c || !b3c ||
      !rel_logits || !conn_logits)
    return EGTR_E_ARG;
  if (triplet_dist != nullptr && node_cls == nullptr) return EGTR_E_ARG;
  if (batch <= 0 || num_query <= 0 || num_slots <= 0 || num_rel <= 0) return EGTR_E_ARG;
  if (hidden != kHd || num_rel > 64 || num_slots > 10) return EGTR_E_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long total = (long long)batch * num_query * num_query;
  const dim3 grid((unsigned)((total + 31) / 32), 2);
#define EGTR_TB(TT)                                                                                                  \
  case TT:                                                                                                           \
    if (num_rel <= 32)                                                                                               \
      hipLaunchKernelGGL((rel_head_fwd_bf16w<TT, 1>), grid, dim3(64), 0, st, gate_q, gate_k, uq, uk, b1, w2r, b2r, w3r, \
                         b3r, w2c, b2c, w3c, b3c, triplet_dist, node_cls, batch, num_query, num_rel, num_cls_plus1,  \
                         rel_logits, conn_logits, gate_mean);                                                        \
    else                                                                                                             \
      hipLaunchKernelGGL((rel_head_fwd_bf16w<TT, 2>), grid, dim3(64), 0, st, gate_q, gate_k, uq, uk, b1, w2r, b2r, w3r, \
                         b3r, w2c, b2c, w3c, b3c, triplet_dist, node_cls, batch, num_query, num_rel, num_cls_plus1,  \
                         rel_logits, conn_logits, gate_mean);                                                        \
    break;
  switch (num_slots) {
    EGTR_TB(1) EGTR_TB(2) EGTR_TB(3) EGTR_TB(4) EGTR_TB(5) EGTR_TB(6) EGTR_TB(7) EGTR_TB(8) EGTR_TB(9) EGTR_TB(10)
    default: return EGTR_E_UNSUPPORTED;
  }
#undef EGTR_TB
  return egtr_check_launch();
}


extern "C" int egtr_rel_head_forward_bf16x6_f32(egtr_stream_t stream, const float* gate_q, const float* gate_k,
                                                const float* uq, const float* uk, const float* b1,
                                                const uint16_t* w2x_rel, const float* b2r, const uint16_t* w3x_rel,
                                                const float* b3r, const uint16_t* w2x_conn, const float* b2c,
                                                const float* w3c, const float* b3c, const float* triplet_dist,
                                                const int64_t* node_cls, int batch, int num_query, int num_slots,
                                                int hidden, int num_rel, int num_cls_plus1, float* rel_logits,
                                                float* conn_logits, float* gate_mean, int apply_sigmoid) {
  if (!gate_q || !gate_k || !uq || !uk || !b1 || !w2x_rel || !b2r || !w3x_rel || !b3r || !w2x_conn || !b2c || !w3c ||
      !b3c || !rel_logits || !conn_logits)
    return EGTR_E_ARG;
  if (triplet_dist != nullptr && node_cls == nullptr) return EGTR_E_ARG;
  if (batch <= 0 || num_query <= 0 || num_slots <= 0 || num_rel <= 0) return EGTR_E_ARG;
  // 9 slots (8 decoder layers + 1) is the largest instantiation that stays within 256 registers without scratch
  if (hidden != kHd || num_rel > 64 || num_slots > 9) return EGTR_E_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long tiles = (long long)batch * ((num_query + 7) / 8) * ((num_query + 3) / 4);
  if (tiles >= (1ll << 31)) return EGTR_E_UNSUPPORTED;
  const dim3 grid((unsigned)tiles, 2);
#define EGTR_TX(TT)                                                                                                  \
  case TT:                                                                                                           \
    if (num_rel <= 32)                                                                                               \
      hipLaunchKernelGGL((rel_head_fwd_x6<TT, 1>), grid, dim3(64 * kRhWaves), 0, st, gate_q, gate_k, uq, uk, b1,     \
                         w2x_rel, b2r, w3x_rel, b3r, w2x_conn, b2c, w3c, b3c, triplet_dist, node_cls, batch,         \
                         num_query, num_rel, num_cls_plus1, rel_logits, conn_logits, gate_mean, apply_sigmoid,       \
                         (float*)nullptr, (float*)nullptr);                                                          \
    else                                                                                                             \
      hipLaunchKernelGGL((rel_head_fwd_x6<TT, 2>), grid, dim3(64 * kRhWaves), 0, st, gate_q, gate_k, uq, uk, b1,     \
                         w2x_rel, b2r, w3x_rel, b3r, w2x_conn, b2c, w3c, b3c, triplet_dist, node_cls, batch,         \
                         num_query, num_rel, num_cls_plus1, rel_logits, conn_logits, gate_mean, apply_sigmoid,       \
                         (float*)nullptr, (float*)nullptr);                                                          \
    break;
  switch (num_slots) {
    EGTR_TX(1) EGTR_TX(2) EGTR_TX(3) EGTR_TX(4) EGTR_TX(5) EGTR_TX(6) EGTR_TX(7) EGTR_TX(8) EGTR_TX(9)
    default: return EGTR_E_UNSUPPORTED;
  }
#undef EGTR_TX
  return egtr_check_launch();
}

// Training forward on the split arithmetic: as above, and the post-ReLU activations of both layers are stored for the backward.
// num_slots 4 (three decoder layers: the small fixtures) or 7 (six: the reference's configuration); others: EGTR_E_UNSUPPORTED
// (the caller keeps the exact-f32 kernel, egtr_rel_head_forward_save_f32).
extern "C" int egtr_rel_head_forward_bf16x6_save_f32(egtr_stream_t stream, const float* gate_q, const float* gate_k,
                                                     const float* uq, const float* uk, const float* b1,
                                                     const uint16_t* w2x_rel, const float* b2r, const uint16_t* w3x_rel,
                                                     const float* b3r, const uint16_t* w2x_conn, const float* b2c,
                                                     const float* w3c, const float* b3c, const float* triplet_dist,
                                                     const int64_t* node_cls, int batch, int num_query, int num_slots,
                                                     int hidden, int num_rel, int num_cls_plus1, float* rel_logits,
                                                     float* conn_logits, float* gate_mean, float* h1_save,
                                                     float* h2_save) {
  if (!gate_q || !gate_k || !uq || !uk || !b1 || !w2x_rel || !b2r || !w3x_rel || !b3r || !w2x_conn || !b2c || !w3c ||
      !b3c || !rel_logits || !conn_logits || !h1_save || !h2_save)
    return EGTR_E_ARG;
  if (triplet_dist != nullptr && node_cls == nullptr) return EGTR_E_ARG;
  if (batch <= 0 || num_query <= 0 || num_rel <= 0) return EGTR_E_ARG;
  if (hidden != kHd || num_rel > 64 || (num_slots != 4 && num_slots != 7)) return EGTR_E_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long tiles = (long long)batch * ((num_query + 7) / 8) * ((num_query + 3) / 4);
  if (tiles >= (1ll << 31)) return EGTR_E_UNSUPPORTED;
  const dim3 grid((unsigned)tiles, 2);
#define EGTR_TXS(TT, OTT)                                                                                            \
  hipLaunchKernelGGL((rel_head_fwd_x6<TT, OTT, true>), grid, dim3(64 * kRhWaves), 0, st, gate_q, gate_k, uq, uk, b1,  \
                     w2x_rel, b2r, w3x_rel, b3r, w2x_conn, b2c, w3c, b3c, triplet_dist, node_cls, batch, num_query,   \
                     num_rel, num_cls_plus1, rel_logits, conn_logits, gate_mean, 0, h1_save, h2_save)
  if (num_slots == 4) {
    if (num_rel <= 32) EGTR_TXS(4, 1); else EGTR_TXS(4, 2);
  } else {
    if (num_rel <= 32) EGTR_TXS(7, 1); else EGTR_TXS(7, 2);
  }
#undef EGTR_TXS
  return egtr_check_launch();
}

// The operand streams of rel_head_fwd_x6 from the fp32 weights in ONE launch (training rebuilds them every step):
//   w2x [8 nt][16 t][3 piece][2 hf][32 pi][8 e]      = piece of W2[32 nt + pi][16 t + 8 hf + e]          (both MLPs)
//   w3x [8 nt][2 kb][OT][3 piece][2 hf][32 pi][8 e]  = piece of W3[32 ot + pi][32 nt + 16 kb + (e & 3) + 8 (e >> 2) + 4 hf]
// (rows of W3 beyond num_rel are zero), pieces rounded to nearest (hi + mid + lo == w to fp32 precision).
namespace {
__global__ __launch_bounds__(256) void rel_head_streams_f32(const float* __restrict__ w2r, const float* __restrict__ w2c,
                                                            const float* __restrict__ w3r, int R, int OT,
                                                            unsigned short* __restrict__ w2xr,
                                                            unsigned short* __restrict__ w2xc,
                                                            unsigned short* __restrict__ w3x) {
  const int n2 = kHd * kHd, n3 = 8 * 2 * OT * 2 * 32 * 8;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < 2 * n2 + n3; i += gridDim.x * 256) {
    if (i < 2 * n2) {
      const int which = i / n2, j = i - which * n2;        // j walks the DESTINATION order without the piece index
      const int e = j & 7, pi = (j >> 3) & 31, hf = (j >> 8) & 1, t = (j >> 9) & 15, nt = j >> 13;
      const float w = (which ? w2c : w2r)[(32 * nt + pi) * kHd + 16 * t + 8 * hf + e];
      const xs::Split3 sp = xs::split3_rne(w);
      unsigned short* d = (which ? w2xc : w2xr) + (size_t)((nt * 16 + t) * 3) * 512 + (hf * 32 + pi) * 8 + e;
      d[0] = (unsigned short)(sp.hi >> 16);
      d[512] = (unsigned short)(sp.mid >> 16);
      d[1024] = (unsigned short)(sp.lo >> 16);
    } else {
      const int j = i - 2 * n2;
      const int e = j & 7, pi = (j >> 3) & 31, hf = (j >> 8) & 1;
      const int rest = j >> 9, ot = rest % OT, kb = (rest / OT) & 1, nt = rest / (2 * OT);
      const int row = 32 * ot + pi, col = 32 * nt + 16 * kb + (e & 3) + 8 * (e >> 2) + 4 * hf;
      const float w = row < R ? w3r[row * kHd + col] : 0.f;
      const xs::Split3 sp = xs::split3_rne(w);
      unsigned short* d = w3x + (size_t)(((nt * 2 + kb) * OT + ot) * 3) * 512 + (hf * 32 + pi) * 8 + e;
      d[0] = (unsigned short)(sp.hi >> 16);
      d[512] = (unsigned short)(sp.mid >> 16);
      d[1024] = (unsigned short)(sp.lo >> 16);
    }
  }
}
}  // namespace

extern "C" int egtr_rel_head_streams_f32(egtr_stream_t stream, const float* w2_rel, const float* w2_conn,
                                         const float* w3_rel, int hidden, int num_rel, uint16_t* w2x_rel,
                                         uint16_t* w2x_conn, uint16_t* w3x_rel) {
  if (!w2_rel || !w2_conn || !w3_rel || !w2x_rel || !w2x_conn || !w3x_rel || num_rel <= 0) return EGTR_E_ARG;
  if (hidden != kHd || num_rel > 64) return EGTR_E_UNSUPPORTED;
  const int OT = num_rel <= 32 ? 1 : 2;
  hipLaunchKernelGGL(rel_head_streams_f32, dim3(256), dim3(256), 0, static_cast<hipStream_t>(stream), w2_rel, w2_conn,
                     w3_rel, num_rel, OT, w2x_rel, w2x_conn, w3x_rel);
  return egtr_check_launch();
}
