// EGTR relation head of a bf16 model, all three layers on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16).
//
// Same algebra as rel_head.hip (model/egtr.py:366-416):
//   h1[i,j,:] = relu(b1 + sum_t g[i,j,t] (uq[i,t,:] + uk[j,t,:])),  g = sigmoid(gate_q[i,t] + gate_k[j,t])
//   rel = W3r relu(W2r h1 + b2r) + b3r (+ triplet_dist[cls_i, cls_j, :]),   conn = w3c . relu(W2c h1' + b2c) + b3c
// rel_head_fwd_bf16w (rel_head.hip) builds h1 on the VALU from fp32 tables: every pair reads T rows of uk (9 KB at T = 9) and
// every wave streams W2 (128 KB) from L2 -- 37 GB of L2 -> L1 traffic for the stress batch (16 x 300^2 pairs), 3.8 ms at the
// ~10 TB/s the L2s deliver.  Here
//   * layer 1 is a matrix product as well: a wave owns 4 subjects x 8 objects = 32 pairs, i.e. 12 per-query "row sets"
//     (T <= 16 slots each); with U the 16 x 256 slot table of a row set and G its gate block,
//         h1^T[ch][pair] = sum over the 12 row sets  U_rs^T[ch][slot] . G_rs[slot][pair],   G_rs[slot][pair] = g[pair][slot] if
//     the pair uses row set rs, else 0 -- one K = 16 step per row set and 32-channel tile, 96 MFMAs per 32 pairs (layer 2: 128).
//     The tables arrive pre-packed in operand order (rel_head_pack_tables: bf16, [row][mlp][tile][half][channel][8 slots]) so a
//     lane's operand is one 16-byte load: 96 KB per 32 pairs instead of 300 KB;
//   * W2 of the workgroup's MLP lives in LDS (128 KB, staged once by a persistent workgroup of 8 waves, in operand order);
//   * the accumulators of a layer are the next layer's operands in place: D[row = channel][col = pair] of layers 1 / 2 holds,
//     per lane (pair, half), channels {0-3, 8-11} + 4 half (registers 0..7) and {16-19, 24-27} + 4 half (8..15) of a 32-wide
//     tile -- two K = 16 steps whose k slot e stands for channel (e & 3) + 8 (e >> 2) + 4 half (+ 16); the W2 / W3 operands
//     are stored / gathered in that order;
//   * layer 3 runs NON-transposed (A = h2, B = W3): D[row = pair][col = relation], so a store instruction writes 32
//     consecutive floats of two pairs -- no LDS staging of the output tile.
// g and the tables are rounded to bf16 as operands (the reference's bf16 model holds both in bf16); accumulation is fp32.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kHd = 256;          // hidden width of both MLPs
constexpr int kSlots = 16;        // slots per row set in the packed tables (T <= 16)
constexpr int kWaves = 8;
constexpr int kRowBytes = 2 * 8 * 2 * 32 * 16;   // one packed row: [mlp 2][tile 8][half 2][channel 32] x 16 bytes = 16 KiB

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 pack8(const float (&f)[8]) {
  bf16x8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (__bf16)f[i];
  return v;
}
__device__ __forceinline__ float bf16_bits(unsigned short u) { return __uint_as_float(((unsigned)u) << 16); }

// Table fragments are requested a whole channel tile ahead: asm loads (program order is kept) with counted waits, the
// destination registers written by the hardware while the previous tile is multiplied (tools/check_async_loads.py).
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ f32x4v gload_frag(const char* sbase, unsigned voff) {
  f32x4v v;
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=&v"(v) : "v"(voff), "s"(sbase), "n"(OFF));
  return v;
}
template <int N>
__device__ __forceinline__ void vm_wait(f32x4v& v) {
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v) : "n"(N));
}

// ---- tables -> operand order ----------------------------------------------------------------------------------------
// u: [rows, T, 512] (fp32 or bf16 bits) -> packed [rows][mlp][tile c][half][channel 32][8 slots] bf16, slot = 8 half + e,
// zero for slots >= T.  One thread per 16-byte operand.
template <typename InT>
__global__ __launch_bounds__(256) void rel_head_pack_tables(const InT* __restrict__ u, int rows, int T,
                                                            uint4* __restrict__ packed) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;   // ((((row * 2 + mlp) * 8 + c) * 2 + hf) * 32 + ch)
  if (idx >= (long long)rows * 1024) return;
  const int ch = (int)(idx & 31), hf = (int)((idx >> 5) & 1), c = (int)((idx >> 6) & 7), mlp = (int)((idx >> 9) & 1);
  const long long row = idx >> 10;
  const InT* src = u + (size_t)row * T * 512 + mlp * 256 + c * 32 + ch;
  float f[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int t = 8 * hf + e;
    float v = 0.f;
    if (t < T) {
      if constexpr (sizeof(InT) == 4) v = (float)src[(size_t)t * 512];
      else v = bf16_bits((unsigned short)src[(size_t)t * 512]);
    }
    f[e] = v;
  }
  const bf16x8 p = pack8(f);
  packed[idx] = __builtin_bit_cast(uint4, p);
}

// ---- forward ----------------------------------------------------------------------------------------------------------
struct RhArgs {
  const float* gate_q;       // [B, N, T]
  const float* gate_k;
  const char* uq;            // packed tables, kRowBytes per (b, n)
  const char* uk;
  const float* b1;           // [512]
  const unsigned short* w2r; // [256, 256] bf16
  const float* b2r;
  const unsigned short* w3r; // [R, 256] bf16
  const float* b3r;
  const unsigned short* w2c;
  const float* b2c;
  const unsigned short* w3c; // [256]
  const float* b3c;
  const float* triplet;      // [C1, C1, R] or null
  const int64_t* node_cls;   // [B, N]
  float* rel_logits;         // [B, N, N, R]
  float* conn_logits;        // [B, N, N]
  float* gate_mean;          // [T] (zeroed by the caller) or null
  int B, N, R, C1;
};

// W3LDS: W3 of the relation MLP in LDS as well, in operand order ([n tile][kb][half][relation < R] x 16 bytes = 512 R bytes
// behind W2; R <= 60 fits the 160 KiB): its fragments then cost an LDS read instead of an exposed L2 round trip per n tile.
template <int T, int OT, bool W3LDS>
__global__ __launch_bounds__(64 * kWaves) void rel_head_fwd_bf16p(RhArgs A) {
  extern __shared__ __attribute__((aligned(16))) char s_w2[];   // [n tile 8][k step 16][lane 64] x 16 bytes = 128 KiB
  __shared__ __attribute__((aligned(16))) float s_b1[kHd];      // this MLP's layer-1 / layer-2 biases: LDS reads do not
  __shared__ __attribute__((aligned(16))) float s_b2[kHd];      // share a counter with the table fragments in flight
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pi = lane & 31, hf = lane >> 5;
  const int mlp = blockIdx.x & 1, wg = blockIdx.x >> 1, nwg = gridDim.x >> 1;
  const unsigned short* w2 = mlp ? A.w2c : A.w2r;
  const float* b2 = mlp ? A.b2c : A.b2r;
  const int N = A.N, R = A.R;

  // W2 -> LDS in operand order: fragment (nt, ks = 2 c + kb), lane (n, half): W2[nt*32 + n][32 c + 16 kb + (e&3) + 8 (e>>2) + 4 half]
  for (int f = tid; f < 8 * 16 * 64; f += 64 * kWaves) {
    const int l = f & 63, ks = (f >> 6) & 15, nt = f >> 10;
    const unsigned short* p = w2 + (size_t)(nt * 32 + (l & 31)) * kHd + (ks >> 1) * 32 + (ks & 1) * 16 + 4 * (l >> 5);
    const uint2 lo = *reinterpret_cast<const uint2*>(p), hi = *reinterpret_cast<const uint2*>(p + 8);
    reinterpret_cast<uint4*>(s_w2)[f] = make_uint4(lo.x, lo.y, hi.x, hi.y);
  }
  if (tid < kHd) {
    s_b1[tid] = A.b1[mlp * kHd + tid];
    s_b2[tid] = b2[tid];
  }
  uint4* const s_w3 = reinterpret_cast<uint4*>(s_w2 + 8 * 16 * 64 * 16);
  if (W3LDS && mlp == 0) {
    for (int f = tid; f < 32 * R; f += 64 * kWaves) {
      const int ro = f % R, q = f / R, h = q & 1, kb = (q >> 1) & 1, nt = q >> 2;
      const unsigned short* p = A.w3r + (size_t)ro * kHd + nt * 32 + 16 * kb + 4 * h;
      const uint2 lo = *reinterpret_cast<const uint2*>(p), hi = *reinterpret_cast<const uint2*>(p + 8);
      s_w3[f] = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
  }
  __syncthreads();

  // The eight waves of a workgroup take a 2 x 4 block of pair tiles (8 subjects x 32 objects) per step: a subject row set is then
  // wanted by four waves and an object row set by two at about the same time and comes out of the L1 for all but the first
  // (40 distinct row sets per step instead of 68 with eight tiles in a row: the kernel is bound by L2 -> L1 operand traffic).
#ifdef EGTR_RH_TILES_IN_A_ROW
  const int it_n = (N + 3) >> 2, jt_n = (N + 7) >> 3;
  const int sit_n = it_n, sjt_n = (jt_n + 7) >> 3;
#else
  const int it_n = (N + 3) >> 2, jt_n = (N + 7) >> 3;
  const int sit_n = (it_n + 1) >> 1, sjt_n = (jt_n + 3) >> 2;
#endif
  const int supers_img = sit_n * sjt_n;
  const long long nsupers = (long long)A.B * supers_img;
  const float inv_total = 1.f / ((float)A.B * (float)N * (float)N);
  const int il = pi >> 3, jl = pi & 7;

  for (long long sup = wg; sup < nsupers; sup += nwg) {
    const int b = __builtin_amdgcn_readfirstlane((int)(sup / supers_img));
    const int rem = __builtin_amdgcn_readfirstlane((int)(sup - (long long)b * supers_img));
#ifdef EGTR_RH_TILES_IN_A_ROW
    const int it = rem / sjt_n, jt = (rem - it * sjt_n) * 8 + wave;
#else
    const int it = (rem / sjt_n) * 2 + (wave >> 2), jt = (rem % sjt_n) * 4 + (wave & 3);
#endif
    if (it >= it_n || jt >= jt_n) continue;   // (wave-uniform; no workgroup barrier inside the loop)
    const int i = it * 4 + il, j = jt * 8 + jl;
    const bool valid = i < N && j < N;
    const int ic = min(i, N - 1), jc = min(j, N - 1);
    const size_t qi = (size_t)b * N + ic, kj = (size_t)b * N + jc;

    // ---- gates: g[t] = sigmoid(gate_q[i, t] + gate_k[j, t]); bf16 operand halves (slots 8 half .. 8 half + 7)
    float g[kSlots];
#pragma unroll
    for (int t = 0; t < kSlots; ++t) {
      if (t < T) {
        const float x = A.gate_q[qi * T + t] + A.gate_k[kj * T + t];
        g[t] = valid ? 1.f / (1.f + __expf(-x)) : 0.f;
      } else {
        g[t] = 0.f;
      }
    }
    if (A.gate_mean != nullptr && mlp == 0) {
#pragma unroll
      for (int t = 0; t < T; ++t) {
        float v = hf == 0 ? g[t] : 0.f;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
        if (lane == 0) unsafeAtomicAdd(A.gate_mean + t, v * inv_total);
      }
    }
    bf16x8 gop;
    {
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = hf ? g[8 + e] : g[e];
      gop = pack8(f);
    }
    bf16x8 zero8;
#pragma unroll
    for (int e = 0; e < 8; ++e) zero8[e] = (__bf16)0.f;

    // ---- layer 1: 8 channel tiles x 12 row sets; the result of tile c becomes k steps 2 c, 2 c + 1 of layer 2 ----------
    bf16x8 h1b[16];
    {
      // row sets: subjects it*4 .. +3 from uq, objects jt*8 .. +7 from uk.  One scalar base per table (biased by 4 KiB so that
      // the eight tile offsets c * 1 KiB - 4 KiB fit the signed 13-bit immediate), a 32-bit lane offset per row set.
      const char* bq = A.uq + mlp * (kRowBytes / 2) + 4096;
      const char* bk = A.uk + mlp * (kRowBytes / 2) + 4096;
      unsigned voff[12];
#pragma unroll
      for (int r = 0; r < 4; ++r)
        voff[r] = (unsigned)(b * N + min(it * 4 + r, N - 1)) * (unsigned)kRowBytes + (unsigned)((hf * 32 + pi) * 16);
#pragma unroll
      for (int r = 0; r < 8; ++r)
        voff[4 + r] = (unsigned)(b * N + min(jt * 8 + r, N - 1)) * (unsigned)kRowBytes + (unsigned)((hf * 32 + pi) * 16);
      // A ring of 12 fragments: the fragment of (tile c + 1, row set rs) is requested into the register the product of
      // (tile c, row set rs) has just read, so 12 loads (12 KiB per wave) are in flight throughout the layer.
      f32x4v fa[12];
      asm volatile("s_nop 4");   // the bases derive from v_readfirstlane results: VALU-written SGPR -> VMEM address hazard
#define EGTR_L1_ISSUE(C, RS) fa[RS] = gload_frag<(C) * 1024 - 4096>((RS) < 4 ? bq : bk, voff[RS]);
#define EGTR_L1_STEP(C, RS)                                                                                      \
      vm_wait<((C) < 7 ? 11 : 11 - (RS))>(fa[RS]);                                                                \
      acc = mfma_bf16(__builtin_bit_cast(bf16x8, fa[RS]), ((RS) < 4 ? (il == (RS)) : (jl == (RS) - 4)) ? gop : zero8, acc); \
      if ((C) < 7) { EGTR_L1_ISSUE(((C) < 7 ? (C) + 1 : 7), RS) }
#define EGTR_L1_TILE(C)                                                                                          \
      {                                                                                                          \
        f32x16 acc;                                                                                              \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[r] = 0.f;                                             \
        EGTR_L1_STEP(C, 0) EGTR_L1_STEP(C, 1) EGTR_L1_STEP(C, 2) EGTR_L1_STEP(C, 3) EGTR_L1_STEP(C, 4)            \
        EGTR_L1_STEP(C, 5) EGTR_L1_STEP(C, 6) EGTR_L1_STEP(C, 7) EGTR_L1_STEP(C, 8) EGTR_L1_STEP(C, 9)            \
        EGTR_L1_STEP(C, 10) EGTR_L1_STEP(C, 11)                                                                  \
        float h[16];                                                                                             \
        _Pragma("unroll") for (int rq = 0; rq < 4; ++rq) {                                                       \
          const float4 bb = *reinterpret_cast<const float4*>(s_b1 + (C) * 32 + 8 * rq + 4 * hf);                 \
          h[4 * rq + 0] = egtr_relu(acc[4 * rq + 0] + bb.x);                                                     \
          h[4 * rq + 1] = egtr_relu(acc[4 * rq + 1] + bb.y);                                                     \
          h[4 * rq + 2] = egtr_relu(acc[4 * rq + 2] + bb.z);                                                     \
          h[4 * rq + 3] = egtr_relu(acc[4 * rq + 3] + bb.w);                                                     \
        }                                                                                                        \
        const float lo[8] = {h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]};                                    \
        const float hi[8] = {h[8], h[9], h[10], h[11], h[12], h[13], h[14], h[15]};                              \
        h1b[2 * (C)] = pack8(lo);                                                                                \
        h1b[2 * (C) + 1] = pack8(hi);                                                                            \
      }
#ifdef EGTR_RH_ABL_NO_L1
      if (A.B < 0) {
#endif
      EGTR_L1_ISSUE(0, 0) EGTR_L1_ISSUE(0, 1) EGTR_L1_ISSUE(0, 2) EGTR_L1_ISSUE(0, 3) EGTR_L1_ISSUE(0, 4) EGTR_L1_ISSUE(0, 5)
      EGTR_L1_ISSUE(0, 6) EGTR_L1_ISSUE(0, 7) EGTR_L1_ISSUE(0, 8) EGTR_L1_ISSUE(0, 9) EGTR_L1_ISSUE(0, 10) EGTR_L1_ISSUE(0, 11)
      EGTR_L1_TILE(0) EGTR_L1_TILE(1) EGTR_L1_TILE(2) EGTR_L1_TILE(3) EGTR_L1_TILE(4) EGTR_L1_TILE(5) EGTR_L1_TILE(6)
      EGTR_L1_TILE(7)
#ifdef EGTR_RH_ABL_NO_L1
      } else {
        for (int t2 = 0; t2 < 16; ++t2) h1b[t2] = gop;
      }
#endif
#undef EGTR_L1_TILE
#undef EGTR_L1_STEP
#undef EGTR_L1_ISSUE
    }

    // ---- layers 2 and 3 ---------------------------------------------------------------------------------------------
    f32x16 racc[OT];
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
      for (int r = 0; r < 16; ++r) racc[ot][r] = 0.f;
    float cacc = 0.f;
#ifdef EGTR_RH_ABL_NO_L23
    if (A.B < 0)
#endif
#pragma unroll 1
    for (int nt = 0; nt < kHd / 32; ++nt) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const uint4* wf = reinterpret_cast<const uint4*>(s_w2) + nt * 16 * 64 + lane;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) acc = mfma_bf16(__builtin_bit_cast(bf16x8, wf[ks * 64]), h1b[ks], acc);
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const float4 bb = *reinterpret_cast<const float4*>(s_b2 + nt * 32 + 8 * rq + 4 * hf);
        acc[4 * rq + 0] = egtr_relu(acc[4 * rq + 0] + bb.x);
        acc[4 * rq + 1] = egtr_relu(acc[4 * rq + 1] + bb.y);
        acc[4 * rq + 2] = egtr_relu(acc[4 * rq + 2] + bb.z);
        acc[4 * rq + 3] = egtr_relu(acc[4 * rq + 3] + bb.w);
      }
      if (mlp == 0) {
        // relation logits, D[row = pair][col = relation]: A = h2 (this lane's pair), B = W3 rows gathered in k-slot order
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          const float hv[8] = {acc[8 * kb + 0], acc[8 * kb + 1], acc[8 * kb + 2], acc[8 * kb + 3],
                               acc[8 * kb + 4], acc[8 * kb + 5], acc[8 * kb + 6], acc[8 * kb + 7]};
          const bf16x8 hb = pack8(hv);
#pragma unroll
          for (int ot = 0; ot < OT; ++ot) {
            const int ro = ot * 32 + pi;
            const bool rok = ro < R;
            uint4 wv;
            if constexpr (W3LDS) {
              wv = s_w3[((nt * 2 + kb) * 2 + hf) * R + (rok ? ro : 0)];
              if (!rok) wv = make_uint4(0u, 0u, 0u, 0u);
            } else {
              const unsigned short* w3p = A.w3r + (size_t)(rok ? ro : 0) * kHd + nt * 32 + 16 * kb + 4 * hf;
              uint2 lo = *reinterpret_cast<const uint2*>(w3p), hi = *reinterpret_cast<const uint2*>(w3p + 8);
              if (!rok) { lo = make_uint2(0u, 0u); hi = make_uint2(0u, 0u); }
              wv = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
            racc[ot] = mfma_bf16(hb, __builtin_bit_cast(bf16x8, wv), racc[ot]);
          }
        }
      } else {
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const uint2 wv = *reinterpret_cast<const uint2*>(A.w3c + nt * 32 + 8 * rq + 4 * hf);
          cacc += bf16_bits((unsigned short)(wv.x & 0xffffu)) * acc[4 * rq + 0] +
                  bf16_bits((unsigned short)(wv.x >> 16)) * acc[4 * rq + 1] +
                  bf16_bits((unsigned short)(wv.y & 0xffffu)) * acc[4 * rq + 2] +
                  bf16_bits((unsigned short)(wv.y >> 16)) * acc[4 * rq + 3];
        }
      }
    }

    if (mlp == 1) {
      cacc += __shfl_xor(cacc, 32);
      if (valid && hf == 0) A.conn_logits[((size_t)b * N + i) * N + j] = cacc + A.b3c[0];
      continue;
    }
    // ---- relation epilogue: register r of lane (relation ro, half) is pair (r & 3) + 8 (r >> 2) + 4 half -------------------
    int tb = -1;
    if (A.triplet != nullptr) tb = ((int)A.node_cls[qi] * A.C1 + (int)A.node_cls[kj]) * R;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int p0 = (r & 3) + 8 * (r >> 2);          // pair of half 0; half 1: + 4
      const int tb0 = __builtin_amdgcn_readlane(tb, p0), tb1 = __builtin_amdgcn_readlane(tb, p0 + 4);
      const int pr = p0 + 4 * hf, tbp = hf ? tb1 : tb0;
      const int pil = pr >> 3, pjl = pr & 7;
      const int gi = it * 4 + pil, gj = jt * 8 + pjl;
#ifdef EGTR_RH_ABL_NO_STORE
      if (gi < N && gj < N && A.B < 0) {
#else
      if (gi < N && gj < N) {
#endif
        float* dst = A.rel_logits + (((size_t)b * N + gi) * N + gj) * R;
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) {
          const int ro = ot * 32 + pi;
          if (ro < R) {
            float v = racc[ot][r] + A.b3r[ro];
            if (tbp >= 0) v += A.triplet[tbp + ro];
            dst[ro] = v;
          }
        }
      }
    }
  }
}

}  // namespace

extern "C" int egtr_rel_head_pack_tables_bf16(egtr_stream_t stream, const void* u, int u_is_bf16, int rows, int num_slots,
                                              uint16_t* packed) {
  if (!u || !packed) return EGTR_E_ARG;
  if (rows <= 0 || num_slots <= 0) return EGTR_E_ARG;
  if (num_slots > kSlots) return EGTR_E_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long n = (long long)rows * 1024;
  const dim3 grid((unsigned)((n + 255) / 256));
  if (u_is_bf16)
    hipLaunchKernelGGL(rel_head_pack_tables<unsigned short>, grid, dim3(256), 0, st, static_cast<const unsigned short*>(u),
                       rows, num_slots, reinterpret_cast<uint4*>(packed));
  else
    hipLaunchKernelGGL(rel_head_pack_tables<float>, grid, dim3(256), 0, st, static_cast<const float*>(u), rows, num_slots,
                       reinterpret_cast<uint4*>(packed));
  return egtr_check_launch();
}

extern "C" int egtr_rel_head_forward_bf16p(egtr_stream_t stream, const float* gate_q, const float* gate_k,
                                           const uint16_t* uq_packed, const uint16_t* uk_packed, const float* b1,
                                           const uint16_t* w2r, const float* b2r, const uint16_t* w3r, const float* b3r,
                                           const uint16_t* w2c, const float* b2c, const uint16_t* w3c, const float* b3c,
                                           const float* triplet_dist, const int64_t* node_cls, int batch, int num_query,
                                           int num_slots, int hidden, int num_rel, int num_cls_plus1, float* rel_logits,
                                           float* conn_logits, float* gate_mean) {
  if (!gate_q || !gate_k || !uq_packed || !uk_packed || !b1 || !w2r || !b2r || !w3r || !b3r || !w2c || !b2c || !w3c ||
      !b3c || !rel_logits || !conn_logits)
    return EGTR_E_ARG;
  if (triplet_dist != nullptr && node_cls == nullptr) return EGTR_E_ARG;
  if (batch <= 0 || num_query <= 0 || num_slots <= 0 || num_rel <= 0) return EGTR_E_ARG;
  if (hidden != kHd || num_rel > 64 || num_slots > 10) return EGTR_E_UNSUPPORTED;
  if ((long long)batch * num_query * num_query * num_rel >= (1ll << 40)) return EGTR_E_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  RhArgs A;
  A.gate_q = gate_q; A.gate_k = gate_k;
  A.uq = reinterpret_cast<const char*>(uq_packed); A.uk = reinterpret_cast<const char*>(uk_packed);
  A.b1 = b1; A.w2r = w2r; A.b2r = b2r; A.w3r = w3r; A.b3r = b3r; A.w2c = w2c; A.b2c = b2c; A.w3c = w3c; A.b3c = b3c;
  A.triplet = triplet_dist; A.node_cls = node_cls; A.rel_logits = rel_logits; A.conn_logits = conn_logits;
  A.gate_mean = gate_mean; A.B = batch; A.N = num_query; A.R = num_rel; A.C1 = num_cls_plus1;
  const long long it_n = (num_query + 3) / 4, jt_n = (num_query + 7) / 8;
  const long long ntiles = (long long)batch * ((it_n + 1) / 2) * ((jt_n + 3) / 4) * kWaves;   // 2 x 4 tile blocks per workgroup step
  // persistent workgroups, one per CU (128 KiB of LDS): even ids the relation MLP, odd ids the connectivity MLP
  int cus = 256;
  {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
  }
  long long per_mlp = (ntiles + kWaves - 1) / kWaves;
  if (per_mlp > cus / 2) per_mlp = cus / 2;
  if (per_mlp < 1) per_mlp = 1;
  const dim3 grid((unsigned)(2 * per_mlp));
  constexpr int kLds = 8 * 16 * 64 * 16;
  const bool w3lds = num_rel <= 60;
  const int lds = kLds + (w3lds ? 512 * num_rel : 0);
  static unsigned long long raised[11][4] = {};
#define EGTR_TP_ONE(TT, OTV, WL)                                                                                 \
  {                                                                                                              \
    if (int e = egtr_raise_dynamic_lds(reinterpret_cast<const void*>(&rel_head_fwd_bf16p<TT, OTV, WL>),          \
                                       kLds + (WL ? 512 * 60 : 0), &raised[TT][2 * (OTV - 1) + (WL ? 1 : 0)]))   \
      return e;                                                                                                  \
    hipLaunchKernelGGL((rel_head_fwd_bf16p<TT, OTV, WL>), grid, dim3(64 * kWaves), lds, st, A);                  \
  }
#define EGTR_TP(TT)                                                                                              \
  case TT:                                                                                                       \
    if (num_rel <= 32) EGTR_TP_ONE(TT, 1, true)                                                                  \
    else if (w3lds) EGTR_TP_ONE(TT, 2, true)                                                                     \
    else EGTR_TP_ONE(TT, 2, false)                                                                               \
    break;
  switch (num_slots) {
    EGTR_TP(1) EGTR_TP(2) EGTR_TP(3) EGTR_TP(4) EGTR_TP(5) EGTR_TP(6) EGTR_TP(7) EGTR_TP(8) EGTR_TP(9) EGTR_TP(10)
    default: return EGTR_E_UNSUPPORTED;
  }
#undef EGTR_TP
#undef EGTR_TP_ONE
  return egtr_check_launch();
}
