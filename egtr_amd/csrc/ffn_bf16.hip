// The encoder layer's feed-forward block of a bf16 model as ONE kernel:
//     y = LayerNorm(x + fc2(relu(fc1(x))))   [+ y + pos]            (model/deformable_detr.py:1335-1345, d_model = 256)
// bf16 storage (raw bits), fp32 accumulation on v_mfma_f32_32x32x16_bf16.  The vendor route of the stress configuration
// (M = 16 x 22 223 rows) runs two GEMMs with a [M, 1024] bf16 activation written and read in between (1.46 GB of the 1.8 GB
// the block moves: both GEMMs are bound by it, 507 us per layer), then a residual + LayerNorm pass (another 130 us).  Here the
// hidden activation never leaves the registers:
//   * a workgroup of 4 waves owns 256 rows, a wave two blocks of 32 (one wave per SIMD with the whole register file: 8 waves
//     of one block each had 20 registers left for weight fragments, every product waited for its LDS read: 759 us); the
//     wave's two 32 x 256 input panels stay in registers for the whole block as the B-operand fragments of fc1 (lane (row, half)
//     holds x[row][16 s + 8 half + 0..7]: 16-byte loads from the row-major tensor);
//   * the hidden dimension is walked in tiles of 32 units.  fc1 of a tile, transposed: D1[unit][row] = W1[unit][:] . x[row][:]
//     (16 K = 16 steps, A = rows of W1 as they lie in memory); bias + ReLU + rounding to bf16 on the accumulator, whose registers
//     0..7 / 8..15 ARE the B operands of two K = 16 steps of fc2 (k slot e <-> unit (e & 3) + 8 (e >> 2) + 4 half (+ 16));
//     fc2 accumulates the tile into the 256 x 32 output accumulators (8 tiles x 2 steps, A = W2 gathered in that slot order);
//   * W1 / W2 slices of a hidden tile (16 KiB each; pre-packed in operand order by egtr_ffn_pack_weights_bf16, a derived
//     constant of the weights) are staged through LDS by all 512 threads, double-buffered:
//     requested a tile ahead into registers, written after the products of the current tile, one barrier per tile.  Every
//     workgroup streams the 1 MiB of weights once per 256 rows (1.4 GB of L2 reads per layer at M = 355 568);
//   * epilogue in the accumulator layout (lane = row, registers = channels): + bias2, rounded to bf16 like the reference's fc2
//     output, + x (re-read, 8 bytes per register quad), rounded, LayerNorm statistics over the lane's 128 channels + one
//     cross-half exchange, y (and y + pos[row % pos_rows], rounded from the rounded y) stored 8 bytes at a time.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "common.h"
#include "x6_common.h"

namespace {

using x6::bf16x8;
using x6::f32x16;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kD = 256;       // d_model
constexpr int kWaves = 4;     // one wave per SIMD: 512 registers each (the 2 x 8 x 16 fc2 accumulators alone are 256)
constexpr int kRB = 2;        // 32-row blocks per wave: every weight fragment read from LDS feeds two products
constexpr int kRowsWg = 32 * kRB * kWaves;
constexpr int kLds = 128 * 1024;     // two weight slices (64 KiB) during the products; the 256 x 256 bf16 output tile behind them
constexpr int kSlice = 32 * 1024;   // one hidden tile: W1 [16 k steps][64 lanes] + W2 [8 channel tiles][2 k steps][64 lanes], 16 B each

__device__ __forceinline__ float bf2f(unsigned short u) { return __uint_as_float(((unsigned)u) << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {  // round to nearest even; NaN stays NaN
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }
// two floats -> packed bf16 on the hardware converter (v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN), and back
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 v;
  v[0] = (__bf16)a;
  v[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float lo_bf(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hi_bf(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ f32x16 mfma_bf16(uint4 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), b, c, 0, 0, 0);
}

#ifdef EGTR_FFN_TIMING
// debugging aid (tools/ffn_variants.sh): shader-clock cycles of wave 0 per phase, summed over the workgroups of a launch
__device__ unsigned long long g_ffn_stamps[12];
#define FFN_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define FFN_ADD(i, a, b) t_sum[i] += (b) - (a)
#else
#define FFN_T(v) do { } while (0)
#define FFN_ADD(i, a, b) do { } while (0)
#endif

struct FfnArgs {
  const unsigned short* x;      // [M, 256] input = residual
  const uint4* wpk;             // packed weights: [F / 32 hidden tiles][2048 operands] x 16 bytes (ffn_pack_weights)
  const unsigned short* b1;     // [F]
  const unsigned short* b2;     // [256]
  const unsigned short* gamma;  // [256]
  const unsigned short* beta;
  const unsigned short* pos;    // [pos_rows, 256] or null
  unsigned short* y;            // [M, 256]
  unsigned short* y_pos;        // [M, 256] or null
  int M, F, pos_rows;
  float eps;
};

// operand (16 bytes) number f of hidden tile j: f < 1024: W1 fragment (ks = f >> 6, lane = f & 63);
// f >= 1024: W2 fragment (ct = (f - 1024) >> 7, kb = ((f - 1024) >> 6) & 1, lane).  Gathered ONCE into the packed stream by
// ffn_pack_weights (the W2 operand is two 8-byte pieces of a row 2 KiB away from its neighbour's: fetched per workgroup
// and tile it cost 8x its bytes in 64-byte sectors -- 6.4 GB of L2 reads per launch, 992 us; packed: 1.4 GB).
__global__ __launch_bounds__(256) void ffn_pack_weights(const unsigned short* __restrict__ w1,
                                                        const unsigned short* __restrict__ w2, int F,
                                                        uint4* __restrict__ packed) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= (F >> 5) * 2048) return;
  const int j = idx >> 11, f = idx & 2047;
  const int l = f & 63, li = l & 31, hf = l >> 5;
  if (f < 1024) {
    const int ks = f >> 6;
    packed[idx] = *reinterpret_cast<const uint4*>(w1 + (size_t)(j * 32 + li) * kD + 16 * ks + 8 * hf);
    return;
  }
  const int g = f - 1024, ct = g >> 7, kb = (g >> 6) & 1;
  const unsigned short* p = w2 + (size_t)(ct * 32 + li) * F + j * 32 + 16 * kb + 4 * hf;
  const uint2 lo = *reinterpret_cast<const uint2*>(p), hi = *reinterpret_cast<const uint2*>(p + 8);
  packed[idx] = make_uint4(lo.x, lo.y, hi.x, hi.y);
}

__global__ __launch_bounds__(64 * kWaves) __attribute__((amdgpu_waves_per_eu(1, 1))) void ffn_bf16_kernel(FfnArgs A) {
  extern __shared__ __attribute__((aligned(16))) char s_w[];   // two slices of kSlice bytes
  __shared__ float s_b1[1024 + 4 * kD];                           // b1 (F <= 1024), then b2, gamma, beta as fp32
  float* const s_b2 = s_b1 + 1024;
  float* const s_g = s_b2 + kD;
  float* const s_be = s_g + kD;
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, hf = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row0 = blockIdx.x * kRowsWg + wave * (32 * kRB);
  const int ntiles = A.F >> 5;

#ifdef EGTR_FFN_STAGGER
  // Workgroups of the first wave of the grid start EGTR_FFN_STAGGER x 4 us apart in 8 groups: one workgroup per CU and equal
  // work keep them in lock step otherwise -- every CU loads its panel, multiplies and stores at the same moments, and the memory
  // system idles during the products and saturates during the epilogues.
  if (blockIdx.x < 256) {
    const int naps = (int)((blockIdx.x >> 3) & 7) * EGTR_FFN_STAGGER;
#ifndef EGTR_FFN_STAGGER_UNIT
#define EGTR_FFN_STAGGER_UNIT 127
#endif
    for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(EGTR_FFN_STAGGER_UNIT);
  }
#endif
#ifdef EGTR_FFN_TIMING
  unsigned long long t_sum[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  FFN_T(t_start);
  for (int i = tid; i < A.F; i += 64 * kWaves) s_b1[i] = bf2f(A.b1[i]);
  if (tid < kD) {
    s_b2[tid] = bf2f(A.b2[tid]);
    s_g[tid] = bf2f(A.gamma[tid]);
    s_be[tid] = bf2f(A.beta[tid]);
  }
  // slice 0 -> LDS (LDS-DMA: 64 lanes x 16 bytes per instruction, no registers; a wave moves a quarter of a slice)
  const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((x6::lds_char*)s_w);
  const char* const gsl = reinterpret_cast<const char*>(A.wpk) + wave * (kSlice / kWaves);
  const unsigned voff = (unsigned)lane * 16u;
  auto dma_piece = [&](int j, int i) {   // piece i (1 KiB) of this wave's share of slice j
    x6::dma16s(gsl + (size_t)j * kSlice + i * 1024, voff, lds0 + (unsigned)((j & 1) * kSlice + wave * (kSlice / kWaves) + i * 1024));
  };
  constexpr int kPieces = kSlice / kWaves / 1024;
  static_assert(kPieces == 8, "eight DMA instructions per wave and slice");
#pragma unroll
  for (int i = 0; i < kPieces; ++i) dma_piece(0, i);
  // the wave's input panels: 16 fragments per row block (made opaque: left visible as loads from a read-only pointer the
  // compiler re-loads them inside the loop instead of keeping 128 registers)
  bf16x8 xb[kRB][16];
  {
    u32x4 xv[kRB][16];
#pragma unroll
    for (int rb = 0; rb < kRB; ++rb) {
      const int row = min(row0 + 32 * rb + li, A.M - 1);
      const u32x4* xr = reinterpret_cast<const u32x4*>(A.x + (size_t)row * kD + 8 * hf);
#pragma unroll
      for (int s = 0; s < 16; ++s) xv[rb][s] = xr[2 * s];   // all 32 requests first ...
    }
#pragma unroll
    for (int rb = 0; rb < kRB; ++rb)
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        asm volatile("" : "+v"(xv[rb][s]));                  // ... then pinned (each pin waits for its load)
        xb[rb][s] = __builtin_bit_cast(bf16x8, xv[rb][s]);
      }
  }
  x6::wait_vm<0>();
  __syncthreads();

  f32x16 acc2[kRB][8];
#pragma unroll
  for (int rb = 0; rb < kRB; ++rb)
#pragma unroll
    for (int ct = 0; ct < 8; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[rb][ct][r] = 0.f;

  // One hidden tile in PINNED program order (sched_barrier(0) between the items; left alone, hipcc reads every LDS operand right
  // in front of its product and waits for it: 64 exposed LDS round trips per tile): a ring of four weight fragments is read four
  // products ahead -- the fc2 fragments of the tile behind the last fc1 products -- and the DMA instructions of the next slice
  // issue in the shadow of the first fc1 products.
  FFN_T(t_loop);
  FFN_ADD(0, t_start, t_loop);
  for (int j = 0; j < ntiles; ++j) {
    FFN_T(t_a);
    const uint4* cur = reinterpret_cast<const uint4*>(s_w + (j & 1) * kSlice) + lane;
    const bool more = j + 1 < ntiles;
    f32x16 acc[kRB];
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
    uint4 wf[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) wf[u] = cur[u * 64];
    __builtin_amdgcn_sched_barrier(0);
    x6::static_for<16>([&](auto KS) {
      constexpr int ks = decltype(KS)::value;
      acc[0] = mfma_bf16(wf[ks & 3], xb[0][ks], ks == 0 ? zero16 : acc[0]);   // (a constant C operand: no zeroing pass)
      __builtin_amdgcn_sched_barrier(0);
#ifndef EGTR_FFN_ABL_NO_DMA
      if constexpr (ks < kPieces) {
        if (more) dma_piece(j + 1, ks);
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
      acc[1] = mfma_bf16(wf[ks & 3], xb[1][ks], ks == 0 ? zero16 : acc[1]);
      __builtin_amdgcn_sched_barrier(0);
      wf[ks & 3] = cur[(ks + 4) * 64];   // fragments 16 .. 19 are the first four of fc2
      __builtin_amdgcn_sched_barrier(0);
    });
    FFN_T(t_b);
    // bias + ReLU + rounding (relu(bf16(v)) == bf16(relu(v))): the accumulators become fc2's B operands
    bf16x8 hb[kRB][2];
#pragma unroll
    for (int rb = 0; rb < kRB; ++rb) {
      float h[16];
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const float4 bb = *reinterpret_cast<const float4*>(s_b1 + j * 32 + 8 * rq + 4 * hf);
        // v_max_f32: a NaN unit becomes 0 here, but a non-finite input row still ends non-finite through the residual x
        h[4 * rq + 0] = __builtin_fmaxf(acc[rb][4 * rq + 0] + bb.x, 0.f);
        h[4 * rq + 1] = __builtin_fmaxf(acc[rb][4 * rq + 1] + bb.y, 0.f);
        h[4 * rq + 2] = __builtin_fmaxf(acc[rb][4 * rq + 2] + bb.z, 0.f);
        h[4 * rq + 3] = __builtin_fmaxf(acc[rb][4 * rq + 3] + bb.w, 0.f);
      }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 8; ++e) hb[rb][kb][e] = (__bf16)h[8 * kb + e];   // round to nearest even
    }
    __builtin_amdgcn_sched_barrier(0);
    FFN_T(t_c);
    // fc2: the tile's contribution to all 256 output channels
    x6::static_for<16>([&](auto F) {
      constexpr int f = decltype(F)::value, ct = f >> 1, kb = f & 1;
#ifdef EGTR_FFN_ABL_NO_FC2
      if (A.M < 0)
#endif
      acc2[0][ct] = mfma_bf16(wf[f & 3], hb[0][kb], acc2[0][ct]);
      __builtin_amdgcn_sched_barrier(0);
#ifdef EGTR_FFN_ABL_NO_FC2
      if (A.M < 0)
#endif
      acc2[1][ct] = mfma_bf16(wf[f & 3], hb[1][kb], acc2[1][ct]);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (f + 4 < 16) {
        wf[f & 3] = cur[(16 + f + 4) * 64];
        __builtin_amdgcn_sched_barrier(0);
      }
    });
    FFN_T(t_d);
#ifndef EGTR_FFN_ABL_NO_BARRIER
    x6::wait_vm<0>();   // this wave's share of slice j + 1 has landed
    __syncthreads();    // ... everybody's; and slice j is free for j + 2
#endif
    FFN_T(t_e);
    FFN_ADD(1, t_a, t_b);
    FFN_ADD(2, t_b, t_c);
    FFN_ADD(3, t_c, t_d);
    FFN_ADD(4, t_d, t_e);
  }

  // ---- epilogue: lane = row li of a row block, registers = channels ct*32 + 8 q + 4 half + 0..3 ----------------------------
  // The residual x comes from the input panel still in registers: channel group (ct, q) of lane (row, half) lies in fragment
  // 2 ct + (q >> 1) of the lane whose half equals q & 1, elements 4 half .. 4 half + 3 -- the own fragment for q & 1 == half, the
  // partner lane's (one 2-dword exchange per fragment) otherwise.  The result leaves through LDS (the weight slices are dead
  // behind the last barrier; a wave owns 16 KiB = one row block): written as 8-byte channel groups, slot index XORed with the
  // row (two lanes per bank pair: the two passes 512 bytes need anyway), read back a ROW per instruction -- 64 lanes x 8 bytes
  // = the 512 contiguous bytes of a row of y, and of pos / y_pos (round 5: 8-byte pieces at a 512-byte stride, 32 lines per
  // instruction, made the epilogue half of the kernel: 165 k of 326 k cycles per workgroup).
  char* const s_out_w = s_w + wave * (kRB * 32 * 512);   // both row blocks of the wave (the dynamic LDS is 128 KiB for this)
  auto epilogue = [&](auto RB) {
    constexpr int rb = decltype(RB)::value;
    char* const s_out = s_out_w + rb * (32 * 512);
    FFN_T(e_0);
    float sum = 0.f;
#pragma unroll
    for (int sfr = 0; sfr < 16; ++sfr) {
      const u32x4 fr = __builtin_bit_cast(u32x4, xb[rb][sfr]);
      const unsigned own0 = hf ? fr[2] : fr[0], own1 = hf ? fr[3] : fr[1];
      const unsigned snd0 = hf ? fr[0] : fr[2], snd1 = hf ? fr[1] : fr[3];
      const unsigned rcv0 = (unsigned)__shfl_xor((int)snd0, 32), rcv1 = (unsigned)__shfl_xor((int)snd1, 32);
      const int ct = sfr >> 1;
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        const int q = 2 * (sfr & 1) + qq;           // q & 1 == qq
        const bool mine = qq == hf;
        const unsigned p0 = mine ? own0 : rcv0, p1 = mine ? own1 : rcv1;
        const float4 bb = *reinterpret_cast<const float4*>(s_b2 + ct * 32 + 8 * q + 4 * hf);
        // bf16 fc2 output, then the bf16 residual sum (the reference's tensors), on the hardware converter (round to nearest even)
        const unsigned f0 = pk_bf16(acc2[rb][ct][4 * q + 0] + bb.x, acc2[rb][ct][4 * q + 1] + bb.y);
        const unsigned f1 = pk_bf16(acc2[rb][ct][4 * q + 2] + bb.z, acc2[rb][ct][4 * q + 3] + bb.w);
        const unsigned s0 = pk_bf16(lo_bf(p0) + lo_bf(f0), hi_bf(p0) + hi_bf(f0));
        const unsigned s1 = pk_bf16(lo_bf(p1) + lo_bf(f1), hi_bf(p1) + hi_bf(f1));
        const float v0 = lo_bf(s0), v1 = hi_bf(s0), v2 = lo_bf(s1), v3 = hi_bf(s1);
        acc2[rb][ct][4 * q + 0] = v0;
        acc2[rb][ct][4 * q + 1] = v1;
        acc2[rb][ct][4 * q + 2] = v2;
        acc2[rb][ct][4 * q + 3] = v3;
        sum += (v0 + v1) + (v2 + v3);
      }
    }
    FFN_T(e_a);
    sum += __shfl_xor(sum, 32);
    const float mean = sum * (1.f / kD);
    float sq = 0.f;
#pragma unroll
    for (int ct = 0; ct < 8; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d = acc2[rb][ct][r] - mean;
        acc2[rb][ct][r] = d;
        sq += d * d;
      }
    sq += __shfl_xor(sq, 32);
    const float rstd = rsqrtf(sq * (1.f / kD) + A.eps);
#pragma unroll
    for (int ct = 0; ct < 8; ++ct)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = ct * 32 + 8 * q + 4 * hf;
        const float4 g4 = *reinterpret_cast<const float4*>(s_g + c), e4 = *reinterpret_cast<const float4*>(s_be + c);
        const unsigned o01 = pk_bf16(acc2[rb][ct][4 * q + 0] * rstd * g4.x + e4.x, acc2[rb][ct][4 * q + 1] * rstd * g4.y + e4.y);
        const unsigned o23 = pk_bf16(acc2[rb][ct][4 * q + 2] * rstd * g4.z + e4.z, acc2[rb][ct][4 * q + 3] * rstd * g4.w + e4.w);
        const int slot = (ct * 8 + 2 * q + hf) ^ li;
        *reinterpret_cast<uint2*>(s_out + li * 512 + slot * 8) = make_uint2(o01, o23);
      }
    FFN_T(e_b);
    FFN_ADD(8, e_0, e_a);
    FFN_ADD(9, e_a, e_b);
  };
  // rows out: lane = 8-byte channel group `lane` of row r.  The position rows of all 64 rows are requested BEFORE the wave issues
  // its first store (loads and stores share one counter on this part and the compiler cannot assume they complete in order: a
  // load issued behind a store was waited for with vmcnt(0), i.e. until every earlier store had COMPLETED -- the rows left one
  // DRAM write round trip at a time, 72 k cycles per workgroup); LDS reads eight rows at a time.
  auto rows_out = [&](auto FULL) {
    constexpr bool full = decltype(FULL)::value;   // all 64 rows of the wave exist: no per-row branch, LDS reads in batches
    const bool with_pos = A.y_pos != nullptr;
    uint2 pv[kRB * 32];
    if (with_pos) {
      // (one division per wave: the rows are consecutive, the position row advances with them and wraps)
      const int rc0 = min(row0, A.M - 1);                 // rows behind M - 1 repeat the last valid one
      const int p0 = rc0 % A.pos_rows;
      const unsigned short* pl = A.pos + 4 * lane;
#pragma unroll
      for (int u = 0; u < kRB * 32; ++u) {
        int p = p0 + (full ? u : min(row0 + u, A.M - 1) - rc0);
        if (A.pos_rows >= kRB * 32) p -= p >= A.pos_rows ? A.pos_rows : 0;
        else p %= A.pos_rows;
        pv[u] = *reinterpret_cast<const uint2*>(pl + (size_t)p * kD);
      }
    }
    unsigned short* const yl = A.y + (size_t)row0 * kD + 4 * lane;
    unsigned short* const pl_out = with_pos ? A.y_pos + (size_t)row0 * kD + 4 * lane : nullptr;
    // the y + pos rows FIRST: their wait for the position rows is a wait for loads only (no store of this wave is in flight yet)
#pragma unroll
    for (int r0 = 0; r0 < kRB * 32; r0 += 8) {
      uint2 yv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) yv[u] = *reinterpret_cast<const uint2*>(s_out_w + (r0 + u) * 512 + ((lane ^ ((r0 + u) & 31)) & 63) * 8);
      if (with_pos) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (full || row0 + r0 + u < A.M)
            *reinterpret_cast<uint2*>(pl_out + (size_t)(r0 + u) * kD) =
                make_uint2(pk_bf16(lo_bf(yv[u].x) + lo_bf(pv[r0 + u].x), hi_bf(yv[u].x) + hi_bf(pv[r0 + u].x)),
                           pk_bf16(lo_bf(yv[u].y) + lo_bf(pv[r0 + u].y), hi_bf(yv[u].y) + hi_bf(pv[r0 + u].y)));
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (full || row0 + r0 + u < A.M) *reinterpret_cast<uint2*>(yl + (size_t)(r0 + u) * kD) = yv[u];
    }
  };
  FFN_T(t_epi);
#ifdef EGTR_FFN_ABL_NO_EPI
  if (A.M < 0)
#endif
  {
    epilogue(std::integral_constant<int, 0>{});
    if constexpr (kRB > 1) epilogue(std::integral_constant<int, 1>{});
    FFN_T(e_r);
    if (row0 + kRB * 32 <= A.M) rows_out(std::true_type{});
    else rows_out(std::false_type{});
    FFN_T(e_s);
    FFN_ADD(10, e_r, e_s);
  }
#ifdef EGTR_FFN_TIMING
  __builtin_amdgcn_s_waitcnt(0);
  FFN_T(t_end);
  FFN_ADD(5, t_epi, t_end);
  FFN_ADD(6, t_start, t_end);
  if (tid == 0) {
    for (int i = 0; i < 7; ++i) atomicAdd(&g_ffn_stamps[i], t_sum[i]);
    for (int i = 8; i < 12; ++i) atomicAdd(&g_ffn_stamps[i], t_sum[i]);
    atomicAdd(&g_ffn_stamps[7], 1ull);
  }
#endif
}

}  // namespace

#ifdef EGTR_FFN_TIMING
extern "C" int egtr_ffn_bf16_stamps(unsigned long long* host_out, int reset) {
  if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_ffn_stamps), sizeof(unsigned long long) * 12) != hipSuccess) return EGTR_E_LAUNCH;
  if (reset) {
    unsigned long long z[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_ffn_stamps), z, sizeof(z)) != hipSuccess) return EGTR_E_LAUNCH;
  }
  return EGTR_OK;
}
#endif

extern "C" long long egtr_ffn_packed_weights_bytes(int ffn_dim) { return ffn_dim > 0 ? (long long)(ffn_dim >> 5) * kSlice : 0; }

extern "C" int egtr_ffn_pack_weights_bf16(egtr_stream_t stream, const uint16_t* w1, const uint16_t* w2, int d_model,
                                          int ffn_dim, uint16_t* packed) {
  if (!w1 || !w2 || !packed) return EGTR_E_ARG;
  if (d_model != kD || ffn_dim <= 0 || ffn_dim > 1024 || (ffn_dim & 31)) return EGTR_E_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(w1) | reinterpret_cast<uintptr_t>(w2) | reinterpret_cast<uintptr_t>(packed)) & 15)
    return EGTR_E_UNSUPPORTED;
  const int n = (ffn_dim >> 5) * 2048;
  hipLaunchKernelGGL(ffn_pack_weights, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), w1, w2, ffn_dim,
                     reinterpret_cast<uint4*>(packed));
  return egtr_check_launch();
}

extern "C" int egtr_ffn_layernorm_bf16(egtr_stream_t stream, const uint16_t* x, const uint16_t* w_packed, const uint16_t* b1,
                                       const uint16_t* b2, const uint16_t* gamma, const uint16_t* beta, float eps,
                                       const uint16_t* pos, int pos_rows, uint16_t* y, uint16_t* y_pos, int M, int d_model,
                                       int ffn_dim) {
  if (!x || !w_packed || !b1 || !b2 || !gamma || !beta || !y) return EGTR_E_ARG;
  if (M <= 0 || (pos != nullptr) != (y_pos != nullptr)) return EGTR_E_ARG;
  if (pos != nullptr && (pos_rows <= 0 || M % pos_rows != 0)) return EGTR_E_ARG;
  if (d_model != kD || ffn_dim <= 0 || ffn_dim > 1024 || (ffn_dim & 31)) return EGTR_E_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w_packed) | reinterpret_cast<uintptr_t>(y) |
       reinterpret_cast<uintptr_t>(pos) | reinterpret_cast<uintptr_t>(y_pos)) & 15)
    return EGTR_E_UNSUPPORTED;
  FfnArgs A;
  A.x = x; A.wpk = reinterpret_cast<const uint4*>(w_packed); A.b1 = b1; A.b2 = b2; A.gamma = gamma; A.beta = beta;
  A.pos = pos; A.y = y; A.y_pos = y_pos; A.M = M; A.F = ffn_dim; A.pos_rows = pos != nullptr ? pos_rows : 1; A.eps = eps;
  static unsigned long long raised = 0;
  if (int e = egtr_raise_dynamic_lds(reinterpret_cast<const void*>(&ffn_bf16_kernel), kLds, &raised)) return e;
  const dim3 grid((unsigned)((M + kRowsWg - 1) / kRowsWg));
  hipLaunchKernelGGL(ffn_bf16_kernel, grid, dim3(64 * kWaves), kLds, static_cast<hipStream_t>(stream), A);
  return egtr_check_launch();
}
