// STUDY CODE (round 3) -- not part of libegtr_hip.so: the stand-alone XS x XS split-bf16 GEMM with an LDS-DMA
// ring that measured what a better main loop buys the token-sized products (nothing: 0.31 -> 0.31-0.39 of the bf16 peak).
// Built only by tools/gemm_x6_bench.hip's command line together with egtr_amd/csrc/xs_split.hip and capi.hip.
// Token-sized fp32 linear layers on the bf16 matrix cores, second generation:  C[M, N] = act(A[M, K] . W[N, K]^T + bias)
// with fp32-level accuracy from exact three-way bf16 splits of both operands and the six leading cross terms (the
// arithmetic of gemm_split.hip / rel_head.hip: every bf16 x bf16 product is exact in fp32, fp32 accumulation, the three
// dropped terms are <= 2^-24 of a product).  Replaces the vendor fp32 GEMMs behind the encoder's nn.Linear layers
// (reference: model/deformable_detr.py:1049-1058, 1102, 1337-1343).
//
// What changed against gemm_split.hip (round 2: 0.31 of the bf16 matrix peak; matrix pipe busy 32 %, 29 % LDS bank
// conflicts from the 8-byte piece stores, every wave of a workgroup in the same phase between two barriers):
//   * BOTH operands arrive PRE-SPLIT in the XS format (xs_format.h): the activations are split by the kernel that
//     produces them (LayerNorm, the MSDA epilogue, this kernel's own epilogue for FFN layer 1 -> layer 2), the weights
//     once per model.  The main loop has no VALU work, no ds_write and no register staging at all: a stage is filled by
//     global_load_lds_dwordx4 (LDS-DMA, one 1 KiB fragment per wave-instruction, lane-linear = conflict-free by
//     construction) and drained by ds_read_b128 straight into MFMA operands.
//   * ONE barrier per K step of 16.  A ring of three LDS stages is filled two steps ahead of the reads (counted
//     `s_waitcnt vmcnt`, never 0 in the steady state; raw s_barrier -- __syncthreads() would drain the DMA queue), and
//     the operand fragments of step s + 1 are read into a second register set while the 24 MFMAs of step s issue.
//   * every operand fragment feeds 2 x 3 (m or n tiles x cross terms) MFMAs: 12 ds_read_b128 per 24 MFMAs per wave.
// Workgroup = 4 waves (2 x 2) on a 128 x 128 tile, each wave 64 x 64 = 2 x 2 MFMA tiles of 32 x 32; 72 KiB of LDS and
// <= 256 registers: two workgroups per CU, i.e. two waves per SIMD from DIFFERENT workgroups (out of phase: one computes
// while the other waits at its barrier).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>
#include <type_traits>

#include "../egtr_amd/csrc/common.h"
#include "../egtr_amd/csrc/xs_format.h"
#include "../egtr_amd/csrc/x6_common.h"

#ifndef X6_ABL
#define X6_ABL 0   // development ablations (tools/gemm_x6_bench.hip): 1 = no steady-state DMA, 2 = no MFMAs, 3 = no barriers
#endif

namespace {

using namespace x6;

constexpr int kMaxProblems = 8;
constexpr int kStages = 3;

struct X6Problem {
  const char* A;      // XS(A): [ceil(M/32)][K/16][3][1 KiB]
  const char* W;      // XS(W): [N/32][K/16][3][1 KiB]
  const float* bias;  // [N] or null
  float* C;           // fp32 [M, ldc] or null
  char* Cxs;          // XS(C) (as the A operand of a following product with K' = N) or null
  int ldc, N, relu;
};
struct X6Problems {
  X6Problem p[kMaxProblems];
};

__device__ __forceinline__ int xcd_tile(int bid, int total) {   // every XCD owns a contiguous range of the tile order
  const int q = total >> 3, r = total & 7, x = bid & 7;
  return x * q + min(x, r) + (bid >> 3);
}

// MT = 32-row MFMA tiles per wave along m (2: workgroup tile 128 x 128).
template <int MT>
__global__ __launch_bounds__(256, 2) void gemm_x6_kernel(X6Problems P, int nprob, int M, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ABLK = 2 * MT;              // 32-row blocks of A per workgroup tile
  constexpr int NFRAG = (ABLK + 4) * 3;     // fragments per stage (A blocks + 4 W blocks, 3 pieces each)
  constexpr int NL = (NFRAG + 3) / 4;       // DMA instructions per wave and stage: the same for every wave, so that the
  constexpr int NSLOT = 4 * NL;             // counted vmcnt waits are immediates; slots >= NFRAG (MT = 1: two of twenty)
  constexpr int STAGE = NSLOT * xs::kFragBytes;   // re-load fragments 0, 1 into a tail of the stage nobody reads
  constexpr int BM = 64 * MT;

  const int KS = K >> 4;
  const int mblocks = (M + BM - 1) / BM;
  int tile = xcd_tile(blockIdx.x, gridDim.x), pi = 0;
  for (; pi + 1 < nprob; ++pi) {
    const int t = (P.p[pi].N >> 7) * mblocks;
    if (tile < t) break;
    tile -= t;
  }
  const X6Problem& G = P.p[pi];
  const int nblocks = G.N >> 7;
  const int nb = tile % nblocks, mb = tile / nblocks;   // the n blocks of one m block are neighbours in the tile order
  const int m0 = mb * BM, n0 = nb * 128;
  const int RB = (M + 31) >> 5;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const unsigned lds_base = (unsigned)reinterpret_cast<uintptr_t>((lds_char*)smem);

  // ---- DMA plan of this wave: fragments f = wave * NL .. + NL - 1 of the stage image [A blocks][W blocks] x [3 pieces]
  const char* gsrc[NL];
  unsigned ldst[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int f = wave * NL + i, fs = f < NFRAG ? f : f - NFRAG, blk = fs / 3, p = fs - 3 * blk;
    const char* base;
    if (blk < ABLK) {
      const int rb = min((m0 >> 5) + blk, RB - 1);   // past the matrix: re-read the last block (its outputs are not stored)
      base = G.A + (size_t)rb * KS * (3 * xs::kFragBytes);
    } else {
      base = G.W + (size_t)((n0 >> 5) + blk - ABLK) * KS * (3 * xs::kFragBytes);
    }
    gsrc[i] = base + p * xs::kFragBytes + lane * 16;
    ldst[i] = lds_base + f * xs::kFragBytes;
  }
  auto issue = [&](int ks, int slot) {
    const size_t goff = (size_t)ks * (3 * xs::kFragBytes);
    const unsigned loff = (unsigned)slot * STAGE;
#pragma unroll
    for (int i = 0; i < NL; ++i) dma16(gsrc[i] + goff, ldst[i] + loff);
  };

  // ---- operand reads: fragment (block, piece) of the stage image at lane * 16
  const char* la = smem + (wm * MT * 3) * xs::kFragBytes + lane * 16;
  const char* lw = smem + ((ABLK + wn * 2) * 3) * xs::kFragBytes + lane * 16;
  auto lds_read = [&](bf16x8 (&a)[MT][3], bf16x8 (&w)[2][3], int slot) {
    const int off = slot * STAGE;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int t = 0; t < MT; ++t) a[t][p] = *reinterpret_cast<const bf16x8*>(la + off + (t * 3 + p) * xs::kFragBytes);
#pragma unroll
      for (int t = 0; t < 2; ++t) w[t][p] = *reinterpret_cast<const bf16x8*>(lw + off + (t * 3 + p) * xs::kFragBytes);
    }
  };

  f32x16 acc[MT][2];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // MFMA roles: A operand = weight piece (i = n), B operand = activation piece (j = m): a lane ends up with one row m and
  // 4 CONSECUTIVE columns n per accumulator quad (float4 / 8-byte piece stores in the epilogue).  Small terms first.
  //
  // One K step, in PINNED program order (sched_barrier(0) between the items: left alone, hipcc sinks the operand reads of
  // the next step behind the MFMAs of this one and re-uses their registers, which serialises read -> wait -> MFMA):
  //   wait (own DMAs of stage s + 1) . lgkmcnt(0) . barrier .
  //   24 MFMAs on the registers of stage s, term-major over the four accumulators (consecutive MFMAs are independent),
  //   with the NL DMA instructions of stage s + 3 behind MFMAs 1, 3, 5, .. and the 4 + 2 MT operand reads of stage s + 1
  //   (into the OTHER register set) two by two behind MFMAs 12 .. 17 -- every non-matrix instruction issues in the
  //   shadow of a 32-cycle MFMA.
  constexpr int NMMA = MT * 2 * 6;
  constexpr int NRD = 3 * (MT + 2);
  auto step = [&](const bf16x8 (&a)[MT][3], const bf16x8 (&w)[2][3], bf16x8 (&an)[MT][3], bf16x8 (&wn_)[2][3], int ks_fill,
                  int slot_fill, int slot_read, auto do_fill, auto do_read) {
    const size_t goff = (size_t)ks_fill * (3 * xs::kFragBytes);
    const unsigned loff = (unsigned)slot_fill * STAGE;
    const int roff = slot_read * STAGE;
    static_for<NMMA>([&](auto I) {
      constexpr int i = decltype(I)::value;
      constexpr int term = i / (MT * 2), t = i % (MT * 2), mt = t / 2, nt = t % 2;
      constexpr int pw = term == 0 ? 2 : (term == 2 || term == 3) ? 1 : 0;
      constexpr int pa = term == 1 ? 2 : (term == 2 || term == 4) ? 1 : 0;
#if X6_ABL != 2 && X6_ABL != 7 && X6_ABL != 8
      acc[mt][nt] = mfma(w[nt][pw], a[mt][pa], acc[mt][nt]);
#else
      asm volatile("" ::"v"(w[nt][pw]), "v"(a[mt][pa]));
#endif
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (X6_ABL != 1 && X6_ABL != 4 && decltype(do_fill)::value && (i & 1) && (i >> 1) < NL) {
        dma16(gsrc[i >> 1] + goff, ldst[i >> 1] + loff);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (X6_ABL != 4 && X6_ABL != 5 && X6_ABL != 8 && decltype(do_read)::value && i >= NMMA / 2 && 2 * (i - NMMA / 2) < NRD) {
#pragma unroll
        for (int j = 2 * (i - NMMA / 2); j < 2 * (i - NMMA / 2) + 2 && j < NRD; ++j) {
          // read order: a[0][p], .., a[MT-1][p], w[0][p], w[1][p] for p = 0, 1, 2
          const int p = j / (MT + 2), u = j % (MT + 2);
          if (u < MT) an[u][p] = *reinterpret_cast<const bf16x8*>(la + roff + (u * 3 + p) * xs::kFragBytes);
          else wn_[u - MT][p] = *reinterpret_cast<const bf16x8*>(lw + roff + ((u - MT) * 3 + p) * xs::kFragBytes);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    });
  };

  bf16x8 a0[MT][3], w0[2][3], a1[MT][3], w1[2][3];
  // Stage s lives in ring slot s % 3; the ring is filled two steps ahead of the reads, the reads one step ahead of the
  // MFMAs.  At the top of step s: this wave's DMAs of stage s + 1 have landed (counted vmcnt: the NL of stage s + 2 stay in
  // flight), its reads of stage s have returned (lgkmcnt(0)); after the barrier that holds for every wave, so the slot of
  // stage s may be refilled (stage s + 3) and stage s + 1 may be read.
  int rd = 0, wr = 2;   // slot to read next (stage s + 1), slot to fill next (stage s + 3)
  auto advance = [&](int& v) { v = (v == kStages - 1) ? 0 : v + 1; };
  auto top = [&](bool more_in_flight) {
    if (X6_ABL == 1 || X6_ABL == 4) wait_vm<0>(); else if (more_in_flight) wait_vm<NL>(); else wait_vm<0>();
    wait_lgkm0();
    if (X6_ABL != 3 && X6_ABL != 4) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  issue(0, 0);
  issue(1, 1);
  top(true);
  issue(2, 2);
  advance(wr);
  lds_read(a0, w0, 0);
  rd = 1;
  int s = 0;
  using T = std::true_type;
  using F = std::false_type;
#pragma unroll 1
  for (; s + 4 < KS; s += 2) {        // steady state: both halves fill (s + 3, s + 4 < KS)
    top(true);
    step(a0, w0, a1, w1, s + 3, wr, rd, T{}, T{});
    advance(wr); advance(rd);
    top(true);
    step(a1, w1, a0, w0, s + 4, wr, rd, T{}, T{});
    advance(wr); advance(rd);
  }
  // tail: KS even, so s == KS - 4 (two more fills: no -- stage s + 3 = KS - 1 only) or s == KS - 2
  if (s + 4 == KS) {
    top(true);
    step(a0, w0, a1, w1, s + 3, wr, rd, T{}, T{});   // fills the last stage
    advance(wr); advance(rd);
    top(true);
    step(a1, w1, a0, w0, 0, 0, rd, F{}, T{});
    advance(rd);
    s += 2;
  }
  top(false);                                        // stage KS - 1 is the only one in flight
  step(a0, w0, a1, w1, 0, 0, rd, F{}, T{});
  step(a1, w1, a0, w0, 0, 0, 0, F{}, F{});

  // ---- epilogue.  D[i = n][j = m]: accumulator r <-> n = (r & 3) + 8 (r >> 2) + 4 hf of the 32-wide n tile, m = lane & 31
  const int li = lane & 31, hf = lane >> 5;
  const float* __restrict__ bias = G.bias;
  float* __restrict__ C = G.C;
  char* __restrict__ Cxs = G.Cxs;
  const bool relu = G.relu != 0;
  const int KSo = G.N >> 4, ldc = G.ldc;
  auto epilogue = [&](auto has_c, auto has_xs) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row = m0 + (wm * MT + mt) * 32 + li;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = n0 + (wn * 2 + nt) * 32 + 8 * q + 4 * hf;
          float4 v = make_float4(acc[mt][nt][4 * q + 0], acc[mt][nt][4 * q + 1], acc[mt][nt][4 * q + 2],
                                 acc[mt][nt][4 * q + 3]);
          if (bias != nullptr) {
            const float4 bv = *reinterpret_cast<const float4*>(bias + col);
            v = make_float4(v.x + bv.x, v.y + bv.y, v.z + bv.z, v.w + bv.w);
          }
          // ReLU as a select: NaN stays NaN like torch.relu (fmaxf would turn it into 0)
          v = make_float4(relu && v.x < 0.f ? 0.f : v.x, relu && v.y < 0.f ? 0.f : v.y, relu && v.z < 0.f ? 0.f : v.z,
                          relu && v.w < 0.f ? 0.f : v.w);
          if (decltype(has_c)::value && row < M) *reinterpret_cast<float4*>(C + (size_t)row * ldc + col) = v;
          if (decltype(has_xs)::value && (row >> 5) < RB)
            xs::store4(Cxs, xs::group_offset(row, col, KSo), v.x, v.y, v.z, v.w);
        }
    }
  };
  if (X6_ABL >= 6 && acc[0][0][0] != 12345.f) return;   // no epilogue
  if (C != nullptr && Cxs != nullptr) epilogue(std::true_type{}, std::true_type{});
  else if (C != nullptr) epilogue(std::true_type{}, std::false_type{});
  else epilogue(std::false_type{}, std::true_type{});
}

int launch_x6(hipStream_t st, const X6Problems& P, int nprob, int M, int K, int force_mt) {
  long long tiles128 = 0, tiles64 = 0;
  for (int i = 0; i < nprob; ++i) {
    tiles128 += (long long)(P.p[i].N >> 7) * ((M + 127) / 128);
    tiles64 += (long long)(P.p[i].N >> 7) * ((M + 63) / 64);
  }
  if (tiles128 <= 0 || tiles64 >= (1ll << 31)) return EGTR_E_UNSUPPORTED;
  constexpr int kLds2 = kStages * 24 * xs::kFragBytes, kLds1 = kStages * 20 * xs::kFragBytes;
  static bool attr_set = false;   // idempotent; a race would only set the same values twice
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x6_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kLds2) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x6_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kLds1) != hipSuccess)
      return egtr_check_launch();
    attr_set = true;
  }
  // 128-row tiles when they give every CU its two workgroups, 64-row tiles otherwise (a lone workgroup per CU = one wave
  // per SIMD exposes every barrier and DMA wait; measured in tools/gemm_x6_bench.hip)
  const bool big = force_mt == 2 || (force_mt != 1 && tiles128 >= 512);
  if (big) hipLaunchKernelGGL(gemm_x6_kernel<2>, dim3((unsigned)tiles128), dim3(256), kLds2, st, P, nprob, M, K);
  else hipLaunchKernelGGL(gemm_x6_kernel<1>, dim3((unsigned)tiles64), dim3(256), kLds1, st, P, nprob, M, K);
  return egtr_check_launch();
}

}  // namespace

extern "C" int egtr_gemm_x6_f32(egtr_stream_t stream, int num_problems, const void* const* a_xs, const void* const* w_xs,
                                const float* const* bias, float* const* c, const int* ldc, void* const* c_xs,
                                const int* N, const int* relu, int M, int K) {
  if (!a_xs || !w_xs || !bias || !c || !ldc || !c_xs || !N || !relu) return EGTR_E_ARG;
  if (num_problems <= 0 || num_problems > kMaxProblems || M <= 0 || K <= 0) return EGTR_E_ARG;
  if (K % 32 || K < 64) return EGTR_E_UNSUPPORTED;
  X6Problems P = {};
  for (int i = 0; i < num_problems; ++i) {
    if (!a_xs[i] || !w_xs[i] || (!c[i] && !c_xs[i]) || N[i] <= 0) return EGTR_E_ARG;
    if (N[i] % 128 || (c[i] && (ldc[i] < N[i] || (ldc[i] & 3) || (reinterpret_cast<uintptr_t>(c[i]) & 15))) ||
        (bias[i] && (reinterpret_cast<uintptr_t>(bias[i]) & 15)) || (reinterpret_cast<uintptr_t>(a_xs[i]) & 15) ||
        (reinterpret_cast<uintptr_t>(w_xs[i]) & 15) || (reinterpret_cast<uintptr_t>(c_xs[i]) & 15))
      return EGTR_E_UNSUPPORTED;
    P.p[i] = X6Problem{static_cast<const char*>(a_xs[i]), static_cast<const char*>(w_xs[i]), bias[i], c[i],
                       static_cast<char*>(c_xs[i]), ldc[i], N[i], relu[i]};
  }
  const char* f = getenv("EGTR_X6_TILE");   // development: 1 / 2 forces 64- / 128-row tiles
  return launch_x6(static_cast<hipStream_t>(stream), P, num_problems, M, K, f ? atoi(f) : 0);
}
