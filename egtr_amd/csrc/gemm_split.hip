// Token-sized fp32 linear layers on the bf16 matrix cores:  C[M, N] = act(A[M, K] . W[N, K]^T + bias)  with fp32 operands
// in, fp32 out and fp32-level accuracy, from exact three-way bf16 splits of both operands (see rel_head.hip,
// rel_head_fwd_x6, for the argument: x = hi + mid + lo exactly, the six leading cross terms kept, every bf16 x bf16
// product exact in fp32, fp32 accumulation; the three dropped terms are <= 2^-24 of the product).
//
// Why: the encoder applies five nn.Linear layers per layer to S = 12 537 token rows (value / output projections,
// sampling offsets + attention weights, the two FFN layers): 1.05 of the 4.1 ms forward, running at 100-125 TFLOP/s in
// hipBLASLt -- 65-80 % of the fp32 matrix peak of this part, which is its fp32 VECTOR rate (157 TFLOP/s).  The bf16
// matrix rate is 16x that; six bf16 MFMAs per K = 16 replace eight fp32 MFMAs of twice the length: 2.67x less matrix
// time for the same result to fp32 rounding.
//
// Tiling: a workgroup of 8 waves owns a 128 x 128 (or 64 x 128) tile of C, each wave 32 x 64 (32 x 32) of it on
// v_mfma_f32_32x32x16_bf16.  K runs in stages of 32: the fp32 A tile is split into its three pieces on the way into
// LDS; W arrives pre-split and pre-tiled from the host (one contiguous 24 KiB block per stage:
// wt[N/128][K/32][3 pieces][128 rows][32] bf16, egtr_amd/ops.py::gemm_split_weights).  The global loads of stage s + 1
// are in flight (inline asm, counted) while stage s is multiplied out of LDS; rows are 80 bytes apart in LDS, which
// makes the 16-byte operand reads of 32 consecutive rows conflict-free.  60 (45) KiB of LDS and <= 128 registers per
// lane: two workgroups = sixteen waves per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "common.h"
#include "xs_format.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kBN = 128, kBK = 32;
constexpr int kPitch = 40;  // bf16 elements per LDS row: 32 + 8 (80 bytes)

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4v gload(const void* p) {
  f32x4v v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v) : "v"(p));
  return v;
}
__device__ __forceinline__ void vm_wait0(f32x4v& v) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(v)); }

// exact truncation split (rel_head.hip): pieces as fp32 bit patterns with zero low halves.  xs::split3_fast (round 4; the
// inf-safe xs::split3 until then): 4 VALU operations per element instead of 10 -- these kernels split every activation
// element N / 128 times over, and the weight-gradient kernel both operands: 3-4.5 % off the forward / data-gradient products
// and 8-9 % off the weight gradients at the training size (abl/gemm_fast.sh, two alternations on one box).  For a non-finite
// element the lower pieces come out NaN (inf - inf), so ITS output row (weight gradient: its column) is NaN instead of a mix of
// +-inf and NaN -- non-finite either way and nothing else is touched (tests/test_gpu_pinning.py).
using Split3 = xs::Split3;
__device__ __forceinline__ Split3 split3(float x) { return xs::split3_fast(x); }
__device__ __forceinline__ unsigned pack_hi16(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// A workgroup is 8 waves on a BM x 128 tile, BM = 128 (wave tile 32 x 64) or 64 (wave tile 32 x 32; chosen when 128-row
// tiles would not give every CU two workgroups).  <= 128 registers per lane: four waves per SIMD, so that the load /
// split / store phases of one wave run under the matrix work of the others -- with one or two waves per SIMD every
// per-stage latency (operand reads, barrier, vmcnt, LDS stores) was exposed and the kernel ran at the vendor fp32 rate.
// MFMA roles: A operand = weight piece (i = n), B operand = activation piece (j = m), so that a lane ends up with 4
// CONSECUTIVE output columns per accumulator quad: the epilogue stores float4.
// Up to kMaxProblems independent products with the same M and K in ONE launch (the value projection and the offsets /
// attention-weights projection of an encoder layer; the six value projections of the decoder): their tiles share the
// grid, so that products which do not fill the chip on their own (196-588 tiles on 512 workgroup slots) fill it together.
// Workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8), each with its own L2.  Logical tile t = xcd_tile(...)
// hands every XCD a CONTIGUOUS range of tiles, so that the tiles running side by side on one XCD are neighbours in the
// tile order below and find the operand rows they share in that XCD's L2 (instead of every n block of a row tile
// fetching the activation rows again through another L2).
__device__ __forceinline__ int xcd_tile(int bid, int total) {
  const int q = total >> 3, r = total & 7, x = bid & 7;
  return x * q + min(x, r) + (bid >> 3);
}

constexpr int kMaxProblems = 16;   // (the relation head's 14 slot products share one grid)
struct GemmProblem {
  const float* A;
  const unsigned short* Wt;
  const float* bias;
  float* C;
  int lda, ldc, N, relu;
  const float* pos;   // optional [pos_rows, K] added to the rows of A on their way into LDS (row % pos_rows): the encoder's
  int pos_rows;       // `hidden_states + position_embeddings` (dd:1041) without a materialised sum
  // epilogue extensions of the training step (egtr_linear_split_bf16_ex_f32), applied in this order; all optional:
  const unsigned char* row_keep;   // [M] bytes: rows with 0 produce zeros (value rows of padded tokens, dd:1052, and their
                                   // gradients in the backward)
  const float* relu_ref;           // [M, ldref]: v = relu_ref > 0 ? v : 0 -- the ReLU backward of the layer whose OUTPUT
  int ldref;                       // relu_ref is, applied to this data-gradient product
  const float* add1;               // [M, ldadd] addends (gradient accumulation of the branches that meet at a tensor);
  const float* add2;               // add1 / add2 may alias C (an element is read and written by the same lane)
  int ldadd;
  float* colpart;                  // [ceil(M / 32), N]: column sums of the stored values over each 32-row block (the bias
                                   // gradient of the consuming layer: summed in a fixed order by a second tiny launch)
};
struct GemmProblems {
  GemmProblem p[kMaxProblems];
};

// (A double-buffered form of this loop -- two LDS stage buffers, one barrier per stage, 120 KiB of LDS = one workgroup per CU --
// and a ping-pong schedule on it were measured 7-15 % slower in round 5 and removed in round 6: DESIGN.md 4.12,
// profiles/r05_gemm_double_buffer_ab.txt; the code is in the history, commit 46d9c2f and before.)
template <int BM>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_split_bf16_f32(
    GemmProblems P, int nprob, int M, int K) {
  // tile -> (problem, n block, m block): n blocks of one m block are neighbours (they read the same activation rows)
  const int mblocks = (M + BM - 1) / BM;
  int tile = xcd_tile(blockIdx.x, gridDim.x), pi = 0;
  for (; pi + 1 < nprob; ++pi) {
    const int t = (P.p[pi].N / kBN) * mblocks;
    if (tile < t) break;
    tile -= t;
  }
  const GemmProblem& G = P.p[pi];
  const int nblocks = G.N / kBN;
  const int nb = tile % nblocks, m0 = (tile / nblocks) * BM;
  const float* __restrict__ A = G.A;
  const unsigned short* __restrict__ Wt = G.Wt;
  const float* __restrict__ bias = G.bias;
  float* __restrict__ C = G.C;
  const int lda = G.lda, ldc = G.ldc;
  const bool RELU = G.relu != 0;
  constexpr int WAVES_M = BM / 32;            // 4 or 2
  constexpr int WAVES_N = 8 / WAVES_M;        // 2 or 4
  constexpr int NT = 4 / WAVES_N;             // 32-column MFMA tiles per wave: 2 or 1
  constexpr int AQ = BM / 64;                 // float4 of A per thread and stage (BM rows x 8 float4 / 512 threads)
  constexpr int kStageElems = 3 * BM * kPitch + 3 * kBN * kPitch;
  __shared__ __attribute__((aligned(16))) __bf16 sbase[kStageElems];
  __bf16* sA = sbase;
  __bf16* sW = sbase + 3 * BM * kPitch;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int li = lane & 31, hf = lane >> 5;
  const int nk = K / kBK;

  // global -> register mapping of one stage
  //   A: BM x 8 float4, thread t takes idx = t + 512 q: row = idx >> 3, c4 = idx & 7
  //   W: 1536 16-byte chunks of the contiguous stage block, thread t takes idx = t + 512 q
  const float* arow[AQ];
#pragma unroll
  for (int q = 0; q < AQ; ++q) {
    const int idx = tid + 512 * q;
    const int row = min(m0 + (idx >> 3), M - 1);
    arow[q] = A + (size_t)row * lda + 4 * (idx & 7);
  }
  const char* wblk = reinterpret_cast<const char*>(Wt) + (size_t)nb * nk * (3 * kBN * kBK * 2) + (size_t)tid * 16;
  const bool has_pos = G.pos != nullptr;   // uniform per workgroup
  const float* prow[AQ];
#pragma unroll
  for (int q = 0; q < AQ; ++q) {
    const int idx = tid + 512 * q;
    const int row = min(m0 + (idx >> 3), M - 1);
    prow[q] = has_pos ? G.pos + (size_t)(row % G.pos_rows) * K + 4 * (idx & 7) : arow[q];
  }

  f32x4v ra[AQ], rp[AQ], rw[3];
  auto issue = [&](int s) {
#pragma unroll
    for (int q = 0; q < AQ; ++q) ra[q] = gload(arow[q] + s * kBK);
    if (has_pos) {
#pragma unroll
      for (int q = 0; q < AQ; ++q) rp[q] = gload(prow[q] + s * kBK);
    }
    const char* wp = wblk + (size_t)s * (3 * kBN * kBK * 2);
#pragma unroll
    for (int q = 0; q < 3; ++q) rw[q] = gload(wp + q * 8192);
  };
  auto stash = [&]() {
    __bf16* sA = sbase;
    __bf16* sW = sA + 3 * BM * kPitch;
#pragma unroll
    for (int q = 0; q < AQ; ++q) vm_wait0(ra[q]);
#pragma unroll
    for (int q = 0; q < 3; ++q) vm_wait0(rw[q]);
    if (has_pos) {
#pragma unroll
      for (int q = 0; q < AQ; ++q) {
        vm_wait0(rp[q]);
        ra[q] += rp[q];
      }
    }
#pragma unroll
    for (int q = 0; q < AQ; ++q) {
      const int idx = tid + 512 * q;
      const int row = idx >> 3, c4 = idx & 7;
      const Split3 s0 = split3(ra[q].x), s1 = split3(ra[q].y), s2 = split3(ra[q].z), s3 = split3(ra[q].w);
      __bf16* p = sA + row * kPitch + 4 * c4;
      *reinterpret_cast<uint2*>(p) = make_uint2(pack_hi16(s0.hi, s1.hi), pack_hi16(s2.hi, s3.hi));
      *reinterpret_cast<uint2*>(p + BM * kPitch) = make_uint2(pack_hi16(s0.mid, s1.mid), pack_hi16(s2.mid, s3.mid));
      *reinterpret_cast<uint2*>(p + 2 * BM * kPitch) = make_uint2(pack_hi16(s0.lo, s1.lo), pack_hi16(s2.lo, s3.lo));
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int idx = tid + 512 * q;          // chunk of the stage block: [piece][row 128][4 chunks of 8 bf16]
      const int piece = idx >> 9, row = (idx >> 2) & 127, c = idx & 3;
      *reinterpret_cast<f32x4v*>(sW + (piece * kBN + row) * kPitch + 8 * c) = rw[q];
    }
  };

  f32x16 acc[NT];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

  issue(0);
  stash();
  __syncthreads();
  const __bf16* pa0 = sA + (wm * 32 + li) * kPitch + 8 * hf;
  const __bf16* pw0 = sW + (wn * (32 * NT) + li) * kPitch + 8 * hf;
  auto compute = [&]() {
    const __bf16* pa = pa0;
    const __bf16* pw = pw0;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[3], w[NT][3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        a[p] = *reinterpret_cast<const bf16x8*>(pa + p * BM * kPitch + 16 * ks);
#pragma unroll
        for (int t = 0; t < NT; ++t)
          w[t][p] = *reinterpret_cast<const bf16x8*>(pw + (p * kBN + t * 32) * kPitch + 16 * ks);
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        f32x16 c = acc[nt];
        c = mfma_bf16(w[nt][2], a[0], c);
        c = mfma_bf16(w[nt][0], a[2], c);
        c = mfma_bf16(w[nt][1], a[1], c);
        c = mfma_bf16(w[nt][1], a[0], c);
        c = mfma_bf16(w[nt][0], a[1], c);
        c = mfma_bf16(w[nt][0], a[0], c);
        acc[nt] = c;
      }
    }
  };
  // every iteration issues the loads of the NEXT stage, multiplies the current one out of LDS and stores the next one
  // (issue and stash unconditionally paired: no path leaves a load in flight); the last stage is peeled
#pragma unroll 1
  for (int s = 0; s + 1 < nk; ++s) {
    issue(s + 1);
    compute();
    __syncthreads();   // every wave has read stage s out of LDS
    stash();
    __syncthreads();
  }
  compute();

  // epilogue: D[i = n][j = m]; accumulator r <-> column n = (r & 3) + 8 (r >> 2) + 4 hf of the 32-wide n tile, row m =
  // lane & 31: one float4 (4 consecutive columns) per accumulator quad
  const int row = m0 + wm * 32 + li;
  const bool in_range = row < M;
  const bool keep = in_range && (G.row_keep == nullptr || G.row_keep[row] != 0);
  const float* __restrict__ relu_ref = G.relu_ref;
  const float* add1 = G.add1;
  const float* add2 = G.add2;
  float* __restrict__ colpart = G.colpart;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int col = nb * kBN + wn * (32 * NT) + nt * 32 + 8 * q + 4 * hf;
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (bias != nullptr) bv = *reinterpret_cast<const float4*>(bias + col);
      float4 v = make_float4(acc[nt][4 * q + 0] + bv.x, acc[nt][4 * q + 1] + bv.y, acc[nt][4 * q + 2] + bv.z,
                             acc[nt][4 * q + 3] + bv.w);
      if (RELU) v = make_float4(egtr_relu(v.x), egtr_relu(v.y), egtr_relu(v.z), egtr_relu(v.w));
      if (!keep) v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (relu_ref != nullptr && in_range) {
        const float4 t = *reinterpret_cast<const float4*>(relu_ref + (size_t)row * G.ldref + col);
        v = make_float4(t.x > 0.f ? v.x : 0.f, t.y > 0.f ? v.y : 0.f, t.z > 0.f ? v.z : 0.f, t.w > 0.f ? v.w : 0.f);
      }
      if (add1 != nullptr && in_range) {
        const float4 t = *reinterpret_cast<const float4*>(add1 + (size_t)row * G.ldadd + col);
        v = make_float4(v.x + t.x, v.y + t.y, v.z + t.z, v.w + t.w);
      }
      if (add2 != nullptr && in_range) {
        const float4 t = *reinterpret_cast<const float4*>(add2 + (size_t)row * G.ldadd + col);
        v = make_float4(v.x + t.x, v.y + t.y, v.z + t.z, v.w + t.w);
      }
      if (in_range) *reinterpret_cast<float4*>(C + (size_t)row * ldc + col) = v;
      if (colpart != nullptr) {   // uniform per workgroup: sum over the wave's 32 rows (the 32 lanes that share hf)
        float4 s = in_range ? v : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
          s.x += __shfl_xor(s.x, o);
          s.y += __shfl_xor(s.y, o);
          s.z += __shfl_xor(s.z, o);
          s.w += __shfl_xor(s.w, o);
        }
        // (a 32-row block that starts at or beyond M does not exist in the [ceil(M / 32), N] buffer)
        if (li == 0 && m0 + wm * 32 < M) *reinterpret_cast<float4*>(colpart + (size_t)((m0 >> 5) + wm) * G.N + col) = s;
      }
    }
}

}  // namespace

namespace {
int launch_grouped(hipStream_t st, const GemmProblems& P, int nprob, int M, int K) {
  long long tiles128 = 0, tiles64 = 0;
  for (int i = 0; i < nprob; ++i) {
    tiles128 += (long long)(P.p[i].N / kBN) * ((M + 127) / 128);
    tiles64 += (long long)(P.p[i].N / kBN) * ((M + 63) / 64);
  }
  if (tiles64 >= (1ll << 31)) return EGTR_E_UNSUPPORTED;
  // 128-row tiles when they fill the chip twice over (two co-resident workgroups per CU overlap each other's load /
  // store phases), 64-row tiles otherwise.  (128-row tiles for the K = 1024 product too measured 46.0 vs 53.0 us stand-alone
  // but 262.6 vs 264.5 images/s in the forward, same box, alternating builds: not taken; again with the XCD-aware tile
  // order: 259.0 vs 262.8 / 261.5 images/s.)
  // Round 4: "fill the chip twice over" relaxed to 480 of the 512 workgroup slots.  The encoder layer's grouped launch at 600x1000
  // (value 256 -> 256 + offsets / weights 256 -> 384, M = 12 537) has 490 128-row tiles = 980 64-row tiles: one nearly full round
  // of the larger tiles beats 1.91 rounds of the smaller ones -- 42.9 -> 37.2 us per launch, 281.9 -> 284.7 images/s end to end
  // (three alternations on one box, abl/infer_ab.sh).
  if (tiles128 >= 480)
    hipLaunchKernelGGL(gemm_split_bf16_f32<128>, dim3((unsigned)tiles128), dim3(512), 0, st, P, nprob, M, K);
  else
    hipLaunchKernelGGL(gemm_split_bf16_f32<64>, dim3((unsigned)tiles64), dim3(512), 0, st, P, nprob, M, K);
  return egtr_check_launch();
}
bool problem_ok(const float* x, int ldx, const uint16_t* w, const float* bias, float* y, int ldy, int K, int N) {
  return x && w && y && ldx >= K && ldy >= N && N > 0 && N % kBN == 0 && !(ldx & 3) && !(ldy & 3) &&
         !(reinterpret_cast<uintptr_t>(x) & 15) && !(reinterpret_cast<uintptr_t>(y) & 15) &&
         !(bias && (reinterpret_cast<uintptr_t>(bias) & 15));
}
// ---- weight gradient of a token-sized linear layer: gw[n][k] = sum_m g[m][n] x[m][k] (M ~ 50 000 rows, N, K a few
// hundred) with the same six-term split arithmetic.  Both operands arrive "reduction-major" (m is the slow index), so the
// transpose happens on the way in: a thread loads 8 consecutive rows m of ONE column (dword loads, a wave covers 64
// consecutive columns = 256 contiguous bytes per row), splits them and stores the 8 bf16 of a piece as one 16-byte LDS
// row segment -- LDS then holds [column][m] tiles exactly like the forward kernel's [row][k] tiles and the MFMA part is
// the same.  The M rows are cut into chunks (split-K) so that (N/128)(K/128) x chunks workgroups fill the chip; every
// workgroup writes its 128 x 128 partial to the workspace and wgrad_reduce_f32 sums the chunks in a fixed order.
// MFMA roles: A operand = x piece (i = k), B operand = g piece (j = n): float4 stores along k of the row-major [N, K]
// result.
// EXT: `xpos` (optional, [pos_rows, K]): x[m] + xpos[m % pos_rows] is the operand (the layer's input was `hidden + pos`, added
// on load in the forward too); `row_keep` (optional, [M] bytes): rows of g with 0 count as zero rows (padded tokens).  A
// template parameter, not a runtime test: with the tests in the load loop the plain kernel lost its batched loads (80 launches
// per train step: 4.1 -> 5.0 ms).
template <bool EXT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void wgrad_split_bf16_f32(
    const float* __restrict__ G, int ldg, const float* __restrict__ X, int ldx, float* __restrict__ partial, int M, int N,
    int K, int rows_per_chunk, const float* __restrict__ xpos, int pos_rows, const unsigned char* __restrict__ row_keep) {
  constexpr int BM = 128;
  const int ktiles = K / kBN, tiles = (N / BM) * ktiles;
  const int lb = xcd_tile(blockIdx.x, gridDim.x);
  const int chunk = lb / tiles, t = lb - chunk * tiles;
  const int n0 = (t / ktiles) * BM, k0 = (t % ktiles) * kBN;
  const int mbeg = chunk * rows_per_chunk, mend = min(M, mbeg + rows_per_chunk);
  const int nstage = (mend - mbeg + kBK - 1) / kBK;
  __shared__ __attribute__((aligned(16))) __bf16 sG[3 * BM * kPitch];
  __shared__ __attribute__((aligned(16))) __bf16 sX[3 * kBN * kPitch];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;     // 4 x 2 waves: 32 rows (n) x 64 columns (k) each
  const int li = lane & 31, hf = lane >> 5;
  const int col = tid & 127, mq = tid >> 7;    // loader: column of the tile, rows 8 mq .. 8 mq + 7 of the stage
  const float* gp = G + n0 + col;
  const float* xp = X + k0 + col;
  const float* pp = (EXT && xpos != nullptr) ? xpos + k0 + col : nullptr;

  float rg[8], rx[8];
  auto issue = [&](int s) {
    const int m = mbeg + s * kBK + mq * 8;
    if constexpr (EXT) {
      float rp[8];
      unsigned char rk[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {   // all loads of the stage first (no test between them), then the arithmetic
        const bool ok = m + e < mend;
        const unsigned r = (unsigned)(ok ? m + e : mbeg);
        rg[e] = gp[(size_t)r * ldg];
        rx[e] = xp[(size_t)r * ldx];
        rp[e] = pp != nullptr ? pp[(size_t)(r % (unsigned)pos_rows) * K] : 0.f;
        rk[e] = row_keep != nullptr ? row_keep[r] : (unsigned char)1;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const bool ok = m + e < mend;
        rg[e] = (ok && rk[e] != 0) ? rg[e] : 0.f;
        rx[e] = ok ? rx[e] + rp[e] : 0.f;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const bool ok = m + e < mend;
        const size_t r = (size_t)(ok ? m + e : mbeg);
        const float a = gp[r * ldg], b = xp[r * ldx];
        rg[e] = ok ? a : 0.f;
        rx[e] = ok ? b : 0.f;
      }
    }
  };
  auto stash_one = [&](const float (&v)[8], __bf16* base) {
    Split3 sp[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) sp[e] = split3(v[e]);
    __bf16* p = base + col * kPitch + mq * 8;
    *reinterpret_cast<uint4*>(p) = make_uint4(pack_hi16(sp[0].hi, sp[1].hi), pack_hi16(sp[2].hi, sp[3].hi),
                                              pack_hi16(sp[4].hi, sp[5].hi), pack_hi16(sp[6].hi, sp[7].hi));
    *reinterpret_cast<uint4*>(p + BM * kPitch) =
        make_uint4(pack_hi16(sp[0].mid, sp[1].mid), pack_hi16(sp[2].mid, sp[3].mid), pack_hi16(sp[4].mid, sp[5].mid),
                   pack_hi16(sp[6].mid, sp[7].mid));
    *reinterpret_cast<uint4*>(p + 2 * BM * kPitch) =
        make_uint4(pack_hi16(sp[0].lo, sp[1].lo), pack_hi16(sp[2].lo, sp[3].lo), pack_hi16(sp[4].lo, sp[5].lo),
                   pack_hi16(sp[6].lo, sp[7].lo));
  };
  auto stash = [&]() {
    stash_one(rg, sG);
    stash_one(rx, sX);
  };

  f32x16 acc[2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

  const __bf16* pa = sG + (wm * 32 + li) * kPitch + 8 * hf;
  const __bf16* pw = sX + (wn * 64 + li) * kPitch + 8 * hf;
  auto compute = [&]() {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[3], w[2][3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        a[p] = *reinterpret_cast<const bf16x8*>(pa + p * BM * kPitch + 16 * ks);
#pragma unroll
        for (int q = 0; q < 2; ++q) w[q][p] = *reinterpret_cast<const bf16x8*>(pw + (p * kBN + q * 32) * kPitch + 16 * ks);
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        f32x16 c = acc[nt];
        c = mfma_bf16(w[nt][2], a[0], c);
        c = mfma_bf16(w[nt][0], a[2], c);
        c = mfma_bf16(w[nt][1], a[1], c);
        c = mfma_bf16(w[nt][1], a[0], c);
        c = mfma_bf16(w[nt][0], a[1], c);
        c = mfma_bf16(w[nt][0], a[0], c);
        acc[nt] = c;
      }
    }
  };
  if (nstage > 0) {
    issue(0);
    stash();
    __syncthreads();
#pragma unroll 1
    for (int s = 0; s + 1 < nstage; ++s) {
      issue(s + 1);
      compute();
      __syncthreads();
      stash();
      __syncthreads();
    }
    compute();
  }
  // D[i = k][j = n]: accumulator r <-> k = (r & 3) + 8 (r >> 2) + 4 hf of the 32-wide k tile, n = lane & 31
  float* out = partial + ((size_t)chunk * N + n0 + wm * 32 + li) * K + k0 + wn * 64 + 4 * hf;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<float4*>(out + nt * 32 + 8 * q) =
          make_float4(acc[nt][4 * q + 0], acc[nt][4 * q + 1], acc[nt][4 * q + 2], acc[nt][4 * q + 3]);
}

// out[i] = sum over chunks of partial[c][i] (float4 columns; 16 columns x 16 chunk lanes per workgroup, fixed order)
__global__ __launch_bounds__(256) void wgrad_reduce_f32(const float4* __restrict__ partial, int chunks, int n4,
                                                        float4* __restrict__ out) {
  __shared__ float4 sm[256];
  const int cl = threadIdx.x & 15, kl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < n4) {
#pragma unroll 4
    for (int k = kl; k < chunks; k += 16) {
      const float4 v = partial[(size_t)k * n4 + c];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  sm[threadIdx.x] = s;
  __syncthreads();
  if (kl == 0 && c < n4) {
#pragma unroll
    for (int k = 1; k < 16; ++k) {
      const float4 v = sm[k * 16 + cl];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    out[c] = s;
  }
}

struct WgradPlan {
  int chunks, rows_per_chunk;
};
WgradPlan wgrad_plan(int M, int N, int K) {
  const int tiles = (N / 128) * (K / 128);
  // two workgroups per CU; measured for 256 x 256 / 256 x 1024 over 50 148 rows with 256 / 384 / 512 / 768 / 1024 / 1536
  // workgroups: 53.6 / 56.4 / 53.2 / 59.8 / 64.4 / 69.8 us and 181.8 / 190.4 / 175.5 / 181.6 / 183.7 / 192.7 us
  int chunks = std::max(1, (512 + tiles - 1) / tiles);
  chunks = std::min(chunks, (M + kBK - 1) / kBK);
  const int rpc = ((M + chunks - 1) / chunks + kBK - 1) / kBK * kBK;
  return WgradPlan{(M + rpc - 1) / rpc, rpc};
}

// W [N, K] fp32 (or its transpose: `transposed`, element (n, k) at w[k * ldw + n]) -> the operand stream of the kernel
// above, [N/128][K/32][3 pieces][128][32] bf16.  Pieces are rounded to nearest even like ops._split3_bf16 (the residuals
// stay exact in fp32); one thread per element, consecutive threads along k of one row -> 64-byte output segments.
__device__ __forceinline__ unsigned bf16_rne(float x) {
  const unsigned u = __float_as_uint(x);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0u;
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
// `pair`: the second half of the grid writes the tiling of W^T behind the first one (out + 3 N K), so that the forward
// of a training step prepares the operand of its data-gradient product in the same launch.
__global__ __launch_bounds__(256) void tile_weights_f32(const float* __restrict__ w, int ldw, int transposed, int N, int K,
                                                        unsigned short* __restrict__ out, int pair) {
  int idx = blockIdx.x * 256 + threadIdx.x;
  if (pair && idx >= N * K) {   // W^T [K, N] read in place
    idx -= N * K;
    out += (size_t)3 * N * K;
    transposed = 1;
    const int t = N;
    N = K;
    K = t;
  }
  if (idx >= N * K) return;
  const int kk = idx & 31, nn = (idx >> 5) & 127, blk = idx >> 12;
  const int ktiles = K / kBK, kt = blk % ktiles, nt = blk / ktiles;
  const int n = nt * kBN + nn, k = kt * kBK + kk;
  const float x = transposed ? w[(size_t)k * ldw + n] : w[(size_t)n * ldw + k];
  const unsigned hi = bf16_rne(x);
  const float r1 = x - __uint_as_float(hi << 16);
  const unsigned mid = bf16_rne(r1);
  const unsigned lo = bf16_rne(r1 - __uint_as_float(mid << 16));
  unsigned short* o = out + (size_t)blk * (3 * kBN * kBK) + nn * kBK + kk;
  o[0] = (unsigned short)hi;
  o[kBN * kBK] = (unsigned short)mid;
  o[2 * kBN * kBK] = (unsigned short)lo;
}

// Several weights in ONE launch (a training step re-tiles every encoder weight after each optimizer step: 5 launches + 2
// concatenations per layer before): problem i = the pair tiling (W and W^T) of a [N_i, K_i] weight whose first `split` rows come
// from w and the rest from w2 (the offsets | attention-weights projection as one [384, 256] weight without a materialised cat).
constexpr int kMaxTileProblems = 8;
struct TileProblem {
  const float* w;
  const float* w2;
  unsigned short* out;
  int ldw, ldw2, split, N, K, first_block;
};
struct TileProblems {
  TileProblem p[kMaxTileProblems];
};
__global__ __launch_bounds__(256) void tile_weights_multi_f32(TileProblems P, int nprob) {
  int pi = 0;
  while (pi + 1 < nprob && (int)blockIdx.x >= P.p[pi + 1].first_block) ++pi;
  const TileProblem& T = P.p[pi];
  int idx = ((int)blockIdx.x - T.first_block) * 256 + threadIdx.x;
  int N = T.N, K = T.K;
  unsigned short* out = T.out;
  const bool transposed = idx >= N * K;   // second half: the tiling of W^T [K, N] behind the first one
  if (transposed) {
    idx -= N * K;
    out += (size_t)3 * N * K;
    const int t = N;
    N = K;
    K = t;
  }
  if (idx >= N * K) return;
  const int kk = idx & 31, nn = (idx >> 5) & 127, blk = idx >> 12;
  const int ktiles = K / kBK, kt = blk % ktiles, nt = blk / ktiles;
  const int n = nt * kBN + nn, k = kt * kBK + kk;
  const int r = transposed ? k : n, c = transposed ? n : k;   // element (r, c) of the untransposed weight
  const float x = r < T.split ? T.w[(size_t)r * T.ldw + c] : T.w2[(size_t)(r - T.split) * T.ldw2 + c];
  const unsigned hi = bf16_rne(x);
  const float r1 = x - __uint_as_float(hi << 16);
  const unsigned mid = bf16_rne(r1);
  const unsigned lo = bf16_rne(r1 - __uint_as_float(mid << 16));
  unsigned short* o = out + (size_t)blk * (3 * kBN * kBK) + nn * kBK + kk;
  o[0] = (unsigned short)hi;
  o[kBN * kBK] = (unsigned short)mid;
  o[2 * kBN * kBK] = (unsigned short)lo;
}

}  // namespace

extern "C" int egtr_gemm_split_tile_weights_multi_f32(egtr_stream_t stream, int num_weights, const float* const* w,
                                                      const int* ldw, const float* const* w2, const int* ldw2,
                                                      const int* split_rows, const int* N, const int* K,
                                                      uint16_t* const* w_tiled_pair) {
  if (!w || !ldw || !N || !K || !w_tiled_pair || num_weights <= 0 || num_weights > kMaxTileProblems) return EGTR_E_ARG;
  TileProblems P = {};
  long long blocks = 0;
  for (int i = 0; i < num_weights; ++i) {
    const float* second = w2 != nullptr ? w2[i] : nullptr;
    const int split = second != nullptr ? split_rows[i] : N[i];
    if (!w[i] || !w_tiled_pair[i] || N[i] <= 0 || K[i] <= 0 || ldw[i] < K[i] || split <= 0 || split > N[i] ||
        (second != nullptr && (!ldw2 || ldw2[i] < K[i])))
      return EGTR_E_ARG;
    if (N[i] % kBN != 0 || K[i] % kBN != 0 || (long long)N[i] * K[i] > (1LL << 29)) return EGTR_E_UNSUPPORTED;
    P.p[i] = TileProblem{w[i], second, reinterpret_cast<unsigned short*>(w_tiled_pair[i]), ldw[i], second ? ldw2[i] : 0, split,
                         N[i], K[i], (int)blocks};
    blocks += 2 * ((long long)N[i] * K[i] / 256);
  }
  if (blocks >= (1ll << 31)) return EGTR_E_UNSUPPORTED;
  hipLaunchKernelGGL(tile_weights_multi_f32, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), P,
                     num_weights);
  return egtr_check_launch();
}

extern "C" int egtr_linear_split_bf16_f32(egtr_stream_t stream, const float* x, int ldx, const uint16_t* w_tiled,
                                          const float* bias, float* y, int ldy, int M, int K, int N, int relu) {
  if (!x || !w_tiled || !y) return EGTR_E_ARG;
  if (M <= 0 || K <= 0 || N <= 0 || ldx < K || ldy < N) return EGTR_E_ARG;
  if (K % kBK != 0 || !problem_ok(x, ldx, w_tiled, bias, y, ldy, K, N)) return EGTR_E_UNSUPPORTED;
  GemmProblems P = {};
  P.p[0] = GemmProblem{x, w_tiled, bias, y, ldx, ldy, N, relu, nullptr, 1, nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr};
  return launch_grouped(static_cast<hipStream_t>(stream), P, 1, M, K);
}

extern "C" int egtr_linear_split_bf16_ex_f32(egtr_stream_t stream, int num_problems, const float* const* x, const int* ldx,
                                             const uint16_t* const* w_tiled, const float* const* bias, float* const* y,
                                             const int* ldy, const int* N, const int* relu, int M, int K,
                                             const float* const* pos, const int* pos_rows,
                                             const unsigned char* const* row_keep, const float* const* relu_ref,
                                             const int* ldref, const float* const* add1, const float* const* add2,
                                             const int* ldadd, float* const* colpart) {
  if (!x || !ldx || !w_tiled || !bias || !y || !ldy || !N || !relu) return EGTR_E_ARG;
  if (num_problems <= 0 || num_problems > kMaxProblems || M <= 0 || K <= 0) return EGTR_E_ARG;
  if ((pos != nullptr && pos_rows == nullptr) || (relu_ref != nullptr && ldref == nullptr) ||
      ((add1 != nullptr || add2 != nullptr) && ldadd == nullptr))
    return EGTR_E_ARG;
  if (K % kBK != 0) return EGTR_E_UNSUPPORTED;
  GemmProblems P = {};
  for (int i = 0; i < num_problems; ++i) {
    if (!x[i] || !w_tiled[i] || !y[i]) return EGTR_E_ARG;
    if (!problem_ok(x[i], ldx[i], w_tiled[i], bias[i], y[i], ldy[i], K, N[i])) return EGTR_E_UNSUPPORTED;
    const float* p = pos != nullptr ? pos[i] : nullptr;
    if (p != nullptr && pos_rows[i] <= 0) return EGTR_E_ARG;
    if (p != nullptr && (reinterpret_cast<uintptr_t>(p) & 15)) return EGTR_E_UNSUPPORTED;
    const float* rr = relu_ref != nullptr ? relu_ref[i] : nullptr;
    const float* a1 = add1 != nullptr ? add1[i] : nullptr;
    const float* a2 = add2 != nullptr ? add2[i] : nullptr;
    float* cp = colpart != nullptr ? colpart[i] : nullptr;
    if ((rr && (ldref[i] < N[i] || (ldref[i] & 3) || (reinterpret_cast<uintptr_t>(rr) & 15))) ||
        ((a1 || a2) && (ldadd[i] < N[i] || (ldadd[i] & 3))) || (reinterpret_cast<uintptr_t>(a1) & 15) ||
        (reinterpret_cast<uintptr_t>(a2) & 15) || (reinterpret_cast<uintptr_t>(cp) & 15))
      return EGTR_E_UNSUPPORTED;
    P.p[i] = GemmProblem{x[i], w_tiled[i], bias[i], y[i], ldx[i], ldy[i], N[i], relu[i], p, p != nullptr ? pos_rows[i] : 1,
                         row_keep != nullptr ? row_keep[i] : nullptr, rr, rr ? ldref[i] : 0, a1, a2,
                         (a1 || a2) ? ldadd[i] : 0, cp};
  }
  return launch_grouped(static_cast<hipStream_t>(stream), P, num_problems, M, K);
}

extern "C" int egtr_linear_split_bf16_grouped_pos_f32(egtr_stream_t stream, int num_problems, const float* const* x,
                                                      const int* ldx, const uint16_t* const* w_tiled,
                                                      const float* const* bias, float* const* y, const int* ldy,
                                                      const int* N, const int* relu, int M, int K,
                                                      const float* const* pos, const int* pos_rows) {
  return egtr_linear_split_bf16_ex_f32(stream, num_problems, x, ldx, w_tiled, bias, y, ldy, N, relu, M, K, pos, pos_rows,
                                       nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
}

extern "C" int egtr_gemm_split_tile_weights_f32(egtr_stream_t stream, const float* w, int ldw, int transposed, int N, int K,
                                                uint16_t* w_tiled) {
  if (!w || !w_tiled || N <= 0 || K <= 0 || ldw < (transposed ? N : K)) return EGTR_E_ARG;
  if (N % kBN != 0 || K % kBK != 0 || (long long)N * K > (1LL << 30)) return EGTR_E_UNSUPPORTED;
  hipLaunchKernelGGL(tile_weights_f32, dim3(N * K / 256), dim3(256), 0, static_cast<hipStream_t>(stream), w, ldw,
                     transposed, N, K, reinterpret_cast<unsigned short*>(w_tiled), 0);
  return egtr_check_launch();
}

extern "C" int egtr_gemm_split_tile_weights_pair_f32(egtr_stream_t stream, const float* w, int ldw, int N, int K,
                                                     uint16_t* w_tiled_pair) {
  if (!w || !w_tiled_pair || N <= 0 || K <= 0 || ldw < K) return EGTR_E_ARG;
  if (N % kBN != 0 || K % kBN != 0 || (long long)N * K > (1LL << 29)) return EGTR_E_UNSUPPORTED;
  hipLaunchKernelGGL(tile_weights_f32, dim3(2 * (N * K / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), w, ldw, 0,
                     N, K, reinterpret_cast<unsigned short*>(w_tiled_pair), 1);
  return egtr_check_launch();
}

extern "C" long long egtr_linear_split_bf16_wgrad_workspace_floats(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0 || N % 128 || K % 128) return 0;
  return (long long)wgrad_plan(M, N, K).chunks * N * K;
}

extern "C" int egtr_linear_split_bf16_wgrad_ex_f32(egtr_stream_t stream, const float* g, int ldg, const float* x, int ldx,
                                                   float* grad_weight, float* workspace, int M, int N, int K,
                                                   const float* x_pos, int pos_rows, const unsigned char* row_keep) {
  if (!g || !x || !grad_weight || !workspace || M <= 0 || N <= 0 || K <= 0 || ldg < N || ldx < K) return EGTR_E_ARG;
  if (x_pos != nullptr && pos_rows <= 0) return EGTR_E_ARG;
  if (N % 128 || K % 128 || (reinterpret_cast<uintptr_t>(grad_weight) & 15) || (reinterpret_cast<uintptr_t>(workspace) & 15))
    return EGTR_E_UNSUPPORTED;
  const WgradPlan pl = wgrad_plan(M, N, K);
  const long long wgs = (long long)pl.chunks * (N / 128) * (K / 128);
  if (wgs >= (1ll << 31)) return EGTR_E_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (x_pos != nullptr || row_keep != nullptr)
    hipLaunchKernelGGL(wgrad_split_bf16_f32<true>, dim3((unsigned)wgs), dim3(512), 0, st, g, ldg, x, ldx, workspace, M, N, K,
                       pl.rows_per_chunk, x_pos, pos_rows, row_keep);
  else
    hipLaunchKernelGGL(wgrad_split_bf16_f32<false>, dim3((unsigned)wgs), dim3(512), 0, st, g, ldg, x, ldx, workspace, M, N, K,
                       pl.rows_per_chunk, x_pos, pos_rows, row_keep);
  int rc = egtr_check_launch();
  if (rc != EGTR_OK) return rc;
  const int n4 = N * K / 4;
  hipLaunchKernelGGL(wgrad_reduce_f32, dim3((n4 + 15) / 16), dim3(256), 0, st, reinterpret_cast<const float4*>(workspace),
                     pl.chunks, n4, reinterpret_cast<float4*>(grad_weight));
  return egtr_check_launch();
}

extern "C" int egtr_linear_split_bf16_wgrad_f32(egtr_stream_t stream, const float* g, int ldg, const float* x, int ldx,
                                                float* grad_weight, float* workspace, int M, int N, int K) {
  return egtr_linear_split_bf16_wgrad_ex_f32(stream, g, ldg, x, ldx, grad_weight, workspace, M, N, K, nullptr, 1, nullptr);
}
