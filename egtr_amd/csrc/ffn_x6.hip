// The encoder layer's feed-forward block as ONE kernel, fp32 in / fp32 out with fp32-level accuracy on the bf16 matrix cores:
//     y = LayerNorm(x + fc2(relu(fc1(x))))   [+ y + pos]          (model/deformable_detr.py:1335-1345, d_model = 256)
// (the LayerNorm / residual part is optional: without it y = fc2(relu(fc1(x)))).  The reference runs two token-sized
// nn.Linear layers with a [S, 1024] fp32 activation (51 MB at S = 12 537) written and read in between, plus an add and a
// LayerNorm kernel.  Here a workgroup owns 64 token rows for the whole block and the hidden activation never leaves the CU:
//
//   * the 64 x 256 input panel is split ONCE into its three bf16 pieces (xs_format.h): hi and mid live in LDS as MFMA
//     operand fragments (64 KiB), lo in registers (each wave: the 16 fragments of its 32 rows);
//   * the hidden dimension is walked in chunks of 64 units: layer 1 for the chunk (K = 256: 4 stages of 4 k-steps), bias +
//     ReLU + split of the 64 x 64 chunk into LDS (24 KiB), layer 2 accumulating the chunk into the 64 x 256 output tile
//     (K = 64: 4 stages of 1 k-step, 4 accumulator tiles per wave);
//   * the weights arrive pre-split in the XS format and stream through a ring of three 24 KiB LDS stages filled by LDS-DMA
//     two stages ahead (counted vmcnt, raw s_barrier, one barrier per stage): every stage, of either layer, is 24 fragments
//     and 24 MFMAs per wave, so the two layers form ONE uniform stream of 8 stages per chunk;
//   * the six-term split product (x6_common.h) everywhere: fp32 operands, fp32 accumulation, error of an fp32 GEMM.
// LDS: 64 (panel) + 24 (hidden chunk) + 72 (ring) = 160 KiB, one workgroup of four waves per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"
#include "x6_common.h"
#include "xs_format.h"

namespace {
using namespace x6;

constexpr int kD = 256;                          // d_model: K of layer 1, N of layer 2
constexpr int kRows = 64;                        // token rows per workgroup
constexpr int kKS = kD / 16;                     // k-steps over d_model
constexpr int kFrag = xs::kFragBytes;
constexpr int kPanel = 2 * kKS * 2 * kFrag;      // [row block 2][k-step 16][hi, mid]        64 KiB
constexpr int kHbuf = 2 * 4 * 3 * kFrag;         // XS(hidden chunk): [row block 2][k-step 4][3]   24 KiB
constexpr int kStage = 24 * kFrag;               // one weight stage                              24 KiB
constexpr int kLds = kPanel + kHbuf + 3 * kStage;
static_assert(kLds == 160 * 1024, "the whole LDS of a CU");
constexpr int NL = 6;                            // DMA instructions per wave and stage
typedef float f32x8 __attribute__((ext_vector_type(8)));

struct FfnArgs {
  const float* x;        // [M, ldx] fp32
  const float* res;      // residual rows [M, ldr] added before the LayerNorm (the FFN: x itself)
  const char* w1;        // XS(W1 [F, 256])
  const float* b1;       // [F]
  const char* w2;        // XS(W2 [256, F])
  const float* b2;       // [256]
  const float* gamma;    // LayerNorm weight / bias [256], or null: no residual, no LayerNorm
  const float* beta;
  const float* pos;      // [pos_rows, 256] or null
  float* out;            // [M, 256]
  float* out_pos;        // [M, 256] = out + pos[row % pos_rows], or null
  int M, ldx, ldr, F, pos_rows;
  float eps;
  // ffn_x6_kernel<true> (encoder layer tail): x is the attention context; y1 = LayerNorm1(res + x . Wp^T + bp) is the
  // FFN's input AND its residual (res / ldr = the layer input, the residual of LayerNorm1)
  const char* wp;        // XS(Wp [256, 256])
  const float* bp;       // [256]
  const float* gamma1;   // LayerNorm1 weight / bias
  const float* beta1;
  float eps1;
};

// Logical 16-byte slot x (= row + 32 * k-group) of a panel fragment of k-step ks -> where it is stored.  build_panel's lane
// owns four consecutive channels of ONE row: a store instruction covers the 16 k-steps x 2 k-groups of a row, i.e. fragments
// 2 KiB apart -- unswizzled, the 16 lanes of a ds_write_b64 group would share 16 bytes' worth of banks (8-way conflict; the
// stores of the prologue alone cost ~3 us per launch).  With the XOR they cover all 32 banks, and the MFMA operand read (slot =
// lane) stays a permutation of the fragment's 64 slots inside each ds_read_b128 lane group: conflict-free as before.
__device__ __forceinline__ int swz(int x, int ks) { return x ^ (ks & 15) ^ ((x >> 5) << 2); }

// The 64 x 256 fp32 input panel of a workgroup -> MFMA operand fragments (slots swizzled, swz()): hi / mid pieces into `panel`
// ([row block 2][k-step 16][hi, mid] KiB), lo pieces of row block 0 / 1 into lo0 / lo1 ([k-step 16] KiB each, parking
// places from which every wave then reads the 16 lo fragments of ITS row block into registers).  No synchronisation inside.
// split3_fast: a non-finite input makes the lower pieces NaN, so its output ROW is NaN rather than +-inf / NaN -- non-finite
// either way, and only that row (tests/test_gpu_pinning.py).
__device__ __forceinline__ void build_panel(const float* __restrict__ x, int ldx, int M, int r0, int tid, int lane,
                                            char* panel, char* lo0, char* lo1) {
  float4 v[16];
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int row = min(r0 + it * 4 + (tid >> 6), M - 1);
    v[it] = *reinterpret_cast<const float4*>(x + (size_t)row * ldx + 4 * lane);
  }
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int row = it * 4 + (tid >> 6), rb = row >> 5, r = row & 31;
    const int ks = lane >> 2, off = (swz(r + 32 * ((lane >> 1) & 1), ks) << 4) + (lane & 1) * 8;
    const xs::Split3 s0 = xs::split3_fast(v[it].x), s1 = xs::split3_fast(v[it].y), s2 = xs::split3_fast(v[it].z),
                     s3 = xs::split3_fast(v[it].w);
    char* p = panel + ((rb * kKS + ks) * 2) * kFrag + off;
    *reinterpret_cast<uint2*>(p) = make_uint2(xs::pack_hi16(s0.hi, s1.hi), xs::pack_hi16(s2.hi, s3.hi));
    *reinterpret_cast<uint2*>(p + kFrag) = make_uint2(xs::pack_hi16(s0.mid, s1.mid), xs::pack_hi16(s2.mid, s3.mid));
    char* q = (rb == 0 ? lo0 : lo1) + ks * kFrag + off;
    *reinterpret_cast<uint2*>(q) = make_uint2(xs::pack_hi16(s0.lo, s1.lo), xs::pack_hi16(s2.lo, s3.lo));
  }
}

// Output tile of a workgroup: y = acc + bias2; with LayerNorm parameters y = LayerNorm(res + y) over the 256 channels of a
// row (dd:1329-1330, 1343-1345) and optionally y + pos.  D[i = n][j = m]: accumulator r <-> n = (r & 3) + 8 (r >> 2) + 4 hf
// of the 32-wide tile, m = lane & 31: a lane holds 64 channels of its row, lane ^ 32 another 64, the wave next door (wn ^ 1)
// the other 128 -- row sums through a cross-lane swap and one exchange through `red` (1 KiB of idle LDS), mean first, then
// the centred sum of squares.
// res_of(t, q): the residual values of register group (t, q) when they are NOT read from A.res (encoder tail: LayerNorm1's
// result still in registers); USE_RES_OF selects.
template <bool USE_RES_OF, class ResFn>
__device__ __forceinline__ void final_epilogue(const f32x16 (&acc2)[4], const FfnArgs& A, int r0, int wave, int lane,
                                               float* red, ResFn&& res_of) {
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, hf = lane >> 5;
  const int row = r0 + wm * 32 + li;
  float4 y[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int col = (4 * wn + t) * 32 + 8 * q + 4 * hf;
      const float4 b = A.b2 != nullptr ? *reinterpret_cast<const float4*>(A.b2 + col) : make_float4(0.f, 0.f, 0.f, 0.f);
      y[t][q] = make_float4(acc2[t][4 * q + 0] + b.x, acc2[t][4 * q + 1] + b.y, acc2[t][4 * q + 2] + b.z,
                            acc2[t][4 * q + 3] + b.w);
    }
  if (A.gamma != nullptr) {
    const float* xr = A.res + (size_t)min(row, A.M - 1) * A.ldr;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // the residual rows: from memory, or (encoder tail) LayerNorm1's result still in registers
        float4 r;
        if constexpr (USE_RES_OF) r = res_of(t, q);
        else r = *reinterpret_cast<const float4*>(xr + (4 * wn + t) * 32 + 8 * q + 4 * hf);
        y[t][q].x += r.x; y[t][q].y += r.y; y[t][q].z += r.z; y[t][q].w += r.w;
      }
    __syncthreads();                                      // every wave is done with the LDS `red` aliases
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) sum += (y[t][q].x + y[t][q].y) + (y[t][q].z + y[t][q].w);
    sum += __shfl_xor(sum, 32);
    if (hf == 0) red[wave * 32 + li] = sum;
    __syncthreads();
    const float mean = (sum + red[(wave ^ 1) * 32 + li]) * (1.f / kD);
    float sq = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        y[t][q].x -= mean; y[t][q].y -= mean; y[t][q].z -= mean; y[t][q].w -= mean;
        sq += (y[t][q].x * y[t][q].x + y[t][q].y * y[t][q].y) + (y[t][q].z * y[t][q].z + y[t][q].w * y[t][q].w);
      }
    sq += __shfl_xor(sq, 32);
    if (hf == 0) red[128 + wave * 32 + li] = sq;
    __syncthreads();
    const float rstd = rsqrtf((sq + red[128 + (wave ^ 1) * 32 + li]) * (1.f / kD) + A.eps);
    const float* pr = A.out_pos != nullptr ? A.pos + (size_t)(min(row, A.M - 1) % A.pos_rows) * kD : nullptr;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = (4 * wn + t) * 32 + 8 * q + 4 * hf;
        const float4 g = *reinterpret_cast<const float4*>(A.gamma + col), b = *reinterpret_cast<const float4*>(A.beta + col);
        y[t][q] = make_float4(y[t][q].x * rstd * g.x + b.x, y[t][q].y * rstd * g.y + b.y, y[t][q].z * rstd * g.z + b.z,
                              y[t][q].w * rstd * g.w + b.w);
        if (pr != nullptr && row < A.M) {
          const float4 p = *reinterpret_cast<const float4*>(pr + col);
          *reinterpret_cast<float4*>(A.out_pos + (size_t)row * kD + col) =
              make_float4(y[t][q].x + p.x, y[t][q].y + p.y, y[t][q].z + p.z, y[t][q].w + p.w);
        }
      }
  }
  if (row < A.M) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(A.out + (size_t)row * kD + (4 * wn + t) * 32 + 8 * q + 4 * hf) = y[t][q];
  }
}

// LayerNorm over the 256 channels of the rows of a 64 x 256 tile held in the accumulator layout (see final_epilogue), in
// place.  `red`: 1 KiB of LDS nobody else touches meanwhile.  Three workgroup barriers.
__device__ __forceinline__ void layernorm_rows(float4 (&y)[4][4], const float* gamma, const float* beta, float eps, int wave,
                                               int lane, float* red) {
  const int wn = wave & 1, li = lane & 31, hf = lane >> 5;
  __syncthreads();
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) sum += (y[t][q].x + y[t][q].y) + (y[t][q].z + y[t][q].w);
  sum += __shfl_xor(sum, 32);
  if (hf == 0) red[wave * 32 + li] = sum;
  __syncthreads();
  const float mean = (sum + red[(wave ^ 1) * 32 + li]) * (1.f / kD);
  float sq = 0.f;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      y[t][q].x -= mean; y[t][q].y -= mean; y[t][q].z -= mean; y[t][q].w -= mean;
      sq += (y[t][q].x * y[t][q].x + y[t][q].y * y[t][q].y) + (y[t][q].z * y[t][q].z + y[t][q].w * y[t][q].w);
    }
  sq += __shfl_xor(sq, 32);
  if (hf == 0) red[128 + wave * 32 + li] = sq;
  __syncthreads();
  const float rstd = rsqrtf((sq + red[128 + (wave ^ 1) * 32 + li]) * (1.f / kD) + eps);
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int col = (4 * wn + t) * 32 + 8 * q + 4 * hf;
      const float4 g = *reinterpret_cast<const float4*>(gamma + col), b = *reinterpret_cast<const float4*>(beta + col);
      y[t][q] = make_float4(y[t][q].x * rstd * g.x + b.x, y[t][q].y * rstd * g.y + b.y, y[t][q].z * rstd * g.z + b.z,
                            y[t][q].w * rstd * g.w + b.w);
    }
}

// PROJ = false: the FFN block of the header.  PROJ = true: the whole TAIL of an encoder layer in one launch --
//     y1 = self_attn_layer_norm(hidden + output_proj(context));  out = final_layer_norm(y1 + fc2(relu(fc1(y1))))  [+ pos]
// (model/deformable_detr.py:1102, 1326-1345): sixteen weight stages of the output projection run first on the same ring
// (context panel in, 64 x 256 tile in the layer-2 accumulators), LayerNorm1 happens in registers, its result is split into the
// panel in place of the context (lo pieces through the idle chunk buffer, one row block at a time) and kept in registers as
// the FFN's residual -- y1 never goes to memory, and one prologue, one epilogue burst and one launch disappear.
template <bool PROJ>
__global__ __launch_bounds__(256, 1) void ffn_x6_kernel(FfnArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const panel = smem;
  char* const hbuf = smem + kPanel;
  char* const ring = smem + kPanel + kHbuf;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, hf = lane >> 5;
  const unsigned lds_ring = (unsigned)reinterpret_cast<uintptr_t>((lds_char*)ring);
  const int r0 = blockIdx.x * kRows;
  const int nchunk = A.F >> 6, KSf = A.F >> 4;

  // ---- weight stream.  Global stage g: [PROJ: g < 16 = k-step g of the output projection, all 256 outputs ([8 n-blocks][3
  //      pieces]), then] stage 8 c + s of the FFN -- s < 4: layer 1, hidden units 64 c .., k-steps 4 s .. 4 s + 3 of d_model
  //      ([2 n-blocks][4 k-steps][3 pieces]); s >= 4: layer 2, k-step 4 c + (s - 4) of the hidden dimension, all 256 outputs
  //      ([8 n-blocks][3 pieces]).  Wave w moves fragments 6 w .. 6 w + 5 of the stage image: contiguous in LDS, and in
  //      global memory one 6 KiB run (layer 1) or two 3 KiB runs (layer 2, projection).
  const unsigned voff = lane * 16;
  // Workgroup b walks the hidden chunks in the rotated order b, b + 1, .. (mod nchunk): workgroups that run side by side
  // then stream DIFFERENT weight fragments at any moment (all of them fetching the same 24 KiB stage at once makes a few
  // L2 channels the bottleneck: 91 -> 8x us at S = 12 537).  The sum over chunks is the same set in a rotated order.
  const int crot = blockIdx.x % nchunk;
  auto chunk_of = [&](int c) { const int t = c + crot; return t >= nchunk ? t - nchunk : t; };
  constexpr int NP = PROJ ? kKS : 0;           // stages of the output projection in front of the FFN stream
  const int nstages = NP + 8 * nchunk;
  const size_t nb_stride = (size_t)KSf * 3 * kFrag;
  // source of this wave's six fragments of global stage g: fragment j at base + (j / 3) * stride + (j % 3) KiB -- scalar
  // arithmetic, the same instruction sequence for every kind of stage (no branch between MFMAs).  Beyond the last stage:
  // the last one again (never read; keeps every wait count an immediate).
  auto stage_src = [&](int g, const char*& base, size_t& stride) {
    g = g < nstages ? g : nstages - 1;
    const bool pj = g < NP;
    const int gg = g - NP, c = chunk_of(pj ? 0 : gg >> 3), st = gg & 7;
    const bool l1 = st < 4;
    const char* b1 = A.w1 + ((size_t)((2 * c + (wave >> 1)) * kKS + 4 * (st & 3) + 2 * (wave & 1)) * 3) * kFrag;
    const char* b2 = A.w2 + ((size_t)(2 * wave * KSf + 4 * c + (st & 3)) * 3) * kFrag;
    base = l1 ? b1 : b2;
    stride = l1 ? (size_t)3 * kFrag : nb_stride;
    if constexpr (PROJ) {
      base = pj ? A.wp + ((size_t)(2 * wave * kKS + g) * 3) * kFrag : base;
      stride = pj ? (size_t)kKS * 3 * kFrag : stride;
    }
  };
  auto issue = [&](int g, int slot) {
    const unsigned dst = lds_ring + (unsigned)slot * kStage + (unsigned)wave * (NL * kFrag);
    const char* base;
    size_t stride;
    stage_src(g, base, stride);
#pragma unroll
    for (int i = 0; i < NL; ++i) dma16s(base + (i / 3) * stride + (i % 3) * kFrag, voff, dst + i * kFrag);
  };
  issue(0, 0);
  issue(1, 1);

  build_panel(A.x, A.ldx, A.M, r0, tid, lane, panel, hbuf, ring + 2 * kStage);
  __syncthreads();
  bf16x8 lo[kKS];
  {
    const char* q = (wm == 0 ? hbuf : ring + 2 * kStage);
#pragma unroll
    for (int ks = 0; ks < kKS; ++ks) lo[ks] = *reinterpret_cast<const bf16x8*>(q + ks * kFrag + (swz(lane, ks) << 4));
  }

  f32x16 acc2[4], acc1[2];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[t][r] = 0.f;

  // hi (piece 0) / mid (piece 1) fragment of k-step ks of this wave's 32 panel rows
  auto pfrag = [&](int ks, int piece) {
    return *reinterpret_cast<const bf16x8*>(panel + ((wm * kKS + ks) * 2 + piece) * kFrag + (swz(lane, ks) << 4));
  };
  const char* const ph = hbuf + (wm * 4 * 3) * kFrag + lane * 16;      // hidden-chunk fragments of this wave's 32 rows
  // weight fragments of a stage, as this wave reads them: layer 1 = [k-step 4][piece 3] of n-block wn, layer 2 / projection
  // = [n tile 4][piece 3] of n-blocks 4 wn ..: in both images fragment (u, p) sits at ((4 wn + u) * 3 + p) KiB
  const char* const pw = ring + (4 * wn * 3) * kFrag + lane * 16;
  auto frag = [](const char* p) { return *reinterpret_cast<const bf16x8*>(p); };

  // One stage in PINNED program order (sched_barrier(0) between the items; left alone, hipcc reads every operand right
  // before its use and chains six dependent MFMAs on one accumulator): 24 MFMAs, consecutive ones on different
  // accumulators; behind MFMAs 1, 3, .., 11 the six DMA instructions of stage g + 3; behind MFMAs 12 .. 17 the twelve
  // weight fragments of stage g + 1 (other register set), two at a time; the operand A fragments of the next stage where
  // nothing orders them behind the next barrier.  Everything but the matrix instructions issues in the shadow of an MFMA.
  // S: 0 .. 3 = layer 1 step S; 4 .. 7 = layer 2 step S - 4; 8 + ks = projection k-step ks (PROJ).
  // a / anx: operand A of this / the next stage: layer 1 = [k-step 4][hi, mid] (lo comes from the registers); layer 2 and
  // projection = the three pieces of ONE k-step in a[0][0], a[0][1], a[1][0].
  auto stage = [&](auto S, const bf16x8 (&w)[4][3], bf16x8 (&wnx)[4][3], const bf16x8 (&a)[4][2], bf16x8 (&anx)[4][2], int g,
                   int slot_next, int slot_fill) {
    constexpr int s = decltype(S)::value;
    const char* const wn_src = pw + slot_next * kStage;
    const unsigned dst = lds_ring + (unsigned)slot_fill * kStage + (unsigned)wave * (NL * kFrag);
    const char* src;
    size_t sstride;
    stage_src(g + 3, src, sstride);
    static_for<24>([&](auto I) {
      constexpr int i = decltype(I)::value;
      // the six cross terms, small ones first: (w piece, a piece)
      constexpr int pwt[6] = {2, 0, 1, 1, 0, 0}, pat[6] = {0, 2, 1, 0, 1, 0};
      if constexpr (s < 4) {
        // MFMA i: k-step pair (i / 12), term (i % 12) / 2, k-step parity i & 1 -> accumulator i & 1 (four accumulators, one
        // per k-step, measured slower: 92.4 vs 87.3 us -- register pressure)
        constexpr int pair = i / 12, term = (i % 12) / 2, par = i & 1, kl = 2 * pair + par;
        if constexpr (pat[term] == 2) acc1[par] = mfma(w[kl][pwt[term]], lo[4 * s + kl], acc1[par]);
        else acc1[par] = mfma(w[kl][pwt[term]], a[kl][pat[term]], acc1[par]);
        __builtin_amdgcn_sched_barrier(0);
      } else {
        constexpr int term = i / 4, t = i % 4;   // term-major over the four output tiles
        constexpr int ap = pat[term];
        acc2[t] = mfma(w[t][pwt[term]], a[ap >> 1][ap & 1], acc2[t]);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr ((i & 1) && (i >> 1) < NL) {
        constexpr int j = i >> 1;
        dma16s(src + (j / 3) * sstride + (j % 3) * kFrag, voff, dst + j * kFrag);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (i >= 12 && i < 18) {
        constexpr int j = 2 * (i - 12);
        wnx[j / 3][j % 3] = frag(wn_src + j * kFrag);
        wnx[(j + 1) / 3][(j + 1) % 3] = frag(wn_src + (j + 1) * kFrag);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (s < 8) {
        // layer 1 reads the read-only input panel, steps 1 .. 3 of layer 2 a hidden chunk finished at least a stage ago
        // (step 0: after the barrier)
        constexpr int sx = (s + 1) & 7;
        if constexpr (i >= 18 && i < 22 && sx < 4) {
          constexpr int kl = i - 18;
          anx[kl][0] = pfrag(4 * sx + kl, 0);
          anx[kl][1] = pfrag(4 * sx + kl, 1);
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (i == 18 && sx > 4) {
          anx[0][0] = frag(ph + ((sx - 4) * 3 + 0) * kFrag);
          anx[0][1] = frag(ph + ((sx - 4) * 3 + 1) * kFrag);
          anx[1][0] = frag(ph + ((sx - 4) * 3 + 2) * kFrag);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else if constexpr (s - 8 + 1 < kKS) {
        // projection: the next k-step of the context panel (after the last one the panel is rebuilt first)
        if constexpr (i == 18) {
          anx[0][0] = pfrag(s - 8 + 1, 0);
          anx[0][1] = pfrag(s - 8 + 1, 1);
          anx[1][0] = lo[s - 8 + 1];
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    });
  };

  // Stage g lives in ring slot g % 3.  At the top of stage g: this wave's DMAs of stage g + 1 have landed (those of g + 2
  // may stay in flight: counted vmcnt), everybody's LDS traffic of stage g - 1 is finished (lgkmcnt(0) + barrier): the
  // weights of stage g + 1 may be read (into the other register set) and the slot of stage g refilled with stage g + 3.
  bf16x8 w0[4][3], w1[4][3], a0[4][2], a1[4][2];
  f32x8 bias[4];
  wait_vm<NL>();                      // stage 0 (stage 1 in flight)
  wait_lgkm0();
  __builtin_amdgcn_s_barrier();
  issue(2, 2);
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int p = 0; p < 3; ++p) w0[u][p] = frag(pw + (u * 3 + p) * kFrag);
  int slot = 0;   // ring slot of the stage being multiplied
  int g = 0;      // its global index
  auto stage_top = [&]() {
    // this wave's DMAs of stage g + 1 have landed (those of g + 2 stay in flight), everybody's LDS traffic of stage g - 1
    // is finished (lgkmcnt(0) + barrier)
    wait_vm<NL>();
    wait_lgkm0();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  float4 y1[4][4];   // PROJ: LayerNorm1's result in the accumulator layout = the residual of the final LayerNorm
  if constexpr (PROJ) {
    a0[0][0] = pfrag(0, 0);
    a0[0][1] = pfrag(0, 1);
    a0[1][0] = lo[0];
    static_for<kKS>([&](auto KS) {
      constexpr int ks = decltype(KS)::value;
      const int sn = slot == 2 ? 0 : slot + 1;
      stage_top();
      if constexpr ((ks & 1) == 0) stage(std::integral_constant<int, 8 + ks>{}, w0, w1, a0, a1, g, sn, slot);
      else stage(std::integral_constant<int, 8 + ks>{}, w1, w0, a1, a0, g, sn, slot);
      slot = sn;
      ++g;
    });
    // ---- y1 = LayerNorm1(res + context . Wp^T + bp) in registers (accumulator layout, see final_epilogue)
    const int row = r0 + wm * 32 + li;
    const float* xr = A.res + (size_t)min(row, A.M - 1) * A.ldr;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = (4 * wn + t) * 32 + 8 * q + 4 * hf;
        const float4 b = *reinterpret_cast<const float4*>(A.bp + col), r = *reinterpret_cast<const float4*>(xr + col);
        y1[t][q] = make_float4(acc2[t][4 * q + 0] + b.x + r.x, acc2[t][4 * q + 1] + b.y + r.y, acc2[t][4 * q + 2] + b.z + r.z,
                               acc2[t][4 * q + 3] + b.w + r.w);
        acc2[t][4 * q + 0] = 0.f; acc2[t][4 * q + 1] = 0.f; acc2[t][4 * q + 2] = 0.f; acc2[t][4 * q + 3] = 0.f;
      }
    float* red = reinterpret_cast<float*>(hbuf + 16 * kFrag);     // the chunk buffer is idle until the first hand-over
    layernorm_rows(y1, A.gamma1, A.beta1, A.eps1, wave, lane, red);   // its first barrier: everybody is done with the context panel
    // ---- y1 -> the panel (hi / mid in place of the context), lo through hbuf[0, 16 KiB) one row block at a time.
    //      Accumulator register group (t, q) = channels 32 (4 wn + t) + 8 q + 4 hf .. + 3 of row li: k-step 2 (4 wn + t) +
    //      (q >> 1), k-group q & 1, bytes 8 hf .. 8 hf + 7 of the row's slot.
    uint2 plo[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const xs::Split3 s0 = xs::split3_fast(y1[t][q].x), s1 = xs::split3_fast(y1[t][q].y), s2 = xs::split3_fast(y1[t][q].z),
                         s3 = xs::split3_fast(y1[t][q].w);
        const int ks = 2 * (4 * wn + t) + (q >> 1);
        char* d = panel + ((wm * kKS + ks) * 2) * kFrag + (swz(li + 32 * (q & 1), ks) << 4) + 8 * hf;
        *reinterpret_cast<uint2*>(d) = make_uint2(xs::pack_hi16(s0.hi, s1.hi), xs::pack_hi16(s2.hi, s3.hi));
        *reinterpret_cast<uint2*>(d + kFrag) = make_uint2(xs::pack_hi16(s0.mid, s1.mid), xs::pack_hi16(s2.mid, s3.mid));
        plo[t][q] = make_uint2(xs::pack_hi16(s0.lo, s1.lo), xs::pack_hi16(s2.lo, s3.lo));
      }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      if (wm == rb) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int ks = 2 * (4 * wn + t) + (q >> 1);
            *reinterpret_cast<uint2*>(hbuf + ks * kFrag + (swz(li + 32 * (q & 1), ks) << 4) + 8 * hf) = plo[t][q];
          }
      }
      __syncthreads();
      if (wm == rb) {
#pragma unroll
        for (int ks = 0; ks < kKS; ++ks) lo[ks] = *reinterpret_cast<const bf16x8*>(hbuf + ks * kFrag + (swz(lane, ks) << 4));
      }
      __syncthreads();
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    a0[u][0] = pfrag(u, 0);
    a0[u][1] = pfrag(u, 1);
  }
#pragma unroll 1
  for (int c = 0; c < nchunk; ++c) {
    static_for<8>([&](auto S) {
      constexpr int s = decltype(S)::value;
      const int sn = slot == 2 ? 0 : slot + 1;       // slot of stage g + 1
      stage_top();
      if constexpr (s == 4) {          // the hidden chunk was written by the stage before: visible after this barrier
        a0[0][0] = frag(ph + 0 * kFrag);
        a0[0][1] = frag(ph + 1 * kFrag);
        a0[1][0] = frag(ph + 2 * kFrag);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (s == 0) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc1[t][r] = 0.f;
        // the chunk's 32 layer-1 bias values of this wave (wave-uniform) as SCALAR loads, three stages ahead of their use.
        // (A vector load here would be waited for with vmcnt(0) by the compiler -- draining the two stages of DMA in
        // flight: measured 35 of 91 us.)  Sound only while the compiler does not spill these SGPRs while the loads are in
        // flight: tests/test_build_resources.py fails the build if this kernel has any SGPR spill.
        const float* bp = A.b1 + 64 * chunk_of(c) + 32 * wn;
        asm volatile("s_load_dwordx8 %0, %4, 0x0\n\ts_load_dwordx8 %1, %4, 0x20\n\ts_load_dwordx8 %2, %4, 0x40\n\t"
                     "s_load_dwordx8 %3, %4, 0x60"
                     : "=&s"(bias[0]), "=&s"(bias[1]), "=&s"(bias[2]), "=&s"(bias[3])
                     : "s"(bp));
      }
      if constexpr ((s & 1) == 0) stage(S, w0, w1, a0, a1, g, sn, slot);
      else stage(S, w1, w0, a1, a0, g, sn, slot);
      if constexpr (s == 3) {
        // bias + ReLU + split of the wave's 32 x 32 piece of the hidden chunk -> XS fragments in LDS.  The 32 bias values
        // are wave-uniform: SCALAR loads (they do not touch the vmcnt queue the DMA waits are counted on).
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(bias[0]), "+s"(bias[1]), "+s"(bias[2]), "+s"(bias[3]));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float h[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float blo = bias[q][j], bhi = bias[q][4 + j];   // bias[q] = values 8 q .. 8 q + 7 of the 32
            h[j] = egtr_relu(acc1[0][4 * q + j] + acc1[1][4 * q + j] + (hf ? bhi : blo));
          }
          // hidden index inside the chunk: 32 wn + 8 q + 4 hf + j -> k-step 2 wn + (q >> 1), k-group q & 1
          char* dst = hbuf + ((wm * 4 + 2 * wn + (q >> 1)) * 3) * kFrag + (q & 1) * 512 + li * 16 + hf * 8;
          const xs::Split3 s0 = xs::split3_fast(h[0]), s1 = xs::split3_fast(h[1]), s2 = xs::split3_fast(h[2]),
                           s3 = xs::split3_fast(h[3]);
          *reinterpret_cast<uint2*>(dst) = make_uint2(xs::pack_hi16(s0.hi, s1.hi), xs::pack_hi16(s2.hi, s3.hi));
          *reinterpret_cast<uint2*>(dst + kFrag) = make_uint2(xs::pack_hi16(s0.mid, s1.mid), xs::pack_hi16(s2.mid, s3.mid));
          *reinterpret_cast<uint2*>(dst + 2 * kFrag) = make_uint2(xs::pack_hi16(s0.lo, s1.lo), xs::pack_hi16(s2.lo, s3.lo));
        }
      }
      slot = sn;
      ++g;
    });
  }
  wait_vm<0>();   // the surplus re-loads of the tail must have landed before this workgroup's LDS can be handed on
  final_epilogue<PROJ>(acc2, A, r0, wave, lane, reinterpret_cast<float*>(ring), [&](int t, int q) { return y1[t][q]; });
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


// y_w = x . W_w^T + b_w for nw stacked 256 -> 256 linear layers applied to the SAME rows (nw = 1: optionally followed by
// LayerNorm(res + y), the attention block's output projection with its residual + LayerNorm, dd:1102, 1326-1330; nw = 6: the
// decoder's six cross-attention value projections of the encoder output, dd:1048-1049) with the same machinery: the input
// panel is split once (hi / mid in LDS, lo in registers) and re-used by every weight, the weights run as 16 nw stages of one
// k-step ([8 n-blocks][3 pieces] = 24 KiB, 24 MFMAs per wave) through the three-slot DMA ring, the 64 x 256 output tile sits
// in four accumulators per wave; the stores of weight w drain while weight w + 1 is multiplied.
__global__ __launch_bounds__(256, 1) void proj_x6_kernel(FfnArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const panel = smem;
  char* const ring = smem + kPanel;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const unsigned lds_ring = (unsigned)reinterpret_cast<uintptr_t>((lds_char*)ring);
  const int r0 = blockIdx.x * kRows;
  const unsigned voff = lane * 16;
  const int nw = A.F;
  // stage (w, ks): fragments (nb, p) of k-step ks of weight w at (((8 w + nb) * 16 + ks) * 3 + p) KiB; a wave moves
  // n-blocks 2 wave, 2 wave + 1.  Past the last stage: re-load it (never read; keeps the wait counts immediates).
  auto src_of = [&](int w, int ks, int j) {
    if (ks >= kKS) { ks -= kKS; ++w; }
    if (w >= nw) { w = nw - 1; ks = kKS - 1; }
    return A.w2 + ((size_t)((8 * w + 2 * wave + j / 3) * kKS + ks) * 3 + j % 3) * kFrag;
  };
  auto issue = [&](int ks, int slot) {
    const unsigned dst = lds_ring + (unsigned)slot * kStage + (unsigned)wave * (NL * kFrag);
#pragma unroll
    for (int i = 0; i < NL; ++i) dma16s(src_of(0, ks, i), voff, dst + i * kFrag);
  };
  issue(0, 0);
  issue(1, 1);
  build_panel(A.x, A.ldx, A.M, r0, tid, lane, panel, ring + 2 * kStage, ring + 2 * kStage + kKS * kFrag);
  __syncthreads();
  bf16x8 lo[kKS];
  {
    const char* q = ring + 2 * kStage + wm * (kKS * kFrag);
#pragma unroll
    for (int ks = 0; ks < kKS; ++ks) lo[ks] = *reinterpret_cast<const bf16x8*>(q + ks * kFrag + (swz(lane, ks) << 4));
  }
  auto pfrag = [&](int ks, int piece) {
    return *reinterpret_cast<const bf16x8*>(panel + ((wm * kKS + ks) * 2 + piece) * kFrag + (swz(lane, ks) << 4));
  };
  const char* const pw = ring + (4 * wn * 3) * kFrag + lane * 16;
  auto frag = [](const char* p) { return *reinterpret_cast<const bf16x8*>(p); };
  bf16x8 w0[4][3], w1[4][3];
  wait_vm<NL>();
  wait_lgkm0();
  __builtin_amdgcn_s_barrier();
  issue(2, 2);
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int p = 0; p < 3; ++p) w0[u][p] = frag(pw + (u * 3 + p) * kFrag);
  bf16x8 ahi = pfrag(0, 0), amid = pfrag(0, 1);
  int slot = 0;   // ring slot of the stage being multiplied
#pragma unroll 1
  for (int w = 0; w < nw; ++w) {
    f32x16 acc2[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[t][r] = 0.f;
    static_for<kKS>([&](auto S) {
      constexpr int ks = decltype(S)::value;
      const int sn = slot == 2 ? 0 : slot + 1;
      wait_vm<NL>();
      wait_lgkm0();
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      const bf16x8 (&wc)[4][3] = (ks & 1) ? w1 : w0;
      bf16x8 (&wnx)[4][3] = (ks & 1) ? w0 : w1;
      const bf16x8 a0 = ahi, a1 = amid;
      const unsigned dst = lds_ring + (unsigned)slot * kStage + (unsigned)wave * (NL * kFrag);
      const char* const wn_src = pw + sn * kStage;
      static_for<24>([&](auto I) {
        constexpr int i = decltype(I)::value;
        constexpr int pwt[6] = {2, 0, 1, 1, 0, 0}, pat[6] = {0, 2, 1, 0, 1, 0};
        constexpr int term = i / 4, t = i % 4;
        if constexpr (pat[term] == 2) acc2[t] = mfma(wc[t][pwt[term]], lo[ks], acc2[t]);
        else acc2[t] = mfma(wc[t][pwt[term]], pat[term] == 0 ? a0 : a1, acc2[t]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr ((i & 1) && (i >> 1) < NL) {
          dma16s(src_of(w, ks + 3, i >> 1), voff, dst + (i >> 1) * kFrag);
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (i >= 12 && i < 18) {
          constexpr int j = 2 * (i - 12);
          wnx[j / 3][j % 3] = frag(wn_src + j * kFrag);
          wnx[(j + 1) / 3][(j + 1) % 3] = frag(wn_src + (j + 1) * kFrag);
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (i == 18) {   // hi / mid of the next k-step (the first one again for the next weight)
          ahi = pfrag((ks + 1) & (kKS - 1), 0);
          amid = pfrag((ks + 1) & (kKS - 1), 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      });
      slot = sn;
    });
    FfnArgs Aw = A;
    Aw.out = A.out + (size_t)w * A.M * kD;
    Aw.b2 = A.b2 != nullptr ? A.b2 + w * kD : nullptr;
    if (w + 1 == nw) wait_vm<0>();   // the surplus re-loads of the tail must have landed before the LDS is handed on
    final_epilogue<false>(acc2, Aw, r0, wave, lane, reinterpret_cast<float*>(ring), [](int, int) { return float4{}; });
  }
}

}  // namespace

extern "C" int egtr_ffn_x6_f32(egtr_stream_t stream, const float* x, int ldx, const void* w1_xs, const float* b1,
                               const void* w2_xs, const float* b2, const float* ln_gamma, const float* ln_beta, float eps,
                               const float* pos, int pos_rows, float* out, float* out_pos, int M, int d_model, int ffn_dim) {
  if (!x || !w1_xs || !b1 || !w2_xs || !b2 || !out || M <= 0 || ldx < d_model) return EGTR_E_ARG;
  if ((ln_gamma == nullptr) != (ln_beta == nullptr) || (out_pos && (!pos || pos_rows <= 0 || !ln_gamma))) return EGTR_E_ARG;
  if (d_model != kD || ffn_dim <= 0 || ffn_dim % 64 || (ldx & 3) || (reinterpret_cast<uintptr_t>(x) & 15) ||
      (reinterpret_cast<uintptr_t>(out) & 15) || (reinterpret_cast<uintptr_t>(w1_xs) & 15) ||
      (reinterpret_cast<uintptr_t>(w2_xs) & 15) || (reinterpret_cast<uintptr_t>(b2) & 15) ||
      (out_pos && (reinterpret_cast<uintptr_t>(out_pos) & 15)) || (pos && (reinterpret_cast<uintptr_t>(pos) & 15)))
    return EGTR_E_UNSUPPORTED;
  static unsigned long long lds_raised = 0;   // per-device
  if (int st = egtr_raise_dynamic_lds(reinterpret_cast<const void*>(ffn_x6_kernel<false>), kLds, &lds_raised)) return st;
  FfnArgs a{x, x, static_cast<const char*>(w1_xs), b1, static_cast<const char*>(w2_xs), b2, ln_gamma, ln_beta, pos, out,
            out_pos, M, ldx, ldx, ffn_dim, pos_rows, eps, nullptr, nullptr, nullptr, nullptr, 0.f};
  hipLaunchKernelGGL(ffn_x6_kernel<false>, dim3((M + kRows - 1) / kRows), dim3(256), kLds, static_cast<hipStream_t>(stream), a);
  return egtr_check_launch();
}

extern "C" int egtr_encoder_tail_x6_f32(egtr_stream_t stream, const float* context, int ldc, const float* hidden, int ldh,
                                        const void* wp_xs, const float* bp, const float* ln1_gamma, const float* ln1_beta,
                                        float eps1, const void* w1_xs, const float* b1, const void* w2_xs, const float* b2,
                                        const float* ln2_gamma, const float* ln2_beta, float eps2, const float* pos,
                                        int pos_rows, float* out, float* out_pos, int M, int d_model, int ffn_dim) {
  if (!context || !hidden || !wp_xs || !bp || !ln1_gamma || !ln1_beta || !w1_xs || !b1 || !w2_xs || !b2 || !ln2_gamma ||
      !ln2_beta || !out || M <= 0 || ldc < d_model || ldh < d_model)
    return EGTR_E_ARG;
  if (out_pos && (!pos || pos_rows <= 0)) return EGTR_E_ARG;
  if (d_model != kD || ffn_dim <= 0 || ffn_dim % 64 || (ldc & 3) || (ldh & 3)) return EGTR_E_UNSUPPORTED;
  for (const void* p : {(const void*)context, (const void*)hidden, wp_xs, (const void*)bp, (const void*)ln1_gamma,
                        (const void*)ln1_beta, w1_xs, w2_xs, (const void*)b2, (const void*)ln2_gamma, (const void*)ln2_beta,
                        (const void*)out, (const void*)out_pos, (const void*)pos})
    if (reinterpret_cast<uintptr_t>(p) & 15) return EGTR_E_UNSUPPORTED;
  static unsigned long long lds_raised = 0;   // per-device
  if (int st = egtr_raise_dynamic_lds(reinterpret_cast<const void*>(ffn_x6_kernel<true>), kLds, &lds_raised)) return st;
  FfnArgs a{context, hidden, static_cast<const char*>(w1_xs), b1, static_cast<const char*>(w2_xs), b2, ln2_gamma, ln2_beta,
            pos, out, out_pos, M, ldc, ldh, ffn_dim, pos_rows, eps2, static_cast<const char*>(wp_xs), bp, ln1_gamma,
            ln1_beta, eps1};
  hipLaunchKernelGGL(ffn_x6_kernel<true>, dim3((M + kRows - 1) / kRows), dim3(256), kLds, static_cast<hipStream_t>(stream), a);
  return egtr_check_launch();
}

extern "C" int egtr_proj_ln_x6_f32(egtr_stream_t stream, const float* x, int ldx, const void* w_xs, const float* bias,
                                   const float* residual, int ldr, const float* ln_gamma, const float* ln_beta, float eps,
                                   const float* pos, int pos_rows, float* out, float* out_pos, int M, int d_model) {
  if (!x || !w_xs || !bias || !out || M <= 0 || ldx < d_model) return EGTR_E_ARG;
  if ((ln_gamma == nullptr) != (ln_beta == nullptr) || (ln_gamma && (!residual || ldr < d_model)) ||
      (out_pos && (!pos || pos_rows <= 0 || !ln_gamma)))
    return EGTR_E_ARG;
  if (d_model != kD || (ldx & 3) || (ldr & 3) || (reinterpret_cast<uintptr_t>(x) & 15) ||
      (reinterpret_cast<uintptr_t>(out) & 15) || (reinterpret_cast<uintptr_t>(w_xs) & 15) ||
      (reinterpret_cast<uintptr_t>(bias) & 15) || (reinterpret_cast<uintptr_t>(residual) & 15) ||
      (out_pos && (reinterpret_cast<uintptr_t>(out_pos) & 15)) || (pos && (reinterpret_cast<uintptr_t>(pos) & 15)))
    return EGTR_E_UNSUPPORTED;
  constexpr int lds = kPanel + 3 * kStage + 8 * kFrag;   // + 8 KiB: parking place of the second row block's lo pieces
  static unsigned long long lds_raised = 0;   // per-device
  if (int st = egtr_raise_dynamic_lds(reinterpret_cast<const void*>(proj_x6_kernel), lds, &lds_raised)) return st;
  FfnArgs a{x, residual, nullptr, nullptr, static_cast<const char*>(w_xs), bias, ln_gamma, ln_beta, pos, out, out_pos,
            M, ldx, ldr, 1, pos_rows, eps};
  hipLaunchKernelGGL(proj_x6_kernel, dim3((M + kRows - 1) / kRows), dim3(256), lds, static_cast<hipStream_t>(stream), a);
  return egtr_check_launch();
}

extern "C" int egtr_proj_multi_x6_f32(egtr_stream_t stream, const float* x, int ldx, const void* w_xs, const float* bias,
                                      float* out, int M, int d_model, int num_weights) {
  if (!x || !w_xs || !out || M <= 0 || ldx < d_model || num_weights <= 0) return EGTR_E_ARG;
  if (d_model != kD || (ldx & 3) || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(out) & 15) ||
      (reinterpret_cast<uintptr_t>(w_xs) & 15) || (bias && (reinterpret_cast<uintptr_t>(bias) & 15)))
    return EGTR_E_UNSUPPORTED;
  constexpr int lds = kPanel + 3 * kStage + 8 * kFrag;
  static unsigned long long lds_raised = 0;   // per-device
  if (int st = egtr_raise_dynamic_lds(reinterpret_cast<const void*>(proj_x6_kernel), lds, &lds_raised)) return st;
  FfnArgs a{x, nullptr, nullptr, nullptr, static_cast<const char*>(w_xs), bias, nullptr, nullptr, nullptr, out, nullptr,
            M, ldx, 0, num_weights, 0, 0.f};
  hipLaunchKernelGGL(proj_x6_kernel, dim3((M + kRows - 1) / kRows), dim3(256), lds, static_cast<hipStream_t>(stream), a);
  return egtr_check_launch();
}
