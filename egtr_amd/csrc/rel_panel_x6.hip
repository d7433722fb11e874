// EGTR relation head (model/egtr.py:366-416), inference forward, as a row-panel kernel on the bf16 matrix cores with fp32-level
// accuracy: the successor of rel_head_fwd_x6 (rel_head.hip; same algebra -- separable gate logit, layer 1 folded into per-query
// tables uq / uk -- see the header there).  What changed is who streams the layer-2 weights and how often:
//
//   rel_head_fwd_x6: a tile of 32 pairs per workgroup, every wave pulls its half of W2 (192 KiB) from L2 into REGISTERS per
//   tile: 0.96 GB of L1 traffic at N = 200, 114 us, 0.25 of the bf16 peak.
//   here: a workgroup of four waves owns a PANEL of 64 pairs (8 subjects x 8 objects) of one of the two MLPs for the whole
//   chain, exactly like the encoder's feed-forward kernel (ffn_x6.hip):
//     * layer 1 (fp32 VALU: gates, gated sum over the slots, ReLU) builds the 64 x 256 hidden-1 panel ONCE, split into its
//       three bf16 pieces -- hi / mid as MFMA operand fragments in LDS (64 KiB), lo in registers;
//     * hidden-2 is walked in four chunks of 64 units: four weight stages of W2 (K = 256), bias + ReLU + split of the 64 x 64
//       chunk into LDS (24 KiB), then -- relation MLP -- ONE stage of W3 accumulating the chunk into the 64 x 64 output tile
//       (connectivity MLP: its single output is a VALU dot product on the accumulators);
//     * W2 / W3 arrive in the XS format (xs_format.h) and stream through a ring of three 24 KiB LDS stages filled by LDS-DMA
//       three stages ahead; every stage is 24 fragments and 24 MFMAs per wave, the program order inside a stage is pinned
//       (MFMAs interleaved with the DMA issues, the next stage's operand reads) as in ffn_x6.hip.
//   One W2 + W3 stream (480 KiB) now feeds 64 pairs with all four waves sharing it through LDS: 0.6 GB of LDS-DMA traffic
//   instead of 0.96 GB of register loads, and no wave waits on its own weight loads.
// The panel's fragments are XOR-swizzled (16-byte slot ^ k-step, swz()) so that layer 1 -- lane = channel quad, i.e. all 16
// k-steps of ONE row per store instruction -- writes them without bank conflicts; the MFMA operand read applies the same XOR.
// LDS: 64 (panel) + 24 (hidden-2 chunk) + 72 (ring) = 160 KiB, one workgroup per CU; grid = 2 x ceil(N/8)^2 x B panels
// (N = 200: 1250 = 4.9 rounds of 256 CUs), the longer relation-MLP panels first.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"
#include "x6_common.h"
#include "xs_format.h"

#ifdef REL_TIMING
#define REL_PHASE(k)                                                                           \
  do {                                                                                         \
    if (A.tdbg != nullptr && threadIdx.x == 0 && (blockIdx.x & 63) == 0)                        \
      A.tdbg[(blockIdx.x >> 6) * 16 + (k)] = (long long)__builtin_readcyclecounter();           \
    __builtin_amdgcn_sched_barrier(0);                                                         \
  } while (0)
#define REL_STAMP(k)                                                                           \
  do {                                                                                         \
    if (A.tdbg != nullptr && threadIdx.x == 0 && blockIdx.x == 0 && c == 0)                     \
      A.tdbg[1024 + s10 * 4 + (k)] = (long long)__builtin_readcyclecounter();                   \
    __builtin_amdgcn_sched_barrier(0);                                                         \
  } while (0)
#else
#define REL_PHASE(k)
#define REL_STAMP(k)
#endif

namespace {
using namespace x6;

constexpr int kHd = 256;                         // hidden width of both MLPs
constexpr int kKS = kHd / 16;                    // k-steps over a hidden layer
constexpr int kFrag = xs::kFragBytes;
constexpr int kPanel = 2 * kKS * 2 * kFrag;      // [row block 2][k-step 16][hi, mid]              64 KiB
constexpr int kHbuf = 2 * 4 * 3 * kFrag;         // XS(hidden-2 chunk): [row block 2][k-step 4][3]  24 KiB
constexpr int kStage = 24 * kFrag;               // one weight stage                                24 KiB
constexpr int kLds = kPanel + kHbuf + 3 * kStage;
static_assert(kLds == 160 * 1024, "the whole LDS of a CU");
constexpr int NL = 6;                            // DMA instructions per wave and stage
constexpr int kOutStride = 65;                   // floats per pair of the staged relation tile
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct RelArgs {
  const float* gate_q;   // [B N, T]
  const float* gate_k;
  const float* uq;       // [B N, T, 2 * 256]: W1[:, :d] q^ of the relation MLP, then of the connectivity MLP
  const float* uk;
  const float* b1;       // [2 * 256]
  const char* w2[2];     // XS(W2 [256, 256]) of the relation / connectivity MLP
  const float* b2[2];
  const char* w3r;       // XS(W3 of the relation MLP, rows zero-padded to 64)
  const float* b3r;      // [R]
  const float* w3c;      // [256]
  const float* b3c;      // [1]
  const float* triplet;  // [C1, C1, R] or null
  const int64_t* node_cls;
  int B, N, R, C1;
  float* rel;            // [B, N, N, R]
  float* conn;           // [B, N, N]
  float* gate_mean;      // [T] (pre-zeroed) or null
  int apply_sigmoid;
  long long* tdbg;       // development: cycle stamps (REL_TIMING builds, tools/rel_panel_bench.hip), else null
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// logical 16-byte slot x (= row + 32 * k-group) of a panel fragment of k-step ks -> where it is stored: the layer-1 writer
// (lane = channel quad: 16 k-steps x 2 k-groups x 2 halves of ONE row per store instruction) hits every bank once, the MFMA
// operand read (slot = lane) stays a permutation of the fragment's 64 slots inside each ds_read_b128 lane group.
__device__ __forceinline__ int swz(int x, int ks) { return x ^ (ks & 15) ^ ((x >> 5) << 2); }

// Layer 1 for the panel: wave w builds the 16 pairs of subjects 4 (w >> 1) .. + 3 x objects 4 (w & 1) .. + 3 (rows 8 iu + 4 (w & 1)
// + ju of row block w >> 1), lane l the channels 4 l .. 4 l + 3:
//     h1 = relu(b1 + sum_t g[i,j,t] (uq[i,t,:] + uk[j,t,:]))                  (egtr.py:380-398 folded, see rel_head.hip).
// The 16 table rows of a slot t (8 subjects, 8 objects; 1 KiB each) are needed by several waves: they come in ONCE per
// workgroup, by LDS-DMA (wave w moves rows 4 w .. 4 w + 3), into 16 KiB buffers carved out of the LDS that is idle during the
// build (the hidden-2 chunk buffer, ring slot 2 and the panel, which is written last) -- 112 KiB of table reads per panel
// instead of 280 KiB of per-wave register loads.
// At the end the same LDS takes the lo pieces (lo0 = hbuf: row block 0, lo1 = slot2: row block 1) for the hand-over to registers.
template <int T, class Fn>
__device__ __forceinline__ void build_h1(const RelArgs& A, int mlp, int b, int i0, int j0, int wave, int lane, char* panel,
                                         char* hbuf, char* slot2, Fn&& after_slots) {
  const int N = A.N;
  const unsigned voff = lane * 16;
  constexpr int kDepth = 7;   // table-row buffers
  // the gate logits of this wave's pairs (lane l & 15 <-> subject iu = l >> 2, object ju = l & 3) are requested FIRST, as asm
  // loads with their own counted wait below: in front of the 28 row DMAs in the memory pipeline instead of behind them
  const int gi_raw = i0 + 4 * (wave >> 1) + ((lane >> 2) & 3), gj_raw = j0 + 4 * (wave & 1) + (lane & 3);
  const int gi = min(gi_raw, N - 1), gj = min(gj_raw, N - 1);
  float gq[T], gk[T];
  {
    const float* pq = A.gate_q + ((size_t)b * N + gi) * T;
    const float* pk = A.gate_k + ((size_t)b * N + gj) * T;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      asm volatile("global_load_dword %0, %1, off" : "=v"(gq[t]) : "v"(pq + t) : "memory");
      asm volatile("global_load_dword %0, %1, off" : "=v"(gk[t]) : "v"(pk + t) : "memory");
    }
  }
  // buffer k < 3: subject rows / object rows in the idle chunk buffer and ring slot 2; buffers 3 .. 6: the panel itself, which
  // is only written after the last slot has been consumed
  auto sub_of = [&](int k) { return k == 0 ? hbuf : k == 1 ? slot2 : k == 2 ? hbuf + 16 * kFrag : panel + (k - 3) * 16 * kFrag; };
  auto obj_of = [&](int k) {
    return k == 0 ? hbuf + 8 * kFrag : k == 1 ? slot2 + 8 * kFrag : k == 2 ? slot2 + 16 * kFrag : panel + ((k - 3) * 16 + 8) * kFrag;
  };
  const float* const tab = wave < 2 ? A.uq : A.uk;
  const int q0 = (wave < 2 ? i0 : j0) + 4 * (wave & 1);
  auto issue_rows = [&](int t, int k) {
    const unsigned dst = (unsigned)reinterpret_cast<uintptr_t>((lds_char*)(wave < 2 ? sub_of(k) : obj_of(k))) + (wave & 1) * 4 * kFrag;
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const float* src = tab + (((size_t)b * N + min(q0 + x, N - 1)) * T + t) * (2 * kHd) + mlp * kHd;
      dma16s(reinterpret_cast<const char*>(src), voff, dst + x * kFrag);
    }
  };
  // every slot's rows are requested up front (T <= 7; beyond that the buffer of slot t - 1 takes slot t + 6): the tables sit
  // in L2 / MALL at ~2 k cycles, a slot's arithmetic is ~0.6 k -- two slots of run-ahead had left the build latency-bound
#pragma unroll
  for (int t = 0; t < (T < kDepth ? T : kDepth); ++t) issue_rows(t, t);

  REL_PHASE(8);
  // wait for the gate logits only (the row DMAs issued after them stay in flight)
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(gq[0]), "+v"(gk[0]) : "i"(4 * (T < kDepth ? T : kDepth)));
#pragma unroll
  for (int t = 1; t < T; ++t) asm volatile("" : "+v"(gq[t]), "+v"(gk[t]));
  float g[T];
#pragma unroll
  for (int t = 0; t < T; ++t) g[t] = sigmoidf_(gq[t] + gk[t]);
  if (A.gate_mean != nullptr && mlp == 0) {
    const bool valid = lane < 16 && gi_raw < N && gj_raw < N;
    const float inv = 1.f / ((float)A.B * (float)N * (float)N);
#pragma unroll
    for (int t = 0; t < T; ++t) {
      float v = valid ? g[t] : 0.f;
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o);
      if (lane == 0) unsafeAtomicAdd(A.gate_mean + t, v * inv);
    }
  }
  REL_PHASE(9);
  const float4 bias4 = reinterpret_cast<const float4*>(A.b1 + mlp * kHd)[lane];
  f32x2 acc[16][2];
#pragma unroll
  for (int pp = 0; pp < 16; ++pp) {
    acc[pp][0] = f32x2{bias4.x, bias4.y};
    acc[pp][1] = f32x2{bias4.z, bias4.w};
  }
  static_for<T>([&](auto TT) {
    constexpr int t = decltype(TT)::value;
    // this wave's rows of slot t have landed (younger slots may be in flight); after the barrier so have everybody's, and
    // everybody is done reading slot t - 1, whose buffer takes slot t + kDepth - 1
    constexpr int last_issued = t == 0 ? kDepth - 1 : t + kDepth - 2;
    wait_vm<4 * ((last_issued < T - 1 ? last_issued : T - 1) - t)>();
    wait_lgkm0();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (t == 0) REL_PHASE(11);
    if constexpr (t == 1) REL_PHASE(12);
    if constexpr (t >= 1 && t + kDepth - 1 < T) {
      issue_rows(t + kDepth - 1, (t - 1) % kDepth);
      __builtin_amdgcn_sched_barrier(0);
    }
    float4 ua[4], kk[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      ua[x] = *reinterpret_cast<const float4*>(sub_of(t % kDepth) + (4 * (wave >> 1) + x) * kFrag + lane * 16);
      kk[x] = *reinterpret_cast<const float4*>(obj_of(t % kDepth) + (4 * (wave & 1) + x) * kFrag + lane * 16);
    }
#pragma unroll
    for (int iu = 0; iu < 4; ++iu)
#pragma unroll
      for (int ju = 0; ju < 4; ++ju) {
        const float gt = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g[t]), iu * 4 + ju));
        const f32x2 gg = {gt, gt};
        const f32x2 u01 = {ua[iu].x + kk[ju].x, ua[iu].y + kk[ju].y}, u23 = {ua[iu].z + kk[ju].z, ua[iu].w + kk[ju].w};
        acc[iu * 4 + ju][0] = gg * u01 + acc[iu * 4 + ju][0];
        acc[iu * 4 + ju][1] = gg * u23 + acc[iu * 4 + ju][1];
      }
  });
  REL_PHASE(10);
  after_slots();   // nothing of the build is in flight any more: the first weight stages start now and land during the split
  wait_lgkm0();
  __builtin_amdgcn_s_barrier();   // every wave is done with the table buffers: they take the lo pieces now
  __builtin_amdgcn_sched_barrier(0);
  const int rb = wave >> 1;
  const int ks = lane >> 2, half = (lane >> 1) & 1;
#pragma unroll
  for (int iu = 0; iu < 4; ++iu)
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {      // two pairs = eight values per staged split
      float x[8];
      uint2 phi[2], pmid[2], plo[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const f32x2 a01 = acc[iu * 4 + 2 * jp + u][0], a23 = acc[iu * 4 + 2 * jp + u][1];
        x[4 * u + 0] = a01.x; x[4 * u + 1] = a01.y; x[4 * u + 2] = a23.x; x[4 * u + 3] = a23.y;
      }
      xs::split_staged<8, true>(x, phi, pmid, plo);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int r = iu * 8 + 4 * (wave & 1) + 2 * jp + u;                   // row inside the row block
        const int off = (swz(r + 32 * half, ks) << 4) + (lane & 1) * 8;
        char* p = panel + ((rb * kKS + ks) * 2) * kFrag + off;
        *reinterpret_cast<uint2*>(p) = phi[u];
        *reinterpret_cast<uint2*>(p + kFrag) = pmid[u];
        *reinterpret_cast<uint2*>((rb == 0 ? hbuf : slot2) + ks * kFrag + off) = plo[u];
      }
    }
}

// MLP = 0: relation (five stages per chunk: 4 x W2, 1 x W3), MLP = 1: connectivity (four stages per chunk)
template <int T, int MLP>
__device__ __forceinline__ void rel_panel_body(const RelArgs& A, int tile, char* smem) {
  constexpr int SPC = MLP == 0 ? 5 : 4;   // stages per chunk
  constexpr int kChunks = kHd / 64;
  char* const panel = smem;
  char* const hbuf = smem + kPanel;
  char* const ring = smem + kPanel + kHbuf;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, hf = lane >> 5;
  const unsigned lds_ring = (unsigned)reinterpret_cast<uintptr_t>((lds_char*)ring);
  const int N = A.N;
  const int tn = (N + 7) >> 3;
  const int b = tile / (tn * tn);
  const int trem = tile - b * tn * tn;
  const int i0 = (trem / tn) * 8, j0 = (trem % tn) * 8;

  const unsigned voff = lane * 16;
  const char* const w2 = A.w2[MLP];
  const int crot = tile & (kChunks - 1);   // neighbouring panels stream different weight fragments at any moment
  auto chunk_of = [&](int c) { return (c + crot) & (kChunks - 1); };
  // stage (chunk cc, s): s < 4: W2 rows 64 cc .., k-steps 4 s .. ([2 n-blocks][4 k-steps][3 pieces]); s == 4: W3, all 64
  // outputs, k-steps 4 cc .. of the hidden-2 dimension (the same image).  Wave w moves fragments 6 w .. 6 w + 5: one 6 KiB run.
  auto src_of = [&](int cc, int s) {
    return s < 4 ? w2 + ((size_t)((2 * cc + (wave >> 1)) * kKS + 4 * s + 2 * (wave & 1)) * 3) * kFrag
                 : A.w3r + ((size_t)((wave >> 1) * kKS + 4 * cc + 2 * (wave & 1)) * 3) * kFrag;
  };
  auto issue = [&](int c, int s, int slot) {
    const unsigned dst = lds_ring + (unsigned)slot * kStage + (unsigned)wave * (NL * kFrag);
    const char* src = src_of(chunk_of(c), s);
#pragma unroll
    for (int i = 0; i < NL; ++i) dma16s(src + i * kFrag, voff, dst + i * kFrag);
  };
  REL_PHASE(0);
  // Layer-2 bias (and, connectivity MLP, the output weights) of the hidden units this lane will hold, for the four chunks in
  // the order this workgroup walks them: unit 64 cc + 32 wn + 8 q + 4 hf + j <-> accumulator register 4 q + j.  Loaded here,
  // first thing, and pinned in registers before the weight stream starts: nothing inside the main loop may touch the vmcnt
  // queue the DMA waits are counted on.  (Scalar loads issued a few stages ahead, as ffn_x6.hip does for its 16 chunks, are
  // not safe here: under this kernel's SGPR pressure the compiler spills the destination registers before the data lands.)
  float bv[kChunks][16], wv[MLP == 1 ? kChunks : 1][16];
#pragma unroll
  for (int cl = 0; cl < kChunks; ++cl)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = 64 * chunk_of(cl) + 32 * wn + 8 * q + 4 * hf;
      const float4 v = *reinterpret_cast<const float4*>(A.b2[MLP] + n);
      bv[cl][4 * q + 0] = v.x; bv[cl][4 * q + 1] = v.y; bv[cl][4 * q + 2] = v.z; bv[cl][4 * q + 3] = v.w;
      if constexpr (MLP == 1) {
        const float4 u = *reinterpret_cast<const float4*>(A.w3c + n);
        wv[cl][4 * q + 0] = u.x; wv[cl][4 * q + 1] = u.y; wv[cl][4 * q + 2] = u.z; wv[cl][4 * q + 3] = u.w;
      }
    }
  __builtin_amdgcn_sched_barrier(0);

  build_h1<T>(A, MLP, b, i0, j0, wave, lane, panel, hbuf, ring + 2 * kStage, [&]() {
#pragma unroll
    for (int cl = 0; cl < kChunks; ++cl)
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        asm volatile("" : "+v"(bv[cl][k]));
        if constexpr (MLP == 1) asm volatile("" : "+v"(wv[cl][k]));
      }
    __builtin_amdgcn_sched_barrier(0);
    issue(0, 0, 0);
    issue(0, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
  });
  REL_PHASE(1);
  __syncthreads();
  REL_PHASE(2);
  bf16x8 lo[kKS];
  {
    const char* q = (wm == 0 ? hbuf : ring + 2 * kStage);
#pragma unroll
    for (int ks = 0; ks < kKS; ++ks) lo[ks] = *reinterpret_cast<const bf16x8*>(q + ks * kFrag + (swz(lane, ks) << 4));
  }

  f32x16 acc1[2], racc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) racc[t][r] = 0.f;
  float cacc = 0.f;

  auto frag = [](const char* p) { return *reinterpret_cast<const bf16x8*>(p); };
  // hi / mid fragment of k-step ks of this wave's 32 panel rows (swizzled)
  auto pfrag = [&](int ks, int piece) {
    return frag(panel + ((wm * kKS + ks) * 2 + piece) * kFrag + (swz(lane, ks) << 4));
  };
  const char* const ph = hbuf + (wm * 4 * 3) * kFrag + lane * 16;      // hidden-2 chunk fragments of this wave's 32 rows
  const char* const pw = ring + (4 * wn * 3) * kFrag + lane * 16;      // stage image: fragment (k-step u, piece p) of n-block wn

  bf16x8 w0[4][3], w1[4][3], a0[4][2], a1[4][2], alo[4];

  auto stage = [&](auto S, const bf16x8 (&w)[4][3], bf16x8 (&wnx)[4][3], const bf16x8 (&a)[4][2], bf16x8 (&anx)[4][2], int c,
                   int slot_next, int slot_fill) {
    constexpr int s10 = decltype(S)::value, s = s10 % SPC;
    const char* const wn_src = pw + slot_next * kStage;
    const unsigned dst = lds_ring + (unsigned)slot_fill * kStage + (unsigned)wave * (NL * kFrag);
    constexpr int s3 = (s10 + 3) % SPC;
    int c3 = c + (s10 + 3) / SPC;
    c3 = chunk_of(c3 >= kChunks ? kChunks - 1 : c3);     // past the last stage: re-load a stage (never read)
    const char* const src = src_of(c3, s3);
    static_for<24>([&](auto I) {
      constexpr int i = decltype(I)::value;
      constexpr int pwt[6] = {2, 0, 1, 1, 0, 0}, pat[6] = {0, 2, 1, 0, 1, 0};   // the six cross terms, small ones first
      constexpr int pair = i / 12, term = (i % 12) / 2, par = i & 1, kl = 2 * pair + par;
      if constexpr (s < 4) {
        if constexpr (pat[term] == 2) acc1[par] = mfma(w[kl][pwt[term]], lo[4 * s + kl], acc1[par]);
        else acc1[par] = mfma(w[kl][pwt[term]], a[kl][pat[term]], acc1[par]);
      } else {
        if constexpr (pat[term] == 2) racc[par] = mfma(w[kl][pwt[term]], alo[kl], racc[par]);
        else racc[par] = mfma(w[kl][pwt[term]], a[kl][pat[term]], racc[par]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr ((i & 1) && (i >> 1) < NL) {
        dma16s(src + (i >> 1) * kFrag, voff, dst + (i >> 1) * kFrag);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (i >= 12 && i < 18) {
        constexpr int j = 2 * (i - 12);
        wnx[j / 3][j % 3] = frag(wn_src + j * kFrag);
        wnx[(j + 1) / 3][(j + 1) % 3] = frag(wn_src + (j + 1) * kFrag);
        __builtin_amdgcn_sched_barrier(0);
      }
      // operand A of the next stage when that is a W2 stage: the read-only panel.  (A W3 stage reads the hidden-2 chunk
      // behind its own barrier.)
      constexpr int sx = (s10 + 1) % SPC;
      if constexpr (i >= 18 && i < 22 && sx < 4) {
        constexpr int k2 = i - 18;
        anx[k2][0] = pfrag(4 * sx + k2, 0);
        anx[k2][1] = pfrag(4 * sx + k2, 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    });
  };

  wait_vm<NL>();                      // stage 0 (stage 1 in flight)
  wait_lgkm0();
  __builtin_amdgcn_s_barrier();
  issue(0, 2, 2);
#pragma unroll
  for (int u = 0; u < 4; ++u) {
#pragma unroll
    for (int p = 0; p < 3; ++p) w0[u][p] = frag(pw + (u * 3 + p) * kFrag);
    a0[u][0] = pfrag(u, 0);
    a0[u][1] = pfrag(u, 1);
  }
  int slot = 0;   // ring slot of the stage being multiplied
  REL_PHASE(3);
#pragma unroll 1
  for (int c = 0; c < kChunks; c += 2) {
    static_for<2 * SPC>([&](auto S) {
      constexpr int s10 = decltype(S)::value, s = s10 % SPC, cl = s10 / SPC;
      const int sn = slot == 2 ? 0 : slot + 1;
      REL_STAMP(0);
      wait_vm<NL>();
      REL_STAMP(1);
      wait_lgkm0();
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      REL_STAMP(2);
      bf16x8 (&acur)[4][2] = (s10 & 1) ? a1 : a0;
      if constexpr (s == 4) {          // the hidden-2 chunk was written by the stage before: visible after this barrier
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) {
          acur[k2][0] = frag(ph + (k2 * 3 + 0) * kFrag);
          acur[k2][1] = frag(ph + (k2 * 3 + 1) * kFrag);
          alo[k2] = frag(ph + (k2 * 3 + 2) * kFrag);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (s == 0) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc1[t][r] = 0.f;
      }
      if constexpr ((s10 & 1) == 0) stage(S, w0, w1, a0, a1, c, sn, slot);
      else stage(S, w1, w0, a1, a0, c, sn, slot);
      REL_STAMP(3);
      if constexpr (s == 3) {
        // bias + ReLU of the wave's 32 pairs x 32 hidden-2 units; hidden index inside the chunk: 32 wn + 8 q + 4 hf + j
        float h[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) h[k] = acc1[0][k] + acc1[1][k];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 16; ++k) h[k] += c == 0 ? bv[cl][k] : bv[2 + cl][k];
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MLP == 0) {
          // -> XS fragments of the chunk: register group q <-> k-step 2 wn + (q >> 1), k-group q & 1
          uint2 phi[4], pmid[4], plo[4];
          xs::split_staged<16, true>(h, phi, pmid, plo);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            char* dst = hbuf + ((wm * 4 + 2 * wn + (q >> 1)) * 3) * kFrag + (q & 1) * 512 + li * 16 + hf * 8;
            *reinterpret_cast<uint2*>(dst) = phi[q];
            *reinterpret_cast<uint2*>(dst + kFrag) = pmid[q];
            *reinterpret_cast<uint2*>(dst + 2 * kFrag) = plo[q];
          }
        } else {
          // connectivity output layer (one unit, egtr.py:414-416): dot product with the chunk's 32 weights of this wave
          float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int k = 0; k < 16; ++k) part[k & 3] += egtr_relu(h[k]) * (c == 0 ? wv[cl][k] : wv[2 + cl][k]);
          cacc += (part[0] + part[1]) + (part[2] + part[3]);
        }
      }
      slot = sn;
    });
  }
  REL_PHASE(4);
  wait_vm<0>();   // the surplus re-loads of the tail must have landed before the ring is re-used / handed on
  __syncthreads();
  REL_PHASE(5);

  if constexpr (MLP == 1) {
    float* red = reinterpret_cast<float*>(ring);
    cacc += __shfl_xor(cacc, 32);
    if (wn == 1 && hf == 0) red[wm * 32 + li] = cacc;
    __syncthreads();
    if (wn == 0 && hf == 0) {
      const int pp = wm * 32 + li, i = i0 + (pp >> 3), j = j0 + (pp & 7);
      if (i < N && j < N) {
        const float v = cacc + red[wm * 32 + li] + A.b3c[0];
        A.conn[((size_t)b * N + i) * N + j] = A.apply_sigmoid ? sigmoidf_(v) : v;
      }
    }
  } else {
    // relation tile: staged through LDS, written as runs of R floats per pair with the output bias and the Neural-Motifs
    // frequency bias triplet_dist[cls_i, cls_j, :] added on the way out (egtr.py:405-413)
    float* s_out = reinterpret_cast<float*>(ring);
#pragma unroll
    for (int r = 0; r < 16; ++r)
      s_out[(wm * 32 + li) * kOutStride + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf] = racc[0][r] + racc[1][r];
    __syncthreads();
    const int R = A.R;
    const float b3 = lane < R ? A.b3r[lane] : 0.f;
    // wave w writes pairs w, w + 4, ..: all sixteen staged values first (independent LDS reads), then the stores
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = s_out[(wave + 4 * k) * kOutStride + lane];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int pp = wave + 4 * k, i = i0 + (pp >> 3), j = j0 + (pp & 7);
      if (i >= N || j >= N || lane >= R) continue;
      const size_t qi = (size_t)b * N + i, kj = (size_t)b * N + j;
      float y = v[k] + b3;
      if (A.triplet != nullptr) y += A.triplet[((size_t)A.node_cls[qi] * A.C1 + (size_t)A.node_cls[kj]) * R + lane];
      A.rel[(qi * N + j) * R + lane] = A.apply_sigmoid ? sigmoidf_(y) : y;
    }
  }
}

template <int T>
__global__ __launch_bounds__(256, 1) void rel_panel_x6_kernel(RelArgs A, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int u = blockIdx.x;
  if (u < ntiles) rel_panel_body<T, 0>(A, u, smem);
  else rel_panel_body<T, 1>(A, u - ntiles, smem);
  REL_PHASE(6);
}

template <int T>
int launch_panel(hipStream_t st, const RelArgs& a, int ntiles) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(rel_panel_x6_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kLds) != hipSuccess)
      return 1;
    attr_set = true;
  }
  hipLaunchKernelGGL((rel_panel_x6_kernel<T>), dim3(2 * ntiles), dim3(256), kLds, st, a, ntiles);
  return 0;
}

}  // namespace

long long* g_rel_tdbg = nullptr;   // development hook (tools/rel_panel_bench.hip)

extern "C" int egtr_rel_head_forward_panel_x6_f32(egtr_stream_t stream, const float* gate_q, const float* gate_k,
                                                  const float* uq, const float* uk, const float* b1, const void* w2_xs_rel,
                                                  const float* b2r, const void* w3_xs_rel, const float* b3r,
                                                  const void* w2_xs_conn, const float* b2c, const float* w3c, const float* b3c,
                                                  const float* triplet_dist, const int64_t* node_cls, int batch,
                                                  int num_query, int num_slots, int hidden, int num_rel, int num_cls_plus1,
                                                  float* rel_logits, float* conn_logits, float* gate_mean,
                                                  int apply_sigmoid) {
  if (!gate_q || !gate_k || !uq || !uk || !b1 || !w2_xs_rel || !b2r || !w3_xs_rel || !b3r || !w2_xs_conn || !b2c || !w3c ||
      !b3c || !rel_logits || !conn_logits)
    return EGTR_E_ARG;
  if (triplet_dist != nullptr && node_cls == nullptr) return EGTR_E_ARG;
  if (batch <= 0 || num_query <= 0 || num_slots <= 0 || num_rel <= 0) return EGTR_E_ARG;
  if (hidden != kHd || num_rel > 64 || num_slots > 9) return EGTR_E_UNSUPPORTED;
  for (const void* p : {(const void*)uq, (const void*)uk, (const void*)b1, (const void*)w2_xs_rel, (const void*)w3_xs_rel,
                        (const void*)w2_xs_conn, (const void*)b2r, (const void*)b2c, (const void*)w3c})
    if (reinterpret_cast<uintptr_t>(p) & 15) return EGTR_E_UNSUPPORTED;
  const long long tn = (num_query + 7) / 8, ntiles = (long long)batch * tn * tn;
  if (2 * ntiles >= (1ll << 31)) return EGTR_E_UNSUPPORTED;
  RelArgs a{gate_q, gate_k, uq, uk, b1, {static_cast<const char*>(w2_xs_rel), static_cast<const char*>(w2_xs_conn)},
            {b2r, b2c}, static_cast<const char*>(w3_xs_rel), b3r, w3c, b3c, triplet_dist, node_cls, batch, num_query,
            num_rel, num_cls_plus1, rel_logits, conn_logits, gate_mean, apply_sigmoid, g_rel_tdbg};
  hipStream_t st = static_cast<hipStream_t>(stream);
  int rc = 0;
#define EGTR_TP(TT) case TT: rc = launch_panel<TT>(st, a, (int)ntiles); break;
  switch (num_slots) {
    EGTR_TP(1) EGTR_TP(2) EGTR_TP(3) EGTR_TP(4) EGTR_TP(5) EGTR_TP(6) EGTR_TP(7) EGTR_TP(8) EGTR_TP(9)
    default: return EGTR_E_UNSUPPORTED;
  }
#undef EGTR_TP
  (void)rc;
  return egtr_check_launch();
}
