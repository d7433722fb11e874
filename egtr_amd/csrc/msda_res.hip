// MSDA forward, "resident coarse levels" variant for gfx950.
//
// Why: the wave-per-query kernel (msda.hip) is bound by the vector-L1 gather rate: 4 corners x 16
// samples x 128 B per (query, head) = 841 MB per encoder launch against a measured L1 ceiling of 30.5 TB/s.  Half of
// those samples (levels 2 and 3 of the 4-level pyramid) address only 768 of the 12 537 pixels.  Here a workgroup is
// bound to ONE head and keeps that head's slice of the coarsest levels -- as many as fit, 128 B per pixel -- resident
// in LDS for its whole life (98 KB at 600x1000); their corners are served by ds_read_b128 (256 B/clk/CU) and only the
// fine levels go through L1.  No windows, no bounding boxes, no fallbacks: any sampling location of a resident level is
// in LDS, so the kernel is as robust as the wave-per-query one and works for any query set.
//
// Mapping: 12 waves per workgroup, one workgroup per CU (persistent), workgroup b -> XCD b % 8 (dispatch order);
// heads are interleaved so that every XCD serves all 8 heads (a single head per XCD would concentrate its reads on
// one 128-B sub-line of every 1 KiB pixel row = a quarter of the memory channels).  The query groups of a head are
// split into contiguous ranges over the workgroups of that head.  A wave owns 8 queries x 1 head per pass:
// lane = (query j = lane >> 3, channel quad c = lane & 7), so 8 lanes read one aligned 128-B line per corner, exactly as
// in the wave-per-query kernel; lane (j, c) computes the geometry of samples 2c, 2c+1 of query j and stages
// {4 corner offsets, 4 bilinear x attention weights} records in LDS ([wave][query][sample], padded to 17 entries).
// LDS rows are 128 B; the two 64-B halves of a row are swapped when bit 1 of the pixel index is set, which halves the
// 2-way bank conflicts between the 4 half-rows that one ds_read_b128 lane group touches.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"
#include "msda_common.h"

using namespace egtr_msda;

namespace {

constexpr int kResPx = 832;   // LDS-resident pixels of one head (128 B each)
constexpr int kQ = 8;         // queries per wave pass
constexpr int kRec = 17;      // record entries per query (16 samples + 1 pad)

// kRW waves per workgroup; GCH = global-memory samples (x 4 corners x 16 B per lane) requested per batch.
template <int kRW, int GCH>
__global__ __launch_bounds__(kRW * 64) void msda_fwd_res_f32(
    const float* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
    const float* __restrict__ loc, const float* __restrict__ attn, float* __restrict__ out, int B, int Lq, int S,
    int L, int P) {
  __shared__ __attribute__((aligned(16))) float4 s_val[kResPx * 8];
  __shared__ __attribute__((aligned(16))) int4 s_off[kRW * kQ * kRec];
  __shared__ __attribute__((aligned(16))) float4 s_w[kRW * kQ * kRec];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane >> 3, c = lane & 7;
  LevelGeom G;
  load_geom(shapes, lsi, L, G);

  // Resident levels: the coarsest levels, as long as the tail [start_l, S) of the pixel list fits the budget and the
  // levels are laid out back to back (anything else: nothing is resident, every corner comes from global memory).
  int res_lvl = L, res_px0 = S;
  {
    int end = S;
    for (int l = L - 1; l >= 0; --l) {
      const int st = SEL_S(G, l), n = SEL_H(G, l) * SEL_W(G, l);
      if (st + n != end || S - st > kResPx) break;
      res_lvl = l;
      res_px0 = st;
      end = st;
    }
  }
  const int nres = S - res_px0;

  // workgroup -> (head, part): XCD x = b % 8 serves the x-th contiguous eighth of the query groups (one horizontal
  // stripe of every level in its private L2) for all 8 heads; the stripe is split again over the XCD's workgroups of
  // the same head.
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int head = idx & 7;
  const int nsub = gridDim.x >> 6;
  const int nparts = nsub * 8;
  const int part = xcd * nsub + (idx >> 3);
  // Passes of this wave: a pass is 8 consecutive queries; the waves stride over the workgroup's contiguous range of such
  // groups.  (Giving the 12 waves the 12 rows of an 8-wide tile instead -- a compact L1 footprint -- was measured
  // slower, 37 vs 32 us: 15 % padding and 5-vs-4 tiles per workgroup cost more than the L1 misses saved.)
  const int nunits = (Lq + kQ - 1) / kQ;
  const int u0 = (int)(((long long)part * nunits) / nparts), u1 = (int)(((long long)(part + 1) * nunits) / nparts);
  const int npass = (u1 - u0 - wave + kRW - 1) / kRW;
  auto pass_query = [&](int i) -> int {  // query of lane (j, *) in the wave's i-th pass; -1: none
    if (i >= npass) return -1;
    const int qq = (u0 + wave + i * kRW) * kQ + j;
    return qq < Lq ? qq : -1;
  };

  int4* my_off = s_off + wave * (kQ * kRec) + j * kRec;
  float4* my_w = s_w + wave * (kQ * kRec) + j * kRec;
  const int resbias = res_px0 * 128 + head * 16;  // (byte offset >> 3) of resident pixel 0 of this head
  const char* lds_val = reinterpret_cast<const char*>(s_val);

  for (int b = 0; b < B; ++b) {
    const char* vbase = reinterpret_cast<const char*>(value) + (size_t)b * S * 1024;
    if (b > 0) __syncthreads();  // everyone is done reading the previous image's resident levels
    for (int i = tid; i < nres * 8; i += kRW * 64) {
      const int px = i >> 3, cc = i & 7;
      const float4 v = *reinterpret_cast<const float4*>(vbase + (size_t)(res_px0 + px) * 1024 + head * 128 + cc * 16);
      s_val[px * 8 + (cc ^ (((px >> 1) & 1) << 2))] = v;
    }
    __syncthreads();

    // loc / attn (read once, streamed from HBM) are requested kPF passes ahead
    constexpr int kPF = 2;
    float4 lcq[kPF];
    float2 awq[kPF];
#pragma unroll
    for (int u = 0; u < kPF; ++u) {
      lcq[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      awq[u] = make_float2(0.f, 0.f);
    }
    auto fetch = [&](int i, float4& l4, float2& a2) {
      const int qq = pass_query(i);
      if (i < npass) {
        const size_t r = ((size_t)b * Lq + max(qq, 0)) * 8 + head;
        l4 = reinterpret_cast<const float4*>(loc)[r * 8 + c];
        a2 = reinterpret_cast<const float2*>(attn)[r * 8 + c];
      }
    };
#pragma unroll
    for (int u = 0; u < kPF; ++u) fetch(u, lcq[u], awq[u]);
    const int nglob = res_lvl * P;  // samples [0, nglob) read global memory, [nglob, 16) the resident levels
    for (int ib = 0; ib < npass; ib += kPF) {
#pragma unroll
     for (int u = 0; u < kPF; ++u) {
      const int i = ib + u;
      if (i >= npass) break;
      const int q = pass_query(i);
      const bool live = q >= 0;
      const size_t row = ((size_t)b * Lq + max(q, 0)) * 8 + head;
      const float4 lc = lcq[u];
      const float2 aw = awq[u];
      fetch(i + kPF, lcq[u], awq[u]);

      // stage 1: lane (j, c) -> samples 2c, 2c+1 of query j
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int s = 2 * c + k;
        const int lvl = s / P;
        const SampleGeom sg = sample_geom<1024, 128>(k ? lc.z : lc.x, k ? lc.w : lc.y, SEL_H(G, lvl), SEL_W(G, lvl),
                                                     SEL_S(G, lvl), head);
        const float a = live ? (k ? aw.y : aw.x) : 0.f;
        int4 o = make_int4(sg.off[0], sg.off[1], sg.off[2], sg.off[3]);
        if (lvl >= res_lvl) {
          // LDS byte offset of the pixel row, bit 6 = "halves swapped": the reader XORs its quad offset in
          const int t0 = (o.x >> 3) - resbias, t1 = (o.y >> 3) - resbias, t2 = (o.z >> 3) - resbias,
                    t3 = (o.w >> 3) - resbias;
          o = make_int4(t0 | ((t0 >> 2) & 64), t1 | ((t1 >> 2) & 64), t2 | ((t2 >> 2) & 64), t3 | ((t3 >> 2) & 64));
        }
        my_off[s] = o;
        my_w[s] = make_float4(sg.ok[0] ? sg.w[0] * a : 0.f, sg.ok[1] ? sg.w[1] * a : 0.f,
                              sg.ok[2] ? sg.w[2] * a : 0.f, sg.ok[3] ? sg.w[3] * a : 0.f);
      }
      // LDS ops of one wave execute in order; the fences only stop the compiler from reordering across lanes.
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

      // stage 2: lane (j, c) -> channels 4c .. 4c+3 of (query j, head).  The corners of up to GCH global-memory samples
      // (GCH x 4 x 16 B per lane) are requested first; the resident levels are gathered from LDS while they are in flight.
      const char* glb = vbase + c * 16;
      const int cl = c * 16;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#define EGTR_FMA4(W_, V0, V1, V2, V3)                                        \
      acc.x += W_.x * V0.x + W_.y * V1.x + W_.z * V2.x + W_.w * V3.x; \
      acc.y += W_.x * V0.y + W_.y * V1.y + W_.z * V2.y + W_.w * V3.y; \
      acc.z += W_.x * V0.z + W_.y * V1.z + W_.z * V2.z + W_.w * V3.z; \
      acc.w += W_.x * V0.w + W_.y * V1.w + W_.z * V2.w + W_.w * V3.w;
#define EGTR_LDS_SAMPLE(S_)                                                            \
      {                                                                              \
        const int4 o = my_off[S_];                                                   \
        const float4 w = my_w[S_];                                                   \
        const float4 v0 = *reinterpret_cast<const float4*>(lds_val + (o.x ^ cl));    \
        const float4 v1 = *reinterpret_cast<const float4*>(lds_val + (o.y ^ cl));    \
        const float4 v2 = *reinterpret_cast<const float4*>(lds_val + (o.z ^ cl));    \
        const float4 v3 = *reinterpret_cast<const float4*>(lds_val + (o.w ^ cl));    \
        EGTR_FMA4(w, v0, v1, v2, v3)                                                 \
      }
      // P is a multiple of 4 (L*P = 16, L <= 4), so the global samples come in whole batches of 4 (of GCH = 8 when their
      // number allows); each batch is requested (4 x GCH loads per lane), then an equal share of the resident samples
      // is gathered from LDS, then the batch is accumulated.  No branch separates a load from its use (hipcc waits
      // vmcnt(0) at such joins).
      int sr = nglob;  // next resident sample
#define EGTR_GLOBAL_BATCHES(NB)                                                         \
      {                                                                               \
        const int nbatch = nglob / NB;                                                \
        const int share = nbatch ? (16 - nglob) / nbatch : 0;                         \
        for (int s0 = 0; s0 < nglob; s0 += NB) {                   \
          float4 gv[NB][4];                                                           \
          _Pragma("unroll") for (int i = 0; i < NB; ++i) {                            \
            const int4 o = my_off[s0 + i];                                            \
            gv[i][0] = *reinterpret_cast<const float4*>(glb + (unsigned)o.x);         \
            gv[i][1] = *reinterpret_cast<const float4*>(glb + (unsigned)o.y);         \
            gv[i][2] = *reinterpret_cast<const float4*>(glb + (unsigned)o.z);         \
            gv[i][3] = *reinterpret_cast<const float4*>(glb + (unsigned)o.w);         \
          }                                                                           \
          {                                                                           \
            const int e = sr + share;                                                 \
            _Pragma("unroll 2") for (; sr < e; ++sr) EGTR_LDS_SAMPLE(sr)              \
          }                                                                           \
          _Pragma("unroll") for (int i = 0; i < NB; ++i) {                            \
            const float4 w = my_w[s0 + i];                                            \
            EGTR_FMA4(w, gv[i][0], gv[i][1], gv[i][2], gv[i][3])                      \
          }                                                                           \
        }                                                                             \
      }
      if (GCH == 8 && (nglob & 7) == 0) EGTR_GLOBAL_BATCHES(8) else EGTR_GLOBAL_BATCHES(4)
#undef EGTR_GLOBAL_BATCHES
#pragma unroll 4
      for (; sr < 16; ++sr) EGTR_LDS_SAMPLE(sr)
#undef EGTR_LDS_SAMPLE
#undef EGTR_FMA4
      if (live) reinterpret_cast<float4*>(out)[row * 8 + c] = acc;
      // the next pass overwrites this wave's records: same wave, LDS ops in order
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
     }
    }
  }
}

}  // namespace

// Launcher used by egtr_msda_forward_f32_variant (msda.hip) for M = 8, D = 32, L*P = 16.
int egtr_launch_msda_fwd_res_f32(hipStream_t st, const float* value, const int64_t* shapes, const int64_t* lsi,
                                 const float* loc, const float* attn, float* out, int B, int Lq, int S, int L, int P) {
  hipLaunchKernelGGL((msda_fwd_res_f32<12, 4>), dim3(256), dim3(12 * 64), 0, st, value, shapes, lsi, loc, attn, out, B,
                     Lq, S, L, P);
  return egtr_check_launch();
}
