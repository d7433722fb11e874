// MSDA forward, "region" kernel for gfx950 (variant 14): the adaptive kernel for encoder-shaped launches (queries = the
// pixels of the feature levels).
//
// Why: the wave-per-query kernel (msda.hip) is bound by the vector L1 / texture addresser -- every 16-byte-per-lane
// gather instruction costs ~17-19 CU cycles whatever it fetches (tools/l1_mask.hip: masking lanes or repeating a line
// inside an instruction saves nothing), and it needs 64 of them per (query, head): 822 MB through a 64 B/clk/CU pipe
// = 27 us per encoder launch at best.  The LDS serves the same 16 B per lane in 4 cycles, but the earlier LDS designs
// (8x8 query tiles with bounding-box windows; lane = (query, channel quad) with per-sample records in LDS) turned out
// to be bound by INSTRUCTION ISSUE (~5 cycles per instruction per SIMD at 2-4 waves per SIMD): records, cross-lane
// traffic, per-tile bookkeeping and re-staged halos cost more instructions than the gather itself.
//
// This kernel minimises instructions per (query, head) unit:
//   * item = (image, region, head).  The image plane is cut into RY x RX regions in normalised coordinates; a region
//     owns the queries of ALL four levels whose pixel lies in it (~196 queries at 600x1000 with 8x8 regions).
//   * per source level the item needs the pixels under its region plus a fixed halo (kHalo px of THAT level: sampling
//     offsets are expressed in pixels of the sampled level, deformable_detr.py:1067-1073).  The window depends on the
//     region only, so its copy starts at once.  Out-of-level window pixels and padded tokens are stored as zeros
//     (== cuh:55-78 per-corner range checks, dd:1052 masked_fill), so staged samples need no per-corner masks.
//   * the four source levels pass one after the other through ONE window buffer (68 KB: two workgroups per CU);
//   * LANE = one (query, head) unit with all 32 channels: the lane computes the geometry of its own samples (no
//     redundancy, no records, no cross-lane traffic), and gathers each bilinear corner with 8 ds_read_b128 whose
//     channel-quad plane is an instruction-offset immediate -- the window is stored as 8 planes [quad][pixel] of 16-byte
//     entries, so the 16 lanes of a hardware ds_read_b128 group (x-adjacent queries -> consecutive pixels) hit 16
//     different 4-bank groups; accumulation in 32 registers with packed FMAs, sample order 0..15 as in the reference.
//     ~42 instructions per unit against ~160 for the record-based form.
//   * samples that fall outside their window are gathered from global memory by the lanes concerned, so results never
//     depend on the halo; and a region whose sampled outlier rate is high (irregular offsets) switches, as a whole and
//     consistently across its 8 heads' workgroups, to the wave-per-query scheme (8 heads per wave, L1 gathers).
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

#include "common.h"
#include "msda_common.h"

namespace {

using namespace egtr_msda;

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) f32x4* lds_f4p;   // explicit address spaces: LDS and global reads of one
typedef const __attribute__((address_space(1))) f32x4* glb_f4p;   // value must never be merged into flat loads

constexpr int kRW = 4;                 // waves per workgroup: 256 lanes = up to 256 (query, head) units
constexpr int kRT = kRW * 64;          // threads
constexpr int kWinPx = 576;            // window pixels per level: (10 + 11) x (16 + 11) = 567 at 600x1000 with 8 x 8 regions
constexpr int kZero = 2;               // all-zero pixels in front of the window (target of invalid samples)
constexpr int kNP = kZero + kWinPx + 1;   // pixels per channel-quad plane; ODD: conflict-free plane-scattered ds_write_b128
constexpr int kPlane = kNP * 16;       // bytes per plane
constexpr int kHalo = 5;               // pixels of the sampled level around the region (levels 1..3)
constexpr int kHalo0 = 4;              // level 0 (its window is the LDS-capacity limit: 546 px at 600x1000, 575 at 800x1333)
constexpr int kProbeQ = 16;            // queries sampled (x 8 heads x 16 samples) for the per-region mode decision
static_assert(kNP % 2 == 1 && 7 * kPlane + 16 < 65536, "plane offsets are ds_read immediates");
// LDS carve-up in float4 units
constexpr int kOffWin = 0;                                   // 8 planes x kNP entries
constexpr int kOffMisc = 8 * kNP;                            // 16: counters + the region's rectangles / windows
constexpr int kLdsF4 = kOffMisc + 16;
// s_misc (ints): [0] outliers, [1] valid samples of the probe, [4+l] queries of level l, [8+4l ..] query rectangle of
// level l {y0, x0, h, w}, [24+5s ..] window of source level s {y0, x0, h, w, staged}
constexpr int kMiscN = 4, kMiscRect = 8, kMiscWin = 24;
static_assert(kLdsF4 * 16 <= 81920, "two workgroups per CU");
// wave-per-query fallback: per wave 2 x 136 records of 16 B, aliased onto the window
constexpr int kHeadStride = 17, kWaveEntries = 8 * kHeadStride;
static_assert(2 * kRW * kWaveEntries <= 8 * kNP, "fallback records fit in the window buffer");

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

#ifdef EGTR_REGION_PROF
__device__ unsigned long long g_prof[32];
#define PROF_T(slot)                                                          \
  if (tid == 0) {                                                             \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();             \
    atomicAdd(&g_prof[slot], now_ - tprev_);                                  \
    tprev_ = now_;                                                            \
  }
#else
#define PROF_T(slot)
#endif

struct Rect { int y0, x0, h, w; };     // query rectangle of one level inside the region
struct Win { int y0, x0, h, w; bool staged; };   // window of one source level (level pixel coordinates)

// The region's tables live in LDS (wave-uniform values that would otherwise occupy ~60 scalar registers for the whole
// kernel); readers get them with broadcast ds_reads where they need them.
__device__ __forceinline__ Rect lds_rect(const int* misc, int l) {
  const int4 v = *reinterpret_cast<const int4*>(misc + kMiscRect + 4 * l);
  return Rect{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ Win lds_win(const int* misc, int s) {
  const int* p = misc + kMiscWin + 5 * s;
  return Win{p[0], p[1], p[2], p[3], p[4] != 0};
}

// idx (0 .. nq-1) -> query index within the image
__device__ __forceinline__ int region_query(const int* misc, const LevelGeom& G, bool grid, int lin0, int idx) {
  if (!grid) return lin0 + idx;
  const int4 n = *reinterpret_cast<const int4*>(misc + kMiscN);
  int l = 0, t = idx;
  if (t >= n.x) { t -= n.x; l = 1;
    if (t >= n.y) { t -= n.y; l = 2;
      if (t >= n.z) { t -= n.z; l = 3; } } }
  const Rect r = lds_rect(misc, l);
  const int ry = (int)(((float)t + 0.5f) * __frcp_rn((float)r.w));   // exact for t, w < 4096
  const int rx = t - ry * r.w;
  return SEL_S(G, l) + (r.y0 + ry) * SEL_W(G, l) + r.x0 + rx;
}

// Window of source level s for a region: union over the query levels of [first centre - halo, last centre + halo + 1],
// clipped to [-1, size] (valid samples never touch anything else).
__device__ __forceinline__ Win make_window(const int* misc, const LevelGeom& G, bool grid, int s) {
  const float Ws = (float)SEL_W(G, s), Hs = (float)SEL_H(G, s);
  float xlo = 1e9f, xhi = -1e9f, ylo = 1e9f, yhi = -1e9f;
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    const Rect r = lds_rect(misc, l);
    if (r.h > 0 && r.w > 0) {
      const float Wl = (float)SEL_W(G, l), Hl = (float)SEL_H(G, l);
      xlo = fminf(xlo, ((float)r.x0 + 0.5f) * Ws / Wl - 0.5f);
      xhi = fmaxf(xhi, ((float)(r.x0 + r.w) - 0.5f) * Ws / Wl - 0.5f);
      ylo = fminf(ylo, ((float)r.y0 + 0.5f) * Hs / Hl - 0.5f);
      yhi = fmaxf(yhi, ((float)(r.y0 + r.h) - 0.5f) * Hs / Hl - 0.5f);
    }
  }
  Win w;
  const int halo = s == 0 ? kHalo0 : kHalo;
  w.x0 = max((int)floorf(xlo) - halo, -1);
  w.y0 = max((int)floorf(ylo) - halo, -1);
  const int x1 = min((int)floorf(xhi) + halo + 1, SEL_W(G, s));
  const int y1 = min((int)floorf(yhi) + halo + 1, SEL_H(G, s));
  w.w = x1 - w.x0 + 1;
  w.h = y1 - w.y0 + 1;
  w.staged = grid && w.w > 0 && w.h > 0 && w.w * w.h <= kWinPx;
  return w;
}

template <bool FUSED>
__device__ __forceinline__ void softmax16(float2& aw) {
  float m = fmaxf(aw.x, aw.y);
  m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0xB1, 0xf, 0xf, false)));
  m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x4E, 0xf, 0xf, false)));
  m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x141, 0xf, 0xf, false)));
  const float e0 = expf(aw.x - m), e1 = expf(aw.y - m);
  float sum = e0 + e1;
  sum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum), 0xB1, 0xf, 0xf, false));
  sum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum), 0x4E, 0xf, 0xf, false));
  sum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum), 0x141, 0xf, 0xf, false));
  aw = make_float2(e0 / sum, e1 / sum);
}

// Wave-per-query scheme (the msda.hip kernel's body): lane = (head, channel quad), records of the query's 8 x 16
// samples in this wave's LDS slice, 64 gathers of the 8 heads' 128-B lines.
template <bool FUSED>
__device__ __forceinline__ void query_all_heads(const float* __restrict__ value, const float* __restrict__ loc,
                                                const float* __restrict__ attn, float* __restrict__ out,
                                                const float* __restrict__ ref, float* __restrict__ attn_out, int ld_off,
                                                int ld_logit, const unsigned* __restrict__ kb, const LevelGeom& G, int b,
                                                int q /* within the image */, int Lq, int S, int lane, int4* my_off,
                                                float4* my_w) {
  const size_t gq = (size_t)b * Lq + q;
  float4 lc = reinterpret_cast<const float4*>(loc + gq * (FUSED ? ld_off : 256))[lane];
  float2 aw = reinterpret_cast<const float2*>(attn + gq * (FUSED ? ld_logit : 128))[lane];
  const int head_s = lane >> 3, s0 = (lane & 7) * 2;
  if (FUSED) {
    const int lvl = s0 >> 2;
    const float2 r = *reinterpret_cast<const float2*>(ref + (gq * 4 + lvl) * 2);
    const float fw = (float)SEL_W(G, lvl), fh = (float)SEL_H(G, lvl);
    lc = make_float4(r.x + lc.x / fw, r.y + lc.y / fh, r.x + lc.z / fw, r.y + lc.w / fh);
    softmax16<FUSED>(aw);
    if (attn_out != nullptr) reinterpret_cast<float2*>(attn_out + gq * 128)[lane] = aw;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int s = s0 + j;
    const int lvl = s >> 2;
    const SampleGeom g = sample_geom<1024, 128>(j ? lc.z : lc.x, j ? lc.w : lc.y, SEL_H(G, lvl), SEL_W(G, lvl),
                                                SEL_S(G, lvl), head_s);
    const float a = j ? aw.y : aw.x;
    bool k0 = g.ok[0], k1 = g.ok[1], k2 = g.ok[2], k3 = g.ok[3];
    if (kb != nullptr) {
      const int p0 = g.off[0] >> 10, p1 = g.off[1] >> 10, p2 = g.off[2] >> 10, p3 = g.off[3] >> 10;
      k0 = k0 && ((kb[p0 >> 5] >> (p0 & 31)) & 1u);
      k1 = k1 && ((kb[p1 >> 5] >> (p1 & 31)) & 1u);
      k2 = k2 && ((kb[p2 >> 5] >> (p2 & 31)) & 1u);
      k3 = k3 && ((kb[p3 >> 5] >> (p3 & 31)) & 1u);
    }
    my_off[head_s * kHeadStride + s] = make_int4(g.off[0], g.off[1], g.off[2], g.off[3]);
    my_w[head_s * kHeadStride + s] = make_float4(k0 ? g.w[0] * a : 0.f, k1 ? g.w[1] * a : 0.f,
                                                 k2 ? g.w[2] * a : 0.f, k3 ? g.w[3] * a : 0.f);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const char* vlane = reinterpret_cast<const char*>(value) + (size_t)b * S * 1024 + (lane & 7) * 16;
  const int4* ro = my_off + (lane >> 3) * kHeadStride;
  const float4* rw = my_w + (lane >> 3) * kHeadStride;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
  for (int s = 0; s < 16; ++s) {
    const int4 o = ro[s];
    const float4 w = rw[s];
    const float4 v0 = *reinterpret_cast<const float4*>(vlane + (unsigned)o.x);
    const float4 v1 = *reinterpret_cast<const float4*>(vlane + (unsigned)o.y);
    const float4 v2 = *reinterpret_cast<const float4*>(vlane + (unsigned)o.z);
    const float4 v3 = *reinterpret_cast<const float4*>(vlane + (unsigned)o.w);
    acc.x += w.x * v0.x + w.y * v1.x + w.z * v2.x + w.w * v3.x;
    acc.y += w.x * v0.y + w.y * v1.y + w.z * v2.y + w.w * v3.y;
    acc.z += w.x * v0.z + w.y * v1.z + w.z * v2.z + w.w * v3.z;
    acc.w += w.x * v0.w + w.y * v1.w + w.z * v2.w + w.w * v3.w;
  }
  reinterpret_cast<float4*>(out + gq * 256)[lane] = acc;
  // the records of this wave are rewritten by its next query: keep the reads above ahead of those writes
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// mode_force: 0 = adaptive (probe), 1 = always the window scheme (outliers per sample from global), 2 = always the
// wave-per-query scheme (tests / A-B timing).
template <bool FUSED>
__global__ __launch_bounds__(kRT, 2) void msda_fwd_region_f32(
    const float* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
    const float* __restrict__ loc, const float* __restrict__ attn, float* __restrict__ out, int B, int Lq, int S,
    const float* __restrict__ ref, float* __restrict__ attn_out, int ld_off, int ld_logit,
    const unsigned* __restrict__ keep_bits, int RY, int RX, int nblk, int mode_force) {
  __shared__ __attribute__((aligned(16))) float4 s_mem[kLdsF4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = rfl(tid >> 6);
  const int item = rfl(xcd_remap(blockIdx.x, nblk));
#ifdef EGTR_REGION_PROF
  unsigned long long tprev_ = __builtin_amdgcn_s_memtime();
#endif
  const int R_ = RY * RX;
  const int b = rfl(item / (R_ * 8));
  const int rem = item - b * R_ * 8;
  const int reg = rem >> 3, head = rem & 7;
  const int ry = rfl(reg / RX), rx = reg - ry * RX;
  LevelGeom G;
  load_geom(shapes, lsi, 4, G);

  // ---- the region's queries and windows (tables in LDS) ----------------------------------------------------------------
  int* s_misc = reinterpret_cast<int*>(s_mem + kOffMisc);
  bool grid;
  int lin0 = 0, nq;
  {
    const int s0 = G.H0 * G.W0, s1 = G.H1 * G.W1, s2 = G.H2 * G.W2, s3 = G.H3 * G.W3;
    grid = (s0 + s1 + s2 + s3 == Lq) && (G.s0 == 0) && (G.s1 == s0) && (G.s2 == s0 + s1) && (G.s3 == s0 + s1 + s2);
    if (tid < 4) {
      const int H = SEL_H(G, tid), W = SEL_W(G, tid);
      const int y0 = (ry * H) / RY, x0 = (rx * W) / RX;
      const int h = ((ry + 1) * H) / RY - y0, w = ((rx + 1) * W) / RX - x0;
      *reinterpret_cast<int4*>(s_misc + kMiscRect + 4 * tid) = make_int4(y0, x0, h, w);
      s_misc[kMiscN + tid] = h * w;
    }
    if (tid == 0) { s_misc[0] = 0; s_misc[1] = 0; }
  }
  const int nwords = (S + 31) >> 5;
  const bool masked = FUSED && keep_bits != nullptr;
  const unsigned* kb_g = masked ? keep_bits + (size_t)b * nwords : nullptr;
  // the zero pixels (entries 0, 1 of every plane)
  if (tid < 16) s_mem[(tid >> 1) * kNP + (tid & 1)] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  if (tid < 4) {
    const Win w = make_window(s_misc, G, grid, tid);
    int* p = s_misc + kMiscWin + 5 * tid;
    p[0] = w.y0; p[1] = w.x0; p[2] = w.h; p[3] = w.w; p[4] = w.staged ? 1 : 0;
  }
  if (grid) {
    const int4 n = *reinterpret_cast<const int4*>(s_misc + kMiscN);
    nq = rfl(n.x + n.y + n.z + n.w);
  } else {  // arbitrary query list: consecutive chunks, wave-per-query scheme only
    lin0 = (int)(((long long)reg * Lq) / R_);
    nq = (int)(((long long)(reg + 1) * Lq) / R_) - lin0;
  }
  __syncthreads();

  const char* vimg = reinterpret_cast<const char*>(value) + (size_t)b * S * 1024 + head * 128;
  typedef const __attribute__((address_space(3))) char* lds_cp;
  typedef const __attribute__((address_space(1))) char* glb_cp;

  // Window copy of source level s into the 8 channel-quad planes, in two halves so that the global loads of level s + 1
  // are in flight while level s is gathered: fill_load issues up to NPC loads per lane (flat window pixel index
  // pi = it * 32 + tid / 8, lane = (pixel, channel quad)), fill_store writes them to LDS after the barrier.  Pixels
  // outside the level and padded tokens become zeros (cuh:55-78, dd:1052).
  auto fill_load = [&](int s, float4* v, auto npc_tag) {
    constexpr int NPC = decltype(npc_tag)::value;
    const Win w = lds_win(s_misc, s);
    const int Hs = SEL_H(G, s), Ws = SEL_W(G, s), ss = SEL_S(G, s);
    const int wy0 = rfl(w.y0), wx0 = rfl(w.x0), ww = rfl(w.w), npx = w.staged ? rfl(w.w * w.h) : 0;
    const char* srcq = vimg + (tid & 7) * 16;
    // window coordinates of this lane's first pixel, then +32 pixels per load with carries (no division per load)
    int pi = tid >> 3;
    int wy = (int)(((float)pi + 0.5f) * __frcp_rn((float)ww));   // exact for pi, ww < 4096
    int wx = pi - wy * ww;
    const int dy = (kRT / 8) / ww, dxr = (kRT / 8) - dy * ww;     // wave-uniform
#pragma unroll
    for (int it = 0; it < NPC; ++it) {
      v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (pi < npx) {
        const int y = wy0 + wy, x = wx0 + wx;
        bool live = (unsigned)y < (unsigned)Hs && (unsigned)x < (unsigned)Ws;
        const int p = ss + y * Ws + x;
        if (live && masked) live = (kb_g[p >> 5] >> (p & 31)) & 1u;
        if (live) v[it] = *reinterpret_cast<const float4*>(srcq + (size_t)p * 1024);
      }
      pi += kRT / 8;
      wx += dxr;
      wy += dy;
      if (wx >= ww) { wx -= ww; wy += 1; }
    }
  };
  auto fill_store = [&](int s, const float4* v, auto npc_tag) {
    constexpr int NPC = decltype(npc_tag)::value;
    const Win w = lds_win(s_misc, s);
    const int npx = w.staged ? rfl(w.w * w.h) : 0;
    float4* dstq = s_mem + (tid & 7) * kNP + kZero + (tid >> 3);
#pragma unroll
    for (int it = 0; it < NPC; ++it)
      if (it * (kRT / 8) + (tid >> 3) < npx) dstq[it * (kRT / 8)] = v[it];
  };
  constexpr int kNpc0 = (kWinPx + kRT / 8 - 1) / (kRT / 8);   // level-0 window: up to kWinPx pixels (17 loads per lane)
  constexpr int kNpcN = 12;                                     // prefetched windows of the coarser levels (<= 384 pixels)
  typedef std::integral_constant<int, kNpc0> Npc0;
  typedef std::integral_constant<int, kNpcN> NpcN;

  // ---- mode decision: the same sample of the region's queries (all 8 heads) in each of its 8 workgroups ----------------
  bool window_mode = grid && nq <= kRT && nq > 0;
  if (mode_force == 2) window_mode = false;
  if (window_mode && mode_force == 0) {
    const int k = tid >> 4, sub = tid & 15, ph = sub >> 1, half = sub & 1;
    const int pq = region_query(s_misc, G, grid, lin0, (k * nq) / kProbeQ);
    const size_t gq = (size_t)b * Lq + pq;
    int nout = 0, nval = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {        // samples 8*half + 2i, +1  (level 2*half + (i >> 1))
      const int lvl = 2 * half + (i >> 1);
      float4 lc = *reinterpret_cast<const float4*>(loc + gq * (FUSED ? ld_off : 256) + ph * 32 + half * 16 + i * 4);
      const float fw = (float)SEL_W(G, lvl), fh = (float)SEL_H(G, lvl);
      if (FUSED) {
        const float2 r = *reinterpret_cast<const float2*>(ref + (gq * 4 + lvl) * 2);
        lc = make_float4(r.x + lc.x / fw, r.y + lc.y / fh, r.x + lc.z / fw, r.y + lc.w / fh);
      }
      const Win w = lds_win(s_misc, lvl);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float x = (j ? lc.z : lc.x) * fw - 0.5f, y = (j ? lc.w : lc.y) * fh - 0.5f;
        const bool val = (y > -1.f) && (x > -1.f) && (y < fh) && (x < fw);
        if (val) {
          const int y0 = (int)floorf(y), x0 = (int)floorf(x);
          const bool in = w.staged && y0 >= w.y0 && x0 >= w.x0 && y0 + 1 < w.y0 + w.h && x0 + 1 < w.x0 + w.w;
          nval += 1;
          nout += in ? 0 : 1;
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      nout += __shfl_xor(nout, o);
      nval += __shfl_xor(nval, o);
    }
    if (lane == 0) {
      atomicAdd(&s_misc[0], nout);
      atomicAdd(&s_misc[1], nval);
    }
    __syncthreads();
    window_mode = s_misc[0] * 16 <= s_misc[1];   // <= 1/16 of the sampled (valid) samples outside their window
  }

  if (!window_mode) {
    // ---- wave-per-query scheme: this workgroup takes every 8th query of the region, all heads -------------------------
    int4* my_off = reinterpret_cast<int4*>(s_mem + kOffWin) + wave * kWaveEntries;
    float4* my_w = s_mem + kOffWin + kRW * kWaveEntries + wave * kWaveEntries;
    for (int idx = head + 8 * wave; idx < nq; idx += 8 * kRW) {
      const int q = region_query(s_misc, G, grid, lin0, idx);
      query_all_heads<FUSED>(value, loc, attn, out, ref, attn_out, ld_off, ld_logit, kb_g, G, b, q, Lq, S, lane, my_off,
                             my_w);
    }
    return;
  }

  PROF_T(0)   // prologue: geometry, tables, probe
  // ---- window scheme: lane = (query, head) unit ----------------------------------------------------------------------
  float4 w0v[kNpc0];
  fill_load(0, w0v, Npc0());   // level 0's window loads are in flight with this unit's operand loads
  const int idx = tid;
  const bool live = idx < nq;
  int q = 0;
  float4 lcs[8];        // sampling locations / offsets of the 16 samples, (x, y) pairs in sample order
  float aws[16];        // attention weights (softmaxed)
  float2 rps[4];        // reference points per level (FUSED)
  if (live) {
    q = region_query(s_misc, G, grid, lin0, idx);
    const size_t gq = (size_t)b * Lq + q;
    const float4* pl = reinterpret_cast<const float4*>(loc + gq * (FUSED ? ld_off : 256) + head * 32);
    const float4* pa = reinterpret_cast<const float4*>(attn + gq * (FUSED ? ld_logit : 128) + head * 16);
#pragma unroll
    for (int i = 0; i < 8; ++i) lcs[i] = pl[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 a = pa[i];
      aws[4 * i + 0] = a.x; aws[4 * i + 1] = a.y; aws[4 * i + 2] = a.z; aws[4 * i + 3] = a.w;
    }
    if (FUSED) {
      const float4* pr = reinterpret_cast<const float4*>(ref + gq * 8);
      const float4 r0 = pr[0], r1 = pr[1];
      rps[0] = make_float2(r0.x, r0.y); rps[1] = make_float2(r0.z, r0.w);
      rps[2] = make_float2(r1.x, r1.y); rps[3] = make_float2(r1.z, r1.w);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) lcs[i] = make_float4(9.f, 9.f, 9.f, 9.f);   // far outside: invalid
#pragma unroll
    for (int i = 0; i < 16; ++i) aws[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) rps[i] = make_float2(0.f, 0.f);
  }
  if (FUSED) {
    // softmax over the unit's 16 logits, lane-private (same operations as the wave-per-query kernel's DPP form)
    float m = aws[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) m = fmaxf(m, aws[i]);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { aws[i] = __expf(aws[i] - m); }   // (2 ulp; the wave-per-query kernel uses expf)
    // the DPP butterfly adds (e0 + e1) per lane first, then lanes ^1, ^2, mirror: same association here
    {
      float p2[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) p2[i] = aws[2 * i] + aws[2 * i + 1];
      float p4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) p4[i] = p2[2 * i] + p2[2 * i + 1];
      const float lo = p4[0] + p4[1], hi = p4[2] + p4[3];
      sum = lo + hi;
    }
    const float rsum = 1.0f / sum;
#pragma unroll
    for (int i = 0; i < 16; ++i) aws[i] = aws[i] * rsum;
    if (attn_out != nullptr && live) {
      float4* po = reinterpret_cast<float4*>(attn_out + ((size_t)b * Lq + q) * 128 + head * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) po[i] = make_float4(aws[4 * i], aws[4 * i + 1], aws[4 * i + 2], aws[4 * i + 3]);
    }
  }

  PROF_T(1)   // operand loads + softmax
  f32x2 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = (f32x2){0.f, 0.f};
  const lds_cp lwin = (lds_cp)(s_mem);
  const glb_cp gimg = (glb_cp)(vimg);

  // The level loop is NOT unrolled: the whole kernel has to stay well inside the 64 KB instruction cache (a fully
  // unrolled build was ~100 KB of straight-line code executed once per workgroup: instruction fetch, not LDS or the
  // ALUs, set its time -- 60 us).  The level's operands are selected into level-local registers at the top.
  fill_store(0, w0v, Npc0());
#pragma unroll 1
  for (int s = 0; s < 4; ++s) {
    const Win w = lds_win(s_misc, s);
    const int Hs = SEL_H(G, s), Ws = SEL_W(G, s), ss = SEL_S(G, s);
    const float fw = (float)Ws, fh = (float)Hs;
    const int wy0 = rfl(w.y0), wx0 = rfl(w.x0), wh = rfl(w.h), ww = rfl(w.w);
    const bool staged = w.staged;
    // this level's operands are ALWAYS in lcs[0..1] / aws[0..3] / rps[0]: the arrays are rotated at the end of the pass
    // (a selection by the loop counter would be turned into a scratch-memory table by the compiler)
    const float4 la = lcs[0], lb = lcs[1];
    const float a4[4] = {aws[0], aws[1], aws[2], aws[3]};
    const float2 rp = rps[0];
    __syncthreads();    // window s is in LDS
    PROF_T(2)
    // the loads of the NEXT level's window are in flight during this level's gather (when they fit the registers)
    float4 nxt[kNpcN];
    bool nxt_ok = false;
    if (s + 1 < 4) {
      const Win wn = lds_win(s_misc, s + 1);
      nxt_ok = rfl(wn.w * wn.h) <= kNpcN * (kRT / 8);   // wave-uniform
      if (nxt_ok) fill_load(s + 1, nxt, NpcN());
    }

    // geometry of the unit's 4 samples of this level
    int p00[4], prow[4];
    float wc[4][4];
    bool outl[4];
    int sy0[4], sx0[4];
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
      const float4 l4 = pt < 2 ? la : lb;
      float lx = (pt & 1) ? l4.z : l4.x, ly = (pt & 1) ? l4.w : l4.y;
      if (FUSED) {
        lx = rp.x + lx / fw;
        ly = rp.y + ly / fh;
      }
      float x = lx * fw - 0.5f, y = ly * fh - 0.5f;
      const bool val = (y > -1.f) && (x > -1.f) && (y < fh) && (x < fw);
      x = val ? x : 0.f;
      y = val ? y : 0.f;
      const float a = val ? a4[pt] : 0.f;
      const float yf = floorf(y), xf = floorf(x);
      const float lh = y - yf, lw = x - xf, hh = 1.f - lh, hw = 1.f - lw;
      const int y0 = (int)yf, x0 = (int)xf;
      wc[pt][0] = hh * hw * a;
      wc[pt][1] = hh * lw * a;
      wc[pt][2] = lh * hw * a;
      wc[pt][3] = lh * lw * a;
      const bool in = staged && y0 >= wy0 && x0 >= wx0 && y0 + 1 < wy0 + wh && x0 + 1 < wx0 + ww;
      outl[pt] = val && !in;
      // window pixel of (y0, x0); invalid samples read the zero pixels with zero weights
      p00[pt] = (val && in) ? kZero + (y0 - wy0) * ww + (x0 - wx0) : 0;
      prow[pt] = (val && in) ? ww : 0;
      sy0[pt] = y0;
      sx0[pt] = x0;
    }
    PROF_T(3)   // next window's loads issued + geometry
    if (!__any((int)(outl[0] || outl[1] || outl[2] || outl[3]))) {
      // every sample of the wave is in the window: 16 corners, each 8 ds_read_b128 (plane = offset immediate) + 16
      // packed FMAs, the reads of corner c + 1 in flight while corner c is accumulated
      auto corner_ptr = [&](int c) -> lds_cp {
        const int pt = c >> 2, cn = c & 3;
        return lwin + (p00[pt] + ((cn & 2) ? prow[pt] : 0) + (cn & 1)) * 16;
      };
      f32x4 vb[2][8];
      {
        const lds_cp pc_ = corner_ptr(0);
#pragma unroll
        for (int qd = 0; qd < 8; ++qd) vb[0][qd] = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(pc_ + qd * kPlane);
      }
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        if (c + 1 < 16) {
          const lds_cp pc_ = corner_ptr(c + 1);
#pragma unroll
          for (int qd = 0; qd < 8; ++qd)
            vb[(c + 1) & 1][qd] = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(pc_ + qd * kPlane);
        }
        const float wgt = wc[c >> 2][c & 3];
#pragma unroll
        for (int qd = 0; qd < 8; ++qd) {
          acc[2 * qd] += (f32x2){wgt, wgt} * (f32x2){vb[c & 1][qd].x, vb[c & 1][qd].y};
          acc[2 * qd + 1] += (f32x2){wgt, wgt} * (f32x2){vb[c & 1][qd].z, vb[c & 1][qd].w};
        }
        __builtin_amdgcn_sched_barrier(0);   // two corners (16 reads) in flight at most
      }
    } else {
#ifdef EGTR_REGION_PROF
      if (lane == 0) atomicAdd(&g_prof[16 + s], 1ull);
      {
        const int no = (int)outl[0] + (int)outl[1] + (int)outl[2] + (int)outl[3];
        atomicAdd(&g_prof[20 + s], (unsigned long long)no);
        if (!staged && lane == 0) atomicAdd(&g_prof[24 + s], 1ull);
      }
#endif
      // some lane has a sample outside its window: those lanes read that sample's corners from global memory (out-of-range
      // corners and padded tokens folded into the weights, cuh:55-78 / dd:1052), the others from the window
#pragma unroll 1
      for (int pt = 0; pt < 4; ++pt) {
        const bool ol = pt == 0 ? outl[0] : pt == 1 ? outl[1] : pt == 2 ? outl[2] : outl[3];
        const int y0 = pt == 0 ? sy0[0] : pt == 1 ? sy0[1] : pt == 2 ? sy0[2] : sy0[3];
        const int x0 = pt == 0 ? sx0[0] : pt == 1 ? sx0[1] : pt == 2 ? sx0[2] : sx0[3];
        const int pp0 = pt == 0 ? p00[0] : pt == 1 ? p00[1] : pt == 2 ? p00[2] : p00[3];
        const int pr0 = pt == 0 ? prow[0] : pt == 1 ? prow[1] : pt == 2 ? prow[2] : prow[3];
        float w4[4];
#pragma unroll
        for (int cn = 0; cn < 4; ++cn) w4[cn] = pt == 0 ? wc[0][cn] : pt == 1 ? wc[1][cn] : pt == 2 ? wc[2][cn] : wc[3][cn];
        unsigned gpix = 0;
        int dx = 0, dy = 0;
        if (ol) {
          const int ya = max(y0, 0), yb = min(y0 + 1, Hs - 1), xa = max(x0, 0), xb = min(x0 + 1, Ws - 1);
          const int g00 = ss + ya * Ws + xa;
          dx = xb - xa;
          dy = yb - ya;
          bool k0 = y0 >= 0 && x0 >= 0, k1 = y0 >= 0 && x0 + 1 <= Ws - 1, k2 = y0 + 1 <= Hs - 1 && x0 >= 0,
               k3 = y0 + 1 <= Hs - 1 && x0 + 1 <= Ws - 1;
          if (masked) {
            const int g01 = g00 + dx, g10 = g00 + dy * Ws, g11 = g10 + dx;
            k0 = k0 && ((kb_g[g00 >> 5] >> (g00 & 31)) & 1u);
            k1 = k1 && ((kb_g[g01 >> 5] >> (g01 & 31)) & 1u);
            k2 = k2 && ((kb_g[g10 >> 5] >> (g10 & 31)) & 1u);
            k3 = k3 && ((kb_g[g11 >> 5] >> (g11 & 31)) & 1u);
          }
          w4[0] = k0 ? w4[0] : 0.f;
          w4[1] = k1 ? w4[1] : 0.f;
          w4[2] = k2 ? w4[2] : 0.f;
          w4[3] = k3 ? w4[3] : 0.f;
          gpix = (unsigned)g00;
        }
#pragma unroll 1
        for (int cn = 0; cn < 4; ++cn) {
          f32x4 v[8];
          if (ol) {
            const glb_cp gp = gimg + ((size_t)gpix + ((cn & 1) ? dx : 0) + ((cn & 2) ? dy * Ws : 0)) * 1024;
#pragma unroll
            for (int qd = 0; qd < 8; ++qd) v[qd] = *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(gp + qd * 16);
          } else {
            const lds_cp pc_ = lwin + (pp0 + ((cn & 2) ? pr0 : 0) + (cn & 1)) * 16;
#pragma unroll
            for (int qd = 0; qd < 8; ++qd) v[qd] = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(pc_ + qd * kPlane);
          }
          const float wgt = cn == 0 ? w4[0] : cn == 1 ? w4[1] : cn == 2 ? w4[2] : w4[3];
#pragma unroll
          for (int qd = 0; qd < 8; ++qd) {
            acc[2 * qd] += (f32x2){wgt, wgt} * (f32x2){v[qd].x, v[qd].y};
            acc[2 * qd + 1] += (f32x2){wgt, wgt} * (f32x2){v[qd].z, v[qd].w};
          }
        }
      }
    }
    PROF_T(4)   // gather
    if (s + 1 < 4) {
      __syncthreads();  // this level's gathers are done with the window buffer
      if (nxt_ok) {
        fill_store(s + 1, nxt, NpcN());
      } else {
        float4 tmp[kNpc0];
        fill_load(s + 1, tmp, Npc0());
        fill_store(s + 1, tmp, Npc0());
      }
    }
    PROF_T(5)   // next window's stores
#pragma unroll
    for (int i = 0; i < 6; ++i) lcs[i] = lcs[i + 2];
#pragma unroll
    for (int i = 0; i < 12; ++i) aws[i] = aws[i + 4];
#pragma unroll
    for (int i = 0; i < 3; ++i) rps[i] = rps[i + 1];
  }

  if (live) {
    float4* po = reinterpret_cast<float4*>(out + ((size_t)b * Lq + q) * 256 + head * 32);
#pragma unroll
    for (int qd = 0; qd < 8; ++qd) po[qd] = make_float4(acc[2 * qd].x, acc[2 * qd].y, acc[2 * qd + 1].x, acc[2 * qd + 1].y);
  }
  PROF_T(14)  // output stores issued
#ifdef EGTR_REGION_PROF
  if (tid == 0) atomicAdd(&g_prof[15], 1ull);
#endif
}

}  // namespace

// Launcher (declared in msda.hip).  fused: loc / attn are raw offsets / logits with row strides ld_off / ld_logit and
// ref the reference points [B, Lq, 4, 2]; otherwise finished sampling locations / attention weights.
// mode: 0 adaptive, 1 window scheme forced, 2 wave-per-query scheme forced.
int egtr_launch_msda_fwd_region_f32(hipStream_t st, const float* value, const int64_t* shapes, const int64_t* lsi,
                                    const float* loc, const float* attn, float* out, int B, int Lq, int S,
                                    const float* ref, float* attn_out, int ld_off, int ld_logit,
                                    const unsigned* keep_bits, int mode) {
  // regions: the smallest power of two that leaves <= ~200 queries per region (<= 256 is what a workgroup holds)
  int R = 4;
  while ((long long)R * 200 < Lq && R < 4096) R <<= 1;
  int k = 0;
  while ((1 << k) < R) ++k;
  const int RY = 1 << (k / 2), RX = R / RY;
  const long long nblk = (long long)B * R * 8;
  if (nblk >= (1ll << 31)) return EGTR_E_UNSUPPORTED;
  if (ref != nullptr)
    hipLaunchKernelGGL(msda_fwd_region_f32<true>, dim3((unsigned)nblk), dim3(kRT), 0, st, value, shapes, lsi, loc, attn,
                       out, B, Lq, S, ref, attn_out, ld_off, ld_logit, keep_bits, RY, RX, (int)nblk, mode);
  else
    hipLaunchKernelGGL(msda_fwd_region_f32<false>, dim3((unsigned)nblk), dim3(kRT), 0, st, value, shapes, lsi, loc, attn,
                       out, B, Lq, S, nullptr, nullptr, 256, 128, nullptr, RY, RX, (int)nblk, mode);
  return egtr_check_launch();
}

#ifdef EGTR_REGION_PROF
extern "C" int egtr_debug_region_prof(unsigned long long* out32) {
  unsigned long long z[32] = {0};
  if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_prof), sizeof(z)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)) != hipSuccess) return -1;
  return 0;
}
#endif
