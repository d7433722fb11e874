// MSDA forward, "region" kernel for gfx950 (variant 14): the adaptive kernel for encoder-shaped launches (queries = the
// pixels of the feature levels).
//
// Why: the wave-per-query kernel (msda.hip) is bound by the vector L1 / texture addresser -- every 16-byte-per-lane
// gather instruction costs ~16-19 CU cycles whatever it fetches (tools/l1_mask.hip: masking lanes or repeating a line
// inside an instruction saves nothing), and it needs 64 of them per (query, head): 822 MB through a 64 B/clk/CU pipe
// = 27 us per encoder launch at best.  The LDS serves the same 16 B per lane in 4 cycles.  Earlier LDS designs (8x8
// query tiles with bounding-box windows) lost the gain to per-tile bookkeeping: small tiles re-stage their halo, and the
// loc -> bounding box -> copy -> gather dependency chain is paid per tile.
//
// This kernel makes the work item as LARGE as the LDS allows and its window STATIC:
//   * item = (image, region, head).  The image plane is cut into RY x RX regions in normalised coordinates; a region
//     owns the queries of ALL four levels whose pixel lies in it (~196 queries at 600x1000 with 8x8 regions).
//   * per source level the item needs the pixels under its region plus a fixed halo (kHalo px of THAT level: sampling
//     offsets are expressed in pixels of the sampled level, deformable_detr.py:1067-1073).  The window depends on the
//     region only, so its copy starts at once -- nothing waits for sampling locations.  Out-of-level window pixels and
//     padded tokens are stored as zeros (== cuh:55-78 per-corner range checks, dd:1052 masked_fill), so staged samples
//     need no per-corner masks.
//   * the four source levels are processed one after the other through ONE window buffer (<= 70 KB: two workgroups per
//     CU); accumulators stay in registers across the passes, sample order 0..15 as in the reference.
//   * gather: lane = (query, channel quad) -- 8 queries per wave step -- one 16-byte record pair per sample in LDS
//     (4 premultiplied bilinear x attention weights, 2 window pixel indices), 4 ds_read_b128 + 8 packed FMAs per sample.
//   * samples that fall outside their window (large offsets) are gathered from global memory by the lanes concerned,
//     per sample, so results never depend on the halo; and a region whose sampled outlier rate is high (irregular
//     offsets: a trained model's far-reaching heads) switches, as a whole and consistently across its 8 heads'
//     workgroups, to the wave-per-query scheme (8 heads per wave, L1 gathers) -- "window where it fits, L1 where it
//     does not" inside one launch.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>
#include <stdlib.h>

#include "common.h"
#include "msda_common.h"

namespace {

using namespace egtr_msda;

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) f32x4* lds_f4p;   // explicit address spaces: LDS and global reads of one
typedef const __attribute__((address_space(1))) f32x4* glb_f4p;   // value must never be merged into flat loads

constexpr int kRW = 8;                 // waves per workgroup
constexpr int kRT = kRW * 64;          // threads
constexpr int kMaxSteps = 4;           // 8-query groups per wave -> at most 256 queries per region
constexpr int kWinPx = 528;            // window pixels per level (x 128 B)
constexpr int kZero = 2;               // all-zero pixels in front of the window (target of invalid samples)
constexpr int kHalo = 5;               // pixels of the sampled level around the region
constexpr int kProbeQ = 32;            // queries sampled (x 8 heads x 16 samples) for the per-region mode decision
constexpr int kFillIters = (kWinPx + 63) / 64;
// LDS carve-up in float4 units
constexpr int kOffWin = 0;                                   // (kZero + kWinPx) * 8
constexpr int kOffRecW = (kZero + kWinPx) * 8;               // [wave][group 2][query 8][sample 4] float4 weights
constexpr int kOffRecA = kOffRecW + kRW * 64;                // [wave][group 2][query 8][sample 4] uint2 addresses
constexpr int kOffMisc = kOffRecA + kRW * 32;                // 16: counters + the region's rectangles / windows
constexpr int kLdsF4 = kOffMisc + 16;
// s_misc (ints): [0] outliers, [1] valid samples of the probe, [4+l] queries of level l, [8+4l ..] query rectangle of
// level l {y0, x0, h, w}, [24+5s ..] window of source level s {y0, x0, h, w, staged}
constexpr int kMiscN = 4, kMiscRect = 8, kMiscWin = 24;
static_assert(kLdsF4 * 16 <= 81920, "two workgroups per CU");
// wave-per-query fallback: per wave 2 x 136 records of 16 B, aliased onto the window
constexpr int kHeadStride = 17, kWaveEntries = 8 * kHeadStride;
static_assert(2 * kRW * kWaveEntries <= (kZero + kWinPx) * 8, "fallback records fit in the window buffer");

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

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
  w.x0 = max((int)floorf(xlo) - kHalo, -1);
  w.y0 = max((int)floorf(ylo) - kHalo, -1);
  const int x1 = min((int)floorf(xhi) + kHalo + 1, SEL_W(G, s));
  const int y1 = min((int)floorf(yhi) + kHalo + 1, SEL_H(G, s));
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

// mode_force: 0 = adaptive (probe), 1 = always the window path (outliers per sample from global), 2 = always the
// wave-per-query path (tests / A-B timing).
template <bool FUSED>
__global__ __launch_bounds__(kRT, 4) void msda_fwd_region_f32(
    const float* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
    const float* __restrict__ loc, const float* __restrict__ attn, float* __restrict__ out, int B, int Lq, int S,
    const float* __restrict__ ref, float* __restrict__ attn_out, int ld_off, int ld_logit,
    const unsigned* __restrict__ keep_bits, int RY, int RX, int nblk, int mode_force, int abl) {
  __shared__ __attribute__((aligned(16))) float4 s_mem[kLdsF4];
  const int tid = threadIdx.x, lane = tid & 63, c4 = lane & 7, col = lane >> 3;
  const int wave = rfl(tid >> 6);
  const int item = rfl(xcd_remap(blockIdx.x, nblk));
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
  if (tid < 16) s_mem[kOffWin + tid] = make_float4(0.f, 0.f, 0.f, 0.f);  // the zero pixels
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

  const char* vhead = reinterpret_cast<const char*>(value) + (size_t)b * S * 1024 + head * 128 + c4 * 16;
  float4* s_win = s_mem + kOffWin;

  // Window copy of source level s by LDS-DMA: 64 pixels per workgroup iteration, wave w owns the contiguous 1 KiB piece
  // [it * 512 + w * 64, +64) float4 of the window, lane = (pixel, channel quad).  Pixels outside the level and padded
  // tokens are written as zeros by their lanes instead (cuh:55-78, dd:1052).  Nothing is waited for here.
  auto issue_fill = [&](int s) {
    const Win w = lds_win(s_misc, s);
    if (!w.staged || (abl & 1)) return;
    const int Hs = SEL_H(G, s), Ws = SEL_W(G, s), ss = SEL_S(G, s);
    const int wy0 = rfl(w.y0), wx0 = rfl(w.x0), ww = rfl(w.w), wh = rfl(w.h);
    if (!masked) {
      // row-structured copy: wave w takes window rows w, w + 8, ...; one DMA instruction = 8 consecutive pixels of a row
      // (lane = (pixel, channel quad)); everything but the lane's column offset is wave-uniform (scalar) arithmetic
      const int pc = lane >> 3;
      const int npc = (ww + 7) >> 3;
      for (int r = wave; r < wh; r += kRW) {
        const int y = wy0 + r;
        const bool rowin = (unsigned)y < (unsigned)Hs;
        const char* grow = vhead + (size_t)(ss + y * Ws + wx0) * 1024;   // pixel (y, wx0) of this head / quad
        for (int pp = 0; pp < npc; ++pp) {
          const int wxp = 8 * pp + pc;
          const bool colin = wxp < ww;
          const bool live = rowin && colin && (unsigned)(wx0 + wxp) < (unsigned)Ws;
          const int widx = kZero + r * ww + 8 * pp;   // first window pixel of the piece
          if (live) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(grow + (size_t)wxp * 1024),
                                             (__attribute__((address_space(3))) void*)(s_win + widx * 8), 16, 0, 0);
          } else if (colin) {
            s_win[(widx + pc) * 8 + c4] = make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
      }
      return;
    }
    // padded image: per-pixel mask test (64 pixels per workgroup iteration, wave w owns piece it * 8 + w)
    const int npx = ww * wh;
    const float inv = __frcp_rn((float)ww);
    for (int it = 0; it * 64 < npx; ++it) {
      const int pi = it * 64 + (tid >> 3);
      if (pi < npx) {
        const int wy = (int)(((float)pi + 0.5f) * inv);
        const int wx = pi - wy * ww;
        const int y = wy0 + wy, x = wx0 + wx;
        bool live = (unsigned)y < (unsigned)Hs && (unsigned)x < (unsigned)Ws;
        const int p = ss + y * Ws + x;
        if (live) live = (kb_g[p >> 5] >> (p & 31)) & 1u;
        if (live) {
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vhead + (size_t)p * 1024),
                                           (__attribute__((address_space(3))) void*)(s_win + kZero * 8 + it * 512 + wave * 64),
                                           16, 0, 0);
        } else {
          s_win[(kZero + pi) * 8 + c4] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    }
  };

  // ---- mode decision: the same sample of the region's queries (all 8 heads) in each of its 8 workgroups ----------------
  bool window_mode = grid && nq <= kRW * 8 * kMaxSteps && nq > 0;
  if (mode_force == 2) window_mode = false;
  if (window_mode) issue_fill(0);   // in flight under the probe and the pre-pass
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
    __syncthreads();   // a window copy may be in flight: its buffer is reused for this scheme's records
    int4* my_off = reinterpret_cast<int4*>(s_mem + kOffWin) + wave * kWaveEntries;
    float4* my_w = s_mem + kOffWin + kRW * kWaveEntries + wave * kWaveEntries;
    const unsigned* kb = kb_g;
    for (int idx = head + 8 * wave; idx < nq; idx += 8 * kRW) {
      const int q = region_query(s_misc, G, grid, lin0, idx);
      query_all_heads<FUSED>(value, loc, attn, out, ref, attn_out, ld_off, ld_logit, kb, G, b, q, Lq, S, lane, my_off,
                             my_w);
    }
    return;
  }

  // ---- window scheme --------------------------------------------------------------------------------------------------
  // pre-pass, once per query group: lane (col, c4) owns samples 2c4, 2c4+1 (both of level c4 >> 1) of its query and keeps
  // their finished records in registers: 4 bilinear x attention weights and one address code each --
  //   window sample : (pixel index of (y0, x0) in the window) | (pixel index of (y0+1, x0)) << 16
  //   invalid sample: 0 (the zero pixels; weights are 0)
  //   outlier       : 0x80000000 | clamped top-left pixel << 2 | dx << 1 | dy  (gathered from global memory)
  const int ngroups = (nq + 7) >> 3;
  const int my_lvl = c4 >> 1;
  float wq[kMaxSteps][2][4];
  unsigned code[kMaxSteps][2];
  int qv[kMaxSteps];
  f32x2 acc0[kMaxSteps], acc1[kMaxSteps];
  {
    const int Hs = SEL_H(G, my_lvl), Ws = SEL_W(G, my_lvl), ss = SEL_S(G, my_lvl);
    const float fw = (float)Ws, fh = (float)Hs;
    const Win w = lds_win(s_misc, my_lvl);
    float4 lcs[kMaxSteps];
    float2 aws[kMaxSteps], rps[kMaxSteps];
#pragma unroll
    for (int st = 0; st < kMaxSteps; ++st) {   // every group's loads first: one memory round trip
      const int g = wave + st * kRW;
      const int idx = g * 8 + col;
      acc0[st] = (f32x2){0.f, 0.f};
      acc1[st] = (f32x2){0.f, 0.f};
      qv[st] = -1;
      lcs[st] = make_float4(9.f, 9.f, 9.f, 9.f);   // far outside: invalid
      aws[st] = make_float2(0.f, 0.f);
      rps[st] = make_float2(0.f, 0.f);
      if (g < ngroups && idx < nq && !(abl & 4)) {
        const int q = region_query(s_misc, G, grid, lin0, idx);
        qv[st] = q;
        const size_t gq = (size_t)b * Lq + q;
        lcs[st] = *reinterpret_cast<const float4*>(loc + gq * (FUSED ? ld_off : 256) + head * 32 + c4 * 4);
        aws[st] = *reinterpret_cast<const float2*>(attn + gq * (FUSED ? ld_logit : 128) + head * 16 + c4 * 2);
        if (FUSED) rps[st] = *reinterpret_cast<const float2*>(ref + (gq * 4 + my_lvl) * 2);
      }
    }
#pragma unroll
    for (int st = 0; st < kMaxSteps; ++st) {
      __builtin_amdgcn_sched_barrier(0);   // one group's geometry at a time (register budget)
      float4 lc = lcs[st];
      float2 aw = aws[st];
      if (FUSED) {  // all lanes take part in the DPP reduce (padding slots hold zeros)
        if (qv[st] >= 0) {
          const float2 r = rps[st];
          lc = make_float4(r.x + lc.x / fw, r.y + lc.y / fh, r.x + lc.z / fw, r.y + lc.w / fh);
        }
        softmax16<FUSED>(aw);
        if (qv[st] < 0) aw = make_float2(0.f, 0.f);
        if (attn_out != nullptr && qv[st] >= 0)
          *reinterpret_cast<float2*>(attn_out + ((size_t)b * Lq + qv[st]) * 128 + head * 16 + c4 * 2) = aw;
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float x = (j ? lc.z : lc.x) * fw - 0.5f, y = (j ? lc.w : lc.y) * fh - 0.5f;
        const bool val = (y > -1.f) && (x > -1.f) && (y < fh) && (x < fw);
        x = val ? x : 0.f;
        y = val ? y : 0.f;
        const float a = val ? (j ? aw.y : aw.x) : 0.f;
        const float yf = floorf(y), xf = floorf(x);
        const float lh = y - yf, lw = x - xf, hh = 1.f - lh, hw = 1.f - lw;
        const int y0 = (int)yf, x0 = (int)xf;
        float w0 = hh * hw * a, w1 = hh * lw * a, w2 = lh * hw * a, w3 = lh * lw * a;
        const bool in = w.staged && y0 >= w.y0 && x0 >= w.x0 && y0 + 1 < w.y0 + w.h && x0 + 1 < w.x0 + w.w;
        unsigned cd = 0u;
        if (val && in) {
          const int p00 = kZero + (y0 - w.y0) * w.w + (x0 - w.x0);
          cd = (unsigned)p00 | ((unsigned)(p00 + w.w) << 16);
        } else if (val) {
          // outside the window: out-of-range corners and padded tokens folded into the weights (cuh:55-78, dd:1052)
          const int ya = max(y0, 0), yb = min(y0 + 1, Hs - 1), xa = max(x0, 0), xb = min(x0 + 1, Ws - 1);
          const int p00 = ss + ya * Ws + xa;
          const int dx = xb - xa, dy = yb - ya;
          bool k0 = y0 >= 0 && x0 >= 0, k1 = y0 >= 0 && x0 + 1 <= Ws - 1, k2 = y0 + 1 <= Hs - 1 && x0 >= 0,
               k3 = y0 + 1 <= Hs - 1 && x0 + 1 <= Ws - 1;
          if (masked) {
            const int p01 = p00 + dx, p10 = p00 + dy * Ws, p11 = p10 + dx;
            k0 = k0 && ((kb_g[p00 >> 5] >> (p00 & 31)) & 1u);
            k1 = k1 && ((kb_g[p01 >> 5] >> (p01 & 31)) & 1u);
            k2 = k2 && ((kb_g[p10 >> 5] >> (p10 & 31)) & 1u);
            k3 = k3 && ((kb_g[p11 >> 5] >> (p11 & 31)) & 1u);
          }
          w0 = k0 ? w0 : 0.f;
          w1 = k1 ? w1 : 0.f;
          w2 = k2 ? w2 : 0.f;
          w3 = k3 ? w3 : 0.f;
          cd = 0x80000000u | (unsigned)((p00 << 2) | (dx << 1) | dy);
        }
        wq[st][j][0] = w0;
        wq[st][j][1] = w1;
        wq[st][j][2] = w2;
        wq[st][j][3] = w3;
        code[st][j] = cd;
      }
    }
  }

  float4* rec_w = s_mem + kOffRecW + wave * 64;                          // [group 2][query col 8][sample p 4]
  uint2* rec_a = reinterpret_cast<uint2*>(s_mem + kOffRecA) + wave * 64;   // [group 2][query col 8][sample p 4]
  typedef const __attribute__((address_space(3))) char* lds_cp;
  const lds_cp lbase = (lds_cp)(s_win) + c4 * 16;   // + a record's byte address = this lane's quad of that pixel
  const glb_f4p gval = (glb_f4p)(vhead);

#pragma unroll 1
  for (int s = 0; s < 4; ++s) {
    const int Ws = SEL_W(G, s);
    if (s > 0) {
      __syncthreads();  // the previous level's gathers are done with the window buffer
      issue_fill(s);
    }
    __syncthreads();    // window copy landed (the fence of the barrier drains the DMA), zero pixels written

#pragma unroll
    for (int pr = 0; pr < kMaxSteps; pr += 2) {       // two query groups at a time: their LDS latencies overlap
      if (wave + pr * kRW >= ngroups || (abl & 2)) continue;   // wave-uniform
      if (my_lvl == s) {
        // owner lanes publish their two samples' records (points 2(c4&1), +1 of level s) for both groups
        const int p = 2 * (c4 & 1);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int st = pr + u;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            rec_w[u * 32 + col * 4 + p + j] = make_float4(wq[st][j][0], wq[st][j][1], wq[st][j][2], wq[st][j][3]);
            const unsigned cd = code[st][j];
            // byte addresses of the two window rows (bit 31 of .x: outlier, .x then carries the global code)
            rec_a[u * 32 + col * 4 + p + j] =
                (cd >> 31) ? make_uint2(cd, 0u) : make_uint2((cd & 0xffffu) << 7, (cd >> 16) << 7);
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

      const bool two = wave + (pr + 1) * kRW < ngroups;   // wave-uniform (odd number of groups: the second is empty)
      uint4 A01[2], A23[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        A01[u] = *reinterpret_cast<const uint4*>(rec_a + u * 32 + col * 4);
        A23[u] = *reinterpret_cast<const uint4*>(rec_a + u * 32 + col * 4 + 2);
      }
      const bool any_out = __any((int)((A01[0].x | A01[0].z | A23[0].x | A23[0].z | A01[1].x | A01[1].z | A23[1].x |
                                        A23[1].z) >> 31));
      if (!any_out) {
        // every sample of both groups is in the window: 4 ds_read_b128 + 8 packed FMAs per sample, nothing else
#pragma unroll
        for (int p = 0; p < 4; ++p) {
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            if (u == 1 && !two) continue;
            const int st = pr + u;
            const unsigned a0s = p == 0 ? A01[u].x : p == 1 ? A01[u].z : p == 2 ? A23[u].x : A23[u].z;
            const unsigned a1s = p == 0 ? A01[u].y : p == 1 ? A01[u].w : p == 2 ? A23[u].y : A23[u].w;
            const float4 wv = rec_w[u * 32 + col * 4 + p];
            const lds_f4p r0 = (lds_f4p)(lbase + a0s), r1 = (lds_f4p)(lbase + a1s);
            const f32x4 v00 = r0[0], v01 = r0[8], v10 = r1[0], v11 = r1[8];
            f32x2 a0 = acc0[st], a1 = acc1[st];
            a0 += (f32x2){wv.x, wv.x} * (f32x2){v00.x, v00.y};
            a1 += (f32x2){wv.x, wv.x} * (f32x2){v00.z, v00.w};
            a0 += (f32x2){wv.y, wv.y} * (f32x2){v01.x, v01.y};
            a1 += (f32x2){wv.y, wv.y} * (f32x2){v01.z, v01.w};
            a0 += (f32x2){wv.z, wv.z} * (f32x2){v10.x, v10.y};
            a1 += (f32x2){wv.z, wv.z} * (f32x2){v10.z, v10.w};
            a0 += (f32x2){wv.w, wv.w} * (f32x2){v11.x, v11.y};
            a1 += (f32x2){wv.w, wv.w} * (f32x2){v11.z, v11.w};
            acc0[st] = a0;
            acc1[st] = a1;
          }
          __builtin_amdgcn_sched_barrier(0);   // one sample of each group in flight at a time (register budget)
        }
      } else {
        // some lane has a sample outside its window: those lanes gather that sample from global memory
#pragma unroll 1
        for (int p = 0; p < 4; ++p) {
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            if (u == 1 && !two) continue;
            const int st = pr + u;
            const uint2 ar = rec_a[u * 32 + col * 4 + p];
            const float4 wv = rec_w[u * 32 + col * 4 + p];
            f32x4 v00, v01, v10, v11;
            if (ar.x >> 31) {
              const unsigned c = ar.x & 0x7fffffffu;
              const glb_f4p gp = gval + (size_t)(c >> 2) * 64;
              const size_t ox = (size_t)((c >> 1) & 1u) * 64, oy = (size_t)(c & 1u) * Ws * 64;
              v00 = gp[0];
              v01 = gp[ox];
              v10 = gp[oy];
              v11 = gp[oy + ox];
            } else {
              const lds_f4p r0 = (lds_f4p)(lbase + ar.x), r1 = (lds_f4p)(lbase + ar.y);
              v00 = r0[0]; v01 = r0[8]; v10 = r1[0]; v11 = r1[8];
            }
            f32x2 a0 = acc0[st], a1 = acc1[st];
            a0 += (f32x2){wv.x, wv.x} * (f32x2){v00.x, v00.y};
            a1 += (f32x2){wv.x, wv.x} * (f32x2){v00.z, v00.w};
            a0 += (f32x2){wv.y, wv.y} * (f32x2){v01.x, v01.y};
            a1 += (f32x2){wv.y, wv.y} * (f32x2){v01.z, v01.w};
            a0 += (f32x2){wv.z, wv.z} * (f32x2){v10.x, v10.y};
            a1 += (f32x2){wv.z, wv.z} * (f32x2){v10.z, v10.w};
            a0 += (f32x2){wv.w, wv.w} * (f32x2){v11.x, v11.y};
            a1 += (f32x2){wv.w, wv.w} * (f32x2){v11.z, v11.w};
            acc0[st] = a0;
            acc1[st] = a1;
          }
        }
      }
      // this wave's records are rewritten by its next pair of groups
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }

#pragma unroll
  for (int st = 0; st < kMaxSteps; ++st) {
    if (qv[st] >= 0)
      *reinterpret_cast<float4*>(out + ((size_t)b * Lq + qv[st]) * 256 + head * 32 + c4 * 4) =
          make_float4(acc0[st].x, acc0[st].y, acc1[st].x, acc1[st].y);
  }
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
  static const int abl = [] { const char* e = getenv("EGTR_REGION_ABLATE"); return e ? atoi(e) : 0; }();  // timing only
  if (ref != nullptr)
    hipLaunchKernelGGL(msda_fwd_region_f32<true>, dim3((unsigned)nblk), dim3(kRT), 0, st, value, shapes, lsi, loc, attn,
                       out, B, Lq, S, ref, attn_out, ld_off, ld_logit, keep_bits, RY, RX, (int)nblk, mode, abl);
  else
    hipLaunchKernelGGL(msda_fwd_region_f32<false>, dim3((unsigned)nblk), dim3(kRT), 0, st, value, shapes, lsi, loc, attn,
                       out, B, Lq, S, nullptr, nullptr, 256, 128, nullptr, RY, RX, (int)nblk, mode, abl);
  return egtr_check_launch();
}
