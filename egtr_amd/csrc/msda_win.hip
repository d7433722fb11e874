// MSDA forward, "window" kernel for gfx950 (variants 8 / 9): the fourth LDS design, built on what the first three
// measured.
//
// The wave-per-query kernel (msda.hip) pulls 8 KiB per (query, head) through the vector L1 (841 MB per encoder
// launch at 600x1000) and is bound by the L1 gather / miss rate, not by HBM.  Here a workgroup owns a TH x TW tile of
// encoder queries x ONE head, stages the per-level bounding windows of the tile's samples once in LDS and serves all
// 16 x 4 corner reads from there (ds_read_b128: 256 B/clk/CU, 4x the L1 rate).  Differences to msda_tile.hip:
//   * ONE 16-byte record per sample {LDS addr of row 0 | row 1 << 16, lw, lh, attn} instead of two (offsets +
//     weights): 5 instead of 6 ds_read_b128 per sample; the four bilinear weights are rebuilt with 4 multiplies.
//     Out-of-range corners need no masks: the window is the bounding box of the UNCLAMPED corners and the staging
//     copy zero-fills pixels outside the level (and padded tokens in the fused path); entirely invalid samples point
//     at a two-pixel zero region.
//   * gather lanes are mapped so that each hardware ds_read_b128 lane group ({0-3,12-15,20-27}, ...) holds the 8
//     channel quads of TWO x-adjacent queries: their pixels are equal (broadcast) or neighbours (different halves of
//     the 64 banks) -> conflict-free for the regular part of the sampling pattern.
//   * small workgroups (4 x 8 queries = 256 threads, 40 KB LDS -> 4 per CU, or 8 x 8) that are NOT persistent: the
//     dependent global round trips of one work item (loc/attn -> bounding boxes -> window copy) are hidden by the
//     other workgroups of the CU instead of by prefetch code.
//   * work order per XCD: coarse-level tiles first (their level-0 windows do not fit and are gathered from global
//     memory = the slow items), then that XCD's raster chunk of every level: compact footprint per private L2.
// Any level whose window does not fit the LDS budget is gathered from global memory, so results never depend on the
// windows, only the speed does.  M = 8, D = 32, L*P = 16, P even.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>
#include <stdlib.h>

#include "common.h"
#include "msda_common.h"

using namespace egtr_msda;

namespace {

#ifndef EGTR_WIN_ABLATE
#define EGTR_WIN_ABLATE 0  // timing ablations (wrong results): 1 = no window copy, 2 = no gather
#endif

constexpr int kZeroPx = 2;  // all-zero pixels at the start of the window buffer (target of invalid samples)

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ int dpp_xor8_min(int v) {
  // lanes l and l^8 of a row of 16: row_ror:8
  return min(v, __builtin_amdgcn_update_dpp(v, v, 0x128, 0xf, 0xf, false));
}
__device__ __forceinline__ int dpp_xor8_max(int v) {
  return max(v, __builtin_amdgcn_update_dpp(v, v, 0x128, 0xf, 0xf, false));
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// loads with an explicit address space (LDS / global): they can never be merged into flat loads
__device__ __forceinline__ float4 lds_ld4(const __attribute__((address_space(3))) char* p) {
  const f32x4 v = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(p);
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 glb_ld4(const __attribute__((address_space(1))) char* p) {
  const f32x4 v = *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(p);
  return make_float4(v.x, v.y, v.z, v.w);
}

// acc[0..1] += w * v[0..1]: one v_pk_fma_f32 with the scalar weight broadcast by op_sel
__device__ __forceinline__ f32x2 pk_fma(f32x2 v, float w, f32x2 acc) {
  const f32x2 ww = {w, w};
  return __builtin_elementwise_fma(v, ww, acc);
}

template <bool FUSED, int TH, int TW, int WINPX, int WPS>
__global__ __launch_bounds__(TH * TW * 8, WPS) void msda_fwd_win_f32(
    const float* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
    const float* __restrict__ loc, const float* __restrict__ attn, float* __restrict__ out, int B, int Lq, int S,
    int L, int P, const float* __restrict__ ref, float* __restrict__ attn_out, int ld_off, int ld_logit,
    const unsigned char* __restrict__ keep, const unsigned* __restrict__ keep_bits,
    unsigned long long* __restrict__ prof) {
  constexpr int TQ = TH * TW;
  constexpr int NW = TQ / 8;    // waves per workgroup
  constexpr int RS = TQ + 1;    // record stride per sample (+1: conflict-free record writes)
  static_assert(TW == 8, "a wave gathers one row of 8 x-adjacent queries");
  static_assert((kZeroPx + WINPX) * 128 < 65536, "LDS addresses are packed into 16 bits");
  __shared__ __attribute__((aligned(16))) float4 s_win[(kZeroPx + WINPX) * 8];
  __shared__ __attribute__((aligned(16))) float4 s_w[16 * RS];  // [sample][query] bilinear x attention weights
  __shared__ __attribute__((aligned(16))) uint4 s_a[4 * RS];    // [sample / 4][query][sample % 4] packed addresses
  __shared__ int s_bbox[16];                                     // [level][ymin, ymax, xmin, xmax]

  const int tid = threadIdx.x, ql = tid >> 3, c4 = tid & 7;
  const int wave = rfl(tid >> 6);
  // gather-phase lane -> (query slot, channel quad): see the header comment
  int gq, gc;
  {
    const int lane = tid & 63, l5 = lane & 31;
    if (l5 < 4) { gq = 0; gc = l5; }
    else if (l5 < 12) { gq = 2; gc = l5 - 4; }
    else if (l5 < 16) { gq = 0; gc = l5 - 8; }
    else if (l5 < 20) { gq = 3; gc = l5 - 16; }
    else if (l5 < 28) { gq = 1; gc = l5 - 20; }
    else { gq = 3; gc = l5 - 24; }
    gq += (lane >> 5) * 4 + (tid >> 6) * 8;
  }
  LevelGeom G;
  load_geom(shapes, lsi, L, G);
  const TileMap tm = make_tile_map<TH, TW>(G, L, Lq);
  // tiles per level and this XCD's raster chunk of each level
  const int xcd = blockIdx.x & 7;
  const int n0 = tm.grid2d ? tm.nt0 : tm.ntiles, n1 = tm.grid2d ? tm.nt1 : 0, n2 = tm.grid2d ? tm.nt2 : 0,
            n3 = tm.grid2d ? tm.nt3 : 0;
  const int lo0 = (xcd * n0) >> 3, lo1 = (xcd * n1) >> 3, lo2 = (xcd * n2) >> 3, lo3 = (xcd * n3) >> 3;
  const int c0 = (((xcd + 1) * n0) >> 3) - lo0, c1 = (((xcd + 1) * n1) >> 3) - lo1,
            c2 = (((xcd + 1) * n2) >> 3) - lo2, c3 = (((xcd + 1) * n3) >> 3) - lo3;
  const int per_img = (c0 + c1 + c2 + c3) * 8;
  const int nwork = B * per_img;
  const int nwords = (S + 31) >> 5;
  // per-lane constants of the geometry phase: samples 2*c4, 2*c4+1 lie in level `lvl` (P is even)
  const int lvl = (2 * c4) / P;
  const int H = SEL_H(G, lvl), W = SEL_W(G, lvl), st = SEL_S(G, lvl);
  const float fW = (float)W, fH = (float)H;
  if (tid < kZeroPx * 8) s_win[tid] = make_float4(0.f, 0.f, 0.f, 0.f);

  for (int k = blockIdx.x >> 3; k < nwork; k += gridDim.x >> 3) {
    // ---- decode the work item: (batch, tile, head); coarse levels first ------------------------------------------
    const int b = (B == 1) ? 0 : k / per_img;
    const int r = k - b * per_img;
    const int head = r & 7;
    int t = r >> 3, tile;
    if (t < c3) tile = n0 + n1 + n2 + lo3 + t;
    else if ((t -= c3) < c2) tile = n0 + n1 + lo2 + t;
    else if ((t -= c2) < c1) tile = n0 + lo1 + t;
    else tile = lo0 + (t - c1);
    const char* vbase = reinterpret_cast<const char*>(value) + (size_t)b * S * 1024 + head * 128;
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    if (prof) t0 = __builtin_amdgcn_s_memtime();

    // ---- P0: loc / attn of (query ql, head), samples 2*c4 and 2*c4+1 ----------------------------------------------
    const int q = tile_query<TH, TW>(tm, G, tile, ql, Lq);
    float4 lc = make_float4(9.f, 9.f, 9.f, 9.f);  // far outside -> invalid
    float2 aw = make_float2(0.f, 0.f);
    if (q >= 0) {
      const size_t qg = (size_t)b * Lq + q;
      if (FUSED) {
        lc = reinterpret_cast<const float4*>(loc + qg * ld_off)[head * 8 + c4];
        aw = reinterpret_cast<const float2*>(attn + qg * ld_logit)[head * 8 + c4];
        const float2 rp = *reinterpret_cast<const float2*>(ref + (qg * L + lvl) * 2);
        lc = make_float4(rp.x + lc.x / fW, rp.y + lc.y / fH, rp.x + lc.z / fW, rp.y + lc.w / fH);
      } else {
        lc = reinterpret_cast<const float4*>(loc + (qg * 8 + head) * 32)[c4];
        aw = reinterpret_cast<const float2*>(attn + (qg * 8 + head) * 16)[c4];
      }
    }
    if (FUSED) {
      // softmax over the 16 logits of the head: 8 lanes x 2 (same arithmetic as msda_fwd_q64_f32<true>)
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
      if (q >= 0 && attn_out != nullptr)
        reinterpret_cast<float2*>(attn_out + (((size_t)b * Lq + q) * 8 + head) * 16)[c4] = aw;
      if (q < 0) aw = make_float2(0.f, 0.f);
    }

    if (tid < 16) s_bbox[tid] = (tid & 1) ? INT_MIN : INT_MAX;
    __syncthreads();  // also: every thread has finished the previous item's LDS reads
    if (prof) t1 = __builtin_amdgcn_s_memtime();

    // ---- A: geometry (cuh:38-78 / 268-288) + per-level bounding boxes of the UNCLAMPED corners --------------------
    int y0[2], x0[2];
    float wq[2][4];  // hh*hw, hh*lw, lh*hw, lh*lw, each x attention
    bool val[2];
    int ymin = INT_MAX, ymax = INT_MIN, xmin = INT_MAX, xmax = INT_MIN;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float x = (j ? lc.z : lc.x) * fW - 0.5f, y = (j ? lc.w : lc.y) * fH - 0.5f;
      val[j] = (y > -1.f) && (x > -1.f) && (y < fH) && (x < fW);
      const float yf = floorf(y), xf = floorf(x);
      const float lh = y - yf, lw = x - xf, hh = 1.f - lh, hw = 1.f - lw;
      const float a = val[j] ? (j ? aw.y : aw.x) : 0.f;
      y0[j] = val[j] ? (int)yf : 0;
      x0[j] = val[j] ? (int)xf : 0;
      // invalid samples (incl. NaN locations): zero weights whatever the fractions are
      wq[j][0] = val[j] ? hh * hw * a : 0.f;
      wq[j][1] = val[j] ? hh * lw * a : 0.f;
      wq[j][2] = val[j] ? lh * hw * a : 0.f;
      wq[j][3] = val[j] ? lh * lw * a : 0.f;
      ymin = min(ymin, val[j] ? y0[j] : INT_MAX);
      ymax = max(ymax, val[j] ? y0[j] + 1 : INT_MIN);
      xmin = min(xmin, val[j] ? x0[j] : INT_MAX);
      xmax = max(xmax, val[j] ? x0[j] + 1 : INT_MIN);
    }
    // lanes l, l^8, l^16, l^32 hold the same c4 (same level): reduce over the 8 queries of the wave
    ymin = dpp_xor8_min(ymin);
    ymax = dpp_xor8_max(ymax);
    xmin = dpp_xor8_min(xmin);
    xmax = dpp_xor8_max(xmax);
#pragma unroll
    for (int o = 16; o < 64; o <<= 1) {
      ymin = min(ymin, __shfl_xor(ymin, o));
      ymax = max(ymax, __shfl_xor(ymax, o));
      xmin = min(xmin, __shfl_xor(xmin, o));
      xmax = max(xmax, __shfl_xor(xmax, o));
    }
    if ((tid & 63) < 8 && ymin <= ymax) {
      atomicMin(&s_bbox[lvl * 4 + 0], ymin);
      atomicMax(&s_bbox[lvl * 4 + 1], ymax);
      atomicMin(&s_bbox[lvl * 4 + 2], xmin);
      atomicMax(&s_bbox[lvl * 4 + 3], xmax);
    }
    __syncthreads();
    if (prof) t2 = __builtin_amdgcn_s_memtime();

    // ---- B: pack the windows (coarse level first) into the LDS budget; records; window copy -----------------------
    int wy0[4], wx0[4], ww[4], wh[4], base[4];
    unsigned staged = 0;
    {
      int off = 0;
#pragma unroll
      for (int l = 3; l >= 0; --l) {
        wy0[l] = rfl(s_bbox[l * 4 + 0]);
        const int wy1 = rfl(s_bbox[l * 4 + 1]);
        wx0[l] = rfl(s_bbox[l * 4 + 2]);
        const int wx1 = rfl(s_bbox[l * 4 + 3]);
        const bool empty = (l >= L) || (wy0[l] > wy1);
        ww[l] = empty ? 0 : (wx1 - wx0[l] + 1);
        wh[l] = empty ? 0 : (wy1 - wy0[l] + 1);
        base[l] = off;
        if (!empty && off + ww[l] * wh[l] <= WINPX) {
          staged |= 1u << l;
          off += ww[l] * wh[l];
        } else {
          wh[l] = 0;  // not staged: nothing to copy
        }
      }
    }
    // window copy, asynchronous global -> LDS (global_load_lds_dwordx4: 64 lanes x 16 B land contiguously at M0):
    // one wave-instruction = 8 consecutive pixels of one window row; rows are dealt round-robin to the waves.
    // Pixels outside the level (the zero apron) and padded tokens are zero-filled with ordinary LDS stores.
#pragma unroll
    for (int l = 3; l >= 0; --l) {
      if (wh[l] == 0) continue;
      const int Hl = sel4(G.H0, G.H1, G.H2, G.H3, l), Wl = sel4(G.W0, G.W1, G.W2, G.W3, l);
      const int sl = sel4(G.s0, G.s1, G.s2, G.s3, l);
      for (int rr = wave; rr < wh[l]; rr += NW) {
        const int yy = wy0[l] + rr;
        const bool rowok = (unsigned)yy < (unsigned)Hl;
        const char* grow = vbase + (size_t)(sl + yy * Wl) * 1024;
        const int lrow = kZeroPx + base[l] + rr * ww[l];
        for (int cb = 0; cb < ww[l]; cb += 8) {
          const int cc = cb + ql - wave * 8;  // this lane's window column (ql - wave*8 = lane >> 3)
          const int xx = wx0[l] + cc;
          bool inb = rowok && cc < ww[l] && (unsigned)xx < (unsigned)Wl;
          if (FUSED && inb) {
            const int pix = sl + yy * Wl + xx;
            if (keep_bits != nullptr) inb = (keep_bits[(size_t)b * nwords + (pix >> 5)] >> (pix & 31)) & 1u;
            else if (keep != nullptr) inb = keep[(size_t)b * S + pix] != 0;
          }
          if (inb) {
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(grow + (unsigned)(xx * 1024 + c4 * 16)),
                (__attribute__((address_space(3))) void*)(s_win + (lrow + cb) * 8), 16, 0, 0);
          } else if (cc < ww[l]) {
            s_win[(lrow + cc) * 8 + c4] = make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
      }
    }
    {
      // records of this thread's two samples
      const bool st_l = (staged >> lvl) & 1u;
      const int by0 = sel4(wy0[0], wy0[1], wy0[2], wy0[3], lvl), bx0 = sel4(wx0[0], wx0[1], wx0[2], wx0[3], lvl);
      const int bww = sel4(ww[0], ww[1], ww[2], ww[3], lvl), bbase = sel4(base[0], base[1], base[2], base[3], lvl);
      unsigned code[2];
      if (st_l) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int a00 = (kZeroPx + bbase + (y0[j] - by0) * bww + (x0[j] - bx0)) * 128;
          code[j] = val[j] ? (unsigned)(a00 | ((a00 + bww * 128) << 16)) : 0u;
        }
      } else {
        // global records: clamped top-left pixel << 2 | dx << 1 | dy; out-of-range corners (and padded tokens) are
        // folded into the weights
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int ya = max(y0[j], 0), yb = min(y0[j] + 1, H - 1), xa = max(x0[j], 0), xb = min(x0[j] + 1, W - 1);
          const int p00 = st + ya * W + xa;
          const int dx = xb - xa, dy = yb - ya;
          bool k0 = y0[j] >= 0 && x0[j] >= 0, k1 = y0[j] >= 0 && x0[j] + 1 <= W - 1,
               k2 = y0[j] + 1 <= H - 1 && x0[j] >= 0, k3 = y0[j] + 1 <= H - 1 && x0[j] + 1 <= W - 1;
          if (FUSED && val[j] && (keep_bits != nullptr || keep != nullptr)) {
            const int p01 = p00 + dx, p10 = p00 + dy * W, p11 = p10 + dx;
            if (keep_bits != nullptr) {
              const unsigned* kb = keep_bits + (size_t)b * nwords;
              k0 = k0 && ((kb[p00 >> 5] >> (p00 & 31)) & 1u);
              k1 = k1 && ((kb[p01 >> 5] >> (p01 & 31)) & 1u);
              k2 = k2 && ((kb[p10 >> 5] >> (p10 & 31)) & 1u);
              k3 = k3 && ((kb[p11 >> 5] >> (p11 & 31)) & 1u);
            } else {
              const unsigned char* kp = keep + (size_t)b * S;
              k0 = k0 && kp[p00];
              k1 = k1 && kp[p01];
              k2 = k2 && kp[p10];
              k3 = k3 && kp[p11];
            }
          }
          wq[j][0] = k0 ? wq[j][0] : 0.f;
          wq[j][1] = k1 ? wq[j][1] : 0.f;
          wq[j][2] = k2 ? wq[j][2] : 0.f;
          wq[j][3] = k3 ? wq[j][3] : 0.f;
          code[j] = val[j] ? (unsigned)((p00 << 2) | (dx << 1) | dy) : 0u;
        }
      }
      s_w[(2 * c4) * RS + ql] = make_float4(wq[0][0], wq[0][1], wq[0][2], wq[0][3]);
      s_w[(2 * c4 + 1) * RS + ql] = make_float4(wq[1][0], wq[1][1], wq[1][2], wq[1][3]);
      reinterpret_cast<uint2*>(s_a + (c4 >> 1) * RS + ql)[c4 & 1] = make_uint2(code[0], code[1]);
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wave's window pieces have landed
    __syncthreads();
    if (prof) t3 = __builtin_amdgcn_s_memtime();

    // ---- C: gather.  Lane (gq, gc); the LDS / global choice is uniform per level ----------------------------------
    {
      const int qo = tile_query<TH, TW>(tm, G, tile, gq, Lq);
      const char* win = reinterpret_cast<const char*>(s_win);
      const unsigned coff = gc * 16;
      const char* glb = vbase + gc * 16;
      f32x2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
      const int G4 = P >> 2;  // groups of 4 samples per level (P is 4, 8 or 16)
      for (int l = 0; l < L; ++l) {
        if ((staged >> l) & 1u) {
          for (int g = 0; g < G4; ++g) {
            const int sg = l * G4 + g;
            const uint4 A = s_a[sg * RS + gq];
            const unsigned cd[4] = {A.x, A.y, A.z, A.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float4 w = s_w[(sg * 4 + j) * RS + gq];
              const unsigned a0 = (cd[j] & 0xffffu) | coff, a1 = (cd[j] >> 16) | coff;
              const float4 v0 = *reinterpret_cast<const float4*>(win + a0);
              const float4 v1 = *reinterpret_cast<const float4*>(win + a0 + 128);
              const float4 v2 = *reinterpret_cast<const float4*>(win + a1);
              const float4 v3 = *reinterpret_cast<const float4*>(win + a1 + 128);
              acc0 = pk_fma(f32x2{v0.x, v0.y}, w.x, acc0);
              acc1 = pk_fma(f32x2{v0.z, v0.w}, w.x, acc1);
              acc0 = pk_fma(f32x2{v1.x, v1.y}, w.y, acc0);
              acc1 = pk_fma(f32x2{v1.z, v1.w}, w.y, acc1);
              acc0 = pk_fma(f32x2{v2.x, v2.y}, w.z, acc0);
              acc1 = pk_fma(f32x2{v2.z, v2.w}, w.z, acc1);
              acc0 = pk_fma(f32x2{v3.x, v3.y}, w.w, acc0);
              acc1 = pk_fma(f32x2{v3.z, v3.w}, w.w, acc1);
            }
          }
        } else {
          const unsigned Wb = (unsigned)sel4(G.W0, G.W1, G.W2, G.W3, l) * 1024u;
          for (int pp = 0; pp < P; ++pp) {
            const int s = l * P + pp;
            const unsigned cdw = reinterpret_cast<const unsigned*>(s_a + (s >> 2) * RS + gq)[s & 3];
            const float4 w = s_w[s * RS + gq];
            const unsigned o00 = (cdw >> 2) << 10;
            const unsigned dxb = (cdw & 2u) ? 1024u : 0u, dyb = (cdw & 1u) ? Wb : 0u;
            const float4 v0 = *reinterpret_cast<const float4*>(glb + o00);
            const float4 v1 = *reinterpret_cast<const float4*>(glb + o00 + dxb);
            const float4 v2 = *reinterpret_cast<const float4*>(glb + o00 + dyb);
            const float4 v3 = *reinterpret_cast<const float4*>(glb + o00 + dyb + dxb);
            acc0 = pk_fma(f32x2{v0.x, v0.y}, w.x, acc0);
            acc1 = pk_fma(f32x2{v0.z, v0.w}, w.x, acc1);
            acc0 = pk_fma(f32x2{v1.x, v1.y}, w.y, acc0);
            acc1 = pk_fma(f32x2{v1.z, v1.w}, w.y, acc1);
            acc0 = pk_fma(f32x2{v2.x, v2.y}, w.z, acc0);
            acc1 = pk_fma(f32x2{v2.z, v2.w}, w.z, acc1);
            acc0 = pk_fma(f32x2{v3.x, v3.y}, w.w, acc0);
            acc1 = pk_fma(f32x2{v3.z, v3.w}, w.w, acc1);
          }
        }
      }
      if (qo >= 0)
        reinterpret_cast<float4*>(out + (((size_t)b * Lq + qo) * 8 + head) * 32)[gc] =
            make_float4(acc0.x, acc0.y, acc1.x, acc1.y);
    }
    if (prof) {
      const unsigned long long t4 = __builtin_amdgcn_s_memtime();
      if (tid == 0) {
        atomicAdd(prof + 0, t1 - t0);
        atomicAdd(prof + 1, t2 - t1);
        atomicAdd(prof + 2, t3 - t2);
        atomicAdd(prof + 3, t4 - t3);
        atomicAdd(prof + 4, 1ull);
        atomicAdd(prof + 5, (unsigned long long)__popc(staged));
      }
    }
    // the next iteration's first barrier orders its LDS writes after every thread's reads of this one
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Pipelined form (variants 11 / 12): persistent workgroups, every LDS structure double-buffered.  While the workgroup
// gathers item j from buffer j&1, the window copy (LDS-DMA) of item j+1 lands in the other buffer and the loc / attn
// rows of item j+2 are in flight into registers, so no global round trip is exposed in steady state:
//     X: [DMA(j) landed, records(j) visible]                           (workgroup barrier)
//        geometry + bounding boxes of item j+1 (registers -> LDS atomics); issue loc/attn loads of item j+2
//     Y: [bounding boxes complete]                                      (workgroup barrier)
//        pack windows, issue DMA(j+1), write records(j+1)
//        gather item j from LDS, wait for DMA(j+1), store the outputs
// Barriers are raw s_barrier + lgkmcnt(0): a __syncthreads() fence would drain the in-flight DMA / output stores.
// What the counters of the first pipelined build said: the kernel is bound by INSTRUCTION ISSUE (one
// VALU and one SALU instruction per SIMD per 4 cycles, one instruction per wave per ~5 cycles), not by LDS or memory,
// so everything around the 8 v_pk_fma_f32 + 5 ds_read_b128 per sample is written for instruction count: the item's
// coordinates live in scalar registers, the window copy is a scalar row loop around ONE global_load_lds per 8 pixels
// with loop-invariant lane offsets, bounding boxes are reduced with two DPP steps + LDS atomics, and padding masks are
// handled per image (an image without padded tokens -- every bs=1 inference -- never looks at the mask again).
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }


// NBUF = 2: the double-buffered pipeline described above (2 workgroups per CU).  NBUF = 1 (variant 13): the same lean
// code single-buffered -- geometry -> Y -> pack + copy + records -> wait -> X -> gather -- with 4 workgroups per CU, so that
// the phases of DIFFERENT workgroups overlap (a gathering workgroup keeps the LDS pipe busy while its neighbours run
// scalar packing code or wait for their window copy); only the loc / attn rows of the next item are prefetched.
// EPOCH = work items a workgroup decodes at once into its LDS table.
template <bool FUSED, int TH, int TW, int WINPX, int WPS, bool PROF = false, int NBUF = 2, int EPOCH = 64>
__global__ __launch_bounds__(TH * TW * 8, WPS) void msda_fwd_winp_f32(
    const float* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
    const float* __restrict__ loc, const float* __restrict__ attn, float* __restrict__ out, int B, int Lq, int S,
    const float* __restrict__ ref, int ld_off, int ld_logit, const unsigned* __restrict__ keep_bits,
    unsigned long long* __restrict__ prof) {
  // PROF: wave 0 of every workgroup accumulates shader-clock ticks into prof[0..7]: barrier X, geometry, barrier Y,
  // pack + copy issue + records, gather, DMA wait, whole workgroup, items
  unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tk = 0, tstart = 0;
  auto tick = [&](int slot) {
    if (PROF) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      pt[slot] += now - tk;
      tk = now;
    }
  };
  if (PROF) tstart = tk = __builtin_amdgcn_s_memtime();
  // L = 4 levels x P = 4 points only (the launcher routes every other shape to the wave-per-query kernel)
  constexpr int TQ = TH * TW;
  constexpr int NT = TQ * 8;
  constexpr int NW = TQ / 8;    // waves per workgroup
  constexpr int RS = TQ + 1;    // record stride per sample (+1: conflict-free record writes)
  constexpr int WPXB = kZeroPx + WINPX;  // pixels per window buffer
  static_assert(TW == 8, "a wave gathers one row of 8 x-adjacent queries");
  static_assert(WPXB * 128 < 65536, "LDS addresses are packed into 16 bits");
  __shared__ __attribute__((aligned(16))) float4 s_win[NBUF][WPXB * 8];
  __shared__ __attribute__((aligned(16))) float4 s_w[NBUF][16 * RS];  // [buffer][sample][query] weights
  __shared__ __attribute__((aligned(16))) uint4 s_a[NBUF][4 * RS];    // [buffer][level][query][point] addresses
  __shared__ __attribute__((aligned(16))) int4 s_bbox[2][4];       // [parity][level]{ymin, ymax, xmin, xmax}
  __shared__ __attribute__((aligned(16))) int4 s_item[EPOCH][2];  // {b, head, qbase, wq}, {wlim, hlim, -, -}
  __shared__ unsigned s_padded;                                     // bit b: image b has padded tokens

  const int tid = threadIdx.x, c4 = tid & 7;
  const int lane = tid & 63, col = lane >> 3;   // geometry / copy phases: lane = (query column, channel quad)
  const int wave = rfl(tid >> 6);               // = query row of the tile
  const int ql = wave * 8 + col;
  int gcol, gc;                                 // gather phase: lane = (query column, channel quad), see header
  {
    const int l5 = lane & 31;
    if (l5 < 4) { gcol = 0; gc = l5; }
    else if (l5 < 12) { gcol = 2; gc = l5 - 4; }
    else if (l5 < 16) { gcol = 0; gc = l5 - 8; }
    else if (l5 < 20) { gcol = 3; gc = l5 - 16; }
    else if (l5 < 28) { gcol = 1; gc = l5 - 20; }
    else { gcol = 3; gc = l5 - 24; }
    gcol += (lane >> 5) * 4;
  }
  const int gq = wave * 8 + gcol;
  LevelGeom G;
  load_geom(shapes, lsi, 4, G);
  const int nwords = (S + 31) >> 5;
  const int kstride = gridDim.x >> 3;
  const int lvl = c4 >> 1;        // level of this thread's two samples (2*c4, 2*c4+1; P = 4)
  const int H = SEL_H(G, lvl), W = SEL_W(G, lvl), st = SEL_S(G, lvl);
  const float fW = (float)W, fH = (float)H;
  const bool masked = FUSED && keep_bits != nullptr;

  // ---- work-item table: lane t decodes this workgroup's t-th item of the epoch (vector ALU, once) -----------------
  int nwork;
  {
    const TileMap tm = make_tile_map<TH, TW>(G, 4, Lq);
    const int xcd = blockIdx.x & 7;
    const int n0 = tm.grid2d ? tm.nt0 : tm.ntiles, n1 = tm.grid2d ? tm.nt1 : 0, n2 = tm.grid2d ? tm.nt2 : 0,
              n3 = tm.grid2d ? tm.nt3 : 0;
    const int lo0 = (xcd * n0) >> 3, lo1 = (xcd * n1) >> 3, lo2 = (xcd * n2) >> 3, lo3 = (xcd * n3) >> 3;
    const int c0 = (((xcd + 1) * n0) >> 3) - lo0, c1 = (((xcd + 1) * n1) >> 3) - lo1,
              c2 = (((xcd + 1) * n2) >> 3) - lo2, c3 = (((xcd + 1) * n3) >> 3) - lo3;
    const int per_img = (c0 + c1 + c2 + c3) * 8;
    nwork = B * per_img;
    if ((int)(blockIdx.x >> 3) >= nwork) return;
    // work items are decoded EPOCH at a time (one epoch unless the batch is large) into the LDS table
    auto fill_table = [&](int kfirst) {
      if (tid < EPOCH) {
        const int k = kfirst + tid * kstride;
        int4 a = make_int4(0, 0, 0, 8), b4 = make_int4(0, 0, 0, 0);
        if (k < nwork) {
          const int b = k / per_img;
          const int r = k - b * per_img;
          const int head = r & 7;
          int t = r >> 3;
          if (!tm.grid2d) {
            const int qb = (lo0 + t) * TQ;
            a = make_int4(b, head, qb, 8);
            b4 = make_int4(Lq - qb, TH, 1, 0);
          } else {
            int lt;
            if (t < c3) { lt = 3; t += lo3; }
            else if ((t -= c3) < c2) { lt = 2; t += lo2; }
            else if ((t -= c2) < c1) { lt = 1; t += lo1; }
            else { lt = 0; t = t - c1 + lo0; }
            const int tw = sel4(tm.tw0, tm.tw1, tm.tw2, tm.tw3, lt);
            const int Ht = SEL_H(G, lt), Wt = SEL_W(G, lt), st_t = SEL_S(G, lt);
            const int ty = t / tw, tx = t - ty * tw;
            a = make_int4(b, head, st_t + ty * TH * Wt + tx * 8, Wt);
            b4 = make_int4(Wt - tx * 8, Ht - ty * TH, 0, 0);
          }
        }
        s_item[tid][0] = a;
        s_item[tid][1] = b4;
      }
    };

    // ---- helpers (all inlined) -----------------------------------------------------------------------------------
    struct Item { int b, head, qbase, wq, wlim, hlim, linear; };
    auto get_item = [&](int slot) -> Item {
      const int4 a = s_item[slot][0];
      const int4 c = s_item[slot][1];
      Item it;
      it.b = rfl(a.x); it.head = rfl(a.y); it.qbase = rfl(a.z); it.wq = rfl(a.w);
      it.wlim = rfl(c.x); it.hlim = rfl(c.y); it.linear = rfl(c.z);
      return it;
    };
    // query index of tile slot (row = wave, column cl); -1 = padding slot.  Linear mode: wq = 8, hlim = TH.
    auto slot_query = [&](const Item& it, int cl) -> int {
      const int o = wave * it.wq + cl;
      const bool ok = it.linear ? (o < it.wlim) : (wave < it.hlim && cl < it.wlim);
      return ok ? it.qbase + o : -1;
    };
    auto load_la = [&](const Item& it, bool live, float4& lc, float2& aw, float2& rp, int& qv) {
      lc = make_float4(9.f, 9.f, 9.f, 9.f);  // far outside -> invalid
      aw = make_float2(0.f, 0.f);
      rp = make_float2(0.f, 0.f);
      qv = live ? slot_query(it, col) : -1;
      if (qv >= 0) {
        // wave-uniform 64-bit bases + 32-bit lane offsets (the launcher guarantees Lq * row pitch < 2^31 bytes)
        if (FUSED) {
          const char* lb = reinterpret_cast<const char*>(loc + (size_t)it.b * Lq * ld_off + it.head * 32);
          const char* ab_ = reinterpret_cast<const char*>(attn + (size_t)it.b * Lq * ld_logit + it.head * 16);
          const char* rb = reinterpret_cast<const char*>(ref + (size_t)it.b * Lq * 8);
          lc = *reinterpret_cast<const float4*>(lb + (unsigned)(qv * (ld_off * 4) + c4 * 16));
          aw = *reinterpret_cast<const float2*>(ab_ + (unsigned)(qv * (ld_logit * 4) + c4 * 8));
          rp = *reinterpret_cast<const float2*>(rb + (unsigned)(qv * 32 + lvl * 8));
        } else {
          const char* lb = reinterpret_cast<const char*>(loc + ((size_t)it.b * Lq * 8 + it.head) * 32);
          const char* ab_ = reinterpret_cast<const char*>(attn + ((size_t)it.b * Lq * 8 + it.head) * 16);
          lc = *reinterpret_cast<const float4*>(lb + (unsigned)(qv * 1024 + c4 * 16));
          aw = *reinterpret_cast<const float2*>(ab_ + (unsigned)(qv * 512 + c4 * 8));
        }
      }
    };

    int y0[2], x0[2];
    float wq[2][4];
    bool val[2];
    // geometry of this thread's two samples (cuh:38-78 / 268-288) + bounding-box atomics into s_bbox[par]
    auto geom_bbox = [&](float4 lc, float2 aw, float2 rp, int q, int par) {
      if (FUSED) {
        if (q >= 0) lc = make_float4(rp.x + lc.x / fW, rp.y + lc.y / fH, rp.x + lc.z / fW, rp.y + lc.w / fH);
        float m = fmaxf(aw.x, aw.y);
        m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0xB1, 0xf, 0xf, false)));
        m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x4E, 0xf, 0xf, false)));
        m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x141, 0xf, 0xf, false)));
        const float e0 = expf(aw.x - m), e1 = expf(aw.y - m);
        float sum = e0 + e1;
        sum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum), 0xB1, 0xf, 0xf, false));
        sum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum), 0x4E, 0xf, 0xf, false));
        sum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum), 0x141, 0xf, 0xf, false));
        aw = (q >= 0) ? make_float2(e0 / sum, e1 / sum) : make_float2(0.f, 0.f);
      }
      int ymin = INT_MAX, ymax = INT_MIN, xmin = INT_MAX, xmax = INT_MIN;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float x = (j ? lc.z : lc.x) * fW - 0.5f, y = (j ? lc.w : lc.y) * fH - 0.5f;
        val[j] = (y > -1.f) && (x > -1.f) && (y < fH) && (x < fW);
        // invalid samples (outside, NaN): a harmless in-range position with zero attention -> all weights 0, finite
        x = val[j] ? x : 0.f;
        y = val[j] ? y : 0.f;
        const float a = val[j] ? (j ? aw.y : aw.x) : 0.f;
        const float yf = floorf(y), xf = floorf(x);
        const float lh = y - yf, lw = x - xf, hh = 1.f - lh, hw = 1.f - lw;
        y0[j] = (int)yf;
        x0[j] = (int)xf;
        wq[j][0] = hh * hw * a;
        wq[j][1] = hh * lw * a;
        wq[j][2] = lh * hw * a;
        wq[j][3] = lh * lw * a;
        ymin = min(ymin, val[j] ? y0[j] : INT_MAX);
        ymax = max(ymax, val[j] ? y0[j] + 1 : INT_MIN);
        xmin = min(xmin, val[j] ? x0[j] : INT_MAX);
        xmax = max(xmax, val[j] ? x0[j] + 1 : INT_MIN);
      }
      // the 8 queries of the wave sit in lanes l, l^8, l^16, l^32: fold l^8 with DPP, the rest with LDS atomics
      ymin = dpp_xor8_min(ymin);
      ymax = dpp_xor8_max(ymax);
      xmin = dpp_xor8_min(xmin);
      xmax = dpp_xor8_max(xmax);
      if ((lane & 8) == 0 && ymin <= ymax) {
        int* bb = reinterpret_cast<int*>(&s_bbox[par][lvl]);
        atomicMin(bb + 0, ymin);
        atomicMax(bb + 1, ymax);
        atomicMin(bb + 2, xmin);
        atomicMax(bb + 3, xmax);
      }
    };

    // pack windows of the item into buffer `buf`, issue the window copy, write the records; returns the staged mask
    auto pack_stage = [&](const Item& it, int buf, int par) -> unsigned {
      const char* vbase = reinterpret_cast<const char*>(value) + (size_t)it.b * S * 1024 + it.head * 128;
      int wy0[4], wx0[4], ww[4], wh[4], base[4];
      unsigned staged = 0;
      // an image with padded tokens is served from global memory with per-corner mask checks (all levels unstaged)
      const bool allow = !masked || !((s_padded >> min(it.b, 31)) & 1u);
      {
        int off = 0;
#pragma unroll
        for (int l = 3; l >= 0; --l) {
          const int4 bb = s_bbox[par][l];
          wy0[l] = rfl(bb.x);
          const int wy1 = rfl(bb.y);
          wx0[l] = rfl(bb.z);
          const int wx1 = rfl(bb.w);
          const bool empty = wy0[l] > wy1;
          ww[l] = empty ? 0 : (wx1 - wx0[l] + 1);
          wh[l] = empty ? 0 : (wy1 - wy0[l] + 1);
          base[l] = off;
          if (!empty && allow && off + ww[l] * wh[l] <= WINPX) {
            staged |= 1u << l;
            off += ww[l] * wh[l];
          } else {
            wh[l] = 0;
          }
        }
      }
      float4* winb = s_win[buf];
      {
        // window copy: wave w copies level 3 - (w & 3) (rows dealt over the NW / 4 waves that share the level), so the
        // scalar set-up below runs once per wave instead of once per level per wave
        const int l = 3 - (wave & 3), rsub = wave >> 2;
        constexpr int NSUB = NW / 4 > 0 ? NW / 4 : 1;
        const int wwl = sel4(ww[0], ww[1], ww[2], ww[3], l), whl = sel4(wh[0], wh[1], wh[2], wh[3], l);
        if (whl != 0) {
          const int wy0l = sel4(wy0[0], wy0[1], wy0[2], wy0[3], l), wx0l = sel4(wx0[0], wx0[1], wx0[2], wx0[3], l);
          const int basel = sel4(base[0], base[1], base[2], base[3], l);
          const int Hl = sel4(G.H0, G.H1, G.H2, G.H3, l), Wl = sel4(G.W0, G.W1, G.W2, G.W3, l);
          const int sl = sel4(G.s0, G.s1, G.s2, G.s3, l);
          // rows of the window inside the level: [r_lo, r_hi)
          const int r_lo = max(0, -wy0l), r_hi = min(whl, Hl - wy0l);
          for (int cb = 0; cb < wwl; cb += 8) {
            const int cc = cb + col, xx = wx0l + cc;
            if (cc < wwl && (unsigned)xx < (unsigned)Wl) {
              const unsigned voff = (unsigned)(xx * 1024 + c4 * 16);
              const int r0 = r_lo + rsub;
              const char* grow = vbase + (size_t)(sl + (wy0l + r0) * Wl) * 1024;
              int lds = (kZeroPx + basel + r0 * wwl + cb) * 8;
              for (int rr = r0; rr < r_hi; rr += NSUB) {
#if EGTR_WIN_ABLATE != 1
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(grow + voff),
                                                 (__attribute__((address_space(3))) void*)(winb + lds), 16, 0, 0);
#endif
                grow += (size_t)NSUB * Wl * 1024;
                lds += NSUB * wwl * 8;
              }
            }
          }
          // zero apron: window pixels outside the level (only tiles whose samples straddle an image border)
          if (rsub == 0 && (wy0l < 0 || wy0l + whl > Hl || wx0l < 0 || wx0l + wwl > Wl)) {
            const int npx = wwl * whl;
            const float inv = __frcp_rn((float)wwl);
            for (int p = col; p < npx; p += 8) {
              const int rr = (int)(((float)p + 0.5f) * inv);  // exact for p, ww < 4096
              const int cc = p - rr * wwl;
              if ((unsigned)(wy0l + rr) >= (unsigned)Hl || (unsigned)(wx0l + cc) >= (unsigned)Wl)
                winb[(kZeroPx + basel + p) * 8 + c4] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
          }
        }
      }
      {
        const bool st_l = (staged >> lvl) & 1u;
        unsigned code[2];
        if (st_l) {
          // a00 = (kZeroPx + base + (y0 - wy0) * ww + (x0 - wx0)) * 128 = K_l + (y0 * ww + x0) * 128
          const int bww = sel4(ww[0], ww[1], ww[2], ww[3], lvl);
          const int K = sel4((kZeroPx + base[0] - wy0[0] * ww[0] - wx0[0]) * 128,
                             (kZeroPx + base[1] - wy0[1] * ww[1] - wx0[1]) * 128,
                             (kZeroPx + base[2] - wy0[2] * ww[2] - wx0[2]) * 128,
                             (kZeroPx + base[3] - wy0[3] * ww[3] - wx0[3]) * 128, lvl);
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int a00 = K + (y0[j] * bww + x0[j]) * 128;
            code[j] = val[j] ? (unsigned)(a00 | ((a00 + bww * 128) << 16)) : 0u;
          }
        } else {
          // global records: clamped top-left pixel << 2 | dx << 1 | dy; out-of-range corners and padded tokens are
          // folded into the weights
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int ya = max(y0[j], 0), yb = min(y0[j] + 1, H - 1), xa = max(x0[j], 0), xb = min(x0[j] + 1, W - 1);
            const int p00 = st + ya * W + xa;
            const int dx = xb - xa, dy = yb - ya;
            bool k0 = y0[j] >= 0 && x0[j] >= 0, k1 = y0[j] >= 0 && x0[j] + 1 <= W - 1,
                 k2 = y0[j] + 1 <= H - 1 && x0[j] >= 0, k3 = y0[j] + 1 <= H - 1 && x0[j] + 1 <= W - 1;
            if (masked && val[j]) {
              const int p01 = p00 + dx, p10 = p00 + dy * W, p11 = p10 + dx;
              const unsigned* kb = keep_bits + (size_t)it.b * nwords;
              k0 = k0 && ((kb[p00 >> 5] >> (p00 & 31)) & 1u);
              k1 = k1 && ((kb[p01 >> 5] >> (p01 & 31)) & 1u);
              k2 = k2 && ((kb[p10 >> 5] >> (p10 & 31)) & 1u);
              k3 = k3 && ((kb[p11 >> 5] >> (p11 & 31)) & 1u);
            }
            wq[j][0] = k0 ? wq[j][0] : 0.f;
            wq[j][1] = k1 ? wq[j][1] : 0.f;
            wq[j][2] = k2 ? wq[j][2] : 0.f;
            wq[j][3] = k3 ? wq[j][3] : 0.f;
            code[j] = val[j] ? (unsigned)((p00 << 2) | (dx << 1) | dy) : 0u;
          }
        }
        s_w[buf][(2 * c4) * RS + ql] = make_float4(wq[0][0], wq[0][1], wq[0][2], wq[0][3]);
        s_w[buf][(2 * c4 + 1) * RS + ql] = make_float4(wq[1][0], wq[1][1], wq[1][2], wq[1][3]);
        reinterpret_cast<uint2*>(s_a[buf] + lvl * RS + ql)[c4 & 1] = make_uint2(code[0], code[1]);
      }
      return staged;
    };

    auto gather = [&](const Item& it, int buf, unsigned staged, bool wait_dma) {
      const int qo = slot_query(it, gcol);
      // explicit address spaces: the two branches of the generic path must not be merged into flat loads
      typedef const __attribute__((address_space(3))) char* lds_cp;
      typedef const __attribute__((address_space(1))) char* glb_cp;
      const lds_cp win = (lds_cp)(s_win[buf]);
      const float4* wb = s_w[buf] + gq;
      const uint4* ab = s_a[buf] + gq;
      const unsigned coff = gc * 16;
      const glb_cp glb = (glb_cp)(reinterpret_cast<const char*>(value) + (size_t)it.b * S * 1024 + it.head * 128 + gc * 16);
      f32x2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
#if EGTR_WIN_ABLATE == 2
      staged = 0x10u;  // timing ablation: no gather at all (wrong results)
#endif
      if (staged == 0xFu && WPS <= 2) {
        // every window is in LDS (the regular case).  Software pipeline over the levels: the 20 ds_read_b128 of
        // level l+1 are in flight while the 32 v_pk_fma_f32 of level l issue (two register buffers).
        uint4 A[4];
#pragma unroll
        for (int l = 0; l < 4; ++l) A[l] = ab[l * RS];
        float4 wv[2][4], vv[2][4][4];
        auto issue = [&](int l, int bf) {
          const unsigned cd[4] = {A[l].x, A[l].y, A[l].z, A[l].w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            wv[bf][j] = wb[(l * 4 + j) * RS];
            const unsigned a0 = (cd[j] & 0xffffu) | coff, a1 = (cd[j] >> 16) | coff;
            vv[bf][j][0] = lds_ld4(win + a0);
            vv[bf][j][1] = lds_ld4(win + a0 + 128);
            vv[bf][j][2] = lds_ld4(win + a1);
            vv[bf][j][3] = lds_ld4(win + a1 + 128);
          }
        };
        issue(0, 0);
#pragma unroll
        for (int l = 0; l < 4; ++l) {
          if (l < 3) issue(l + 1, (l + 1) & 1);
          const int bf = l & 1;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc0 = pk_fma(f32x2{vv[bf][j][0].x, vv[bf][j][0].y}, wv[bf][j].x, acc0);
            acc1 = pk_fma(f32x2{vv[bf][j][0].z, vv[bf][j][0].w}, wv[bf][j].x, acc1);
            acc0 = pk_fma(f32x2{vv[bf][j][1].x, vv[bf][j][1].y}, wv[bf][j].y, acc0);
            acc1 = pk_fma(f32x2{vv[bf][j][1].z, vv[bf][j][1].w}, wv[bf][j].y, acc1);
            acc0 = pk_fma(f32x2{vv[bf][j][2].x, vv[bf][j][2].y}, wv[bf][j].z, acc0);
            acc1 = pk_fma(f32x2{vv[bf][j][2].z, vv[bf][j][2].w}, wv[bf][j].z, acc1);
            acc0 = pk_fma(f32x2{vv[bf][j][3].x, vv[bf][j][3].y}, wv[bf][j].w, acc0);
            acc1 = pk_fma(f32x2{vv[bf][j][3].z, vv[bf][j][3].w}, wv[bf][j].w, acc1);
          }
        }
      } else if (staged == 0xFu) {
        // every window is in LDS, 4 waves per SIMD (<= 128 registers): two samples (10 ds_read_b128) at a time; the
        // other waves of the SIMD cover the LDS latency
#pragma unroll
        for (int l = 0; l < 4; ++l) {
          const uint4 A = ab[l * RS];
          const unsigned cd[4] = {A.x, A.y, A.z, A.w};
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            float4 w[2], v[2][4];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              w[j] = wb[(l * 4 + 2 * h + j) * RS];
              const unsigned a0 = (cd[2 * h + j] & 0xffffu) | coff, a1 = (cd[2 * h + j] >> 16) | coff;
              v[j][0] = lds_ld4(win + a0);
              v[j][1] = lds_ld4(win + a0 + 128);
              v[j][2] = lds_ld4(win + a1);
              v[j][3] = lds_ld4(win + a1 + 128);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              acc0 = pk_fma(f32x2{v[j][0].x, v[j][0].y}, w[j].x, acc0);
              acc1 = pk_fma(f32x2{v[j][0].z, v[j][0].w}, w[j].x, acc1);
              acc0 = pk_fma(f32x2{v[j][1].x, v[j][1].y}, w[j].y, acc0);
              acc1 = pk_fma(f32x2{v[j][1].z, v[j][1].w}, w[j].y, acc1);
              acc0 = pk_fma(f32x2{v[j][2].x, v[j][2].y}, w[j].z, acc0);
              acc1 = pk_fma(f32x2{v[j][2].z, v[j][2].w}, w[j].z, acc1);
              acc0 = pk_fma(f32x2{v[j][3].x, v[j][3].y}, w[j].w, acc0);
              acc1 = pk_fma(f32x2{v[j][3].z, v[j][3].w}, w[j].w, acc1);
            }
          }
        }
      } else {
#pragma unroll 1
        for (int l = 0; l < ((staged & 0x10u) ? 0 : 4); ++l) {
          const uint4 A = ab[l * RS];
          const unsigned cd[4] = {A.x, A.y, A.z, A.w};
          const bool in_lds = (staged >> l) & 1u;
          const unsigned Wb = (unsigned)sel4(G.W0, G.W1, G.W2, G.W3, l) * 1024u;
#pragma unroll 1
          for (int h = 0; h < 2; ++h) {
            float4 w[2], v[2][4];
            if (in_lds) {
#pragma unroll
              for (int j = 0; j < 2; ++j) {
                const unsigned c = h ? (j ? cd[3] : cd[2]) : (j ? cd[1] : cd[0]);
                w[j] = wb[(l * 4 + 2 * h + j) * RS];
                const unsigned a0 = (c & 0xffffu) | coff, a1 = (c >> 16) | coff;
                v[j][0] = lds_ld4(win + a0);
                v[j][1] = lds_ld4(win + a0 + 128);
                v[j][2] = lds_ld4(win + a1);
                v[j][3] = lds_ld4(win + a1 + 128);
              }
            } else {
#pragma unroll
              for (int j = 0; j < 2; ++j) {
                const unsigned c = h ? (j ? cd[3] : cd[2]) : (j ? cd[1] : cd[0]);
                w[j] = wb[(l * 4 + 2 * h + j) * RS];
                const unsigned o00 = (c >> 2) << 10;
                const unsigned dxb = (c & 2u) ? 1024u : 0u, dyb = (c & 1u) ? Wb : 0u;
                v[j][0] = glb_ld4(glb + o00);
                v[j][1] = glb_ld4(glb + o00 + dxb);
                v[j][2] = glb_ld4(glb + o00 + dyb);
                v[j][3] = glb_ld4(glb + o00 + dyb + dxb);
              }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              acc0 = pk_fma(f32x2{v[j][0].x, v[j][0].y}, w[j].x, acc0);
              acc1 = pk_fma(f32x2{v[j][0].z, v[j][0].w}, w[j].x, acc1);
              acc0 = pk_fma(f32x2{v[j][1].x, v[j][1].y}, w[j].y, acc0);
              acc1 = pk_fma(f32x2{v[j][1].z, v[j][1].w}, w[j].y, acc1);
              acc0 = pk_fma(f32x2{v[j][2].x, v[j][2].y}, w[j].z, acc0);
              acc1 = pk_fma(f32x2{v[j][2].z, v[j][2].w}, w[j].z, acc1);
              acc0 = pk_fma(f32x2{v[j][3].x, v[j][3].y}, w[j].w, acc0);
              acc1 = pk_fma(f32x2{v[j][3].z, v[j][3].w}, w[j].w, acc1);
            }
          }
        }
      }
      if (PROF) {
        // make the timer see the FMAs (they depend on the last LDS reads)
        asm volatile("" :: "v"(acc0), "v"(acc1));
        tick(4);
      }
      // the next item's window copy had the whole gather to land; wait for it BEFORE the stores join the queue
      if (wait_dma) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      tick(5);
      if (qo >= 0) {
        char* ob = reinterpret_cast<char*>(out + ((size_t)it.b * Lq * 8 + it.head) * 32);
        *reinterpret_cast<float4*>(ob + (unsigned)(qo * 1024 + gc * 16)) = make_float4(acc0.x, acc0.y, acc1.x, acc1.y);
      }
    };

    // ---- prologue ------------------------------------------------------------------------------------------------
    if (tid < 8) reinterpret_cast<int4*>(s_bbox)[tid] = make_int4(INT_MAX, INT_MIN, INT_MAX, INT_MIN);
    if (tid < kZeroPx * 8) {
      s_win[0][tid] = make_float4(0.f, 0.f, 0.f, 0.f);
      s_win[NBUF - 1][tid] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (tid == 0) s_padded = 0;
    fill_table((int)(blockIdx.x >> 3));
    wg_barrier();
    if (masked) {
      // which images have padded tokens at all (one pass over the bit mask per workgroup; bit 31 = images >= 31)
      for (int b = 0; b < B; ++b) {
        bool pad = false;
        for (int i = tid; i < nwords; i += NT) {
          const unsigned full = (i == nwords - 1 && (S & 31)) ? ((1u << (S & 31)) - 1u) : 0xffffffffu;
          pad = pad || ((keep_bits[(size_t)b * nwords + i] & full) != full);
        }
        if (pad) atomicOr(&s_padded, 1u << min(b, 31));
      }
    }
    // ---- epochs x pipeline -----------------------------------------------------------------------------------------
    for (int kfirst = (int)(blockIdx.x >> 3); kfirst < nwork; kfirst += EPOCH * kstride) {
      if (kfirst != (int)(blockIdx.x >> 3)) {
        wg_barrier();  // every wave has finished the previous epoch (its table and buffers are free)
        fill_table(kfirst);
        if (tid < 8) reinterpret_cast<int4*>(s_bbox)[tid] = make_int4(INT_MAX, INT_MIN, INT_MAX, INT_MIN);
        wg_barrier();
      }
      const int nit = min(EPOCH, (nwork - kfirst + kstride - 1) / kstride);  // items of this epoch
      Item cur = get_item(0);
      float4 lc;
      float2 aw, rp;
      int qv;
      load_la(cur, true, lc, aw, rp, qv);
      if (NBUF == 1) {
        for (int j = 0; j < nit; ++j) {
          if (PROF) tk = __builtin_amdgcn_s_memtime();
          geom_bbox(lc, aw, rp, qv, j & 1);
          Item nxt = cur;
          if (j + 1 < nit) nxt = get_item(j + 1);
          load_la(nxt, j + 1 < nit, lc, aw, rp, qv);
          tick(1);
          wg_barrier();  // Y: bounding boxes complete; every wave has finished the previous item's gather
          tick(2);
          const unsigned staged = pack_stage(cur, 0, j & 1);
          tick(3);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          tick(5);
          wg_barrier();  // X: window landed, records visible
          tick(0);
          if (tid < 4) s_bbox[j & 1][tid] = make_int4(INT_MAX, INT_MIN, INT_MAX, INT_MIN);  // next used by item j+2
          gather(cur, 0, staged, false);
          if (PROF) pt[7] += 1;
          cur = nxt;
        }
        continue;
      }
      geom_bbox(lc, aw, rp, qv, 0);
      Item nxt = cur;
      if (nit > 1) nxt = get_item(1);
      load_la(nxt, nit > 1, lc, aw, rp, qv);
      wg_barrier();
      unsigned staged_cur = pack_stage(cur, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      for (int j = 0;; ++j) {
        if (PROF) tk = __builtin_amdgcn_s_memtime();
        wg_barrier();  // X: DMA(j) landed (each wave waited for its own pieces), records(j) visible
        tick(0);
        if (tid < 4) s_bbox[j & 1][tid] = make_int4(INT_MAX, INT_MIN, INT_MAX, INT_MIN);  // next used by item j+2
        const bool live_next = j + 1 < nit;
        const Item it_next = nxt;
        unsigned staged_next = 0;
        if (live_next) {
          geom_bbox(lc, aw, rp, qv, (j + 1) & 1);
          const bool more = j + 2 < nit;
          if (more) nxt = get_item(j + 2);
          load_la(nxt, more, lc, aw, rp, qv);
          tick(1);
          wg_barrier();  // Y
          tick(2);
          staged_next = pack_stage(it_next, (NBUF - 1) & (j + 1), (j + 1) & 1);
          tick(3);
        }
        gather(cur, (NBUF - 1) & j, staged_cur, live_next);
        if (PROF) pt[7] += 1;
        if (!live_next) break;
        cur = it_next;
        staged_cur = staged_next;
      }
    }
    if (PROF && tid == 0) {
      pt[6] = __builtin_amdgcn_s_memtime() - tstart;
      for (int i = 0; i < 8; ++i) atomicAdd(prof + i, pt[i]);
    }
  }
}

int pick_grid(int B, int S, int tq) {
  // upper estimate of the number of (tile, head) items (the level shapes live in device memory); the kernel strides
  // over the real count, surplus workgroups exit at once
  long long items = (long long)B * 8 * ((S + tq - 1) / tq);
  items += items / 3 + 64;
  if (items > 65536) items = 65536;
  return (int)((items + 7) & ~7ll);
}

}  // namespace

// kind 0: 4 x 8 query tiles (256 threads, 4 workgroups per CU); kind 1: 8 x 8 tiles (512 threads, 2 per CU);
// kind 2: 4 x 8 tiles with a larger window budget (3 workgroups per CU).
int egtr_launch_msda_fwd_win_f32(hipStream_t st, const float* value, const int64_t* shapes, const int64_t* lsi,
                                 const float* loc, const float* attn, float* out, int B, int Lq, int S, int L, int P,
                                 int kind, const float* ref, float* attn_out, int ld_off, int ld_logit,
                                 const unsigned char* keep, const unsigned* keep_bits, unsigned long long* prof) {
  const bool fused = ref != nullptr;
#define EGTR_WIN_LAUNCH(F, TH_, WINPX_, WPS_)                                                                      \
  hipLaunchKernelGGL((msda_fwd_win_f32<F, TH_, 8, WINPX_, WPS_>), dim3(pick_grid(B, S, TH_ * 8)), dim3(TH_ * 64), 0, \
                     st, value, shapes, lsi, loc, attn, out, B, Lq, S, L, P, ref, attn_out, ld_off, ld_logit, keep,  \
                     keep_bits, prof)
  if (kind == 3 || kind == 4 || kind == 5) {
    // persistent: exactly the resident workgroups (2 per CU at 80 KB LDS / 1 per CU at 160 KB)
    hipDeviceProp_t prop;
    int dev = 0;
    static int ncu = 0;
    if (ncu == 0) {
      if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return EGTR_E_LAUNCH;
      ncu = prop.multiProcessorCount;
    }
#define EGTR_WINP_LAUNCH(F, TH_, WINPX_, WPS_, PERCU)                                                                \
  hipLaunchKernelGGL((msda_fwd_winp_f32<F, TH_, 8, WINPX_, WPS_>), dim3((ncu * PERCU) & ~7), dim3(TH_ * 64), 0, st,   \
                     value, shapes, lsi, loc, attn, out, B, Lq, S, ref, ld_off, ld_logit, keep_bits,                \
                     (unsigned long long*)nullptr)
    if (L != 4 || P != 4 || attn_out != nullptr || (keep != nullptr && keep_bits == nullptr)) return EGTR_E_UNSUPPORTED;
    {  // 32-bit lane offsets inside one image
      const long long pitch = std::max<long long>(1024, 4ll * std::max(ld_off, ld_logit));
      if ((long long)Lq * pitch >= (1ll << 31)) return EGTR_E_UNSUPPORTED;
    }
    if (kind == 3 && prof != nullptr && !fused) {
      hipLaunchKernelGGL((msda_fwd_winp_f32<false, 4, 8, 224, 2, true>), dim3((ncu * 2) & ~7), dim3(256), 0, st, value,
                         shapes, lsi, loc, attn, out, B, Lq, S, ref, ld_off, ld_logit, keep_bits, prof);
      return egtr_check_launch();
    }
    if (kind == 5 && prof != nullptr && !fused) {
      hipLaunchKernelGGL((msda_fwd_winp_f32<false, 4, 8, 224, 4, true, 1, 32>), dim3((ncu * 4) & ~7), dim3(256), 0, st,
                         value, shapes, lsi, loc, attn, out, B, Lq, S, ref, ld_off, ld_logit, keep_bits, prof);
      return egtr_check_launch();
    }
    if (kind == 5) {
      if (fused)
        hipLaunchKernelGGL((msda_fwd_winp_f32<true, 4, 8, 224, 4, false, 1, 32>), dim3((ncu * 4) & ~7), dim3(256), 0,
                           st, value, shapes, lsi, loc, attn, out, B, Lq, S, ref, ld_off, ld_logit, keep_bits,
                           (unsigned long long*)nullptr);
      else
        hipLaunchKernelGGL((msda_fwd_winp_f32<false, 4, 8, 224, 4, false, 1, 32>), dim3((ncu * 4) & ~7), dim3(256), 0,
                           st, value, shapes, lsi, loc, attn, out, B, Lq, S, ref, ld_off, ld_logit, keep_bits,
                           (unsigned long long*)nullptr);
      return egtr_check_launch();
    }
    if (kind == 3) {
      if (fused) EGTR_WINP_LAUNCH(true, 4, 224, 2, 2); else EGTR_WINP_LAUNCH(false, 4, 224, 2, 2);
    } else {
      if (fused) EGTR_WINP_LAUNCH(true, 8, 456, 2, 1); else EGTR_WINP_LAUNCH(false, 8, 456, 2, 1);
    }
#undef EGTR_WINP_LAUNCH
    return egtr_check_launch();
  }
  if (kind == 1) {
    if (fused) EGTR_WIN_LAUNCH(true, 8, 464, 4); else EGTR_WIN_LAUNCH(false, 8, 464, 4);
  } else if (kind == 2) {
    if (fused) EGTR_WIN_LAUNCH(true, 4, 328, 3); else EGTR_WIN_LAUNCH(false, 4, 328, 3);
  } else {
    if (fused) EGTR_WIN_LAUNCH(true, 4, 232, 4); else EGTR_WIN_LAUNCH(false, 4, 232, 4);
  }
#undef EGTR_WIN_LAUNCH
  return egtr_check_launch();
}
