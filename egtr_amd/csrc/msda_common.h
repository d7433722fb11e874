// Shared device helpers of the MSDA kernels (msda.hip, msda_tile.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace egtr_msda {

constexpr int kWaves = 4;  // waves (= queries) per workgroup in the wave-per-query kernels

// Level geometry of up to four levels, held in scalar registers (plain members, never indexed dynamically,
// so nothing is spilled to scratch).
struct LevelGeom {
  int H0, H1, H2, H3, W0, W1, W2, W3, s0, s1, s2, s3;
};

// Select one of four wave-uniform values by a per-lane level index.
__device__ __forceinline__ int sel4(int a0, int a1, int a2, int a3, int l) {
  int r = a0;
  r = (l == 1) ? a1 : r;
  r = (l == 2) ? a2 : r;
  r = (l == 3) ? a3 : r;
  return r;
}
#define SEL_H(G, l) sel4(G.H0, G.H1, G.H2, G.H3, l)
#define SEL_W(G, l) sel4(G.W0, G.W1, G.W2, G.W3, l)
#define SEL_S(G, l) sel4(G.s0, G.s1, G.s2, G.s3, l)

// XCD-aware bijective remap: block b runs on XCD b%8 (observed dispatch order, used for speed only);
// give XCD x the x-th contiguous chunk of logical work items.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

struct SampleGeom {
  int off[4];    // byte offsets (within one batch image) of the 4 clamped corners, head+level folded in
  float w[4];    // bilinear weights hh*hw, hh*lw, lh*hw, lh*lw
  bool ok[4];    // per-corner in-range (and sample valid)
  float lh, lw;
  bool valid;
  int y0, x0;    // unclamped integer top-left corner (0 when the sample is invalid)
};

// Geometry of one sample (reference cuh:38-78 / 268-288).  x,y already scaled: x = loc_x*W - 0.5.
template <int ROW_BYTES /* M*D*sizeof(elt) */, int HEAD_BYTES /* D*sizeof(elt) */>
__device__ __forceinline__ SampleGeom sample_geom(float lx, float ly, int H, int W, int start, int head) {
  SampleGeom g;
  const float x = lx * (float)W - 0.5f;
  const float y = ly * (float)H - 0.5f;
  g.valid = (y > -1.f) && (x > -1.f) && (y < (float)H) && (x < (float)W);
  const float yf = floorf(y), xf = floorf(x);
  g.lh = y - yf;
  g.lw = x - xf;
  const float hh = 1.f - g.lh, hw = 1.f - g.lw;
  // NaN / huge coordinates: valid == false, so every weight is zeroed; clamp keeps addresses in range.
  int y0 = g.valid ? (int)yf : 0, x0 = g.valid ? (int)xf : 0;
  const int y1 = y0 + 1, x1 = x0 + 1;
  g.y0 = y0;
  g.x0 = x0;
  const bool y0ok = y0 >= 0, x0ok = x0 >= 0, y1ok = y1 <= H - 1, x1ok = x1 <= W - 1;
  g.ok[0] = g.valid && y0ok && x0ok;
  g.ok[1] = g.valid && y0ok && x1ok;
  g.ok[2] = g.valid && y1ok && x0ok;
  g.ok[3] = g.valid && y1ok && x1ok;
  g.w[0] = hh * hw;
  g.w[1] = hh * g.lw;
  g.w[2] = g.lh * hw;
  g.w[3] = g.lh * g.lw;
  const int y0c = max(y0, 0), x0c = max(x0, 0), y1c = min(y1, H - 1), x1c = min(x1, W - 1);
  const int r0 = (start + y0c * W) * ROW_BYTES + head * HEAD_BYTES;
  const int r1 = (start + y1c * W) * ROW_BYTES + head * HEAD_BYTES;
  g.off[0] = r0 + x0c * ROW_BYTES;
  g.off[1] = r0 + x1c * ROW_BYTES;
  g.off[2] = r1 + x0c * ROW_BYTES;
  g.off[3] = r1 + x1c * ROW_BYTES;
  return g;
}

__device__ __forceinline__ void load_geom(const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
                                          int L, LevelGeom& g) {
  const int l1 = (1 < L) ? 1 : 0, l2 = (2 < L) ? 2 : 0, l3 = (3 < L) ? 3 : 0;
  g.H0 = (int)shapes[0];
  g.W0 = (int)shapes[1];
  g.s0 = (int)lsi[0];
  g.H1 = (int)shapes[2 * l1];
  g.W1 = (int)shapes[2 * l1 + 1];
  g.s1 = (int)lsi[l1];
  g.H2 = (int)shapes[2 * l2];
  g.W2 = (int)shapes[2 * l2 + 1];
  g.s2 = (int)lsi[l2];
  g.H3 = (int)shapes[2 * l3];
  g.W3 = (int)shapes[2 * l3 + 1];
  g.s3 = (int)lsi[l3];
}

// ---- query tiles of the tile x head kernels (msda_tile.hip, msda_lane.hip) ------------------------------------------
// Encoder mode (Lq == sum H_l W_l and the levels are laid out back to back): queries are the pixels of the levels and a
// tile is a TH x TW pixel block of one level.  Otherwise a tile is TH*TW consecutive queries.
struct TileMap {
  bool grid2d;
  int ntiles;               // tiles per batch image
  int nt0, nt1, nt2, nt3;   // tiles per level (2-D mode)
  int tw0, tw1, tw2, tw3;   // tiles per row of each level
};

template <int TH = 8, int TW = 8>
__device__ __forceinline__ TileMap make_tile_map(const LevelGeom& G, int L, int Lq) {
  TileMap m;
  const int s0 = G.H0 * G.W0, s1 = (L > 1) ? G.H1 * G.W1 : 0, s2 = (L > 2) ? G.H2 * G.W2 : 0,
            s3 = (L > 3) ? G.H3 * G.W3 : 0;
  m.grid2d = (s0 + s1 + s2 + s3 == Lq) && (G.s0 == 0) && (L < 2 || G.s1 == s0) && (L < 3 || G.s2 == s0 + s1) &&
             (L < 4 || G.s3 == s0 + s1 + s2);
  m.tw0 = (G.W0 + TW - 1) / TW;
  m.tw1 = (G.W1 + TW - 1) / TW;
  m.tw2 = (G.W2 + TW - 1) / TW;
  m.tw3 = (G.W3 + TW - 1) / TW;
  m.nt0 = m.tw0 * ((G.H0 + TH - 1) / TH);
  m.nt1 = (L > 1) ? m.tw1 * ((G.H1 + TH - 1) / TH) : 0;
  m.nt2 = (L > 2) ? m.tw2 * ((G.H2 + TH - 1) / TH) : 0;
  m.nt3 = (L > 3) ? m.tw3 * ((G.H3 + TH - 1) / TH) : 0;
  m.ntiles = m.grid2d ? (m.nt0 + m.nt1 + m.nt2 + m.nt3) : ((Lq + TH * TW - 1) / (TH * TW));
  return m;
}

// Query index (within the batch image) of slot `ql` (0..63) of tile `tile`; -1 if the slot is padding.
template <int TH = 8, int TW = 8>
__device__ __forceinline__ int tile_query(const TileMap& m, const LevelGeom& G, int tile, int ql, int Lq) {
  if (!m.grid2d) {
    const int q = tile * (TH * TW) + ql;
    return q < Lq ? q : -1;
  }
  int lvl = 0, t = tile;
  if (t >= m.nt0) { t -= m.nt0; lvl = 1;
    if (t >= m.nt1) { t -= m.nt1; lvl = 2;
      if (t >= m.nt2) { t -= m.nt2; lvl = 3; } } }
  const int tw = sel4(m.tw0, m.tw1, m.tw2, m.tw3, lvl);
  const int H = SEL_H(G, lvl), W = SEL_W(G, lvl), st = SEL_S(G, lvl);
  const int ty = t / tw, tx = t - ty * tw;
  const int qy = ty * TH + ql / TW, qx = tx * TW + ql % TW;
  return (qy < H && qx < W) ? st + qy * W + qx : -1;
}

// Tile -> (level, top-left pixel) in encoder mode.
template <int TH, int TW>
__device__ __forceinline__ void tile_origin(const TileMap& m, int tile, int& lvl, int& y0, int& x0) {
  int l = 0, t = tile;
  if (t >= m.nt0) { t -= m.nt0; l = 1;
    if (t >= m.nt1) { t -= m.nt1; l = 2;
      if (t >= m.nt2) { t -= m.nt2; l = 3; } } }
  const int tw = sel4(m.tw0, m.tw1, m.tw2, m.tw3, l);
  const int ty = t / tw;
  lvl = l;
  y0 = ty * TH;
  x0 = (t - ty * tw) * TW;
}


}  // namespace egtr_msda
