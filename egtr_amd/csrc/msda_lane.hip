// MSDA forward, "lane per query" variant for gfx950: one lane owns one (query, head) with all 32 channels, and the
// bilinear corners are gathered from LDS-staged sampling windows.
//
// Why: the wave-per-query kernel (msda.hip) gathers 841 MB per encoder launch through the vector L1 at
// ~80 % of the measured L1 gather ceiling (30.5 TB/s) = 20 % of the HBM roofline at best.  LDS serves ds_read_b128 at
// 256 B/clk/CU (~150 TB/s chip-wide), and neighbouring encoder queries sample neighbouring pixels, so a 16 x 4*NW tile
// of queries of one head needs, per level, only a small window of pixels.  The first LDS kernel (msda_tile.hip, 8 lanes
// per query) paid for that with per-sample records broadcast through LDS (1/3 of its LDS cycles), 2x the VALU work
// (window packing, record addresses) and bank conflicts.  This kernel removes those three costs:
//   * lane = query: every lane computes the geometry of its own 16 samples in registers -- no records, no exchange;
//     the 32 channels of a corner are 8 ds_read_b128 that share one address register;
//   * window layout [channel quad][pixel] (8 planes of 16-byte entries): the quad index is the instruction's immediate
//     offset (k * PLANE), so a corner costs one address and ZERO per-read VALU; the 16 lanes that the hardware
//     services together (MI355X_MICROARCH.md, LDS table: {0-3,12-15,20-27}, ...) are made the 16 queries of one tile
//     row, which read 16 consecutive (or pairwise identical) pixels = 16 distinct 16-byte bank groups: conflict-free
//     for regular offsets, whatever the window pitch;
//   * staging: lane i computes the descriptor of the wave's i-th chunk (<= 8 pixels of one window row) once; the copy
//     loop then costs three v_readlane per 1 KiB.
// Work item = (batch, tile, head); NW waves per workgroup share one window; persistent launch, XCD-aware work order.
// A level whose window does not fit the LDS budget (a level-1 tile sampling level 0, or wildly scattered learned
// offsets) is gathered from global memory instead, so results never depend on the windows -- only the speed does.
// loc / attn rows (128 B / 64 B per (query, head)) are read coalesced (8 / 4 lanes per row) and transposed to one row
// per lane through padded LDS rows; the output goes back the same way, so every global access is a full 128-B line.
// Code size matters: fully unrolled over the 16 samples this kernel was 75 KB (> the 64 KB instruction cache) and ran
// 3x slower than the wave-per-query kernel; the level loops below are rolled (4 samples per body).
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>

#include "common.h"
#include "msda_common.h"

using namespace egtr_msda;

namespace {

template <int CTRL>
__device__ __forceinline__ int dpp_mov(int v) {
  return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false);
}
// min / max over the 16 lanes of a DPP row (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror), then over
// the 4 rows of the wave by readlane + scalar min / max: the result is wave-uniform (SGPR).
__device__ __forceinline__ int wave_min(int v) {
  v = min(v, dpp_mov<0xB1>(v));
  v = min(v, dpp_mov<0x4E>(v));
  v = min(v, dpp_mov<0x141>(v));
  v = min(v, dpp_mov<0x140>(v));
  return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
             min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int wave_max(int v) {
  v = max(v, dpp_mov<0xB1>(v));
  v = max(v, dpp_mov<0x4E>(v));
  v = max(v, dpp_mov<0x141>(v));
  v = max(v, dpp_mov<0x140>(v));
  return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
             max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// component-wise selects by a wave-uniform level index (kept scalar so that the arrays stay in registers)
__device__ __forceinline__ float sel4f(float a0, float a1, float a2, float a3, int l) {
  float r = a0;
  r = (l == 1) ? a1 : r;
  r = (l == 2) ? a2 : r;
  r = (l == 3) ? a3 : r;
  return r;
}
#define EGTR_LEVEL_LOC(LX, LY, l)                                                                     \
  const float LX[4] = {sel4f(lc[0].x, lc[2].x, lc[4].x, lc[6].x, l), sel4f(lc[0].z, lc[2].z, lc[4].z, lc[6].z, l), \
                       sel4f(lc[1].x, lc[3].x, lc[5].x, lc[7].x, l), sel4f(lc[1].z, lc[3].z, lc[5].z, lc[7].z, l)}; \
  const float LY[4] = {sel4f(lc[0].y, lc[2].y, lc[4].y, lc[6].y, l), sel4f(lc[0].w, lc[2].w, lc[4].w, lc[6].w, l), \
                       sel4f(lc[1].y, lc[3].y, lc[5].y, lc[7].y, l), sel4f(lc[1].w, lc[3].w, lc[5].w, lc[7].w, l)};

// Lane -> (tile row within the wave, x within the row) such that the four 16-lane groups a ds_read_b128 is serviced
// in are the four rows: group {0-3,12-15,20-27} -> row 0 (x = 0..15), {4-11,16-19,28-31} -> row 1, same + 2 for lanes 32-63.
__device__ __forceinline__ void lane_slot(int lane, int& row, int& x) {
  const int j = lane & 31;
  int g, p;
  if (j < 4) { g = 0; p = j; }
  else if (j < 12) { g = 1; p = j - 4; }
  else if (j < 16) { g = 0; p = j - 8; }
  else if (j < 20) { g = 1; p = j - 8; }
  else if (j < 28) { g = 0; p = j - 12; }
  else { g = 1; p = j - 16; }
  row = (lane >> 5) * 2 + g;
  x = p;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- the gather pipeline of one staged level (4 samples) ----------------------------------------------------------------
// 32 units per lane; unit U = sample U>>3, corner (U>>1)&3, channel quads 4*(U&1) .. +3: 4 ds_read_b128 that share one
// address register, the quad being the instruction's immediate offset.  Units are issued 3 ahead of the FMAs that
// consume them (12 reads in flight per wave; lgkmcnt counts 15 at most).  The reads are inline asm: hipcc hoists plain
// LDS loads to the top of the block and spills them, and it cannot count asm memory operations, so the waits are
// explicit; the "+v" ties make the FMAs depend on the wait statement, and the empty asm after the FMAs pins them there
// (SelectionDAG orders only chained nodes; without it the FMAs drift hundreds of instructions down and the ring
// values get parked in AGPRs).
template <int OFF>
__device__ __forceinline__ f32x4 lds_read_b128(unsigned addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int N>
__device__ __forceinline__ void lgkm_wait(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}

struct LevelWin {  // wave-uniform description of one level and its staged window
  int H, W;                 // level size
  int wx0, wy0, mw, mh;     // window origin, width - 1, height - 1
  int pitch, base;          // window row pitch and first pixel index in the planes
};

template <int PLANE>
struct LevelGather {
  f32x4 ring[4][4];
  unsigned ca[4];   // LDS byte addresses of the 4 corners of the sample being issued
  float wt[2][4];   // bilinear x attention weights of the sample being consumed / the one being issued

  template <int U>
  __device__ __forceinline__ void issue(const float (&lx)[4], const float (&ly)[4], const float (&at)[4],
                                        const LevelWin& g, unsigned lds0) {
    constexpr int p = U >> 3, j = (U >> 1) & 3, hq = U & 1;
    if constexpr ((U & 7) == 0) {
      const SampleGeom sg = sample_geom<1024, 128>(lx[p], ly[p], g.H, g.W, 0, 0);
#pragma unroll
      for (int k = 0; k < 4; ++k) wt[p & 1][k] = sg.ok[k] ? sg.w[k] * at[p] : 0.f;
      const int xa = min(max(sg.x0 - g.wx0, 0), g.mw), xb = min(max(sg.x0 + 1 - g.wx0, 0), g.mw);
      const int ya = min(max(sg.y0 - g.wy0, 0), g.mh), yb = min(max(sg.y0 + 1 - g.wy0, 0), g.mh);
      const int ra = g.base + ya * g.pitch, rb = g.base + yb * g.pitch;
      ca[0] = lds0 + (unsigned)(ra + xa) * 16u;
      ca[1] = lds0 + (unsigned)(ra + xb) * 16u;
      ca[2] = lds0 + (unsigned)(rb + xa) * 16u;
      ca[3] = lds0 + (unsigned)(rb + xb) * 16u;
    }
    if constexpr (hq == 0 || 7 * PLANE <= 65535) {
      ring[U & 3][0] = lds_read_b128<(hq * 4 + 0) * PLANE>(ca[j]);
      ring[U & 3][1] = lds_read_b128<(hq * 4 + 1) * PLANE>(ca[j]);
      ring[U & 3][2] = lds_read_b128<(hq * 4 + 2) * PLANE>(ca[j]);
      ring[U & 3][3] = lds_read_b128<(hq * 4 + 3) * PLANE>(ca[j]);
    } else {  // the 16-bit offset field does not reach plane 7: rebase on plane 4
      const unsigned a4 = ca[j] + 4u * PLANE;
      ring[U & 3][0] = lds_read_b128<0 * PLANE>(a4);
      ring[U & 3][1] = lds_read_b128<1 * PLANE>(a4);
      ring[U & 3][2] = lds_read_b128<2 * PLANE>(a4);
      ring[U & 3][3] = lds_read_b128<3 * PLANE>(a4);
    }
  }

  template <int U>
  __device__ __forceinline__ void step(f32x4 (&acc)[8], const float (&lx)[4], const float (&ly)[4],
                                       const float (&at)[4], const LevelWin& g, unsigned lds0) {
    if constexpr (U + 3 < 32) issue<U + 3>(lx, ly, at, g, lds0);
    constexpr int ahead = (U + 3 < 32) ? 3 : 31 - U;  // units issued after U that may still be in flight
    f32x4(&r)[4] = ring[U & 3];
    lgkm_wait<4 * ahead>(r[0], r[1], r[2], r[3]);
    const float w = wt[(U >> 3) & 1][(U >> 1) & 3];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      f32x4& a = acc[(U & 1) * 4 + kk];
      a.x = fmaf(w, r[kk].x, a.x);
      a.y = fmaf(w, r[kk].y, a.y);
      a.z = fmaf(w, r[kk].z, a.z);
      a.w = fmaf(w, r[kk].w, a.w);
    }
    asm volatile("" : "+v"(acc[(U & 1) * 4 + 0]), "+v"(acc[(U & 1) * 4 + 1]), "+v"(acc[(U & 1) * 4 + 2]),
                 "+v"(acc[(U & 1) * 4 + 3]));
    if constexpr (U + 1 < 32) step<U + 1>(acc, lx, ly, at, g, lds0);
  }

  __device__ __forceinline__ void run(f32x4 (&acc)[8], const float (&lx)[4], const float (&ly)[4],
                                      const float (&at)[4], const LevelWin& g, unsigned lds0) {
    issue<0>(lx, ly, at, g, lds0);
    issue<1>(lx, ly, at, g, lds0);
    issue<2>(lx, ly, at, g, lds0);
    step<0>(acc, lx, ly, at, g, lds0);
  }
};

// PROF: shader-clock cycles of thread 0 summed over the work items: prof[0] loc/attn + bounding boxes, [1] window
// packing + staging, [2] gather, [3] output, [4] work items, [5] work items with all 4 windows staged, [6] their gather
// cycles, [7] their total cycles.
template <bool PROF, int NW, int WINPX>
__global__ __launch_bounds__(64 * NW) void msda_fwd_lane_f32(
    const float* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
    const float* __restrict__ loc, const float* __restrict__ attn, float* __restrict__ out, int B, int Lq, int S,
    unsigned long long* __restrict__ prof) {
  constexpr int TW = 16, TH = 4 * NW, TQ = 64 * NW;
  constexpr int PLANE = WINPX * 16 + 16;    // bytes per channel-quad plane; +16: planes start 4 banks apart
  constexpr int CPW = 48;                   // staging chunk slots per wave (one descriptor lane each)
  constexpr int BATCH = CPW / 2;            // staging loads in flight per wave
  constexpr int LOCROW = 144, ATTROW = 80;  // padded row strides of the loc / attn / out transposes (conflict-free b128)
  constexpr int SCRATCH = 64 * (LOCROW + ATTROW);  // per wave, aliases the window planes
  static_assert(WINPX % 8 == 0 && NW * SCRATCH <= 8 * PLANE && CPW <= 64 && CPW % 2 == 0, "LDS layout");
  __shared__ __attribute__((aligned(16))) char smem[8 * PLANE];
  __shared__ __attribute__((aligned(16))) int s_bbox[NW][16];  // [wave][level*4 + {ymin, ymax, xmin, xmax}]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uni(tid >> 6);
  LevelGeom G;
  load_geom(shapes, lsi, 4, G);
  const TileMap tm = make_tile_map<TH, TW>(G, 4, Lq);
  const int nwork = B * tm.ntiles * 8;
  int lrow, lx_;
  lane_slot(lane, lrow, lx_);
  const int dy = wave * 4 + lrow;
  char* scr = smem + wave * SCRATCH;  // loc rows / out rows of this wave
  char* scrA = scr + 64 * LOCROW;     // attn rows
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const int sub = lane >> 3, quad = lane & 7;

  for (int work = blockIdx.x; work < nwork; work += gridDim.x) {
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    if (PROF) t0 = __builtin_amdgcn_s_memtime();
    const int wlog = xcd_remap(work, nwork);
    const int head = wlog & 7;
    const int t = wlog >> 3;
    const int b = t / tm.ntiles, tile = t - b * tm.ntiles;
    int q = -1;
    if (tm.grid2d) {
      int tl, ty0, tx0;
      tile_origin<TH, TW>(tm, tile, tl, ty0, tx0);
      const int qy = ty0 + dy, qx = tx0 + lx_;
      if (qy < sel4(G.H0, G.H1, G.H2, G.H3, tl) && qx < sel4(G.W0, G.W1, G.W2, G.W3, tl))
        q = sel4(G.s0, G.s1, G.s2, G.s3, tl) + qy * sel4(G.W0, G.W1, G.W2, G.W3, tl) + qx;
    } else {
      const int qq = tile * TQ + wave * 64 + lane;
      q = qq < Lq ? qq : -1;
    }
    const int qh = (q >= 0) ? (b * Lq + q) * 8 + head : -1;  // (query, head) row index; < 2^30 (checked by the host)
    const char* vb = reinterpret_cast<const char*>(value) + (size_t)b * S * 1024 + head * 128;

    // ---- 1: loc / attn rows -> one row per lane (coalesced loads, padded LDS transpose) -------------------------
    float4 lc[8], aw[4];
    {
      float4 tl4[8], ta4[4];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int qs = __shfl(qh, 8 * i + sub);
        // padding slots read row 0 (branch-free) and are replaced by a far-outside location -> invalid samples
        const float4 v = reinterpret_cast<const float4*>(loc)[(size_t)max(qs, 0) * 8 + quad];
        tl4[i] = (qs >= 0) ? v : make_float4(9.f, 9.f, 9.f, 9.f);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int qs = __shfl(qh, 16 * i + (lane >> 2));
        const float4 v = reinterpret_cast<const float4*>(attn)[(size_t)max(qs, 0) * 4 + (lane & 3)];
        ta4[i] = (qs >= 0) ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
        *reinterpret_cast<float4*>(scr + (8 * i + sub) * LOCROW + quad * 16) = tl4[i];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *reinterpret_cast<float4*>(scrA + (16 * i + (lane >> 2)) * ATTROW + (lane & 3) * 16) = ta4[i];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int k = 0; k < 8; ++k) lc[k] = *reinterpret_cast<const float4*>(scr + lane * LOCROW + k * 16);
#pragma unroll
      for (int k = 0; k < 4; ++k) aw[k] = *reinterpret_cast<const float4*>(scrA + lane * ATTROW + k * 16);
    }

    // ---- 2: per-level bounding box of the valid corners of the tile's samples (rolled over the levels) ---------------
    int bymin[4], bymax[4], bxmin[4], bxmax[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      bymin[l] = bxmin[l] = INT_MAX;
      bymax[l] = bxmax[l] = INT_MIN;
    }
#pragma unroll 1
    for (int l = 0; l < 4; ++l) {
      const int H = SEL_H(G, l), W = SEL_W(G, l);
      EGTR_LEVEL_LOC(lx, ly, l)
      int ymin = INT_MAX, ymax = INT_MIN, xmin = INT_MAX, xmax = INT_MIN;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const SampleGeom g = sample_geom<1024, 128>(lx[p], ly[p], H, W, 0, 0);
        if (g.ok[0] || g.ok[1] || g.ok[2] || g.ok[3]) {
          ymin = min(ymin, max(g.y0, 0));
          ymax = max(ymax, min(g.y0 + 1, H - 1));
          xmin = min(xmin, max(g.x0, 0));
          xmax = max(xmax, min(g.x0 + 1, W - 1));
        }
      }
      const int a = wave_min(ymin), bb = wave_max(ymax), c = wave_min(xmin), d = wave_max(xmax);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (l == k) {
          bymin[k] = a;
          bymax[k] = bb;
          bxmin[k] = c;
          bxmax[k] = d;
        }
      }
    }
    if (NW > 1) {
      // exchange the per-wave boxes through LDS (plain stores, no atomics)
      if (lane == 0) {
#pragma unroll
        for (int l = 0; l < 4; ++l)
          *reinterpret_cast<int4*>(&s_bbox[wave][l * 4]) = make_int4(bymin[l], bymax[l], bxmin[l], bxmax[l]);
      }
      __syncthreads();  // B: boxes of every wave visible; every wave is done with its loc / attn scratch rows
#pragma unroll
      for (int w = 0; w < NW; ++w) {
#pragma unroll
        for (int l = 0; l < 4; ++l) {
          const int4 o = *reinterpret_cast<const int4*>(&s_bbox[w][l * 4]);
          bymin[l] = min(bymin[l], uni(o.x));
          bymax[l] = max(bymax[l], uni(o.y));
          bxmin[l] = min(bxmin[l], uni(o.z));
          bxmax[l] = max(bxmax[l], uni(o.w));
        }
      }
    }
    if (PROF) t1 = __builtin_amdgcn_s_memtime();

    // ---- 3: pack the windows into the LDS budget (wave-uniform, scalar registers) ----------------------------------
    // Staging unit = chunk: up to 8 consecutive pixels of ONE window row (8 lanes per pixel).  A level is staged when
    // its pixels fit the LDS budget and its chunks fit the chunk slots (CPW per wave).
    int wy0[4], wx0[4], ww[4], wh[4], base[4], cpr[4], cbeg[4], cend[4];
    unsigned staged = 0;
    {
      int off = 1, coff = 0;  // pixel 0 is all-zero: the target of the corners of empty levels
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        const int y0 = bymin[l], y1 = bymax[l], x0 = bxmin[l], x1 = bxmax[l];
        const bool empty = y0 > y1;
        wy0[l] = empty ? 0 : y0;
        wx0[l] = empty ? 0 : x0;
        ww[l] = empty ? 1 : x1 - x0 + 1;
        wh[l] = empty ? 1 : y1 - y0 + 1;
        cpr[l] = (ww[l] + 7) >> 3;
        const int n = empty ? 0 : ww[l] * wh[l];
        const int nc = empty ? 0 : wh[l] * cpr[l];
        cbeg[l] = coff;
        if (off + n <= WINPX && coff + nc <= CPW * NW) {
          staged |= 1u << l;
          base[l] = empty ? 0 : off;
          off += n;
          coff += nc;
        } else {
          base[l] = 0;
        }
        cend[l] = coff;
      }
    }

    // ---- 4: stage the windows: coalesced global reads (8 lanes = one 128-B pixel row of this head), written to the
    //         [quad][pixel] planes.  Lane i first computes the descriptor of this wave's i-th chunk {global byte offset,
    //         LDS byte offset, pixels}; the copy loop then needs only three v_readlane per chunk. -----------------------
    {
      int dsrc, ddst, dcnt;
      {
        const int g = lane * NW + wave;
        const int l = (g >= cend[0] ? 1 : 0) + (g >= cend[1] ? 1 : 0) + (g >= cend[2] ? 1 : 0);
        const int cprl = sel4(cpr[0], cpr[1], cpr[2], cpr[3], l);
        const int gl = g - sel4(cbeg[0], cbeg[1], cbeg[2], cbeg[3], l);
        const int r = (int)(((float)gl + 0.5f) * __builtin_amdgcn_rcpf((float)cprl));  // exact: gl, cprl < 4096
        const int c0 = (gl - r * cprl) * 8;
        const int wwl = sel4(ww[0], ww[1], ww[2], ww[3], l);
        const int Wl = sel4(G.W0, G.W1, G.W2, G.W3, l);
        const int row = sel4(G.s0, G.s1, G.s2, G.s3, l) + (sel4(wy0[0], wy0[1], wy0[2], wy0[3], l) + r) * Wl +
                        sel4(wx0[0], wx0[1], wx0[2], wx0[3], l);
        dsrc = (row + c0) * 1024;
        ddst = (sel4(base[0], base[1], base[2], base[3], l) + r * wwl + c0) * 16;
        dcnt = (g < cend[3]) ? min(8, wwl - c0) : 0;
      }
      if (tid < 8) *reinterpret_cast<float4*>(smem + tid * PLANE) = make_float4(0.f, 0.f, 0.f, 0.f);
      const int myc = (cend[3] - wave + NW - 1) / NW;  // chunks of this wave (uniform)
      // two batches: each issues all of its loads before its first store (4 x BATCH staging registers)
#pragma unroll 1
      for (int h = 0; h < 2; ++h) {
        float4 tmp[BATCH];
#pragma unroll
        for (int ii = 0; ii < BATCH; ++ii) {
          const int i = h * BATCH + ii;
          tmp[ii] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (i < myc) {
            // lanes past the end of the row re-read its last pixel (branch-free; they do not store)
            const int src = __builtin_amdgcn_readlane(dsrc, i), cnt = __builtin_amdgcn_readlane(dcnt, i);
            tmp[ii] = *reinterpret_cast<const float4*>(vb + (unsigned)src +
                                                       (unsigned)(min(sub, cnt - 1) * 1024 + quad * 16));
          }
        }
#pragma unroll
        for (int ii = 0; ii < BATCH; ++ii) {
          const int i = h * BATCH + ii;
          if (i < myc) {
            const int dst = __builtin_amdgcn_readlane(ddst, i), cnt = __builtin_amdgcn_readlane(dcnt, i);
            if (sub < cnt) *reinterpret_cast<float4*>(smem + quad * PLANE + dst + sub * 16) = tmp[ii];
          }
        }
      }
    }
    __syncthreads();  // C: windows staged
    if (PROF) t2 = __builtin_amdgcn_s_memtime();

    // ---- 5: gather, level by level (rolled): per corner one LDS address, 8 ds_read_b128 with immediate plane
    //         offsets, 32 FMAs; a level that is not staged reads its corners from global memory. ----------------------
    f32x4 acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int l = 0; l < 4; ++l) {
      EGTR_LEVEL_LOC(lx, ly, l)
      const float at[4] = {sel4f(aw[0].x, aw[1].x, aw[2].x, aw[3].x, l), sel4f(aw[0].y, aw[1].y, aw[2].y, aw[3].y, l),
                           sel4f(aw[0].z, aw[1].z, aw[2].z, aw[3].z, l), sel4f(aw[0].w, aw[1].w, aw[2].w, aw[3].w, l)};
      const int H = SEL_H(G, l), W = SEL_W(G, l);
      if ((staged >> l) & 1u) {
        LevelWin g;
        g.H = H;
        g.W = W;
        g.wx0 = sel4(wx0[0], wx0[1], wx0[2], wx0[3], l);
        g.wy0 = sel4(wy0[0], wy0[1], wy0[2], wy0[3], l);
        g.pitch = sel4(ww[0], ww[1], ww[2], ww[3], l);
        g.mw = g.pitch - 1;
        g.mh = sel4(wh[0], wh[1], wh[2], wh[3], l) - 1;
        g.base = sel4(base[0], base[1], base[2], base[3], l);
        LevelGather<PLANE> lg;
        lg.run(acc, lx, ly, at, g, lds0);
      } else {
        const int st = SEL_S(G, l);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const SampleGeom sg = sample_geom<1024, 128>(lx[p], ly[p], H, W, st, 0);
          const char* c0 = vb + (unsigned)sg.off[0];
          const char* c1 = vb + (unsigned)sg.off[1];
          const char* c2 = vb + (unsigned)sg.off[2];
          const char* c3 = vb + (unsigned)sg.off[3];
          const float w0 = sg.ok[0] ? sg.w[0] * at[p] : 0.f, w1 = sg.ok[1] ? sg.w[1] * at[p] : 0.f;
          const float w2 = sg.ok[2] ? sg.w[2] * at[p] : 0.f, w3 = sg.ok[3] ? sg.w[3] * at[p] : 0.f;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float4 v0 = *reinterpret_cast<const float4*>(c0 + k * 16);
            const float4 v1 = *reinterpret_cast<const float4*>(c1 + k * 16);
            const float4 v2 = *reinterpret_cast<const float4*>(c2 + k * 16);
            const float4 v3 = *reinterpret_cast<const float4*>(c3 + k * 16);
            acc[k].x += w0 * v0.x + w1 * v1.x + w2 * v2.x + w3 * v3.x;
            acc[k].y += w0 * v0.y + w1 * v1.y + w2 * v2.y + w3 * v3.y;
            acc[k].z += w0 * v0.z + w1 * v1.z + w2 * v2.z + w3 * v3.z;
            acc[k].w += w0 * v0.w + w1 * v1.w + w2 * v2.w + w3 * v3.w;
          }
        }
      }
    }
    __syncthreads();  // D: every wave is done reading the windows (the out transpose reuses the scratch rows)
    if (PROF) t3 = __builtin_amdgcn_s_memtime();

    // ---- 6: output rows -> coalesced 128-B stores ------------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < 8; ++k)
      *reinterpret_cast<float4*>(scr + lane * LOCROW + k * 16) = make_float4(acc[k].x, acc[k].y, acc[k].z, acc[k].w);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int sl = 8 * i + sub;
      const int qs = __shfl(qh, sl);
      const float4 v = *reinterpret_cast<const float4*>(scr + sl * LOCROW + quad * 16);
      if (qs >= 0) reinterpret_cast<float4*>(out)[(size_t)qs * 8 + quad] = v;
    }
    if (PROF) {
      const unsigned long long t4 = __builtin_amdgcn_s_memtime();
      if (tid == 0) {
        atomicAdd(prof + 0, t1 - t0);
        atomicAdd(prof + 1, t2 - t1);
        atomicAdd(prof + 2, t3 - t2);
        atomicAdd(prof + 3, t4 - t3);
        atomicAdd(prof + 4, 1ull);
        if (staged == 0xFu) {
          atomicAdd(prof + 5, 1ull);
          atomicAdd(prof + 6, t3 - t2);
          atomicAdd(prof + 7, t4 - t0);
        }
      }
    }
  }
}

template <bool PROF, int NW, int WINPX>
int launch_lane(hipStream_t st, const float* value, const int64_t* shapes, const int64_t* lsi, const float* loc,
                const float* attn, float* out, int B, int Lq, int S, unsigned long long* prof) {
  constexpr int per_cu = (160 * 1024) / (8 * (WINPX * 16 + 16) + 128);
  hipLaunchKernelGGL((msda_fwd_lane_f32<PROF, NW, WINPX>), dim3(256 * per_cu), dim3(64 * NW), 0, st, value, shapes,
                     lsi, loc, attn, out, B, Lq, S, prof);
  return egtr_check_launch();
}

}  // namespace

// Launchers used by egtr_msda_forward_f32_variant (msda.hip) for M = 8, D = 32, L = P = 4.
//   kind 0: 2 waves per workgroup (16 x 8 query tiles), 632-pixel windows, 2 workgroups per CU
//   kind 1: 1 wave per workgroup (16 x 4 query tiles), 312-pixel windows, 4 workgroups per CU
int egtr_launch_msda_fwd_lane_f32(hipStream_t st, const float* value, const int64_t* shapes, const int64_t* lsi,
                                  const float* loc, const float* attn, float* out, int B, int Lq, int S, int kind,
                                  unsigned long long* prof) {
  if (prof) {
    if (kind == 1) return launch_lane<true, 1, 312>(st, value, shapes, lsi, loc, attn, out, B, Lq, S, prof);
    return launch_lane<true, 2, 632>(st, value, shapes, lsi, loc, attn, out, B, Lq, S, prof);
  }
  if (kind == 1) return launch_lane<false, 1, 312>(st, value, shapes, lsi, loc, attn, out, B, Lq, S, nullptr);
  return launch_lane<false, 2, 632>(st, value, shapes, lsi, loc, attn, out, B, Lq, S, nullptr);
}
