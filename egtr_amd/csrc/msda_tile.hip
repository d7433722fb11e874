// MSDA forward/backward, "tile x head" variant for gfx950: LDS-staged sampling windows.
//
// Why: the wave-per-query kernel (msda.hip) gathers 4 corners x 16 samples x 128 B = 8 KiB per (query, head)
// through the vector L1 / texture addresser, i.e. 841 MB per encoder launch at 600x1000 against 45 MB of
// compulsory traffic; rocprofv3 shows it bound by the TA/L1 gather rate (~64 B/clk/CU), not by HBM
// (DESIGN.md 4.1).  Neighbouring encoder queries sample neighbouring pixels, so a tile of 8x8 queries of one head
// touches, per level, a window of only ~(8 s + 5)^2 pixels (s = scale between the query's level and the sampled
// level).  This kernel stages those windows ONCE in LDS (160 KiB/CU, 256 B/clk for ds_read_b128 = 4x the L1
// rate) and serves the 16 x 4 corner reads of every query from there.
//
// Work item = (batch, query tile of 64, head); 512 threads = 64 queries x 8 lanes (lane = 4 fp32 channels).
//   A  each thread computes the bilinear geometry of 2 of its query's 16 samples and reduces, per level, the
//      bounding box of all in-range corners of the tile (xor-shuffles over the queries of a wave + LDS min/max);
//   B  windows are packed into the LDS budget (level by level; a level whose window does not fit -- e.g. level-1
//      queries sampling level 0, or wildly scattered learned offsets -- is served straight from global memory, so
//      the result never depends on the windows, only the speed does); per-sample records {4 corner byte offsets,
//      4 bilinear x attention weights} are written (offsets point into the LDS window or into `value`);
//      the windows are copied global -> LDS, 16 B per lane, 128-B lines;
//   C  gather: per sample two broadcast ds_read_b128 for the record + four 16-B corner reads (LDS or global).
// Persistent launch: 512 workgroups (2 per CU) stride over the work items; XCD x (workgroup b runs on XCD b % 8)
// walks the x-th contiguous chunk of (tile, head) items = a compact spatial region with all 8 heads.  (Giving each
// XCD ONE head of every tile instead was measured 10x worse in load latency: a fixed 128-B sub-line of every 1 KiB
// pixel row camps on a quarter of the memory channels.)
// Encoder mode (Lq == sum H_l W_l): queries are the pixels of the levels, tiles are 8x8 pixel blocks.
// Otherwise tiles are 64 consecutive queries (correct for any query set; windows simply rarely fit).
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>

#include "common.h"
#include "msda_common.h"

using namespace egtr_msda;

namespace {

constexpr int kTQ = 64;         // queries per tile
constexpr int kThreads = 512;   // kTQ x 8 lanes
constexpr int kWinPx = 360;     // LDS window budget in pixels (128 B each); pixel 0 is an all-zero pixel
constexpr int kRecStride = 17;  // record entries per query (16 samples + 1 pad: conflict-free ds_read_b128)

// PROF: accumulate per-phase shader-clock cycles of thread 0 of every workgroup into prof[0..3] (A, B, C, count).
//   TH x TW  query tile (encoder mode), WINPX = LDS window budget in pixels, WPS = min waves per SIMD for the register
//   allocator (blocks per CU x waves per block / 4).  <8,8,360,4>: 64-query tiles, 80 KB LDS, 2 blocks per CU;
//   <4,4,160,3>: 16-query tiles (2 waves), 29 KB LDS, 5 blocks per CU -- more independent work items in flight per CU
//   to overlap the global round trips (loc/attn, then the windows) that a work item makes back to back.
template <bool PROF, int TH, int TW, int WINPX, int WPS>
__global__ __launch_bounds__(TH * TW * 8, WPS) void msda_fwd_tile_f32(
    const float* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
    const float* __restrict__ loc, const float* __restrict__ attn, float* __restrict__ out, int B, int Lq, int S,
    int L, int P, unsigned long long* __restrict__ prof) {
  constexpr int TQ = TH * TW;
  constexpr int NPASS = (WINPX + TQ - 1) / TQ;
  __shared__ __attribute__((aligned(16))) float4 s_win[WINPX * 8];
  __shared__ __attribute__((aligned(16))) int4 s_off[TQ * kRecStride];
  __shared__ __attribute__((aligned(16))) float4 s_wt[TQ * kRecStride];
  __shared__ int s_bbox[16];  // [level][ymin, ymax, xmin, xmax]

  const int tid = threadIdx.x, ql = tid >> 3, c4 = tid & 7;
  LevelGeom G;
  load_geom(shapes, lsi, L, G);
  const TileMap tm = make_tile_map<TH, TW>(G, L, Lq);
  const int nwork = B * tm.ntiles * 8;
  if (tid < 8) s_win[tid] = make_float4(0.f, 0.f, 0.f, 0.f);  // the all-zero pixel

  // loc / attn of the NEXT work item are prefetched into registers while the current one is staged and gathered
  float4 lc_next = make_float4(9.f, 9.f, 9.f, 9.f);  // far outside -> invalid
  float2 aw_next = make_float2(0.f, 0.f);
  auto prefetch = [&](int w) {
    lc_next = make_float4(9.f, 9.f, 9.f, 9.f);
    aw_next = make_float2(0.f, 0.f);
    if (w < nwork) {
      const int wl = xcd_remap(w, nwork);
      const int hd = wl & 7, tt = wl >> 3;
      const int bb = tt / tm.ntiles, tl = tt - bb * tm.ntiles;
      const int qq = tile_query<TH, TW>(tm, G, tl, ql, Lq);
      if (qq >= 0) {
        const size_t qh = ((size_t)bb * Lq + qq) * 8 + hd;
        lc_next = reinterpret_cast<const float4*>(loc + qh * 32)[c4];
        aw_next = reinterpret_cast<const float2*>(attn + qh * 16)[c4];
      }
    }
  };
  prefetch(blockIdx.x);

  for (int work = blockIdx.x; work < nwork; work += gridDim.x) {
    // XCD x (= work % 8: workgroup b runs on XCD b % 8 and gridDim.x % 8 == 0) walks the x-th contiguous chunk of
    // (tile, head) items: a compact spatial region per private L2, and all 8 heads (= all 128-B sub-lines of every
    // 1 KiB pixel row) so that memory-channel interleaving is not defeated.
    const int wlog = xcd_remap(work, nwork);
    const int head = wlog & 7;
    const int t = wlog >> 3;
    const int b = t / tm.ntiles, tile = t - b * tm.ntiles;
    const int q = tile_query<TH, TW>(tm, G, tile, ql, Lq);
    const char* vbase = reinterpret_cast<const char*>(value) + (size_t)b * S * 1024;
    const float4 lc = lc_next;
    const float2 aw = aw_next;

    if (tid < 16) s_bbox[tid] = (tid & 1) ? INT_MIN : INT_MAX;
    __syncthreads();
    unsigned long long t0 = 0, t1 = 0, t2 = 0;
    if (PROF) t0 = __builtin_amdgcn_s_memtime();

    // ---- A: geometry of samples 2*c4, 2*c4+1 of (q, head); both lie in the same level (P is even) ----------
    const int lvl = (2 * c4) / P;
    const int H = SEL_H(G, lvl), W = SEL_W(G, lvl), st = SEL_S(G, lvl);
    int y0[2], x0[2];
    float wgt[2][4];
    bool any[2];
    int ymin = INT_MAX, ymax = INT_MIN, xmin = INT_MAX, xmax = INT_MIN;
    {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const SampleGeom g = sample_geom<1024, 128>(j ? lc.z : lc.x, j ? lc.w : lc.y, H, W, st, head);
        const float a = j ? aw.y : aw.x;
        y0[j] = g.y0;
        x0[j] = g.x0;
        any[j] = g.ok[0] || g.ok[1] || g.ok[2] || g.ok[3];
#pragma unroll
        for (int k = 0; k < 4; ++k) wgt[j][k] = g.ok[k] ? g.w[k] * a : 0.f;
        if (any[j]) {
          ymin = min(ymin, max(y0[j], 0));
          ymax = max(ymax, min(y0[j] + 1, H - 1));
          xmin = min(xmin, max(x0[j], 0));
          xmax = max(xmax, min(x0[j] + 1, W - 1));
        }
      }
    }
    // lanes l, l^8, l^16, l^32 hold the same c4 (same level): reduce over the 8 queries of the wave
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
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
    if (PROF) t1 = __builtin_amdgcn_s_memtime();
    prefetch(work + gridDim.x);

    // ---- B: pack windows into the LDS budget (uniform), write records, stage windows ------------------------
    int wy0[4], wx0[4], wy1[4], wx1[4], ww[4], base[4], npx[4];
    unsigned staged = 0;
    {
      int off = 1;  // pixel 0 = zeros
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        wy0[l] = s_bbox[l * 4 + 0];
        wy1[l] = s_bbox[l * 4 + 1];
        wx0[l] = s_bbox[l * 4 + 2];
        wx1[l] = s_bbox[l * 4 + 3];
        const bool empty = (l >= L) || (wy0[l] > wy1[l]);
        ww[l] = empty ? 0 : (wx1[l] - wx0[l] + 1);
        npx[l] = empty ? 0 : ww[l] * (wy1[l] - wy0[l] + 1);
        base[l] = off;
        if (off + npx[l] <= WINPX) {
          staged |= 1u << l;
          off += npx[l];
        } else {
          npx[l] = 0;  // not staged: nothing to copy
        }
      }
    }
    {
      const bool st_l = (staged >> lvl) & 1u;
      const int by0 = sel4(wy0[0], wy0[1], wy0[2], wy0[3], lvl), by1 = sel4(wy1[0], wy1[1], wy1[2], wy1[3], lvl);
      const int bx0 = sel4(wx0[0], wx0[1], wx0[2], wx0[3], lvl), bx1 = sel4(wx1[0], wx1[1], wx1[2], wx1[3], lvl);
      const int bww = sel4(ww[0], ww[1], ww[2], ww[3], lvl), bbase = sel4(base[0], base[1], base[2], base[3], lvl);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int s = 2 * c4 + j;
        int4 o;
        if (!any[j]) {
          o = make_int4(0, 0, 0, 0);  // zero pixel (LDS) / first pixel (global); all weights are 0
          if (!st_l) o = make_int4(head * 128, head * 128, head * 128, head * 128);
        } else if (st_l) {
          const int ya = min(max(y0[j], by0), by1), yb = min(max(y0[j] + 1, by0), by1);
          const int xa = min(max(x0[j], bx0), bx1), xb = min(max(x0[j] + 1, bx0), bx1);
          const int r0 = bbase + (ya - by0) * bww - bx0, r1 = bbase + (yb - by0) * bww - bx0;
          o = make_int4((r0 + xa) * 128, (r0 + xb) * 128, (r1 + xa) * 128, (r1 + xb) * 128);
        } else {
          const int ya = max(y0[j], 0), yb = min(y0[j] + 1, H - 1), xa = max(x0[j], 0), xb = min(x0[j] + 1, W - 1);
          const int r0 = (st + ya * W) * 1024 + head * 128, r1 = (st + yb * W) * 1024 + head * 128;
          o = make_int4(r0 + xa * 1024, r0 + xb * 1024, r1 + xa * 1024, r1 + xb * 1024);
        }
        s_off[ql * kRecStride + s] = o;
        s_wt[ql * kRecStride + s] = make_float4(wgt[j][0], wgt[j][1], wgt[j][2], wgt[j][3]);
      }
    }
    {
      // Copy the staged windows global -> LDS.  Flattened over all levels so that every thread first ISSUES all of
      // its (up to 6) 16-byte loads and only then stores them (one global round trip per work item, not per pass).
      const int total = base[3] + npx[3];  // one past the last staged pixel
      float4 tmp[NPASS];
#pragma unroll
      for (int i = 0; i < NPASS; ++i) {
        const int p = 1 + ql + TQ * i;
        tmp[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p < total) {
          const int l = (p >= base[1] ? 1 : 0) + (p >= base[2] ? 1 : 0) + (p >= base[3] ? 1 : 0);
          const int rel = p - sel4(base[0], base[1], base[2], base[3], l);
          const int wwl = sel4(ww[0], ww[1], ww[2], ww[3], l);
          const int r = (int)(((float)rel + 0.5f) * __frcp_rn((float)wwl));  // exact for rel, wwl < 4096
          const int c = rel - r * wwl;
          const int Wl = sel4(G.W0, G.W1, G.W2, G.W3, l), stl = sel4(G.s0, G.s1, G.s2, G.s3, l);
          const int yy = sel4(wy0[0], wy0[1], wy0[2], wy0[3], l) + r, xx = sel4(wx0[0], wx0[1], wx0[2], wx0[3], l) + c;
          tmp[i] = *reinterpret_cast<const float4*>(vbase + ((size_t)(stl + yy * Wl + xx) * 1024 + head * 128 + c4 * 16));
        }
      }
#pragma unroll
      for (int i = 0; i < NPASS; ++i) {
        const int p = 1 + ql + TQ * i;
        if (p < total) s_win[p * 8 + c4] = tmp[i];
      }
    }
    __syncthreads();
    if (PROF) t2 = __builtin_amdgcn_s_memtime();

    // ---- C: gather (level by level; the LDS / global choice is uniform per level, two separate code paths so
    //      the LDS reads stay ds_read_b128 and never degrade to flat loads) ------------------------------------
    {
      const int4* ro = s_off + ql * kRecStride;
      const float4* rw = s_wt + ql * kRecStride;
      const char* glb = vbase + c4 * 16;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#define EGTR_FMA4()                                                \
      acc.x += w.x * v0.x + w.y * v1.x + w.z * v2.x + w.w * v3.x;  \
      acc.y += w.x * v0.y + w.y * v1.y + w.z * v2.y + w.w * v3.y;  \
      acc.z += w.x * v0.z + w.y * v1.z + w.z * v2.z + w.w * v3.z;  \
      acc.w += w.x * v0.w + w.y * v1.w + w.z * v2.w + w.w * v3.w;
      for (int l = 0; l < L; ++l) {
        if ((staged >> l) & 1u) {
#pragma unroll 4
          for (int pp = 0; pp < P; ++pp) {
            const int4 o = ro[l * P + pp];
            const float4 w = rw[l * P + pp];
            const float4 v0 = s_win[(o.x >> 4) + c4], v1 = s_win[(o.y >> 4) + c4];
            const float4 v2 = s_win[(o.z >> 4) + c4], v3 = s_win[(o.w >> 4) + c4];
            EGTR_FMA4()
          }
        } else {
#pragma unroll 4
          for (int pp = 0; pp < P; ++pp) {
            const int4 o = ro[l * P + pp];
            const float4 w = rw[l * P + pp];
            const float4 v0 = *reinterpret_cast<const float4*>(glb + (unsigned)o.x);
            const float4 v1 = *reinterpret_cast<const float4*>(glb + (unsigned)o.y);
            const float4 v2 = *reinterpret_cast<const float4*>(glb + (unsigned)o.z);
            const float4 v3 = *reinterpret_cast<const float4*>(glb + (unsigned)o.w);
            EGTR_FMA4()
          }
        }
      }
#undef EGTR_FMA4
      if (q >= 0) reinterpret_cast<float4*>(out + (((size_t)b * Lq + q) * 8 + head) * 32)[c4] = acc;
    }
    if (PROF) {
      const unsigned long long t3 = __builtin_amdgcn_s_memtime();
      if (tid == 0) {
        atomicAdd(prof + 0, t1 - t0);
        atomicAdd(prof + 1, t2 - t1);
        atomicAdd(prof + 2, t3 - t2);
        atomicAdd(prof + 3, 1ull);
      }
    }
    // the next iteration's first barrier orders its LDS writes after every thread's reads of this one
  }
}


// ------------------------------------------------------------------------------------------------ backward
// grad_value only (grad_attn / grad_loc come from the wave-per-query kernel in msda.hip, run without its atomics).
// The scatter  grad_value[pixel, c] += sum_{q, sample, corner} w * attn * grad_out[q, c]  is turned into a small dense
// matrix product per work item (query tile of 64, head):  G[pixel, c] = A^T[pixel, q] . T[q, c]  with
//   A[q, pixel] (64 x Np fp32, LDS) = bilinear x attention weight mass query q puts on window pixel `pixel`
//                                     (built without atomics: row q is written by the threads of query q only and the
//                                     levels occupy disjoint column ranges),
//   T[q, c]     (64 x 32)           = grad_out of the tile for this head,
// evaluated on the matrix cores with v_mfma_f32_32x32x2_f32 (exact f32); G is then added to global memory with ONE
// coalesced atomic per window element (32 lanes = 128 contiguous bytes).  The windows of the L levels are concatenated
// into one virtual pixel range that is processed in chunks of kChunk columns, so any window size is handled (more
// chunks, never a different code path).  History: per-(sample, corner, channel) global atomics as in
// the reference (cuh:125-152) = 2.4 ms per encoder launch; LDS ds_add_f32 accumulation = 1.2 ms (LDS float atomics
// are serialised per lane, ~130 cycles per instruction).
constexpr int kChunk = 448;  // columns of A per pass (multiple of 32)

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(kThreads, 2) void msda_bwd_value_tile_f32(
    const float* __restrict__ grad_out, const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
    const float* __restrict__ loc, const float* __restrict__ attn, float* __restrict__ grad_value, int B, int Lq,
    int S, int L, int P) {
  __shared__ __attribute__((aligned(16))) float s_A[kTQ * kChunk];         // A[q][column]
  __shared__ __attribute__((aligned(16))) float4 s_T[kTQ * 8];             // T[q][32 channels]
  __shared__ __attribute__((aligned(16))) float4 s_w[kTQ * kRecStride];    // per sample: 4 weights (x attention)
  __shared__ __attribute__((aligned(8))) int2 s_vp[kTQ * kRecStride];      // per sample: 4 virtual pixel ids (16 bit)
  __shared__ int s_pix[kChunk];                                            // global byte offset of a chunk column
  __shared__ int s_bbox[16];

  const int tid = threadIdx.x, ql = tid >> 3, c4 = tid & 7;
  LevelGeom G;
  load_geom(shapes, lsi, L, G);
  const TileMap tm = make_tile_map(G, L, Lq);
  const int nwork = B * tm.ntiles * 8;

  for (int work = blockIdx.x; work < nwork; work += gridDim.x) {
    const int wlog = xcd_remap(work, nwork);
    const int head = wlog & 7;
    const int t = wlog >> 3;
    const int b = t / tm.ntiles, tile = t - b * tm.ntiles;
    const int q = tile_query(tm, G, tile, ql, Lq);
    char* gvbase = reinterpret_cast<char*>(grad_value) + (size_t)b * S * 1024;

    if (tid < 16) s_bbox[tid] = (tid & 1) ? INT_MIN : INT_MAX;
    __syncthreads();

    // ---- A: geometry of samples 2*c4, 2*c4+1 + per-level bounding boxes ------------------------------------------
    const int lvl = (2 * c4) / P;
    const int H = SEL_H(G, lvl), W = SEL_W(G, lvl), st = SEL_S(G, lvl);
    int y0[2], x0[2];
    float wgt[2][4];
    bool any[2];
    int ymin = INT_MAX, ymax = INT_MIN, xmin = INT_MAX, xmax = INT_MIN;
    {
      float4 lc = make_float4(9.f, 9.f, 9.f, 9.f);
      float2 aw = make_float2(0.f, 0.f);
      float4 go = make_float4(0.f, 0.f, 0.f, 0.f);
      if (q >= 0) {
        const size_t qh = ((size_t)b * Lq + q) * 8 + head;
        lc = reinterpret_cast<const float4*>(loc + qh * 32)[c4];
        aw = reinterpret_cast<const float2*>(attn + qh * 16)[c4];
        go = reinterpret_cast<const float4*>(grad_out + qh * 32)[c4];
      }
      s_T[ql * 8 + c4] = go;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const SampleGeom g = sample_geom<1024, 128>(j ? lc.z : lc.x, j ? lc.w : lc.y, H, W, st, head);
        const float a = j ? aw.y : aw.x;
        y0[j] = g.y0;
        x0[j] = g.x0;
        any[j] = g.ok[0] || g.ok[1] || g.ok[2] || g.ok[3];
#pragma unroll
        for (int k = 0; k < 4; ++k) wgt[j][k] = g.ok[k] ? g.w[k] * a : 0.f;
        if (any[j]) {
          ymin = min(ymin, max(y0[j], 0));
          ymax = max(ymax, min(y0[j] + 1, H - 1));
          xmin = min(xmin, max(x0[j], 0));
          xmax = max(xmax, min(x0[j] + 1, W - 1));
        }
      }
    }
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
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

    // ---- B: concatenate the level windows into one virtual pixel range; per-sample records ------------------------
    int wy0[4], wx0[4], ww[4], vb[5];
    vb[0] = 0;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      wy0[l] = s_bbox[l * 4 + 0];
      const int wy1 = s_bbox[l * 4 + 1];
      wx0[l] = s_bbox[l * 4 + 2];
      const int wx1 = s_bbox[l * 4 + 3];
      const bool empty = (l >= L) || (wy0[l] > wy1);
      ww[l] = empty ? 0 : (wx1 - wx0[l] + 1);
      vb[l + 1] = vb[l] + (empty ? 0 : ww[l] * (wy1 - wy0[l] + 1));
    }
    const int ntot = vb[4];
    {
      const int by0 = sel4(wy0[0], wy0[1], wy0[2], wy0[3], lvl), bx0 = sel4(wx0[0], wx0[1], wx0[2], wx0[3], lvl);
      const int bww = sel4(ww[0], ww[1], ww[2], ww[3], lvl), bvb = sel4(vb[0], vb[1], vb[2], vb[3], lvl);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int s = 2 * c4 + j;
        int v0 = 0, v1 = 0, v2 = 0, v3 = 0;
        if (any[j]) {  // clamped corners lie inside the level's bounding box by construction
          const int ya = max(y0[j], 0), yb = min(y0[j] + 1, H - 1), xa = max(x0[j], 0), xb = min(x0[j] + 1, W - 1);
          const int r0 = bvb + (ya - by0) * bww - bx0, r1 = bvb + (yb - by0) * bww - bx0;
          v0 = r0 + xa; v1 = r0 + xb; v2 = r1 + xa; v3 = r1 + xb;
        }
        s_vp[ql * kRecStride + s] = make_int2(v0 | (v1 << 16), v2 | (v3 << 16));
        s_w[ql * kRecStride + s] = make_float4(wgt[j][0], wgt[j][1], wgt[j][2], wgt[j][3]);
      }
    }

    // ---- C: chunks of kChunk virtual pixels: zero A, build A, MFMA, one atomic per element ----------------------
    for (int k0 = 0; k0 < ntot; k0 += kChunk) {
      const int ncols = min(kChunk, ntot - k0);
      const int mtiles = (ncols + 31) >> 5;
      __syncthreads();  // records visible (first chunk) / previous chunk's MFMA reads of A and s_pix finished
      {
        const int ncol4 = mtiles * 8;
        for (int e = tid; e < kTQ * ncol4; e += kThreads) {
          const int r = e / ncol4, c = e - r * ncol4;
          reinterpret_cast<float4*>(s_A + r * kChunk)[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (tid < ncols) {  // global byte offset of chunk column `tid`
          const int p = k0 + tid;
          const int l = (p >= vb[1] ? 1 : 0) + (p >= vb[2] ? 1 : 0) + (p >= vb[3] ? 1 : 0);
          const int rel = p - sel4(vb[0], vb[1], vb[2], vb[3], l);
          const int wwl = sel4(ww[0], ww[1], ww[2], ww[3], l);
          const int r = (int)(((float)rel + 0.5f) * __frcp_rn((float)wwl));
          const int c = rel - r * wwl;
          const int Wl = sel4(G.W0, G.W1, G.W2, G.W3, l), stl = sel4(G.s0, G.s1, G.s2, G.s3, l);
          const int yy = sel4(wy0[0], wy0[1], wy0[2], wy0[3], l) + r, xx = sel4(wx0[0], wx0[1], wx0[2], wx0[3], l) + c;
          s_pix[tid] = (stl + yy * Wl + xx) * 1024 + head * 128;
        }
      }
      __syncthreads();
      // build A: thread (q, level l = c4) owns the samples of level l of query q -> no cross-thread conflicts
      if (c4 < L) {
        float* arow = s_A + ql * kChunk;
        for (int pp = 0; pp < P; ++pp) {
          const int2 v = s_vp[ql * kRecStride + c4 * P + pp];
          const float4 w = s_w[ql * kRecStride + c4 * P + pp];
          const unsigned c0 = (unsigned)((v.x & 0xffff) - k0), c1 = (unsigned)(((unsigned)v.x >> 16) - k0);
          const unsigned c2 = (unsigned)((v.y & 0xffff) - k0), c3 = (unsigned)(((unsigned)v.y >> 16) - k0);
          if (w.x != 0.f && c0 < (unsigned)ncols) arow[c0] += w.x;
          if (w.y != 0.f && c1 < (unsigned)ncols) arow[c1] += w.y;
          if (w.z != 0.f && c2 < (unsigned)ncols) arow[c2] += w.z;
          if (w.w != 0.f && c3 < (unsigned)ncols) arow[c3] += w.w;
        }
      }
      __syncthreads();
      {
        const int wave = tid >> 6, lane = tid & 63, li = lane & 31, hf = lane >> 5;
        const float* tmat = reinterpret_cast<const float*>(s_T);
        for (int mt = wave; mt < mtiles; mt += kThreads / 64) {
          f32x16 acc;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = 0.f;
          const float* acol = s_A + mt * 32 + li;
#pragma unroll 8
          for (int s2 = 0; s2 < kTQ / 2; ++s2) {
            const int qq = 2 * s2 + hf;
            // D[i = pixel][j = channel] += A_op[i][k = qq] * B_op[k = qq][j]
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(acol[qq * kChunk], tmat[qq * 32 + li], acc, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int col = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
            if (col < ncols && acc[r] != 0.f)
              unsafeAtomicAdd(reinterpret_cast<float*>(gvbase + (unsigned)s_pix[col]) + li, acc[r]);
          }
        }
      }
    }
  }
}

}  // namespace

// Launcher used by egtr_msda_forward_f32 (msda.hip) for M = 8, D = 32, L*P = 16 when Lq == S (encoder layers).
int egtr_launch_msda_fwd_tile_f32(hipStream_t st, const float* value, const int64_t* shapes, const int64_t* lsi,
                                  const float* loc, const float* attn, float* out, int B, int Lq, int S, int L,
                                  int P) {
  hipLaunchKernelGGL((msda_fwd_tile_f32<false, 8, 8, 360, 4>), dim3(512), dim3(512), 0, st, value, shapes, lsi, loc,
                     attn, out, B, Lq, S, L, P, (unsigned long long*)nullptr);
  return egtr_check_launch();
}

// variant 4: 16-query (4 x 4) tiles, 2 waves per workgroup, 5 workgroups per CU
int egtr_launch_msda_fwd_tile16_f32(hipStream_t st, const float* value, const int64_t* shapes, const int64_t* lsi,
                                    const float* loc, const float* attn, float* out, int B, int Lq, int S, int L,
                                    int P) {
  hipLaunchKernelGGL((msda_fwd_tile_f32<false, 4, 4, 160, 3>), dim3(1280), dim3(128), 0, st, value, shapes, lsi, loc,
                     attn, out, B, Lq, S, L, P, (unsigned long long*)nullptr);
  return egtr_check_launch();
}

// Profiling hook: same kernel with per-phase cycle accounting (see include/egtr_hip.h).
extern "C" int egtr_msda_tile_phase_cycles(egtr_stream_t stream, const float* value, const int64_t* spatial_shapes,
                                           const int64_t* level_start_index, const float* sampling_loc,
                                           const float* attn_weight, int batch, int spatial_size, int num_levels,
                                           int num_query, int num_point, float* out, unsigned long long* cycles) {
  if (!value || !spatial_shapes || !level_start_index || !sampling_loc || !attn_weight || !out || !cycles)
    return EGTR_E_ARG;
  if (num_levels < 1 || num_levels > 4 || num_levels * num_point != 16) return EGTR_E_UNSUPPORTED;
  hipLaunchKernelGGL((msda_fwd_tile_f32<true, 8, 8, 360, 4>), dim3(512), dim3(512), 0,
                     static_cast<hipStream_t>(stream), value, spatial_shapes, level_start_index, sampling_loc,
                     attn_weight, out, batch, num_query, spatial_size, num_levels, num_point, cycles);
  return egtr_check_launch();
}

// Launcher used by egtr_msda_backward_f32 (msda.hip): grad_value of encoder-shaped calls (M = 8, D = 32, L*P = 16).
int egtr_launch_msda_bwd_value_tile_f32(hipStream_t st, const float* grad_out, const int64_t* shapes,
                                        const int64_t* lsi, const float* loc, const float* attn, float* grad_value,
                                        int B, int Lq, int S, int L, int P) {
  hipLaunchKernelGGL(msda_bwd_value_tile_f32, dim3(256), dim3(kThreads), 0, st, grad_out, shapes, lsi, loc, attn,
                     grad_value, B, Lq, S, L, P);
  return egtr_check_launch();
}
