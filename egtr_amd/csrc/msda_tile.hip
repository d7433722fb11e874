// MSDA backward, grad_value of encoder-shaped calls: query tile x head, the scatter as a small dense product on the
// matrix cores (see the kernel comment below).  The forward counterpart of this decomposition (LDS-staged sampling
// windows) and four further LDS designs were measured slower than the wave-per-query forward kernel and removed
// (DESIGN.md 4.1 keeps the measurements; the code is in the history: commits 76a60e2, aecda1b).
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>

#include "common.h"
#include "msda_common.h"

using namespace egtr_msda;

namespace {

constexpr int kTQ = 64;         // queries per tile
constexpr int kThreads = 512;   // kTQ x 8 lanes
constexpr int kRecStride = 17;  // record entries per query (16 samples + 1 pad: conflict-free ds_read_b128)

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
// kChunk: columns of A per pass (multiple of 32).  448 -> 115 KB of A, one workgroup per CU.  Measured round 2 with two
// workgroups per CU instead (chunk 160 / 128 / 96, grid 512): 321 / 309 / 376 us at B = 1 against 284 us, 653 / 642 /
// 750 us at B = 4 against 657 us (whole backward) -- the extra chunk passes cost what the second workgroup hides.
constexpr int kChunk = 448;
constexpr int kBwdGrid = 256;  // persistent workgroups, one per CU

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Work counters of the launches in flight: 8 per launch (one per head = XCD), a ring of kCounterSlots launches.  A slot is
// zeroed by the wave-per-query kernel that precedes its value-tile kernel on the stream (msda.hip).
// Launches recorded into a HIP graph keep their slot for the graph's lifetime (the pointer is baked into the graph and reused on
// every replay), so they draw from a region of their own that is never handed out twice: an eager launch on another stream
// can then not wrap the ring onto a slot a replay is using (two launch pairs sharing counters would skip or repeat tiles).
constexpr int kCounterSlots = 1024;   // eager launches: a ring
constexpr int kGraphSlots = 8192;     // captured launches: one-way
__device__ unsigned g_tile_counters[(kCounterSlots + kGraphSlots) * 8];

// GO_BF16: grad_out holds raw bfloat16 (the backward of a bf16 model); widened on load, everything else is unchanged.
template <bool GO_BF16>
__global__ __launch_bounds__(kThreads, 2) void msda_bwd_value_tile_f32(
    const void* __restrict__ grad_out_, const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
    const float* __restrict__ loc, const float* __restrict__ attn, float* __restrict__ grad_value, int B, int Lq,
    int S, int L, int P, unsigned* __restrict__ counters) {
  __shared__ __attribute__((aligned(16))) float s_A[kTQ * kChunk];         // A[q][column]
  __shared__ __attribute__((aligned(16))) float4 s_T[kTQ * 8];             // T[q][32 channels]
  __shared__ __attribute__((aligned(16))) float4 s_w[kTQ * kRecStride];    // per sample: 4 weights (x attention)
  __shared__ __attribute__((aligned(8))) int2 s_vp[kTQ * kRecStride];      // per sample: 4 virtual pixel ids (16 bit)
  __shared__ __attribute__((aligned(16))) int s_pix[kChunk];                                            // global byte offset of a chunk column
  __shared__ int s_bbox[16];
  __shared__ unsigned s_km[kTQ / 2];                                       // per query pair: column tiles of the chunk it touches
  __shared__ int s_grab[2];                                                // next item's index, handed from thread 0 to the workgroup

  const int tid = threadIdx.x, ql = tid >> 3, c4 = tid & 7;
  LevelGeom G;
  load_geom(shapes, lsi, L, G);
  const TileMap tm = make_tile_map(G, L, Lq);
  // Items are handed out dynamically, largest first.  head = XCD: workgroups are dealt to the eight XCDs round-robin by id and
  // the grid is a multiple of 8, so blockIdx & 7 is the XCD this workgroup runs on; it takes the items of THAT head, so that all
  // atomics on the 128-byte head slice [.., head, :] of a pixel come from one XCD (437 -> 394 us per launch at B = 4 in round 3;
  // the atomics themselves still execute on the fabric side -- WRITE_SIZE unchanged at 205 MB, L2 hit ~0).  Item j of a head is
  // (image j % B, tile ntiles - 1 - j / B): the tile list is ordered level 0 ... level 3 and the window of a coarse-level tile
  // is many times that of a level-0 tile (1 chunk against ~14), so the big items go first and the 32 workgroups of an XCD pull
  // the next item from the head's counter when they finish one (round 4; with the static round-robin of before, the slowest
  // workgroup took 814 k cycles against a mean of 685 k).  An index is drawn two items ahead of its use -- the operands of
  // the next item are in flight while the current one is multiplied out, and the atomic's result is picked up an item later.
  const int head = blockIdx.x & 7;
  const int nitems = B * tm.ntiles;
  unsigned* ctr = counters + head;
  // A is zero whenever a chunk starts building it: zeroed once here, and every element a chunk filled is zeroed again by the
  // lane that consumed it in the MFMA phase (round 4: the 115 KB zero pass per chunk and its barrier are gone)
  for (int e = tid; e < kTQ * kChunk / 4; e += kThreads) reinterpret_cast<float4*>(s_A)[e] = make_float4(0.f, 0.f, 0.f, 0.f);

  // operands of the NEXT work item are requested while the current one is multiplied out (round 4: the loads of loc / attn /
  // grad_out were exposed at the top of every item, with one workgroup per CU nothing else runs meanwhile)
  float4 lc_n = make_float4(9.f, 9.f, 9.f, 9.f), go_n = make_float4(0.f, 0.f, 0.f, 0.f);
  float2 aw_n = make_float2(0.f, 0.f);
  auto fetch = [&](int item_) {
    lc_n = make_float4(9.f, 9.f, 9.f, 9.f);
    aw_n = make_float2(0.f, 0.f);
    go_n = make_float4(0.f, 0.f, 0.f, 0.f);
    if (item_ < nitems) {
      const int b_ = item_ % B, tile_ = tm.ntiles - 1 - item_ / B;
      const int q_ = tile_query(tm, G, tile_, ql, Lq);
      if (q_ >= 0) {
        const size_t qh = ((size_t)b_ * Lq + q_) * 8 + head;
        lc_n = reinterpret_cast<const float4*>(loc + qh * 32)[c4];
        aw_n = reinterpret_cast<const float2*>(attn + qh * 16)[c4];
        if constexpr (GO_BF16) {
          const uint2 gb = reinterpret_cast<const uint2*>(static_cast<const uint16_t*>(grad_out_) + qh * 32)[c4];
          go_n = make_float4(__uint_as_float(gb.x << 16), __uint_as_float(gb.x & 0xffff0000u), __uint_as_float(gb.y << 16),
                             __uint_as_float(gb.y & 0xffff0000u));
        } else {
          go_n = reinterpret_cast<const float4*>(static_cast<const float*>(grad_out_) + qh * 32)[c4];
        }
      }
    }
  };
  // The first two items of a workgroup are fixed -- w and npx + w for the w-th of the npx workgroups of its XCD, so that the
  // largest items land on different workgroups -- and the counter hands out the indices from 2 npx on.  Thread 0 keeps the
  // index requested during the previous item in `pending` and publishes it at the top of the next one.
  const int npx = (int)(gridDim.x >> 3), w = (int)(blockIdx.x >> 3);
  const unsigned drawn0 = 2u * (unsigned)npx;
  unsigned pending = (unsigned)(npx + w);
  int cur = w;
  int par = 0;   // hand-over slot of this item (the other one may still be read by a wave that is behind)
  fetch(cur);

  while (cur < nitems) {
    const int b = cur % B;
    char* gvbase = reinterpret_cast<char*>(grad_value) + (size_t)b * S * 1024;

    if (tid < 16) s_bbox[tid] = (tid & 1) ? INT_MIN : INT_MAX;
    if (tid == 0) {
      s_grab[par] = (int)pending;              // requested during the previous item: arrived long ago
      pending = atomicAdd(ctr, 1u) + drawn0;   // picked up at the top of the next item
    }
    __syncthreads();
    const int nxt = s_grab[par];

    // ---- A: geometry of samples 2*c4, 2*c4+1 + per-level bounding boxes ------------------------------------------
    const int lvl = (2 * c4) / P;
    const int H = SEL_H(G, lvl), W = SEL_W(G, lvl), st = SEL_S(G, lvl);
    int y0[2], x0[2];
    float wgt[2][4];
    bool any[2];
    int ymin = INT_MAX, ymax = INT_MIN, xmin = INT_MAX, xmax = INT_MIN;
    {
      const float4 lc = lc_n, go = go_n;
      const float2 aw = aw_n;
      fetch(nxt);   // in flight until the next iteration reads lc_n / aw_n / go_n
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
      __syncthreads();  // records visible (first chunk) / previous chunk's MFMA phase (reads + re-zeroing of A, s_pix) finished
      {
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
      // build A: the samples of level l of query q belong to TWO threads (c4 = 2 l, 2 l + 1): one adds the two top corners of
      // every sample, the other the two bottom corners.  Within one instruction the two threads work on the SAME sample, whose
      // top and bottom corners are different pixels whenever both carry weight (a clamped duplicate has weight 0 and is
      // skipped), different queries are different rows of A and different levels disjoint column ranges: no two lanes of an
      // instruction touch one element, and a wave's LDS operations execute in program order -- plain read-modify-writes, no
      // atomics.  (Round 4: one thread per (query, level) did all four corners: a chain of 16 dependent LDS round trips.)
      unsigned touched = 0;  // bit t: this thread put weight into column tile t (32 columns) of the chunk
      if ((c4 >> 1) < L) {
        float* arow = s_A + ql * kChunk;
        const int lvl_b = c4 >> 1, bottom = c4 & 1;
        for (int pp = 0; pp < P; ++pp) {
          const int2 v = s_vp[ql * kRecStride + lvl_b * P + pp];
          const float4 w = s_w[ql * kRecStride + lvl_b * P + pp];
          const int vv = bottom ? v.y : v.x;
          const float wa = bottom ? w.z : w.x, wb = bottom ? w.w : w.y;
          const unsigned ca = (unsigned)((vv & 0xffff) - k0), cb = (unsigned)(((unsigned)vv >> 16) - k0);
          if (wa != 0.f && ca < (unsigned)ncols) { arow[ca] += wa; touched |= 1u << (ca >> 5); }
          if (wb != 0.f && cb < (unsigned)ncols) { arow[cb] += wb; touched |= 1u << (cb >> 5); }
        }
      }
      // Which column tiles does the query PAIR (2 s2, 2 s2 + 1) = one k step of the MFMA loop = 16 consecutive lanes touch?
      // OR over the 16 lanes (two quad permutes, then the two mirrors: OR does not care which lane a value came from).
      touched |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)touched, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
      touched |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)touched, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
      touched |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)touched, 0x141, 0xF, 0xF, true);  // row_half_mirror
      touched |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)touched, 0x140, 0xF, 0xF, true);  // row_mirror
      if ((tid & 15) == 0) s_km[tid >> 4] = touched;
      __syncthreads();
      {
        const int wave = tid >> 6, lane = tid & 63, li = lane & 31, hf = lane >> 5;
        const float* tmat = reinterpret_cast<const float*>(s_T);
        const unsigned kmv = s_km[li];  // lane l (and l + 32): the column tiles query pair l touches
        for (int mt = wave; mt < mtiles; mt += kThreads / 64) {
          // Only the k steps (query pairs) that put weight into this column tile are multiplied: the rest of the tile's
          // 64 x 32 block of A is zero.  Rows of A are sparse -- 64 nonzeros in a window of up to thousands of columns for
          // the coarse-level tiles -- so most (column tile, k step) blocks are empty (round 4; DESIGN 4.2 has the counts).
          unsigned km = (unsigned)__builtin_amdgcn_ballot_w64(((kmv >> mt) & 1u) != 0);  // low half = high half
          if (km == 0) continue;   // nobody touched the tile: nothing to add, nothing to re-zero
          f32x16 acc;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = 0.f;
          float* acol = s_A + mt * 32 + li;
          // D[i = pixel][j = channel] += A_op[i][k = qq] * B_op[k = qq][j], four set bits at a time: the operands of the NEXT
          // four are requested before the current four products are issued (an LDS round trip is ~200 cycles here, a product
          // 64).  A group's missing steps repeat its first one with a zero A operand.
          // Every element is set back to zero right behind its read -- this lane is its only consumer, and A has to be zero
          // when the next chunk / item starts filling it.  A group's missing steps repeat its first one: they read the zero just
          // written (a wave's LDS operations execute in order) and add nothing.
          float av[4], tv[4];
          auto take4 = [&]() {
            const int first = __builtin_ctz(km);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              int s2 = km != 0 ? __builtin_ctz(km) : first;
              km &= km - 1;
              asm volatile("" : "+s"(s2));   // opaque: no per-case code paths, the four read pairs are issued back to back
              const int qq = 2 * s2 + hf;
              float* ap = acol + __mul24(qq, kChunk);   // 24-bit multiply: hipcc otherwise picks a 64-bit multiply-add whose
              av[u] = *ap;                              // unused high half aliased a pending read's register (a stall per group)
              *ap = 0.f;
              tv[u] = tmat[qq * 32 + li];
            }
          };
          take4();
          for (;;) {
            const float a0 = av[0], a1 = av[1], a2 = av[2], a3 = av[3], t0 = tv[0], t1 = tv[1], t2 = tv[2], t3 = tv[3];
            const bool more = km != 0;
            if (more) take4();
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, t0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, t1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, t2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, t3, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (!more) break;
          }
          // one atomic per element of the 32 x 32 result; the 16 column offsets of this lane first, as four 16-byte reads
          int4 po[4];
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) po[g4] = *reinterpret_cast<const int4*>(&s_pix[mt * 32 + 8 * g4 + 4 * hf]);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int col = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
            const int4 p4 = po[r >> 2];
            const int pix = (r & 3) == 0 ? p4.x : (r & 3) == 1 ? p4.y : (r & 3) == 2 ? p4.z : p4.w;
            if (col < ncols && acc[r] != 0.f) unsafeAtomicAdd(reinterpret_cast<float*>(gvbase + (unsigned)pix) + li, acc[r]);
          }
        }
      }
    }
    cur = nxt;
    par ^= 1;
  }
}

}  // namespace

// Eight zero-initialised-by-the-caller work counters for one launch pair: the next slot of the ring on the current device.
unsigned* egtr_msda_tile_counters(hipStream_t st) {
  static unsigned* base[64] = {};
  static unsigned next_slot = 0, next_graph_slot = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (base[dev] == nullptr) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_tile_counters)) != hipSuccess) return nullptr;
    base[dev] = static_cast<unsigned*>(p);
  }
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) == hipSuccess && cap == hipStreamCaptureStatusActive) {
    const unsigned g = __atomic_fetch_add(&next_graph_slot, 1u, __ATOMIC_RELAXED);
    if (g >= (unsigned)kGraphSlots) return nullptr;   // ~1000 captured train steps: the caller reports EGTR_E_LAUNCH
    return base[dev] + (kCounterSlots + g) * 8;
  }
  const unsigned slot = __atomic_fetch_add(&next_slot, 1u, __ATOMIC_RELAXED) % kCounterSlots;
  return base[dev] + slot * 8;
}

// Launcher used by egtr_msda_backward_f32 (msda.hip): grad_value of encoder-shaped calls (M = 8, D = 32, L*P = 16).
int egtr_launch_msda_bwd_value_tile_f32(hipStream_t st, const void* grad_out, bool grad_out_bf16, const int64_t* shapes,
                                        const int64_t* lsi, const float* loc, const float* attn, float* grad_value,
                                        int B, int Lq, int S, int L, int P, unsigned* counters) {
  if (grad_out_bf16)
    hipLaunchKernelGGL(msda_bwd_value_tile_f32<true>, dim3(kBwdGrid), dim3(kThreads), 0, st, grad_out, shapes, lsi, loc, attn,
                       grad_value, B, Lq, S, L, P, counters);
  else
    hipLaunchKernelGGL(msda_bwd_value_tile_f32<false>, dim3(kBwdGrid), dim3(kThreads), 0, st, grad_out, shapes, lsi, loc, attn,
                       grad_value, B, Lq, S, L, P, counters);
  return egtr_check_launch();
}
