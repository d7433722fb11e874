// The tail of a ResNet bottleneck on channels-last fp32 data as ONE kernel (inference, frozen batch norm folded into the weights):
//     z = relu( relu(a + shift2) . W3^T + shift3 + shortcut )
// a = the raw output of the 3x3 convolution [M = B H W, K = planes], W3 the 1x1 convolution [N = 4 planes, K], shortcut the
// block input (or the downsample branch) [M, N]  (reference: model/deformable_detr.py:735-760 -- the timm ResNet-50 backbone
// with frozen batch norm; a bottleneck ends conv3 -> bn3 -> += shortcut -> relu).  Before: an in-place shift + ReLU pass over a,
// a vendor fp32 GEMM, a shift + shortcut + ReLU pass over z -- three launches and two extra round trips of the activation
// through memory per block, 21-43 us per block at 600 x 1000 (tools/conv3_fused_ab.py).
//
// Arithmetic: the six-term split-bf16 product of the other x6 kernels (x6_common.h): fp32 operands, fp32 accumulation, the
// error of an fp32 GEMM.
//
// Shape of the work: K is SMALL (64 .. 512) and M x N is everything, so a workgroup keeps its whole BM x K activation panel
// resident and nothing is staged per K slice:
//   * 4 waves; the panel (BM = 64 or 32 rows) is loaded once, shifted, rectified, split into its three bf16 pieces and parked in
//     LDS as [piece][row][K + 8] (rows 2K + 16 bytes apart: the 16-byte operand reads of 32 consecutive rows hit distinct
//     banks, the 8-byte stores of a row are contiguous);
//   * a wave owns 32 NTW output columns for all BM rows.  Its weight fragments never touch LDS: W3 arrives pre-split in the XS
//     format (xs_format.h: 1 KiB MFMA-operand fragments, fragment (n / 32, k / 16, piece)), one 16-byte load per lane and
//     fragment straight into registers, two k-steps ahead of the products that consume them (plain loads: the compiler
//     counts vmcnt).  All workgroups of a column block read the same K x BN x 6 bytes: L2 hits.  The workgroup -> tile order
//     (xcd_tile: every XCD gets a contiguous range) is chosen on the host: column blocks of a row panel side by side when the
//     activation (4 M K bytes) is the larger operand, row panels of a column block side by side when the split weights
//     (6 N K bytes: 6 MiB at N = 2048, K = 512 -- more than an XCD's L2) are;
//   * MFMA roles: A operand = activation piece (i = row), B operand = weight piece (j = column) -- the other way round than in
//     gemm_split / ffn_x6 -- so that a lane holds ONE output column and the 32 lanes of a half wave 128 consecutive bytes of a
//     row: the shortcut reads and the stores of the epilogue (+ shift3 + shortcut, ReLU) are whole cache lines per
//     instruction.  (With 4 consecutive columns per lane a float4 instruction touched 32 rows x 32 bytes.)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"
#include "x6_common.h"
#include "xs_format.h"

namespace {
using namespace x6;
typedef float f32x4v __attribute__((ext_vector_type(4)));

struct TailArgs {
  const float* a;         // [M, lda]
  const float* a_shift;   // [K] or null
  const char* w;          // XS(W [N, K]), pieces rounded to nearest
  const float* bias;      // [N] or null
  const float* shortcut;  // [M, lds] or null
  float* y;               // [M, ldy]
  int M, N, lda, ldsc, ldy;
  int relu_in, relu_out;
  int row_major;   // tile order: 1 = the column blocks of a row panel are neighbours, 0 = the row panels of a column block
};

__device__ __forceinline__ int xcd_tile(int bid, int total) {
  const int q = total >> 3, r = total & 7, x = bid & 7;
  return x * q + min(x, r) + (bid >> 3);
}

#ifdef EGTR_TAIL_TIMING
// debugging aid (tools/tail_timing.py): per workgroup of the LAST launch (no atomics: they would be the bottleneck), the
// real-time counter (100 MHz) at entry and exit and the shader-clock stamps of wave 0 between the phases
constexpr int kTailRecWg = 4096;
__device__ unsigned long long g_tail_rec[kTailRecWg][8];
#define TAIL_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#else
#define TAIL_T(v) do { } while (0)
#endif

template <int BM, int NTW, int KS, int NW = 4>
__global__ __launch_bounds__(64 * NW) void conv_tail_x6_kernel(TailArgs A) {
  constexpr int K = 16 * KS;
  constexpr int MT = BM / 32;            // 32-row MFMA tiles of the panel
  constexpr int BN = 32 * NTW * NW;      // output columns per workgroup
  constexpr int NT = 64 * NW;            // threads
  constexpr int kPitch = K + 8;          // bf16 elements per LDS row
  constexpr int C4 = K / 4;              // float4 per panel row
  constexpr int NQ = BM * C4 / NT;       // float4 per thread
  // weight fragments are requested PF k-steps ahead (K = 64: all of them up front).  Measured and not kept (600 x 1000, inside
  // the forward): PF = 6-8 for K >= 256 together with one LDS-DMA touch per 128-byte line of the wave's weight block at kernel
  // start (all first-touch misses in flight at once): 16.5 -> 19.6 us (K = 256), 20.4 -> 28.3 us (K = 512) per launch.
#ifndef EGTR_TAIL_PF_LONG
#define EGTR_TAIL_PF_LONG 3
#endif
  constexpr int PF = KS <= 4 ? 4 : EGTR_TAIL_PF_LONG;
  static_assert(NT % C4 == 0 && NQ >= 1, "a thread keeps one column group of the panel");
  extern __shared__ __attribute__((aligned(16))) char s_raw[];
  __bf16* const sA = reinterpret_cast<__bf16*>(s_raw);   // [3][BM][kPitch]

  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, hf = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef EGTR_TAIL_TIMING
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  TAIL_T(t0);
  const int mblocks = (A.M + BM - 1) / BM;
  const int tile = xcd_tile(blockIdx.x, gridDim.x);
  // neighbours in the tile order run side by side on one XCD and share its L2: whichever operand is larger is the one
  // that must not be fetched once per partner (the host decides: panel bytes 4 M K against weight bytes 6 N K)
  const int nblocks = A.N / BN;
  const int nb = A.row_major ? tile % nblocks : tile / mblocks;
  const int m0 = (A.row_major ? tile / nblocks : tile - nb * mblocks) * BM;
  const int nt0 = nb * (BN / 32) + wave * NTW;           // first 32-column tile of this wave

  // Requests in the order the data is needed -- a wave's loads return in order, so whatever is requested first is what the
  // first wait waits for: the panel rows (thread t owns column group c4 = t % C4 of rows t / C4 + (NT / C4) q), then the weight
  // fragments of the first k-steps, then shift3 and the shortcut values (needed last, after the products).
  const int c4 = tid % C4, r0 = tid / C4;
  constexpr int CH = NQ < 8 ? NQ : 8;   // panel loads in flight per thread
  f32x4v v[CH];
  auto request_panel = [&](int q0) {
#pragma unroll
    for (int q = 0; q < CH; ++q) {
      const int row = min(m0 + r0 + (NT / C4) * (q0 + q), A.M - 1);
      v[q] = *reinterpret_cast<const f32x4v*>(A.a + (size_t)row * A.lda + 4 * c4);
    }
  };
  request_panel(0);
  f32x4v sh = {0.f, 0.f, 0.f, 0.f};
  if (A.a_shift != nullptr) sh = *reinterpret_cast<const f32x4v*>(A.a_shift + 4 * c4);
  __builtin_amdgcn_sched_barrier(0);   // (keeps this order: the scheduler would sort the requests by address readiness)

  // weight fragment (n tile, k-step, piece) of this lane
  const char* const wlane = A.w + (size_t)nt0 * KS * (3 * xs::kFragBytes) + lane * 16;
  bf16x8 w[PF + 1][NTW][3];
  auto load_w = [&](int ks, bf16x8 (&dst)[NTW][3]) {
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        dst[t][p] = *reinterpret_cast<const bf16x8*>(wlane + ((size_t)(t * KS + ks) * 3 + p) * xs::kFragBytes);
  };
  static_for<PF>([&](auto i_) {
    constexpr int i = decltype(i_)::value;
    if constexpr (i < KS) load_w(i, w[i]);
  });
  float bz[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) bz[t] = A.bias != nullptr ? A.bias[(nt0 + t) * 32 + li] : 0.f;
  // the shortcut values of this lane's outputs, requested now (D[i = m][j = n]: lane l holds column n = l & 31 of a 32-wide
  // tile, accumulator r row (r & 3) + 8 (r >> 2) + 4 (l >> 5)): they arrive under the panel build and the products
  float sc[MT][NTW][16];
  if (A.shortcut != nullptr) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = min(m0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf, A.M - 1);
          sc[m][t][r] = A.shortcut[(size_t)row * A.ldsc + (nt0 + t) * 32 + li];
        }
  } else {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[m][t][r] = 0.f;
  }
  __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise sinks the requests to their first use: no prefetch left)

  // the panel: shift, ReLU, split, park
  {
    const bool relu_in = A.relu_in != 0;
#pragma unroll
    for (int q0 = 0; q0 < NQ; q0 += CH) {
      if (q0 > 0) request_panel(q0);
#pragma unroll
      for (int q = 0; q < CH; ++q) {
        f32x4v t = v[q] + sh;
        if (relu_in) t = f32x4v{egtr_relu(t.x), egtr_relu(t.y), egtr_relu(t.z), egtr_relu(t.w)};
        const xs::Split3 s0 = xs::split3_fast(t.x), s1 = xs::split3_fast(t.y), s2 = xs::split3_fast(t.z),
                         s3 = xs::split3_fast(t.w);
        __bf16* p = sA + (r0 + (NT / C4) * (q0 + q)) * kPitch + 4 * c4;
        *reinterpret_cast<uint2*>(p) = make_uint2(xs::pack_hi16(s0.hi, s1.hi), xs::pack_hi16(s2.hi, s3.hi));
        *reinterpret_cast<uint2*>(p + BM * kPitch) = make_uint2(xs::pack_hi16(s0.mid, s1.mid), xs::pack_hi16(s2.mid, s3.mid));
        *reinterpret_cast<uint2*>(p + 2 * BM * kPitch) = make_uint2(xs::pack_hi16(s0.lo, s1.lo), xs::pack_hi16(s2.lo, s3.lo));
      }
    }
  }
  TAIL_T(t1);
  __syncthreads();
  TAIL_T(t2);

  f32x16 acc[MT][NTW];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;

  // operand fragments of the panel one k-step ahead of their products, weight fragments PF steps ahead; the scheduling
  // barrier keeps the requests of a step in front of its products
  const __bf16* const pa0 = sA + li * kPitch + 8 * hf;
  bf16x8 a[2][MT][3];
  auto read_a = [&](int ks, bf16x8 (&dst)[MT][3]) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        dst[m][p] = *reinterpret_cast<const bf16x8*>(pa0 + (p * BM + m * 32) * kPitch + 16 * ks);
  };
  read_a(0, a[0]);
  static_for<KS>([&](auto ks_) {
    constexpr int ks = decltype(ks_)::value;
    if constexpr (ks + PF < KS) load_w(ks + PF, w[(ks + PF) % (PF + 1)]);
    if constexpr (ks + 1 < KS) read_a(ks + 1, a[(ks + 1) & 1]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int t = 0; t < NTW; ++t) acc[m][t] = mfma6(a[ks & 1][m], w[ks % (PF + 1)][t], acc[m][t]);
    __builtin_amdgcn_sched_barrier(0);
  });

  TAIL_T(t3);
  // epilogue: the 32 lanes of a half wave write 128 CONSECUTIVE bytes of one row per instruction
  const bool relu_out = A.relu_out != 0;
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const int col = (nt0 + t) * 32 + li;
    const float b = bz[t];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
        float v = acc[m][t][r] + b + sc[m][t][r];
        if (relu_out) v = egtr_relu(v);
        if (row < A.M) A.y[(size_t)row * A.ldy + col] = v;
      }
    }
  }
#ifdef EGTR_TAIL_TIMING
  TAIL_T(t4);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TAIL_T(t5);
  if (tid == 0 && blockIdx.x < kTailRecWg) {
    unsigned long long* r = g_tail_rec[blockIdx.x];
    r[0] = rt0;
    r[1] = __builtin_amdgcn_s_memrealtime();
    r[2] = t1 - t0;   // requests + panel build (the loads' latency is waited for here)
    r[3] = t2 - t1;   // barrier
    r[4] = t3 - t2;   // products
    r[5] = t4 - t3;   // epilogue up to the last store instruction
    r[6] = t5 - t4;   // the stores' completion
    r[7] = 1;
  }
#endif
}

template <int BM, int NTW, int KS, int NW = 4>
int launch(hipStream_t st, const TailArgs& A) {
  static unsigned long long raised = 0;
  constexpr int lds = 3 * BM * (16 * KS + 8) * 2;
  auto kern = conv_tail_x6_kernel<BM, NTW, KS, NW>;
  if (lds > 64 * 1024) {
    const int rc = egtr_raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds, &raised);
    if (rc != EGTR_OK) return rc;
  }
  const long long tiles = (long long)((A.M + BM - 1) / BM) * (A.N / (32 * NTW * NW));
  if (tiles >= (1ll << 31)) return EGTR_E_UNSUPPORTED;
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(64 * NW), lds, st, A);
  return egtr_check_launch();
}

// tile choice (tools/conv3_fused_ab.py, 600 x 1000, stand-alone and inside the forward): 32 x 128 tiles -- many small
// workgroups, four to five resident per CU at different phases -- beat 64-row panels and 256-column blocks at every layer
// but the last, where K = 512 leaves one workgroup per CU (100 KiB of panel) and the wider block halves the panel builds
template <int KS>
int dispatch(hipStream_t st, const TailArgs& A, int force_bm, int force_ntw) {
  if (A.N % 128 != 0) {   // N % 64 == 0 (a bottleneck's first 1x1 convolution at 64 planes): two waves x 32 columns
    if (force_bm == 64 || force_ntw == 2) return EGTR_E_UNSUPPORTED;
    return launch<32, 1, KS, 2>(st, A);
  }
  const bool n256 = A.N % 256 == 0;
  int bm = 32;
  int ntw = (KS >= 32 && n256) ? 2 : 1;
  if (force_bm) bm = force_bm;
  if (force_ntw) ntw = force_ntw;
  if (ntw == 2 && !n256) return EGTR_E_UNSUPPORTED;
  if (bm == 64 && KS > 16) return EGTR_E_UNSUPPORTED;   // 3 x 64 x (K + 8) x 2 bytes must fit the LDS
  if constexpr (KS <= 16) {
    if (bm == 64) return ntw == 2 ? launch<64, 2, KS>(st, A) : launch<64, 1, KS>(st, A);
  }
  if constexpr (KS >= 32) {
    // K = 512: the panel fills the LDS (one workgroup per CU), so the 256-column block is 8 waves x 32 columns -- twice the
    // waves streaming weight fragments and building the panel (16.6 -> 14.5 us per launch at M = 608)
    if (bm == 32 && ntw == 2) return launch<32, 1, KS, 8>(st, A);
  }
  if (bm == 32) return ntw == 2 ? launch<32, 2, KS>(st, A) : launch<32, 1, KS>(st, A);
  return EGTR_E_ARG;
}

}  // namespace

#ifdef EGTR_TAIL_TIMING
extern "C" int egtr_conv_tail_stamps(unsigned long long* host_out, int reset) {   // host_out: kTailRecWg x 8
  if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_tail_rec), sizeof(unsigned long long) * kTailRecWg * 8) != hipSuccess)
    return EGTR_E_LAUNCH;
  if (reset) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_tail_rec)) != hipSuccess ||
        hipMemset(p, 0, sizeof(unsigned long long) * kTailRecWg * 8) != hipSuccess)
      return EGTR_E_LAUNCH;
  }
  return EGTR_OK;
}
#endif

extern "C" int egtr_conv1x1_tail_x6_f32(egtr_stream_t stream, const float* a, int lda, const float* a_shift, int relu_in,
                                        const void* w_xs, const float* bias, const float* shortcut, int ld_shortcut,
                                        int relu_out, float* y, int ldy, int M, int K, int N, int tile_rows, int tile_cols) {
  if (!a || !w_xs || !y || M <= 0 || K <= 0 || N <= 0 || lda < K || ldy < N || (shortcut && ld_shortcut < N)) return EGTR_E_ARG;
  if ((tile_rows != 0 && tile_rows != 32 && tile_rows != 64) || (tile_cols != 0 && tile_cols != 128 && tile_cols != 256))
    return EGTR_E_ARG;
  if ((K != 64 && K != 128 && K != 256 && K != 512) || N % 64 || (lda & 3) || (ldy & 3) || (shortcut && (ld_shortcut & 3)) ||
      (reinterpret_cast<uintptr_t>(a) & 15) || (reinterpret_cast<uintptr_t>(y) & 15) || (reinterpret_cast<uintptr_t>(w_xs) & 15) ||
      (reinterpret_cast<uintptr_t>(a_shift) & 15) || (reinterpret_cast<uintptr_t>(bias) & 15) ||
      (reinterpret_cast<uintptr_t>(shortcut) & 15))
    return EGTR_E_UNSUPPORTED;
  TailArgs A{a, a_shift, static_cast<const char*>(w_xs), bias, shortcut, y, M, N, lda, ld_shortcut, ldy, relu_in, relu_out,
             (long long)M * K * 4 >= (long long)N * K * 6 ? 1 : 0};
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int fn = tile_cols / 128;
  switch (K) {
    case 64: return dispatch<4>(st, A, tile_rows, fn);
    case 128: return dispatch<8>(st, A, tile_rows, fn);
    case 256: return dispatch<16>(st, A, tile_rows, fn);
    default: return dispatch<32>(st, A, tile_rows, fn);
  }
}
