// Multi-scale deformable attention (MSDA) for gfx950 / CDNA4 -- forward and backward.
//
// What it computes (reference: model/custom_kernel/cuda/ms_deform_im2col_cuda.cuh, formulas restated in
// SURVEY.md Appendix B):
//   out[b,q,m,:] = sum_{l<L} sum_{p<P} attn[b,q,m,l,p] * bilinear(value_l[b,:,m,:], loc[b,q,m,l,p]*(W_l,H_l) - 0.5)
// with zero contribution from samples outside (-1,H)x(-1,W) and from individual out-of-range corners.
//
// Design (NOT the reference's one-thread-per-output-element / D-thread-block scheme):
//   * one 64-lane wavefront owns one query: lane = (head m = lane>>3, channel quad c4 = lane&7), so a group of
//     8 lanes reads one aligned 128-byte line (32 fp32 channels of one head of one pixel) per bilinear corner
//     with a single global_load_dwordx4, and the 8 groups = the 8 heads.  (M = 8, D = 32, L*P = 16.)
//   * the query's 256 location floats and 128 attention weights are fetched once, coalesced (1 KiB + 512 B per
//     wave): lane i computes the sample geometry of head i>>3, samples 2*(i&7) and 2*(i&7)+1, and stages
//     per-sample records {4 clamped corner byte-offsets, 4 (bilinear x attention) weights} in LDS;
//     the gather loop then reads each record with two broadcast ds_read_b128 per sample.
//   * workgroup -> query mapping is XCD-aware: consecutive blockIdx values round-robin over the 8 XCDs, so
//     block b is given the (b%8)-th contiguous chunk of queries; each XCD's private 4 MiB L2 then serves one
//     horizontal stripe of every feature level instead of the whole 12.8 MB value tensor.
//   * backward: same mapping; grad_attn / grad_loc are reduced over the 8 lanes of a head with DPP
//     (quad_perm / row_half_mirror) instead of the reference's shared-memory serial reduce (cuh:376-393);
//     grad_value uses hardware fp32 atomics (global_atomic_add_f32) like the reference's atomicAdd.
//   * any other (M, D, L, P) goes through a simple generic kernel (one thread per output element).
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdlib.h>

#include "common.h"
#include "msda_common.h"

namespace {

using namespace egtr_msda;

// LDS record layout: [wave][head][sample] 16-byte entries, head stride padded by one entry so that the four
// heads served together by one ds_read_b128 lane group land on distinct banks (MI355X_MICROARCH.md, LDS).
constexpr int kMaxBitWords = 1024;              // padding-mask bits kept in LDS by the fused forward (S <= 32768)
constexpr int kHeadStride = 17;                 // entries (16 samples + 1 pad)
constexpr int kWaveEntries = 8 * kHeadStride;   // per array per wave

// ------------------------------------------------------------------------------------------------ forward
// fp32, M = 8, D = 32, L*P = 16 (L <= 4).  One wave per query, kWaves queries per workgroup.
//
// FUSED: `loc` holds the raw sampling offsets (output of the sampling_offsets Linear, same [.., M, L, P, 2] layout) and
// `attn` the raw attention logits; the kernel itself forms  loc = ref[q, l, :] + offset / (W_l, H_l)  and the softmax
// over the L*P logits of a head (deformable_detr.py:1055-1073) -- the 16 logits of a head live in 8 adjacent lanes,
// reduced with DPP -- instead of four elementwise launches and a 19 MB round trip per encoder layer.  `attn_out`
// (optional) receives the softmaxed weights.
// SPLIT: short query lists (decoder: 200 queries = 200 waves on 1024 SIMDs, one dependent chain of 4 x 16 gathers each):
// one query per workgroup, wave w gathers samples 4w .. 4w+3 (16 loads in flight at once) and the four partial sums
// are added in sample order through LDS.
constexpr long long kSplitMaxQueries = 1024;
// BOX (FUSED only): `ref` holds 4-d reference boxes [nq, L, 4] = (cx, cy, w, h) and the locations are
// ref.xy + offset / P * ref.wh * 0.5 (deformable_detr.py:1074-1081, iterative box refinement / two-stage heads).
template <bool FUSED, bool SPLIT = false, bool BOX = false>
__global__ __launch_bounds__(kWaves * 64) void msda_fwd_q64_f32(
    const float* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
    const float* __restrict__ loc, const float* __restrict__ attn, float* __restrict__ out, int nq_total,
    int Lq, int S, int L, int P, int nblk, const float* __restrict__ ref, float* __restrict__ attn_out, int ld_off,
    int ld_logit, const unsigned char* __restrict__ keep, const unsigned* __restrict__ keep_bits,
    const float* __restrict__ vbias) {
  __shared__ __attribute__((aligned(16))) int4 s_off[kWaves * kWaveEntries];
  __shared__ __attribute__((aligned(16))) float4 s_w[kWaves * kWaveEntries];
  __shared__ unsigned s_bits[FUSED ? kMaxBitWords : 1];
  // (the wave index as a scalar: the query, its image, its mask row and the branches on them become wave-uniform)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int blk = xcd_remap(blockIdx.x, nblk);
  const int q = SPLIT ? blk : blk * kWaves + wave;
  const int nwords = (S + 31) >> 5;
  bool bits_in_lds = false;
  if (FUSED && keep_bits != nullptr && nwords <= kMaxBitWords) {
    // padding mask of the image of this workgroup's first query, one bit per token (1.5 KB at 600x1000), in LDS
    const int b0 = min(SPLIT ? blk : blk * kWaves, nq_total - 1) / Lq;
    for (int i = threadIdx.x; i < nwords; i += kWaves * 64) s_bits[i] = keep_bits[(size_t)b0 * nwords + i];
    __syncthreads();
    bits_in_lds = q < nq_total && q / Lq == b0;
  }
  if (q >= nq_total) return;  // wave-uniform; no workgroup barrier below
  LevelGeom G;
  load_geom(shapes, lsi, L, G);
  const int b = q / Lq;
  const char* vbase = reinterpret_cast<const char*>(value) + (size_t)b * S * (256 * 4);

  // stage 1: lane i -> head i>>3, samples 2*(i&7), 2*(i&7)+1
  // FUSED: offsets / logits may be column blocks of one wider Linear output (row strides ld_off / ld_logit floats)
  float4 lc = reinterpret_cast<const float4*>(loc + (size_t)q * (FUSED ? ld_off : 256))[lane];
  float2 aw = reinterpret_cast<const float2*>(attn + (size_t)q * (FUSED ? ld_logit : 128))[lane];
  const int head_s = lane >> 3, s0 = (lane & 7) * 2;
  if (FUSED) {
    const int lvl = s0 / P;  // both samples of the lane lie in one level (P is even)
    if (BOX) {
      const float4 r = *reinterpret_cast<const float4*>(ref + ((size_t)q * L + lvl) * 4);
      const float fp = (float)P;  // same operation order as the reference: ((offset / P) * wh) * 0.5
      lc = make_float4(r.x + lc.x / fp * r.z * 0.5f, r.y + lc.y / fp * r.w * 0.5f, r.x + lc.z / fp * r.z * 0.5f,
                       r.y + lc.w / fp * r.w * 0.5f);
    } else {
      const float2 r = *reinterpret_cast<const float2*>(ref + ((size_t)q * L + lvl) * 2);
      // round 6: reciprocal multiplies and the hardware exponential below (as the bf16 kernel since round 5).  The counters
      // (profiles/r06_sq_pmc.txt) put this kernel at 809 VALU instructions per query of which the gather loop is 180: four
      // IEEE divisions, two expf and two more divisions per lane were ~150 of the rest.  1-2 ulp of fp32 against the
      // reference's `offset / normalizer` and softmax: 1e-7 pixels / 1e-7 of a weight.
      const float iw = __frcp_rn((float)SEL_W(G, lvl)), ih = __frcp_rn((float)SEL_H(G, lvl));
      lc = make_float4(r.x + lc.x * iw, r.y + lc.y * ih, r.x + lc.z * iw, r.y + lc.w * ih);
    }
    // softmax over the 16 logits of the head: 8 lanes x 2
    float m = fmaxf(aw.x, aw.y);
    m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0xB1, 0xf, 0xf, false)));
    m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x4E, 0xf, 0xf, false)));
    m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x141, 0xf, 0xf, false)));
    const float e0 = __expf(aw.x - m), e1 = __expf(aw.y - m);
    float sum = e0 + e1;
    sum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum), 0xB1, 0xf, 0xf, false));
    sum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum), 0x4E, 0xf, 0xf, false));
    sum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum), 0x141, 0xf, 0xf, false));
    const float inv_sum = __frcp_rn(sum);
    aw = make_float2(e0 * inv_sum, e1 * inv_sum);
    if (attn_out != nullptr && (!SPLIT || wave == 0)) reinterpret_cast<float2*>(attn_out + (size_t)q * 128)[lane] = aw;
  }
  int4* my_off = s_off + wave * kWaveEntries;
  float4* my_w = s_w + wave * kWaveEntries;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int s = s0 + j;
    const int lvl = s / P;
    const SampleGeom g = sample_geom<1024, 128>(j ? lc.z : lc.x, j ? lc.w : lc.y, SEL_H(G, lvl), SEL_W(G, lvl),
                                                SEL_S(G, lvl), head_s);
    const float a = j ? aw.y : aw.x;
    bool k0 = g.ok[0], k1 = g.ok[1], k2 = g.ok[2], k3 = g.ok[3];
    // padded tokens contribute nothing (deformable_detr.py:1050-1052 zeroes their value rows; zeroing their weights
    // is the same sum and saves a pass over the value tensor)
    if (FUSED && keep_bits != nullptr) {
      const int p0 = g.off[0] >> 10, p1 = g.off[1] >> 10, p2 = g.off[2] >> 10, p3 = g.off[3] >> 10;
      // (bitwise, not short-circuit: the clamped offsets are always readable, and four independent reads beat four branches)
      unsigned w0, w1, w2, w3;
      if (bits_in_lds) {
        w0 = s_bits[p0 >> 5]; w1 = s_bits[p1 >> 5]; w2 = s_bits[p2 >> 5]; w3 = s_bits[p3 >> 5];
        // (keeps the optimiser from sinking the two branches' reads into one FLAT load of a selected pointer)
        asm volatile("" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3));
      } else {
        const unsigned* kb = keep_bits + (size_t)b * nwords;
        w0 = kb[p0 >> 5]; w1 = kb[p1 >> 5]; w2 = kb[p2 >> 5]; w3 = kb[p3 >> 5];
      }
      k0 = k0 & (bool)((w0 >> (p0 & 31)) & 1u);
      k1 = k1 & (bool)((w1 >> (p1 & 31)) & 1u);
      k2 = k2 & (bool)((w2 >> (p2 & 31)) & 1u);
      k3 = k3 & (bool)((w3 >> (p3 & 31)) & 1u);
    } else if (FUSED && keep != nullptr) {
      const unsigned char* kp = keep + (size_t)b * S;
      const unsigned char c0 = kp[g.off[0] >> 10], c1 = kp[g.off[1] >> 10], c2 = kp[g.off[2] >> 10], c3 = kp[g.off[3] >> 10];
      k0 = k0 & (c0 != 0);
      k1 = k1 & (c1 != 0);
      k2 = k2 & (c2 != 0);
      k3 = k3 & (c3 != 0);
    }
    my_off[head_s * kHeadStride + s] = make_int4(g.off[0], g.off[1], g.off[2], g.off[3]);
    my_w[head_s * kHeadStride + s] = make_float4(k0 ? g.w[0] * a : 0.f, k1 ? g.w[1] * a : 0.f,
                                                 k2 ? g.w[2] * a : 0.f, k3 ? g.w[3] * a : 0.f);
  }
  // LDS ops of one wave execute in order; the fences only stop the compiler from reordering across lanes.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // stage 2: lane -> (head = lane>>3, channel quad = lane&7)
  const int head = lane >> 3;
  const char* vlane = vbase + (lane & 7) * 16;
  const int4* ro = my_off + head * kHeadStride;
  const float4* rw = my_w + head * kHeadStride;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  // vbias (optional, [256]): `value` is the bias-free projection W x and the bias b of value_proj is applied here:
  //   sum_s w_s (v_s + b) = sum_s w_s v_s + b sum_s w_s   with w_s the in-range, unpadded corner weights
  // (dd:1048-1052: padded rows of value are zero, not b; out-of-range corners contribute nothing).
  float wsum = 0.f;
  if (SPLIT) {
    __shared__ float4 s_part[kWaves - 1][64];
#pragma unroll
    for (int i = 0; i < 16 / kWaves; ++i) {
      const int s = wave * (16 / kWaves) + i;
      const int4 o = ro[s];
      const float4 w = rw[s];
      wsum += (w.x + w.y) + (w.z + w.w);
      const float4 v0 = *reinterpret_cast<const float4*>(vlane + (unsigned)o.x);
      const float4 v1 = *reinterpret_cast<const float4*>(vlane + (unsigned)o.y);
      const float4 v2 = *reinterpret_cast<const float4*>(vlane + (unsigned)o.z);
      const float4 v3 = *reinterpret_cast<const float4*>(vlane + (unsigned)o.w);
      acc.x += w.x * v0.x + w.y * v1.x + w.z * v2.x + w.w * v3.x;
      acc.y += w.x * v0.y + w.y * v1.y + w.z * v2.y + w.w * v3.y;
      acc.z += w.x * v0.z + w.y * v1.z + w.z * v2.z + w.w * v3.z;
      acc.w += w.x * v0.w + w.y * v1.w + w.z * v2.w + w.w * v3.w;
    }
    if (vbias != nullptr) {
      const float4 vb = reinterpret_cast<const float4*>(vbias)[lane];
      acc.x += vb.x * wsum;
      acc.y += vb.y * wsum;
      acc.z += vb.z * wsum;
      acc.w += vb.w * wsum;
    }
    if (wave > 0) s_part[wave - 1][lane] = acc;
    __syncthreads();  // q is the same for the four waves: nobody has returned
    if (wave == 0) {
#pragma unroll
      for (int w = 0; w < kWaves - 1; ++w) {
        const float4 p = s_part[w][lane];
        acc.x += p.x;
        acc.y += p.y;
        acc.z += p.z;
        acc.w += p.w;
      }
      reinterpret_cast<float4*>(out + (size_t)q * 256)[lane] = acc;
    }
    return;
  }
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
  if (vbias != nullptr) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float4 w = rw[s];
      wsum += (w.x + w.y) + (w.z + w.w);
    }
    const float4 vb = reinterpret_cast<const float4*>(vbias)[lane];
    acc.x += vb.x * wsum;
    acc.y += vb.y * wsum;
    acc.z += vb.z * wsum;
    acc.w += vb.w * wsum;
  }
  reinterpret_cast<float4*>(out + (size_t)q * 256)[lane] = acc;
}

// bf16 storage, fp32 accumulate; M = 8, D = 32, L*P = 16: a head row is 64 bytes = 4 lanes x 16 B, so one wave
// owns TWO queries (lanes 0-31 / 32-63), lane = (query half, head = (lane>>2)&7, channel octet = lane&3).
__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  // round-to-nearest-even (inputs are finite sums)
  unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
  ua += 0x7fffu + ((ua >> 16) & 1u);
  ub += 0x7fffu + ((ub >> 16) & 1u);
  return (ua >> 16) | (ub & 0xffff0000u);
}

// FUSED: `loc` / `attn` hold the raw bf16 outputs of the sampling_offsets / attention_weights Linears (row strides
// ld_off / ld_logit elements) and `ref` the bf16 reference points [nq, L, 2]; softmax and loc = ref + off / (W, H) are
// formed here in fp32 (the PyTorch composition rounds every intermediate to bf16); `keep` (optional, bytes per token):
// padded tokens are skipped == their value rows zeroed (dd:1052).
// PC: the number of points per level as a compile-time constant (4: the model's), 0 = the runtime value -- the level of a sample
// is s / P, a division by a runtime integer otherwise (~20 instructions each, several per record).
template <bool FUSED, int PC>
__global__ __launch_bounds__(kWaves * 64) void msda_fwd_q32_bf16(
    const uint16_t* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
    const void* __restrict__ loc, const void* __restrict__ attn, uint16_t* __restrict__ out, int nq_total,
    int Lq, int S, int L, int P_rt, int nblk, const uint16_t* __restrict__ ref, int ld_off, int ld_logit,
    const unsigned char* __restrict__ keep, const unsigned* __restrict__ keep_bits) {
  const int P = PC ? PC : P_rt;
  __shared__ __attribute__((aligned(16))) int4 s_off[kWaves * 2 * kWaveEntries];
  __shared__ __attribute__((aligned(16))) float4 s_w[kWaves * 2 * kWaveEntries];
  __shared__ unsigned s_bits[FUSED ? kMaxBitWords : 1];
  // (the wave index as a scalar: the query index, the image it belongs to and the mask row derived from it are wave-uniform,
  // and so are the branches on them -- as a vector value they cost ~1100 VALU instructions and 71 exec-mask branches of prologue)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int blk = xcd_remap(blockIdx.x, nblk);
  const int qpair = (blk * kWaves + wave) * 2;
  // padding mask as ONE BIT per token, the image of the workgroup's first query staged in LDS (2.8 KB at 800x1333): the
  // byte-mask form costs eight scattered single-byte loads per lane and query pair -- 40 % of this kernel's L1 accesses
  // (866 vs 512 per query, profiles/r05_msda_bf16_pmc.json before the change) in a kernel that runs at the L1's rate
  const int nwords = (S + 31) >> 5;
  int b0 = 0;
  bool bits_staged = false;
  if (FUSED && keep_bits != nullptr && nwords <= kMaxBitWords) {
    b0 = min(blk * kWaves * 2, nq_total - 1) / Lq;
    for (int i = threadIdx.x; i < nwords; i += kWaves * 64) s_bits[i] = keep_bits[(size_t)b0 * nwords + i];
    __syncthreads();
    bits_staged = true;
  }
  if (qpair >= nq_total) return;
  LevelGeom G;
  load_geom(shapes, lsi, L, G);
  int4* my_off = s_off + wave * 2 * kWaveEntries;
  float4* my_w = s_w + wave * 2 * kWaveEntries;
  // stage 1: two queries x 128 (head,sample) records over 64 lanes = 4 records per lane
#pragma unroll
  for (int qq = 0; qq < 2; ++qq) {
    const int q = min(qpair + qq, nq_total - 1);
    const int head_s = lane >> 3, s0 = (lane & 7) * 2;
    float4 lc;
    float2 aw;
    if (FUSED) {
      const int lvl = s0 / P;
      const uint2 o = reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(loc) + (size_t)q * ld_off)[lane];
      const unsigned lg = reinterpret_cast<const unsigned*>(reinterpret_cast<const uint16_t*>(attn) + (size_t)q * ld_logit)[lane];
      const unsigned rp = reinterpret_cast<const unsigned*>(ref + ((size_t)q * L + lvl) * 2)[0];
      const float rx = bf16_lo(rp), ry = bf16_hi(rp);
      // (reciprocal multiplies and the hardware exponential: this kernel is VALU-bound and its operands are bf16 -- a
      // 1-ulp fp32 difference against the division / expf of the fp32 kernel is three orders below their rounding)
      const float iw = __frcp_rn((float)SEL_W(G, lvl)), ih = __frcp_rn((float)SEL_H(G, lvl));
      lc = make_float4(rx + bf16_lo(o.x) * iw, ry + bf16_hi(o.x) * ih, rx + bf16_lo(o.y) * iw, ry + bf16_hi(o.y) * ih);
      aw = make_float2(bf16_lo(lg), bf16_hi(lg));
      float m = fmaxf(aw.x, aw.y);
      m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0xB1, 0xf, 0xf, false)));
      m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x4E, 0xf, 0xf, false)));
      m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x141, 0xf, 0xf, false)));
      const float e0 = __expf(aw.x - m), e1 = __expf(aw.y - m);
      float sum = e0 + e1;
      sum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum), 0xB1, 0xf, 0xf, false));
      sum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum), 0x4E, 0xf, 0xf, false));
      sum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum), 0x141, 0xf, 0xf, false));
      const float inv = __frcp_rn(sum);
      aw = make_float2(e0 * inv, e1 * inv);
    } else {
      lc = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(loc) + (size_t)q * 256)[lane];
      aw = reinterpret_cast<const float2*>(reinterpret_cast<const float*>(attn) + (size_t)q * 128)[lane];
    }
    const bool use_bits = FUSED && keep_bits != nullptr;
    const unsigned char* kp = (FUSED && keep != nullptr && !use_bits) ? keep + (size_t)(q / Lq) * S : nullptr;
    const bool in_lds = __builtin_amdgcn_readfirstlane((int)(bits_staged && q / Lq == b0)) != 0;   // a scalar branch, two code paths
    const unsigned* kb = use_bits ? keep_bits + (size_t)(q / Lq) * nwords : nullptr;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int s = s0 + j;
      const int lvl = s / P;
      const SampleGeom g = sample_geom<512, 64>(j ? lc.z : lc.x, j ? lc.w : lc.y, SEL_H(G, lvl), SEL_W(G, lvl),
                                                SEL_S(G, lvl), head_s);
      const float a = j ? aw.y : aw.x;
      bool k0 = g.ok[0], k1 = g.ok[1], k2 = g.ok[2], k3 = g.ok[3];
      // (bitwise, not short-circuit: the clamped offsets are always readable, and four independent reads beat four branches)
      if (use_bits) {
        const int p0 = g.off[0] >> 9, p1 = g.off[1] >> 9, p2 = g.off[2] >> 9, p3 = g.off[3] >> 9;
        unsigned w0, w1, w2, w3;
        if (in_lds) {
          w0 = s_bits[p0 >> 5]; w1 = s_bits[p1 >> 5]; w2 = s_bits[p2 >> 5]; w3 = s_bits[p3 >> 5];
          // (keeps the optimiser from sinking the two branches' reads into one FLAT load of a selected pointer: a flat load
          // counts on both wait counters and costs an LDS read the latency of a global one)
          asm volatile("" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3));
        } else {
          w0 = kb[p0 >> 5]; w1 = kb[p1 >> 5]; w2 = kb[p2 >> 5]; w3 = kb[p3 >> 5];
        }
        k0 = k0 & (bool)((w0 >> (p0 & 31)) & 1u);
        k1 = k1 & (bool)((w1 >> (p1 & 31)) & 1u);
        k2 = k2 & (bool)((w2 >> (p2 & 31)) & 1u);
        k3 = k3 & (bool)((w3 >> (p3 & 31)) & 1u);
      } else if (kp != nullptr) {
        const unsigned char c0 = kp[g.off[0] >> 9], c1 = kp[g.off[1] >> 9], c2 = kp[g.off[2] >> 9], c3 = kp[g.off[3] >> 9];
        k0 = k0 & (c0 != 0);
        k1 = k1 & (c1 != 0);
        k2 = k2 & (c2 != 0);
        k3 = k3 & (c3 != 0);
      }
      my_off[qq * kWaveEntries + head_s * kHeadStride + s] = make_int4(g.off[0], g.off[1], g.off[2], g.off[3]);
      my_w[qq * kWaveEntries + head_s * kHeadStride + s] =
          make_float4(k0 ? g.w[0] * a : 0.f, k1 ? g.w[1] * a : 0.f, k2 ? g.w[2] * a : 0.f, k3 ? g.w[3] * a : 0.f);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int half = lane >> 5, head = (lane >> 2) & 7;
  const int q = qpair + half;
  const bool live = q < nq_total;
  const int qc = live ? q : nq_total - 1;
  const int b = qc / Lq;
  // The loop below is VALU-bound as much as L1-bound (round 5: ~95 VALU instructions per sample and lane against 16 loads),
  // so the arithmetic is written for the packed pipe -- a uint of two bf16 channels is unpacked into a float2 and multiplied
  // by the broadcast corner weight with ONE v_pk_fma_f32 -- and the addresses are a wave-uniform base + a 32-bit lane offset
  // (one v_add_u32 per load instead of a 64-bit add; the launcher checks B * S * 512 < 2^32).
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const char* vbase = reinterpret_cast<const char*>(value);
  const unsigned lane_off = (unsigned)b * (unsigned)S * 512u + (lane & 3) * 16;
  const int4* ro = my_off + half * kWaveEntries + head * kHeadStride;
  const float4* rw = my_w + half * kWaveEntries + head * kHeadStride;
  f32x2 acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#ifndef EGTR_MSDA_BF16_PKFMA
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#endif
  // kBatch samples = 4 kBatch corner loads are requested before the first is consumed (the compiler otherwise waits for each
  // sample's four loads right behind their issue)
  constexpr int kBatch = 4;   // 16 loads in flight per lane: the kernel is short of memory-level parallelism (16 waves per CU)
#pragma unroll 1
  for (int s = 0; s < 16; s += kBatch) {
    int4 o[kBatch];
    float4 w[kBatch];
    uint4 v[kBatch][4];
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
      o[u] = ro[s + u];
      w[u] = rw[s + u];
    }
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
      v[u][0] = *reinterpret_cast<const uint4*>(vbase + (lane_off + (unsigned)o[u].x));
      v[u][1] = *reinterpret_cast<const uint4*>(vbase + (lane_off + (unsigned)o[u].y));
      v[u][2] = *reinterpret_cast<const uint4*>(vbase + (lane_off + (unsigned)o[u].z));
      v[u][3] = *reinterpret_cast<const uint4*>(vbase + (lane_off + (unsigned)o[u].w));
    }
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
      const float wc[4] = {w[u].x, w[u].y, w[u].z, w[u].w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const unsigned a[4] = {v[u][c].x, v[u][c].y, v[u][c].z, v[u][c].w};
#ifndef EGTR_MSDA_BF16_PKFMA
        // v_dot2_f32_bf16 on the packed word as it was loaded: the corner weight, rounded to bf16, sits in the low / high half of
        // the other operand (the half it does not occupy is zero): one instruction per channel, no unpacking -- 10 instead of 13
        // VALU instructions per corner (round 5: plain bf16 entry 654 -> 571 us at the stress shape, tools/msda_bf16_dot2_ab.sh;
        // -DEGTR_MSDA_BF16_PKFMA restores v_pk_fma_f32 on unpacked pairs with the fp32 weight)
        bf16x2 wl;
        wl[0] = (__bf16)wc[c];
        wl[1] = (__bf16)0.f;
        const unsigned wlo = __builtin_bit_cast(unsigned, wl), whi = wlo << 16;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          acc[k][0] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a[k]), __builtin_bit_cast(bf16x2, wlo), acc[k][0], false);
          acc[k][1] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a[k]), __builtin_bit_cast(bf16x2, whi), acc[k][1], false);
        }
#else
        const f32x2 wv = {wc[c], wc[c]};
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = __builtin_elementwise_fma(f32x2{bf16_lo(a[k]), bf16_hi(a[k])}, wv, acc[k]);
#endif
      }
    }
  }
  if (live) {
    uint4 r;
    r.x = pack_bf16(acc[0][0], acc[0][1]);
    r.y = pack_bf16(acc[1][0], acc[1][1]);
    r.z = pack_bf16(acc[2][0], acc[2][1]);
    r.w = pack_bf16(acc[3][0], acc[3][1]);
    reinterpret_cast<uint4*>(out + (size_t)q * 256)[lane & 31] = r;
  }
}

// Generic shapes: one thread per output element (b,q,m,c); correctness path for unusual (M, D, L, P).
template <typename T>
__device__ __forceinline__ float ld_elt(const T* p);
template <>
__device__ __forceinline__ float ld_elt<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float ld_elt<uint16_t>(const uint16_t* p) { return __uint_as_float(((unsigned)*p) << 16); }

template <typename T>
__global__ void msda_fwd_generic(const T* __restrict__ value, const int64_t* __restrict__ shapes,
                                     const int64_t* __restrict__ lsi, const T* __restrict__ loc,
                                     const T* __restrict__ attn, T* __restrict__ out, long long n, int S,
                                     int M, int D, int L, int Lq, int P) {
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % D);
    long long t = idx / D;
    const int m = (int)(t % M);
    t /= M;  // t = b*Lq + q
    const int b = (int)(t / Lq);
    const T* vb = value + (size_t)b * S * M * D + m * D + c;
    const T* lp = loc + (size_t)(t * M + m) * L * P * 2;
    const T* ap = attn + (size_t)(t * M + m) * L * P;
    T col = 0;
    for (int l = 0; l < L; ++l) {
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1], st = (int)lsi[l];
      for (int p = 0; p < P; ++p) {
        const T x = lp[(l * P + p) * 2] * W - (T)0.5, y = lp[(l * P + p) * 2 + 1] * H - (T)0.5;
        if (!(y > (T)-1 && x > (T)-1 && y < (T)H && x < (T)W)) continue;
        const T yf = floor(y), xf = floor(x);
        const int y0 = (int)yf, x0 = (int)xf, y1 = y0 + 1, x1 = x0 + 1;
        const T lh = y - yf, lw = x - xf, hh = (T)1 - lh, hw = (T)1 - lw;
        T v1 = 0, v2 = 0, v3 = 0, v4 = 0;
        if (y0 >= 0 && x0 >= 0) v1 = vb[(size_t)(st + y0 * W + x0) * M * D];
        if (y0 >= 0 && x1 <= W - 1) v2 = vb[(size_t)(st + y0 * W + x1) * M * D];
        if (y1 <= H - 1 && x0 >= 0) v3 = vb[(size_t)(st + y1 * W + x0) * M * D];
        if (y1 <= H - 1 && x1 <= W - 1) v4 = vb[(size_t)(st + y1 * W + x1) * M * D];
        const T val = hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4;
        col += val * ap[l * P + p];
      }
    }
    out[idx] = col;
  }
}

// ------------------------------------------------------------------------------------------------ backward
// Sum over the 8 lanes of a head group; every lane of the group ends with the total.
__device__ __forceinline__ float group8_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  return v;
}


// fp32, M = 8, D = 32, L*P = 16.  One wave per query.  Per (head, sample) the wave keeps in LDS the four corner offsets
// and THREE coefficient vectors over the corners, written once by the lane that owns the sample:
//   A = masked bilinear weights, B = attn * W * {-hh, +hh, -lh, +lh} (masked), C = attn * H * {-hw, -lw, +hw, +lw} (masked).
// With d_k = <grad_out[head, :], value[corner k, head, :]> (4 multiply-adds per lane + one 8-lane DPP reduction per
// corner) the three gradients of the sample are  grad_attn = A.d,  grad_loc.x = B.d,  grad_loc.y = C.d  (cuh:108-158 with
// the channel sum taken first): lanes 0 / 1 / 2 of the head's group read A / B / C and form one of them each.  (Round 4.
// Until then every lane formed val / dh / dw per CHANNEL and three reductions followed: ~125 VALU instructions per
// sample made the kernel VALU-bound at 200 us for B = 4; now ~35.)  Entries are swizzled (slot = sample ^ head) so that
// the heads served by one ds_read_b128 pass use different banks without padding: 4 waves x 8 KiB = 32 KiB, five
// workgroups per CU.  A sample's three results overwrite the x components of its own A / B / C (dead by then) and are
// stored coalesced after the loop.
// SPLIT > 1 (short query lists, e.g. the decoder's 200 queries per image): SPLIT waves share one query, each taking
// 16 / SPLIT consecutive samples (= one level for SPLIT = 4), so that a B x 200-query call fills the chip (800 waves of
// serial atomics -> 3200) -- every wave still builds all 16 records (cheap) but gathers / scatters only its own.
// BF: grad_out and value hold raw bfloat16 (the backward of a bf16 model): rows of 512 bytes instead of 1024 -- the record
// offsets, built for the fp32 grad_value rows, are halved for the value gather -- widened on load; everything behind the loads
// is the fp32 kernel (same instructions, same results as widening the operands first).
template <bool VALUE_ATOMICS, int SPLIT = 1, bool BF = false>
__global__ __launch_bounds__(kWaves * 64) void msda_bwd_q64_f32(
    const void* __restrict__ grad_out_, const void* __restrict__ value_, const int64_t* __restrict__ shapes,
    const int64_t* __restrict__ lsi, const float* __restrict__ loc, const float* __restrict__ attn,
    float* __restrict__ grad_value, float* __restrict__ grad_loc, float* __restrict__ grad_attn, int nq_total,
    int Lq, int S, int L, int P, int nblk, unsigned* __restrict__ zero8, int zero_value_rows) {
  __shared__ __attribute__((aligned(16))) int4 s_off[kWaves * 128];
  // the work counters of the value-tile kernel that runs behind this launch (msda_tile.hip) are reset here: no launch of their own
  if (zero8 != nullptr && blockIdx.x == 0 && threadIdx.x < 8) zero8[threadIdx.x] = 0u;
  __shared__ __attribute__((aligned(16))) float4 s_cf[kWaves * 128 * 3];                      // A | B | C per entry
  __shared__ __attribute__((aligned(16))) float4 s_a[VALUE_ATOMICS ? kWaves * 128 : 1];      // {bits, lh, lw, attn}
  __shared__ __attribute__((aligned(16))) float s_go[VALUE_ATOMICS ? kWaves * 256 : 4];      // grad_out row, [head][32]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c4_ = lane & 7;
  const int blk = xcd_remap(blockIdx.x, nblk);
  const int gw = blk * kWaves + wave;
  const int q = gw / SPLIT, part = gw % SPLIT;
  constexpr int NS = 16 / SPLIT;  // samples per wave
  if (q >= nq_total) return;
  LevelGeom G;
  load_geom(shapes, lsi, L, G);
  const int b = q / Lq;
  const size_t boff = (size_t)b * S * (256 * 4);
  const char* vbase = reinterpret_cast<const char*>(value_) + (BF ? boff / 2 : boff);
  char* gvbase = reinterpret_cast<char*>(grad_value) + boff;
  auto widen4 = [](uint2 u) {
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
  };
  auto corner = [&](int off) {   // the lane's 4 channels of one corner
    if constexpr (BF) return widen4(*reinterpret_cast<const uint2*>(vbase + c4_ * 8 + ((unsigned)off >> 1)));
    else return *reinterpret_cast<const float4*>(vbase + c4_ * 16 + (unsigned)off);
  };

  const float4 lc = reinterpret_cast<const float4*>(loc + (size_t)q * 256)[lane];
  const float2 aw = reinterpret_cast<const float2*>(attn + (size_t)q * 128)[lane];
  // encoder-shaped calls without atomics here (Lq == S, the value-tile kernel accumulates grad_value behind this launch): query q
  // clears row q of grad_value, 1 KiB per wave -- the caller's 51 MB zero-fill launch folded into this gather-bound kernel
  if (!VALUE_ATOMICS && zero_value_rows)
    reinterpret_cast<float4*>(gvbase + (size_t)(q - b * Lq) * 1024)[lane] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int head_s = lane >> 3, s0 = (lane & 7) * 2;
  int4* my_off = s_off + wave * 128;
  float4* my_cf = s_cf + wave * 128 * 3;
  float4* my_a = s_a + (VALUE_ATOMICS ? wave * 128 : 0);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int s = s0 + j;
    const int lvl = s / P;
    const int H = SEL_H(G, lvl), W = SEL_W(G, lvl);
    const SampleGeom g = sample_geom<1024, 128>(j ? lc.z : lc.x, j ? lc.w : lc.y, H, W, SEL_S(G, lvl), head_s);
    const int e = head_s * 16 + (s ^ head_s);
    const float a = j ? aw.y : aw.x;
    const float aW = a * (float)W, aH = a * (float)H;  // the "* W" / "* H" of cuh:157-158
    const float hh = 1.f - g.lh, hw = 1.f - g.lw;
    my_off[e] = make_int4(g.off[0], g.off[1], g.off[2], g.off[3]);
    // an out-of-range corner contributes 0 everywhere (cuh:121-150)
    my_cf[e * 3 + 0] = make_float4(g.ok[0] ? g.w[0] : 0.f, g.ok[1] ? g.w[1] : 0.f, g.ok[2] ? g.w[2] : 0.f,
                                   g.ok[3] ? g.w[3] : 0.f);
    my_cf[e * 3 + 1] = make_float4(g.ok[0] ? -hh * aW : 0.f, g.ok[1] ? hh * aW : 0.f, g.ok[2] ? -g.lh * aW : 0.f,
                                   g.ok[3] ? g.lh * aW : 0.f);
    my_cf[e * 3 + 2] = make_float4(g.ok[0] ? -hw * aH : 0.f, g.ok[1] ? -g.lw * aH : 0.f, g.ok[2] ? hw * aH : 0.f,
                                   g.ok[3] ? g.lw * aH : 0.f);
    if (VALUE_ATOMICS) {
      const int bits = (g.ok[0] ? 1 : 0) | (g.ok[1] ? 2 : 0) | (g.ok[2] ? 4 : 0) | (g.ok[3] ? 8 : 0);
      my_a[e] = make_float4(__int_as_float(bits), g.lh, g.lw, a);
    }
  }
  const int head = lane >> 3, c4 = lane & 7;
  float4 g;
  if constexpr (BF) g = widen4(reinterpret_cast<const uint2*>(static_cast<const uint16_t*>(grad_out_) + (size_t)q * 256)[lane]);
  else g = reinterpret_cast<const float4*>(static_cast<const float*>(grad_out_) + (size_t)q * 256)[lane];
  if (VALUE_ATOMICS) reinterpret_cast<float4*>(s_go + wave * 256)[lane] = g;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  const int which = c4 < 3 ? c4 : 0;  // lanes 0 / 1 / 2 of the group: grad_attn / grad_loc.x / grad_loc.y
#pragma unroll 4
  for (int s = part * NS; s < (part + 1) * NS; ++s) {
    const int e = head * 16 + (s ^ head);
    const int4 o = my_off[e];
    const float4 v0 = corner(o.x), v1 = corner(o.y), v2 = corner(o.z), v3 = corner(o.w);
    if (VALUE_ATOMICS) {
      // grad_value scatter (cuh:125-152) in a lane = channel layout: one atomic instruction covers the 32 contiguous
      // channels of TWO heads = two whole 128-byte lines.  (Device-scope float atomics execute memory-side on this
      // part, one request per touched line: the (head, channel-quad) layout of the gather above would touch 8 lines
      // with each of its 4 instructions -- 4x the line requests for the same 256 values.)
      const float* gs = s_go + wave * 256;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int h2 = 2 * k + (lane >> 5), ch = lane & 31;
        const int e2 = h2 * 16 + (s ^ h2);
        const int4 o2 = my_off[e2];
        const float4 r2 = my_a[e2];
        const int b2 = __float_as_int(r2.x);
        const float lh2 = r2.y, lw2 = r2.z, hh2 = 1.f - lh2, hw2 = 1.f - lw2;
        const float t2 = gs[h2 * 32 + ch] * r2.w;  // top = grad_out * attn (cuh:114)
        if (b2 & 1) unsafeAtomicAdd(reinterpret_cast<float*>(gvbase + (unsigned)o2.x) + ch, hh2 * hw2 * t2);
        if (b2 & 2) unsafeAtomicAdd(reinterpret_cast<float*>(gvbase + (unsigned)o2.y) + ch, hh2 * lw2 * t2);
        if (b2 & 4) unsafeAtomicAdd(reinterpret_cast<float*>(gvbase + (unsigned)o2.z) + ch, lh2 * hw2 * t2);
        if (b2 & 8) unsafeAtomicAdd(reinterpret_cast<float*>(gvbase + (unsigned)o2.w) + ch, lh2 * lw2 * t2);
      }
    }
    const float4 cf = my_cf[e * 3 + which];
    float d0 = g.x * v0.x + g.y * v0.y + g.z * v0.z + g.w * v0.w;
    float d1 = g.x * v1.x + g.y * v1.y + g.z * v1.z + g.w * v1.w;
    float d2 = g.x * v2.x + g.y * v2.y + g.z * v2.z + g.w * v2.w;
    float d3 = g.x * v3.x + g.y * v3.y + g.z * v3.z + g.w * v3.w;
    d0 = group8_sum(d0);
    d1 = group8_sum(d1);
    d2 = group8_sum(d2);
    d3 = group8_sum(d3);
    const float r = cf.x * d0 + cf.y * d1 + cf.z * d2 + cf.w * d3;
    // lanes 3..7 of a group repeat lane 0's result and store it to the same place: no branch in the loop, so the
    // compiler can keep the next samples' gathers in flight
    reinterpret_cast<float*>(my_cf + e * 3 + which)[0] = r;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // result of (head h, sample s), component c (0 = grad_attn, 1 / 2 = grad_loc x / y)
  auto res = [&](int h, int s, int c) { return reinterpret_cast<const float*>(my_cf + (h * 16 + (s ^ h)) * 3 + c)[0]; };
  if (SPLIT == 1) {
    const int sa = 2 * c4, sb = 2 * c4 + 1;
    reinterpret_cast<float4*>(grad_loc + (size_t)q * 256)[lane] =
        make_float4(res(head, sa, 1), res(head, sa, 2), res(head, sb, 1), res(head, sb, 2));
    reinterpret_cast<float2*>(grad_attn + (size_t)q * 128)[lane] = make_float2(res(head, sa, 0), res(head, sb, 0));
  } else {
    // this wave's samples: per head 2 * NS location gradients and NS attention gradients
    for (int e = lane; e < 8 * 2 * NS; e += 64) {
      const int h = e / (2 * NS), t = e % (2 * NS), s = part * NS + t / 2;
      grad_loc[(size_t)q * 256 + (h * 16 + s) * 2 + (t & 1)] = res(h, s, 1 + (t & 1));
    }
    for (int e = lane; e < 8 * NS; e += 64) {
      const int h = e / NS, s = part * NS + e % NS;
      grad_attn[(size_t)q * 128 + h * 16 + s] = res(h, s, 0);
    }
  }
}

// Generic backward: one thread per (b,q,m,l,p) sample, loops over the D channels; atomics for grad_value.
template <typename T>
__global__ void msda_bwd_generic(const T* __restrict__ grad_out, const T* __restrict__ value,
                                     const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
                                     const T* __restrict__ loc, const T* __restrict__ attn,
                                     T* __restrict__ grad_value, T* __restrict__ grad_loc,
                                     T* __restrict__ grad_attn, long long n, int S, int M, int D, int L, int Lq,
                                     int P) {
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n;
       idx += (long long)gridDim.x * blockDim.x) {
    long long t = idx / P;
    const int l = (int)(t % L);
    t /= L;
    const int m = (int)(t % M);
    t /= M;  // b*Lq + q
    const int b = (int)(t / Lq);
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1], st = (int)lsi[l];
    const T x = loc[idx * 2] * W - (T)0.5, y = loc[idx * 2 + 1] * H - (T)0.5;
    T ga = 0, gw = 0, gh = 0;
    if (y > (T)-1 && x > (T)-1 && y < (T)H && x < (T)W) {
      const T yf = floor(y), xf = floor(x);
      const int y0 = (int)yf, x0 = (int)xf, y1 = y0 + 1, x1 = x0 + 1;
      const T lh = y - yf, lw = x - xf, hh = (T)1 - lh, hw = (T)1 - lw;
      const bool k0 = y0 >= 0 && x0 >= 0, k1 = y0 >= 0 && x1 <= W - 1, k2 = y1 <= H - 1 && x0 >= 0,
                 k3 = y1 <= H - 1 && x1 <= W - 1;
      const size_t bo = (size_t)b * S * M * D + m * D;
      const size_t o0 = bo + (size_t)(st + y0 * W + x0) * M * D, o1 = bo + (size_t)(st + y0 * W + x1) * M * D,
                   o2 = bo + (size_t)(st + y1 * W + x0) * M * D, o3 = bo + (size_t)(st + y1 * W + x1) * M * D;
      const T a = attn[idx];
      const T* go = grad_out + (size_t)(t * M + m) * D;
      for (int c = 0; c < D; ++c) {
        const T v0 = k0 ? value[o0 + c] : (T)0, v1 = k1 ? value[o1 + c] : (T)0, v2 = k2 ? value[o2 + c] : (T)0,
                v3 = k3 ? value[o3 + c] : (T)0;
        const T top = go[c] * a;
        if (k0) unsafeAtomicAdd(grad_value + o0 + c, hh * hw * top);
        if (k1) unsafeAtomicAdd(grad_value + o1 + c, hh * lw * top);
        if (k2) unsafeAtomicAdd(grad_value + o2 + c, lh * hw * top);
        if (k3) unsafeAtomicAdd(grad_value + o3 + c, lh * lw * top);
        ga += go[c] * (hh * hw * v0 + hh * lw * v1 + lh * hw * v2 + lh * lw * v3);
        gw += (-hh * v0 + hh * v1 - lh * v2 + lh * v3) * top;
        gh += (-hw * v0 - lw * v1 + hw * v2 + lw * v3) * top;
      }
      gw *= W;
      gh *= H;
    }
    grad_attn[idx] = ga;
    grad_loc[idx * 2] = gw;
    grad_loc[idx * 2 + 1] = gh;
  }
}

bool fast_shape(int M, int D, int L, int P) { return M == 8 && D == 32 && L >= 1 && L <= 4 && L * P == 16; }

}  // namespace

// variant: 0 = automatic (wave-per-query when M = 8, D = 32, L*P = 16, else generic), 1 = wave-per-query,
//          3 = generic one-thread-per-element (A/B parity tests).  The LDS-window designs that used to sit behind other
//          variant numbers were measured slower and removed.
extern "C" int egtr_msda_forward_f32_variant(egtr_stream_t stream, const float* value,
                                             const int64_t* spatial_shapes, const int64_t* level_start_index,
                                             const float* sampling_loc, const float* attn_weight, int batch,
                                             int spatial_size, int num_heads, int channels, int num_levels,
                                             int num_query, int num_point, float* out, int variant) {
  if (!value || !spatial_shapes || !level_start_index || !sampling_loc || !attn_weight || !out) return EGTR_E_ARG;
  if (batch <= 0 || spatial_size <= 0 || num_heads <= 0 || channels <= 0 || num_levels <= 0 || num_query <= 0 ||
      num_point <= 0)
    return EGTR_E_ARG;
  if (variant != 0 && variant != 1 && variant != 3) return EGTR_E_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long nq = (long long)batch * num_query;
  const bool fast = fast_shape(num_heads, channels, num_levels, num_point) &&
                    (long long)spatial_size * 1024 < (1ll << 31) && nq < (1ll << 27);
  if (variant == 0) variant = fast ? 1 : 3;
  if (variant == 1 && !fast) return EGTR_E_UNSUPPORTED;
  if (variant == 1 && nq <= kSplitMaxQueries && num_levels * num_point == 16) {
    hipLaunchKernelGGL((msda_fwd_q64_f32<false, true>), dim3((int)nq), dim3(kWaves * 64), 0, st, value, spatial_shapes,
                       level_start_index, sampling_loc, attn_weight, out, (int)nq, num_query, spatial_size,
                       num_levels, num_point, (int)nq, (const float*)nullptr, (float*)nullptr, 256, 128,
                       (const unsigned char*)nullptr, (const unsigned*)nullptr, (const float*)nullptr);
  } else if (variant == 1) {
    const int nblk = (int)((nq + kWaves - 1) / kWaves);
    hipLaunchKernelGGL(msda_fwd_q64_f32<false>, dim3(nblk), dim3(kWaves * 64), 0, st, value, spatial_shapes,
                       level_start_index, sampling_loc, attn_weight, out, (int)nq, num_query, spatial_size,
                       num_levels, num_point, nblk, (const float*)nullptr, (float*)nullptr, 256, 128,
                       (const unsigned char*)nullptr, (const unsigned*)nullptr, (const float*)nullptr);
  } else {
    const long long n = nq * num_heads * channels;
    const int threads = 256;
    const int blocks = (int)std::min<long long>((n + threads - 1) / threads, 65535ll * 16);
    hipLaunchKernelGGL(msda_fwd_generic<float>, dim3(blocks), dim3(threads), 0, st, value, spatial_shapes,
                       level_start_index, sampling_loc, attn_weight, out, n, spatial_size, num_heads, channels,
                       num_levels, num_query, num_point);
  }
  return egtr_check_launch();
}

namespace {
template <bool BOX>
int launch_fused_f32(egtr_stream_t stream, const float* value, const int64_t* spatial_shapes,
                     const int64_t* level_start_index, const float* sampling_offsets, const float* attn_logits,
                     const float* reference_points, int batch, int spatial_size, int num_heads, int channels,
                     int num_levels, int num_query, int num_point, float* out, float* attn_weight_out, int ld_offsets,
                     int ld_logits, const unsigned char* keep_mask, const unsigned* keep_bits,
                     const float* value_bias) {
  if (!value || !spatial_shapes || !level_start_index || !sampling_offsets || !attn_logits || !reference_points ||
      !out)
    return EGTR_E_ARG;
  if (batch <= 0 || spatial_size <= 0 || num_query <= 0) return EGTR_E_ARG;
  const long long nq = (long long)batch * num_query;
  if (ld_offsets < 256 || ld_logits < 128 || (ld_offsets & 3) || (ld_logits & 1)) return EGTR_E_ARG;
  if (!fast_shape(num_heads, channels, num_levels, num_point) || (num_point & 1) ||
      (long long)spatial_size * 1024 >= (1ll << 31) || nq >= (1ll << 27))
    return EGTR_E_UNSUPPORTED;
  if (nq <= kSplitMaxQueries) {  // fewer waves than SIMDs: split each query's samples over a workgroup
    hipLaunchKernelGGL((msda_fwd_q64_f32<true, true, BOX>), dim3((int)nq), dim3(kWaves * 64), 0,
                       static_cast<hipStream_t>(stream), value, spatial_shapes, level_start_index, sampling_offsets,
                       attn_logits, out, (int)nq, num_query, spatial_size, num_levels, num_point, (int)nq,
                       reference_points, attn_weight_out, ld_offsets, ld_logits, keep_mask, keep_bits, value_bias);
    return egtr_check_launch();
  }
  const int nblk = (int)((nq + kWaves - 1) / kWaves);
  hipLaunchKernelGGL((msda_fwd_q64_f32<true, false, BOX>), dim3(nblk), dim3(kWaves * 64), 0,
                     static_cast<hipStream_t>(stream), value, spatial_shapes, level_start_index, sampling_offsets,
                     attn_logits, out, (int)nq, num_query, spatial_size, num_levels, num_point, nblk,
                     reference_points, attn_weight_out, ld_offsets, ld_logits, keep_mask, keep_bits, value_bias);
  return egtr_check_launch();
}
}  // namespace

extern "C" int egtr_msda_forward_fused_vbias_f32(egtr_stream_t stream, const float* value,
                                                 const int64_t* spatial_shapes, const int64_t* level_start_index,
                                                 const float* sampling_offsets, const float* attn_logits,
                                                 const float* reference_points, int batch, int spatial_size,
                                                 int num_heads, int channels, int num_levels, int num_query,
                                                 int num_point, float* out, float* attn_weight_out, int ld_offsets,
                                                 int ld_logits, const unsigned char* keep_mask,
                                                 const unsigned* keep_bits, const float* value_bias) {
  return launch_fused_f32<false>(stream, value, spatial_shapes, level_start_index, sampling_offsets, attn_logits,
                                 reference_points, batch, spatial_size, num_heads, channels, num_levels, num_query,
                                 num_point, out, attn_weight_out, ld_offsets, ld_logits, keep_mask, keep_bits,
                                 value_bias);
}

extern "C" int egtr_msda_forward_fused_box_f32(egtr_stream_t stream, const float* value,
                                               const int64_t* spatial_shapes, const int64_t* level_start_index,
                                               const float* sampling_offsets, const float* attn_logits,
                                               const float* reference_boxes, int batch, int spatial_size,
                                               int num_heads, int channels, int num_levels, int num_query,
                                               int num_point, float* out, float* attn_weight_out, int ld_offsets,
                                               int ld_logits, const unsigned char* keep_mask,
                                               const unsigned* keep_bits, const float* value_bias) {
  return launch_fused_f32<true>(stream, value, spatial_shapes, level_start_index, sampling_offsets, attn_logits,
                                reference_boxes, batch, spatial_size, num_heads, channels, num_levels, num_query,
                                num_point, out, attn_weight_out, ld_offsets, ld_logits, keep_mask, keep_bits,
                                value_bias);
}

extern "C" int egtr_msda_forward_f32(egtr_stream_t stream, const float* value, const int64_t* spatial_shapes,
                                     const int64_t* level_start_index, const float* sampling_loc,
                                     const float* attn_weight, int batch, int spatial_size, int num_heads,
                                     int channels, int num_levels, int num_query, int num_point, float* out) {
  return egtr_msda_forward_f32_variant(stream, value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                                       batch, spatial_size, num_heads, channels, num_levels, num_query, num_point,
                                       out, 0);
}

extern "C" int egtr_msda_forward_bf16(egtr_stream_t stream, const uint16_t* value, const int64_t* spatial_shapes,
                                      const int64_t* level_start_index, const float* sampling_loc,
                                      const float* attn_weight, int batch, int spatial_size, int num_heads,
                                      int channels, int num_levels, int num_query, int num_point, uint16_t* out) {
  if (!value || !spatial_shapes || !level_start_index || !sampling_loc || !attn_weight || !out) return EGTR_E_ARG;
  if (batch <= 0 || spatial_size <= 0 || num_query <= 0) return EGTR_E_ARG;
  if (!fast_shape(num_heads, channels, num_levels, num_point) || (long long)spatial_size * 512 >= (1ll << 31) ||
      (long long)(batch + 1) * spatial_size * 512 >= (1ll << 32))
    return EGTR_E_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long nq = (long long)batch * num_query;
  const int nblk = (int)((nq + 2 * kWaves - 1) / (2 * kWaves));
  if (num_point == 4)
    hipLaunchKernelGGL((msda_fwd_q32_bf16<false, 4>), dim3(nblk), dim3(kWaves * 64), 0, st, value, spatial_shapes,
                       level_start_index, (const void*)sampling_loc, (const void*)attn_weight, out, (int)nq, num_query,
                       spatial_size, num_levels, num_point, nblk, (const uint16_t*)nullptr, 256, 128,
                       (const unsigned char*)nullptr, (const unsigned*)nullptr);
  else
    hipLaunchKernelGGL((msda_fwd_q32_bf16<false, 0>), dim3(nblk), dim3(kWaves * 64), 0, st, value, spatial_shapes,
                       level_start_index, (const void*)sampling_loc, (const void*)attn_weight, out, (int)nq, num_query,
                       spatial_size, num_levels, num_point, nblk, (const uint16_t*)nullptr, 256, 128,
                       (const unsigned char*)nullptr, (const unsigned*)nullptr);
  return egtr_check_launch();
}

extern "C" int egtr_msda_forward_fused_bf16(egtr_stream_t stream, const uint16_t* value, const int64_t* spatial_shapes,
                                            const int64_t* level_start_index, const uint16_t* sampling_offsets,
                                            const uint16_t* attn_logits, const uint16_t* reference_points, int batch,
                                            int spatial_size, int num_heads, int channels, int num_levels,
                                            int num_query, int num_point, uint16_t* out, int ld_offsets, int ld_logits,
                                            const unsigned char* keep_mask, const unsigned* keep_bits) {
  if (!value || !spatial_shapes || !level_start_index || !sampling_offsets || !attn_logits || !reference_points ||
      !out)
    return EGTR_E_ARG;
  if (batch <= 0 || spatial_size <= 0 || num_query <= 0) return EGTR_E_ARG;
  if (ld_offsets < 256 || ld_logits < 128 || (ld_offsets & 3) || (ld_logits & 1)) return EGTR_E_ARG;
  const long long nq = (long long)batch * num_query;
  if (!fast_shape(num_heads, channels, num_levels, num_point) || (num_point & 1) ||
      (long long)spatial_size * 512 >= (1ll << 31) || nq >= (1ll << 27) ||
      (long long)(batch + 1) * spatial_size * 512 >= (1ll << 32))   // (32-bit lane offsets into the value tensor)
    return EGTR_E_UNSUPPORTED;
  const int nblk = (int)((nq + 2 * kWaves - 1) / (2 * kWaves));
  if (num_point == 4)
    hipLaunchKernelGGL((msda_fwd_q32_bf16<true, 4>), dim3(nblk), dim3(kWaves * 64), 0, static_cast<hipStream_t>(stream), value,
                       spatial_shapes, level_start_index, (const void*)sampling_offsets, (const void*)attn_logits, out,
                       (int)nq, num_query, spatial_size, num_levels, num_point, nblk, reference_points, ld_offsets,
                       ld_logits, keep_mask, keep_bits);
  else
    hipLaunchKernelGGL((msda_fwd_q32_bf16<true, 0>), dim3(nblk), dim3(kWaves * 64), 0, static_cast<hipStream_t>(stream), value,
                       spatial_shapes, level_start_index, (const void*)sampling_offsets, (const void*)attn_logits, out,
                       (int)nq, num_query, spatial_size, num_levels, num_point, nblk, reference_points, ld_offsets,
                       ld_logits, keep_mask, keep_bits);
  return egtr_check_launch();
}

int egtr_launch_msda_bwd_value_tile_f32(hipStream_t st, const void* grad_out, bool grad_out_bf16, const int64_t* shapes,
                                        const int64_t* lsi, const float* loc, const float* attn, float* grad_value,
                                        int B, int Lq, int S, int L, int P, unsigned* counters);
unsigned* egtr_msda_tile_counters(hipStream_t st);   // eight work counters private to one launch pair (msda_tile.hip)

// variant: 0 = automatic, 1 = wave-per-query with per-sample global atomics (reference scheme), 2 = two kernels:
// wave-per-query for grad_attn / grad_loc (no atomics) + query-tile x head MFMA accumulation of grad_value
// (msda_tile.hip), 3 = generic.  Automatic = 2 for encoder-shaped calls (Lq == S: queries are the pixels, so tiles
// have compact windows), 1 for short / arbitrary query lists, 3 for shapes other than M = 8, D = 32, L*P = 16.
namespace {
// zero_inside: grad_value arrives uninitialised and is cleared here -- by the wave-per-query kernel itself where the value-tile
// kernel follows it (variant 2), by a memset on the stream otherwise
// bf16_ops: grad_out / value are raw bfloat16 (fast shapes only: EGTR_E_UNSUPPORTED otherwise, the caller widens and comes back)
int msda_backward_f32_impl(egtr_stream_t stream, const void* grad_out, const void* value,
                           const int64_t* spatial_shapes, const int64_t* level_start_index,
                           const float* sampling_loc, const float* attn_weight, int batch,
                           int spatial_size, int num_heads, int channels, int num_levels,
                           int num_query, int num_point, float* grad_value,
                           float* grad_sampling_loc, float* grad_attn_weight, int variant, bool zero_inside,
                           bool bf16_ops = false) {
  if (!grad_out || !value || !spatial_shapes || !level_start_index || !sampling_loc || !attn_weight || !grad_value ||
      !grad_sampling_loc || !grad_attn_weight)
    return EGTR_E_ARG;
  if (batch <= 0 || spatial_size <= 0 || num_heads <= 0 || channels <= 0 || num_levels <= 0 || num_query <= 0 ||
      num_point <= 0)
    return EGTR_E_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long nq = (long long)batch * num_query;
  const bool fast = fast_shape(num_heads, channels, num_levels, num_point) &&
                    (long long)spatial_size * 1024 < (1ll << 31) && nq < (1ll << 27);
  if (variant == 0) variant = fast ? ((num_query == spatial_size && num_query >= 256) ? 2 : 1) : 3;
  if ((variant == 1 || variant == 2) && !fast) return EGTR_E_UNSUPPORTED;
  if (bf16_ops && variant == 3) return EGTR_E_UNSUPPORTED;
  const bool zero_in_kernel = zero_inside && variant == 2 && num_query == spatial_size;
  if (zero_inside && !zero_in_kernel &&
      hipMemsetAsync(grad_value, 0, (size_t)batch * spatial_size * num_heads * channels * sizeof(float), st) != hipSuccess)
    return EGTR_E_LAUNCH;
  if (variant == 2) {
    const int nblk = (int)((nq + kWaves - 1) / kWaves);
    unsigned* counters = egtr_msda_tile_counters(st);
    if (counters == nullptr) return EGTR_E_LAUNCH;
    if (bf16_ops)
      hipLaunchKernelGGL((msda_bwd_q64_f32<false, 1, true>), dim3(nblk), dim3(kWaves * 64), 0, st, grad_out, value,
                         spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_value, grad_sampling_loc,
                         grad_attn_weight, (int)nq, num_query, spatial_size, num_levels, num_point, nblk, counters,
                         zero_in_kernel ? 1 : 0);
    else
      hipLaunchKernelGGL(msda_bwd_q64_f32<false>, dim3(nblk), dim3(kWaves * 64), 0, st, grad_out, value, spatial_shapes,
                         level_start_index, sampling_loc, attn_weight, grad_value, grad_sampling_loc,
                         grad_attn_weight, (int)nq, num_query, spatial_size, num_levels, num_point, nblk, counters,
                         zero_in_kernel ? 1 : 0);
    const int st1 = egtr_check_launch();
    if (st1 != EGTR_OK) return st1;
    return egtr_launch_msda_bwd_value_tile_f32(st, grad_out, bf16_ops, spatial_shapes, level_start_index, sampling_loc,
                                               attn_weight, grad_value, batch, num_query, spatial_size, num_levels,
                                               num_point, counters);
  }
  if (variant == 1 && nq * 4 <= 16384 && (num_point == 4 || num_point == 8 || num_point == 16)) {
    // short query list: 4 waves per query (one level's samples each for L = P = 4)
    const int nblk = (int)((nq * 4 + kWaves - 1) / kWaves);
    if (bf16_ops)
      hipLaunchKernelGGL((msda_bwd_q64_f32<true, 4, true>), dim3(nblk), dim3(kWaves * 64), 0, st, grad_out, value,
                         spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_value, grad_sampling_loc,
                         grad_attn_weight, (int)nq, num_query, spatial_size, num_levels, num_point, nblk, (unsigned*)nullptr, 0);
    else
      hipLaunchKernelGGL((msda_bwd_q64_f32<true, 4>), dim3(nblk), dim3(kWaves * 64), 0, st, grad_out, value,
                         spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_value, grad_sampling_loc,
                         grad_attn_weight, (int)nq, num_query, spatial_size, num_levels, num_point, nblk, (unsigned*)nullptr, 0);
  } else if (variant == 1) {
    const int nblk = (int)((nq + kWaves - 1) / kWaves);
    if (bf16_ops)
      hipLaunchKernelGGL((msda_bwd_q64_f32<true, 1, true>), dim3(nblk), dim3(kWaves * 64), 0, st, grad_out, value,
                         spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_value, grad_sampling_loc,
                         grad_attn_weight, (int)nq, num_query, spatial_size, num_levels, num_point, nblk, (unsigned*)nullptr, 0);
    else
      hipLaunchKernelGGL(msda_bwd_q64_f32<true>, dim3(nblk), dim3(kWaves * 64), 0, st, grad_out, value, spatial_shapes,
                         level_start_index, sampling_loc, attn_weight, grad_value, grad_sampling_loc,
                         grad_attn_weight, (int)nq, num_query, spatial_size, num_levels, num_point, nblk, (unsigned*)nullptr, 0);
  } else {
    const long long n = nq * num_heads * num_levels * num_point;
    const int threads = 256;
    const int blocks = (int)std::min<long long>((n + threads - 1) / threads, 65535ll * 16);
    hipLaunchKernelGGL(msda_bwd_generic<float>, dim3(blocks), dim3(threads), 0, st, static_cast<const float*>(grad_out),
                       static_cast<const float*>(value), spatial_shapes,
                       level_start_index, sampling_loc, attn_weight, grad_value, grad_sampling_loc,
                       grad_attn_weight, n, spatial_size, num_heads, channels, num_levels, num_query, num_point);
  }
  return egtr_check_launch();
}
}  // namespace

extern "C" int egtr_msda_backward_f32_variant(egtr_stream_t stream, const float* grad_out, const float* value,
                                              const int64_t* spatial_shapes, const int64_t* level_start_index,
                                              const float* sampling_loc, const float* attn_weight, int batch,
                                              int spatial_size, int num_heads, int channels, int num_levels,
                                              int num_query, int num_point, float* grad_value,
                                              float* grad_sampling_loc, float* grad_attn_weight, int variant) {
  return msda_backward_f32_impl(stream, grad_out, value, spatial_shapes, level_start_index, sampling_loc, attn_weight, batch,
                                spatial_size, num_heads, channels, num_levels, num_query, num_point, grad_value,
                                grad_sampling_loc, grad_attn_weight, variant, false);
}

extern "C" int egtr_msda_backward_out_f32(egtr_stream_t stream, const float* grad_out, const float* value,
                                          const int64_t* spatial_shapes, const int64_t* level_start_index,
                                          const float* sampling_loc, const float* attn_weight, int batch,
                                          int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                                          int num_point, float* grad_value, float* grad_sampling_loc,
                                          float* grad_attn_weight) {
  return msda_backward_f32_impl(stream, grad_out, value, spatial_shapes, level_start_index, sampling_loc, attn_weight, batch,
                                spatial_size, num_heads, channels, num_levels, num_query, num_point, grad_value,
                                grad_sampling_loc, grad_attn_weight, 0, true);
}

extern "C" int egtr_msda_backward_f32(egtr_stream_t stream, const float* grad_out, const float* value,
                                      const int64_t* spatial_shapes, const int64_t* level_start_index,
                                      const float* sampling_loc, const float* attn_weight, int batch,
                                      int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                                      int num_point, float* grad_value, float* grad_sampling_loc,
                                      float* grad_attn_weight) {
  return egtr_msda_backward_f32_variant(stream, grad_out, value, spatial_shapes, level_start_index, sampling_loc,
                                        attn_weight, batch, spatial_size, num_heads, channels, num_levels, num_query,
                                        num_point, grad_value, grad_sampling_loc, grad_attn_weight, 0);
}


// ---- float64 entries: the reference dispatches AT_DISPATCH_FLOATING_TYPES (ms_deform_attn_cuda.cu:67, 137), so a
// gradcheck-style caller of the extension passes double.  Served by the generic kernels (any M, D, L, P).
extern "C" int egtr_msda_forward_f64(egtr_stream_t stream, const double* value, const int64_t* spatial_shapes,
                                     const int64_t* level_start_index, const double* sampling_loc,
                                     const double* attn_weight, int batch, int spatial_size, int num_heads,
                                     int channels, int num_levels, int num_query, int num_point, double* out) {
  if (!value || !spatial_shapes || !level_start_index || !sampling_loc || !attn_weight || !out) return EGTR_E_ARG;
  if (batch <= 0 || spatial_size <= 0 || num_heads <= 0 || channels <= 0 || num_levels <= 0 || num_query <= 0 ||
      num_point <= 0)
    return EGTR_E_ARG;
  const long long n = (long long)batch * num_query * num_heads * channels;
  const int blocks = (int)std::min<long long>((n + 255) / 256, 65535ll * 16);
  hipLaunchKernelGGL(msda_fwd_generic<double>, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), value,
                     spatial_shapes, level_start_index, sampling_loc, attn_weight, out, n, spatial_size, num_heads,
                     channels, num_levels, num_query, num_point);
  return egtr_check_launch();
}

extern "C" int egtr_msda_backward_f64(egtr_stream_t stream, const double* grad_out, const double* value,
                                      const int64_t* spatial_shapes, const int64_t* level_start_index,
                                      const double* sampling_loc, const double* attn_weight, int batch,
                                      int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                                      int num_point, double* grad_value, double* grad_sampling_loc,
                                      double* grad_attn_weight) {
  if (!grad_out || !value || !spatial_shapes || !level_start_index || !sampling_loc || !attn_weight || !grad_value ||
      !grad_sampling_loc || !grad_attn_weight)
    return EGTR_E_ARG;
  if (batch <= 0 || spatial_size <= 0 || num_heads <= 0 || channels <= 0 || num_levels <= 0 || num_query <= 0 ||
      num_point <= 0)
    return EGTR_E_ARG;
  const long long n = (long long)batch * num_query * num_heads * num_levels * num_point;
  const int blocks = (int)std::min<long long>((n + 255) / 256, 65535ll * 16);
  hipLaunchKernelGGL(msda_bwd_generic<double>, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), grad_out,
                     value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_value, grad_sampling_loc,
                     grad_attn_weight, n, spatial_size, num_heads, channels, num_levels, num_query, num_point);
  return egtr_check_launch();
}

// ---- bf16 backward (the reference has no half / bf16 kernel; the stress configuration trains in bf16): bf16 value and
// upstream gradient, fp32 sampling geometry and fp32 gradients.  The fast shapes (M = 8, D = 32, L * P = 16) read the bf16
// operands directly (msda_bwd_q64_f32<.., BF = true>, msda_bwd_value_tile_f32<true>: widened on load, identical arithmetic
// behind it); other shapes widen them once into `workspace` (B*S*M*D + B*Lq*M*D floats) for the generic fp32 kernel.
namespace {
__global__ __launch_bounds__(256) void widen_bf16(const uint16_t* __restrict__ a, float* __restrict__ oa, long long na,
                                                  const uint16_t* __restrict__ b, float* __restrict__ ob, long long nb) {
  const long long n8 = (na + nb) / 8;   // both counts are multiples of 8 (M * D = 256 elements per row)
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    const long long e = i * 8;
    const bool first = e < na;
    const uint4 v = *reinterpret_cast<const uint4*>(first ? a + e : b + (e - na));
    float* o = first ? oa + e : ob + (e - na);
    reinterpret_cast<float4*>(o)[0] = make_float4(bf16_lo(v.x), bf16_hi(v.x), bf16_lo(v.y), bf16_hi(v.y));
    reinterpret_cast<float4*>(o)[1] = make_float4(bf16_lo(v.z), bf16_hi(v.z), bf16_lo(v.w), bf16_hi(v.w));
  }
}
}  // namespace

extern "C" long long egtr_msda_backward_bf16_workspace_floats(int batch, int spatial_size, int num_heads, int channels,
                                                             int num_levels, int num_query, int num_point) {
  if (batch <= 0 || spatial_size <= 0 || num_heads <= 0 || channels <= 0 || num_query <= 0) return 0;
  const long long nq = (long long)batch * num_query;
  if (fast_shape(num_heads, channels, num_levels, num_point) && (long long)spatial_size * 1024 < (1ll << 31) && nq < (1ll << 27))
    return 0;   // native bf16 operands
  return (long long)batch * spatial_size * num_heads * channels + nq * num_heads * channels;
}

extern "C" int egtr_msda_backward_bf16(egtr_stream_t stream, const uint16_t* grad_out, const uint16_t* value,
                                       const int64_t* spatial_shapes, const int64_t* level_start_index,
                                       const float* sampling_loc, const float* attn_weight, int batch,
                                       int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                                       int num_point, float* grad_value, float* grad_sampling_loc,
                                       float* grad_attn_weight, float* workspace) {
  if (!grad_out || !value) return EGTR_E_ARG;
  if (batch <= 0 || spatial_size <= 0 || num_heads <= 0 || channels <= 0 || num_query <= 0) return EGTR_E_ARG;
  {
    const int st0 = msda_backward_f32_impl(stream, grad_out, value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                                           batch, spatial_size, num_heads, channels, num_levels, num_query, num_point,
                                           grad_value, grad_sampling_loc, grad_attn_weight, 0, false, true);
    if (st0 != EGTR_E_UNSUPPORTED) return st0;
  }
  if (!workspace) return EGTR_E_ARG;
  const long long nv = (long long)batch * spatial_size * num_heads * channels;
  const long long ng = (long long)batch * num_query * num_heads * channels;
  if ((nv & 7) || (ng & 7)) return EGTR_E_UNSUPPORTED;
  const int blocks = (int)std::min<long long>(((nv + ng) / 8 + 255) / 256, 65535ll);
  hipLaunchKernelGGL(widen_bf16, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), value, workspace, nv,
                     grad_out, workspace + nv, ng);
  const int st = egtr_check_launch();
  if (st != EGTR_OK) return st;
  return egtr_msda_backward_f32(stream, workspace + nv, workspace, spatial_shapes, level_start_index, sampling_loc,
                                attn_weight, batch, spatial_size, num_heads, channels, num_levels, num_query, num_point,
                                grad_value, grad_sampling_loc, grad_attn_weight);
}
