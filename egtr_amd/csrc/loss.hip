// Relation / connectivity losses of the SGG criterion on the device: loss AND gradient in a handful of
// passes over pred_rel, no host synchronisation, no index lists.
//
// Reference (model/egtr.py:754-923, training mode with rel_sample_negatives / rel_sample_nonmatching and *_largest=True --
// the configuration train_egtr.py uses): per image it permutes pred_rel / target_rel into "matched queries first" order,
// materialises three nonzero() index lists (true relations of the matched block; false candidates of the block; every
// element with an unmatched subject or object: ~2 M x 3 int64), copies counts to the host, runs two topk's over gathered
// scores and a BCE over the concatenated gather.  The result is a MEAN over a SET of elements, so it can be evaluated in
// place:   loss_rel = sum_{selected e} BCE(x_e, t_e * w_a w_b) / #selected,
//   selected = true relations of the block  U  the k1 largest-logit false candidates of the block  U  the k2 largest-logit
//   elements outside the block, k1 = min(80 n_true, n_false), k2 = min(80 n_true, n_outside) (0 when n_true = 0).
// "k largest" is a threshold: a 3-pass radix select (11 + 11 + 10 bits of the order-preserving integer image of the fp32
// logit) finds the k-th largest key exactly; elements above it are selected, elements equal to it take tickets until k
// is reached (ties have equal logits; which of them the reference's topk returns is unspecified too).
// Launches: prep | 3 x (histogram, find) | final (loss terms, dense d loss / d pred_rel, pair flags) | connectivity.
// All counts stay on the device; the only inputs from the host are tensor shapes.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "common.h"

namespace {

constexpr int kLT = 256;
constexpr int kRowsPerWg = 1;    // query rows (N*R elements each) per workgroup of the histogram / final passes (8 until round 4: 100
                                 // workgroups at B = 4, N = 200 left 60 % of the CUs idle: final pass 391 us, histogram passes 200 us each)
constexpr int kBins = 2048;

// workspace layout per image (ints), see egtr_relation_loss_workspace_bytes()
struct ImgState {
  int T;            // matched pairs
  int n_true, n_false;
  int k[2];         // elements to select: [0] false candidates of the block, [1] outside the block
  int krem[2];      // still to find below the current prefix
  unsigned prefix[2];   // key bits fixed so far (left-aligned)
  unsigned thr[2];  // final threshold key
  int need[2];      // elements equal to thr to take
  int ticket[2];    // tickets handed out to elements equal to thr
  int n_sel;        // n_true + k[0] + k[1]
  int pad[4];
};
static_assert(sizeof(ImgState) == 80, "layout");

__device__ __forceinline__ unsigned key_of(float x) {   // order-preserving: larger float <-> larger unsigned
  const unsigned u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ float bce_logits(float x, float y) {   // BCEWithLogitsLoss(reduction="none")
  return fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
}

constexpr int kPrepT = 1024;   // one workgroup per image, sixteen waves: the T x T x R count below is ~500 dependent gathers per thread with four
// ---- prep: per image  tq (query -> target row), wq (1 - sigmoid(matching cost)), mq (matched), block counts ------------
__global__ __launch_bounds__(kPrepT) void rel_loss_prep(const int64_t* __restrict__ pred_idx,
                                                     const int64_t* __restrict__ tgt_idx,
                                                     const float* __restrict__ match_cost,
                                                     const int* __restrict__ out_off,
                                                     const float* const* __restrict__ target_rel, int N, int R,
                                                     float nonmatching_cost, int k_neg, int k_nm, ImgState* st,
                                                     int* tq, float* wq, unsigned char* mq, int* total_sel) {
  __shared__ int s_cnt[2];
  __shared__ int s_scan[kPrepT];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int o0 = out_off[b];
  int T = out_off[b + 1] - o0;
  // The device matcher writes -1 indices for an image whose cost matrix holds NaN / -inf (scipy raises there): such an
  // image is treated as having NO matches here (never an out-of-bounds access); the Python side turns the matcher's
  // status into NaN loss terms for that output set (SceneGraphGenerationLoss.forward) and a ValueError
  // (DeformableDetrHungarianMatcher.raise_if_invalid).
  int bad_local = 0;
  for (int t = tid; t < T; t += kPrepT) {
    const long long q = pred_idx[o0 + t], g = tgt_idx[o0 + t];
    bad_local |= (q < 0 || q >= N || g < 0 || g >= N) ? 1 : 0;
  }
  if (__syncthreads_or(bad_local)) T = 0;
  int* tq_b = tq + (size_t)b * N;
  float* wq_b = wq + (size_t)b * N;
  unsigned char* mq_b = mq + (size_t)b * N;
  if (tid < 2) s_cnt[tid] = 0;
  for (int q = tid; q < N; q += kPrepT) { mq_b[q] = 0; wq_b[q] = 1.0f - 1.0f / (1.0f + expf(-nonmatching_cost)); }
  __syncthreads();
  for (int t = tid; t < T; t += kPrepT) {
    const int q = (int)pred_idx[o0 + t];
    mq_b[q] = 1;
    tq_b[q] = (int)tgt_idx[o0 + t];
    wq_b[q] = 1.0f - 1.0f / (1.0f + expf(-match_cost[o0 + t]));     // 1 - cost.sigmoid()  (egtr:844, 920)
  }
  __syncthreads();
  // unmatched queries in ascending order take target rows T, T + 1, ... (egtr:761-768)
  int base = 0;
  for (int q0 = 0; q0 < N; q0 += kPrepT) {
    const int q = q0 + tid;
    const int un = (q < N && !mq_b[q]) ? 1 : 0;
    s_scan[tid] = un;
    __syncthreads();
    for (int o = 1; o < kPrepT; o <<= 1) {
      const int v = tid >= o ? s_scan[tid - o] : 0;
      __syncthreads();
      s_scan[tid] += v;
      __syncthreads();
    }
    if (un) tq_b[q] = T + base + s_scan[tid] - 1;
    base += s_scan[kPrepT - 1];
    __syncthreads();
  }
  // counts over the matched block: true = target != 0, false candidates = target != 1 (egtr:838-840)
  const float* rel = target_rel[b];
  int nt = 0, nf = 0;
  const int tot = T * T * R;
  for (int e = tid; e < tot; e += kPrepT) {
    const int a = e / (T * R), rem = e - a * T * R, c = rem / R, r = rem - c * R;
    const float t = rel[((size_t)tgt_idx[o0 + a] * N + (size_t)tgt_idx[o0 + c]) * R + r];
    nt += (t != 0.f);
    nf += (t != 1.0f);
  }
  for (int o = 32; o > 0; o >>= 1) { nt += __shfl_xor(nt, o); nf += __shfl_xor(nf, o); }
  if ((tid & 63) == 0) { atomicAdd(&s_cnt[0], nt); atomicAdd(&s_cnt[1], nf); }
  __syncthreads();
  if (tid == 0) {
    ImgState s;
    s.T = T;
    s.n_true = s_cnt[0];
    s.n_false = s_cnt[1];
    const long long n_nm = ((long long)N * N - (long long)T * T) * R;
    const long long want0 = (long long)s.n_true * k_neg, want1 = (long long)s.n_true * k_nm;
    s.k[0] = s.n_true > 0 ? (int)(want0 < s.n_false ? want0 : s.n_false) : 0;       // egtr:852-858 min(n_true * k, #cands)
    s.k[1] = s.n_true > 0 ? (int)(want1 < n_nm ? want1 : n_nm) : 0;
    s.krem[0] = s.k[0];
    s.krem[1] = s.k[1];
    s.prefix[0] = s.prefix[1] = 0u;
    s.thr[0] = s.thr[1] = 0u;
    s.need[0] = s.need[1] = 0;
    s.ticket[0] = s.ticket[1] = 0;
    s.n_sel = s.n_true + s.k[0] + s.k[1];
    s.pad[0] = s.pad[1] = s.pad[2] = s.pad[3] = 0;
    st[b] = s;
    atomicAdd(total_sel, s.n_sel);
  }
}

// ---- histogram pass: digit PASS (0: bits 31..21, 1: bits 20..10, 2: bits 9..0) of the candidates under the prefix --------
template <int PASS>
__global__ __launch_bounds__(kLT) void rel_loss_hist(const float* __restrict__ pred_rel,
                                                     const float* const* __restrict__ target_rel, int N, int R,
                                                     const ImgState* __restrict__ st, const int* __restrict__ tq,
                                                     const unsigned char* __restrict__ mq, unsigned* __restrict__ hist) {
  __shared__ unsigned s_h[2][kBins];
  const int b = blockIdx.y, tid = threadIdx.x;
  const ImgState s = st[b];
  if (s.krem[0] <= 0 && s.krem[1] <= 0) return;      // nothing (left) to select in this image
  for (int i = tid; i < 2 * kBins; i += kLT) (&s_h[0][0])[i] = 0u;
  __syncthreads();
  const int* tq_b = tq + (size_t)b * N;
  const unsigned char* mq_b = mq + (size_t)b * N;
  const float* rel = target_rel[b];
  const float* pr = pred_rel + (size_t)b * N * N * R;
  const int NR = N * R;
  const float invR = 1.0f / (float)R;
  constexpr int kShift = PASS == 0 ? 21 : (PASS == 1 ? 10 : 0);
  constexpr unsigned kMask = PASS == 2 ? 1023u : 2047u;
  for (int row = 0; row < kRowsPerWg; ++row) {
    const int qa = blockIdx.x * kRowsPerWg + row;
    if (qa >= N) break;
    const bool ma = mq_b[qa] != 0;
    if (!ma && s.krem[1] <= 0) continue;
    const size_t trow = (size_t)tq_b[qa] * N;
    for (int e = tid; e < NR; e += kLT) {
      const int qb = (int)(((float)e + 0.5f) * invR);   // exact: e < 2^22
      const bool blk = ma && mq_b[qb];
      int prob;
      if (blk) {
        if (s.krem[0] <= 0) continue;
        const float t = rel[(trow + tq_b[qb]) * R + (e - qb * R)];
        if (!(t != 1.0f)) continue;
        prob = 0;
      } else {
        if (s.krem[1] <= 0) continue;
        prob = 1;
      }
      const unsigned key = key_of(pr[(size_t)qa * NR + e]);
      if (PASS == 1 && (key >> 21) != (s.prefix[prob] >> 21)) continue;
      if (PASS == 2 && (key >> 10) != (s.prefix[prob] >> 10)) continue;
      atomicAdd(&s_h[prob][(key >> kShift) & kMask], 1u);
    }
  }
  __syncthreads();
  unsigned* gh = hist + ((size_t)b * 3 + PASS) * 2 * kBins;
  for (int i = tid; i < 2 * kBins; i += kLT) {
    const unsigned v = (&s_h[0][0])[i];
    if (v) atomicAdd(&gh[i], v);
  }
}

// ---- find: the digit in which the k-th largest candidate lies (one workgroup per (problem, image)) ----------------------
template <int PASS>
__global__ __launch_bounds__(kLT) void rel_loss_find(ImgState* st, const unsigned* __restrict__ hist) {
  __shared__ unsigned s_sum[kLT];
  const int prob = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int krem = st[b].krem[prob];
  if (krem <= 0) return;
  const unsigned* h = hist + (((size_t)b * 3 + PASS) * 2 + prob) * kBins;
  constexpr int kPer = kBins / kLT;   // 8 bins per thread, thread 0 owns the TOP bins
  unsigned loc[kPer];
  unsigned tot = 0;
#pragma unroll
  for (int i = 0; i < kPer; ++i) {
    loc[i] = h[kBins - 1 - (tid * kPer + i)];
    tot += loc[i];
  }
  s_sum[tid] = tot;
  __syncthreads();
  for (int o = 1; o < kLT; o <<= 1) {
    const unsigned v = tid >= o ? s_sum[tid - o] : 0u;
    __syncthreads();
    s_sum[tid] += v;
    __syncthreads();
  }
  const unsigned before = s_sum[tid] - tot;   // candidates in bins above this thread's range
  if (before < (unsigned)krem && before + tot >= (unsigned)krem) {
    unsigned c = before;
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      if (c + loc[i] >= (unsigned)krem) {
        const unsigned digit = (unsigned)(kBins - 1 - (tid * kPer + i));
        constexpr int kShift = PASS == 0 ? 21 : (PASS == 1 ? 10 : 0);
        const unsigned pref = st[b].prefix[prob] | (digit << kShift);
        st[b].prefix[prob] = pref;
        st[b].krem[prob] = krem - (int)c;         // still to take among the elements of this digit
        if (PASS == 2) {
          st[b].thr[prob] = pref;
          st[b].need[prob] = krem - (int)c;      // elements equal to the threshold key to take
        }
        break;
      }
      c += loc[i];
    }
  }
}

// ---- final pass: loss terms, dense gradient, pair flags (target connectivity in query order) ---------------------------
__global__ __launch_bounds__(kLT) void rel_loss_final(const float* __restrict__ pred_rel,
                                                      const float* const* __restrict__ target_rel, int N, int R,
                                                      ImgState* st, const int* __restrict__ tq,
                                                      const float* __restrict__ wq, const unsigned char* __restrict__ mq,
                                                      const int* __restrict__ total_sel, float* __restrict__ grad_rel,
                                                      unsigned char* __restrict__ pairflag, double* __restrict__ partial) {
  __shared__ double s_red[kLT / 64];
  const int b = blockIdx.y, tid = threadIdx.x;
  const ImgState s = st[b];
  const int* tq_b = tq + (size_t)b * N;
  const float* wq_b = wq + (size_t)b * N;
  const unsigned char* mq_b = mq + (size_t)b * N;
  const float* rel = target_rel[b];
  const float* pr = pred_rel + (size_t)b * N * N * R;
  float* gr = grad_rel + (size_t)b * N * N * R;
  unsigned char* pf = pairflag + (size_t)b * N * N;
  const int NR = N * R;
  const float invR = 1.0f / (float)R;
  const float inv_cnt = 1.0f / (float)(*total_sel);    // mean over every selected element of the batch (egtr:814)
  double acc = 0.0;
  for (int row = 0; row < kRowsPerWg; ++row) {
    const int qa = blockIdx.x * kRowsPerWg + row;
    if (qa >= N) break;
    const bool ma = mq_b[qa] != 0;
    const size_t trow = (size_t)tq_b[qa] * N;
    const float wa = wq_b[qa];
    for (int e = tid; e < NR; e += kLT) {
      const int qb = (int)(((float)e + 0.5f) * invR);
      const float t = rel[(trow + tq_b[qb]) * R + (e - qb * R)];
      if (t != 0.f) pf[(size_t)qa * N + qb] = 1;      // benign race: every writer stores 1
      const bool blk = ma && mq_b[qb];
      const float x = pr[(size_t)qa * NR + e];
      int mult = 0;
      const int prob = blk ? 0 : 1;
      if (blk && t != 0.f) mult = 1;                   // true relation of the matched block (egtr:838)
      if ((!blk || t != 1.0f) && s.k[prob] > 0) {      // sampled by score (egtr:852-895)
        const unsigned key = key_of(x);
        if (key > s.thr[prob]) mult += 1;
        else if (key == s.thr[prob] && atomicAdd(&st[b].ticket[prob], 1) < s.need[prob]) mult += 1;
      }
      float g = 0.f;
      if (mult) {
        const float y = t * (wa * wq_b[qb]);           // target * weight[a] * weight[b]  (egtr:918-922)
        acc += (double)((float)mult * bce_logits(x, y));
        g = (float)mult * (1.0f / (1.0f + expf(-x)) - y) * inv_cnt;
      }
      gr[(size_t)qa * NR + e] = g;
    }
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((tid & 63) == 0) s_red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) {
    double t = 0.0;
    for (int w = 0; w < kLT / 64; ++w) t += s_red[w];
    partial[(size_t)b * gridDim.x + blockIdx.x] = t;
  }
}

// ---- connectivity BCE (egtr:786-793, 815) + final reduction of both losses (last workgroup, fixed order) ----------------
__global__ __launch_bounds__(kLT) void conn_loss(const float* __restrict__ pred_conn,
                                                 const unsigned char* __restrict__ pairflag, long long n_pairs,
                                                 float* __restrict__ grad_conn, double* __restrict__ partial_conn,
                                                 const double* __restrict__ partial_rel, int n_partial_rel,
                                                 const int* __restrict__ total_sel, unsigned* __restrict__ done,
                                                 float* __restrict__ loss_out) {
  __shared__ double s_red[kLT / 64];
  __shared__ bool s_last;
  const int tid = threadIdx.x;
  const float inv = 1.0f / (float)n_pairs;
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * kLT + tid; i < n_pairs; i += (long long)gridDim.x * kLT) {
    const float x = pred_conn[i], y = pairflag[i] ? 1.0f : 0.0f;
    acc += (double)bce_logits(x, y);
    grad_conn[i] = (1.0f / (1.0f + expf(-x)) - y) * inv;
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((tid & 63) == 0) s_red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) {
    double t = 0.0;
    for (int w = 0; w < kLT / 64; ++w) t += s_red[w];
    partial_conn[blockIdx.x] = t;
    __threadfence();
    s_last = atomicAdd(done, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (s_last) {
    // The last workgroup sums the partials of both losses -- every thread a fixed strided share, then lanes and waves in a fixed
    // order: bit-reproducible.  (Round 4: thread 0 alone walked the ~1 400 partials with dependent volatile loads: 125 of the
    // launch's 132 us.)
    __shared__ double s_fin[2][kLT / 64];
    __threadfence();
    double sc = 0.0, sr = 0.0;
    const volatile double* pc = partial_conn;   // written by other workgroups of this launch (released above)
    for (unsigned i = tid; i < gridDim.x; i += kLT) sc += pc[i];
    for (int i = tid; i < n_partial_rel; i += kLT) sr += partial_rel[i];
    for (int o = 32; o > 0; o >>= 1) {
      sc += __shfl_xor(sc, o);
      sr += __shfl_xor(sr, o);
    }
    if ((tid & 63) == 0) {
      s_fin[0][tid >> 6] = sc;
      s_fin[1][tid >> 6] = sr;
    }
    __syncthreads();
    if (tid == 0) {
      double tc = 0.0, tr = 0.0;
      for (int w = 0; w < kLT / 64; ++w) {
        tc += s_fin[0][w];
        tr += s_fin[1][w];
      }
      loss_out[0] = (float)(tr / (double)(*total_sel));   // 0 / 0 = NaN: the reference's mean of an empty tensor
      loss_out[1] = (float)(tc / (double)n_pairs);
    }
  }
}

}  // namespace

extern "C" long long egtr_relation_loss_workspace_bytes(int batch, int num_query) {
  if (batch <= 0 || num_query <= 0) return 0;
  const long long B = batch, N = num_query;
  const long long rows = (N + kRowsPerWg - 1) / kRowsPerWg;
  long long bytes = 256;                       // total_sel, done
  bytes += B * (long long)sizeof(ImgState);
  bytes += B * 3 * 2 * kBins * 4;              // histograms
  bytes += B * N * 4 * 2;                      // tq, wq
  bytes += ((B * N + 15) / 16) * 16;           // mq
  bytes += ((B * N * N + 15) / 16) * 16;       // pairflag
  bytes += (B * rows + 1024) * 8;              // partial sums (relations + connectivity)
  return bytes + 256;
}

// target_rel: DEVICE array of `batch` device pointers (image b's dense [N, N, R] target).  pred_idx / tgt_idx /
// match_cost / out_offsets: the matcher's packed outputs (egtr_hungarian_match_f32).  workspace: device,
// egtr_relation_loss_workspace_bytes() bytes, contents irrelevant (zeroed here).  loss_out[0] = loss_rel,
// loss_out[1] = loss_connectivity; grad_rel [B,N,N,R] / grad_conn [B,N,N] = d loss / d logits.
extern "C" int egtr_relation_loss_f32(egtr_stream_t stream, const float* pred_rel, const float* pred_conn,
                                      const float* const* target_rel, const int64_t* pred_idx, const int64_t* tgt_idx,
                                      const float* match_cost, const int* out_offsets, int batch, int num_query,
                                      int num_rel, float nonmatching_cost, int sample_negatives, int sample_nonmatching,
                                      float* loss_out, float* grad_rel, float* grad_conn, void* workspace) {
  if (!pred_rel || !pred_conn || !target_rel || !pred_idx || !tgt_idx || !match_cost || !out_offsets || !loss_out ||
      !grad_rel || !grad_conn || !workspace)
    return EGTR_E_ARG;
  if (batch <= 0 || num_query <= 0 || num_rel <= 0 || sample_negatives < 0 || sample_nonmatching < 0) return EGTR_E_ARG;
  if ((long long)num_query * num_rel >= (1 << 22) || num_query > 4096) return EGTR_E_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long B = batch, N = num_query;
  const int rows = (int)((N + kRowsPerWg - 1) / kRowsPerWg);
  const long long wsb = egtr_relation_loss_workspace_bytes(batch, num_query);
  if (hipMemsetAsync(workspace, 0, (size_t)wsb, st) != hipSuccess) return EGTR_E_LAUNCH;
  char* p = static_cast<char*>(workspace);
  int* total_sel = reinterpret_cast<int*>(p);
  unsigned* done = reinterpret_cast<unsigned*>(p + 64);
  p += 256;
  ImgState* state = reinterpret_cast<ImgState*>(p);
  p += B * sizeof(ImgState);
  unsigned* hist = reinterpret_cast<unsigned*>(p);
  p += B * 3 * 2 * kBins * 4;
  int* tq = reinterpret_cast<int*>(p);
  p += B * N * 4;
  float* wq = reinterpret_cast<float*>(p);
  p += B * N * 4;
  unsigned char* mq = reinterpret_cast<unsigned char*>(p);
  p += ((B * N + 15) / 16) * 16;
  unsigned char* pairflag = reinterpret_cast<unsigned char*>(p);
  p += ((B * N * N + 15) / 16) * 16;
  p = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(p) + 7) & ~(uintptr_t)7);
  double* partial_rel = reinterpret_cast<double*>(p);
  double* partial_conn = partial_rel + B * rows;
  hipLaunchKernelGGL(rel_loss_prep, dim3(batch), dim3(kPrepT), 0, st, pred_idx, tgt_idx, match_cost, out_offsets, target_rel,
                     num_query, num_rel, nonmatching_cost, sample_negatives, sample_nonmatching, state, tq, wq, mq,
                     total_sel);
  const dim3 gh(rows, batch), gf(2, batch);
  hipLaunchKernelGGL(rel_loss_hist<0>, gh, dim3(kLT), 0, st, pred_rel, target_rel, num_query, num_rel, state, tq, mq, hist);
  hipLaunchKernelGGL(rel_loss_find<0>, gf, dim3(kLT), 0, st, state, hist);
  hipLaunchKernelGGL(rel_loss_hist<1>, gh, dim3(kLT), 0, st, pred_rel, target_rel, num_query, num_rel, state, tq, mq, hist);
  hipLaunchKernelGGL(rel_loss_find<1>, gf, dim3(kLT), 0, st, state, hist);
  hipLaunchKernelGGL(rel_loss_hist<2>, gh, dim3(kLT), 0, st, pred_rel, target_rel, num_query, num_rel, state, tq, mq, hist);
  hipLaunchKernelGGL(rel_loss_find<2>, gf, dim3(kLT), 0, st, state, hist);
  hipLaunchKernelGGL(rel_loss_final, gh, dim3(kLT), 0, st, pred_rel, target_rel, num_query, num_rel, state, tq, wq, mq,
                     total_sel, grad_rel, pairflag, partial_rel);
  const long long n_pairs = B * N * N;
  const int cblocks = (int)std::min<long long>((n_pairs + kLT - 1) / kLT, 1024);
  hipLaunchKernelGGL(conn_loss, dim3(cblocks), dim3(kLT), 0, st, pred_conn, pairflag, n_pairs, grad_conn, partial_conn,
                     partial_rel, (int)(B * rows), total_sel, done, loss_out);
  return egtr_check_launch();
}

// ---- detection losses of one output set (labels / boxes / cardinality, egtr:611-659, 661-670, 692-712) ------------------
// The reference evaluates them as ~40 small tensor ops per output set (7 sets with auxiliary losses), plus their autograd
// graph.  One workgroup per image: target-class map in LDS from the matcher's packed indices, sigmoid focal loss + its
// gradient over the [N, C] logits, argmax-based cardinality count, L1 + generalised-IoU loss + gradients of the matched
// boxes.  Sums are formed in a fixed order (per-thread partial -> LDS -> thread 0): bit-reproducible.  The caller divides
// nothing: every sum and every gradient already carries 1 / num_boxes.
namespace {

constexpr int kDT = 1024;   // one workgroup per image: sixteen waves (round 4; four left ~120 dependent iterations per thread)
constexpr int kDetMaxN = 2048;

__device__ __forceinline__ float softplusf(float z) { return fmaxf(z, 0.f) + log1pf(expf(-fabsf(z))); }

__global__ __launch_bounds__(kDT) void det_loss_f32(
    const float* __restrict__ logits, const float* __restrict__ boxes, const long long* __restrict__ pred_idx,
    const long long* __restrict__ tgt_idx, const int* __restrict__ moff, const long long* __restrict__ tlabels,
    const float* __restrict__ tboxes, const int* __restrict__ toff, int N, int C, float alpha, float inv_num_boxes,
    float* __restrict__ d_logits, float* __restrict__ d_l1, float* __restrict__ d_giou, float* __restrict__ out) {
  __shared__ int s_cls[kDetMaxN];
  __shared__ double s_red[3][kDT / 64];
  __shared__ int s_cnt[kDT / 64];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = moff[b], m1 = moff[b + 1], t0 = toff[b];
  for (int n = tid; n < N; n += kDT) s_cls[n] = C;                 // "no object" (egtr:640-642)
  for (int e = tid; e < N * 4; e += kDT) {
    d_l1[(size_t)b * N * 4 + e] = 0.f;
    d_giou[(size_t)b * N * 4 + e] = 0.f;
  }
  __syncthreads();
  // invalid entries (-1 from a matcher that refused the image's cost matrix) are skipped, never dereferenced
  const int nt_img = toff[b + 1] - t0;
  for (int k = m0 + tid; k < m1; k += kDT) {
    const long long n = pred_idx[k], g = tgt_idx[k];
    if (n < 0 || n >= N || g < 0 || g >= nt_img) continue;
    const int cls = (int)tlabels[t0 + (int)g];
    s_cls[(int)n] = (cls >= 0 && cls < C) ? cls : C;
  }
  __syncthreads();

  // ---- sigmoid focal loss, gamma = 2 (egtr:647-656, dd:2687-2722) --------------------------------------------------
  float fsum = 0.f;
  const float* lg = logits + (size_t)b * N * C;
  float* dl = d_logits + (size_t)b * N * C;
  for (int e = tid; e < N * C; e += kDT) {
    const int n = e / C, c = e - n * C;
    const float x = lg[e];
    const bool t = s_cls[n] == c;
    const float p = 1.f / (1.f + expf(-x));
    const float ce = t ? softplusf(-x) : softplusf(x);            // = -log p_t
    const float pt = t ? p : 1.f - p;
    const float om = 1.f - pt;
    const float a = alpha >= 0.f ? (t ? alpha : 1.f - alpha) : 1.f;
    fsum += a * om * om * ce;
    const float g = a * om * om * (2.f * pt * (-ce) - om);         // d f / d x up to the sign of dp_t/dx
    dl[e] = (t ? g : -g) * inv_num_boxes;
  }

  // ---- cardinality (egtr:661-670): rows whose argmax is not the LAST class; first maximum on ties, like torch.argmax ---
  int cnt = 0;
  for (int n = wave; n < N; n += kDT / 64) {
    float best = -INFINITY;
    int bi = C;
    for (int c = lane; c < C; c += 64) {
      const float v = lg[(size_t)n * C + c];
      if (v > best) { best = v; bi = c; }
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const float ov = __shfl_xor(best, o);
      const int oi = __shfl_xor(bi, o);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    cnt += (bi != C - 1) ? 1 : 0;
  }
  if (lane == 0) s_cnt[wave] = cnt;

  // ---- matched boxes: L1 + generalised IoU (egtr:692-712, dd util generalized_box_iou) -----------------------------------
  float l1sum = 0.f, gsum = 0.f;
  for (int k = m0 + tid; k < m1; k += kDT) {
    if (pred_idx[k] < 0 || pred_idx[k] >= N || tgt_idx[k] < 0 || tgt_idx[k] >= nt_img) continue;
    const int n = (int)pred_idx[k];
    const float4 s = *reinterpret_cast<const float4*>(boxes + ((size_t)b * N + n) * 4);
    const float4 t = *reinterpret_cast<const float4*>(tboxes + (size_t)(t0 + (int)tgt_idx[k]) * 4);
    l1sum += fabsf(s.x - t.x) + fabsf(s.y - t.y) + fabsf(s.z - t.z) + fabsf(s.w - t.w);
    auto sgn = [](float d) { return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); };
    *reinterpret_cast<float4*>(d_l1 + ((size_t)b * N + n) * 4) =
        make_float4(sgn(s.x - t.x) * inv_num_boxes, sgn(s.y - t.y) * inv_num_boxes, sgn(s.z - t.z) * inv_num_boxes,
                    sgn(s.w - t.w) * inv_num_boxes);
    // corners
    const float x0 = s.x - 0.5f * s.z, y0 = s.y - 0.5f * s.w, x1 = s.x + 0.5f * s.z, y1 = s.y + 0.5f * s.w;
    const float u0 = t.x - 0.5f * t.z, v0 = t.y - 0.5f * t.w, u1 = t.x + 0.5f * t.z, v1 = t.y + 0.5f * t.w;
    const float as = (x1 - x0) * (y1 - y0), at = (u1 - u0) * (v1 - v0);
    const float iwr = fminf(x1, u1) - fmaxf(x0, u0), ihr = fminf(y1, v1) - fmaxf(y0, v0);
    const float iw = fmaxf(iwr, 0.f), ih = fmaxf(ihr, 0.f);
    const float inter = iw * ih, uni = as + at - inter;
    const float ew = fmaxf(fmaxf(x1, u1) - fminf(x0, u0), 0.f), eh = fmaxf(fmaxf(y1, v1) - fminf(y0, v0), 0.f);
    const float ea = ew * eh;
    const float giou = inter / uni - (ea - uni) / ea;
    gsum += 1.f - giou;
    // d(iw) / d(x0, x1), d(ew) / d(x0, x1) and the same for y
    const float iw_x1 = (iwr > 0.f && x1 < u1) ? 1.f : 0.f, iw_x0 = (iwr > 0.f && x0 > u0) ? -1.f : 0.f;
    const float ih_y1 = (ihr > 0.f && y1 < v1) ? 1.f : 0.f, ih_y0 = (ihr > 0.f && y0 > v0) ? -1.f : 0.f;
    const float ew_x1 = x1 > u1 ? 1.f : 0.f, ew_x0 = x0 < u0 ? -1.f : 0.f;
    const float eh_y1 = y1 > v1 ? 1.f : 0.f, eh_y0 = y0 < v0 ? -1.f : 0.f;
    auto dgiou = [&](float d_as, float d_inter, float d_ea) {
      const float d_uni = d_as - d_inter;
      return (d_inter * uni - inter * d_uni) / (uni * uni) + (d_uni * ea - uni * d_ea) / (ea * ea);
    };
    const float g_x0 = dgiou(-(y1 - y0), iw_x0 * ih, ew_x0 * eh), g_x1 = dgiou((y1 - y0), iw_x1 * ih, ew_x1 * eh);
    const float g_y0 = dgiou(-(x1 - x0), iw * ih_y0, ew * eh_y0), g_y1 = dgiou((x1 - x0), iw * ih_y1, ew * eh_y1);
    // loss = 1 - giou; corners -> (cx, cy, w, h)
    *reinterpret_cast<float4*>(d_giou + ((size_t)b * N + n) * 4) =
        make_float4(-(g_x0 + g_x1) * inv_num_boxes, -(g_y0 + g_y1) * inv_num_boxes,
                    -0.5f * (g_x1 - g_x0) * inv_num_boxes, -0.5f * (g_y1 - g_y0) * inv_num_boxes);
  }
  // fixed order: the lanes of a wave (butterfly), then the waves in index order
  double fd = (double)fsum, ld = (double)l1sum, gd = (double)gsum;
  for (int o = 32; o > 0; o >>= 1) {
    fd += __shfl_xor(fd, o);
    ld += __shfl_xor(ld, o);
    gd += __shfl_xor(gd, o);
  }
  if (lane == 0) {
    s_red[0][wave] = fd;
    s_red[1][wave] = ld;
    s_red[2][wave] = gd;
  }
  __syncthreads();
  if (tid == 0) {
    double f = 0.0, l = 0.0, g = 0.0;
    for (int i = 0; i < kDT / 64; ++i) { f += s_red[0][i]; l += s_red[1][i]; g += s_red[2][i]; }
    int c = 0;
    for (int w = 0; w < kDT / 64; ++w) c += s_cnt[w];
    out[b * 4 + 0] = (float)(f * inv_num_boxes);
    out[b * 4 + 1] = (float)(l * inv_num_boxes);
    out[b * 4 + 2] = (float)(g * inv_num_boxes);
    out[b * 4 + 3] = (float)c;
  }
}

}  // namespace

extern "C" int egtr_detection_loss_f32(egtr_stream_t stream, const float* logits, const float* pred_boxes,
                                       const int64_t* pred_idx, const int64_t* tgt_idx, const int* match_offsets,
                                       const int64_t* target_labels, const float* target_boxes,
                                       const int* target_offsets, int batch, int num_query, int num_classes,
                                       float focal_alpha, float num_boxes, float* grad_logits, float* grad_boxes_l1,
                                       float* grad_boxes_giou, float* out) {
  if (!logits || !pred_boxes || !pred_idx || !tgt_idx || !match_offsets || !target_labels || !target_boxes ||
      !target_offsets || !grad_logits || !grad_boxes_l1 || !grad_boxes_giou || !out)
    return EGTR_E_ARG;
  if (batch <= 0 || num_query <= 0 || num_classes <= 0 || !(num_boxes > 0.f)) return EGTR_E_ARG;
  if (num_query > kDetMaxN) return EGTR_E_UNSUPPORTED;
  hipLaunchKernelGGL(det_loss_f32, dim3(batch), dim3(kDT), 0, static_cast<hipStream_t>(stream), logits, pred_boxes,
                     reinterpret_cast<const long long*>(pred_idx), reinterpret_cast<const long long*>(tgt_idx),
                     match_offsets, reinterpret_cast<const long long*>(target_labels), target_boxes, target_offsets,
                     num_query, num_classes, focal_alpha, 1.0f / num_boxes, grad_logits, grad_boxes_l1,
                     grad_boxes_giou, out);
  return egtr_check_launch();
}
