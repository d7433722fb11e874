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
constexpr int kRowsPerWg = 8;    // query rows (N*R elements each) per workgroup of the histogram / final passes
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

// ---- prep: per image  tq (query -> target row), wq (1 - sigmoid(matching cost)), mq (matched), block counts ------------
__global__ __launch_bounds__(kLT) void rel_loss_prep(const int64_t* __restrict__ pred_idx,
                                                     const int64_t* __restrict__ tgt_idx,
                                                     const float* __restrict__ match_cost,
                                                     const int* __restrict__ out_off,
                                                     const float* const* __restrict__ target_rel, int N, int R,
                                                     float nonmatching_cost, int k_neg, int k_nm, ImgState* st,
                                                     int* tq, float* wq, unsigned char* mq, int* total_sel) {
  __shared__ int s_cnt[2];
  __shared__ int s_scan[kLT];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int o0 = out_off[b], T = out_off[b + 1] - o0;
  int* tq_b = tq + (size_t)b * N;
  float* wq_b = wq + (size_t)b * N;
  unsigned char* mq_b = mq + (size_t)b * N;
  if (tid < 2) s_cnt[tid] = 0;
  for (int q = tid; q < N; q += kLT) { mq_b[q] = 0; wq_b[q] = 1.0f - 1.0f / (1.0f + expf(-nonmatching_cost)); }
  __syncthreads();
  for (int t = tid; t < T; t += kLT) {
    const int q = (int)pred_idx[o0 + t];
    mq_b[q] = 1;
    tq_b[q] = (int)tgt_idx[o0 + t];
    wq_b[q] = 1.0f - 1.0f / (1.0f + expf(-match_cost[o0 + t]));     // 1 - cost.sigmoid()  (egtr:844, 920)
  }
  __syncthreads();
  // unmatched queries in ascending order take target rows T, T + 1, ... (egtr:761-768)
  int base = 0;
  for (int q0 = 0; q0 < N; q0 += kLT) {
    const int q = q0 + tid;
    const int un = (q < N && !mq_b[q]) ? 1 : 0;
    s_scan[tid] = un;
    __syncthreads();
    for (int o = 1; o < kLT; o <<= 1) {
      const int v = tid >= o ? s_scan[tid - o] : 0;
      __syncthreads();
      s_scan[tid] += v;
      __syncthreads();
    }
    if (un) tq_b[q] = T + base + s_scan[tid] - 1;
    base += s_scan[kLT - 1];
    __syncthreads();
  }
  // counts over the matched block: true = target != 0, false candidates = target != 1 (egtr:838-840)
  const float* rel = target_rel[b];
  int nt = 0, nf = 0;
  const int tot = T * T * R;
  for (int e = tid; e < tot; e += kLT) {
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
  if (s_last && tid == 0) {
    __threadfence();
    double sc = 0.0, sr = 0.0;
    const volatile double* pc = partial_conn;   // written by other workgroups of this launch (released above)
    for (unsigned i = 0; i < gridDim.x; ++i) sc += pc[i];
    for (int i = 0; i < n_partial_rel; ++i) sr += partial_rel[i];
    loss_out[0] = (float)(sr / (double)(*total_sel));   // 0 / 0 = NaN: the reference's mean of an empty tensor
    loss_out[1] = (float)(sc / (double)n_pairs);
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
  hipLaunchKernelGGL(rel_loss_prep, dim3(batch), dim3(kLT), 0, st, pred_idx, tgt_idx, match_cost, out_offsets, target_rel,
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
