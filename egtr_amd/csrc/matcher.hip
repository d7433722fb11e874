// Hungarian matcher on the device: cost matrix + linear sum assignment in ONE launch, no host round trip.
//
// Reference: DeformableDetrHungarianMatcher.forward (model/deformable_detr.py:2925-3015) builds the focal / L1 / GIoU cost
// matrix with ~15 tensor ops, copies it to the host (":2985 .cpu()", a device synchronisation per training step, Ld times
// with auxiliary losses) and calls scipy.optimize.linear_sum_assignment per image.  Here one workgroup per image
//   1. evaluates its [N, T_b] block of the cost matrix in fp32 with the reference's operation order (no FMA contraction;
//      sigmoid / log / the adaptive-smoothing offset exactly as written at :2949-2999) and keeps it in LDS as float64 --
//      scipy converts the float32 matrix to double before solving;
//   2. solves the assignment with the algorithm scipy implements (Crouse's shortest augmenting path, restated and pinned
//      against scipy in oracle/lsa.py): same transposition rule, same reverse-ordered "remaining" list, same tie rule,
//      same float64 operation order for the reduced costs and the dual updates -- so the indices are those scipy returns,
//      ties included.  The scan over the remaining columns, which is the inner loop, is spread over the 64 lanes of one
//      wave (columns j, j+64, ...) and finished with a lexicographic wave reduction (cost, unassigned first, position in
//      the remaining list); everything sequential in the algorithm (the path, the augmentation) stays sequential;
//   3. writes (query index, target index, matching cost) triples sorted by query index, which is the order scipy reports
//      for a transposed problem.
// LDS: nr * nc doubles for the matrix (nr = min(N, T), nc = max(N, T)): 48 KB at N = 200, T = 30; up to ~150 KB.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "common.h"

namespace {

#pragma clang fp contract(off)

constexpr int kMT = 256;          // threads per workgroup
constexpr int kMaxSide = 1024;    // max(N, T)
constexpr int kLdsBytes = 160 * 1024 - 1024;

struct Best {
  double val;
  int score;   // unassigned: 0x40000000 + pos (later position wins), assigned: 0x3fffffff - pos (earlier wins)
  int j;
};

__device__ __forceinline__ bool better(const Best& a, const Best& b) {   // a strictly preferred to b
  return a.val < b.val || (a.val == b.val && a.score > b.score);
}

__device__ __forceinline__ Best shfl_best(const Best& x, int mask) {
  Best r;
  r.val = __shfl_xor(x.val, mask);
  r.score = __shfl_xor(x.score, mask);
  r.j = __shfl_xor(x.j, mask);
  return r;
}

// cost of (query n, target t) in fp32, operation order of dd:2949-2982 (+ :2989-2999 when `smooth`)
__device__ __forceinline__ float pair_cost(const float* __restrict__ logit_row, const float* __restrict__ box,
                                           long long cls, const float* __restrict__ tb, float w_class, float w_bbox,
                                           float w_giou, int smooth, float cost_min, float inv_sig) {
  const float x = logit_row[cls];
  const float p = 1.0f / (1.0f + expf(-x));                                   // .sigmoid()
  const float neg = (0.75f * (p * p)) * (-logf((1.0f - p) + 1e-8f));           // (1 - alpha) * p**2 * -(1 - p + 1e-8).log()
  const float q = 1.0f - p;
  const float pos = (0.25f * (q * q)) * (-logf(p + 1e-8f));                    // alpha * (1 - p)**2 * -(p + 1e-8).log()
  const float class_cost = pos - neg;
  // torch.cdist(p = 1)
  const float l1 = ((fabsf(box[0] - tb[0]) + fabsf(box[1] - tb[1])) + fabsf(box[2] - tb[2])) + fabsf(box[3] - tb[3]);
  // generalized_box_iou(center_to_corners_format(.), center_to_corners_format(.))  (model/util.py:89-124)
  const float ax0 = box[0] - 0.5f * box[2], ay0 = box[1] - 0.5f * box[3], ax1 = box[0] + 0.5f * box[2],
              ay1 = box[1] + 0.5f * box[3];
  const float bx0 = tb[0] - 0.5f * tb[2], by0 = tb[1] - 0.5f * tb[3], bx1 = tb[0] + 0.5f * tb[2],
              by1 = tb[1] + 0.5f * tb[3];
  const float area1 = (ax1 - ax0) * (ay1 - ay0), area2 = (bx1 - bx0) * (by1 - by0);
  const float iw = fmaxf(fminf(ax1, bx1) - fmaxf(ax0, bx0), 0.f), ih = fmaxf(fminf(ay1, by1) - fmaxf(ay0, by0), 0.f);
  const float inter = iw * ih;
  const float uni = (area1 + area2) - inter;
  const float iou = inter / uni;
  const float ew = fmaxf(fmaxf(ax1, bx1) - fminf(ax0, bx0), 0.f), eh = fmaxf(fmaxf(ay1, by1) - fminf(ay0, by0), 0.f);
  const float earea = ew * eh;
  const float giou = iou - (earea - uni) / earea;
  float c = (w_bbox * l1 + w_class * class_cost) + w_giou * (-giou);            // dd:2976-2980
  if (smooth) c = (c - cost_min) + inv_sig;                                      // dd:2999
  return c;
}

// One workgroup per image.  Outputs are packed: image b owns entries [out_off[b], out_off[b] + min(N, T_b)).
// MGLOBAL: the float64 matrix does not fit in LDS and lives in caller-provided scratch (image b at N * tgt_off[b]).
template <bool MGLOBAL>
__global__ __launch_bounds__(kMT) void hungarian_match_f32(
    const float* __restrict__ logits, const float* __restrict__ boxes, const int64_t* __restrict__ tgt_ids,
    const float* __restrict__ tgt_boxes, const int* __restrict__ tgt_off, const int* __restrict__ out_off, int N, int K,
    float w_class, float w_bbox, float w_giou, int smooth, float cost_min, float inv_sig,
    int64_t* __restrict__ pred_idx, int64_t* __restrict__ tgt_idx, float* __restrict__ match_cost,
    float* __restrict__ cost_out, const float* __restrict__ cost_in, int* __restrict__ status,
    double* __restrict__ scratch) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int t0 = tgt_off[b], T = tgt_off[b + 1] - t0;
  const int o0 = out_off[b];
  if (T <= 0 || N <= 0) return;
  const bool transpose = T < N;                     // scipy: "tall rectangular cost matrix must be transposed"
  const int nr = transpose ? T : N, nc = transpose ? N : T;
  // LDS carve-up
  double* M = MGLOBAL ? scratch + (size_t)N * t0 : reinterpret_cast<double*>(s_raw);   // [nr][nc]
  double* u = reinterpret_cast<double*>(s_raw) + (MGLOBAL ? 0 : (size_t)nr * nc);       // [nr]
  double* v = u + nr;                                          // [nc]
  double* spc = v + nc;                                        // [nc]
  int* path = reinterpret_cast<int*>(spc + nc);                // [nc]
  int* row4col = path + nc;                                    // [nc]
  int* pos = row4col + nc;                                     // [nc] position of column j in `remaining`
  int* remaining = pos + nc;                                   // [nc]
  int* col4row = remaining + nc;                               // [nr]
  int* flagR = col4row + nr;                                   // [nr] SR
  int* flagC = flagR + nr;                                     // [nc] SC
  int* s_ctl = flagC + nc;                                     // [4]

  // ---- 1. cost block (all threads) ----------------------------------------------------------------------------------
  int bad = 0;
  for (int e = tid; e < N * T; e += kMT) {
    const int n = e / T, t = e - n * T;
    float c;
    if (cost_in != nullptr) {
      c = cost_in[(size_t)N * t0 + e];              // tests: solve a given matrix ([N, T_b] blocks packed by image)
    } else {
      c = pair_cost(logits + ((size_t)b * N + n) * K, boxes + ((size_t)b * N + n) * 4, tgt_ids[t0 + t],
                    tgt_boxes + (size_t)(t0 + t) * 4, w_class, w_bbox, w_giou, smooth, cost_min, inv_sig);
    }
    if (cost_out != nullptr) cost_out[(size_t)N * t0 + e] = c;
    if (c != c || c == -INFINITY) bad = 1;           // scipy: "matrix contains invalid numeric entries"
    if (transpose) M[(size_t)t * nc + n] = (double)c; else M[(size_t)n * nc + t] = (double)c;
  }
  for (int i = tid; i < nr; i += kMT) { u[i] = 0.0; col4row[i] = -1; }
  for (int j = tid; j < nc; j += kMT) { v[j] = 0.0; row4col[j] = -1; path[j] = -1; }
  if (tid == 0) s_ctl[0] = 0;
  __syncthreads();
  if (bad) atomicOr(&s_ctl[0], 1);
  __syncthreads();
  if (s_ctl[0]) {   // no assignment for this image: indices -1 (the host wrapper reports it when asked)
    if (tid == 0 && status != nullptr) status[b] = 1;
    for (int i = tid; i < nr; i += kMT) { pred_idx[o0 + i] = -1; tgt_idx[o0 + i] = -1; match_cost[o0 + i] = 0.f; }
    return;
  }

  // ---- 2. assignment: wave 0 ----------------------------------------------------------------------------------------
  if (tid < 64) {
    for (int cur = 0; cur < nr; ++cur) {
      for (int j = lane; j < nc; j += 64) {
        spc[j] = INFINITY;
        flagC[j] = 0;
        remaining[j] = nc - j - 1;
        pos[nc - j - 1] = j;                         // column (nc - j - 1) sits at position j
      }
      for (int i = lane; i < nr; i += 64) flagR[i] = 0;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      double minVal = 0.0;
      int num_remaining = nc;
      int i = cur, sink = -1;
      while (sink == -1) {
        if (lane == 0) flagR[i] = 1;
        const double ui = u[i];
        const double* Mi = M + (size_t)i * nc;
        Best best;
        best.val = INFINITY;
        best.score = -1;
        best.j = -1;
        for (int j = lane; j < nc; j += 64) {
          if (!flagC[j]) {
            const double r = ((minVal + Mi[j]) - ui) - v[j];
            double s = spc[j];
            if (r < s) {
              path[j] = i;
              spc[j] = r;
              s = r;
            }
            Best c;
            c.val = s;
            c.score = (row4col[j] == -1) ? (0x40000000 + pos[j]) : (0x3fffffff - pos[j]);
            c.j = j;
            // the sequential scan starts from lowest = +inf with a strict "<": an infinite candidate is never chosen
            if (s < INFINITY && (best.j < 0 || better(c, best))) best = c;
          }
        }
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) {
          const Best o = shfl_best(best, m);
          if (o.j >= 0 && (best.j < 0 || better(o, best))) best = o;
        }
        if (best.j < 0) {                            // infeasible (every remaining reduced cost is +inf)
          sink = -2;
          break;
        }
        minVal = best.val;
        const int j = best.j;
        const int r4c = row4col[j];
        if (lane == 0) {
          flagC[j] = 1;
          const int index = pos[j];
          const int last = remaining[num_remaining - 1];
          remaining[index] = last;
          pos[last] = index;
        }
        --num_remaining;
        if (r4c == -1) sink = j; else i = r4c;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      if (sink < 0) {
        if (lane == 0) { s_ctl[0] = 2; if (status != nullptr) status[b] = 2; }
        break;
      }
      // dual update
      for (int i2 = lane; i2 < nr; i2 += 64) {
        if (i2 == cur) u[i2] = u[i2] + minVal;
        else if (flagR[i2]) u[i2] = u[i2] + (minVal - spc[col4row[i2]]);
      }
      for (int j2 = lane; j2 < nc; j2 += 64)
        if (flagC[j2]) v[j2] = v[j2] - (minVal - spc[j2]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // augment the previous solution along the path (sequential)
      if (lane == 0) {
        int j = sink;
        while (true) {
          const int i2 = path[j];
          row4col[j] = i2;
          const int tmp = col4row[i2];
          col4row[i2] = j;
          j = tmp;
          if (i2 == cur) break;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  __syncthreads();
  if (s_ctl[0]) {
    for (int i = tid; i < nr; i += kMT) { pred_idx[o0 + i] = -1; tgt_idx[o0 + i] = -1; match_cost[o0 + i] = 0.f; }
    return;
  }

  // ---- 3. outputs, sorted by query index ----------------------------------------------------------------------------
  if (tid == 0 && status != nullptr) status[b] = 0;
  for (int i = tid; i < nr; i += kMT) {
    const int c = col4row[i];
    if (transpose) {       // row i = target i, c = its query: rank by query index (all distinct)
      int rank = 0;
      for (int k = 0; k < nr; ++k) rank += col4row[k] < c;
      pred_idx[o0 + rank] = c;
      tgt_idx[o0 + rank] = i;
      match_cost[o0 + rank] = (float)M[(size_t)i * nc + c];
    } else {               // row i = query i
      pred_idx[o0 + i] = i;
      tgt_idx[o0 + i] = c;
      match_cost[o0 + i] = (float)M[(size_t)i * nc + c];
    }
  }
}

size_t lds_bytes(int N, int T, bool with_matrix) {
  const size_t nr = (size_t)(T < N ? T : N), nc = (size_t)(T < N ? N : T);
  return (with_matrix ? nr * nc * 8 : 0) + (nr + 2 * nc) * 8 + (5 * nc + 2 * nr + 4) * 4 + 16;
}

}  // namespace

extern "C" long long egtr_hungarian_match_scratch_doubles(int num_query, int max_targets, long long total_targets) {
  if (num_query <= 0 || max_targets <= 0) return 0;
  return lds_bytes(num_query, max_targets, true) > (size_t)kLdsBytes ? (long long)num_query * total_targets : 0;
}

extern "C" int egtr_hungarian_match_f32(egtr_stream_t stream, const float* logits, const float* boxes,
                                        const int64_t* tgt_ids, const float* tgt_boxes, const int* tgt_offsets,
                                        const int* out_offsets, int batch, int num_query, int num_logits,
                                        int max_targets, float class_cost, float bbox_cost, float giou_cost,
                                        int smoothing, float cost_min, float inverse_sigmoid_smoothing,
                                        int64_t* pred_idx, int64_t* tgt_idx, float* match_cost, float* cost_out,
                                        const float* cost_in, int* status, double* scratch) {
  if (!tgt_offsets || !out_offsets || !pred_idx || !tgt_idx || !match_cost) return EGTR_E_ARG;
  if (cost_in == nullptr && (!logits || !boxes || !tgt_ids || !tgt_boxes)) return EGTR_E_ARG;
  if (batch <= 0 || num_query <= 0 || max_targets < 0 || (cost_in == nullptr && num_logits <= 0)) return EGTR_E_ARG;
  if (max_targets == 0) return EGTR_OK;
  if (num_query > kMaxSide || max_targets > kMaxSide) return EGTR_E_UNSUPPORTED;
  const bool mglobal = lds_bytes(num_query, max_targets, true) > (size_t)kLdsBytes;
  if (mglobal && scratch == nullptr) return EGTR_E_ARG;   // egtr_hungarian_match_scratch_doubles() says how much
  const size_t lds = lds_bytes(num_query, max_targets, !mglobal);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (mglobal) {
    hipLaunchKernelGGL(hungarian_match_f32<true>, dim3(batch), dim3(kMT), lds, st, logits, boxes, tgt_ids, tgt_boxes,
                       tgt_offsets, out_offsets, num_query, num_logits, class_cost, bbox_cost, giou_cost, smoothing,
                       cost_min, inverse_sigmoid_smoothing, pred_idx, tgt_idx, match_cost, cost_out, cost_in, status,
                       scratch);
    return egtr_check_launch();
  }
  if (lds > 64 * 1024) {
    static unsigned long long lds_raised = 0;   // dynamic LDS above 64 KB has to be requested once per device
    if (int rc = egtr_raise_dynamic_lds(reinterpret_cast<const void*>(hungarian_match_f32<false>), kLdsBytes, &lds_raised))
      return rc;
  }
  hipLaunchKernelGGL(hungarian_match_f32<false>, dim3(batch), dim3(kMT), lds, st, logits, boxes, tgt_ids, tgt_boxes,
                     tgt_offsets, out_offsets, num_query, num_logits, class_cost, bbox_cost, giou_cost, smoothing,
                     cost_min, inverse_sigmoid_smoothing, pred_idx, tgt_idx, match_cost, cost_out, cost_in, status,
                     nullptr);
  return egtr_check_launch();
}
