// Post-processing routines on the device: the evaluator inputs of the reference.
//
// egtr_bbox_overlaps_f64: the reference's native CPU routine lib/fpn/box_intersections_cpu/bbox.pyx (bbox_overlaps,
// :21-61, and bbox_intersections, :64-108) -- float64, "+1 pixel" box convention, zero where the boxes do not overlap.
// One thread per (box n, query k) pair, the same operation order as the Cython loops, so results are bit-identical
// (IEEE double add / mul / div; no FMA contraction: the products are rounded before they are added).
#include <hip/hip_runtime.h>

#include "common.h"

namespace {

#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void bbox_overlaps_f64(const double* __restrict__ boxes,
                                                         const double* __restrict__ query, int N, int K, int mode,
                                                         double* __restrict__ out) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)N * K) return;
  const int n = (int)(idx / K), k = (int)(idx - (long long)n * K);
  const double bx0 = boxes[n * 4 + 0], by0 = boxes[n * 4 + 1], bx1 = boxes[n * 4 + 2], by1 = boxes[n * 4 + 3];
  const double qx0 = query[k * 4 + 0], qy0 = query[k * 4 + 1], qx1 = query[k * 4 + 2], qy1 = query[k * 4 + 3];
  const double box_area = (qx1 - qx0 + 1) * (qy1 - qy0 + 1);                 // bbox.pyx:44-47
  double r = 0.0;
  const double iw = fmin(bx1, qx1) - fmax(bx0, qx0) + 1;                     // :49-52
  if (iw > 0) {
    const double ih = fmin(by1, qy1) - fmax(by0, qy0) + 1;                   // :54-57
    if (ih > 0) {
      if (mode == 0) {
        const double ua = (bx1 - bx0 + 1) * (by1 - by0 + 1) + box_area - iw * ih;   // :59-63
        r = iw * ih / ua;                                                    // :64
      } else {
        r = iw * ih / box_area;                                              // :107 (bbox_intersections)
      }
    }
  }
  out[idx] = r;
}

}  // namespace

extern "C" int egtr_bbox_overlaps_f64(egtr_stream_t stream, const double* boxes, const double* query_boxes,
                                      int num_boxes, int num_query, int mode, double* out) {
  if (num_boxes < 0 || num_query < 0 || (mode != 0 && mode != 1)) return EGTR_E_ARG;
  if (num_boxes == 0 || num_query == 0) return EGTR_OK;  // empty result, nothing to launch
  if (!boxes || !query_boxes || !out) return EGTR_E_ARG;
  const long long n = (long long)num_boxes * num_query;
  if (n >= (1ll << 31)) return EGTR_E_UNSUPPORTED;
  hipLaunchKernelGGL(bbox_overlaps_f64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), boxes, query_boxes, num_boxes, num_query, mode, out);
  return egtr_check_launch();
}
