// The XS operand format's stand-alone conversion pass: fp32 row-major -> exact three-way bf16 split in 1 KiB MFMA-operand
// fragments (xs_format.h).  Used once per weight (ops.xs_split, cached per module) by the row-panel kernels of ffn_x6.hip
// (reference layers: model/deformable_detr.py:1049, 1102, 1337-1343); activations are split by the kernels that consume them.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"
#include "xs_format.h"

namespace {

// ---- fp32 row-major -> XS (the stand-alone split pass: layer-0 input of the encoder, weights, tests) -----------------
// One thread per group of 4 consecutive k of one row; `pos` (optional, [pos_rows, K], row r uses pos[r % pos_rows]): also
// writes XS(x + pos).
template <bool RNE>
__global__ __launch_bounds__(256) void split_tile_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ pos,
                                                         int pos_rows, int rows, int K, char* __restrict__ out,
                                                         char* __restrict__ out_pos) {
  const int k4 = K >> 2;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)rows * k4) return;
  const int row = (int)(idx / k4), k = (int)(idx - (long long)row * k4) * 4;
  const float4 v = *reinterpret_cast<const float4*>(x + (size_t)row * ldx + k);
  const size_t off = xs::group_offset(row, k, K >> 4);
  if (out != nullptr) xs::store4<RNE>(out, off, v.x, v.y, v.z, v.w);
  if (out_pos != nullptr) {
    const float4 p = *reinterpret_cast<const float4*>(pos + (size_t)(row % pos_rows) * K + k);
    xs::store4<RNE>(out_pos, off, v.x + p.x, v.y + p.y, v.z + p.z, v.w + p.w);
  }
}

}  // namespace

extern "C" long long egtr_xs_bytes(int rows, int K) {
  if (rows <= 0 || K <= 0 || K % 16) return 0;
  return xs::buffer_bytes(rows, K);
}

extern "C" int egtr_xs_split_f32(egtr_stream_t stream, const float* x, int ldx, const float* pos, int pos_rows, int rows,
                                 int K, void* xs_out, void* xs_pos_out, int round_to_nearest) {
  if (!x || rows <= 0 || K <= 0 || ldx < K || (!xs_out && !xs_pos_out)) return EGTR_E_ARG;
  if (xs_pos_out && (!pos || pos_rows <= 0)) return EGTR_E_ARG;
  if (K % 16 || (ldx & 3) || (reinterpret_cast<uintptr_t>(x) & 15) || (pos && (reinterpret_cast<uintptr_t>(pos) & 15)) ||
      (reinterpret_cast<uintptr_t>(xs_out) & 15) || (reinterpret_cast<uintptr_t>(xs_pos_out) & 15))
    return EGTR_E_UNSUPPORTED;
  const long long n = (long long)rows * (K / 4);
  if ((n + 255) / 256 >= (1ll << 31)) return EGTR_E_UNSUPPORTED;
  const dim3 grid((unsigned)((n + 255) / 256));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (round_to_nearest)
    hipLaunchKernelGGL(split_tile_kernel<true>, grid, dim3(256), 0, st, x, ldx, pos, pos_rows, rows, K,
                       static_cast<char*>(xs_out), static_cast<char*>(xs_pos_out));
  else
    hipLaunchKernelGGL(split_tile_kernel<false>, grid, dim3(256), 0, st, x, ldx, pos, pos_rows, rows, K,
                       static_cast<char*>(xs_out), static_cast<char*>(xs_pos_out));
  return egtr_check_launch();
}
