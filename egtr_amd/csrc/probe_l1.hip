// TEST / MEASUREMENT ONLY (include/egtr_hip_test.h): the vector-L1 gather ceiling the MSDA kernels are priced against.
//
// Every 8-lane group of a wave reads a different, L1-RESIDENT 128-byte line with one global_load_dwordx4 per lane -- the
// access shape of the wave-per-query MSDA gather (8 heads x 128 B per bilinear corner) with every miss taken out: what the
// CU's vector memory path returns per second when nothing but the return path limits it.  bench.py launches it in the same
// run as the MSDA timing and reports `roofline.l1_gather_ceiling_gbs` (until round 5 a constant measured once in round 1,
// profiles/r01_l1_gather_bandwidth.txt: 30.5 TB/s).  Study version with two more access modes: tools/l1_bw.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"
#include "../../include/egtr_hip_test.h"

namespace {
constexpr int kLines = 96;   // 12 KiB per workgroup, two workgroups per CU: L1-resident (32 KiB per CU)

__global__ __launch_bounds__(256) void l1_gather_probe(const float4* __restrict__ buf, float* __restrict__ out, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float4* base = buf + (size_t)blockIdx.x * kLines * 8;   // 8 float4 per 128-byte line
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  unsigned h = lane * 2654435761u + wave * 97u;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      h = h * 1664525u + 1013904223u;                         // per-group pseudo-random line
      const unsigned g = __shfl(h, lane & ~7);                // the same line for the 8 lanes of a group
      const int idx = (int)((g >> 8) % (unsigned)kLines) * 8 + (lane & 7);
      const float4 v = base[idx];
      acc.x += v.x;
      acc.y += v.y;
      acc.z += v.z;
      acc.w += v.w;
    }
  }
  if (acc.x == 12345.f) out[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;   // never true on a zero buffer: keeps the loads
}
}  // namespace

extern "C" long long egtr_test_l1_gather_buffer_bytes(int blocks) { return (long long)blocks * kLines * 128; }

extern "C" int egtr_test_l1_gather_bandwidth(egtr_stream_t stream, const void* buf, float* out, int iters, int blocks,
                                             long long* bytes_moved) {
  if (buf == nullptr || out == nullptr || iters <= 0 || blocks <= 0) return EGTR_E_ARG;
  hipLaunchKernelGGL(l1_gather_probe, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const float4*>(buf), out, iters);
  if (bytes_moved != nullptr) *bytes_moved = (long long)blocks * 256 * 16 * 16 * iters;
  return egtr_check_launch();
}
