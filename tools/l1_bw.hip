// Micro-benchmark: vector-L1 hit bandwidth per CU for 16-byte-per-lane loads, (a) fully coalesced (a wave reads 1 KiB
// contiguous), (b) MSDA-like (each 8-lane group reads a different, L1-resident 128-B line) and (c) like (b) but every
// line has the same offset within its 1 KiB row (one head of the [pixel][8 heads][32 ch] layout).  Build + run on the GPU
// box:  hipcc --offload-arch=gfx950 -O3 tools/l1_bw.hip -o /tmp/l1_bw && /tmp/l1_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void l1_read(const float4* __restrict__ buf, float* __restrict__ out, int iters,
                                               int lines /* 128-B lines in this block's private region */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float4* base = buf + (size_t)blockIdx.x * lines * 8;  // 8 float4 per 128-B line
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  unsigned h = lane * 2654435761u + wave * 97u;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      int idx;
      if (MODE == 0) {
        idx = (((it * 16 + u) * 4 + wave) * 8 % lines) * 8 + lane;  // 8 consecutive lines per wave
        idx = idx % (lines * 8);
      } else {
        h = h * 1664525u + 1013904223u;                               // per-group pseudo-random line
        const unsigned g = __shfl(h, lane & ~7);                      // same line for the 8 lanes of a group
        idx = (int)((g >> 8) % (unsigned)lines) * 8 + (lane & 7);
        if (MODE == 2) idx = ((int)((g >> 8) % (unsigned)(lines / 8)) * 8 + (int)(blockIdx.x & 7)) * 8 + (lane & 7);
      }
      const float4 v = base[idx];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  if (acc.x == 12345.f) out[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

int main() {
  const int blocks = 256 * 2;
  int lines = 96;  // 12 KiB per block: L1-resident (32 KiB per CU, 2 blocks per CU)
  float4* buf; float* out;
  hipMalloc(&buf, (size_t)blocks * 768 * 128);
  hipMalloc(&out, 4096);
  hipMemset(buf, 0, (size_t)blocks * 768 * 128);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 3; ++mode) {
    lines = (mode == 2) ? 768 : 96;  // mode 2: 96 lines of one head out of a 96 KiB region
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(l1_read<0>, dim3(blocks), dim3(256), 0, 0, buf, out, iters, lines);
      else if (mode == 1) hipLaunchKernelGGL(l1_read<1>, dim3(blocks), dim3(256), 0, 0, buf, out, iters, lines);
      else hipLaunchKernelGGL(l1_read<2>, dim3(blocks), dim3(256), 0, 0, buf, out, iters, lines);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * 256 * 16.0 * 16 * iters;
    printf("mode %d (%s): %.3f ms, %.1f TB/s aggregate, %.1f B/clk/CU at 2.4 GHz (%.1f at 2.1 GHz)\n", mode,
           mode == 2 ? "8 random lines per wave, all at the same offset of their 1 KiB row" : mode ? "8 random L1-resident lines per wave" : "coalesced 1 KiB per wave", ms, bytes / ms / 1e9,
           bytes / ms / 1e-3 / 256 / 2.4e9, bytes / ms / 1e-3 / 256 / 2.1e9);
  }
  return 0;
}
