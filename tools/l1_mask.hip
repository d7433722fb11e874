// Micro-benchmark: what a 16-byte-per-lane gather instruction costs in the vector L1 / texture addresser when
//   (a) only some of its eight 8-lane groups are active (exec-masked load), and
//   (b) several 8-lane groups read the SAME 128-B line (duplicates inside one instruction).
// This decides whether de-duplicating the bilinear corners of x-adjacent MSDA queries inside a wave (masked loads +
// cross-lane moves) can lower the L1 time of the gather.  Every line is L1-resident (12 KiB per workgroup).
//   hipcc --offload-arch=gfx950 -O3 tools/l1_mask.hip -o /tmp/l1_mask && /tmp/l1_mask
#include <hip/hip_runtime.h>
#include <cstdio>

// ACTIVE: number of 8-lane groups that execute the load (1..8).  DISTINCT: number of distinct lines among the active
// groups (groups g and g' read the same line when g % DISTINCT == g' % DISTINCT).
template <int ACTIVE, int DISTINCT>
__global__ __launch_bounds__(256) void l1_read(const float4* __restrict__ buf, float* __restrict__ out, int iters,
                                               int lines) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float4* base = buf + (size_t)blockIdx.x * lines * 8;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  unsigned h = (lane & ~7) * 2654435761u + wave * 97u + 12345u;
  const int grp = lane >> 3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      h = h * 1664525u + 1013904223u;
      const unsigned g = __shfl(h, (grp % DISTINCT) * 8);  // same line for the groups that share
      const int idx = (int)((g >> 8) % (unsigned)lines) * 8 + (lane & 7);
      if (grp < ACTIVE) {
        const float4 v = base[idx];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
  }
  if (acc.x == 12345.f) out[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int ACTIVE, int DISTINCT>
static void run(const float4* buf, float* out, const char* what) {
  const int blocks = 512, lines = 96, iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((l1_read<ACTIVE, DISTINCT>), dim3(blocks), dim3(256), 0, 0, buf, out, iters, lines);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  const double instr = (double)blocks * 4 * 16.0 * iters;                 // wave-level load instructions
  const double cyc = best * 1e-3 * 2.1e9 / (instr / 256.0);               // per instruction per CU at ~2.1 GHz
  printf("%-58s %8.3f ms  %6.2f CU-cycles per wave-instruction (2.1 GHz)  useful %.1f TB/s\n", what, best, cyc,
         instr * ACTIVE * 128.0 / best / 1e9);
}

int main() {
  float4* buf;
  float* out;
  hipMalloc(&buf, (size_t)512 * 96 * 128);
  hipMalloc(&out, 4096);
  hipMemset(buf, 0, (size_t)512 * 96 * 128);
  run<8, 8>(buf, out, "8 groups active, 8 distinct lines (baseline)");
  run<4, 4>(buf, out, "4 groups active (exec mask), 4 distinct lines");
  run<2, 2>(buf, out, "2 groups active, 2 distinct lines");
  run<1, 1>(buf, out, "1 group active, 1 line");
  run<8, 4>(buf, out, "8 groups active, 4 distinct lines (pairs share)");
  run<8, 2>(buf, out, "8 groups active, 2 distinct lines");
  run<8, 1>(buf, out, "8 groups active, all read the same line");
  return 0;
}
