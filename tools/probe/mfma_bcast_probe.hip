// Probe: A-operand broadcast of v_mfma_f32_4x4x1_16b_f32 (cbsz / abid) on gfx950.
//   cbsz = 4, abid = b: every block multiplies with block b's A values;  cbsz = 3: blocks 0-7 use block abid, 8-15 block 8 + abid.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_bcast_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CBSZ, int ABID>
__global__ void probe(float* out) {
  const int l = threadIdx.x;
  f32x4 z = {0.f, 0.f, 0.f, 0.f};
  f32x4 d = __builtin_amdgcn_mfma_f32_4x4x1f32((float)l, 1.f, z, CBSZ, ABID, 0);
  for (int i = 0; i < 4; ++i) out[i * 64 + l] = d[i];
}

template <int CBSZ, int ABID>
bool run(float* d) {
  hipLaunchKernelGGL((probe<CBSZ, ABID>), dim3(1), dim3(64), 0, 0, d);
  float h[256];
  if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return false;
  bool ok = true;
  for (int i = 0; i < 4; ++i)
    for (int l = 0; l < 64; ++l) {
      const int blk = l / 4, group = CBSZ == 4 ? 0 : (blk / 8) * 8;
      const int src = (group + ABID) * 4 + i;
      if ((int)h[i * 64 + l] != src) ok = false;
    }
  printf("cbsz %d abid %d: D[i][lane] takes A from lane 4*(group + abid) + i: %s\n", CBSZ, ABID, ok ? "CONFIRMED" : "DIFFERENT");
  if (!ok) {
    for (int i = 0; i < 4; ++i) {
      printf("  reg %d:", i);
      for (int l = 0; l < 64; ++l) printf(" %d", (int)h[i * 64 + l]);
      printf("\n");
    }
  }
  return ok;
}

int main() {
  float* d;
  if (hipMalloc(&d, 1024) != hipSuccess) return 2;
  bool ok = run<4, 0>(d);
  ok = run<4, 5>(d) && ok;
  ok = run<4, 15>(d) && ok;
  ok = run<3, 0>(d) && ok;
  ok = run<3, 6>(d) && ok;
  return ok ? 0 : 1;
}
