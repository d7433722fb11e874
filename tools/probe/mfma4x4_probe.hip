// Probe: operand / result layout of v_mfma_f32_4x4x1_16b_f32 on gfx950, and an sc1 (agent-scope, L2-served) 16-byte load.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma4x4_probe.hip -o build/mfma4x4_probe && build/mfma4x4_probe
// Expected (CDNA3 ISA, "4x4x1 16 blocks"): block = lane / 4; A[i] in lane 4*block + i; B[j] in lane 4*block + j;
// D[i][j] of a block: VGPR i, lane 4*block + j.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(float* out) {
  const int l = threadIdx.x;
  f32x4 z = {0.f, 0.f, 0.f, 0.f};
  f32x4 da = __builtin_amdgcn_mfma_f32_4x4x1f32((float)l, 1.f, z, 0, 0, 0);   // which A lane feeds D[reg][lane]
  f32x4 db = __builtin_amdgcn_mfma_f32_4x4x1f32(1.f, (float)l, z, 0, 0, 0);   // which B lane
  for (int i = 0; i < 4; ++i) {
    out[(0 * 4 + i) * 64 + l] = da[i];
    out[(1 * 4 + i) * 64 + l] = db[i];
  }
}

__global__ void sc1_probe(const float4* in, float4* out) {
  float4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(in + threadIdx.x) : "memory");
  out[threadIdx.x] = v;
}

int main() {
  float* d;
  hipMalloc(&d, 8 * 64 * 4);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  float h[8 * 64];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  bool ok = true;
  for (int i = 0; i < 4; ++i)
    for (int l = 0; l < 64; ++l) {
      const int ea = (l / 4) * 4 + i, eb = l;
      if ((int)h[(0 * 4 + i) * 64 + l] != ea || (int)h[(1 * 4 + i) * 64 + l] != eb) ok = false;
    }
  printf("4x4x1 layout D[reg i][lane l] = A[lane 4*(l/4)+i] * B[lane l]: %s\n", ok ? "CONFIRMED" : "DIFFERENT");
  if (!ok) {
    for (int i = 0; i < 4; ++i) {
      printf("reg %d A-lane:", i);
      for (int l = 0; l < 64; ++l) printf(" %d", (int)h[(0 * 4 + i) * 64 + l]);
      printf("\nreg %d B-lane:", i);
      for (int l = 0; l < 64; ++l) printf(" %d", (int)h[(1 * 4 + i) * 64 + l]);
      printf("\n");
    }
  }
  float4 *a, *b;
  hipMalloc(&a, 64 * 16);
  hipMalloc(&b, 64 * 16);
  float4 ha[64];
  for (int i = 0; i < 64; ++i) ha[i] = make_float4(i, i + 0.25f, i + 0.5f, i + 0.75f);
  hipMemcpy(a, ha, sizeof(ha), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(sc1_probe, dim3(1), dim3(64), 0, 0, a, b);
  float4 hb[64];
  hipMemcpy(hb, b, sizeof(hb), hipMemcpyDeviceToHost);
  bool ok2 = true;
  for (int i = 0; i < 64; ++i) ok2 = ok2 && hb[i].x == ha[i].x && hb[i].w == ha[i].w;
  printf("sc1 16-byte load: %s\n", ok2 ? "ok" : "WRONG");
  return ok && ok2 ? 0 : 1;
}
