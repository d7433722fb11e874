// Probe of ds_read_b64_tr_b16 on gfx950: LDS holds element value = its own element index (as raw u16); every lane reads
// with a chosen address; prints what each lane received.  Build: hipcc --offload-arch=gfx950 tr_probe.hip -o tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void probe(unsigned long long* out, int pitch_elems) {
  __shared__ unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  asm volatile("" ::: "memory");
  __syncthreads();
  const unsigned base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned short*)lds;
  const int l = threadIdx.x;
  // lane l of each 16-lane group g: row = (l & 15) >> 2, 4-element segment = (l & 3); group g -> rows 4 g .. 4 g + 3
  const int g = l >> 4, r = (l & 15) >> 2, seg = l & 3;
  const unsigned addr = base + (unsigned)(((4 * g + r) * pitch_elems + seg * 4) * 2);
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  if (lds[l] == 0xffff) v = 0;   // keeps the LDS image alive
  out[l] = v;
}

int main() {
  unsigned long long* d;
  hipMalloc(&d, 64 * 8);
  const int pitch = 16;
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, pitch);
  unsigned long long h[64];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("pitch %d elements; lane: provided (row, col0) -> received 4 elements as (row, col)\n", pitch);
  for (int l = 0; l < 64; ++l) {
    const int g = l >> 4, r = (l & 15) >> 2, seg = l & 3;
    printf("lane %2d gave (%2d,%2d):", l, 4 * g + r, seg * 4);
    for (int e = 0; e < 4; ++e) {
      const unsigned idx = (unsigned)((h[l] >> (16 * e)) & 0xffff);
      printf(" (%2u,%2u)", idx / pitch, idx % pitch);
    }
    printf("\n");
  }
  return 0;
}
