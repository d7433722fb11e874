// Micro-benchmark: cycles per VALU instruction of ONE wave per SIMD as a function of the number of independent dependency
// chains it interleaves (development probe: the epilogues of the x6 kernels are long single chains as hipcc emits them).
//   hipcc --offload-arch=gfx950 -O3 tools/valu_chain.hip -o build/valu_chain && build/valu_chain
#include <hip/hip_runtime.h>

#include <cstdio>

template <int CH, int KIND>
__global__ __launch_bounds__(256) void probe(float* out, long long* cyc, int iters) {
  float v[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) v[c] = threadIdx.x * 0.001f + c;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 64 / CH; ++u)
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[c]));
        else if (KIND == 1) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(v[c]));
        else asm volatile("v_sub_f32 %0, %0, %0" : "+v"(v[c]));
      }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) s += v[c];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int CH, int KIND>
void run(const char* name, float* out, long long* cyc) {
  const int iters = 200;
  hipLaunchKernelGGL((probe<CH, KIND>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long h;
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-10s chains=%d: %.2f cycles per instruction\n", name, CH, (double)h / (iters * 64.0));
}

int main() {
  float* out;
  long long* cyc;
  hipMalloc(&out, 256 * 256 * 4);
  hipMalloc(&cyc, 8);
  run<1, 0>("v_fma_f32", out, cyc);
  run<2, 0>("v_fma_f32", out, cyc);
  run<4, 0>("v_fma_f32", out, cyc);
  run<8, 0>("v_fma_f32", out, cyc);
  run<1, 1>("v_and_b32", out, cyc);
  run<2, 1>("v_and_b32", out, cyc);
  run<4, 1>("v_and_b32", out, cyc);
  run<1, 2>("v_sub_f32", out, cyc);
  run<2, 2>("v_sub_f32", out, cyc);
  return 0;
}
