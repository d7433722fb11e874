// Micro-benchmark: how fast can a CU pull L2-resident operand fragments into LDS?  (development probe for the LDS-DMA rings of csrc/ffn_x6.hip / tools/gemm_x6.hip)
//   mode 0: global_load_lds_dwordx4 (LDS-DMA), NL instructions per wave in flight, counted vmcnt
//   mode 1: global_load_dwordx4 into registers + ds_write_b128
//   mode 2: global_load_dwordx4 into registers only (no LDS write)
// Every workgroup (256 threads) walks `iters` stages of NL KiB per wave over a window of a `buf_mb` MiB buffer.
//   hipcc --offload-arch=gfx950 -O3 tools/dma_bw.hip -o build/dma_bw && build/dma_bw
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(2);                                                                  \
    }                                                                           \
  } while (0)

typedef __attribute__((address_space(3))) char lds_char;
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

template <int MODE, int NL>
__global__ __launch_bounds__(256) void probe(const char* __restrict__ buf, size_t buf_bytes, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds_base = (unsigned)reinterpret_cast<uintptr_t>((lds_char*)smem);
  // window: workgroup b starts at b * 12 KiB (neighbours overlap like the tiles of a GEMM row), stage = 4 * NL KiB
  size_t pos = ((size_t)blockIdx.x * 12288) % (buf_bytes - (size_t)4 * NL * 1024 * 4);
  const size_t stage = (size_t)4 * NL * 1024;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    const char* src = buf + pos + (size_t)wave * NL * 1024 + lane * 16;
    const unsigned slot = (unsigned)(it & 1) * (unsigned)stage;
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < NL; ++i) dma16(src + i * 1024, lds_base + slot + (wave * NL + i) * 1024);
      asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NL) : "memory");   // the previous stage has landed
    } else {
      f32x4 r[NL];
#pragma unroll
      for (int i = 0; i < NL; ++i) r[i] = *reinterpret_cast<const f32x4*>(src + i * 1024);
      if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < NL; ++i) *reinterpret_cast<f32x4*>(smem + slot + (wave * NL + i) * 1024 + lane * 16) = r[i];
      } else {
#pragma unroll
        for (int i = 0; i < NL; ++i) acc += r[i];
      }
    }
    pos += stage;
    if (pos + stage * 2 > buf_bytes) pos = 0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (MODE != 2) acc = *reinterpret_cast<const f32x4*>(smem + threadIdx.x * 16);
  if (acc.x == 12345.f) sink[0] = acc.x + acc.y + acc.z + acc.w;
}

template <int MODE, int NL>
void run(const char* name, const char* buf, size_t bytes, int wgs, int iters, float* sink) {
  const int lds = 2 * 4 * NL * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE, NL>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((probe<MODE, NL>), dim3(wgs), dim3(256), lds, 0, buf, bytes, iters, sink);
  CK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((probe<MODE, NL>), dim3(wgs), dim3(256), lds, 0, buf, bytes, iters, sink);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / 5, total = (double)wgs * iters * 4 * NL * 1024;
  printf("%-26s NL=%d wgs=%4d buf=%6.1f MiB: %8.1f us  %6.2f TB/s  %5.1f B/clk/CU @2.4GHz\n", name, NL, wgs,
         bytes / 1048576.0, us, total / us * 1e-6, total / us * 1e-6 * 1e12 / 256 / 2.4e9);
}

int main() {
  const size_t cap = 64u << 20;
  char* buf;
  float* sink;
  CK(hipMalloc(&buf, cap));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(buf, 1, cap));
  for (size_t mb : {2u, 16u, 64u}) {
    const size_t bytes = mb << 20;
    for (int wgs : {256, 512, 1024}) {
      run<0, 6>("LDS-DMA", buf, bytes, wgs, 400, sink);
      run<1, 6>("regs + ds_write_b128", buf, bytes, wgs, 400, sink);
      run<2, 6>("regs only", buf, bytes, wgs, 400, sink);
    }
  }
  run<0, 12>("LDS-DMA", buf, 2u << 20, 512, 400, sink);
  run<0, 3>("LDS-DMA", buf, 2u << 20, 512, 400, sink);
  run<0, 3>("LDS-DMA", buf, 2u << 20, 2048, 400, sink);
  return 0;
}
