// Micro-benchmark: what does a barrier between workgroups cost when all of them sit on ONE XCD (one L2) compared with a
// device-wide one?  (Round 2 measured a device-scope hand-off at ~17 us inside the decoder -- about a kernel boundary; a
// persistent decoder-layer kernel confined to one XCD would need only the L2 of that XCD as its meeting point.)
// Workgroups are dealt to the XCDs round-robin by their id, so "blockIdx % 8 == x" selects one XCD.
//   mode 0: participants = every 8th workgroup (one XCD, 32 of them): arrival = an atomic executed in that XCD's L2, polled
//           with agent-scope loads (past the L1, served by the L2), the data hand-over through the L2 as well, no fences;
//   mode 1: the same participants with __threadfence() (device-scope release / acquire) around the barrier;
//   mode 2: 32 participants spread over all 8 XCDs (blockIdx < 32), device-scope fences.
//   hipcc --offload-arch=gfx950 -O3 tools/xcd_barrier.hip -o build/xcd_barrier && build/xcd_barrier
#include <hip/hip_runtime.h>

#include <cstdio>

// poll: sc1 = agent scope = past the CU's L1, served by the XCD's L2; sc0 sc1 = system scope = past the L2 as well
template <bool PAST_L2>
__device__ __forceinline__ unsigned poll(const unsigned* p) {
  unsigned v;
  if (PAST_L2) asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
// arrive: an atomic without scope bits executes in the issuing XCD's L2 (enough when every participant shares that L2); with
// sc1 it is forwarded to the memory side, where all XCDs meet
template <bool PAST_L2>
__device__ __forceinline__ void arrive(unsigned* p) {
  const unsigned one = 1;
  if (PAST_L2) asm volatile("global_atomic_add %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" ::"v"(p), "v"(one) : "memory");
  else asm volatile("global_atomic_add %0, %1, off\n\ts_waitcnt vmcnt(0)" ::"v"(p), "v"(one) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(256) void probe(unsigned* counter, float* data, int iters, long long* cyc, unsigned* xcc, unsigned* fail) {
  const bool part = MODE == 2 ? blockIdx.x < 32 : (blockIdx.x & 7) == 0;
  if (!part) return;
  const int rank = MODE == 2 ? blockIdx.x : blockIdx.x >> 3;
  if (threadIdx.x == 0) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    xcc[rank] = id & 0xf;
  }
  const long long t0 = __builtin_readcyclecounter();
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    // a little work whose result the others read after the barrier
    data[rank * 256 + threadIdx.x] = (float)(it + rank);   // plain store: write-through to the L2
    if (MODE != 0) __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
      arrive<MODE != 0>(counter);
      const unsigned want = 32u * (unsigned)(it + 1);
      int spins = 0;
      while (poll<MODE != 0>(counter) < want && ++spins < 2000000) __builtin_amdgcn_s_sleep(1);
      if (spins >= 2000000) fail[0] = 1;   // never hang the GPU: give up, the host reports it
    }
    __syncthreads();
    if (MODE != 0) __threadfence();
    const int other = (rank + 1 + (it & 15)) & 31;
    float v;
    if (MODE != 0) asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(data + other * 256 + threadIdx.x) : "memory");
    else asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(data + other * 256 + threadIdx.x) : "memory");
    acc += v;
    if (v != (float)(it + other)) atomicAdd(&fail[1], 1u);   // the hand-over must deliver THIS iteration's value
    __syncthreads();                                          // (the next store to `data` must not overtake the readers)
    if (threadIdx.x == 0) {                                   // second barrier: everybody has read
      arrive<MODE != 0>(counter + 16);
      const unsigned want = 32u * (unsigned)(it + 1);
      int spins = 0;
      while (poll<MODE != 0>(counter + 16) < want && ++spins < 2000000) __builtin_amdgcn_s_sleep(1);
      if (spins >= 2000000) fail[0] = 1;
    }
    __syncthreads();
  }
  const long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cyc[rank] = t1 - t0;
  if (acc == -1.f) data[0] = acc;
}

template <int MODE>
void run(const char* name) {
  unsigned *counter, *xcc, *fail;
  float* data;
  long long* cyc;
  hipMalloc(&counter, 128);
  hipMalloc(&xcc, 32 * 4);
  hipMalloc(&fail, 8);
  hipMemset(fail, 0, 8);
  hipMalloc(&data, 32 * 256 * 4);
  hipMalloc(&cyc, 32 * 8);
  const int iters = 200;
  for (int rep = 0; rep < 2; ++rep) {
    hipMemset(counter, 0, 128);
    hipMemset(fail, 0, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(256), 0, 0, counter, data, iters, cyc, xcc, fail);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned hx[32];
    long long hc[32];
    hipMemcpy(hx, xcc, sizeof(hx), hipMemcpyDeviceToHost);
    hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
    unsigned mask = 0;
    for (int i = 0; i < 32; ++i) mask |= 1u << hx[i];
    unsigned hfv[2] = {0, 0};
    hipMemcpy(hfv, fail, 8, hipMemcpyDeviceToHost);
    const unsigned hf = hfv[0];
    if (hfv[1]) printf("%-44s %u STALE values read after the barrier\n", name, hfv[1]);
    if (hf) printf("%-44s barrier never completed (participants do not see each other's arrivals)\n", name);
    else if (rep == 1)
      printf("%-44s %6.2f us per (barrier + hand-off + barrier) (%lld cycles), participants on XCD mask 0x%02x\n", name,
             ms * 1e3 / iters, hc[0] / iters, mask);
  }
}

int main() {
  run<0>("one XCD, L2-local (relaxed atomic, sc0 polls)");
  run<1>("one XCD, device-scope fences");
  run<2>("eight XCDs, device-scope fences");
  return 0;
}
