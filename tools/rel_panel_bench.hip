// Stand-alone development harness of csrc/rel_panel_x6.hip: launch time of the row-panel relation head on random inputs and,
// in a -DREL_TIMING build, the cycle stamps of the phases of every 64th workgroup (correctness: tests/test_gpu_kernels.py).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DREL_TIMING] tools/rel_panel_bench.hip egtr_amd/csrc/rel_panel_x6.hip \
//         egtr_amd/csrc/gemm_x6.hip egtr_amd/csrc/capi.hip -o build/rel_panel_bench && build/rel_panel_bench [N T R iters]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../include/egtr_hip.h"

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(2);                                                                  \
    }                                                                           \
  } while (0)

extern long long* g_rel_tdbg;

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 200, T = argc > 2 ? atoi(argv[2]) : 7, R = argc > 3 ? atoi(argv[3]) : 50;
  const int iters = argc > 4 ? atoi(argv[4]) : 100;
  std::mt19937 rng(5);
  std::normal_distribution<float> nd(0.f, 1.f);
  auto mk = [&](size_t n, float sc) {
    std::vector<float> h(n);
    for (auto& v : h) v = nd(rng) * sc;
    float* d;
    CK(hipMalloc(&d, n * 4));
    CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    return d;
  };
  float *gq = mk((size_t)N * T, 1), *gk = mk((size_t)N * T, 1), *uq = mk((size_t)N * T * 512, .5f), *uk = mk((size_t)N * T * 512, .5f);
  float *b1 = mk(512, .1f), *w2r = mk(65536, 1 / 16.f), *b2r = mk(256, .1f), *w3r = mk(64 * 256, 1 / 16.f), *b3r = mk(64, .1f);
  float *w2c = mk(65536, 1 / 16.f), *b2c = mk(256, .1f), *w3c = mk(256, 1 / 16.f), *b3c = mk(1, .1f);
  void *x2r, *x2c, *x3;
  CK(hipMalloc(&x2r, egtr_xs_bytes(256, 256)));
  CK(hipMalloc(&x2c, egtr_xs_bytes(256, 256)));
  CK(hipMalloc(&x3, egtr_xs_bytes(64, 256)));
  if (egtr_xs_split_f32(nullptr, w2r, 256, nullptr, 0, 256, 256, x2r, nullptr, 1) ||
      egtr_xs_split_f32(nullptr, w2c, 256, nullptr, 0, 256, 256, x2c, nullptr, 1) ||
      egtr_xs_split_f32(nullptr, w3r, 256, nullptr, 0, 64, 256, x3, nullptr, 1))
    return 2;
  float *rel, *conn;
  CK(hipMalloc(&rel, (size_t)N * N * R * 4));
  CK(hipMalloc(&conn, (size_t)N * N * 4));
  auto run = [&]() {
    return egtr_rel_head_forward_panel_x6_f32(nullptr, gq, gk, uq, uk, b1, x2r, b2r, x3, b3r, x2c, b2c, w3c, b3c, nullptr,
                                              nullptr, 1, N, T, 256, R, 0, rel, conn, nullptr, 0);
  };
  int rc = run();
  CK(hipDeviceSynchronize());
  if (rc) {
    fprintf(stderr, "rc %d (%s)\n", rc, egtr_last_hip_error());
    return 2;
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) run();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const int tn = (N + 7) / 8;
  printf("N=%d T=%d R=%d: %d panels, %.2f us per launch\n", N, T, R, 2 * tn * tn, ms * 1e3 / iters);
  long long* td;
  const int nwg = 2 * tn * tn, nrec = (nwg + 63) / 64;
  CK(hipMalloc(&td, (1024 + 64) * 8));
  CK(hipMemset(td, 0, (1024 + 64) * 8));
  g_rel_tdbg = td;
  run();
  CK(hipDeviceSynchronize());
  std::vector<long long> h(1024 + 64);
  CK(hipMemcpy(h.data(), td, h.size() * 8, hipMemcpyDeviceToHost));
  if (h[6]) {
    printf("cycle stamps of workgroups 0, 64, ..: start(rel. to wg 0) | build | sync | prologue | main loop | drain | epilogue\n");
    for (int k = 0; k < nrec; ++k) {
      const long long* t = &h[k * 16];
      printf("  wg %4d %s: %8lld | %6lld %6lld %6lld %7lld %6lld %6lld\n", k * 64, k * 64 < tn * tn ? "rel " : "conn", t[0] - h[0],
             t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[6] - t[5]);
      printf("          build: to build start %6lld | gates %6lld | slot 0 landed %6lld | slot 0 math %6lld | slots 1.. %6lld | finalize %6lld\n",
             t[8] - t[0], t[9] - t[8], t[11] - t[9], t[12] - t[11], t[10] - t[12], t[1] - t[10]);
    }
  }
  if (h[1024]) {
    printf("stages of workgroup 0, first chunk pair: DMA wait | lgkm + barrier | MFMA stage | tail (chunk epilogue) \n");
    for (int s = 0; s < 10; ++s) {
      const long long* t = &h[1024 + 4 * s];
      printf("  stage %d: %5lld %5lld %5lld %5lld\n", s, t[1] - t[0], t[2] - t[1], t[3] - t[2], (s < 9 ? t[4] : t[3]) - t[3]);
    }
  }
  return 0;
}
