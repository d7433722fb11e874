// Stand-alone development harness of the encoder-tail kernel (csrc/ffn_x6.hip, ffn_x6_kernel<true>): launch time on random
// inputs (correctness: tests/test_gpu_pinning.py; the cycle-stamp instrumentation of round 3 is in the history, 7146666).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/enc_tail_bench.hip egtr_amd/csrc/ffn_x6.hip \
//         egtr_amd/csrc/xs_split.hip egtr_amd/csrc/capi.hip -o build/enc_tail_bench && build/enc_tail_bench [M F iters with_pos]
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


int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 12537, F = argc > 2 ? atoi(argv[2]) : 1024, iters = argc > 3 ? atoi(argv[3]) : 100;
  const int with_pos = argc > 4 ? atoi(argv[4]) : 0;   // the model runs it without the position output since the lazy-pos GEMM
  const int D = 256;
  std::mt19937 rng(4);
  std::normal_distribution<float> nd(0.f, 1.f);
  auto mk = [&](size_t n, float sc, float off = 0.f) {
    std::vector<float> h(n);
    for (auto& v : h) v = off + nd(rng) * sc;
    float* d;
    CK(hipMalloc(&d, n * 4));
    CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    return d;
  };
  float *ctx = mk((size_t)M * D, 1), *hid = mk((size_t)M * D, 1), *pos = mk((size_t)M * D, 1);
  float *wp = mk(D * D, 1 / 16.f), *bp = mk(D, .3f), *g1 = mk(D, .2f, 1.f), *be1 = mk(D, .2f);
  float *w1 = mk((size_t)F * D, 1 / 16.f), *b1 = mk(F, .3f), *w2 = mk((size_t)D * F, 1 / 32.f), *b2 = mk(D, .3f);
  float *g2 = mk(D, .2f, 1.f), *be2 = mk(D, .2f);
  void *xp, *x1, *x2;
  CK(hipMalloc(&xp, egtr_xs_bytes(D, D)));
  CK(hipMalloc(&x1, egtr_xs_bytes(F, D)));
  CK(hipMalloc(&x2, egtr_xs_bytes(D, F)));
  if (egtr_xs_split_f32(nullptr, wp, D, nullptr, 0, D, D, xp, nullptr, 1) ||
      egtr_xs_split_f32(nullptr, w1, D, nullptr, 0, F, D, x1, nullptr, 1) ||
      egtr_xs_split_f32(nullptr, w2, F, nullptr, 0, D, F, x2, nullptr, 1))
    return 2;
  float *out, *outp;
  CK(hipMalloc(&out, (size_t)M * D * 4));
  CK(hipMalloc(&outp, (size_t)M * D * 4));
  auto run = [&]() {
    return egtr_encoder_tail_x6_f32(nullptr, ctx, D, hid, D, xp, bp, g1, be1, 1e-5f, x1, b1, x2, b2, g2, be2, 1e-5f, with_pos ? pos : nullptr, M, out,
                                    with_pos ? outp : nullptr, M, D, F);
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
  printf("M=%d F=%d: %.2f us per launch\n", M, F, ms * 1e3 / iters);
  return 0;
}
