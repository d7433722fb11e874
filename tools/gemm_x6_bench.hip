// Stand-alone development harness of tools/gemm_x6.hip (no Python, no torch): builds XS operands with the device split
// pass, runs egtr_gemm_x6_f32, checks sampled outputs against a float64 host product and times the launch with HIP events.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_x6_bench.hip tools/gemm_x6.hip egtr_amd/csrc/xs_split.hip egtr_amd/csrc/capi.hip \
//         -o gpurun_out/gemm_x6_bench && gpurun_out/gemm_x6_bench [M K N relu xs_out iters]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../include/egtr_hip.h"

extern "C" long long egtr_xs_bytes(int rows, int K);
extern "C" int egtr_xs_split_f32(egtr_stream_t, const float*, int, const float*, int, int, int, void*, void*, int);
extern "C" int egtr_gemm_x6_f32(egtr_stream_t, int, const void* const*, const void* const*, const float* const*,
                                float* const*, const int*, void* const*, const int*, const int*, int, int);

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
      exit(2);                                                                     \
    }                                                                              \
  } while (0)

static float bf16_bits_to_float(unsigned short b) {
  unsigned u = (unsigned)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 12537, K = argc > 2 ? atoi(argv[2]) : 256, N = argc > 3 ? atoi(argv[3]) : 1024;
  const int relu = argc > 4 ? atoi(argv[4]) : 1, xs_out = argc > 5 ? atoi(argv[5]) : 0, iters = argc > 6 ? atoi(argv[6]) : 200;
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hb(N);
  for (auto& v : hA) v = nd(rng);
  for (auto& v : hW) v = nd(rng) / std::sqrt((float)K);
  for (auto& v : hb) v = nd(rng);
  float *dA, *dW, *db, *dC;
  void *xa, *xw, *xc;
  CK(hipMalloc(&dA, hA.size() * 4));
  CK(hipMalloc(&dW, hW.size() * 4));
  CK(hipMalloc(&db, N * 4));
  CK(hipMalloc(&dC, (size_t)M * N * 4));
  CK(hipMalloc(&xa, egtr_xs_bytes(M, K)));
  CK(hipMalloc(&xw, egtr_xs_bytes(N, K)));
  CK(hipMalloc(&xc, egtr_xs_bytes(M, N)));
  CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dC, 0xff, (size_t)M * N * 4));
  if (egtr_xs_split_f32(nullptr, dA, K, nullptr, 0, M, K, xa, nullptr, 0) ||
      egtr_xs_split_f32(nullptr, dW, K, nullptr, 0, N, K, xw, nullptr, 1)) {
    fprintf(stderr, "split failed\n");
    return 2;
  }
  const void* pa[1] = {xa};
  const void* pw[1] = {xw};
  const float* pb[1] = {db};
  float* pc[1] = {dC};
  void* pcx[1] = {xs_out ? xc : nullptr};
  const int ldc[1] = {N}, nn[1] = {N}, rl[1] = {relu};
  auto run = [&]() { return egtr_gemm_x6_f32(nullptr, 1, pa, pw, pb, pc, ldc, pcx, nn, rl, M, K); };
  int rc = run();
  CK(hipDeviceSynchronize());
  if (rc) {
    fprintf(stderr, "gemm rc %d (%s)\n", rc, egtr_last_hip_error());
    return 2;
  }
  std::vector<float> hC((size_t)M * N);
  CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
  // check: 64 sampled rows incl. the first / last ones, every column, against float64
  double max_err = 0, max_ref = 0;
  long long bad = 0;
  for (int t = 0; t < 64; ++t) {
    const int r = t == 0 ? 0 : t == 1 ? M - 1 : t == 2 ? std::min(M - 1, 127) : t == 3 ? std::min(M - 1, 128) : (int)(rng() % M);
    for (int n = 0; n < N; ++n) {
      double s = hb[n];
      for (int k = 0; k < K; ++k) s += (double)hA[(size_t)r * K + k] * hW[(size_t)n * K + k];
      if (relu && s < 0) s = 0;
      const double e = std::fabs(s - hC[(size_t)r * N + n]);
      if (!(e < 1e-4)) ++bad;
      max_err = std::max(max_err, e);
      max_ref = std::max(max_ref, std::fabs(s));
    }
  }
  printf("M=%d K=%d N=%d relu=%d  max|err| = %.3e (max|ref| %.2f)  bad = %lld\n", M, K, N, relu, max_err, max_ref, bad);
  if (xs_out) {  // XS(C): hi + mid + lo must reproduce C bit for bit
    std::vector<unsigned short> hx(egtr_xs_bytes(M, N) / 2);
    CK(hipMemcpy(hx.data(), xc, hx.size() * 2, hipMemcpyDeviceToHost));
    long long xbad = 0;
    const int KS = N / 16;
    for (int t = 0; t < 64; ++t) {
      const int r = t == 0 ? 0 : t == 1 ? M - 1 : (int)(rng() % M);
      for (int n = 0; n < N; ++n) {
        const size_t f = ((size_t)(r / 32) * KS + n / 16) * 3;
        const size_t in = ((n % 16) / 8) * 256 + (r % 32) * 8 + (n % 8);
        const float s = (bf16_bits_to_float(hx[f * 512 + in]) + bf16_bits_to_float(hx[(f + 1) * 512 + in])) +
                        bf16_bits_to_float(hx[(f + 2) * 512 + in]);
        if (s != hC[(size_t)r * N + n]) ++xbad;
      }
    }
    printf("XS(C) mismatches: %lld\n", xbad);
    bad += xbad;
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 10; ++i) run();
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) run();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / iters, fl = 2.0 * M * K * N;
  printf("%.2f us per launch: %.1f TFLOP/s algorithmic, %.3f of the bf16 dense peak (6x executed)\n", us, fl / us * 1e-6,
         6 * fl / us * 1e-6 / 2500.0);
  return bad ? 1 : 0;
}
