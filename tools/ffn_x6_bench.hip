// Stand-alone development harness of csrc/ffn_x6.hip: y = [LayerNorm(x +] fc2(relu(fc1(x))) [)] against float64 on sampled
// rows, and the launch time.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ffn_x6_bench.hip egtr_amd/csrc/ffn_x6.hip egtr_amd/csrc/xs_split.hip \
//         egtr_amd/csrc/capi.hip -o build/ffn_x6_bench && build/ffn_x6_bench [M F layernorm iters]
#include <hip/hip_runtime.h>

#include <cmath>
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
  const int M = argc > 1 ? atoi(argv[1]) : 12537, F = argc > 2 ? atoi(argv[2]) : 1024;
  const int ln = argc > 3 ? atoi(argv[3]) : 0, iters = argc > 4 ? atoi(argv[4]) : 100;
  const int D = 256;
  std::mt19937 rng(3);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> hx((size_t)M * D), hw1((size_t)F * D), hb1(F), hw2((size_t)D * F), hb2(D), hg(D), hbt(D), hpos((size_t)M * D);
  for (auto& v : hx) v = nd(rng);
  for (auto& v : hw1) v = nd(rng) / 16.f;
  for (auto& v : hb1) v = nd(rng) * 0.3f;
  for (auto& v : hw2) v = nd(rng) / std::sqrt((float)F);
  for (auto& v : hb2) v = nd(rng) * 0.3f;
  for (auto& v : hg) v = 1.f + 0.2f * nd(rng);
  for (auto& v : hbt) v = 0.2f * nd(rng);
  for (auto& v : hpos) v = nd(rng);
  float *dx, *dw1, *db1, *dw2, *db2, *dg, *dbt, *dpos, *dout, *doutp;
  void *xw1, *xw2;
  auto up = [&](float** d, const std::vector<float>& h) {
    CK(hipMalloc(d, h.size() * 4));
    CK(hipMemcpy(*d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  };
  up(&dx, hx); up(&dw1, hw1); up(&db1, hb1); up(&dw2, hw2); up(&db2, hb2); up(&dg, hg); up(&dbt, hbt); up(&dpos, hpos);
  CK(hipMalloc(&dout, (size_t)M * D * 4));
  CK(hipMalloc(&doutp, (size_t)M * D * 4));
  CK(hipMalloc(&xw1, egtr_xs_bytes(F, D)));
  CK(hipMalloc(&xw2, egtr_xs_bytes(D, F)));
  CK(hipMemset(dout, 0xff, (size_t)M * D * 4));
  if (egtr_xs_split_f32(nullptr, dw1, D, nullptr, 0, F, D, xw1, nullptr, 1) ||
      egtr_xs_split_f32(nullptr, dw2, F, nullptr, 0, D, F, xw2, nullptr, 1)) {
    fprintf(stderr, "split failed\n");
    return 2;
  }
  auto run = [&]() {
    return egtr_ffn_x6_f32(nullptr, dx, D, xw1, db1, xw2, db2, ln ? dg : nullptr, ln ? dbt : nullptr, 1e-5f,
                           ln > 1 ? dpos : nullptr, M, dout, ln > 1 ? doutp : nullptr, M, D, F);
  };
  int rc = run();
  CK(hipDeviceSynchronize());
  if (rc) {
    fprintf(stderr, "ffn rc %d (%s)\n", rc, egtr_last_hip_error());
    return 2;
  }
  std::vector<float> ho((size_t)M * D), hop((size_t)M * D);
  CK(hipMemcpy(ho.data(), dout, ho.size() * 4, hipMemcpyDeviceToHost));
  if (ln > 1) CK(hipMemcpy(hop.data(), doutp, hop.size() * 4, hipMemcpyDeviceToHost));
  double max_err = 0, max_ref = 0;
  long long bad = 0;
  std::vector<double> h(F), y(D);
  for (int t = 0; t < 48; ++t) {
    const int r = t == 0 ? 0 : t == 1 ? M - 1 : t == 2 ? std::min(M - 1, 63) : t == 3 ? std::min(M - 1, 64) : (int)(rng() % M);
    for (int f = 0; f < F; ++f) {
      double s = hb1[f];
      for (int k = 0; k < D; ++k) s += (double)hx[(size_t)r * D + k] * hw1[(size_t)f * D + k];
      h[f] = s > 0 ? s : 0;
    }
    for (int n = 0; n < D; ++n) {
      double s = hb2[n];
      for (int f = 0; f < F; ++f) s += h[f] * hw2[(size_t)n * F + f];
      y[n] = s;
    }
    if (ln) {
      double mu = 0, var = 0;
      for (int n = 0; n < D; ++n) { y[n] += hx[(size_t)r * D + n]; mu += y[n]; }
      mu /= D;
      for (int n = 0; n < D; ++n) var += (y[n] - mu) * (y[n] - mu);
      var /= D;
      for (int n = 0; n < D; ++n) y[n] = (y[n] - mu) / std::sqrt(var + 1e-5) * hg[n] + hbt[n];
    }
    for (int n = 0; n < D; ++n) {
      const double e = std::fabs(y[n] - ho[(size_t)r * D + n]);
      if (!(e < 2e-5 * std::max(1.0, std::fabs(y[n])))) ++bad;
      max_err = std::max(max_err, e);
      max_ref = std::max(max_ref, std::fabs(y[n]));
      if (ln > 1) {
        const double e2 = std::fabs(y[n] + hpos[(size_t)r * D + n] - hop[(size_t)r * D + n]);
        if (!(e2 < 2e-5 * std::max(1.0, std::fabs(y[n])))) ++bad;
      }
    }
  }
  printf("M=%d F=%d ln=%d  max|err| = %.3e (max|ref| %.2f)  bad = %lld\n", M, F, ln, max_err, max_ref, bad);
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
  const double us = ms * 1e3 / iters, fl = 4.0 * M * D * F;
  printf("%.2f us per launch: %.1f TFLOP/s algorithmic, %.3f of the bf16 dense peak (6x executed)\n", us, fl / us * 1e-6,
         6 * fl / us * 1e-6 / 2500.0);
  return bad ? 1 : 0;
}
