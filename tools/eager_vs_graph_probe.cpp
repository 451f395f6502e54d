// What does it cost the HOST side to run K = 20 dependent tg_step_i8 launches and wait for them (the driver's
// `bench.py --steps 20` sample) -- as a hipGraph replay (what bench.py does) or as K plain launches from a C loop?
// BASELINE config 2 (S=4, 65 536 games).  Wall clock per sample (std::chrono), median of 200 samples.
// Build: hipcc -O3 -std=c++17 tools/eager_vs_graph_probe.cpp -Lmat_mul_amd/lib -ltensorgame -o eager_vs_graph_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/tensor_game.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int64_t B = argc > 1 ? atoll(argv[1]) : 65536;
  const int S = 4, K = argc > 2 ? atoi(argv[2]) : 20;
  int8_t *st, *tok; uint8_t *done, *ovf;
  CK(hipMalloc(&st, B * 64)); CK(hipMalloc(&tok, (size_t)K * B * 12)); CK(hipMalloc(&done, B)); CK(hipMalloc(&ovf, B));
  std::vector<int8_t> hs(B * 64), ht((size_t)K * B * 12);
  for (auto& x : hs) x = (int8_t)((rand() % 5) - 2);
  // action k+1 undoes action k (u negated), so the state stays small
  for (int k = 0; k < K; k += 2)
    for (size_t i = 0; i < (size_t)B * 12; ++i) {
      const int8_t v = (int8_t)((rand() % 10) < 7 ? 1 : (rand() % 2) * 2);
      ht[(size_t)k * B * 12 + i] = v;
      if (k + 1 < K) ht[(size_t)(k + 1) * B * 12 + i] = (i % 12) < 4 ? (int8_t)(2 - v) : v;
    }
  CK(hipMemcpy(st, hs.data(), hs.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(tok, ht.data(), ht.size(), hipMemcpyHostToDevice));
  CK(hipMemset(ovf, 0, B));
  hipStream_t s; CK(hipStreamCreate(&s));
  auto steps = [&]() {
    for (int k = 0; k < K; ++k)
      if (tg_step_i8(st, st, tok + (size_t)k * B * 12, done, ovf, B, S, 64, 1, s) != 0) { fprintf(stderr, "%s\n", tg_last_error()); exit(1); }
  };
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  steps();
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  auto med = [&](auto run) {
    std::vector<double> t;
    for (int i = 0; i < 230; ++i) {
      CK(hipStreamSynchronize(s));
      const auto t0 = std::chrono::steady_clock::now();
      run();
      CK(hipStreamSynchronize(s));
      const auto t1 = std::chrono::steady_clock::now();
      if (i >= 30) t.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
  };
  const double tg = med([&] { CK(hipGraphLaunch(ge, s)); });
  const double te = med([&] { steps(); });
  auto issue = [&]() {  // host time to ENQUEUE the K launches (no wait)
    std::vector<double> t;
    for (int i = 0; i < 100; ++i) {
      CK(hipStreamSynchronize(s));
      const auto t0 = std::chrono::steady_clock::now();
      steps();
      const auto t1 = std::chrono::steady_clock::now();
      t.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
  };
  printf("S=4 B=%lld, K=%d dependent tg_step_i8 launches + stream synchronize, wall clock, median of 200:\n", (long long)B, K);
  printf("  hipGraph replay : %7.1f us per sample  = %.2f us per step  (%.3g steps/s)\n", tg, tg / K, B * K / (tg * 1e-6));
  printf("  K plain launches: %7.1f us per sample  = %.2f us per step  (%.3g steps/s)   [enqueue alone: %.1f us = %.2f us per launch]\n",
         te, te / K, B * K / (te * 1e-6), issue(), issue() / K);
  return 0;
}
