// Developer harness: times hta_mlp_hmc_sample_f32 on the BASELINE config-4 shape with parts of the
// kernel compiled out (-DHTA_ABL=bits, see mlp_hmc.hip).  Built and run by tools/scratch/mlp_ablate.sh.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "hamiltorch_amd.h"
#if HTA_TIMING || defined(HTA_TIMING_VALU)
extern "C" void hta_dbg_read(unsigned long long* out);
#endif

int main(int argc, char** argv) {
  const int C = argc > 1 ? atoi(argv[1]) : 512, n_in = 8, H = 100, N = 400, M = 4, Nb = 100, L = 10, NT = 20;
  const int D = H * n_in + 2 * H + 1;
  std::vector<float> th((size_t)C * D), X((size_t)N * n_in), Y(N);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) * (1.0f / 16777216.0f)) - 0.5f; };
  for (auto& v : th) v = 0.2f * rnd();
  for (auto& v : X) v = 2.0f * rnd();
  for (int i = 0; i < N; ++i) Y[i] = sinf(X[i * n_in]) + 0.1f * rnd();
  float *dth, *dth0, *dX, *dY; int32_t* rej;
  hipMalloc(&dth, th.size() * 4); hipMalloc(&dth0, th.size() * 4); hipMalloc(&dX, X.size() * 4); hipMalloc(&dY, Y.size() * 4);
  hipMalloc(&rej, C * 4); hipMemset(rej, 0, C * 4);
  hipMemcpy(dth, th.data(), th.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dth0, th.data(), th.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dY, Y.data(), Y.size() * 4, hipMemcpyHostToDevice);
  const float tau[4] = {1, 1, 1, 1};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0, 0);
    int rc = hta_mlp_hmc_sample_f32(dth, dth0, C, n_in, H, 0, HTA_LOSS_REGRESSION, dX, dY, N, M, Nb, tau, 100.0f, 1.0f, 0, nullptr, nullptr, HTA_SPLIT_SYMMETRIC, L, 5e-4f, NT,
                                    rep * NT, 0, 7, 0, nullptr, rej, nullptr, nullptr, nullptr, nullptr);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    if (rc) { printf("error: %s\n", hta_last_error()); return 1; }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  std::vector<int32_t> r(C); hipMemcpy(r.data(), rej, C * 4, hipMemcpyDeviceToHost);
  long tot = 0; for (int v : r) tot += v;
  printf("%s C=%d  %.3f ms per %d trajectories (L=%d, M=%d)  -> %.3f us per gradient  rejected %ld\n", argc > 2 ? argv[2] : "", C, best, NT, L, M,
         best * 1e3 / (NT * (L * 2 * M + 2)), tot);
#if HTA_TIMING || defined(HTA_TIMING_VALU)
  unsigned long long d[16]; hta_dbg_read(d);
  const char* nm[8] = {"rest(axpy,drift,gibbs)", "forward", "barrier 1", "residual stage", "barrier 2", "backward", "tail shuffles", "-"};
  unsigned long long tot2 = 0; for (int k = 0; k < 8; ++k) tot2 += d[k];
  for (int k = 0; k < 8; ++k) printf("   %-22s %10llu ticks  %5.1f %%\n", nm[k], d[k], 100.0 * d[k] / tot2);
#endif
  return 0;
}
