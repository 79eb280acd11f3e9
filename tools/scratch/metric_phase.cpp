// Phase timer of metric_warm_mfma_kernel (developer tool): s_memtime stamps of workgroup 0 at the phase boundaries.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DHTA_TIMING=1 -ffp-contract=on -fno-slp-vectorize -x hip tools/scratch/metric_phase.cpp -x none \
//         $(ls hamiltorch_amd/csrc/build/*.o | grep -v rmhmc_metric_mfma.o) -o tools/scratch/metric_phase.bin
//   (the rest of the library as objects: linked against the shared library the process holds TWO copies of this file's kernels, and the one
//    without the stamps may be the one that is launched)
#include "../../hamiltorch_amd/csrc/rmhmc_metric_mfma.hip"
#include "../../include/hamiltorch_amd.h"
#include <vector>
#include <cmath>
#include <cstdlib>
int main(int argc, char** argv) {
  const int D = argc > 1 ? atoi(argv[1]) : 100, B = argc > 2 ? atoi(argv[2]) : 256, gibbs = argc > 3 ? atoi(argv[3]) : 0;
  const double jitter = argc > 4 ? atof(argv[4]) : 1e-3;
  hta::g_metric_second = argc > 5 ? atoi(argv[5]) : 1;      // 0: the three-product second pass
  std::vector<double> Q(D * D);
  srand(1);
  for (auto& v : Q) v = rand() / (double)RAND_MAX - 0.5;
  for (int k = 0; k < D; ++k) {                       // Gram-Schmidt on the columns
    for (int j = 0; j < k; ++j) { double d = 0; for (int i = 0; i < D; ++i) d += Q[i * D + k] * Q[i * D + j]; for (int i = 0; i < D; ++i) Q[i * D + k] -= d * Q[i * D + j]; }
    double n = 0; for (int i = 0; i < D; ++i) n += Q[i * D + k] * Q[i * D + k]; n = sqrt(n); for (int i = 0; i < D; ++i) Q[i * D + k] /= n;
  }
  std::vector<float> V0(D * D), lam0(D), P(D * D), X(B * D), m(B * D);
  for (int k = 0; k < D; ++k) lam0[k] = 0.5f + 1.5f * k / (D > 1 ? D - 1 : 1);
  for (int i = 0; i < D; ++i) for (int j = 0; j < D; ++j) { double a = 0; for (int k = 0; k < D; ++k) a += Q[i * D + k] * lam0[k] * Q[j * D + k]; P[i * D + j] = (float)a; V0[i * D + j] = (float)Q[i * D + j]; }
  for (auto& v : X) v = 0.1f * (rand() / (float)RAND_MAX - 0.5f);
  for (auto& v : m) v = rand() / (float)RAND_MAX - 0.5f;
  float *dV0, *dl, *dP, *dX, *dm, *dx, *dg, *dp, *dH, *dmu;
  hipMalloc(&dV0, D * D * 4); hipMalloc(&dl, D * 4); hipMalloc(&dP, D * D * 4); hipMalloc(&dX, B * D * 4); hipMalloc(&dm, B * D * 4);
  hipMalloc(&dx, B * D * 4); hipMalloc(&dg, B * D * 4); hipMalloc(&dp, B * D * 4); hipMalloc(&dH, B * 4); hipMalloc(&dmu, D * 4);
  hipMemcpy(dV0, V0.data(), D * D * 4, hipMemcpyHostToDevice); hipMemcpy(dl, lam0.data(), D * 4, hipMemcpyHostToDevice);
  hipMemcpy(dP, P.data(), D * D * 4, hipMemcpyHostToDevice); hipMemcpy(dX, X.data(), B * D * 4, hipMemcpyHostToDevice);
  hipMemcpy(dm, m.data(), B * D * 4, hipMemcpyHostToDevice); hipMemset(dx, 0, B * D * 4); hipMemset(dg, 0, B * D * 4); hipMemset(dmu, 0, D * 4);
  hta::MetricArgsT<float> a; memset(&a, 0, sizeof(a));
  a.B = B; a.D = D; a.metric = 1; a.Hs = dP; a.hs_stride = 0; a.alpha = 1e6; a.has_jitter = jitter > 0; a.jitter = jitter; a.seed = 5; a.draw = 1; a.sub = 2;
  a.Pm = dP; a.mu = dmu; a.V0 = dV0; a.lam0 = dl;
  if (gibbs) a.p_out = dp; else { a.X = dX; a.m = dm; a.upd_x = dx; a.cx = 0.05; a.upd_g = dg; a.cg = -0.05; a.H_out = dH; }
  // argv[6] = L > 0: the TRAJECTORY kernel instead (round 6: momentum draw, H_old, 4 L half steps, H_new in one launch; the stamps that remain
  // are the last evaluation's, slot 31 = the gap between the previous evaluation's last stamp and this one's first)
  const int trajL = argc > 6 ? atoi(argv[6]) : 0;
  hta::g_metric_resident = argc > 7 ? atoi(argv[7]) : 1;
  float *dth, *dthc, *dpmc, *dH1, *dlp;
  hipMalloc(&dth, B * D * 4); hipMalloc(&dthc, B * D * 4); hipMalloc(&dpmc, B * D * 4); hipMalloc(&dH1, B * 4); hipMalloc(&dlp, B * 4);
  hta::MetricTrajArgs ta{dX, dth, dp, dthc, dpmc, dH, dH1, dlp, trajL, 0.05, cosf(2.f), sinf(2.f)};
  if (trajL > 0) { a.X = nullptr; a.m = nullptr; a.upd_x = nullptr; a.upd_g = nullptr; a.H_out = nullptr; a.p_out = nullptr; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, 0);
    for (int k = 0; k < 10; ++k) { int rc = trajL > 0 ? hta::metric_traj_mfma(a, ta, 0) : hta::metric_warm_mfma(a, 0); if (rc) { printf("error %s\n", hta_last_error()); return 1; } }
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("D=%d B=%d gibbs=%d jitter=%g second=%d: %.1f us per launch\n", D, B, gibbs, jitter, hta::g_metric_second, ms * 100);
  }
  long long t[32] = {0}; (void)dx; { hipError_t e = hipMemcpyFromSymbol(t, HIP_SYMBOL(hta::hta_metric_dbg), sizeof(t)); if (e != hipSuccess) printf("hipMemcpyFromSymbol: %s\n", hipGetErrorString(e)); }
  const char* names[32] = {"start", "operands+V0 stage", "logp + V0^T m", "formation", "it0 begin", "it0 gemms", "it0 E", "it0 X", "it1 begin", "it1 products (F E1 | T,S,Gm)", "it1 E", "it1 X update",
                           "it2 begin", "it2 gemms", "it2 E", "it2 X", "it3 begin", "it3 gemms", "it3 E", "it3 X", "refine end", "softabs+solve", "G assembly", "cholesky", "end"};
  long long prev = t[0];
  for (int k = 1; k <= 24; ++k) {
    if (k >= 12 && k <= 19 && t[k] > t[20]) continue;      // the slots of the passes it = 2, 3 hold the solve's sub-phase stamps when those passes do not run
    if (t[k] > prev) { printf("  %-22s %8lld cycles\n", names[k], t[k] - prev); prev = t[k]; }
  }
  printf("  total %lld cycles\n", prev - t[0]);
  if (trajL > 0) printf("  trajectory kernel, L = %d, resident = %d: gap before this evaluation %lld cycles; per evaluation (launch time / (4 L + 3)) see above\n", trajL, hta::g_metric_resident, t[31]);
  printf("  formation product, wave 0 (cycles since the phase began): entry %lld | k loop begins %lld | k loop ends %lld | returned %lld | after the barrier %lld | diagonal added %lld | phase ends %lld\n",
         t[25] - t[2], t[26] - t[2], t[27] - t[2], t[28] - t[2], t[29] - t[2], t[30] - t[2], t[3] - t[2]);
  if (!gibbs)
    printf("  solve sub-phases: soft-abs map + log-det sum %lld | X^T m' %lld | + E^T y %lld | w, quad sum %lld | + E w %lld | X w %lld | V0 stage %lld | V0 x', store %lld\n",
           t[12] - t[20], t[13] - t[12], t[14] - t[13], t[15] - t[14], t[16] - t[15], t[17] - t[16], t[18] - t[17], t[21] - t[18]);
  long long w[16][16]; { hipError_t e = hipMemcpyFromSymbol(w, HIP_SYMBOL(hta::hta_metric_wdbg), sizeof(w)); (void)e; }
  if (w[0][4]) {       // the fast solve ran (round 6): per wave, cycles since the formation phase / the second product began
    for (int k = 0; k < 16; ++k) printf("    wave %2d form: entry %6lld  k loop %6lld .. %6lld  at barrier %6lld  past %6lld  epilogue done %6lld  returned %6lld | second: entry %6lld  k loop %6lld .. %6lld  at barrier %6lld  past %6lld  epilogue done %6lld  returned %6lld\n", k,
        w[k][4] - t[2], w[k][4] - t[2], w[k][5] - t[2], w[k][6] - t[2], w[k][7] - t[2], w[k][8] - t[2], w[k][9] - t[2],
        w[k][10] - t[3], w[k][10] - t[3], w[k][11] - t[3], w[k][12] - t[3], w[k][13] - t[3], w[k][14] - t[3], w[k][15] - t[3]);
    printf("  chain, wave 0 (cycles since it began):");
    for (int k = 12; k <= 18; ++k) printf(" %lld", t[k] - t[9]);
    printf("\n");
  } else
    for (int k = 0; k < 16; ++k) printf("    wave %2d: entry %6lld  k loop %6lld .. %6lld  returned %6lld\n", k, w[k][0] - t[2], w[k][1] - t[2], w[k][2] - t[2], w[k][3] - t[2]);
  return 0;
}
