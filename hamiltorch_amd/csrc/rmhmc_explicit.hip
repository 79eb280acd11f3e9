// Explicit (Tao 2016 / Cobb et al. 2019) RMHMC integrator and sampler for constant-curvature
// (Gaussian) targets: hamiltorch/samplers.py:389-462 (leapfrog), :969-1026 (sample, RMHMC branch).
//
// Host-side driver: one C call ENQUEUES the whole sequence of launches of a run (no
// synchronisation, accept/reject stays on the device).  Per step (S:427-461) with the duplicate
// gradient calls of each half step evaluated once:
//   phi_A/2 : p~.. one metric_eval launch: thc += eh G(th)^-1 pmc ;  pm  -= eh P (th  - mu)
//   phi_B/2 : one launch:                 th  += eh G(thc)^-1 pm ;  pmc -= eh P (thc - mu)
//   phi_C   : element-wise rotation with the reference's SEQUENTIAL update order (S:447-450, Q1)
//   phi_B/2, phi_A/2 again.
// dH/dtheta = -grad log p exactly because dG/dtheta == 0 for this family.
// Jitter sub-streams follow the reference's call order: 0 gibbs, 1 initial H,
// 2 + 8 l + {1, 2, 4, 7} the metric evaluations of step l, 2 + 8 L the final H.
#include <math.h>
#include <map>
#include <mutex>
#include <utility>
#include <vector>
#include "common.hpp"
#include "rmhmc.hpp"

namespace hta {

template <typename T>
__global__ void phi_c_kernel(T* __restrict__ th, T* __restrict__ pm, T* __restrict__ thc, T* __restrict__ pmc, T c,
                             T s, int64_t total) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    T a = th[t], b = pm[t], ac = thc[t], bc = pmc[t];
    phi_c_elem<T>(a, b, ac, bc, c, s);                      // S:447-450
    th[t] = a; pm[t] = b; thc[t] = ac; pmc[t] = bc;
  }
}

// chol(P) of ONE matrix in double precision: a single workgroup, the symmetric part of P in LDS (row stride D + 1), right-looking
// factorisation (column scale + trailing update per step), the lower factor rounded to T and written row-major with a zero upper part.
template <typename T>
__global__ void __launch_bounds__(256) chol_p_kernel(const T* __restrict__ P, T* __restrict__ L, int D) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* A = reinterpret_cast<double*>(smem_raw);
  const int lda = D + 1, tid = threadIdx.x;
  for (int e = tid; e < D * D; e += 256) {
    const int i = e / D, j = e - i * D;
    A[i * lda + j] = 0.5 * ((double)P[(int64_t)i * D + j] + (double)P[(int64_t)j * D + i]);      // the symmetric part, as the kernels read P
  }
  for (int j = 0; j < D; ++j) {
    __syncthreads();
    const double djj = sqrt(A[j * lda + j]);
    const double inv = 1.0 / djj;
    __syncthreads();
    for (int i = j + tid; i < D; i += 256) A[i * lda + j] = (i == j) ? djj : A[i * lda + j] * inv;
    __syncthreads();
    const int r = D - j - 1;
    for (int e = tid; e < r * r; e += 256) {
      const int ii = e / r, kk = e - ii * r;
      if (kk <= ii) {
        const int i = j + 1 + ii, k = j + 1 + kk;
        A[i * lda + k] -= A[i * lda + j] * A[k * lda + j];
      }
    }
  }
  __syncthreads();
  for (int e = tid; e < D * D; e += 256) {
    const int i = e / D, j = e - i * D;
    L[e] = (j <= i) ? (T)A[i * lda + j] : (T)0;
  }
}

template <typename T> struct RmModel {
  const T* P; const T* mu; double log_norm; int metric; double alpha; int has_jitter; double jitter;
  uint64_t seed; uint64_t chain_offset; int64_t C; int D;
  const T* V0; const T* lam0;     // eigenbasis of the jitter-free P (warm start), or NULL
  void* mws; int64_t mws_bytes;   // the metric evaluations' slab area inside the caller's workspace (sizes beyond one CU's LDS), or NULL
};

template <typename T> static MetricArgsT<T> base_args(const RmModel<T>& m, uint32_t draw, uint32_t sub) {
  MetricArgsT<T> a;
  memset(&a, 0, sizeof(a));
  a.B = m.C; a.D = m.D; a.metric = m.metric; a.Hs = m.P; a.hs_stride = 0; a.alpha = m.alpha;
  a.has_jitter = m.has_jitter; a.jitter = m.jitter; a.seed = m.seed; a.chain_offset = m.chain_offset;
  a.draw = draw; a.sub = sub; a.Pm = m.P; a.mu = m.mu; a.log_norm = m.log_norm;
  a.V0 = m.V0; a.lam0 = m.lam0;
  a.workspace = m.mws; a.workspace_bytes = m.mws_bytes;
  return a;
}

// one half step: upd_x += eh G(X)^-1 mvec ;  upd_g -= eh P (X - mu)
template <typename T>
static int half_step(const RmModel<T>& m, uint32_t draw, uint32_t sub, const T* X, const T* mvec, T* upd_x, T* upd_g,
                     double eh, hipStream_t s) {
  MetricArgsT<T> a = base_args(m, draw, sub);
  a.X = X; a.m = mvec; a.upd_x = upd_x; a.cx = eh; a.upd_g = upd_g; a.cg = -eh;
  return metric_eval<T>(a, s);
}

template <typename T>
static int explicit_steps(const RmModel<T>& m, uint32_t draw, T* th, T* pm, T* thc, T* pmc, int steps, double eps,
                          double omega, T* path_theta, T* path_p, hipStream_t s) {
  const int64_t total = m.C * m.D;
  const double eh = 0.5 * eps;
  const float ang = (float)(2.0 * omega * eps);        // S:435-436: float32 cos / sin whatever the state dtype
  const T c = (T)cosf(ang), sn = (T)sinf(ang);
  int grid = (int)((total + 255) / 256); if (grid > 2048) grid = 2048;
  for (int l = 0; l < steps; ++l) {
    const uint32_t k0 = 2u + 8u * (uint32_t)l;
    int rc;
    if ((rc = half_step<T>(m, draw, k0 + 1, th, pmc, thc, pm, eh, s))) return rc;     // phi_A/2  S:429-430
    if ((rc = half_step<T>(m, draw, k0 + 2, thc, pm, th, pmc, eh, s))) return rc;     // phi_B/2  S:432-433
    phi_c_kernel<T><<<grid, 256, 0, s>>>(th, pm, thc, pmc, c, sn, total);             // phi_C    S:447-450
    if ((rc = half_step<T>(m, draw, k0 + 4, thc, pm, th, pmc, eh, s))) return rc;     // phi_B/2  S:454-455
    if ((rc = half_step<T>(m, draw, k0 + 7, th, pmc, thc, pm, eh, s))) return rc;     // phi_A/2  S:457-458
    if (path_theta) (void)hipMemcpyAsync(path_theta + (int64_t)l * total, th, total * sizeof(T), hipMemcpyDeviceToDevice, s);
    if (path_p) (void)hipMemcpyAsync(path_p + (int64_t)l * total, pm, total * sizeof(T), hipMemcpyDeviceToDevice, s);
  }
  HTA_CHECK_LAUNCH("hta_rmhmc_gaussian_leapfrog");
  return HTA_OK;
}

template <typename T>
int rmhmc_leapfrog(T* th, T* pm, T* thc, T* pmc, const T* P, const T* mu, int metric, double alpha, int has_jitter,
                   double jitter, uint64_t seed, uint64_t chain_offset, uint32_t draw, int64_t C, int D, int steps,
                   double eps, double omega, T* path_theta, T* path_p, void* workspace, int64_t workspace_bytes, hipStream_t s) {
  HTA_REQUIRE(th && pm && thc && pmc && P && mu && C > 0 && D > 0 && steps >= 0, "hta_rmhmc_gaussian_leapfrog: bad arguments");
  RmModel<T> m{P, mu, 0.0, metric, alpha, has_jitter, jitter, seed, chain_offset, C, D, nullptr, nullptr, workspace, workspace_bytes};
  return explicit_steps<T>(m, draw, th, pm, thc, pmc, steps, eps, omega, path_theta, path_p, s);
}

// ---- the per-target setup of hta_rmhmc_gaussian_sample and its cache -------------------------------------------------------
struct RmPrepared {
  const void* P; int D, metric, has_jitter, elem; double alpha, jitter;      // what it was prepared for
  int64_t C;                                                                  // ... and for how many chains: the prepared block sits behind 4 C D + 3 C elements
  int K, series; double logdetP;                                              // the fused route's plan (K < 0: not eligible)
  int split;                                                                  // chol(P) is in the workspace (the split momentum draw)
  int fused_keys;                                                             // g_rmhmc_fused at preparation time
};
static std::mutex g_prep_mu;
static std::map<std::pair<int, const void*>, RmPrepared> g_prepared;         // (device, workspace) -> plan
static int current_device() { int d = 0; (void)hipGetDevice(&d); return d; }
static bool prepared_lookup(const void* ws, const void* P, int64_t C, int D, int metric, double alpha, int has_jitter, double jitter,
                            size_t elem, RmPrepared& out) {
  std::lock_guard<std::mutex> lock(g_prep_mu);
  auto it = g_prepared.find({current_device(), ws});
  if (it == g_prepared.end()) return false;
  const RmPrepared& p = it->second;
  if (p.P != P || p.C != C || p.D != D || p.metric != metric || p.alpha != alpha || p.has_jitter != has_jitter || p.jitter != jitter ||
      p.elem != (int)elem || p.fused_keys != g_rmhmc_fused)
    return false;
  out = p;
  return true;
}

// the base layout: augmented state 4 C D | H_old, H_new, log p: 3 C | V0, S, chol(P): 3 D^2 | lam0: D - rounded up to 16 bytes so
// that the metric slabs behind it (sizes beyond one CU's LDS: metric_eval_workspace_bytes) and the pre-drawn momenta start aligned
static int64_t rm_base_bytes(int64_t C, int D, int elem) {
  return ((4 * C * D + 3 * C + 3 * (int64_t)D * D + D) * (int64_t)elem + 15) & ~(int64_t)15;
}

template <typename T>
static int rmhmc_setup(RmModel<T>& m, T* V0, T* lam0, T* Sinv, T* LP, bool want_plan, RmPrepared& pr, hipStream_t s) {
  const char* who = "hta_rmhmc_gaussian_sample";
  const int D = m.D;
  pr = RmPrepared{m.P, D, m.metric, m.has_jitter, (int)sizeof(T), m.alpha, m.jitter, m.C, -1, 0, 0.0, 0, g_rmhmc_fused};
  if (m.metric == HTA_METRIC_SOFTABS || g_rmhmc_fused) {
    // the target's curvature is one matrix for all chains and all evaluation points: diagonalise it once
    MetricArgsT<T> a0 = base_args(m, 0, 0);
    a0.metric = HTA_METRIC_SOFTABS; a0.B = 1; a0.has_jitter = 0; a0.V_out = V0; a0.lamraw_out = lam0;
    int rc0 = metric_eval<T>(a0, s);
    if (rc0) return rc0;
  }
  if (g_rmhmc_fused && want_plan && D <= 1024) {
    T lam_host[1024];
    if (hipMemcpyAsync(lam_host, lam0, D * sizeof(T), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) {
      set_error("%s: reading the spectrum back failed: %s", who, hipGetErrorString(hipGetLastError()));
      return HTA_ERR_LAUNCH;
    }
    pr.K = fused_plan<T>(lam_host, D, m.metric, m.alpha, m.has_jitter, m.jitter, &pr.logdetP, &pr.series);
    if (pr.K >= 0) {
      int rc = inverse_from_eigen<T>(V0, lam0, Sinv, D, s);
      if (rc) return rc;
      if (m.has_jitter && D <= 128) {
        // chol(P) for the split momentum draw (rmhmc_fused.hip: rmhmc_momentum_split_kernel): one 100 x 100 factorisation per TARGET, in
        // double, on the DEVICE (round 6: chol_p_kernel - a single workgroup, the matrix in LDS; rounds 2-5 read P back and factorised
        // on the host: two more synchronisations).  P is positive definite iff its smallest eigenvalue is: the spectrum is on the host
        // already (the plan above), so whether the split draw applies needs no second read-back.
        double lmin = (double)lam_host[0], lmax = fabs((double)lam_host[0]);
        for (int i = 1; i < D; ++i) { lmin = fmin(lmin, (double)lam_host[i]); lmax = fmax(lmax, fabs((double)lam_host[i])); }
        if (lmin > 1e-6 * lmax) {
          const size_t lds = (size_t)D * (D + 1) * sizeof(double);
          static DevOnce chol_attr;
          if (!chol_attr) {
            (void)hipFuncSetAttribute((const void*)chol_p_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            chol_attr = true;
          }
          chol_p_kernel<T><<<1, 256, lds, s>>>(m.P, LP, D);
          HTA_CHECK_LAUNCH(who);
          pr.split = 1;
        }
      }
    }
  }
  return HTA_OK;
}

// hta_rmhmc_gaussian_prepare: run the setup into `workspace` and remember its plan, keyed by (device, workspace)
template <typename T>
int rmhmc_prepare(const T* P, const T* mu, int metric, double alpha, int has_jitter, double jitter, int64_t C, int D,
                  void* workspace, int64_t workspace_bytes, hipStream_t s) {
  const char* who = "hta_rmhmc_gaussian_prepare";
  HTA_REQUIRE(P && mu && C > 0 && D > 0, "%s: bad arguments", who);
  const int64_t total = C * D;
  const int64_t base = rm_base_bytes(C, D, (int)sizeof(T)), mslab = metric_eval_workspace_bytes(C, D, (int)sizeof(T));
  const int64_t need = base + mslab;
  HTA_REQUIRE(workspace && workspace_bytes >= need, "%s: workspace of %lld bytes required", who, (long long)need);
  T* V0 = (T*)workspace + 4 * total + 3 * C; T* lam0 = V0 + (int64_t)D * D; T* Sinv = lam0 + D; T* LP = Sinv + (int64_t)D * D;
  RmModel<T> m{P, mu, 0.0, metric, alpha, has_jitter, jitter, 0, 0, C, D, nullptr, nullptr, mslab ? (char*)workspace + base : nullptr, mslab};
  RmPrepared pr;
  const int rc = rmhmc_setup<T>(m, V0, lam0, Sinv, LP, true, pr, s);
  std::lock_guard<std::mutex> lock(g_prep_mu);
  if (rc) { g_prepared.erase({current_device(), workspace}); return rc; }
  g_prepared[{current_device(), workspace}] = pr;
  return HTA_OK;
}

template <typename T>
int rmhmc_sample(T* cur, const T* theta_init, const T* P, const T* mu, double log_norm, int metric, double alpha,
                 int has_jitter, double jitter, int64_t C, int D, int L, double eps, double omega, int n_traj,
                 int traj_offset, int burn, uint64_t seed, uint64_t chain_offset, T* samples, int32_t* reject_count,
                 T* H_old_out, T* H_new_out, uint8_t* accept_out, void* workspace, int64_t workspace_bytes,
                 hipStream_t s) {
  const char* who = "hta_rmhmc_gaussian_sample";
  HTA_REQUIRE(cur && theta_init && P && mu && reject_count && C > 0 && D > 0 && L >= 0 && n_traj >= 0, "%s: bad arguments", who);
  const int64_t total = C * D;
  const int64_t base = rm_base_bytes(C, D, (int)sizeof(T)), mslab = metric_eval_workspace_bytes(C, D, (int)sizeof(T));
  const int64_t need = base + mslab;
  HTA_REQUIRE(workspace && workspace_bytes >= need, "%s: workspace of %lld bytes required", who, (long long)need);
  T* th = (T*)workspace; T* pm = th + total; T* thc = pm + total; T* pmc = thc + total;
  T* H0 = pmc + total; T* H1 = H0 + C; T* lp1 = H1 + C;
  T* V0 = lp1 + C; T* lam0 = V0 + (int64_t)D * D; T* Sinv = lam0 + D; T* LP = Sinv + (int64_t)D * D;
  RmModel<T> m{P, mu, log_norm, metric, alpha, has_jitter, jitter, seed, chain_offset, C, D, nullptr, nullptr,
               mslab ? (char*)workspace + base : nullptr, mslab};
  // Once per TARGET: the eigenbasis of the curvature matrix (every evaluation only adds its own jitter to the diagonal and
  // starts from that basis), the host-side plan of the fused route (needs the spectrum on the host: one D-element copy and a
  // synchronise) and the shared inverse S.  A caller that keeps sampling one target in several calls prepares its workspace
  // once (hta_rmhmc_gaussian_prepare): the cold Jacobi + inverse are 1.2 ms, 14 % of a 100-trajectory call at 1024 chains.
  RmPrepared pr;
  if (!prepared_lookup(workspace, P, C, D, metric, alpha, has_jitter, jitter, sizeof(T), pr)) {
    const int rcp = rmhmc_setup<T>(m, V0, lam0, Sinv, LP, n_traj > 0, pr, s);
    if (rcp) return rcp;
  }
  if (metric == HTA_METRIC_SOFTABS) { m.V0 = V0; m.lam0 = lam0; }
  if (g_rmhmc_fused && n_traj > 0 && D <= 1024 && pr.K >= 0) {
    // the soft-abs map is the identity on this spectrum (or the metric is the Hessian itself): the whole run is one launch
    // sequence of rmhmc_fused.hip.  Room for pre-drawn momenta: whatever the caller's workspace holds beyond the base
    // layout, else the (unused on this path) augmented-state area at its head: 4 trajectories per pass
    T* p_ws = (T*)((char*)workspace + need);             // behind the base layout and the metric slabs
    int64_t p_elems = workspace_bytes / (int64_t)sizeof(T) - (p_ws - (T*)workspace);
    if (p_elems < total) { p_ws = th; p_elems = 4 * total; }
    return rmhmc_fused_sample<T>(cur, theta_init, P, Sinv, mu, log_norm, pr.logdetP, has_jitter, jitter, pr.K, pr.series, C, D, L, eps,
                                 omega, n_traj, traj_offset, burn, seed, chain_offset, samples, reject_count, H_old_out,
                                 H_new_out, accept_out, p_ws, p_elems, pr.split ? LP : nullptr, s);
  }
  bool traj_kernel = false;
  if constexpr (sizeof(T) == 4) {
    const MetricArgsT<T> probe = base_args(m, 0, 0);
    traj_kernel = metric_traj_mfma_eligible(reinterpret_cast<const MetricArgsT<float>&>(probe));
  }
  for (int t = 0; t < n_traj; ++t) {
    const int n = traj_offset + t;
    int rc;
    bool selected = false;
    if constexpr (sizeof(T) == 4) {
      if (traj_kernel) {       // one launch: draw, H_old, L steps, H_new (rmhmc_metric_mfma.hip: metric_traj_mfma_kernel)
        const MetricArgsT<T> a = base_args(m, (uint32_t)n, 0);
        const float ang = (float)(2.0 * omega * eps);        // S:435-436
        float* rowf = (samples && n > burn) ? (float*)samples + (int64_t)(n - burn) * total : nullptr;
        const MetricTrajArgs ta{(float*)cur, (float*)th, (float*)pm, (float*)thc, (float*)pmc, (float*)H0, (float*)H1, (float*)lp1, L, 0.5 * eps,
                                cosf(ang), sinf(ang), (const float*)theta_init, rowf, reject_count, accept_out ? accept_out + (int64_t)t * C : nullptr,
                                n, burn, g_metric_select ? 1 : 0, 0};
        if ((rc = metric_traj_mfma(reinterpret_cast<const MetricArgsT<float>&>(a), ta, s))) return rc;
        selected = g_metric_select != 0;                   // (the kernel ended with the chain's Metropolis selection)
      }
    }
    if (!traj_kernel) {
    {   // gibbs: p ~ N(0, G(theta))  (S:183-184)
      MetricArgsT<T> a = base_args(m, (uint32_t)n, 0);
      a.p_out = pm;
      if ((rc = metric_eval<T>(a, s))) return rc;
    }
    {   // H_old = rm_hamiltonian(theta, p)  (S:971 -> S:822, halved at S:977)
      MetricArgsT<T> a = base_args(m, (uint32_t)n, 1);
      a.X = cur; a.m = pm; a.H_out = H0;
      if ((rc = metric_eval<T>(a, s))) return rc;
    }
    (void)hipMemcpyAsync(th, cur, total * sizeof(T), hipMemcpyDeviceToDevice, s);     // S:425-426
    (void)hipMemcpyAsync(thc, cur, total * sizeof(T), hipMemcpyDeviceToDevice, s);
    (void)hipMemcpyAsync(pmc, pm, total * sizeof(T), hipMemcpyDeviceToDevice, s);
    if ((rc = explicit_steps<T>(m, (uint32_t)n, th, pm, thc, pmc, L, eps, omega, nullptr, nullptr, s))) return rc;
    {   // H_new on the un-augmented pair (S:989, Q4)
      MetricArgsT<T> a = base_args(m, (uint32_t)n, 2u + 8u * (uint32_t)L);
      a.X = th; a.m = pm; a.H_out = H1; a.logp_out = lp1;
      if ((rc = metric_eval<T>(a, s))) return rc;
    }
    }
    T* row = (samples && n > burn) ? samples + (int64_t)(n - burn) * total : nullptr;
    if (!selected && (rc = mh_select<T>(cur, th, theta_init, H0, H1, lp1, row, reject_count,
                                        accept_out ? accept_out + (int64_t)t * C : nullptr, C, D, n, burn, seed, chain_offset, s)))
      return rc;
    if (H_old_out) (void)hipMemcpyAsync(H_old_out + (int64_t)t * C, H0, C * sizeof(T), hipMemcpyDeviceToDevice, s);
    if (H_new_out) (void)hipMemcpyAsync(H_new_out + (int64_t)t * C, H1, C * sizeof(T), hipMemcpyDeviceToDevice, s);
  }
  HTA_CHECK_LAUNCH(who);
  return HTA_OK;
}

}  // namespace hta

extern "C" {
int64_t hta_rmhmc_workspace_bytes(int64_t C, int D, int elem_size) {
  return hta::rm_base_bytes(C, D, elem_size) + hta::metric_eval_workspace_bytes(C, D, elem_size);
}

#define HTA_DEFINE_ROT(SUF, T)                                                                                  \
  int hta_rmhmc_binding_rotation_##SUF(T* theta, T* p, T* theta_copy, T* p_copy, int64_t total, double eps,      \
                                       double omega, void* stream) {                                             \
    if (!theta || !p || !theta_copy || !p_copy || total <= 0) {                                                  \
      hta::set_error("hta_rmhmc_binding_rotation: NULL pointer / empty state");                                  \
      return HTA_ERR_INVALID;                                                                                    \
    }                                                                                                            \
    const float ang = (float)(2.0 * omega * eps);                                                                \
    int grid = (int)((total + 255) / 256); if (grid > 2048) grid = 2048;                                         \
    hta::phi_c_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>(theta, p, theta_copy, p_copy, (T)cosf(ang),      \
                                                                (T)sinf(ang), total);                            \
    HTA_CHECK_LAUNCH("hta_rmhmc_binding_rotation");                                                              \
    return HTA_OK;                                                                                               \
  }
HTA_DEFINE_ROT(f32, float)
HTA_DEFINE_ROT(f64, double)

#define HTA_DEFINE_RM(SUF, T)                                                                                   \
  int hta_rmhmc_gaussian_leapfrog_##SUF(T* theta, T* p, T* theta_copy, T* p_copy, const T* P, const T* mu,       \
                                        int metric, double alpha, int has_jitter, double jitter, uint64_t seed,  \
                                        uint64_t chain_offset, uint32_t draw, int64_t C, int D, int steps,       \
                                        double eps, double omega, T* path_theta, T* path_p, void* workspace,     \
                                        int64_t workspace_bytes, void* stream) {                                 \
    return hta::rmhmc_leapfrog<T>(theta, p, theta_copy, p_copy, P, mu, metric, alpha, has_jitter, jitter, seed,  \
                                  chain_offset, draw, C, D, steps, eps, omega, path_theta, path_p, workspace,    \
                                  workspace_bytes, (hipStream_t)stream);                                         \
  }                                                                                                              \
  int hta_rmhmc_gaussian_sample_##SUF(T* theta, const T* theta_init, const T* P, const T* mu, double log_norm,   \
                                      int metric, double alpha, int has_jitter, double jitter, int64_t C, int D, \
                                      int L, double eps, double omega, int n_traj, int traj_offset, int burn,    \
                                      uint64_t seed, uint64_t chain_offset, T* samples, int32_t* reject_count,   \
                                      T* H_old, T* H_new, uint8_t* accept, void* workspace,                      \
                                      int64_t workspace_bytes, void* stream) {                                   \
    return hta::rmhmc_sample<T>(theta, theta_init, P, mu, log_norm, metric, alpha, has_jitter, jitter, C, D, L,   \
                                eps, omega, n_traj, traj_offset, burn, seed, chain_offset, samples, reject_count, \
                                H_old, H_new, accept, workspace, workspace_bytes, (hipStream_t)stream);           \
  }
HTA_DEFINE_RM(f32, float)
HTA_DEFINE_RM(f64, double)

#define HTA_DEFINE_PREP(SUF, T)                                                                                 \
  int hta_rmhmc_gaussian_prepare_##SUF(const T* P, const T* mu, int metric, double alpha, int has_jitter,        \
                                       double jitter, int64_t C, int D, void* workspace, int64_t workspace_bytes, \
                                       void* stream) {                                                           \
    return hta::rmhmc_prepare<T>(P, mu, metric, alpha, has_jitter, jitter, C, D, workspace, workspace_bytes,      \
                                 (hipStream_t)stream);                                                           \
  }
HTA_DEFINE_PREP(f32, float)
HTA_DEFINE_PREP(f64, double)

int hta_rmhmc_gaussian_forget(void* workspace) {
  std::lock_guard<std::mutex> lock(hta::g_prep_mu);
  hta::g_prepared.erase({hta::current_device(), workspace});
  return HTA_OK;
}
}
