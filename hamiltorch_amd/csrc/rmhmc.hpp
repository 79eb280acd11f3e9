// Shared declarations of the RMHMC sources (rmhmc_metric.hip, rmhmc_explicit.hip, hmc_pieces.hip).
#pragma once
#include "common.hpp"

namespace hta {

template <typename T> struct MetricArgsT {   // typed view of HtaMetricArgs (include/hamiltorch_amd.h)
  int64_t B; int32_t D; int32_t metric;
  const T* Hs; int64_t hs_stride;
  double alpha;
  int32_t has_jitter; int32_t max_sweeps; double jitter; uint64_t seed; uint64_t chain_offset; uint32_t draw; uint32_t sub;
  const T* X; const T* Pm; const T* mu; double log_norm;
  const T* m;
  T* p_out;
  T* x_out; T* G_out; T* lam_out; T* V_out; T* L_out; T* logdet_out; T* quad_out; T* H_out; T* logp_out;
  T* upd_x; double cx; T* upd_g; double cg;
  const T* V0; const T* lam0; T* lamraw_out;
  T* dmetric_out;
  int64_t v0_stride;           // elements between consecutive systems' V0 (0: one basis shared by all systems)
  void* workspace; int64_t workspace_bytes;   // ABI 10: caller-owned scratch for the sizes whose matrices exceed one CU's LDS (hta_metric_eval_workspace_bytes)
};
static_assert(sizeof(MetricArgsT<float>) == sizeof(HtaMetricArgs), "HtaMetricArgs layout drifted from MetricArgsT");

template <typename T> int metric_eval(const MetricArgsT<T>& a, hipStream_t s);
int64_t metric_eval_workspace_bytes(int64_t B, int D, int elem_size);       // 0 while both matrices of a system fit the LDS of a CU
extern int g_metric_mfma;                                   // tuning key "metric_mfma" (default 1)
extern int g_metric_general;                                // tuning key "metric_general" (default 1): per-system bases on the matrix cores
bool metric_warm_mfma_eligible(const MetricArgsT<float>& a);
int metric_warm_mfma(const MetricArgsT<float>& a, hipStream_t s);     // rmhmc_metric_mfma.hip

// the binding rotation phi_C of one element with the reference's SEQUENTIAL update order (S:447-450)
template <typename T> __device__ __forceinline__ void phi_c_elem(T& a, T& b, T& ac, T& bc, T c, T s) {
  const T h = (T)0.5;
  a = h * ((a + ac) + c * (a - ac) + s * (b - bc));       // S:447 (old values)
  b = h * ((b + bc) - s * (a - ac) + c * (b - bc));       // S:448 (NEW theta)
  ac = h * ((a + ac) - c * (a - ac) - s * (b - bc));      // S:449 (NEW theta, NEW p)
  bc = h * ((b + bc) + s * (a - ac) - c * (b - bc));      // S:450 (NEW theta, p, theta~)
}

// rmhmc_metric_mfma.hip: one launch per trajectory of the eigendecomposition route (fp32, D <= 112)
struct MetricTrajArgs {
  float* cur; float* th; float* pm; float* thc; float* pmc; float* H0; float* H1; float* lp1; int L; double eh; float c, s;
  // the Metropolis selection of the trajectory inside the launch (round 6; `select` = 0: the caller launches mh_select): mh_select_kernel's arguments
  const float* init; float* row; int32_t* rej; uint8_t* acc; int32_t n, burn, select, pad_;
};
extern int g_metric_traj;                                   // tuning key "metric_traj" (default 1)
extern int g_metric_resident;                               // tuning key "metric_resident" (default 1)
extern int g_metric_select;                                 // tuning key "metric_select" (default 1): the trajectory kernel ends with the chain's Metropolis selection
bool metric_traj_mfma_eligible(const MetricArgsT<float>& a);
int metric_traj_mfma(const MetricArgsT<float>& a, const MetricTrajArgs& t, hipStream_t s);

template <typename T>
int mh_select(T* cur, const T* prop, const T* init, const T* Ho, const T* Hn, const T* lpn, T* row, int32_t* rej,
              uint8_t* acc, int64_t C, int D, int n, int burn, uint64_t seed, uint64_t off, hipStream_t s);

void profile_begin(hipStream_t s);
void profile_end(hipStream_t s);

// rmhmc_fused.hip: the identity-soft-abs fast path of the Gaussian-target sampler
extern int g_rmhmc_fused;                                   // tuning key "rmhmc_fused" (default 1)
extern int g_rmhmc_batch;                                   // tuning key "rmhmc_batch" (default 1)
extern int g_rmhmc_momwave;                                 // tuning key "rmhmc_momwave" (default 1)
extern int g_rmhmc_mfma4_waves;                             // tuning key "rmhmc_mfma4_waves" (default 4; 2 = the two-wave kernel)
extern int g_rmhmc_uv;                                      // tuning key "rmhmc_uv" (default 1: rmhmc_uv.hip up to 2 x CUs chains; 0 off; 2 always)
extern int g_rmhmc_pair;                                    // tuning key "rmhmc_pair" (default 1: two half steps per K + 2 product phases)
extern int g_rmhmc_mfma4, g_rmhmc_mfma4_lo, g_rmhmc_mfma4_hi;    // tuning keys "rmhmc_mfma4" (default 1), "rmhmc_mfma4_lo", "rmhmc_mfma4_hi"
extern int g_rmhmc_overlap;                                 // tuning key "rmhmc_overlap" (default 0 since round 3)
extern int g_rmhmc_lean;                                    // tuning key "rmhmc_lean" (default 1): rmhmc_uv / rmhmc_mfma4x4 without lane-predicated stores and padding selects
extern int g_rmhmc_uv_co, g_rmhmc_uv_acc, g_rmhmc_uv_g;    // tuning keys "rmhmc_uv_co" / "rmhmc_uv_acc" / "rmhmc_uv_g" (rmhmc_uv.hip)
extern int g_rmhmc_momsplit;                                // tuning key "rmhmc_momsplit" (default 1): p = chol(P) z1 + sqrt(e) z2
template <typename T>
int fused_plan(const T* lam0_host, int D, int metric, double alpha, int has_jitter, double jitter, double* logdetP, int* series);
template <typename T> int inverse_from_eigen(const T* V0, const T* lam0, T* S, int D, hipStream_t s);
template <typename T>
int rmhmc_fused_sample(T* cur, const T* theta_init, const T* P, const T* Sinv, const T* mu, double log_norm, double logdetP,
                       int has_jitter, double jitter, int K, int series, int64_t C, int D, int L, double eps, double omega,
                       int n_traj, int traj_offset, int burn, uint64_t seed, uint64_t chain_offset, T* samples,
                       int32_t* reject_count, T* H_old, T* H_new, uint8_t* accept, T* p_ws, int64_t p_ws_elems, const T* LP,
                       hipStream_t s);

}  // namespace hta
