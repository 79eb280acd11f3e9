// Argument block of the hta_netn_* entry points (multi-layer Bayesian networks): shared by csrc/netn_hmc.hip (small nets, one
// wave per chain) and csrc/mlp3_mfma.hip (two wide hidden layers on the matrix cores).
#pragma once
#include "common.hpp"

namespace hta {

constexpr int NETN_MAX_LAYERS = 4;      // Linear layers
constexpr int NETN_KMAX = 8;            // parameters per lane: D <= 512
constexpr int NETN_MAX_WIDTH = 64;
constexpr int NETN_NSET = 6;            // matrix instructions per point of the gradient: up to 96 blocks of 4 x 4 weights

template <typename T> struct NetArgs {
  T* theta; const T* theta_init; int64_t C;
  int n_layers; int dims[NETN_MAX_LAYERS + 1]; int act; int loss;
  const T* X; const T* Y; int N; int M; int Nb;
  T tau[2 * NETN_MAX_LAYERS]; T tau_out; T prior_scale;
  int mass_kind; const T* inv_mass; const T* mass_factor;
  int L; T eps; int n_traj; int traj_offset; int burn;
  uint64_t seed; uint64_t chain_offset;
  T* samples; int32_t* reject_count; T* H_old; T* H_new; uint8_t* accept;
  T* grad_out; T* logp_out; int eval_split;
  int integ;
  void* workspace; int64_t workspace_bytes;     // ABI 10: caller-owned scratch (hta_netn_hmc_workspace_bytes; mlp3 route: the momentum slots)
};

// csrc/mlp3_mfma.hip: Linear(n_in, H1)-act-Linear(H1, H2)-act-Linear(H2, 1), Gaussian likelihood, fp32, H1, H2 <= 104
extern int g_mlp3_route;                                      // tuning key "mlp3_route" (default 1)
bool mlp3_eligible(const NetArgs<float>& a);
int mlp3_mfma(const NetArgs<float>& a, hipStream_t s);
int64_t mlp3_workspace_bytes(int64_t C, int n_layers, const int* dims);     // 0 = these shapes never take the matrix-core route

}  // namespace hta
