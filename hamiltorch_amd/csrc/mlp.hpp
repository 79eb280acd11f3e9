// Argument block shared by the two Bayesian-MLP sampler kernels (mlp_hmc.hip: VALU, any dtype/shape;
// mlp_mfma.hip: fp32 on the matrix cores).
#pragma once
#include "common.hpp"

namespace hta {

template <typename T> struct MlpArgs {
  T* theta; const T* theta_init; int64_t C;
  int n_in; int H; int act;
  const T* X; const T* Y; int N;
  int M; int Nb;
  T tau[4]; T tau_out; T prior_scale;
  int mass_kind; const T* inv_mass; const T* mass_factor;
  int L; T eps; int n_traj; int traj_offset; int burn;
  uint64_t seed; uint64_t chain_offset;
  T* samples; int32_t* reject_count; T* H_old; T* H_new; uint8_t* accept;
  T* grad_out; T* logp_out;   // evaluation-only mode (n_traj == 0): d log p_m / d theta [C, D] and log p_m [C] of split `eval_split`
  int eval_split;
};

extern int g_mlp_valu;                                        // tuning key "mlp_valu": 1 = never take the MFMA kernel
bool mlp_mfma_eligible(const MlpArgs<float>& a);
int mlp_mfma(const MlpArgs<float>& a, hipStream_t s);         // mlp_mfma.hip

}  // namespace hta
