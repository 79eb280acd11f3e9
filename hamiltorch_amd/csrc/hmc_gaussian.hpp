// Argument block shared by the fused Gaussian-HMC kernels (hmc_gaussian.hip, hmc_gaussian_quad.hip).
#pragma once
#include "common.hpp"

namespace hta {

template <typename T> struct GaussArgs {
  T* theta; const T* theta_init; const T* P; const T* mu; T log_norm;
  const T* inv_mass; const T* mass_factor;
  int64_t C; int D; int L; T eps; int n_traj; int traj_offset; int burn;
  uint64_t seed; uint64_t chain_offset;
  T* samples; int32_t* reject_count; T* H_old; T* H_new; uint8_t* accept;
  T* p_io;  // leapfrog-only entry: momentum in/out
  T* path_theta; T* path_p;  // leapfrog-only: optional per-step record [steps,C,D] (S:299-300)
  T* ws_z; T* ws_logu;       // optional pre-drawn records [n_traj,C,(z_0..z_{D-1}, log u, pad)]; ws_logu: eig block (lam, Q) or NULL
};

extern int g_small_chains_per_block;
extern int g_force_general;
extern int g_gauss_eig;
extern int g_quad_max_chains;
extern int g_quad_variant;                                  // tuning key "quad_variant" (default 7; 0 / 3: see hmc_gauss_quad_kernel's VAR)
extern int g_fill_blocks;
void profile_begin(hipStream_t s);
void profile_end(hipStream_t s);

}  // namespace hta
