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
  int integ;                  // HTA_SPLIT_SYMMETRIC (plain leapfrog when M == 1) / HTA_SPLIT_RAND / HTA_SPLIT_KMID
  int loss;                   // HTA_LOSS_REGRESSION / HTA_LOSS_BINARY_LOGITS
};

// Stage st of a trajectory: which split's gradient, the kick it scales and the drift that follows it.
//   symmetric (S:499-540): per step m = 0..M-1, M-1..0, half kicks, drift eps / (2 (M-1)) except at the two turning points;
//     M == 1: plain leapfrog (S:281-302), stage 0 kicks eps/2, stages 1..L kick eps (the last half kick is taken back after);
//   RAND (S:547-566): per step, for each subset in the trajectory's random order: half kick, drift eps / M, half kick;
//   KMID (S:572-596): per step M half kicks, ONE drift of eps, M half kicks in reverse order.
template <typename T>
__device__ __forceinline__ void split_stage(int integ, int M, int L, int st, T eps, const int* perm, int& m, T& kick, T& dr) {
  const T heps = (T)0.5 * eps;
  if (integ == HTA_SPLIT_SYMMETRIC && M == 1) { m = 0; kick = (st == 0) ? heps : eps; dr = (st < L) ? eps : (T)0; return; }
  const int s2 = st % (2 * M);
  kick = heps;
  if (integ == HTA_SPLIT_RAND) { m = perm[s2 >> 1]; dr = (s2 & 1) ? (T)0 : eps / (T)M; return; }
  m = (s2 < M) ? s2 : 2 * M - 1 - s2;
  if (integ == HTA_SPLIT_KMID) dr = (s2 == M - 1) ? eps : (T)0;
  else dr = (s2 == M - 1 || s2 == 2 * M - 1) ? (T)0 : eps / (T)((M - 1) * 2);
}
// The same table without the per-stage integer modulo and floating-point divisions (round 6: on BASELINE config 4 a gradient pass is 9.5 k
// cycles and the stage bookkeeping between two passes - `st % (2 M)` and `eps / (2 (M - 1))` expand to ~60 vector instructions that all 14
// waves of the CU execute - was a fifth of the kernel's time, profiles/r06l_cfg4_ticks.txt): the plan holds the quotients, computed once with
// the same fp32 operations (bit-identical stages), and the caller carries s2 = st mod 2 M along (split_next_s2).
template <typename T> struct SplitPlan { int integ, M, L, M2; T eps, heps, epsM, epsS; bool plain; };
template <typename T> __device__ __forceinline__ SplitPlan<T> split_plan(int integ, int M, int L, T eps) {
  SplitPlan<T> p;
  p.integ = integ; p.M = M; p.L = L; p.M2 = 2 * M; p.eps = eps; p.heps = (T)0.5 * eps;
  p.epsM = eps / (T)M; p.epsS = M > 1 ? eps / (T)((M - 1) * 2) : (T)0;
  p.plain = integ == HTA_SPLIT_SYMMETRIC && M == 1;
  return p;
}
__device__ __forceinline__ int split_next_s2(int s2, int M2) { return s2 + 1 == M2 ? 0 : s2 + 1; }
template <typename T>
__device__ __forceinline__ void split_stage_at(const SplitPlan<T>& p, int st, int s2, const int* perm, int& m, T& kick, T& dr) {
  if (p.plain) { m = 0; kick = (st == 0) ? p.heps : p.eps; dr = (st < p.L) ? p.eps : (T)0; return; }
  kick = p.heps;
  if (p.integ == HTA_SPLIT_RAND) { m = perm[s2 >> 1]; dr = (s2 & 1) ? (T)0 : p.epsM; return; }
  m = (s2 < p.M) ? s2 : p.M2 - 1 - s2;
  if (p.integ == HTA_SPLIT_KMID) dr = (s2 == p.M - 1) ? p.eps : (T)0;
  else dr = (s2 == p.M - 1 || s2 == p.M2 - 1) ? (T)0 : p.epsS;
}
// Gradient reuse between stages (round 3).  A stage that kicks WITHOUT a drift is followed, in the reference's loops, by a stage
// that evaluates the SAME subset at the SAME parameters: the turning point of the symmetric scheme (m = M-1 closes the forward
// sweep, S:501-517, and opens the backward one, S:519-535) and its step boundary (m = 0 closes a step and opens the next);
// SPLITTING_KMID's step boundary likewise.  The reference differentiates twice and gets the same gradient twice; the kernels
// evaluate it once and apply both kicks, (p + k1 g) + k2 g - bit for bit what two evaluations give, with (2M - 2) L + 1
// instead of 2 M L gradient passes per trajectory (M = 4: a quarter fewer).  `split_stage_reuses(prev_m, prev_dr, m)`.
template <typename T> __host__ __device__ inline bool split_stage_reuses(int prev_m, T prev_dr, int m) {
  return prev_dr == (T)0 && prev_m == m;
}
// gradient passes a trajectory executes with that reuse (bench.py's executed-flop count)
__host__ inline int split_gradient_passes(int integ, int M, int L) {
  if (integ == HTA_SPLIT_SYMMETRIC && M == 1) return L + 1;
  if (integ == HTA_SPLIT_RAND) return 2 * M * L;
  if (integ == HTA_SPLIT_KMID) return 2 * M * L - (L - 1);
  return (2 * M - 2) * L + 1;
}
__host__ __device__ inline int split_stage_count(int integ, int M, int L) { return (integ == HTA_SPLIT_SYMMETRIC && M == 1) ? L + 1 : L * 2 * M; }

// The likelihood of one point (S:1170-1184) as (delta, e): delta = d log-lik / d f and e with log-lik = -1/2 tau_out e, so that both
// kernels keep ONE accumulation (sum of e) and ONE scaling whatever the likelihood.
//   regression:        r = f - y; delta = -tau_out r;               e = r^2
//   Bernoulli, logits: delta = -tau_out (sigmoid(f) - y);           e = 2 (softplus(f) - y f)
// (the Bernoulli branch is OUT OF LINE: inlined, its exp / log1p sequences raise the register pressure of the sampler kernels,
//  which run at a 128-register cap, for every likelihood - 4 % on BASELINE config 4)
template <typename T> __device__ __attribute__((noinline)) void mlp_point_loss_binary(T f, T y, T tau_out, T& delta, T& e) {
  const T ex = exp(-fabs(f));                              // in (0, 1]: no overflow
  const T sg = f >= (T)0 ? (T)1 / ((T)1 + ex) : ex / ((T)1 + ex);
  delta = -tau_out * (sg - y);
  e = (T)2 * (fmax(f, (T)0) - y * f + log1p(ex));
}
template <typename T> __device__ __forceinline__ void mlp_point_loss(int loss, T f, T y, T tau_out, T& delta, T& e) {
  if (loss == HTA_LOSS_BINARY_LOGITS) {
    mlp_point_loss_binary<T>(f, y, tau_out, delta, e);
  } else {
    const T r = f - y;
    delta = -tau_out * r;
    e = r * r;
  }
}

extern int g_mlp_valu;                                        // tuning key "mlp_valu": 1 = never take the MFMA kernel
bool mlp_mfma_eligible(const MlpArgs<float>& a);
int mlp_mfma(const MlpArgs<float>& a, hipStream_t s);         // mlp_mfma.hip


}  // namespace hta
