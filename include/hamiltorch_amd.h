/* libhamiltorch_amd.so -- C ABI of the MI355X (gfx950) HMC / RMHMC sampling engine.
 *
 * This is the drop-in boundary for the per-trajectory hot path of
 * AdamCobb/hamiltorch 0.4.1 (`hamiltorch/samplers.py`, below `S:`; `hamiltorch/util.py`, `U:`).
 * The reference has no FFI -- its boundary is the Python function `hamiltorch.sample` (S:850) --
 * so these are the entry points a ctypes binding of that function calls
 * (`hamiltorch_amd/_abi.py`; INTEGRATION.md shows the reference-side stub).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (torch tensors kept alive by Python);
 *  - chain-major dense state: theta[C, D], p[C, D], row stride D, no padding;
 *  - every call only ENQUEUES work on `stream` (a hipStream_t); it never synchronises, never
 *    allocates persistent memory and never throws.  Return 0 on success, <0 on error
 *    (`hta_last_error()` gives a thread-local message);
 *  - `_f32` / `_f64` select the arithmetic type (the reference computes in the dtype of
 *    `params_init`; float32 by default);
 *  - random numbers: Philox4x32-10 keyed by (seed, chain_offset + local chain index,
 *    trajectory index n), see csrc/philox.hpp.  `chain_offset` is the global id of this
 *    buffer's first chain, so sharding chains over GPUs does not change any chain's stream.
 */
#ifndef HAMILTORCH_AMD_H
#define HAMILTORCH_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HTA_ABI_VERSION 11

#define HTA_OK 0
#define HTA_ERR_INVALID (-1)   /* bad argument                           */
#define HTA_ERR_LAUNCH (-2)    /* hipLaunch / runtime error              */
#define HTA_ERR_UNSUPPORTED (-3)

/* mass-matrix kinds: how `inv_mass` (S:283-296, S:800-814) and the momentum factor
 * (S:185-201) are stored */
#define HTA_MASS_NONE 0 /* identity; pointers ignored                                          */
#define HTA_MASS_DIAG 1 /* inv_mass[D]; mass_factor[D]   = sqrt(1/inv_mass)   (S:201, S:952)   */
#define HTA_MASS_FULL 2 /* inv_mass[D,D]; mass_factor[D,D] = chol(inverse(inv_mass)), lower,
                           row-major (S:199 via MultivariateNormal.rsample, S:950)             */

typedef struct HtaDeviceInfo {
  int abi_version;
  int device;
  int compute_units;
  int wavefront_size;
  int lds_bytes_per_cu;
  int clock_khz;
  int64_t hbm_bytes;
  char arch[64];
} HtaDeviceInfo;

int hta_abi_version(void);
const char* hta_last_error(void);
int hta_device_info(int device, HtaDeviceInfo* out);

/* ---------------------------------------------------------------------------------------------
 * Generic-callback pieces: the state updates around a user `log_prob_func` evaluated by torch
 * (S:270-278).  One launch each; grad / logp come from the caller.
 * ------------------------------------------------------------------------------------------- */

/* gibbs(): p ~ N(0, M)  (S:185-202).  p[C,D] out. */
int hta_momentum_resample_f32(float* p, int mass_kind, const float* mass_factor, int64_t C, int D,
                              uint64_t seed, uint64_t chain_offset, uint32_t draw, void* stream);
int hta_momentum_resample_f64(double* p, int mass_kind, const double* mass_factor, int64_t C, int D,
                              uint64_t seed, uint64_t chain_offset, uint32_t draw, void* stream);

/* The same draw with the trajectory index read from device memory (`*n_dev`): such a launch can be captured in a HIP
 * graph and replayed once per trajectory (hta_counter_add advances the index inside the graph). */
int hta_momentum_resample_at_f32(float* p, int mass_kind, const float* mass_factor, int64_t C, int D, uint64_t seed,
                                 uint64_t chain_offset, const int32_t* n_dev, void* stream);
int hta_momentum_resample_at_f64(double* p, int mass_kind, const double* mass_factor, int64_t C, int D, uint64_t seed,
                                 uint64_t chain_offset, const int32_t* n_dev, void* stream);
int hta_counter_add(int32_t* counter, int delta, void* stream);

/* The first lines of every run (S:954-961: params = params_init.clone(), ret_params = [params.clone()], num_rejected = 0)
 * as ONE launch instead of three: cur[C,D] <- init, row0[C,D] <- init (the stored row 0; NULL = none),
 * reject_count[C] <- 0 (NULL = none).  elem_size = 4 | 8 bytes per element (ABI 8). */
int hta_run_begin(const void* init, void* cur, void* row0, int32_t* reject_count, int64_t C, int D, int elem_size,
                  void* stream);

/* leapfrog pieces (S:281, S:283-298, S:302, and the split half-kicks S:505-520):
 *   p     += kick  * grad           (skipped when grad == NULL)
 *   theta += drift * M^-1 p         (skipped when drift == 0)
 * in this order, fused into one pass over [C,D]. */
int hta_kick_drift_f32(float* theta, float* p, const float* grad, float kick, float drift, int mass_kind,
                       const float* inv_mass, int64_t C, int D, void* stream);
int hta_kick_drift_f64(double* theta, double* p, const double* grad, double kick, double drift,
                       int mass_kind, const double* inv_mass, int64_t C, int D, void* stream);

/* hamiltonian(): H[c] = -logp[c] + 0.5 p^T M^-1 p  (S:799-815).  logp may be NULL (kinetic only). */
int hta_hamiltonian_f32(const float* p, const float* logp, int mass_kind, const float* inv_mass,
                        float* H, int64_t C, int D, void* stream);
int hta_hamiltonian_f64(const double* p, const double* logp, int mass_kind, const double* inv_mass,
                        double* H, int64_t C, int D, void* stream);

/* Metropolis test + burn/accept bookkeeping of sample() (S:1000-1026, S:1045-1057), per chain:
 *   accept iff finite(H_new), finite(logp_new) and min(0, H_old - H_new) >= log(u), u = Philox(seed, chain, n)
 *   accept: cur <- prop;  reject: cur stays, except the reference quirk: a rejection at
 *   n == burn + 1 resets cur to theta_init (S:1018 reads ret_params[-1] == params_init).
 *   n > burn: the new cur is written to samples_row[C,D] (if not NULL).
 *   reject_count[c] += !accept.   out_accept[c] (may be NULL) = accept. */
int hta_mh_select_f32(float* theta_cur, const float* theta_prop, const float* theta_init, const float* H_old,
                      const float* H_new, const float* logp_new, float* samples_row, int32_t* reject_count,
                      uint8_t* out_accept, int64_t C, int D, int n, int burn, uint64_t seed,
                      uint64_t chain_offset, void* stream);
int hta_mh_select_f64(double* theta_cur, const double* theta_prop, const double* theta_init,
                      const double* H_old, const double* H_new, const double* logp_new, double* samples_row,
                      int32_t* reject_count, uint8_t* out_accept, int64_t C, int D, int n, int burn,
                      uint64_t seed, uint64_t chain_offset, void* stream);
/* n = *n_dev (device memory); samples_base is the START of the [S, C, D] sample buffer (row n - burn is addressed by
 * the kernel when n > burn).  For launches replayed from a HIP graph. */
int hta_mh_select_at_f32(float* theta_cur, const float* theta_prop, const float* theta_init, const float* H_old,
                         const float* H_new, const float* logp_new, float* samples_base, int32_t* reject_count,
                         uint8_t* out_accept, int64_t C, int D, const int32_t* n_dev, int burn, uint64_t seed,
                         uint64_t chain_offset, void* stream);
int hta_mh_select_at_f64(double* theta_cur, const double* theta_prop, const double* theta_init, const double* H_old,
                         const double* H_new, const double* logp_new, double* samples_base, int32_t* reject_count,
                         uint8_t* out_accept, int64_t C, int D, const int32_t* n_dev, int burn, uint64_t seed,
                         uint64_t chain_offset, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused HMC for a dense Gaussian target  log p(x) = log_norm - 0.5 (x-mu)^T P (x-mu)
 * (the model family of BASELINE configs 1-2).  One launch runs `n_traj` whole trajectories
 * (gibbs S:969 -> H S:971 -> leapfrog S:281-302 -> H S:995 -> MH S:1000-1026) for all C chains,
 * with theta / p held in registers for the whole launch.
 *
 *   theta       [C,D] in/out   current state of every chain
 *   theta_init  [C,D]          params_init (only read for the Q2 reset)
 *   P [D,D] (symmetric), mu [D]
 *   samples     [S,C,D] or NULL; the state after trajectory n (> burn) is written to row n - burn
 *               (row 0 = params_init is the caller's), as in S:1007-1024
 *   reject_count[C] in/out; H_old/H_new [n_traj,C] and accept [n_traj,C] optional diagnostics
 *   trajectory indices n = traj_offset .. traj_offset + n_traj - 1  (RNG draw index and burn test)
 *   workspace   optional scratch of >= hta_hmc_gaussian_workspace_bytes(C, D, n_traj, sizeof(T)) bytes
 *               (one 16-byte-aligned record [z_0..z_{D-1}, log u, pad] per trajectory and chain).  When given (and D is in
 *               the register-resident range) all random draws of the launch are produced first by a
 *               full-chip kernel and the latency-bound trajectory kernel only loads them; results are
 *               identical to the inline-RNG path.  NULL = draw inline.
 *               With a workspace and D <= 6 (fp64: 4) the trajectories are integrated in the eigenbasis of the
 *               mass-whitened precision matrix L^-1 P L^-T, M = L L^T (2 D instead of D + D*D multiply-adds per step;
 *               fp32, D <= 4 and C <= 65536: one chain per DPP quad); the workspace then also holds the eigen block.
 *               Same map and draws; results differ from the direct form by rounding only
 *               (hta_set_tuning("gauss_eig", 0) selects the direct form).  6 < D <= 128 (fp64: 96) with identity mass:
 *               the workspace is the eigen area only (the kernels draw inline) and the wave-per-chain kernel
 *               integrates in the eigenbasis of P likewise.
 * ------------------------------------------------------------------------------------------- */
int hta_hmc_gaussian_sample_f32(float* theta, const float* theta_init, const float* P, const float* mu,
                                float log_norm, int mass_kind, const float* inv_mass,
                                const float* mass_factor, int64_t C, int D, int L, float eps, int n_traj,
                                int traj_offset, int burn, uint64_t seed, uint64_t chain_offset,
                                float* samples, int32_t* reject_count, float* H_old, float* H_new,
                                uint8_t* accept, void* workspace, int64_t workspace_bytes, void* stream);
int hta_hmc_gaussian_sample_f64(double* theta, const double* theta_init, const double* P, const double* mu,
                                double log_norm, int mass_kind, const double* inv_mass,
                                const double* mass_factor, int64_t C, int D, int L, double eps, int n_traj,
                                int traj_offset, int burn, uint64_t seed, uint64_t chain_offset,
                                double* samples, int32_t* reject_count, double* H_old, double* H_new,
                                uint8_t* accept, void* workspace, int64_t workspace_bytes, void* stream);

int64_t hta_hmc_gaussian_workspace_bytes(int64_t C, int D, int n_traj, int elem_size);

/* Once-per-TARGET setup of hta_hmc_gaussian_sample, hoisted (ABI 8; the Gaussian-HMC counterpart of
 * hta_rmhmc_gaussian_prepare).  On the eigenbasis route (workspace given, D <= 6, fp64: 4) every sample call first
 * diagonalises the mass-whitened precision matrix into the workspace's eig block - one single-wave kernel, ~5 us plus a launch
 * gap in front of the draw pass: 3 % of a 1000-trajectory call at BASELINE config 2, and the same matrix every time for a
 * caller that keeps sampling one target (hamiltorch.sample()'s loop over trajectories has no such setup: S:965-1026).
 * hta_hmc_gaussian_prepare runs that kernel into `workspace` for calls of `n_traj` trajectories (the eig block sits behind the
 * records: its position depends on n_traj) and remembers (device, eig block) -> (P, mass_kind, mass_factor, D, element type);
 * a sample call with the same workspace, n_traj, P / mass_factor POINTERS, mass kind, D and element type skips the kernel,
 * any other call runs it as before.  Contract: the CONTENTS of P / mass_factor must not change between prepare and sample
 * (prepare again, or hta_hmc_gaussian_forget), and the workspace must not be used for anything else in between.  Results are
 * bit-identical with and without.  Routes without an eig block (D > 6, "gauss_eig" 0): prepare is a no-op. */
int hta_hmc_gaussian_prepare_f32(const float* P, int mass_kind, const float* mass_factor, int64_t C, int D, int n_traj,
                                 void* workspace, int64_t workspace_bytes, void* stream);
int hta_hmc_gaussian_prepare_f64(const double* P, int mass_kind, const double* mass_factor, int64_t C, int D, int n_traj,
                                 void* workspace, int64_t workspace_bytes, void* stream);
int hta_hmc_gaussian_forget(void* workspace);

/* STATUS WORD of a prepared workspace (ABI 10).  The fused launch of the D <= 4 route (csrc/hmc_gaussian.hip:
 * hmc_gauss_quad_fused_kernel, the kernel of BASELINE config 2) hands the pre-drawn records from producer blocks to consumer blocks
 * of ONE grid; the consumers' wait is bounded, and a consumer whose bound expires (producers not scheduled: a shared or preempted GPU)
 * ORs 1 into a uint32 word inside the caller's workspace, stops waiting and - when its trajectories are done - overwrites the sample
 * rows it stored in that launch and its slot of `theta` with NaN.  The word is STICKY: launches never clear it, only
 * hta_hmc_gaussian_prepare does.  The library never synchronises; the caller reads the word (its own memory) whenever it does:
 *   byte offset of the word inside a workspace of hta_hmc_gaussian_workspace_bytes(C, D, n_traj, elem_size) bytes, or -1 for
 *   shapes without that route (D > 4, fp64).  0 = every launch since the preparation got its records.
 * The reference has no such failure mode (one chain, one process: S:965-1026); this is the error contract of the batched path.
 * Debug key "quad_starve" (hta_set_tuning) makes the producers leave at once, for tests. */
int64_t hta_hmc_gaussian_status_offset(int64_t C, int D, int n_traj, int elem_size);

/* leapfrog() only (S:267-304) for the same target: theta, p [C,D] in/out after `steps` steps.
 * path_theta / path_p: optional [steps,C,D] record of every step (the lists of S:299-300, last
 * momentum corrected as in S:302); NULL = final state only.
 * Used by the `samplers.leapfrog` mirror and the T1 parity tests. */
int hta_hmc_gaussian_leapfrog_f32(float* theta, float* p, const float* P, const float* mu, int mass_kind,
                                  const float* inv_mass, int64_t C, int D, int steps, float eps,
                                  float* path_theta, float* path_p, void* stream);
int hta_hmc_gaussian_leapfrog_f64(double* theta, double* p, const double* P, const double* mu, int mass_kind,
                                  const double* inv_mass, int64_t C, int D, int steps, double eps,
                                  double* path_theta, double* path_p, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Riemannian metric evaluation, batched: one workgroup per system, matrices in LDS.
 * Covers fisher() S:108-122 (jitter, eigh, soft-abs map, Q diag Q^T), cholesky_inverse() S:146-148,
 * rm_hamiltonian() S:710-731 and the RMHMC momentum draw S:183-184.  Every output pointer may be NULL.
 * `Hs[b]` is the NEGATIVE Hessian of log p at system b (hs_stride 0 = one matrix shared by all
 * systems, i.e. a Gaussian target); only its lower triangle is read (eigh UPLO='L', S:119).
 * D <= ~140 (fp32) / ~99 (fp64): both matrices of a system live in the 160 KiB LDS of one CU.
 * ------------------------------------------------------------------------------------------- */
#define HTA_METRIC_HESSIAN 0 /* G = Hs (must be positive definite); log|G| and solves by Cholesky  */
#define HTA_METRIC_SOFTABS 1 /* G = Q diag(lam / tanh(alpha lam)) Q^T                              */

typedef struct HtaMetricArgs {
  int64_t B; int32_t D; int32_t metric;
  const void* Hs; int64_t hs_stride;        /* elements between consecutive systems (0 or D*D)      */
  double alpha;                              /* softabs_const                                        */
  int32_t has_jitter; int32_t max_sweeps;    /* max_sweeps 0 = default Jacobi sweep cap              */
  double jitter; uint64_t seed; uint64_t chain_offset; uint32_t draw; uint32_t sub;
                                             /* Hs += diag(jitter * U(0,1)), Philox(seed, chain, draw, sub) */
  const void* X; const void* Pm; const void* mu; double log_norm;
                                             /* optional Gaussian log p / gradient at X[B,D]:
                                                logp = log_norm - 0.5 (X-mu)^T Pm (X-mu), Pd = Pm (X-mu)   */
  const void* m;                             /* [B,D]: solve x = G^-1 m                              */
  void* p_out;                               /* [B,D]: p = chol(G) z, z = Philox normals (seed, chain, draw) */
  void* x_out; void* G_out; void* lam_out; void* V_out; void* L_out;
  void* logdet_out; void* quad_out; void* H_out; void* logp_out;
                                             /* H = -logp + D/2 log 2pi + 1/2 log|G| + 1/2 m^T G^-1 m (S:731) */
  void* upd_x; double cx; void* upd_g; double cg;
                                             /* fused half step: upd_x[b,:] += cx * x ; upd_g[b,:] += cg * Pd */
  const void* V0; const void* lam0;          /* optional warm start (shared Hs only, hs_stride == 0): eigenvectors
                                                [D,D] (columns) and eigenvalues [D] of the jitter-free Hs.  The
                                                solver then diagonalises diag(lam0) + V0^T diag(jitter u) V0, which
                                                is nearly diagonal: 2 Jacobi sweeps instead of ~10; same results    */
  void* lamraw_out;                          /* [B,D] eigenvalues before the soft-abs map                           */
  void* dmetric_out;                         /* [B,D,D]: the symmetric matrix M with
                                                d/dtheta_i [1/2 log|G| + 1/2 m^T G^-1 m] = <d_i Hs, M>, i.e. what the
                                                reference gets by differentiating S:726-731 through eigh (S:398):
                                                M = Q W Q^T, W_kl = 1/2 [k==l] lam~'_k / lam~_k - 1/2 J_kl u_k u_l,
                                                u = Q^T m / lam~, J = divided differences of lam -> lam~ (Daleckii-Krein);
                                                HESSIAN metric: M = 1/2 G^-1 - 1/2 v v^T, v = G^-1 m                */
  int64_t v0_stride;                         /* (ABI 7) elements between consecutive systems' V0; 0 = one shared basis (above).
                                                D*D together with hs_stride = D*D: PER-SYSTEM warm start for general targets -
                                                V0[b] is an approximate orthonormal eigenbasis of Hs[b] (columns), typically the
                                                V_out of the previous evaluation at that chain (V_out may alias V0: the basis is
                                                updated in place); lam0 is not used.  fp32, D <= 112, SOFTABS: the evaluation runs
                                                on the matrix cores (csrc/rmhmc_metric_mfma.hip: V0^T Hs V0, iterative refinement,
                                                Jacobi inside the launch where the refinement's coupling test fails); elsewhere
                                                the hint is ignored (cold Jacobi), results agree to rounding either way.
                                                A first call passes the identity.                                    */
  void* workspace; int64_t workspace_bytes;  /* (ABI 10) caller-owned scratch of >= hta_metric_eval_workspace_bytes(B, D, sizeof(T)) bytes:
                                                needed (else HTA_ERR_INVALID) where a system's two D x D matrices exceed the 160 KiB
                                                LDS of one CU - fp64 from D = 100, fp32 from D = 141: the eigenvector matrix, from
                                                D ~ 140 / ~198 also the work matrix, live in one slab per workgroup (<= 512
                                                workgroups); to D = 254 fp32 / 180 fp64 with per-thread work lists in registers, beyond with
                                                the lists walked at run time (D <= 1024: slow - 0.4 s per call at D = 512 - never an
                                                error; the reference has no limit, S:108-122).  Rounds 3-4 allocated it behind the
                                                ABI (hipMallocAsync per call).  NULL where the function returns 0.            */
} HtaMetricArgs;

int hta_metric_eval_f32(const HtaMetricArgs* args, void* stream);
int hta_metric_eval_f64(const HtaMetricArgs* args, void* stream);
int64_t hta_metric_eval_workspace_bytes(int64_t B, int D, int elem_size);

/* Explicit RMHMC integrator (S:389-462) for a Gaussian target, `steps` steps on the augmented state
 * (theta, p, theta_copy, p_copy), all [C,D] in/out.  omega = explicit_binding_const.
 * path_theta / path_p: optional [steps,C,D] record of (theta, p) after every step (S:460-461). */
/* phi_C of the explicit integrator (S:435-450): the binding rotation by angle 2 omega eps on the augmented state,
 * in the reference's sequential update order, cos / sin rounded to float32 first.  total = C * D elements. */
int hta_rmhmc_binding_rotation_f32(float* theta, float* p, float* theta_copy, float* p_copy, int64_t total,
                                   double eps, double omega, void* stream);
int hta_rmhmc_binding_rotation_f64(double* theta, double* p, double* theta_copy, double* p_copy, int64_t total,
                                   double eps, double omega, void* stream);

int hta_rmhmc_gaussian_leapfrog_f32(float* theta, float* p, float* theta_copy, float* p_copy, const float* P,
                                    const float* mu, int metric, double alpha, int has_jitter, double jitter,
                                    uint64_t seed, uint64_t chain_offset, uint32_t draw, int64_t C, int D,
                                    int steps, double eps, double omega, float* path_theta, float* path_p,
                                    void* workspace, int64_t workspace_bytes, void* stream);
int hta_rmhmc_gaussian_leapfrog_f64(double* theta, double* p, double* theta_copy, double* p_copy, const double* P,
                                    const double* mu, int metric, double alpha, int has_jitter, double jitter,
                                    uint64_t seed, uint64_t chain_offset, uint32_t draw, int64_t C, int D,
                                    int steps, double eps, double omega, double* path_theta, double* path_p,
                                    void* workspace, int64_t workspace_bytes, void* stream);
/* (ABI 10) workspace of hta_rmhmc_gaussian_leapfrog: hta_metric_eval_workspace_bytes(C, D, sizeof(T)) bytes - 0 (NULL) up to
 * D = 99 fp64 / 140 fp32. */

/* sample(sampler=RMHMC, integrator=EXPLICIT) (S:969-1026) for a Gaussian target: enqueues every
 * launch of `n_traj` trajectories; arguments as hta_hmc_gaussian_sample.  workspace:
 * hta_rmhmc_workspace_bytes(C, D, sizeof(T)) bytes at least (required; ABI 10: includes the metric evaluations' slabs for sizes
 * beyond one CU's LDS, hta_metric_eval_workspace_bytes).  Every further C*D*sizeof(T) bytes let the
 * fused path (soft-abs map == identity on the target's spectrum, csrc/rmhmc_fused.hip) draw the momenta of one more
 * trajectory per pass ahead of the chains (one full-chip batch of draws); with the minimum it works in passes of 4.  With
 * hta_set_tuning("rmhmc_overlap", 1) and room for >= 16 trajectories the draws of the next block run on an internal side
 * stream (forked from and joined back into `stream` inside the call) under the trajectories of the current one.
 * Momenta on the fused path with jitter: p = chol(P) z1 + sqrt(jitter u) . z2 (z1, z2: normal sub-streams 0 and 1 of the
 * trajectory, u: jitter sub-stream 0), distributed as the reference's chol(P + diag(jitter u)) z; see "rmhmc_momsplit". */
int64_t hta_rmhmc_workspace_bytes(int64_t C, int D, int elem_size);
int hta_rmhmc_gaussian_sample_f32(float* theta, const float* theta_init, const float* P, const float* mu,
                                  double log_norm, int metric, double alpha, int has_jitter, double jitter,
                                  int64_t C, int D, int L, double eps, double omega, int n_traj, int traj_offset,
                                  int burn, uint64_t seed, uint64_t chain_offset, float* samples,
                                  int32_t* reject_count, float* H_old, float* H_new, uint8_t* accept,
                                  void* workspace, int64_t workspace_bytes, void* stream);
int hta_rmhmc_gaussian_sample_f64(double* theta, const double* theta_init, const double* P, const double* mu,
                                  double log_norm, int metric, double alpha, int has_jitter, double jitter,
                                  int64_t C, int D, int L, double eps, double omega, int n_traj, int traj_offset,
                                  int burn, uint64_t seed, uint64_t chain_offset, double* samples,
                                  int32_t* reject_count, double* H_old, double* H_new, uint8_t* accept,
                                  void* workspace, int64_t workspace_bytes, void* stream);

/* Once-per-TARGET setup of hta_rmhmc_gaussian_sample, hoisted out of it for callers that keep sampling one target in
 * several calls (a long run cut into blocks, bench.py's steps): the cold eigendecomposition of the curvature matrix P, the
 * host-side plan of the fused route (one D-element read-back and a stream synchronise) and the shared inverse are written
 * into `workspace` (the same buffer the sample calls get; >= hta_rmhmc_workspace_bytes) and remembered for it.  A later
 * hta_rmhmc_gaussian_sample on that workspace with the SAME P pointer, D, metric, alpha, jitter and element type skips
 * them (1.2 ms, 14 % of a 100-trajectory call at 1024 chains and D = 100); anything else runs the setup as before.
 * Contract: the CONTENTS of P must not change between prepare and sample (prepare again, or hta_rmhmc_gaussian_forget),
 * and the workspace must not be used for anything else in between.  Results are bit-identical with and without. */
int hta_rmhmc_gaussian_prepare_f32(const float* P, const float* mu, int metric, double alpha, int has_jitter, double jitter,
                                   int64_t C, int D, void* workspace, int64_t workspace_bytes, void* stream);
int hta_rmhmc_gaussian_prepare_f64(const double* P, const double* mu, int metric, double alpha, int has_jitter, double jitter,
                                   int64_t C, int D, void* workspace, int64_t workspace_bytes, void* stream);
int hta_rmhmc_gaussian_forget(void* workspace);

/* ---------------------------------------------------------------------------------------------
 * Fused (split-)HMC for a one-hidden-layer Bayesian MLP with one output (BASELINE config 4):
 * the closures of define_model_log_prob / define_split_model_log_prob (S:1093-1258) and the
 * SPLITTING integrator (S:499-540; M == 1: plain leapfrog S:281-302) for C chains, one workgroup
 * per chain.  theta[C, D], D = H*n_in + 2H + 1 in nn.Module.parameters() order (W1, b1, W2, b2).
 *   X[N, n_in], Y[N]: the M splits are rows [m*Nb, (m+1)*Nb).  act: 0 relu, 1 tanh, 2 sigmoid.
 *   loss_kind: HTA_LOSS_REGRESSION  -1/2 tau_out sum (f - y)^2                         (model_loss 'regression', S:1184)
 *              HTA_LOSS_BINARY_LOGITS -tau_out sum [softplus(f) - y f], y in [0, 1]      ('binary_class_linear_output':
 *                                    binary_cross_entropy_with_logits, reduction 'sum', S:1172)
 *   tau4: HOST pointer to the 4 prior precisions (W1, b1, W2, b2); prior_scale divides the prior of
 *   every split closure (S:1199): num_splits for sample_split_model, 1 for sample_model.
 *   mass_kind: HTA_MASS_NONE or HTA_MASS_DIAG (flat [D] operands).
 *   integrator: HTA_SPLIT_SYMMETRIC = Integrator.SPLITTING (S:499-540), HTA_SPLIT_RAND = SPLITTING_RAND
 *   (S:547-566; the subset order is a Philox permutation per (seed, trajectory), M <= 64),
 *   HTA_SPLIT_KMID = SPLITTING_KMID (S:572-596, M >= 2).
 * Remaining arguments as hta_hmc_gaussian_sample. */
#define HTA_LOSS_REGRESSION 0
#define HTA_LOSS_BINARY_LOGITS 1
#define HTA_LOSS_SOFTMAX_CE 2     /* hta_netn_* only: -tau_out sum_p [logsumexp(f_p) - f_p[y_p]], y_p a class index stored as a float
                                   * ('multi_class_linear_output': CrossEntropyLoss(reduction='sum') on logits, samplers.py:1173-1178) */
#define HTA_SPLIT_SYMMETRIC 0
#define HTA_SPLIT_RAND 1
#define HTA_SPLIT_KMID 2
int hta_mlp_hmc_sample_f32(float* theta, const float* theta_init, int64_t C, int n_in, int H, int act, int loss_kind,
                           const float* X, const float* Y, int N, int M, int Nb, const float* tau4, float tau_out,
                           float prior_scale, int mass_kind, const float* inv_mass, const float* mass_factor,
                           int integrator, int L, float eps, int n_traj, int traj_offset, int burn, uint64_t seed,
                           uint64_t chain_offset, float* samples, int32_t* reject_count, float* H_old,
                           float* H_new, uint8_t* accept, void* stream);
int hta_mlp_hmc_sample_f64(double* theta, const double* theta_init, int64_t C, int n_in, int H, int act, int loss_kind,
                           const double* X, const double* Y, int N, int M, int Nb, const double* tau4,
                           double tau_out, double prior_scale, int mass_kind, const double* inv_mass,
                           const double* mass_factor, int integrator, int L, double eps, int n_traj,
                           int traj_offset, int burn, uint64_t seed, uint64_t chain_offset, double* samples,
                           int32_t* reject_count, double* H_old, double* H_new, uint8_t* accept, void* stream);
/* value and gradient of ONE split closure (S:1145-1199) for every chain: grad_out[C, D], logp_out[C]. */
int hta_mlp_logp_grad_f32(const float* theta, int64_t C, int n_in, int H, int act, int loss_kind, const float* X, const float* Y,
                          int N, int M, int Nb, int split, const float* tau4, float tau_out, float prior_scale,
                          float* grad_out, float* logp_out, void* stream);
int hta_mlp_logp_grad_f64(const double* theta, int64_t C, int n_in, int H, int act, int loss_kind, const double* X,
                          const double* Y, int N, int M, int Nb, int split, const double* tau4, double tau_out,
                          double prior_scale, double* grad_out, double* logp_out, void* stream);

/* ---- Bayesian networks of any small shape (csrc/netn_hmc.hip; ABI 4) ---------------------------------------------------------
 * The same two entry points for fully connected nets of 1 .. 4 Linear layers (none to three hidden layers, one activation kind
 * between them, every width <= 64, at most 512 parameters), one wave per chain: the models of the reference's notebooks that the
 * one-hidden-layer kernels above do not cover - Net([1, 10, 10, 1]) of the split-HMC notebook, the softmax regression
 * Linear(4, 3) with `multi_class_linear_output` (sample_model's default model_loss) of the BNN notebook.  Replaces, like
 * hta_mlp_*, `define_model_log_prob` / `define_split_model_log_prob` + the trajectory loop of `sample`
 * (hamiltorch/samplers.py:1093-1258, :965-1026, :499-596, :281-302).
 *   dims   HOST pointer to n_layers + 1 ints: input width, the layers' output widths
 *   taus   HOST pointer to 2 n_layers prior precisions (weight, bias per layer: the reference's tau_list, util.py:121-122 order)
 *   theta  [C, D], D = sum_l dims[l] dims[l+1] + dims[l+1]: per layer weight[out, in] row-major, then bias[out]
 *   Y      [N, dims[n_layers]] for HTA_LOSS_REGRESSION / HTA_LOSS_BINARY_LOGITS (summed over the outputs),
 *          [N] class indices stored as floats for HTA_LOSS_SOFTMAX_CE
 * everything else as in hta_mlp_hmc_sample / hta_mlp_logp_grad. */
int hta_netn_hmc_sample_f32(float* theta, const float* theta_init, int64_t C, int n_layers, const int* dims, int act, int loss_kind,
                            const float* X, const float* Y, int N, int M, int Nb, const float* taus, float tau_out,
                            float prior_scale, int mass_kind, const float* inv_mass, const float* mass_factor, int integrator,
                            int L, float eps, int n_traj, int traj_offset, int burn, uint64_t seed, uint64_t chain_offset,
                            float* samples, int32_t* reject_count, float* H_old, float* H_new, uint8_t* accept, void* workspace,
                            int64_t workspace_bytes, void* stream);
int hta_netn_hmc_sample_f64(double* theta, const double* theta_init, int64_t C, int n_layers, const int* dims, int act, int loss_kind,
                            const double* X, const double* Y, int N, int M, int Nb, const double* taus, double tau_out,
                            double prior_scale, int mass_kind, const double* inv_mass, const double* mass_factor, int integrator,
                            int L, double eps, int n_traj, int traj_offset, int burn, uint64_t seed, uint64_t chain_offset,
                            double* samples, int32_t* reject_count, double* H_old, double* H_new, uint8_t* accept, void* workspace,
                            int64_t workspace_bytes, void* stream);
int hta_netn_logp_grad_f32(const float* theta, int64_t C, int n_layers, const int* dims, int act, int loss_kind, const float* X,
                           const float* Y, int N, int M, int Nb, int split, const float* taus, float tau_out, float prior_scale,
                           float* grad_out, float* logp_out, void* workspace, int64_t workspace_bytes, void* stream);
int hta_netn_logp_grad_f64(const double* theta, int64_t C, int n_layers, const int* dims, int act, int loss_kind, const double* X,
                           const double* Y, int N, int M, int Nb, int split, const double* taus, double tau_out, double prior_scale,
                           double* grad_out, double* logp_out, void* workspace, int64_t workspace_bytes, void* stream);
/* (ABI 10) `workspace`: caller-owned scratch of >= hta_netn_hmc_workspace_bytes(C, n_layers, dims, sizeof(T)) bytes - non-zero only for
 * the shapes of the matrix-core route (csrc/mlp3_mfma.hip: two hidden layers of <= 104 units, n_in <= 4, one output, fp32: the momentum
 * slots of its workgroups; rounds 3-4 kept a hipMalloc'ed buffer per stream behind the ABI).  NULL / 0 elsewhere. */
int64_t hta_netn_hmc_workspace_bytes(int64_t C, int n_layers, const int* dims, int elem_size);

/* Posterior predictive: out[s, p, :] = f(x_p; theta_s) for S parameter vectors at once (forward only) - replaces the per-sample
 * forward passes of hamiltorch.predict_model (hamiltorch/samplers.py:1530-1552: a Python loop over the samples, one functional
 * call each); SURVEY 8(f) N2.  theta [S, D]: rows of the samples sample_model / sample_split_model returned (flattened in
 * model.parameters() order, U:121-122: weight [out, in] row-major then bias [out] per Linear); dims [n_layers + 1] = (in, h1, ...,
 * out); act 0 relu | 1 tanh | 2 sigmoid between the layers (none after the last); X [N, dims[0]]; out [S, N, dims[n_layers]].
 * Nets with one hidden layer and <= 16 outputs: any width; deeper nets: widths <= 256, <= 8 layers.  The log-probabilities
 * predict_model also returns are element-wise reductions of `out` (hamiltorch_amd/bnn.py). */
int hta_net_forward_f32(const float* theta, int64_t S, int n_layers, const int* dims, int act, const float* X, int N, float* out,
                        void* stream);
int hta_net_forward_f64(const double* theta, int64_t S, int n_layers, const int* dims, int act, const double* X, int N, double* out,
                        void* stream);

/* ---------------------------------------------------------------------------------------------
 * Compiled callbacks (ABI 11): an OPAQUE user `log_prob_func` (the callback contract, S:272-274) inside fused kernels.
 * hamiltorch_amd/jit/ traces the callable once (torch.fx), differentiates the trace itself and writes value + gradient
 * (`params_grad`, S:270-278, S:33-66; for the Riemannian samplers also the Hessian of S:108 and the third-derivative
 * contraction behind S:397-398) as straight-line device code; that text is compiled here by hipRTC for gfx950 INTO the
 * hand-written kernels of csrc/jit/ (hmc_callback.hip.in: the sample() loop of S:965-1026 - momentum draw, leapfrog,
 * both energies, Metropolis, burn-in / Q2 bookkeeping, row stores - one chain per lane, a block of trajectories per launch).
 *
 *   hta_jit_compile   host work: HIP source (+ named headers, compiler options) -> malloc'ed gfx950 code object
 *                     (hta_jit_free releases it; needs no GPU).  On HTA_ERR_INVALID hta_jit_last_log() holds the compiler's
 *                     messages.  HTA_ERR_UNSUPPORTED: libhiprtc.so is not loadable.
 *   hta_jit_load      code object -> module on the CURRENT device (a preparation step: reads the module's info block back
 *                     once); hta_jit_unload frees it.  hta_jit_module_info: {magic, D, sizeof(T), mass kind, kernel set,
 *                     scalar operations, 0, 0}.
 *   hta_jit_hmc_sample  trajectories [traj_offset, traj_offset + n_traj) of plain HMC for C chains; the argument block is
 *                     HtaCbHmcArgs (csrc/jit/jit_args.h; `gcur` / `lp_out` are placed in `workspace`,
 *                     hta_jit_hmc_workspace_bytes(C, D, itemsize) bytes).  Same streams, rules and outputs as the pieces
 *                     above; D / itemsize / mass_kind must be the module's (else HTA_ERR_INVALID, nothing is launched).
 *   hta_jit_derivs    which = 0: logp[C], grad[C,D], neg_hess[C,D,D] (each optional) at theta[C,D]; which = 1:
 *                     contract[C,D] = d_k < Hess log p, M > with M[C,D,D] held fixed.
 *   hta_jit_note_fallback  records in hta_last_route() WHY a callable was not compiled (the caller then runs the pieces path).
 * Launches are enqueued on `stream`; nothing is allocated or synchronised on the launch path. */
#include "../hamiltorch_amd/csrc/jit/jit_args.h"
int hta_jit_available(void);
const char* hta_jit_last_log(void);
int hta_jit_note_fallback(const char* reason);
int hta_jit_compile(const char* source, const char* name, int n_headers, const char* const* header_names,
                    const char* const* header_sources, int n_options, const char* const* options, void** code_out,
                    int64_t* code_bytes);
void hta_jit_free(void* code);
int hta_jit_load(const void* code, int64_t bytes, void** module_out);
int hta_jit_unload(void* module);
int hta_jit_module_info(void* module, int* info_out);
int64_t hta_jit_hmc_workspace_bytes(int64_t C, int D, int itemsize);
/* bytes of HtaCbHmcArgs::pre for a launch of n_traj trajectories: with it the launch's momentum draws and log-uniforms are produced by a
 * kernel of their own in front of the trajectory kernel (the whole GPU instead of the chains' few waves; bit-identical results) */
int64_t hta_jit_hmc_predraw_bytes(int64_t C, int D, int n_traj, int itemsize);
int hta_jit_hmc_sample(void* module, const HtaCbHmcArgs* args, int D, int itemsize, int mass_kind, void* workspace,
                       int64_t workspace_bytes, void* stream);
int hta_jit_derivs(void* module, const HtaCbDerivArgs* args, int which, int D, int itemsize, void* stream);
/* Explicit RMHMC (S:389-462 inside the RMHMC branch of sample(), S:969-1026; Metric.SOFTABS) for a GENERAL target of small dimension
 * (D <= 16) on a compiled callable: trajectories [traj_offset, traj_offset + n_traj), a chain per lane, the 8 L + 3 metric evaluations
 * of a trajectory (Hessian, jitter, eigendecomposition, soft-abs map, solves, the derivative of the metric) in the lane's registers -
 * csrc/jit/rmhmc_callback.hip.in.  `workspace`: hta_jit_rmhmc_workspace_bytes(C, D, itemsize) bytes (log p at the final states). */
int64_t hta_jit_rmhmc_workspace_bytes(int64_t C, int D, int itemsize);
int hta_jit_rmhmc_sample(void* module, const HtaCbRmhmcArgs* args, int D, int itemsize, int has_jitter, void* workspace,
                         int64_t workspace_bytes, void* stream);

/* measurement knobs: "small_chains_per_block" (launch shape of the thread-per-chain kernel), "fill_blocks" (grid cap of the
 * pre-draw pass of the Gaussian path: 256-thread blocks, grid-stride; 4096),
 * "force_general" (route small D through the wave-per-chain kernel), "profile" (N > 0 = record a HIP
 * event pair around the dominant kernel of every N-th fused call, on the launch stream);
 * route selectors kept for the parity tests: "gauss_eig" (1 default; 0 = direct small-D Gaussian kernel, 2 = eigenbasis with
 * one chain per lane only, 3 = quad kernel without the compiled-in step counts), "quad_max_chains" (65536), "rmhmc_fused"
 * (1 default; 0 = per-evaluation Jacobi path, 3 = two chains per workgroup), "rmhmc_overlap" (0 default: momentum draws
 * on the caller's stream; 1 = on an internal side stream under the previous block's trajectories - slower at every measured chain
 * count, the two kernels compete for the same SIMDs), "rmhmc_momsplit" (1 default: with jitter on, the fused route draws
 * p = chol(P) z1 + sqrt(jitter u) . z2 - covariance P + diag(jitter u) = G exactly, the law of S:183-184, chol(P) once per target;
 * 0 = chol(G) z with a factorisation per draw, the reference's arithmetic), "rmhmc_batch" (1 default: 16 chains per workgroup on the matrix cores from 2048 chains on; 0 off, 2 always),
 * "rmhmc_mfma4" (1 default: 4 chains per workgroup on v_mfma_f32_4x4x1_16b for "rmhmc_mfma4_lo" = 1793 (round 4; before: 513) <= chains < "rmhmc_mfma4_hi" = 2049;
 * 0 off, 2 always; "rmhmc_mfma4_waves" 4 default: four waves per group - rows x contraction parity inside a wave; 2 = two waves;
 * "netn_waves" 1 default: waves per chain of the small-network kernel (2 / 4: a chain's sweeps over several waves where they fit);
 * "rmhmc_uv" 1 default: up to 7 x (compute units) chains (2 x with "rmhmc_uv_co" = 0) run one or two per workgroup with their state sets as columns of the
 * matrix instruction - csrc/rmhmc_uv.hip; 0 off, 2 at any chain count;
 * "rmhmc_pair" 1 default: two consecutive half steps share K + 2 product phases, 0 = one half step at a time), "rmhmc_wide" (1 default: the spill-free one-workgroup-per-CU
 * instances of the one-chain kernel when chains <= compute units; 0 = always the two-workgroups-per-CU instances), "rmhmc_momwave" (1 default: one wave per momentum draw, fp32 with jitter, D <= 104; 0 = one workgroup per draw), "mlp_valu" (1 = VALU MLP kernel instead of the MFMA one),
 * "metric_mfma" (1 default: fp32 metric evaluations that share an eigenbasis, and Metric.HESSIAN ones, run on the matrix cores -
 * csrc/rmhmc_metric_mfma.hip, D <= 112; 0 = always the Jacobi kernel of csrc/rmhmc_metric.hip),
 * "metric_general" (1 default: evaluations with per-system curvature and per-system warm bases - HtaMetricArgs::v0_stride - run on
 * that kernel too; 0 = the Jacobi kernel, cold),
 * "metric_traj" (round 4; 1 default: a trajectory of hta_rmhmc_gaussian_sample's eigendecomposition route - momentum draw, H_old, 4 L
 * half steps with the binding rotation, H_new - is ONE launch of metric_traj_mfma_kernel per trajectory (fp32, D <= 112: each chain's
 * workgroup runs the chain's 4 L + 3 evaluations back to back) + the accept/reject launch; 0 = one launch per evaluation, the
 * parity partner: bit-identical results),
 * "metric_second" (round 4; 1 default: where the first refinement pass of a matrix-core metric evaluation moves no eigenvector entry by
 * more than 8e-3, the second pass is second-order perturbation theory in closed form - ONE product F E1 instead of A X, X^T A X and
 * X^T X; truncation error |F| d^2 < 1e-8 in ||A X - X Lam||, below fp32 rounding; 0 = always the three-product pass: the parity
 * partner, equal results to rounding),
 * "metric_sqrtdraw" (round 6; 1 default: the momentum draw of a soft-abs evaluation on a shared basis - HtaMetricArgs::p_out without G_out /
 * V_out / dmetric_out, V0 / lam0 given, v0_stride 0; the draw of hta_rmhmc_gaussian_sample's eigendecomposition route - is p = G^(1/2) z
 * with the SYMMETRIC square root Q diag(sqrt lam~) Q^T: a solve-shaped evaluation (no assembly of G, no Cholesky: 30 k cycles instead of
 * 350 k at D = 100).  The same law N(0, G(theta)) as samplers.py:183-184's chol(G) z from the same z, a different map (as "rmhmc_momsplit"
 * on the fused routes); 0 = chol(G) z),
 * "metric_select" (round 6; 1 default: that trajectory kernel ends with the chain's Metropolis selection - hta_mh_select's rule, uniform and
 * writes, by the chain's own workgroup: ONE launch per trajectory; 0 = the selection is a second launch; bit-identical either way),
 * "metric_resident" (round 6; 1 default: that trajectory kernel takes the chain's four state vectors into the eigenbasis once after the
 * momentum draw - theta' = V0^T (theta - mu), p' = V0^T p - and keeps them in LDS: a solve evaluation has no V0 product and no global
 * traffic but its scalars, theta = mu + V0 theta' at the end; agreement with the launch sequence to fp32 rounding; 0 = the state in the
 * caller's coordinates in global memory: bit-identical to the launch sequence),
 * "metric_bx3" (round 6; 2 default: in the SOLVE evaluations of a Gaussian target on the shared basis - the fast sequence of
 * csrc/rmhmc_metric_mfma.hip: V0 resident, the element-wise passes in the products' epilogues, log p and P d from the eigenbasis -
 * both D^3 products run as three bfloat16 products of operands split hi + lo: the closed-form second pass's F E1 (a correction that is
 * itself below 1e-4: its 2^-16 relative error is under fp32 rounding of the first-order terms) and the formation F = W^T W on the planes
 * of W = diag(sqrt e) V0 (a term of F with relative error 2^-16 moves a solve by 4e-9 of its size in a float64 emulation - less than the
 * closed-form pass's own truncation; tools/scratch/bf16_formation_err.py); 1 = F E1 only, the formation in exact fp32; 0 = both in exact
 * fp32: the parity partner),
 * "mlp3_route" (1 default: Bayesian MLPs with two wide hidden layers run on csrc/mlp3_mfma.hip; 0 = callback path),
 * "quad_variant" (7 default: the quad kernel with wave-uniform base addresses + 32-bit lane offsets, without the NaN guard of
 * the accept compare, and with the row element and the energy butterfly in one interleaved block; 3 = without that block;
 * 0 = the round-1 instance.  Bit-identical results; 78.5 / 74.1 / 73.25 instructions per trajectory of a lone wave at L = 25:
 * 166.3 -> 157.5 us per 1000 trajectories of BASELINE config 2),
 * "rmhmc_lean" (1 default: rmhmc_uv_kernel without the lane predicate around its LDS stores - both lane halves hold the same
 * values - and, with rmhmc_mfma4x4_kernel, without the selects that zero the padding rows, which are exact zeros by
 * construction: the same results bit for bit, ~30 instructions fewer per step of these one-wave-per-SIMD kernels (+1.6 % / +1.2 %);
 * 0 = the round-2 instances, the parity partners),
 * "rmhmc_uv_co" (round 4; 1 default = rmhmc_uv_kernel / rmhmc_uvc*_kernel under a 256-register cap, two workgroups per CU: the phase latency of one is
 * filled by the other's matrix instructions; bit-identical to 0; also extends the kernel's range to 4 x CUs chains),
 * "rmhmc_uv_acc" (2 | 4 accumulator chains per product of rmhmc_uv_kernel; 4 needs no s_nop between dependent matrix
 * instructions; another summation order, equal to rounding),
 * "rmhmc_uv_g" (0 = chains per workgroup by chain count; 1 or 2 force it),
 * "rmhmc_uvc" (round 4; 1 default = one-chain workgroups with K == 2 refinements and jitter run on rmhmc_uvc_kernel: one value per
 * lane in the element-wise work and three product phases per step - the second-order term of a solve rides in the idle columns
 * of the next phase's matrix instructions; equal to the K = 2 iteration up to third-order terms in jitter / lambda_min; two-chain
 * workgroups run on rmhmc_uvc2_kernel: two values per lane, branch-free half steps, the schedule of rmhmc_uv_kernel<2>;
 * 0 = rmhmc_uv_kernel, the parity partner of both),
 * "quad_fused" (round 4; 1 default = on a PREPARED workspace, up to 4096 chains whose record rows are whole 128-byte lines, the quad
 * route of hta_hmc_gaussian_sample produces its draw records inside the trajectory launch - producer blocks behind the consumer
 * blocks, write-through stores, one counter per chunk of whole trajectories, the consumers' look-ahead gated per pass - instead
 * of a pre-draw launch in front of it: bit-identical results, BASELINE config 2 0.175 -> 0.151 ms per call; 0 = the two-launch
 * form; "quad_producers": producer blocks of that launch per 1024 chains, 64; "quad_chunks": hand-over chunks per launch, 8). */
int hta_set_tuning(const char* key, int value);
/* current value of a route key; every key back to its default (test fixtures call this between tests: the keys are
 * process-global).  The environment variable HTA_TUNING_DEFAULTS="key=value,..." moves the DEFAULT of the named keys for the
 * process (read when the library is loaded and by hta_reset_tuning): A/B runs of unmodified callers under another route. */
int hta_get_tuning(const char* key, int* value);
int hta_reset_tuning(void);
/* Name (with template arguments) of the dominant kernel the calling thread's last sampling / evaluation call dispatched
 * to, e.g. "hmc_gauss_quad_kernel<3,false,25>", "rmhmc_uv_kernel<1>", "rmhmc_mfma4x4_kernel<true>", "mlp_mfma_kernel<2,7,0,512>".
 * The string lives in thread-local storage until the next call.  Parity tests and bench.py assert / report the route
 * they exercise with it (the dispatch depends on the chain count and on the process-global tuning keys). */
const char* hta_last_route(void);
/* waits for the recorded event pairs; returns their summed elapsed time and count, then resets. */
int hta_profile_collect(double* total_ms, int* launches);

#ifdef __cplusplus
}
#endif
#endif /* HAMILTORCH_AMD_H */
