/* Kernel-argument blocks of the run-time compiled callback kernels: ONE definition, seen by the library (csrc/jit_runtime.cpp
 * fills them) and by the hipRTC translation unit (csrc/jit/hmc_callback.hip.in reads them).  Plain C, fixed-width fields,
 * no padding surprises: pointers first, then 8-byte scalars, then 4-byte ones. */
#ifndef HTA_JIT_ARGS_H
#define HTA_JIT_ARGS_H

#define HTA_CB_INFO_WORDS 8 /* hta_cb_info[]: {magic, D, sizeof(T), mass kind, kernel set, n_nodes, 0, 0} */
#define HTA_CB_MAGIC 0x48544131 /* "HTA1" */

/* kernel sets (hta_cb_info[4]): which entry points the module exports */
#define HTA_CB_SET_HMC 1    /* hta_cb_hmc_kernel                                            */
#define HTA_CB_SET_DERIVS 2 /* hta_cb_derivs_kernel + hta_cb_contract_kernel (Riemannian)   */
#define HTA_CB_SET_RMHMC 3  /* hta_cb_rmhmc_kernel: explicit RMHMC trajectories, D <= 16     */

typedef struct HtaCbHmcArgs {
  void* cur;               /* [C, D] current state, in / out                                                    */
  const void* init;        /* [C, D] params_init (the reference's Q2 reset, samplers.py:1018)                   */
  const void* inv_mass;    /* (D,) | (D,D) | NULL                                                               */
  const void* mass_factor; /* sqrt(mass) (D,) | chol(mass) (D,D) lower, row-major | NULL                        */
  void* samples;           /* [S, C, D] or NULL                                                                 */
  int* reject_count;       /* [C]                                                                               */
  void* H_old;             /* [C] of the launch's LAST trajectory, or NULL                                      */
  void* H_new;             /* [C] ditto                                                                         */
  unsigned char* accept;   /* [C] ditto                                                                         */
  void* gcur;              /* [C, D] workspace: gradient at the current state                                   */
  void* lp_out;            /* [C] log p at the state the launch ended in (checked against the callback), or NULL */
  long long C;
  double eps;
  unsigned long long seed, chain_offset;
  int L, n_traj, traj_offset, burn;
  int resume;              /* 1: (log p, gradient) at `cur` are in the workspace from the previous launch of this run */
  int reserved;
  void* pre;               /* NULL, or [n_traj, D + 1, C] pre-drawn records of this launch (momentum after the mass factor, log u):
                              filled by hta_cb_predraw_kernel in front of the trajectory kernel (hta_jit_hmc_predraw_bytes)       */
  long long pre_bytes;
} HtaCbHmcArgs;

typedef struct HtaCbRmhmcArgs {
  void* cur;             /* [C, D] current state, in / out                                     */
  const void* init;      /* [C, D] params_init (Q2 reset)                                      */
  void* samples;         /* [S, C, D] or NULL                                                  */
  int* reject_count;     /* [C]                                                                */
  void* H_old;           /* [C] of the launch's last trajectory, or NULL                       */
  void* H_new;
  unsigned char* accept;
  void* lp_out;          /* [C] log p at the state the launch ended in, or NULL                */
  long long C;
  double eps, alpha, jitter, omega;
  unsigned long long seed, chain_offset;
  int L, n_traj, traj_offset, burn;
} HtaCbRmhmcArgs;

typedef struct HtaCbDerivArgs {
  const void* theta; /* [C, D]                                              */
  void* logp;        /* [C] or NULL                                         */
  void* grad;        /* [C, D] or NULL                                      */
  void* neg_hess;    /* [C, D, D] or NULL: -Hessian of log p (samplers.py:108) */
  const void* M;     /* [C, D, D] (contract kernel)                         */
  void* contract;    /* [C, D] or NULL: c_i = d_i < Hess log p, M >, M held fixed */
  void* upd;         /* [C, D] or NULL: upd += coef * (grad_in + c) - the momentum update of S:395-398 fused into the contraction */
  const void* grad_in; /* [C, D]: gradient of log p at theta (with upd)      */
  double coef;
  long long C;
} HtaCbDerivArgs;

#endif
