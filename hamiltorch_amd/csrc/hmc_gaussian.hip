// Fused HMC for a dense Gaussian target: whole trajectories in one launch, state on-chip.
//
// Replaces, for log p(x) = log_norm - 0.5 (x-mu)^T P (x-mu), the per-trajectory body of
// hamiltorch.sample() (hamiltorch/samplers.py:965-1026):
//   gibbs (S:969 -> S:185-202) -> hamiltonian (S:971 -> S:779-815) -> leapfrog (S:973 -> S:281-302)
//   -> hamiltonian (S:995) -> acceptance + MH test (S:1000-1004) -> burn / store bookkeeping (S:1007-1026)
// for C independent chains.  HBM is touched only at trajectory boundaries (one coalesced [C,D]
// sample row per stored trajectory); theta, p, the gradient and both energies live in registers.
//
// Two layouts:
//   small  (D <= 8):   one chain per lane, D compile-time, P/mu in SGPRs (wave-uniform loads).
//                      This is BASELINE config 2 (3-D, 1024 chains): latency/issue bound, 16 waves.
//   general (D <= 1024): one chain per 64-lane wave, lane l owns elements l, l+64, ...;
//                      P streamed row-by-row (coalesced, L1/L2 resident), the offset vector
//                      broadcast through LDS, energies by wave butterfly reduction.
#include "common.hpp"
#include "philox.hpp"

namespace hta {

extern int g_small_chains_per_block;
extern int g_force_general;

template <typename T> struct GaussArgs {
  T* theta; const T* theta_init; const T* P; const T* mu; T log_norm;
  const T* inv_mass; const T* mass_factor;
  int64_t C; int D; int L; T eps; int n_traj; int traj_offset; int burn;
  uint64_t seed; uint64_t chain_offset;
  T* samples; int32_t* reject_count; T* H_old; T* H_new; uint8_t* accept;
  T* p_io;  // leapfrog-only entry: momentum in/out
  T* path_theta; T* path_p;  // leapfrog-only: optional per-step record [steps,C,D] (S:299-300)
};

// =============================================================================================
// small-D: thread per chain
// =============================================================================================
template <typename T, int D, int MASS> struct SmallModel {
  T P[D][D]; T mu[D]; T im[MASS == HTA_MASS_FULL ? D * D : (MASS == HTA_MASS_DIAG ? D : 1)];
  T mf[MASS == HTA_MASS_FULL ? D * D : (MASS == HTA_MASS_DIAG ? D : 1)];
  __device__ __forceinline__ void load(const GaussArgs<T>& a, bool need_mf) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
      mu[i] = a.mu[i];
#pragma unroll
      for (int j = 0; j < D; ++j) P[i][j] = a.P[i * D + j];
    }
    if (MASS == HTA_MASS_DIAG) {
#pragma unroll
      for (int i = 0; i < D; ++i) { im[i] = a.inv_mass[i]; mf[i] = need_mf ? a.mass_factor[i] : (T)0; }
    } else if (MASS == HTA_MASS_FULL) {
#pragma unroll
      for (int i = 0; i < D * D; ++i) { im[i] = a.inv_mass[i]; mf[i] = need_mf ? a.mass_factor[i] : (T)0; }
    }
  }
  // v = M^-1 p
  __device__ __forceinline__ void vel(const T (&p)[D], T (&v)[D]) const {
#pragma unroll
    for (int i = 0; i < D; ++i) {
      if (MASS == HTA_MASS_NONE) v[i] = p[i];
      else if (MASS == HTA_MASS_DIAG) v[i] = im[i] * p[i];
      else { T acc = 0;
#pragma unroll
        for (int k = 0; k < D; ++k) acc += im[i * D + k] * p[k];
        v[i] = acc; }
    }
  }
  // Pd = P (q - mu);  returns 0.5 d^T P d
  __device__ __forceinline__ T curv(const T (&q)[D], T (&Pd)[D]) const {
    T d[D];
#pragma unroll
    for (int i = 0; i < D; ++i) d[i] = q[i] - mu[i];
    T quad = 0;
#pragma unroll
    for (int i = 0; i < D; ++i) {
      T acc = P[i][0] * d[0];
#pragma unroll
      for (int k = 1; k < D; ++k) acc += P[i][k] * d[k];
      Pd[i] = acc;
      quad += d[i] * acc;
    }
    return (T)0.5 * quad;
  }
  __device__ __forceinline__ T kinetic(const T (&p)[D]) const {
    T v[D]; vel(p, v);
    T k = 0;
#pragma unroll
    for (int i = 0; i < D; ++i) k += p[i] * v[i];
    return (T)0.5 * k;
  }
  // leapfrog (S:281-302): grad = -Pd.  On exit Pd holds P(q_L - mu), returns 0.5 d^T P d at q_L.
  template <bool REC = false>
  __device__ __forceinline__ T leapfrog(T (&q)[D], T (&p)[D], T (&Pd)[D], int L, T eps, T* rec_q = nullptr,
                                        T* rec_p = nullptr, int64_t rec_stride = 0) const {
    const T he = (T)0.5 * eps;
    T quad = 0;
#pragma unroll
    for (int i = 0; i < D; ++i) p[i] -= he * Pd[i];                 // S:281
    for (int l = 0; l < L; ++l) {
      T v[D]; vel(p, v);
#pragma unroll
      for (int i = 0; i < D; ++i) q[i] = q[i] + eps * v[i];         // S:284 / S:294 / S:296
      quad = curv(q, Pd);                                            // S:297
#pragma unroll
      for (int i = 0; i < D; ++i) p[i] -= eps * Pd[i];              // S:298
      if (REC) {                                                     // S:299-300
#pragma unroll
        for (int i = 0; i < D; ++i) {
          if (rec_q) rec_q[l * rec_stride + i] = q[i];
          if (rec_p) rec_p[l * rec_stride + i] = p[i];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < D; ++i) p[i] += he * Pd[i];                 // S:302
    return quad;
  }
};

template <typename T, int D, int MASS>
__global__ void hmc_gauss_small_kernel(GaussArgs<T> a) {
  const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (c >= a.C) return;
  SmallModel<T, D, MASS> m;
  m.load(a, true);
  T th[D];
#pragma unroll
  for (int i = 0; i < D; ++i) th[i] = a.theta[c * D + i];
  const uint64_t chain = a.chain_offset + (uint64_t)c;
  int32_t rejected = 0;
  constexpr int NQ = (D + 3) / 4;

  for (int t = 0; t < a.n_traj; ++t) {
    const int n = a.traj_offset + t;
    // ---- gibbs: p ~ N(0, M)  (S:185-202)
    T z[NQ * 4];
#pragma unroll
    for (int b = 0; b < NQ; ++b) {
      T zz[4];
      normal4<T>(philox_block(a.seed, chain, (uint32_t)n, PURPOSE_MOMENTUM, 0, b), zz);
#pragma unroll
      for (int i = 0; i < 4; ++i) z[4 * b + i] = zz[i];
    }
    T p[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
      if (MASS == HTA_MASS_NONE) p[i] = z[i];
      else if (MASS == HTA_MASS_DIAG) p[i] = m.mf[i] * z[i];
      else { T acc = 0;
#pragma unroll
        for (int k = 0; k <= i; ++k) acc += m.mf[i * D + k] * z[k];
        p[i] = acc; }
    }
    // ---- H_old (S:971)
    T q[D], Pd[D];
#pragma unroll
    for (int i = 0; i < D; ++i) q[i] = th[i];
    const T quad0 = m.curv(q, Pd);
    const T h_old = -(a.log_norm - quad0) + m.kinetic(p);
    // ---- leapfrog (S:973)
    const T quad1 = m.leapfrog(q, p, Pd, a.L, a.eps);
    // ---- H_new (S:995), MH (S:1000-1004)
    const T logp1 = a.log_norm - quad1;
    const T h_new = -logp1 + m.kinetic(p);
    const T u = u23<T>(philox_block(a.seed, chain, (uint32_t)n, PURPOSE_MH, 0, 0).x);
    const bool acc = mh_accept<T>(h_old, h_new, logp1, u);
    // ---- bookkeeping (S:1007-1026; Q2 reset at n == burn+1)
    if (acc) {
#pragma unroll
      for (int i = 0; i < D; ++i) th[i] = q[i];
    } else {
      ++rejected;
      if (n == a.burn + 1) {
#pragma unroll
        for (int i = 0; i < D; ++i) th[i] = a.theta_init[c * D + i];
      }
    }
    if (a.samples && n > a.burn) {
      T* row = a.samples + ((int64_t)(n - a.burn) * a.C + c) * D;
#pragma unroll
      for (int i = 0; i < D; ++i) row[i] = th[i];
    }
    if (a.H_old) a.H_old[(int64_t)t * a.C + c] = h_old;
    if (a.H_new) a.H_new[(int64_t)t * a.C + c] = h_new;
    if (a.accept) a.accept[(int64_t)t * a.C + c] = acc ? 1 : 0;
  }
#pragma unroll
  for (int i = 0; i < D; ++i) a.theta[c * D + i] = th[i];
  if (a.reject_count) a.reject_count[c] += rejected;
}

template <typename T, int D, int MASS>
__global__ void leapfrog_gauss_small_kernel(GaussArgs<T> a) {
  const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (c >= a.C) return;
  SmallModel<T, D, MASS> m;
  m.load(a, false);
  T q[D], p[D], Pd[D];
#pragma unroll
  for (int i = 0; i < D; ++i) { q[i] = a.theta[c * D + i]; p[i] = a.p_io[c * D + i]; }
  m.curv(q, Pd);
  m.template leapfrog<true>(q, p, Pd, a.L, a.eps, a.path_theta ? a.path_theta + c * D : nullptr,
                            a.path_p ? a.path_p + c * D : nullptr, a.C * D);
#pragma unroll
  for (int i = 0; i < D; ++i) {
    a.theta[c * D + i] = q[i]; a.p_io[c * D + i] = p[i];
    if (a.path_p && a.L > 0) a.path_p[((int64_t)(a.L - 1) * a.C + c) * D + i] = p[i];  // S:302
  }
}

// =============================================================================================
// general D: one chain per wave; lane l owns elements l + 64 r, r < R
// =============================================================================================
constexpr int GEN_WAVES = 4;

template <typename T, int R, int MASS> struct WaveModel {
  const T* P; const T* mu; const T* im; const T* mf; int D;
  T* bc;  // this wave's LDS broadcast buffer [64*R]
  T muv[R], imd[R];
  __device__ __forceinline__ void init(const GaussArgs<T>& a, T* lds, int lane) {
    P = a.P; mu = a.mu; im = a.inv_mass; mf = a.mass_factor; D = a.D; bc = lds;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int j = lane + 64 * r;
      muv[r] = j < D ? a.mu[j] : (T)0;
      imd[r] = (MASS == HTA_MASS_DIAG && j < D) ? a.inv_mass[j] : (T)1;
    }
  }
  // out_j = sum_k A[j][k] x_k for the lane's elements, A symmetric or accessed as A[k][j] (coalesced)
  // `tri`: only k <= j contributes (lower-triangular factor, accessed as A[j][k]).
  __device__ __forceinline__ void matvec(const T* __restrict__ A, const T (&x)[R], T (&out)[R], int lane,
                                         bool lower_tri) const {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) bc[lane + 64 * r] = x[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) out[r] = 0;
    if (!lower_tri) {
      for (int k = 0; k < D; ++k) {
        const T xk = bc[k];
        const T* row = A + (int64_t)k * D;  // A symmetric: A[k][j] == A[j][k]
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int j = lane + 64 * r;
          if (j < D) out[r] += row[j] * xk;
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int j = lane + 64 * r;
        if (j < D) {
          T acc = 0;
          for (int k = 0; k <= j; ++k) acc += A[(int64_t)j * D + k] * bc[k];
          out[r] = acc;
        }
      }
    }
  }
  __device__ __forceinline__ void vel(const T (&p)[R], T (&v)[R], int lane) const {
    if (MASS == HTA_MASS_FULL) { matvec(im, p, v, lane, false); return; }
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = (MASS == HTA_MASS_DIAG) ? imd[r] * p[r] : p[r];
  }
  __device__ __forceinline__ T curv(const T (&q)[R], T (&Pd)[R], int lane) const {
    T d[R];
#pragma unroll
    for (int r = 0; r < R; ++r) d[r] = (lane + 64 * r < D) ? q[r] - muv[r] : (T)0;
    matvec(P, d, Pd, lane, false);
    T quad = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) quad += d[r] * Pd[r];
    return (T)0.5 * wave_sum(quad);
  }
  __device__ __forceinline__ T kinetic(const T (&p)[R], int lane) const {
    T v[R]; vel(p, v, lane);
    T k = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) k += p[r] * v[r];
    return (T)0.5 * wave_sum(k);
  }
  template <bool REC = false>
  __device__ __forceinline__ T leapfrog(T (&q)[R], T (&p)[R], T (&Pd)[R], int L, T eps, int lane,
                                        T* rec_q = nullptr, T* rec_p = nullptr, int64_t rec_stride = 0) const {
    const T he = (T)0.5 * eps;
    T quad = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) p[r] -= he * Pd[r];
    for (int l = 0; l < L; ++l) {
      T v[R]; vel(p, v, lane);
#pragma unroll
      for (int r = 0; r < R; ++r) q[r] = q[r] + eps * v[r];
      quad = curv(q, Pd, lane);
#pragma unroll
      for (int r = 0; r < R; ++r) p[r] -= eps * Pd[r];
      if (REC) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int j = lane + 64 * r;
          if (j < D && rec_q) rec_q[l * rec_stride + j] = q[r];
          if (j < D && rec_p) rec_p[l * rec_stride + j] = p[r];
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) p[r] += he * Pd[r];
    return quad;
  }
};

template <typename T, int R, int MASS>
__global__ __launch_bounds__(64 * GEN_WAVES) void hmc_gauss_wave_kernel(GaussArgs<T> a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  T* lds = reinterpret_cast<T*>(smem_raw) + wave * 64 * R;
  const int64_t cc = (int64_t)blockIdx.x * GEN_WAVES + wave;
  const bool live = cc < a.C;
  const int64_t c = live ? cc : a.C - 1;  // dead waves shadow the last chain (barriers stay matched), no stores
  WaveModel<T, R, MASS> m;
  m.init(a, lds, lane);
  const int D = a.D;
  T th[R];
#pragma unroll
  for (int r = 0; r < R; ++r) th[r] = (lane + 64 * r < D) ? a.theta[c * D + lane + 64 * r] : (T)0;
  const uint64_t chain = a.chain_offset + (uint64_t)c;
  int32_t rejected = 0;

  for (int t = 0; t < a.n_traj; ++t) {
    const int n = a.traj_offset + t;
    T z[R], p[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int j = lane + 64 * r;
      z[r] = j < D ? normal_elem<T>(a.seed, chain, (uint32_t)n, 0, j) : (T)0;
    }
    if (MASS == HTA_MASS_FULL) m.matvec(a.mass_factor, z, p, lane, true);
    else {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int j = lane + 64 * r;
        p[r] = (MASS == HTA_MASS_DIAG && j < D) ? a.mass_factor[j] * z[r] : z[r];
      }
    }
    T q[R], Pd[R];
#pragma unroll
    for (int r = 0; r < R; ++r) q[r] = th[r];
    const T quad0 = m.curv(q, Pd, lane);
    const T h_old = -(a.log_norm - quad0) + m.kinetic(p, lane);
    const T quad1 = m.leapfrog(q, p, Pd, a.L, a.eps, lane);
    const T logp1 = a.log_norm - quad1;
    const T h_new = -logp1 + m.kinetic(p, lane);
    const T u = u23<T>(philox_block(a.seed, chain, (uint32_t)n, PURPOSE_MH, 0, 0).x);
    const bool acc = mh_accept<T>(h_old, h_new, logp1, u);
    if (acc) {
#pragma unroll
      for (int r = 0; r < R; ++r) th[r] = q[r];
    } else {
      ++rejected;
      if (n == a.burn + 1) {
#pragma unroll
        for (int r = 0; r < R; ++r) if (lane + 64 * r < D) th[r] = a.theta_init[c * D + lane + 64 * r];
      }
    }
    if (live) {
      if (a.samples && n > a.burn) {
        T* row = a.samples + ((int64_t)(n - a.burn) * a.C + c) * D;
#pragma unroll
        for (int r = 0; r < R; ++r) if (lane + 64 * r < D) row[lane + 64 * r] = th[r];
      }
      if (lane == 0) {
        if (a.H_old) a.H_old[(int64_t)t * a.C + c] = h_old;
        if (a.H_new) a.H_new[(int64_t)t * a.C + c] = h_new;
        if (a.accept) a.accept[(int64_t)t * a.C + c] = acc ? 1 : 0;
      }
    }
  }
  if (live) {
#pragma unroll
    for (int r = 0; r < R; ++r) if (lane + 64 * r < D) a.theta[c * D + lane + 64 * r] = th[r];
    if (lane == 0 && a.reject_count) a.reject_count[c] += rejected;
  }
}

template <typename T, int R, int MASS>
__global__ __launch_bounds__(64 * GEN_WAVES) void leapfrog_gauss_wave_kernel(GaussArgs<T> a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  T* lds = reinterpret_cast<T*>(smem_raw) + wave * 64 * R;
  const int64_t cc = (int64_t)blockIdx.x * GEN_WAVES + wave;
  const bool live = cc < a.C;
  const int64_t c = live ? cc : a.C - 1;
  WaveModel<T, R, MASS> m;
  m.init(a, lds, lane);
  const int D = a.D;
  T q[R], p[R], Pd[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int j = lane + 64 * r;
    q[r] = j < D ? a.theta[c * D + j] : (T)0;
    p[r] = j < D ? a.p_io[c * D + j] : (T)0;
  }
  m.curv(q, Pd, lane);
  m.template leapfrog<true>(q, p, Pd, a.L, a.eps, lane, (live && a.path_theta) ? a.path_theta + c * D : nullptr,
                            (live && a.path_p) ? a.path_p + c * D : nullptr, a.C * D);
  if (live) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int j = lane + 64 * r;
      if (j < D) {
        a.theta[c * D + j] = q[r]; a.p_io[c * D + j] = p[r];
        if (a.path_p && a.L > 0) a.path_p[((int64_t)(a.L - 1) * a.C + c) * D + j] = p[r];
      }
    }
  }
}

// =============================================================================================
// dispatch
// =============================================================================================
template <typename T, int D, int MASS> void launch_small(const GaussArgs<T>& a, bool lf_only, hipStream_t s) {
  int block = g_small_chains_per_block > 0 ? g_small_chains_per_block : 64;
  const int grid = (int)((a.C + block - 1) / block);
  if (lf_only) leapfrog_gauss_small_kernel<T, D, MASS><<<grid, block, 0, s>>>(a);
  else hmc_gauss_small_kernel<T, D, MASS><<<grid, block, 0, s>>>(a);
}
template <typename T, int D> void launch_small_m(const GaussArgs<T>& a, int kind, bool lf, hipStream_t s) {
  if (kind == HTA_MASS_NONE) launch_small<T, D, HTA_MASS_NONE>(a, lf, s);
  else if (kind == HTA_MASS_DIAG) launch_small<T, D, HTA_MASS_DIAG>(a, lf, s);
  else launch_small<T, D, HTA_MASS_FULL>(a, lf, s);
}
template <typename T, int R, int MASS> void launch_wave(const GaussArgs<T>& a, bool lf_only, hipStream_t s) {
  const int grid = (int)((a.C + GEN_WAVES - 1) / GEN_WAVES);
  const size_t lds = (size_t)GEN_WAVES * 64 * R * sizeof(T);
  if (lf_only) leapfrog_gauss_wave_kernel<T, R, MASS><<<grid, 64 * GEN_WAVES, lds, s>>>(a);
  else hmc_gauss_wave_kernel<T, R, MASS><<<grid, 64 * GEN_WAVES, lds, s>>>(a);
}
template <typename T, int R> void launch_wave_m(const GaussArgs<T>& a, int kind, bool lf, hipStream_t s) {
  if (kind == HTA_MASS_NONE) launch_wave<T, R, HTA_MASS_NONE>(a, lf, s);
  else if (kind == HTA_MASS_DIAG) launch_wave<T, R, HTA_MASS_DIAG>(a, lf, s);
  else launch_wave<T, R, HTA_MASS_FULL>(a, lf, s);
}

template <typename T> int gaussian_dispatch(const GaussArgs<T>& a, int kind, bool lf_only, hipStream_t s) {
  const char* who = lf_only ? "hta_hmc_gaussian_leapfrog" : "hta_hmc_gaussian_sample";
  HTA_REQUIRE(a.theta && a.P && a.mu, "%s: NULL state/model pointer", who);
  HTA_REQUIRE(a.C > 0 && a.D > 0 && a.D <= 1024, "%s: need C > 0 and 1 <= D <= 1024 (C=%lld D=%d)", who,
              (long long)a.C, a.D);
  HTA_REQUIRE(a.L >= 0, "%s: negative step count", who);
  HTA_REQUIRE(kind >= HTA_MASS_NONE && kind <= HTA_MASS_FULL, "%s: unknown mass kind %d", who, kind);
  HTA_REQUIRE(kind == HTA_MASS_NONE || (a.inv_mass && (lf_only || a.mass_factor)), "%s: mass operand is NULL", who);
  if (!lf_only) HTA_REQUIRE(a.theta_init && a.n_traj >= 0, "%s: bad trajectory arguments", who);
  if (lf_only) HTA_REQUIRE(a.p_io, "%s: momentum is NULL", who);
  if (a.n_traj == 0 && !lf_only) return HTA_OK;
  const int D = a.D;
  // f64 models past D=4 overflow the SGPR file (P alone is 2*D*D SGPRs) and would spill to scratch
  const int small_max = sizeof(T) == 8 ? 4 : 8;
  if (D <= small_max && !g_force_general) {
    switch (D) {
      case 1: launch_small_m<T, 1>(a, kind, lf_only, s); break;
      case 2: launch_small_m<T, 2>(a, kind, lf_only, s); break;
      case 3: launch_small_m<T, 3>(a, kind, lf_only, s); break;
      case 4: launch_small_m<T, 4>(a, kind, lf_only, s); break;
      case 5: launch_small_m<T, 5>(a, kind, lf_only, s); break;
      case 6: launch_small_m<T, 6>(a, kind, lf_only, s); break;
      case 7: launch_small_m<T, 7>(a, kind, lf_only, s); break;
      default: launch_small_m<T, 8>(a, kind, lf_only, s); break;
    }
  } else {
    const int R = (D + 63) / 64;
    if (R <= 1) launch_wave_m<T, 1>(a, kind, lf_only, s);
    else if (R <= 2) launch_wave_m<T, 2>(a, kind, lf_only, s);
    else if (R <= 4) launch_wave_m<T, 4>(a, kind, lf_only, s);
    else if (R <= 8) launch_wave_m<T, 8>(a, kind, lf_only, s);
    else launch_wave_m<T, 16>(a, kind, lf_only, s);
  }
  HTA_CHECK_LAUNCH(who);
  return HTA_OK;
}

}  // namespace hta

extern "C" {

#define HTA_DEFINE_GAUSS(SUF, T)                                                                               \
  int hta_hmc_gaussian_sample_##SUF(T* theta, const T* theta_init, const T* P, const T* mu, T log_norm,         \
                                    int mass_kind, const T* inv_mass, const T* mass_factor, int64_t C, int D,   \
                                    int L, T eps, int n_traj, int traj_offset, int burn, uint64_t seed,         \
                                    uint64_t chain_offset, T* samples, int32_t* reject_count, T* H_old,         \
                                    T* H_new, uint8_t* accept, void* stream) {                                  \
    hta::GaussArgs<T> a{theta, theta_init, P, mu, log_norm, inv_mass, mass_factor, C, D, L, eps, n_traj,         \
                        traj_offset, burn, seed, chain_offset, samples, reject_count, H_old, H_new, accept,     \
                        nullptr, nullptr, nullptr};                                                                               \
    return hta::gaussian_dispatch<T>(a, mass_kind, false, (hipStream_t)stream);                                 \
  }                                                                                                             \
  int hta_hmc_gaussian_leapfrog_##SUF(T* theta, T* p, const T* P, const T* mu, int mass_kind, const T* inv_mass, \
                                      int64_t C, int D, int steps, T eps, T* path_theta, T* path_p,             \
                                      void* stream) {                                                           \
    hta::GaussArgs<T> a{theta, nullptr, P, mu, (T)0, inv_mass, nullptr, C, D, steps, eps, 0, 0, 0, 0, 0,         \
                        nullptr, nullptr, nullptr, nullptr, nullptr, p, path_theta, path_p};                                        \
    return hta::gaussian_dispatch<T>(a, mass_kind, true, (hipStream_t)stream);                                  \
  }

HTA_DEFINE_GAUSS(f32, float)
HTA_DEFINE_GAUSS(f64, double)

}  // extern "C"
