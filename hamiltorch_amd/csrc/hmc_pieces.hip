// State-update kernels around a user log_prob_func evaluated by torch (generic-callback path).
// Each mirrors one statement group of hamiltorch/samplers.py (cited per kernel), batched over
// chains: theta[C,D], p[C,D] chain-major.  All are single-pass, HBM-bound, coalesced.
#include "common.hpp"
#include "philox.hpp"
#include "rmhmc.hpp"

namespace hta {

// ---- gibbs(): p ~ N(0,M)  (samplers.py:185-202) ---------------------------------------------
// NONE/DIAG: one thread per Philox block of 4 elements.
template <typename T, int MASS>
__global__ void resample_kernel(T* __restrict__ p, const T* __restrict__ mf, int64_t C, int D, int nq,
                                uint64_t seed, uint64_t chain_offset, uint32_t draw, const int32_t* __restrict__ n_dev) {
  if (n_dev) draw = (uint32_t)*n_dev;      // trajectory index kept on the device: the launch can sit in a replayed HIP graph
  const int64_t total = C * nq;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = t / nq;
    const int q = (int)(t - c * nq);
    T z[4];
    normal4<T>(philox_block(seed, chain_offset + c, draw, PURPOSE_MOMENTUM, 0, q), z);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = 4 * q + i;
      if (j < D) p[c * D + j] = (MASS == HTA_MASS_DIAG) ? mf[j] * z[i] : z[i];
    }
  }
}

// FULL: p = Lm z, one block per chain, z staged in LDS (Lm lower-triangular, row-major).
template <typename T>
__global__ void resample_full_kernel(T* __restrict__ p, const T* __restrict__ Lm, int64_t C, int D,
                                     uint64_t seed, uint64_t chain_offset, uint32_t draw, const int32_t* __restrict__ n_dev) {
  if (n_dev) draw = (uint32_t)*n_dev;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* z = reinterpret_cast<T*>(smem_raw);
  for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
    const int nq = (D + 3) / 4;
    for (int q = threadIdx.x; q < nq; q += blockDim.x) {
      T zz[4];
      normal4<T>(philox_block(seed, chain_offset + c, draw, PURPOSE_MOMENTUM, 0, q), zz);
#pragma unroll
      for (int i = 0; i < 4; ++i) z[4 * q + i] = zz[i];
    }
    __syncthreads();
    for (int j = threadIdx.x; j < D; j += blockDim.x) {
      T acc = 0;
      for (int k = 0; k <= j; ++k) acc += Lm[(int64_t)j * D + k] * z[k];
      p[c * D + j] = acc;
    }
    __syncthreads();
  }
}

// ---- leapfrog kick / drift (samplers.py:281, 283-298, 302; split 505-520) ---------------------
template <typename T, int MASS>
__global__ void kick_drift_kernel(T* __restrict__ theta, T* __restrict__ p, const T* __restrict__ grad, T kick,
                                  T drift, const T* __restrict__ im, int64_t C, int D) {
  const int64_t total = C * D;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    T pv = p[t];
    if (grad) { pv += kick * grad[t]; p[t] = pv; }
    if (drift != (T)0) {
      const int j = (int)(t % D);
      const T v = (MASS == HTA_MASS_DIAG) ? im[j] * pv : pv;
      theta[t] = theta[t] + drift * v;
    }
  }
}

template <typename T>
__global__ void drift_full_kernel(T* __restrict__ theta, const T* __restrict__ p, T drift,
                                  const T* __restrict__ im, int64_t C, int D) {
  const int64_t total = C * D;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = t / D;
    const int j = (int)(t - c * D);
    T acc = 0;
    for (int k = 0; k < D; ++k) acc += im[(int64_t)j * D + k] * p[c * D + k];
    theta[t] = theta[t] + drift * acc;
  }
}

// ---- hamiltonian(): H = -logp + 0.5 p^T M^-1 p  (samplers.py:799-815) --------------------------
// G lanes cooperate on one chain (G = pow2 >= min(D,64)); butterfly reduction inside the group.
template <typename T, int MASS, int G>
__global__ void hamiltonian_kernel(const T* __restrict__ p, const T* __restrict__ logp,
                                   const T* __restrict__ im, T* __restrict__ H, int64_t C, int D) {
  const int64_t gid = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / G;
  const int lane = threadIdx.x % G;
  const bool live = gid < C;
  const int64_t c = live ? gid : 0;
  T acc = 0;
  for (int j = lane; j < D; j += G) {
    const T pj = p[c * D + j];
    T v;
    if (MASS == HTA_MASS_NONE) v = pj;
    else if (MASS == HTA_MASS_DIAG) v = im[j] * pj;
    else { v = 0; for (int k = 0; k < D; ++k) v += im[(int64_t)j * D + k] * p[c * D + k]; }
    acc += pj * v;
  }
  acc = group_sum<G>(acc);
  if (live && lane == 0) H[c] = (logp ? -logp[c] : (T)0) + (T)0.5 * acc;
}

// ---- MH test + burn bookkeeping (samplers.py:1000-1026, 1045-1057) ------------------------------
template <typename T, int G>
__global__ void mh_select_kernel(T* __restrict__ cur, const T* __restrict__ prop, const T* __restrict__ init,
                                 const T* __restrict__ H_old, const T* __restrict__ H_new,
                                 const T* __restrict__ logp_new, T* __restrict__ row,
                                 int32_t* __restrict__ reject_count, uint8_t* __restrict__ out_accept, int64_t C,
                                 int D, int n, int burn, uint64_t seed, uint64_t chain_offset,
                                 const int32_t* __restrict__ n_dev) {
  if (n_dev) {                 // device-side trajectory index: `row` is then the BASE of the [S, C, D] sample buffer
    n = *n_dev;
    if (row && n > burn) row += (int64_t)(n - burn) * C * D;
  }
  const int64_t c = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / G;
  const int lane = threadIdx.x % G;
  if (c >= C) return;
  const T u = u23<T>(philox_block(seed, chain_offset + c, (uint32_t)n, PURPOSE_MH, 0, 0).x);
  const bool acc = mh_accept<T>(H_old[c], H_new[c], logp_new ? logp_new[c] : (T)0, u);
  const bool reset = (!acc) && (n == burn + 1);  // SURVEY Q2 (samplers.py:1018)
  for (int j = lane; j < D; j += G) {
    T v = acc ? prop[c * D + j] : (reset ? init[c * D + j] : cur[c * D + j]);
    cur[c * D + j] = v;
    if (row && n > burn) row[c * D + j] = v;
  }
  if (lane == 0) {
    if (!acc) reject_count[c] += 1;
    if (out_accept) out_accept[c] = acc ? 1 : 0;
  }
}

static inline int grid_for(int64_t total, int block) {
  int64_t g = (total + block - 1) / block;
  if (g > 256 * 8) g = 256 * 8;  // 256 CUs x 8 blocks, grid-stride the rest
  if (g < 1) g = 1;
  return (int)g;
}
static inline int pow2_group(int D) { int g = 1; while (g < D && g < 64) g <<= 1; return g; }

__global__ void counter_add_kernel(int32_t* counter, int delta) { *counter += delta; }

template <typename T>
int momentum_resample(T* p, int kind, const T* mf, int64_t C, int D, uint64_t seed, uint64_t off, uint32_t draw,
                      hipStream_t s, const int32_t* n_dev = nullptr) {
  HTA_REQUIRE(p && C > 0 && D > 0, "hta_momentum_resample: bad shape C=%lld D=%d", (long long)C, D);
  HTA_REQUIRE(kind == HTA_MASS_NONE || mf, "hta_momentum_resample: mass_factor is NULL");
  const int nq = (D + 3) / 4;
  if (kind == HTA_MASS_NONE)
    resample_kernel<T, HTA_MASS_NONE><<<grid_for(C * nq, 256), 256, 0, s>>>(p, mf, C, D, nq, seed, off, draw, n_dev);
  else if (kind == HTA_MASS_DIAG)
    resample_kernel<T, HTA_MASS_DIAG><<<grid_for(C * nq, 256), 256, 0, s>>>(p, mf, C, D, nq, seed, off, draw, n_dev);
  else if (kind == HTA_MASS_FULL) {
    const int block = D <= 64 ? 64 : (D <= 128 ? 128 : 256);
    const size_t lds = (size_t)nq * 4 * sizeof(T);
    HTA_REQUIRE(lds <= 64 * 1024, "hta_momentum_resample: D=%d too large for the full-mass path", D);
    resample_full_kernel<T><<<(int)(C < 4096 ? C : 4096), block, lds, s>>>(p, mf, C, D, seed, off, draw, n_dev);
  } else HTA_REQUIRE(false, "hta_momentum_resample: unknown mass kind %d", kind);
  HTA_CHECK_LAUNCH("hta_momentum_resample");
  return HTA_OK;
}

__global__ void run_begin_kernel(const uint32_t* __restrict__ init, uint32_t* __restrict__ cur, uint32_t* __restrict__ row0,
                                 int32_t* __restrict__ rej, int64_t words, int64_t C) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < words; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t v = init[i];
    cur[i] = v;
    if (row0) row0[i] = v;
    if (rej && i < C) rej[i] = 0;
  }
}

template <typename T>
int kick_drift(T* theta, T* p, const T* grad, T kick, T drift, int kind, const T* im, int64_t C, int D,
               hipStream_t s) {
  HTA_REQUIRE(theta && p && C > 0 && D > 0, "hta_kick_drift: bad arguments");
  HTA_REQUIRE(kind == HTA_MASS_NONE || im || drift == (T)0, "hta_kick_drift: inv_mass is NULL");
  const int g = grid_for(C * D, 256);
  if (kind == HTA_MASS_NONE) kick_drift_kernel<T, HTA_MASS_NONE><<<g, 256, 0, s>>>(theta, p, grad, kick, drift, im, C, D);
  else if (kind == HTA_MASS_DIAG) kick_drift_kernel<T, HTA_MASS_DIAG><<<g, 256, 0, s>>>(theta, p, grad, kick, drift, im, C, D);
  else if (kind == HTA_MASS_FULL) {
    if (grad) kick_drift_kernel<T, HTA_MASS_NONE><<<g, 256, 0, s>>>(theta, p, grad, kick, (T)0, im, C, D);
    if (drift != (T)0) drift_full_kernel<T><<<g, 256, 0, s>>>(theta, p, drift, im, C, D);
  } else HTA_REQUIRE(false, "hta_kick_drift: unknown mass kind %d", kind);
  HTA_CHECK_LAUNCH("hta_kick_drift");
  return HTA_OK;
}

template <typename T, int MASS>
void launch_ham(const T* p, const T* logp, const T* im, T* H, int64_t C, int D, hipStream_t s) {
  const int G = pow2_group(D);
  const int block = 256;
  const int64_t total = C * G;
  const int grid = (int)((total + block - 1) / block);
  switch (G) {
    case 1: hamiltonian_kernel<T, MASS, 1><<<grid, block, 0, s>>>(p, logp, im, H, C, D); break;
    case 2: hamiltonian_kernel<T, MASS, 2><<<grid, block, 0, s>>>(p, logp, im, H, C, D); break;
    case 4: hamiltonian_kernel<T, MASS, 4><<<grid, block, 0, s>>>(p, logp, im, H, C, D); break;
    case 8: hamiltonian_kernel<T, MASS, 8><<<grid, block, 0, s>>>(p, logp, im, H, C, D); break;
    case 16: hamiltonian_kernel<T, MASS, 16><<<grid, block, 0, s>>>(p, logp, im, H, C, D); break;
    case 32: hamiltonian_kernel<T, MASS, 32><<<grid, block, 0, s>>>(p, logp, im, H, C, D); break;
    default: hamiltonian_kernel<T, MASS, 64><<<grid, block, 0, s>>>(p, logp, im, H, C, D); break;
  }
}

template <typename T>
int hamiltonian(const T* p, const T* logp, int kind, const T* im, T* H, int64_t C, int D, hipStream_t s) {
  HTA_REQUIRE(p && H && C > 0 && D > 0, "hta_hamiltonian: bad arguments");
  HTA_REQUIRE(kind == HTA_MASS_NONE || im, "hta_hamiltonian: inv_mass is NULL");
  if (kind == HTA_MASS_NONE) launch_ham<T, HTA_MASS_NONE>(p, logp, im, H, C, D, s);
  else if (kind == HTA_MASS_DIAG) launch_ham<T, HTA_MASS_DIAG>(p, logp, im, H, C, D, s);
  else if (kind == HTA_MASS_FULL) launch_ham<T, HTA_MASS_FULL>(p, logp, im, H, C, D, s);
  else HTA_REQUIRE(false, "hta_hamiltonian: unknown mass kind %d", kind);
  HTA_CHECK_LAUNCH("hta_hamiltonian");
  return HTA_OK;
}

template <typename T>
int mh_select_impl(T* cur, const T* prop, const T* init, const T* Ho, const T* Hn, const T* lpn, T* row, int32_t* rej,
                   uint8_t* acc, int64_t C, int D, int n, int burn, uint64_t seed, uint64_t off, hipStream_t s,
                   const int32_t* n_dev) {
  HTA_REQUIRE(cur && prop && init && Ho && Hn && rej && C > 0 && D > 0, "hta_mh_select: bad arguments");
  const int G = pow2_group(D);
  const int block = 256;
  const int grid = (int)((C * G + block - 1) / block);
#define HTA_MH(GG) mh_select_kernel<T, GG><<<grid, block, 0, s>>>(cur, prop, init, Ho, Hn, lpn, row, rej, acc, C, D, n, burn, seed, off, n_dev)
  switch (G) {
    case 1: HTA_MH(1); break; case 2: HTA_MH(2); break; case 4: HTA_MH(4); break; case 8: HTA_MH(8); break;
    case 16: HTA_MH(16); break; case 32: HTA_MH(32); break; default: HTA_MH(64); break;
  }
#undef HTA_MH
  HTA_CHECK_LAUNCH("hta_mh_select");
  return HTA_OK;
}

template <typename T>
int mh_select(T* cur, const T* prop, const T* init, const T* Ho, const T* Hn, const T* lpn, T* row, int32_t* rej,
              uint8_t* acc, int64_t C, int D, int n, int burn, uint64_t seed, uint64_t off, hipStream_t s) {
  return mh_select_impl<T>(cur, prop, init, Ho, Hn, lpn, row, rej, acc, C, D, n, burn, seed, off, s, nullptr);
}

template int mh_select<float>(float*, const float*, const float*, const float*, const float*, const float*, float*,
                              int32_t*, uint8_t*, int64_t, int, int, int, uint64_t, uint64_t, hipStream_t);
template int mh_select<double>(double*, const double*, const double*, const double*, const double*, const double*,
                               double*, int32_t*, uint8_t*, int64_t, int, int, int, uint64_t, uint64_t, hipStream_t);

}  // namespace hta

extern "C" {
int hta_momentum_resample_f32(float* p, int k, const float* mf, int64_t C, int D, uint64_t seed, uint64_t off,
                              uint32_t draw, void* s) {
  return hta::momentum_resample<float>(p, k, mf, C, D, seed, off, draw, (hipStream_t)s);
}
int hta_momentum_resample_f64(double* p, int k, const double* mf, int64_t C, int D, uint64_t seed, uint64_t off,
                              uint32_t draw, void* s) {
  return hta::momentum_resample<double>(p, k, mf, C, D, seed, off, draw, (hipStream_t)s);
}
int hta_kick_drift_f32(float* th, float* p, const float* g, float kick, float drift, int k, const float* im,
                       int64_t C, int D, void* s) {
  return hta::kick_drift<float>(th, p, g, kick, drift, k, im, C, D, (hipStream_t)s);
}
int hta_kick_drift_f64(double* th, double* p, const double* g, double kick, double drift, int k, const double* im,
                       int64_t C, int D, void* s) {
  return hta::kick_drift<double>(th, p, g, kick, drift, k, im, C, D, (hipStream_t)s);
}
int hta_hamiltonian_f32(const float* p, const float* lp, int k, const float* im, float* H, int64_t C, int D,
                        void* s) {
  return hta::hamiltonian<float>(p, lp, k, im, H, C, D, (hipStream_t)s);
}
int hta_hamiltonian_f64(const double* p, const double* lp, int k, const double* im, double* H, int64_t C, int D,
                        void* s) {
  return hta::hamiltonian<double>(p, lp, k, im, H, C, D, (hipStream_t)s);
}
int hta_mh_select_f32(float* cur, const float* prop, const float* init, const float* Ho, const float* Hn,
                      const float* lpn, float* row, int32_t* rej, uint8_t* acc, int64_t C, int D, int n, int burn,
                      uint64_t seed, uint64_t off, void* s) {
  return hta::mh_select<float>(cur, prop, init, Ho, Hn, lpn, row, rej, acc, C, D, n, burn, seed, off, (hipStream_t)s);
}
int hta_mh_select_f64(double* cur, const double* prop, const double* init, const double* Ho, const double* Hn,
                      const double* lpn, double* row, int32_t* rej, uint8_t* acc, int64_t C, int D, int n, int burn,
                      uint64_t seed, uint64_t off, void* s) {
  return hta::mh_select<double>(cur, prop, init, Ho, Hn, lpn, row, rej, acc, C, D, n, burn, seed, off, (hipStream_t)s);
}

/* trajectory index on the device: the same two operations for launches captured in a HIP graph and replayed once per
 * trajectory (the generic-callback engine); hta_counter_add advances the index inside the graph. */
int hta_momentum_resample_at_f32(float* p, int k, const float* mf, int64_t C, int D, uint64_t seed, uint64_t off,
                                 const int32_t* n_dev, void* s) {
  if (!n_dev) { hta::set_error("hta_momentum_resample_at: n_dev is NULL"); return HTA_ERR_INVALID; }
  return hta::momentum_resample<float>(p, k, mf, C, D, seed, off, 0, (hipStream_t)s, n_dev);
}
int hta_momentum_resample_at_f64(double* p, int k, const double* mf, int64_t C, int D, uint64_t seed, uint64_t off,
                                 const int32_t* n_dev, void* s) {
  if (!n_dev) { hta::set_error("hta_momentum_resample_at: n_dev is NULL"); return HTA_ERR_INVALID; }
  return hta::momentum_resample<double>(p, k, mf, C, D, seed, off, 0, (hipStream_t)s, n_dev);
}
int hta_mh_select_at_f32(float* cur, const float* prop, const float* init, const float* Ho, const float* Hn,
                         const float* lpn, float* samples_base, int32_t* rej, uint8_t* acc, int64_t C, int D,
                         const int32_t* n_dev, int burn, uint64_t seed, uint64_t off, void* s) {
  if (!n_dev) { hta::set_error("hta_mh_select_at: n_dev is NULL"); return HTA_ERR_INVALID; }
  return hta::mh_select_impl<float>(cur, prop, init, Ho, Hn, lpn, samples_base, rej, acc, C, D, 0, burn, seed, off, (hipStream_t)s, n_dev);
}
int hta_mh_select_at_f64(double* cur, const double* prop, const double* init, const double* Ho, const double* Hn,
                         const double* lpn, double* samples_base, int32_t* rej, uint8_t* acc, int64_t C, int D,
                         const int32_t* n_dev, int burn, uint64_t seed, uint64_t off, void* s) {
  if (!n_dev) { hta::set_error("hta_mh_select_at: n_dev is NULL"); return HTA_ERR_INVALID; }
  return hta::mh_select_impl<double>(cur, prop, init, Ho, Hn, lpn, samples_base, rej, acc, C, D, 0, burn, seed, off, (hipStream_t)s, n_dev);
}
/* what every sample() run starts with (S:954-961: params = params_init.clone(), ret_params = [params.clone()], num_rejected = 0)
 * as ONE launch: cur <- init, row0 <- init, reject_count <- 0.  elem_size 4 or 8; total = C * D elements. */
int hta_run_begin(const void* init, void* cur, void* row0, int32_t* reject_count, int64_t C, int D, int elem_size, void* s) {
  if (!init || !cur || C <= 0 || D <= 0 || (elem_size != 4 && elem_size != 8)) {
    hta::set_error("hta_run_begin: bad arguments");
    return HTA_ERR_INVALID;
  }
  const int64_t words = C * D * (elem_size / 4);
  int grid = (int)((words + 255) / 256); if (grid > 4096) grid = 4096;
  hta::run_begin_kernel<<<grid, 256, 0, (hipStream_t)s>>>((const uint32_t*)init, (uint32_t*)cur, (uint32_t*)row0, reject_count, words, C);
  HTA_CHECK_LAUNCH("hta_run_begin");
  return HTA_OK;
}
int hta_counter_add(int32_t* counter, int delta, void* s) {
  if (!counter) { hta::set_error("hta_counter_add: NULL counter"); return HTA_ERR_INVALID; }
  hta::counter_add_kernel<<<1, 1, 0, (hipStream_t)s>>>(counter, delta);
  HTA_CHECK_LAUNCH("hta_counter_add");
  return HTA_OK;
}
}
