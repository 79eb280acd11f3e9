// Batched Riemannian-metric evaluation: one workgroup per (chain, evaluation point), matrices in LDS.
//
// Replaces, per system b, the chain of torch calls the reference makes in
//   fisher()            (samplers.py:108-122)  Hs = -hessian + jitter*U(0,1) on the diagonal; SOFTABS:
//                                              eigh(lower) -> lam/tanh(alpha lam) -> Q diag Q^T
//   cholesky_inverse()  (samplers.py:146-148)  G^-1 p
//   rm_hamiltonian()    (samplers.py:710-731)  -logp + D/2 log 2pi + 1/2 log|G| + 1/2 p^T G^-1 p
//   gibbs(RMHMC)        (samplers.py:183-184)  p = chol(G) z
// and, for the explicit integrator (samplers.py:429-458), fuses the two first-order updates of a
// half step (x-update by G^-1 m, momentum kick by P (X - mu)) into the same launch.
//
// Eigen-decomposition: cyclic two-sided Jacobi with round-robin (chess-tournament) pair ordering,
// all n/2 disjoint rotations of a round applied together.  Each 2x2 pair-block of A = J^T A J is
// updated by one thread (left and right rotation in registers), only blocks on or above the block
// diagonal are computed and mirrored, V = V J by one thread per (row, pair).  Unconditionally
// convergent and orthogonal to rounding for any symmetric input (degenerate spectra included),
// which is what G = Q f(L) Q^T needs.  In SOLVE mode G^-1 m is formed as Q (Q^T m / lam~): the
// assembled G and its Cholesky factor are only built when the caller asks for them (fisher(),
// the momentum draw); both agree to O(cond * eps) with the reference's Cholesky solve.
#include "common.hpp"
#include "philox.hpp"
#include "rmhmc.hpp"

namespace hta {

constexpr int MT = 1024;  // threads per system (16 waves: the LDS-latency-bound Jacobi rounds need >= 4 waves per SIMD)

template <typename T> struct Eps;
template <> struct Eps<float> { static constexpr float v = 1.1920929e-07f; };
template <> struct Eps<double> { static constexpr double v = 2.220446049250313e-16; };

template <typename T> __device__ __forceinline__ T block_sum(T v, T* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  T tot = 0;
#pragma unroll
  for (int i = 0; i < MT / 64; ++i) tot += red[i];
  return tot;
}

// ------------------------------------------------------------------------------------------------
// In-LDS Cholesky (right-looking), A[n][lda] lower triangle in/out.  Non-PD input -> NaN factor.
template <typename T> __device__ void lds_cholesky(T* A, int n, int lda) {
  const int tid = threadIdx.x;
  for (int j = 0; j < n; ++j) {
    __syncthreads();
    const T djj = sqrt(A[j * lda + j]);
    const T inv = (T)1 / djj;
    __syncthreads();
    for (int i = j + tid; i < n; i += MT) A[i * lda + j] = (i == j) ? djj : A[i * lda + j] * inv;
    __syncthreads();
    // trailing update: A[i][k] -= L[i][j] L[k][j], j < k <= i
    const int r = n - j - 1;
    for (int e = tid; e < r * r; e += MT) {
      const int ii = e / r, kk = e - ii * r;
      if (kk <= ii) {
        const int i = j + 1 + ii, k = j + 1 + kk;
        A[i * lda + k] -= A[i * lda + j] * A[k * lda + j];
      }
    }
  }
  __syncthreads();
}

// x = (L L^T)^-1 b ; b in/out vector in LDS (length n).  Column-oriented substitution.
template <typename T> __device__ void lds_chol_solve(const T* L, int n, int lda, T* v) {
  const int tid = threadIdx.x;
  for (int j = 0; j < n; ++j) {          // forward: L y = b
    __syncthreads();
    const T yj = v[j] / L[j * lda + j];
    __syncthreads();
    if (tid == 0) v[j] = yj;
    for (int i = j + 1 + tid; i < n; i += MT) v[i] -= L[i * lda + j] * yj;
  }
  for (int j = n - 1; j >= 0; --j) {     // backward: L^T x = y
    __syncthreads();
    const T xj = v[j] / L[j * lda + j];
    __syncthreads();
    if (tid == 0) v[j] = xj;
    for (int i = tid; i < j; i += MT) v[i] -= L[j * lda + i] * xj;
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// Round-robin pairing: n even, round r in [0, n-1), slot k in [0, n/2).
__device__ __forceinline__ void rr_pair(int n, int r, int k, int& p, int& q) {
  const int m = n - 1;
  if (k == 0) { p = m; q = r; }
  else { p = (r + k) % m; q = (r - k + m) % m; }
  if (p > q) { const int t = p; p = q; q = t; }
}

// Cyclic Jacobi on A[ne][lda] (symmetric, both triangles valid), V[D][ldv] <- eigenvectors (columns).
// Work split (MT threads): the <= ceil(nblk / MT) pair-blocks a thread owns are decoded ONCE (the
// enumeration does not depend on the round); V rows are walked with pair index = lane, so the hot
// loops contain no integer division.
template <typename T>
__device__ void lds_jacobi(T* A, T* V, int D, int ne, int lda, int ldv, T* cs, int* pq, T* red, int max_sweeps) {
  const int tid = threadIdx.x;
  const int NP = ne / 2;
  const int nblk = NP * (NP + 1) / 2;
  constexpr int MAXB = 4;                       // nblk <= 71*70/2 = 2485 < 4 * 1024
  int blkA[MAXB], blkB[MAXB];
#pragma unroll
  for (int k = 0; k < MAXB; ++k) {
    const int e = tid + k * MT;
    blkA[k] = -1; blkB[k] = 0;
    if (e < nblk) {
      // row-major upper-triangular enumeration: row a holds NP - a blocks
      int a = (int)(((float)(2 * NP + 1) - sqrtf((float)((2 * NP + 1) * (2 * NP + 1) - 8 * e))) * 0.5f);
      if (a < 0) a = 0;
      while (a > 0 && a * NP - a * (a - 1) / 2 > e) --a;
      while ((a + 1) * NP - (a + 1) * a / 2 <= e) ++a;
      blkA[k] = a; blkB[k] = a + (e - (a * NP - a * (a - 1) / 2));
    }
  }
  int npw = 1; while (npw < NP) npw <<= 1;      // pairs padded to a power of two <= 128
  const int vb = tid & (npw - 1), vi0 = tid / npw, vstep = MT / npw;
  T off_prev = (T)-1;
  for (int sweep = 0; sweep < max_sweeps; ++sweep) {
    // convergence: off-diagonal vs diagonal mass
    T off = 0, dg = 0;
    for (int e = tid; e < D * D; e += MT) {
      const int i = e / D, j = e - i * D;
      const T a = A[i * lda + j];
      if (i == j) dg += a * a; else off += a * a;
    }
    off = block_sum(off, red);
    dg = block_sum(dg, red);
    const T tol = (T)64 * Eps<T>::v * Eps<T>::v;
    if (!(off > tol * (dg + off))) break;                              // converged (or NaN input)
    if (sweep >= 4 && off_prev >= (T)0 && off > (T)0.25 * off_prev && off <= (T)1e-6 * (dg + off)) break;  // rounding floor
    off_prev = off;

    for (int r = 0; r < ne - 1; ++r) {
      __syncthreads();
      if (tid < NP) {
        int p, q;
        rr_pair(ne, r, tid, p, q);
        const T app = A[p * lda + p], aqq = A[q * lda + q], apq = A[p * lda + q];
        T c = 1, s = 0;
        if (apq != (T)0) {
          const T theta = (aqq - app) / ((T)2 * apq);
          const T t = copysign((T)1, theta) / (fabs(theta) + sqrt((T)1 + theta * theta));
          c = (T)1 / sqrt((T)1 + t * t);
          s = t * c;
        }
        cs[2 * tid] = c; cs[2 * tid + 1] = s;
        pq[2 * tid] = p; pq[2 * tid + 1] = q;
      }
      __syncthreads();
      // A <- J^T A J, one thread per pair-block (a <= b), mirrored
#pragma unroll
      for (int k = 0; k < MAXB; ++k) {
        const int a = blkA[k], bb = blkB[k];
        if (a >= 0) {
          const int pa = pq[2 * a], qa = pq[2 * a + 1], pb = pq[2 * bb], qb = pq[2 * bb + 1];
          const T ca = cs[2 * a], sa = cs[2 * a + 1], cb = cs[2 * bb], sb = cs[2 * bb + 1];
          const T m00 = A[pa * lda + pb], m01 = A[pa * lda + qb], m10 = A[qa * lda + pb], m11 = A[qa * lda + qb];
          const T t00 = cb * m00 - sb * m01, t01 = sb * m00 + cb * m01;
          const T t10 = cb * m10 - sb * m11, t11 = sb * m10 + cb * m11;
          T n00 = ca * t00 - sa * t10, n01 = ca * t01 - sa * t11;
          T n10 = sa * t00 + ca * t10, n11 = sa * t01 + ca * t11;
          if (a == bb) { n01 = 0; n10 = 0; }
          A[pa * lda + pb] = n00; A[pa * lda + qb] = n01; A[qa * lda + pb] = n10; A[qa * lda + qb] = n11;
          if (a != bb) { A[pb * lda + pa] = n00; A[qb * lda + pa] = n01; A[pb * lda + qa] = n10; A[qb * lda + qa] = n11; }
        }
      }
      // V <- V J : pair = lane, rows strided
      if (vb < NP) {
        const int pb = pq[2 * vb], qb = pq[2 * vb + 1];
        if (qb < D) {     // the padding index (odd D) never rotates
          const T c = cs[2 * vb], s = cs[2 * vb + 1];
          for (int i = vi0; i < D; i += vstep) {
            const T vp = V[i * ldv + pb], vq = V[i * ldv + qb];
            V[i * ldv + pb] = c * vp - s * vq;
            V[i * ldv + qb] = s * vp + c * vq;
          }
        }
      }
    }
    __syncthreads();
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(MT) void metric_eval_kernel(MetricArgsT<T> a, int ne, int lda, int ldv, int v0_lds) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int D = a.D, tid = threadIdx.x;
  T* A = reinterpret_cast<T*>(smem_raw);
  T* V = A + ne * lda;
  T* vec0 = V + D * ldv;          // lam~ (ne)
  T* vec1 = vec0 + ne;            // y / w / solve vector (ne)
  T* vec2 = vec1 + ne;            // d = X - mu, later z (ne)
  T* vec3 = vec2 + ne;            // Pd (ne)
  T* cs = vec3 + ne;              // 2 * NP
  T* red = cs + ne;               // MT / 64
  int* pq = reinterpret_cast<int*>(red + MT / 64);
  const bool softabs = a.metric == 1;
  // warm start: the shared eigenbasis V0 of the jitter-free Hs (LDS copy when it fits, else L2)
  const bool warm = softabs && a.V0 && a.lam0 && a.hs_stride == 0;
  T* V0s = reinterpret_cast<T*>(pq + ne);
  const T* V0 = a.V0;
  int ld0 = D;
  if (warm && v0_lds) {
    for (int e = tid; e < D * D; e += MT) { const int i = e / D, j = e - i * D; V0s[i * ldv + j] = a.V0[e]; }
    V0 = V0s; ld0 = ldv;
  }

  for (int64_t b = blockIdx.x; b < a.B; b += gridDim.x) {
    const uint64_t chain = a.chain_offset + (uint64_t)b;
    __syncthreads();
    // ---- 0. Gaussian log-prob and P (X - mu) at the evaluation point
    T logp = 0;
    if (a.X) {
      for (int i = tid; i < D; i += MT) vec2[i] = a.X[b * D + i] - a.mu[i];
      __syncthreads();
      T part = 0;
      for (int i = tid; i < D; i += MT) {
        T acc = 0;
        for (int k = 0; k < D; ++k) acc += a.Pm[(int64_t)k * D + i] * vec2[k];   // Pm symmetric: coalesced column read
        vec3[i] = acc;
        part += vec2[i] * acc;
      }
      logp = (T)a.log_norm - (T)0.5 * block_sum(part, red);
      if (a.upd_g) for (int i = tid; i < D; i += MT) a.upd_g[b * D + i] += (T)a.cg * vec3[i];
    }
    // ---- 1. Hs (lower triangle, as eigh(UPLO='L')) + jitter on the diagonal  (S:113-119)
    const T* Hs = a.Hs + b * a.hs_stride;
    if (!warm) {
      for (int e = tid; e < ne * ne; e += MT) {
        const int i = e / ne, j = e - i * ne;
        T v = 0;
        if (i < D && j < D) {
          v = (i >= j) ? Hs[(int64_t)i * D + j] : Hs[(int64_t)j * D + i];
          if (i == j && a.has_jitter) v += (T)a.jitter * uniform_elem<T>(a.seed, chain, a.draw, PURPOSE_JITTER, a.sub, i);
        }
        A[i * lda + j] = v;
      }
    } else {
      // A = V0^T (Hs + diag(e)) V0 = diag(lam0) + sum_i e_i v0_i v0_i^T   (v0_i = row i of V0), e = jitter * u
      __syncthreads();
      for (int i = tid; i < D; i += MT)
        vec1[i] = a.has_jitter ? (T)a.jitter * uniform_elem<T>(a.seed, chain, a.draw, PURPOSE_JITTER, a.sub, i) : (T)0;
      __syncthreads();
      for (int e = tid; e < ne * ne; e += MT) {
        const int k = e / ne, l = e - k * ne;
        if (k >= D || l >= D) { A[k * lda + l] = 0; continue; }
        if (l < k) continue;
        T acc = (k == l) ? a.lam0[k] : (T)0;
        if (a.has_jitter)
          for (int i = 0; i < D; ++i) acc += vec1[i] * V0[i * ld0 + k] * V0[i * ld0 + l];
        A[k * lda + l] = acc; A[l * lda + k] = acc;
      }
    }
    T logdet = 0, quad = 0;
    if (softabs) {
      for (int e = tid; e < D * D; e += MT) { const int i = e / D, j = e - i * D; V[i * ldv + j] = (i == j) ? (T)1 : (T)0; }
      __syncthreads();
      lds_jacobi<T>(A, V, D, ne, lda, ldv, cs, pq, red, a.max_sweeps);
      // lam~ = lam / tanh(alpha lam)   (S:120)
      T ld = 0;
      for (int i = tid; i < D; i += MT) {
        const T lam = A[i * lda + i];
        const T lt = ((T)1 / tanh((T)a.alpha * lam)) * lam;
        vec0[i] = lt;
        ld += log(lt);                                                  // S:726
        if (a.lam_out) a.lam_out[b * D + i] = lt;
        if (a.lamraw_out) a.lamraw_out[b * D + i] = lam;
      }
      logdet = block_sum(ld, red);
      if (warm && (a.V_out || a.G_out || a.p_out || a.L_out)) {
        // eigenvectors in the original basis: V <- V0 J  (through the A region; its eigenvalues are in vec0)
        __syncthreads();
        for (int e = tid; e < D * D; e += MT) {
          const int i = e / D, j = e - i * D;
          T acc = 0;
          for (int k = 0; k < D; ++k) acc += V0[i * ld0 + k] * V[k * ldv + j];
          A[i * lda + j] = acc;
        }
        __syncthreads();
        for (int e = tid; e < D * D; e += MT) { const int i = e / D, j = e - i * D; V[i * ldv + j] = A[i * lda + j]; }
        __syncthreads();
      }
      const bool rotated = warm && !(a.V_out || a.G_out || a.p_out || a.L_out);   // V still holds J
      if (a.V_out) for (int e = tid; e < D * D; e += MT) { const int i = e / D, j = e - i * D; a.V_out[b * D * D + e] = V[i * ldv + j]; }
      if (a.m) {                       // x = Q (Q^T m / lam~)
        if (rotated) {                 // m' = V0^T m
          for (int i = tid; i < D; i += MT) vec1[i] = a.m[b * D + i];
          __syncthreads();
          for (int k = tid; k < D; k += MT) {
            T acc = 0;
            for (int i = 0; i < D; ++i) acc += V0[i * ld0 + k] * vec1[i];
            vec2[k] = acc;
          }
        } else {
          for (int i = tid; i < D; i += MT) vec2[i] = a.m[b * D + i];
        }
        __syncthreads();
        T qd = 0;
        for (int k = tid; k < D; k += MT) {
          T acc = 0;
          for (int i = 0; i < D; ++i) acc += V[i * ldv + k] * vec2[i];
          const T w = acc / vec0[k];
          vec1[k] = w;
          qd += acc * w;
        }
        quad = block_sum(qd, red);
        if (rotated) {                 // x = V0 (J w)
          for (int i = tid; i < D; i += MT) {
            T acc = 0;
            for (int k = 0; k < D; ++k) acc += V[i * ldv + k] * vec1[k];
            vec3[i] = acc;
          }
          __syncthreads();
        }
        for (int i = tid; i < D; i += MT) {
          T acc = 0;
          if (rotated) { for (int k = 0; k < D; ++k) acc += V0[i * ld0 + k] * vec3[k]; }
          else { for (int k = 0; k < D; ++k) acc += V[i * ldv + k] * vec1[k]; }
          if (a.x_out) a.x_out[b * D + i] = acc;
          if (a.upd_x) a.upd_x[b * D + i] += (T)a.cx * acc;
        }
      }
      if (a.G_out || a.p_out || a.L_out) {          // G = Q diag(lam~) Q^T  (S:121), into the A region
        __syncthreads();
        for (int e = tid; e < D * D; e += MT) {
          const int i = e / D, j = e - i * D;
          if (j <= i) {
            T acc = 0;
            for (int k = 0; k < D; ++k) acc += V[i * ldv + k] * vec0[k] * V[j * ldv + k];
            A[i * lda + j] = acc; A[j * lda + i] = acc;
          }
        }
        __syncthreads();
        if (a.G_out) for (int e = tid; e < D * D; e += MT) { const int i = e / D, j = e - i * D; a.G_out[b * D * D + e] = A[i * lda + j]; }
      }
    } else {
      __syncthreads();
      if (a.G_out) for (int e = tid; e < D * D; e += MT) { const int i = e / D, j = e - i * D; a.G_out[b * D * D + e] = A[i * lda + j]; }
    }
    const bool need_chol = !softabs || a.p_out || a.L_out;
    if (need_chol) {
      __syncthreads();
      lds_cholesky<T>(A, D, lda);
      if (a.L_out) for (int e = tid; e < D * D; e += MT) { const int i = e / D, j = e - i * D; a.L_out[b * D * D + e] = (j <= i) ? A[i * lda + j] : (T)0; }
      if (!softabs) {
        T ld = 0;
        for (int i = tid; i < D; i += MT) ld += (T)2 * log(A[i * lda + i]);   // slogdet (S:728) for a PD metric
        logdet = block_sum(ld, red);
        if (a.m) {
          for (int i = tid; i < D; i += MT) { vec1[i] = a.m[b * D + i]; vec2[i] = vec1[i]; }
          lds_chol_solve<T>(A, D, lda, vec1);
          T qd = 0;
          for (int i = tid; i < D; i += MT) {
            qd += vec2[i] * vec1[i];
            if (a.x_out) a.x_out[b * D + i] = vec1[i];
            if (a.upd_x) a.upd_x[b * D + i] += (T)a.cx * vec1[i];
          }
          quad = block_sum(qd, red);
        }
      }
      if (a.p_out) {                   // p = L z  (S:184 via MultivariateNormal.rsample)
        __syncthreads();
        for (int i = tid; i < D; i += MT) vec2[i] = normal_elem<T>(a.seed, chain, a.draw, 0, i);
        __syncthreads();
        for (int i = tid; i < D; i += MT) {
          T acc = 0;
          for (int k = 0; k <= i; ++k) acc += A[i * lda + k] * vec2[k];
          a.p_out[b * D + i] = acc;
        }
      }
    }
    if (tid == 0) {
      if (a.logdet_out) a.logdet_out[b] = logdet;
      if (a.quad_out) a.quad_out[b] = quad;
      if (a.logp_out) a.logp_out[b] = logp;
      if (a.H_out) {
        // S:712: ndim * log(2 pi) is evaluated in float32 whatever the state dtype
        const float pi_term = (float)D * 1.8378770351409912f;
        a.H_out[b] = -logp + (T)0.5 * (T)pi_term + (T)0.5 * logdet + (T)0.5 * quad;   // S:731
      }
    }
  }
}

template <typename T> int metric_eval(const MetricArgsT<T>& a, hipStream_t s) {
  HTA_REQUIRE(a.B > 0 && a.D > 0 && a.Hs, "hta_metric_eval: bad shape / NULL Hs (B=%lld D=%d)", (long long)a.B, a.D);
  HTA_REQUIRE(a.metric == 0 || a.metric == 1, "hta_metric_eval: metric must be 0 (HESSIAN) or 1 (SOFTABS)");
  HTA_REQUIRE(!a.X || (a.Pm && a.mu), "hta_metric_eval: X given without Pm / mu");
  HTA_REQUIRE(!a.upd_g || a.X, "hta_metric_eval: upd_g needs X");
  const int D = a.D;
  const int ne = D + (D & 1);
  // LDS: A[ne][lda] + V[D][ldv] + 5 vectors + reduction scratch + pair table; pad the leading dimensions
  // to an odd stride when that still fits in the 160 KiB of one CU
  auto bytes = [&](int lda, int ldv) {
    return (size_t)(ne * lda + D * ldv + 5 * ne + MT / 64) * sizeof(T) + (size_t)ne * sizeof(int) + 64;
  };
  int lda = ne + 1, ldv = (D | 1);
  if (bytes(lda, ldv) > 160 * 1024) { lda = ne; ldv = D; }
  size_t lds = bytes(lda, ldv);
  const bool warm = a.metric == 1 && a.V0 && a.lam0 && a.hs_stride == 0;
  int v0_lds = 0;
  if (warm && lds + (size_t)D * ldv * sizeof(T) <= 160 * 1024) { v0_lds = 1; lds += (size_t)D * ldv * sizeof(T); }
  HTA_REQUIRE(lds <= 160 * 1024, "hta_metric_eval: D=%d does not fit the 160 KiB LDS of a CU for this dtype (max ~140 fp32 / ~99 fp64)", D);
  static bool attr_f = false, attr_d = false;
  bool& done = sizeof(T) == 4 ? attr_f : attr_d;
  if (!done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&metric_eval_kernel<T>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { set_error("hta_metric_eval: hipFuncSetAttribute: %s", hipGetErrorString(e)); return HTA_ERR_LAUNCH; }
    done = true;
  }
  MetricArgsT<T> k = a;
  if (k.max_sweeps <= 0) k.max_sweeps = sizeof(T) == 4 ? 16 : 24;
  const int grid = (int)(a.B < 65536 ? a.B : 65536);
  profile_begin(s);
  metric_eval_kernel<T><<<grid, MT, lds, s>>>(k, ne, lda, ldv, v0_lds);
  profile_end(s);
  HTA_CHECK_LAUNCH("hta_metric_eval");
  return HTA_OK;
}

template int metric_eval<float>(const MetricArgsT<float>&, hipStream_t);
template int metric_eval<double>(const MetricArgsT<double>&, hipStream_t);

}  // namespace hta

extern "C" {
int hta_metric_eval_f32(const HtaMetricArgs* args, void* stream) {
  if (!args) { hta::set_error("hta_metric_eval: NULL args"); return HTA_ERR_INVALID; }
  return hta::metric_eval<float>(*reinterpret_cast<const hta::MetricArgsT<float>*>(args), (hipStream_t)stream);
}
int hta_metric_eval_f64(const HtaMetricArgs* args, void* stream) {
  if (!args) { hta::set_error("hta_metric_eval: NULL args"); return HTA_ERR_INVALID; }
  return hta::metric_eval<double>(*reinterpret_cast<const hta::MetricArgsT<double>*>(args), (hipStream_t)stream);
}
}
