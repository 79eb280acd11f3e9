// Device building blocks shared by the metric-evaluation kernels (rmhmc_metric.hip: any dtype / any symmetric input,
// VALU; rmhmc_metric_mfma.hip: fp32 evaluations that start from a shared eigenbasis, on the matrix cores).
#pragma once
#include "common.hpp"
#include "philox.hpp"
#include "rmhmc.hpp"

namespace hta {

#ifndef HTA_MT
#define HTA_MT 1024
#endif
constexpr int MT = HTA_MT;  // threads per system

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
  const int m = n - 1;                     // r, k < m: one conditional subtract / add replaces the modulo
  int x = r + k; x -= (x >= m) ? m : 0;
  int y = r - k; y += (y < 0) ? m : 0;
  if (k == 0) { x = m; y = r; }
  p = min(x, y); q = max(x, y);
}

template <typename T> struct Vec16;
template <> struct Vec16<float> { typedef float type __attribute__((ext_vector_type(4))); static constexpr int N = 4; };
template <> struct Vec16<double> { typedef double type __attribute__((ext_vector_type(2))); static constexpr int N = 2; };
template <typename T> struct Vec8;      // (c, s) pair
template <> struct Vec8<float> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct Vec8<double> { typedef double type __attribute__((ext_vector_type(2))); };

// Jacobi rotation annihilating a_pq.  This is the serial part of every round (NP lanes of one wave while
// 15 waves wait at the barrier), so fp32 uses the single-instruction reciprocal / rsqrt (1 ulp): the
// similarity transform stays orthogonal to rounding whatever the angle's accuracy, which only affects
// how fast the off-diagonal mass decays.
template <typename T> __device__ __forceinline__ void rotation(T app, T aqq, T apq, T& c, T& s);
template <> __device__ __forceinline__ void rotation<float>(float app, float aqq, float apq, float& c, float& s) {
  const float theta = 0.5f * (aqq - app) * __frcp_rn(apq);
  const float at = fabsf(theta);
  // t = sgn(theta) / (|theta| + sqrt(1 + theta^2)); huge |theta| -> t = 0 (inf-safe)
  const float t = copysignf(__frcp_rn(at + __fsqrt_rn(fmaf(theta, theta, 1.0f))), theta);
  c = __frsqrt_rn(fmaf(t, t, 1.0f));
  s = t * c;
}
template <> __device__ __forceinline__ void rotation<double>(double app, double aqq, double apq, double& c, double& s) {
  const double theta = (aqq - app) / (2.0 * apq);
  const double t = copysign(1.0, theta) / (fabs(theta) + sqrt(1.0 + theta * theta));
  c = 1.0 / sqrt(1.0 + t * t);
  s = t * c;
}

// Cyclic Jacobi.  A[ne][lda]: symmetric, only the UPPER triangle (i <= j) is read and kept up to date.
// VT[D][ldv]: row k = eigenvector k (transposed storage: a rotation mixes two contiguous rows, so it
// moves 16 bytes per LDS instruction); ldv is a multiple of 16 bytes.
// LDS traffic per round is what bounds this loop (16 waves share one LDS), so: (p, q) packed in one word
// and (c, s) in one 8-byte word (2 + 2 loads per pair-block instead of 8), upper-only A (4 loads + 4 stores
// per block, no mirror stores), vectorised VT rows, per-thread work lists decoded once.
template <typename T, int MAXB, int MAXV>
__device__ void lds_jacobi(T* A, T* VT, int D, int ne, int lda, int ldv, T* cs_raw, int* pq, T* red, int max_sweeps) {
  typedef typename Vec16<T>::type V16;
  typedef typename Vec8<T>::type CS;
  constexpr int VN = Vec16<T>::N;
  CS* cs = reinterpret_cast<CS*>(cs_raw);
  const int tid = threadIdx.x;
  const int NP = ne / 2;
  const int nblk = NP * (NP + 1) / 2;
  int blkA[MAXB], blkB[MAXB];                   // MAXB * MT >= nblk, MAXV * MT >= NP * nv (checked at launch)
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
  const int nv = ldv / VN;                      // 16-byte vectors per VT row
  int vtB[MAXV], vtG[MAXV];
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int e = tid + k * MT;
    vtB[k] = (e < NP * nv) ? e / nv : -1;
    vtG[k] = (e < NP * nv) ? e - (e / nv) * nv : 0;
  }
  T off_prev = (T)-1;
  for (int sweep = 0; sweep < max_sweeps; ++sweep) {
    // convergence: off-diagonal vs diagonal mass (upper triangle, off-diagonal counted twice)
    T off = 0, dg = 0;
    for (int e = tid; e < D * D; e += MT) {
      const int i = e / D, j = e - i * D;
      if (j >= i) {
        const T a = A[i * lda + j];
        if (i == j) dg += a * a; else off += (T)2 * a * a;
      }
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
        rr_pair(ne, r, tid, p, q);                                     // p < q
        const T app = A[p * lda + p], aqq = A[q * lda + q], apq = A[p * lda + q];
        T c = 1, s = 0;
        if (apq != (T)0) rotation<T>(app, aqq, apq, c, s);
        CS v; v.x = c; v.y = s;
        cs[tid] = v;
      }
      __syncthreads();
      // A <- J^T A J, one thread per pair-block (a <= b); every element lives at (min, max).  The pair table is
      // recomputed from (r, slot) in registers instead of being read back from LDS: one dependent LDS round trip
      // less on the critical path of every round.
#pragma unroll
      for (int k = 0; k < MAXB; ++k) {
        const int a = blkA[k], bb = blkB[k];
        if (a >= 0) {
          int pa, qa, pb, qb;
          rr_pair(ne, r, a, pa, qa);
          rr_pair(ne, r, bb, pb, qb);
          const CS ra = cs[a], rb = cs[bb];
          const T ca = ra.x, sa = ra.y, cb = rb.x, sb = rb.y;
          const int i00 = min(pa, pb) * lda + max(pa, pb), i01 = min(pa, qb) * lda + max(pa, qb);
          const int i10 = min(qa, pb) * lda + max(qa, pb), i11 = min(qa, qb) * lda + max(qa, qb);
          const T m00 = A[i00], m01 = A[i01], m10 = A[i10], m11 = A[i11];
          const T t00 = cb * m00 - sb * m01, t01 = sb * m00 + cb * m01;
          const T t10 = cb * m10 - sb * m11, t11 = sb * m10 + cb * m11;
          T n00 = ca * t00 - sa * t10, n01 = ca * t01 - sa * t11;
          T n10 = sa * t00 + ca * t10, n11 = sa * t01 + ca * t11;
          if (a == bb) { n01 = 0; n10 = 0; }                           // i01 == i10 here: the annihilated element
          A[i00] = n00; A[i01] = n01; A[i10] = n10; A[i11] = n11;
        }
      }
      // VT rows p, q <- rotation, 16 bytes at a time
#pragma unroll
      for (int k = 0; k < MAXV; ++k) {
        const int vb = vtB[k];
        if (vb >= 0) {
          int pb, qb;
          rr_pair(ne, r, vb, pb, qb);
          if (qb < D) {     // the padding index (odd D) never rotates
            const CS rb = cs[vb];
            V16* rp = reinterpret_cast<V16*>(VT + pb * ldv) + vtG[k];
            V16* rq = reinterpret_cast<V16*>(VT + qb * ldv) + vtG[k];
            const V16 vp = *rp, vq = *rq;
            *rp = rb.x * vp - rb.y * vq;
            *rq = rb.y * vp + rb.x * vq;
          }
        }
      }
    }
    __syncthreads();
  }
  __syncthreads();
}

// The same sweeps with the work lists walked at RUN time (round 5, the instance for every size beyond the register lists of
// lds_jacobi: D > 254 fp32 / 180 fp64, both matrices in the caller's slab): a thread decodes its pair-blocks and its VT vectors
// again in every round.  Slow on purpose - the reference has no size limit (S:108-122), so this is an answer, not an error.
// Same arithmetic, same order inside a round (a round's rotations are disjoint: any assignment of blocks to threads gives the same
// result).  Limits left: one thread per rotation of a round (D <= 2 MT) and the callers' own (D <= MT).
template <typename T>
__device__ void lds_jacobi_dyn(T* A, T* VT, int D, int ne, int lda, int ldv, T* cs_raw, T* red, int max_sweeps) {
  typedef typename Vec16<T>::type V16;
  typedef typename Vec8<T>::type CS;
  constexpr int VN = Vec16<T>::N;
  CS* cs = reinterpret_cast<CS*>(cs_raw);
  const int tid = threadIdx.x;
  const int NP = ne / 2;
  const int nblk = NP * (NP + 1) / 2;
  const int nv = ldv / VN;
  T off_prev = (T)-1;
  for (int sweep = 0; sweep < max_sweeps; ++sweep) {
    T off = 0, dg = 0;
    for (int e = tid; e < D * D; e += MT) {
      const int i = e / D, j = e - i * D;
      if (j >= i) {
        const T a = A[(int64_t)i * lda + j];
        if (i == j) dg += a * a; else off += (T)2 * a * a;
      }
    }
    off = block_sum(off, red);
    dg = block_sum(dg, red);
    const T tol = (T)64 * Eps<T>::v * Eps<T>::v;
    if (!(off > tol * (dg + off))) break;
    if (sweep >= 4 && off_prev >= (T)0 && off > (T)0.25 * off_prev && off <= (T)1e-6 * (dg + off)) break;
    off_prev = off;
    for (int r = 0; r < ne - 1; ++r) {
      __syncthreads();
      if (tid < NP) {
        int p, q;
        rr_pair(ne, r, tid, p, q);
        const T app = A[(int64_t)p * lda + p], aqq = A[(int64_t)q * lda + q], apq = A[(int64_t)p * lda + q];
        T c = 1, s = 0;
        if (apq != (T)0) rotation<T>(app, aqq, apq, c, s);
        CS v; v.x = c; v.y = s;
        cs[tid] = v;
      }
      __syncthreads();
      // row a of the upper block triangle starts at block a NP - a (a - 1) / 2: a thread walks its blocks with (a, bb) carried along
      for (int e = tid; e < nblk; e += MT) {
        int a = (int)(((double)(2 * NP + 1) - sqrt((double)(2 * NP + 1) * (double)(2 * NP + 1) - 8.0 * (double)e)) * 0.5);
        if (a < 0) a = 0;
        while (a > 0 && a * NP - a * (a - 1) / 2 > e) --a;
        while ((a + 1) * NP - (a + 1) * a / 2 <= e) ++a;
        const int bb = a + (e - (a * NP - a * (a - 1) / 2));
        int pa, qa, pb, qb;
        rr_pair(ne, r, a, pa, qa);
        rr_pair(ne, r, bb, pb, qb);
        const CS ra = cs[a], rb = cs[bb];
        const T ca = ra.x, sa = ra.y, cb = rb.x, sb = rb.y;
        const int64_t i00 = (int64_t)min(pa, pb) * lda + max(pa, pb), i01 = (int64_t)min(pa, qb) * lda + max(pa, qb);
        const int64_t i10 = (int64_t)min(qa, pb) * lda + max(qa, pb), i11 = (int64_t)min(qa, qb) * lda + max(qa, qb);
        const T m00 = A[i00], m01 = A[i01], m10 = A[i10], m11 = A[i11];
        const T t00 = cb * m00 - sb * m01, t01 = sb * m00 + cb * m01;
        const T t10 = cb * m10 - sb * m11, t11 = sb * m10 + cb * m11;
        T n00 = ca * t00 - sa * t10, n01 = ca * t01 - sa * t11;
        T n10 = sa * t00 + ca * t10, n11 = sa * t01 + ca * t11;
        if (a == bb) { n01 = 0; n10 = 0; }
        A[i00] = n00; A[i01] = n01; A[i10] = n10; A[i11] = n11;
      }
      for (int e = tid; e < NP * nv; e += MT) {
        const int vb = e / nv, vg = e - vb * nv;
        int pb, qb;
        rr_pair(ne, r, vb, pb, qb);
        if (qb < D) {
          const CS rb = cs[vb];
          V16* rp = reinterpret_cast<V16*>(VT + (int64_t)pb * ldv) + vg;
          V16* rq = reinterpret_cast<V16*>(VT + (int64_t)qb * ldv) + vg;
          const V16 vp = *rp, vq = *rq;
          *rp = rb.x * vp - rb.y * vq;
          *rq = rb.y * vp + rb.x * vq;
        }
      }
    }
    __syncthreads();
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// d lam~ / d lam for the soft-abs map lam~ = lam coth(alpha lam) (S:120):  coth x - x / sinh^2 x with x = alpha lam
// (an odd function of x; series below |x| = 0.3 where the closed form cancels, exp(-2|x|) form elsewhere: no overflow)
template <typename T> __device__ __forceinline__ T softabs_slope(T alpha, T lam) {
  const T x = alpha * lam, ax = fabs(x);
  if (ax < (T)0.3) {
    const T x2 = x * x;
    return x * ((T)(2.0 / 3.0) - x2 * ((T)(4.0 / 45.0) - x2 * ((T)(12.0 / 945.0) - x2 * (T)(8.0 / 4725.0))));
  }
  const T t = exp((T)-2 * ax), om = (T)1 - t;
  const T d = ((T)1 + t) / om - ax * (T)4 * t / (om * om);
  return x < 0 ? -d : d;
}

}  // namespace hta
