// Explicit RMHMC for a Gaussian target when the soft-abs map is the identity: the whole `sample` loop
// (hamiltorch/samplers.py:969-1026 with the integrator S:425-461) for one chain per workgroup, every
// trajectory of a run in ONE launch.
//
// Why this is the same computation as csrc/rmhmc_explicit.hip + rmhmc_metric.hip.  The target's curvature is one
// matrix P for all chains; an evaluation's metric is G = softabs(F), F = P + diag(e), e = jitter u >= 0 (S:113-121).
// lam coth(alpha lam) = lam (1 + 2 exp(-2 alpha lam) + ...): once alpha lam_min(P) >= 20 the correction is below
// 1e-17 relative, i.e. G == F in fp32 and fp64 (and G = F by definition for Metric.HESSIAN).  Then
//   G^-1 m    = (P + E)^-1 m:  x_0 = S m,  x_{k+1} = x_0 - S (e . x_k)  with the shared S = P^-1; contraction factor
//               rho <= jitter / lam_min(P), so K = O(log eps / log rho) products with ONE shared matrix (3 in fp32 at
//               the BASELINE config-3 numbers) instead of an eigendecomposition per chain and evaluation;
//   log |G|   = log |F| and the momentum draw p = chol(G) z (S:183-184): a Cholesky factorisation of F in LDS,
//               three per trajectory (gibbs, H_old, H_new);
//   dH/dtheta = P (theta - mu)  (constant curvature).
// The host (rmhmc_explicit.hip) takes this path only when those conditions hold; everything else - finite alpha,
// indefinite curvature, rho > 1/4, matrices beyond the LDS - keeps the Jacobi path.  Same Philox sub-streams, same
// update order (sequential phi_C, Q1; un-augmented H_new, Q4; first post-burn rejection resets to params_init, Q2).
//
// Layout: P, S and the Cholesky work matrix in LDS (row stride D|1), the four state vectors and scratch vectors in
// LDS, 256 threads: a matrix-vector product is 2 threads per row; the Cholesky trailing update a 16 x 16 thread tile.
#include <math.h>
#include <utility>
#include "common.hpp"
#include "philox.hpp"
#include "rmhmc.hpp"
#include "rmhmc_fused_dev.hpp"

#ifndef HTA_RM_TIMING
#define HTA_RM_TIMING 0   // developer cycle counters (thread 0 of block 0): tools/scratch/rmhmc_time.py prints them
#endif
#if HTA_RM_TIMING
__device__ unsigned long long hta_rm_dbg[16];
extern "C" void hta_rm_dbg_read(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(hta_rm_dbg), sizeof(hta_rm_dbg)); }
#define HTA_XTICK(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); xacc[k] += now_ - xlast; xlast = now_; } while (0)
#define HTA_RTICK(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); ch.tacc[0] += now_ - ch.tlast; ch.tlast = now_; } while (0)
#define HTA_MTICK(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); tacc[k] += now_ - tlast; tlast = now_; } while (0)
#else
#define HTA_RTICK(k) do {} while (0)
#define HTA_MTICK(k) do {} while (0)
#define HTA_XTICK(k) do {} while (0)
#endif

namespace hta {

constexpr int FNT = 256;          // 4 waves: (row block of 64) x (half of the contraction range)
constexpr int FCB = 4;            // Cholesky panel width

template <typename T> __device__ __forceinline__ T fast_rsqrt(T v) { return (T)1 / sqrt(v); }
template <> __device__ __forceinline__ float fast_rsqrt<float>(float v) { return __builtin_amdgcn_rsqf(v); }   // v_rsq_f32, 1 ulp

// W = P + diag(ev) -> its Cholesky factor (strictly-lower part in W, diagonal in dg), right-looking in panels of
// FCB columns: the panel rows are one thread each (the FCB x FCB diagonal block is refactored by every thread from
// LDS: no extra barrier), the rank-FCB trailing update a 16 x 16 thread tile with the panel rows it needs in
// registers.  Returns this thread's share of log |F| (sum over threads = log |F|).
template <typename T>
__device__ __forceinline__ T chol_in_lds(const T* __restrict__ P, const T* ev, T* W, T* dg, int D, int ld, int tid) {
  __syncthreads();
  {
    const float invD = 1.0f / (float)D;
    for (int e0 = tid; e0 < D * D; e0 += 8 * FNT) {       // 8 independent L2 loads in flight per thread
      T t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int e = e0 + u * FNT; t[u] = e < D * D ? P[e] : (T)0; }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u * FNT;
        int i = (int)(((float)e + 0.5f) * invD);             // e / D without the integer divide (e < 2^14)
        const int j = e - i * D;
        if (e < D * D && j <= i) W[i * ld + j] = t[u] + (i == j ? ev[i] : (T)0);
      }
    }
  }
  __syncthreads();
  const int tx = tid & 15, ty = tid >> 4;
  for (int kb = 0; kb < D; kb += FCB) {
    const int nb = min(FCB, D - kb);
    // --- panel: diagonal block factor (registers, every thread), then this thread's row of the panel
    T Ld[FCB][FCB], rinv[FCB];
#pragma unroll
    for (int c = 0; c < FCB; ++c)
#pragma unroll
      for (int c2 = 0; c2 < FCB; ++c2) Ld[c][c2] = (c < nb && c2 <= c) ? W[(kb + c) * ld + kb + c2] : (T)(c == c2);
#pragma unroll
    for (int c = 0; c < FCB; ++c) {
#pragma unroll
      for (int c2 = 0; c2 < c; ++c2) {
        T v = Ld[c][c2];
#pragma unroll
        for (int c3 = 0; c3 < c2; ++c3) v = fma(-Ld[c][c3], Ld[c2][c3], v);
        Ld[c][c2] = v * rinv[c2];
      }
      T v = Ld[c][c];
#pragma unroll
      for (int c3 = 0; c3 < c; ++c3) v = fma(-Ld[c][c3], Ld[c][c3], v);
      rinv[c] = fast_rsqrt<T>(v);
      Ld[c][c] = v * rinv[c];
    }
    T lrow[FCB];
    const int i = kb + FCB + tid;                       // rows below the diagonal block: one thread each
    const bool below = i < D;
    if (below) {
#pragma unroll
      for (int c = 0; c < FCB; ++c) {
        T v = c < nb ? W[i * ld + kb + c] : (T)0;
#pragma unroll
        for (int c3 = 0; c3 < c; ++c3) v = fma(-lrow[c3], Ld[c][c3], v);
        lrow[c] = v * rinv[c];
      }
    }
    __syncthreads();                                     // every thread has read the old panel / diagonal block
    if (below) {
#pragma unroll
      for (int c = 0; c < FCB; ++c) if (c < nb) W[i * ld + kb + c] = lrow[c];
    }
#pragma unroll
    for (int c = 0; c < FCB; ++c) {                      // rows of the diagonal block itself (static indices: Ld stays in registers)
      if (tid == c && c < nb) {
        dg[kb + c] = Ld[c][c];
#pragma unroll
        for (int c2 = 0; c2 < c; ++c2) W[(kb + c) * ld + kb + c2] = Ld[c][c2];
      }
    }
    __syncthreads();
    // --- trailing update with the finished panel: rows ii = r0 + 16 a (a-th slot of ty), columns j = c0 + 16 b (tx)
    const int rbase = kb + FCB + ty, cbase = kb + FCB + tx;
    for (int ii = rbase; ii < D; ii += 16) {
      T* wrow = W + ii * ld;
      T li[FCB];
#pragma unroll
      for (int c = 0; c < FCB; ++c) li[c] = wrow[kb + c];
      for (int j = cbase; j <= ii; j += 32) {            // two column slots per trip: both sets of loads in flight together
        const int j2 = j + 16;
        const bool two = j2 <= ii;
        const T* p1 = W + j * ld + kb;
        const T* p2 = W + (two ? j2 : j) * ld + kb;
        T l1[FCB], l2[FCB];
#pragma unroll
        for (int c = 0; c < FCB; ++c) { l1[c] = p1[c]; l2[c] = p2[c]; }
        T v1 = wrow[j], v2 = wrow[two ? j2 : j];
#pragma unroll
        for (int c = 0; c < FCB; ++c) { v1 = fma(-li[c], l1[c], v1); v2 = fma(-li[c], l2[c], v2); }
        wrow[j] = v1;
        if (two) wrow[j2] = v2;
      }
    }
    __syncthreads();
  }
  return tid < D ? (T)2 * log(dg[tid]) : (T)0;
}


// Sum of the two half-row partials of a row: lanes l (lower k half) and l + 32 (upper half) of one wave.  The total is
// valid in the UPPER lane (the row's owner).  v_permlane32_swap_b32 (gfx950) puts the lower half of `h` under the upper
// lanes in one VALU op - an LDS bpermute round trip here costs more than the barrier it saves.
__device__ __forceinline__ float row_total(float h) {
  const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, h), __builtin_bit_cast(unsigned, h), false, false);
  return h + __builtin_bit_cast(float, r[0]);
}
__device__ __forceinline__ double row_total(double h) { return h + __shfl_xor(h, 32, 64); }

// acc += s * (lane J of this lane's 16-lane row of u): v_fmac_f32 with the DPP row_newbcast control (gfx90a+).  A chunk of
// 16 vector elements then costs ONE 4-byte LDS load per lane (lane l fetches element l & 15) and 16 FMAs, instead of four
// 16-byte loads that every lane repeats: the product passes were LDS-bandwidth bound (8 clocks per 16-byte wave load).
template <int J> __device__ __forceinline__ void fmac_bcast(float& acc, float u, float s) {
  asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(u), "v"(s), "n"(J));
}
template <int... J>
__device__ __forceinline__ void chunk_fma(const float* reg, float u, float (&acc)[2], std::integer_sequence<int, J...>) {
  (fmac_bcast<J>(acc[J & 1], u, reg[J]), ...);
}

// KH: register-resident slice of a matrix column per thread (multiple of 8).  Lane (row, half) keeps P[k][row] and
// S[k][row] for its KH values of k in VGPRs for the whole launch; a matrix-vector product then only streams the
// vector (16-byte LDS reads).  The two halves of a row sit in the SAME wave (lanes l and l + 32): their partial sums
// meet through one lane swap, the upper lane ("owner" of the row) finishes the element-wise work of its element in
// registers - no partial-sum vectors, no combine phases: one barrier per product pass.
// NC chains can share a workgroup (NC = 2): the slices do not depend on the chain, so a pass carries both chains for the
// same register reads and barriers - but not for the same LDS traffic, which is what bounds a pass (see the dispatch).
constexpr int FVC = 16;           // LDS vectors per chain: pm pmc ev d0 d1 w0 w1 | dc + 8 solve vectors of the tracked schedule
                                  // (128 entries each, zero beyond D)

template <typename T, int KH, int NC, bool TRACK = false> struct Fused {
  typedef T V4 __attribute__((ext_vector_type(4)));
  static constexpr int CHS = FVC * 128;
  const FusedArgs<T>& a;
  int D, ld, tid, row, k0, dpar;
  bool own;
  T Preg[KH], Sreg[KH];
  T mu_r, sd_r;                       // mu[row], S[row][row]
  T ev_r[NC];                         // this owner's elements of the current evaluation's jitter
  T scur[NC], sth[NC], spm[NC], sthc[NC], spmc[NC];   // the owner's element of the chain state (theta, p and their copies):
                                      // registers; only the momenta are mirrored in LDS (they are product operands)
  int jslot;                          // next unread evaluation slot of the jitter buffer
  T y[NC], yc[NC], z[NC], zc[NC];     // TRACK: the owner's element of P (theta - mu), P (theta_c - mu), S p, S p_c
  T *pm, *pmc, *ev, *d0, *d1, *w0, *w1, *dc, *ws, *dg, *red, *W, *jb;
  uint64_t chain[NC];
  bool live[NC];
#if HTA_RM_TIMING
  unsigned long long tacc[8] = {0}, tlast = 0;
#endif
  __device__ Fused(const FusedArgs<T>& a_) : a(a_) {}

  __device__ __forceinline__ void load_slices() {           // once per launch (symmetric matrices: column `row`)
    const bool rowok = row < D;
#pragma unroll
    for (int kk = 0; kk < KH; ++kk) {
      const int k = k0 + kk;
      const bool ok = rowok && k < D;
      Preg[kk] = ok ? a.P[(int64_t)k * D + row] : (T)0;
      Sreg[kk] = ok ? a.S[(int64_t)k * D + row] : (T)0;
    }
  }

  // NC x 4 block sums at once (every thread gets all of them)
  __device__ __forceinline__ void block_sums(T (&v)[NC][4]) {
#pragma unroll
    for (int q = 0; q < NC; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[q][e] = wave_sum(v[q][e]);
    __syncthreads();
    if ((tid & 63) == 0) {
#pragma unroll
      for (int q = 0; q < NC; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[(tid >> 6) * 8 + q * 4 + e] = v[q][e];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NC; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        T s = 0;
#pragma unroll
        for (int i = 0; i < FNT / 64; ++i) s += red[i * 8 + q * 4 + e];
        v[q][e] = s;
      }
  }

  // Row `row` of up to three symmetric-matrix x vector products per chain, complete in both lanes of the row:
  //   o1 = P v1,  o2 = S v2,  o3 = (S . S) v3  (element-wise square: the second-order log-det term).
  // No barriers inside; vectors and slices are zero beyond D.
  template <bool WITH_P, int NS>
  __device__ __forceinline__ void products(const T* v1, const T* v2, const T* v3, T (&o1)[NC], T (&o2)[NC], T (&o3)[NC]) {
    T a1[NC][2], a2[NC][2], a3[NC][2];
#pragma unroll
    for (int q = 0; q < NC; ++q) { a1[q][0] = a1[q][1] = a2[q][0] = a2[q][1] = a3[q][0] = a3[q][1] = 0; }
    constexpr int NCH = KH / 4;
#pragma unroll
    for (int q = 0; q < NC; ++q) {
      if (NS <= 1 && sizeof(T) == 4) {
        // fp32: one 4-byte load per lane and chunk of 16 elements, broadcast inside the FMA (see fmac_bcast)
        constexpr int NB = (KH + 15) / 16;
        float u1[NB], u2[NB];
        const int l15 = tid & 15;
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
          const int off = q * CHS + k0 + 16 * cb + ((16 * cb + 8 < KH) ? l15 : (l15 & 7));    // a trailing half chunk has 8 elements
          if (WITH_P) u1[cb] = v1[off];
          if (NS >= 1) u2[cb] = v2[off];
        }
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
          if (16 * cb + 8 < KH) {
            if (WITH_P) chunk_fma(reinterpret_cast<const float*>(Preg) + 16 * cb, u1[cb], reinterpret_cast<float(&)[2]>(a1[q]), std::make_integer_sequence<int, 16>{});
            if (NS >= 1) chunk_fma(reinterpret_cast<const float*>(Sreg) + 16 * cb, u2[cb], reinterpret_cast<float(&)[2]>(a2[q]), std::make_integer_sequence<int, 16>{});
          } else {
            if (WITH_P) chunk_fma(reinterpret_cast<const float*>(Preg) + 16 * cb, u1[cb], reinterpret_cast<float(&)[2]>(a1[q]), std::make_integer_sequence<int, 8>{});
            if (NS >= 1) chunk_fma(reinterpret_cast<const float*>(Sreg) + 16 * cb, u2[cb], reinterpret_cast<float(&)[2]>(a2[q]), std::make_integer_sequence<int, 8>{});
          }
        }
      } else if (NS <= 1) {
        // every 16-byte vector load of this chain first, then the FMAs: left to itself the compiler issues three loads
        // and waits for them, 14 exposed LDS round trips per pass
        V4 u1[NCH], u2[NCH];
#pragma unroll
        for (int cb = 0; cb < NCH; ++cb) {
          if (WITH_P) u1[cb] = *reinterpret_cast<const V4*>(v1 + q * CHS + k0 + 4 * cb);
          if (NS >= 1) u2[cb] = *reinterpret_cast<const V4*>(v2 + q * CHS + k0 + 4 * cb);
        }
#pragma unroll
        for (int cb = 0; cb < NCH; ++cb) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (WITH_P) a1[q][e & 1] = fma(Preg[4 * cb + e], u1[cb][e], a1[q][e & 1]);
            if (NS >= 1) a2[q][e & 1] = fma(Sreg[4 * cb + e], u2[cb][e], a2[q][e & 1]);
          }
        }
      } else {
#pragma unroll
        for (int kk = 0; kk < KH; kk += 4) {
          V4 u1, u2, u3;
          if (WITH_P) u1 = *reinterpret_cast<const V4*>(v1 + q * CHS + k0 + kk);
          u2 = *reinterpret_cast<const V4*>(v2 + q * CHS + k0 + kk);
          u3 = *reinterpret_cast<const V4*>(v3 + q * CHS + k0 + kk);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (WITH_P) a1[q][e & 1] = fma(Preg[kk + e], u1[e], a1[q][e & 1]);
            a2[q][e & 1] = fma(Sreg[kk + e], u2[e], a2[q][e & 1]);
            a3[q][e & 1] = fma(Sreg[kk + e] * Sreg[kk + e], u3[e], a3[q][e & 1]);
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < NC; ++q) {
      if (WITH_P) o1[q] = row_total(a1[q][0] + a1[q][1]);
      if (NS >= 1) o2[q] = row_total(a2[q][0] + a2[q][1]);
      if (NS >= 2) o3[q] = row_total(a3[q][0] + a3[q][1]);
    }
  }

  __device__ __forceinline__ T jitter_elem(int q, uint32_t n, uint32_t sub) {
    return (a.has_jitter && own && live[q]) ? a.jitter * uniform_elem<T>(a.seed, chain[q], n, PURPOSE_JITTER, sub, row) : (T)0;
  }

  // Jitter of the next NSLOT evaluations of the integrator in ONE Philox pass per wave: lane (slot, blk) draws the block
  // of 4 uniforms of rows 32 wave + 4 blk .. + 3 for evaluation `slot` (NC == 1: 8 evaluations = 2 steps; NC == 2: 4
  // evaluations x 2 chains = 1 step).  A wave only ever reads the rows it wrote: no barrier.  (Drawing per evaluation
  // costs ~100 VALU instructions per wave and half step however few lanes need it - a fifth of the half step.)
  static constexpr int NSLOT = 8 / NC;
  __device__ __forceinline__ void refill_jitter(uint32_t n, int l) {
    jslot = 0;
    if (!a.has_jitter) return;
    const int lane = tid & 63, slot = lane >> 3, blk = lane & 7;
    const int q = NC == 1 ? 0 : (slot & 1), e = NC == 1 ? slot : (slot >> 1);
    const int which = e & 3;
    const uint32_t sub = 2u + 8u * (uint32_t)(l + (e >> 2)) + (which == 0 ? 1u : which == 1 ? 2u : which == 2 ? 4u : 7u);
    const int r0 = 32 * (tid >> 6) + 4 * blk;
    const U4 r = philox_block(a.seed, chain[q], n, PURPOSE_JITTER, sub, (uint32_t)(r0 >> 2));
    const V4 val = {a.jitter * u23<T>(r.x), a.jitter * u23<T>(r.y), a.jitter * u23<T>(r.z), a.jitter * u23<T>(r.w)};
    *reinterpret_cast<V4*>(jb + (e * NC + q) * 128 + r0) = val;
  }

  // x = (P + diag(e))^-1 m continued from x0 = S m: xr <- x0 - S (e . x), K times; w buffers alternate (a fast wave's
  // store of the next e . x must not overtake a slow wave's reads of the current one)
  __device__ __forceinline__ void refine(const T (&x0)[NC], T (&xr)[NC]) {
    for (int it = 0; it < a.K; ++it) {
      const T* wr = (it & 1) ? w1 : w0;
      T* ww = (it & 1) ? w0 : w1;
      __syncthreads();
      T sx[NC], u1[NC], u3[NC];
      products<false, 1>(nullptr, wr, nullptr, u1, sx, u3);
      if (own) {
#pragma unroll
        for (int q = 0; q < NC; ++q) { xr[q] = x0[q] - sx[q]; ww[q * CHS + row] = ev_r[q] * xr[q]; }
      }
    }
  }

  // one half step (csrc/rmhmc_explicit.hip:half_step): upd_x += eh G(X)^-1 m ; upd_g -= eh P (X - mu); its jitter is
  // the next slot of the buffer refill_jitter() filled.  X, upd_x: owner registers; m: LDS vector; upd_g: owner register
  // mirrored to its LDS vector.  1 + K barriers.
  __device__ __forceinline__ void half_step(const T (&X)[NC], const T* m, T (&upd_x)[NC], T (&upd_g)[NC], T* upd_g_lds, T eh) {
    T* d = dpar ? d1 : d0;                                 // alternate: with K == 0 nothing else separates two evaluations
    dpar ^= 1;
    if (own) {
#pragma unroll
      for (int q = 0; q < NC; ++q) {
        ev_r[q] = a.has_jitter ? jb[(jslot * NC + q) * 128 + row] : (T)0;
        d[q * CHS + row] = X[q] - mu_r;
      }
    }
    ++jslot;
    HTA_MTICK(4);
    __syncthreads();
    HTA_MTICK(5);
    T Pd[NC], x0[NC], u3[NC], xr[NC];
    products<true, 1>(d, m, nullptr, Pd, x0, u3);
    HTA_MTICK(6);
    if (own) {
#pragma unroll
      for (int q = 0; q < NC; ++q) {
        upd_g[q] -= eh * Pd[q];
        upd_g_lds[q * CHS + row] = upd_g[q];
        xr[q] = x0[q];
        w0[q * CHS + row] = ev_r[q] * x0[q];
      }
    }
    HTA_MTICK(7);
    refine(x0, xr);
    if (own) {
#pragma unroll
      for (int q = 0; q < NC; ++q) upd_x[q] += eh * xr[q];
    }
    HTA_MTICK(3);
  }

  // ---- TRACK: the half steps without their P d and S m products (the schedule and its derivation: rmhmc_mfma4x4_kernel) ----
  // Up to two P products and two S products with ONE pass over the slices: op_i = P vp_i, os_i = S vs_i, complete in both
  // lanes of the row.  Per product the chunks and the order of the sum are those of products().
  template <int NPV, int NSV>
  __device__ __forceinline__ void products_multi(const T* vp0, const T* vp1, const T* vs0, const T* vs1, T (&op0)[NC], T (&op1)[NC],
                                                 T (&os0)[NC], T (&os1)[NC]) {
    constexpr int NV = NPV + NSV;
    const T* vec[4] = {vp0, vp1, vs0, vs1};
    T acc[NC][4][2];
#pragma unroll
    for (int q = 0; q < NC; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[q][i][0] = acc[q][i][1] = 0;
#pragma unroll
    for (int q = 0; q < NC; ++q) {
      if constexpr (sizeof(T) == 4) {
        constexpr int NB = (KH + 15) / 16;
        float u[4][NB];
        const int l15 = tid & 15;
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
          const int off = q * CHS + k0 + 16 * cb + ((16 * cb + 8 < KH) ? l15 : (l15 & 7));
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const bool used = i < 2 ? i < NPV : i - 2 < NSV;
            if (used) u[i][cb] = vec[i][off];
          }
        }
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const bool used = i < 2 ? i < NPV : i - 2 < NSV;
            if (!used) continue;
            const float* reg = reinterpret_cast<const float*>(i < 2 ? Preg : Sreg) + 16 * cb;
            if (16 * cb + 8 < KH) chunk_fma(reg, u[i][cb], reinterpret_cast<float(&)[2]>(acc[q][i]), std::make_integer_sequence<int, 16>{});
            else chunk_fma(reg, u[i][cb], reinterpret_cast<float(&)[2]>(acc[q][i]), std::make_integer_sequence<int, 8>{});
          }
        }
      } else {
        constexpr int NCH = KH / 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bool used = i < 2 ? i < NPV : i - 2 < NSV;
          if (!used) continue;
          V4 uu[NCH];
#pragma unroll
          for (int cb = 0; cb < NCH; ++cb) uu[cb] = *reinterpret_cast<const V4*>(vec[i] + q * CHS + k0 + 4 * cb);
#pragma unroll
          for (int cb = 0; cb < NCH; ++cb)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[q][i][e & 1] = fma((i < 2 ? Preg : Sreg)[4 * cb + e], uu[cb][e], acc[q][i][e & 1]);
        }
      }
    }
    (void)NV;
#pragma unroll
    for (int q = 0; q < NC; ++q) {
      if (NPV >= 1) op0[q] = row_total(acc[q][0][0] + acc[q][0][1]);
      if (NPV >= 2) op1[q] = row_total(acc[q][1][0] + acc[q][1][1]);
      if (NSV >= 1) os0[q] = row_total(acc[q][2][0] + acc[q][2][1]);
      if (NSV >= 2) os1[q] = row_total(acc[q][3][0] + acc[q][3][1]);
    }
  }

  // two independent solves x = (P + E)^-1 m from x_0 = S m, K phases (one barrier and one pass over S each); wa / wb return
  // e . x_(K-1) (zero without jitter).  wb4: this pair's four vectors a0 a1 b0 b1.
  __device__ __forceinline__ void solve2(T* wb4, const T (&ea)[NC], const T (&eb)[NC], const T (&x0a)[NC], const T (&x0b)[NC],
                                         T (&xa)[NC], T (&xb)[NC], T (&wa)[NC], T (&wb)[NC]) {
#pragma unroll
    for (int q = 0; q < NC; ++q) { xa[q] = x0a[q]; xb[q] = x0b[q]; wa[q] = 0; wb[q] = 0; }
    for (int it = 0; it < a.K; ++it) {
      T* A = wb4 + (it & 1) * 128;                         // read in this phase only; rewritten two phases later
      T* B = A + 2 * 128;
      if (own) {
#pragma unroll
        for (int q = 0; q < NC; ++q) {
          wa[q] = ea[q] * xa[q]; wb[q] = eb[q] * xb[q];
          A[q * CHS + row] = wa[q]; B[q * CHS + row] = wb[q];
        }
      }
      HTA_MTICK(4);
      __syncthreads();
      HTA_MTICK(5);
      T ra[NC], rb[NC], u0[NC], u1[NC];
      products_multi<0, 2>(nullptr, nullptr, A, B, u0, u1, ra, rb);
      HTA_MTICK(6);
#pragma unroll
      for (int q = 0; q < NC; ++q) { xa[q] = x0a[q] - ra[q]; xb[q] = x0b[q] - rb[q]; }
    }
  }

  // a pair of half steps (S:429-433 and, with the roles of the copies swapped, S:454-458):
  //   a:  g1 -= eh P (X1 - mu)   X2 += eh (P + E_a)^-1 g2        b:  g2 -= eh P (X2 - mu)   X1 += eh (P + E_b)^-1 g1
  // with y1 = P (X1 - mu), y2 = P (X2 - mu), z1 = S g1, z2 = S g2 kept current; the jitter of a and b: the next two slots
  __device__ __forceinline__ void pair_tracked(T* wb4, T (&X1)[NC], T (&X2)[NC], T (&g1)[NC], T (&g2)[NC], T (&y1)[NC], T (&y2)[NC],
                                               T (&z1)[NC], T (&z2)[NC], T eh) {
    T ea[NC], eb[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) {
      ea[q] = (a.has_jitter && own) ? jb[(jslot * NC + q) * 128 + row] : (T)0;
      eb[q] = (a.has_jitter && own) ? jb[((jslot + 1) * NC + q) * 128 + row] : (T)0;
      g1[q] -= eh * y1[q];                                  // a's momentum update ...
      z1[q] -= eh * (X1[q] - mu_r);                         // ... and S g1 with it (S P = I)
    }
    jslot += 2;
    T xa[NC], xb[NC], wa[NC], wb[NC];
    solve2(wb4, ea, eb, z2, z1, xa, xb, wa, wb);
#pragma unroll
    for (int q = 0; q < NC; ++q) {
      X2[q] += eh * xa[q];                                  // a's position update; P x_a = g2 - e_a . x_a(K-1)
      y2[q] += eh * (g2[q] - wa[q]);
      g2[q] -= eh * y2[q];                                  // b's momentum update
      z2[q] -= eh * (X2[q] - mu_r);
      X1[q] += eh * xb[q];                                  // b's position update
      y1[q] += eh * (g1[q] - wb[q]);
    }
    HTA_MTICK(7);
  }

  // the four tracked products of the current state, afresh (after the rotation phi_C): one barrier, one pass over P and S
  __device__ __forceinline__ void refresh_tracked() {
    if (own) {
#pragma unroll
      for (int q = 0; q < NC; ++q) {
        d1[q * CHS + row] = sth[q] - mu_r; dc[q * CHS + row] = sthc[q] - mu_r;
        pm[q * CHS + row] = spm[q]; pmc[q * CHS + row] = spmc[q];
      }
    }
    HTA_MTICK(4);
    __syncthreads();
    HTA_MTICK(5);
    products_multi<2, 2>(d1, dc, pm, pmc, y, yc, z, zc);
    HTA_MTICK(3);
  }

  __device__ __forceinline__ T factor() { return chol_in_lds<T>(a.P, ev, W, dg, D, ld, tid); }

  // H = -log p + D/2 log 2 pi + 1/2 log|G| + 1/2 m^T G^-1 m  (S:731) at (X, m) per chain, jitter sub-stream `sub`
  __device__ __forceinline__ void hamiltonian(uint32_t n, uint32_t sub, const T (&X)[NC], const T* m, const T (&mr)[NC], T (&H)[NC],
                                              T (&logp)[NC], T (&Pd_out)[NC], T (&Sm_out)[NC]) {
    T* d = (dpar && !TRACK) ? d1 : d0;                      // (TRACK: d1 belongs to the refresh phase)
    dpar ^= 1;
    T dr[NC];
    if (own) {
#pragma unroll
      for (int q = 0; q < NC; ++q) {
        const T e = jitter_elem(q, n, sub);
        ev_r[q] = e; ev[q * CHS + row] = e;
        dr[q] = X[q] - mu_r; d[q * CHS + row] = dr[q];
      }
    }
    const bool series = a.series || !a.has_jitter;
    T ld_exact = 0;
    if (NC == 1 && !series) ld_exact = factor();              // exact log|P + E| (its own barriers; reads ev from LDS)
    __syncthreads();
    T Pd[NC], x0[NC], s2[NC], xr[NC];
    if (series && a.has_jitter) products<true, 2>(d, m, ev, Pd, x0, s2);
    else products<true, 1>(d, m, nullptr, Pd, x0, s2);
    T v[NC][4];
#pragma unroll
    for (int q = 0; q < NC; ++q) { v[q][0] = v[q][1] = v[q][2] = v[q][3] = 0; xr[q] = x0[q]; Pd_out[q] = Pd[q]; Sm_out[q] = x0[q]; }
    if (own) {
#pragma unroll
      for (int q = 0; q < NC; ++q) {
        v[q][0] = dr[q] * Pd[q];
        w0[q * CHS + row] = ev_r[q] * x0[q];
        // log|P + E| = log|P| + tr(SE) - 1/2 tr((SE)^2) + O(D rho^3 / 3)
        if (series && a.has_jitter) v[q][2] = ev_r[q] * (sd_r - (T)0.5 * s2[q]);
      }
    }
    if (NC == 1 && !series) v[0][2] = ld_exact;
    refine(x0, xr);
    if (own) {
#pragma unroll
      for (int q = 0; q < NC; ++q) v[q][1] = mr[q] * xr[q];
    }
    block_sums(v);
    const float pi_term = (float)D * 1.8378770351409912f;     // S:712: float32 whatever the state dtype
#pragma unroll
    for (int q = 0; q < NC; ++q) {
      const T lp = a.log_norm - (T)0.5 * v[q][0];
      const T logdet = series ? a.logdetP + v[q][2] : v[q][2];
      logp[q] = lp;
      H[q] = -lp + (T)0.5 * (T)pi_term + (T)0.5 * logdet + (T)0.5 * v[q][1];
    }
  }
};

template <typename T, int KH, int NC, bool TRACK = false>
__device__ __forceinline__ void fused_body(const FusedArgs<T>& a, int ld, int need_w) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  typedef Fused<T, KH, NC, TRACK> F;
  F ch(a);
  const int D = a.D, tid = threadIdx.x;
  ch.D = D; ch.ld = ld; ch.tid = tid; ch.dpar = 0;
  {
    const int wave = tid >> 6, lane = tid & 63;
    ch.row = 32 * wave + (lane & 31);
    ch.k0 = (lane >> 5) ? KH : 0;
    ch.own = (lane >= 32) && ch.row < D;                    // the upper lane of a row holds the complete sums
  }
  T* v = reinterpret_cast<T*>(smem_raw);
  T** slots[9] = {&ch.pm, &ch.pmc, &ch.ev, &ch.d0, &ch.d1, &ch.w0, &ch.w1, &ch.dc, &ch.ws};    // ws: 8 vectors
  for (int i = 0; i < 9; ++i) *slots[i] = v + i * 128;
  ch.dg = v + NC * F::CHS;
  ch.red = ch.dg + 128;
  ch.jb = ch.red + 32;                                      // [8 evaluation slots][128]
  ch.W = ch.jb + 8 * 128;                                   // only with need_w
  for (int e = tid; e < NC * F::CHS + 128; e += FNT) v[e] = (T)0;
  const int row = ch.row;
  ch.mu_r = row < D ? a.mu[row] : (T)0;
  ch.sd_r = row < D ? a.S[(int64_t)row * D + row] : (T)0;
  const T eh = (T)0.5 * a.eps;
  ch.load_slices();
#if HTA_RM_TIMING
  __syncthreads();
  ch.tlast = __builtin_readcyclecounter();
#endif
  bool have_factor = false;                                 // without jitter chol(P) serves every chain and trajectory
  const int64_t ngroup = (a.C + NC - 1) / NC;
  for (int64_t cg = blockIdx.x; cg < ngroup; cg += gridDim.x) {
    int64_t c[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) {
      c[q] = NC * cg + q;
      ch.live[q] = c[q] < a.C;                               // an odd chain count leaves the last pair half empty
      ch.chain[q] = a.chain_offset + (uint64_t)(ch.live[q] ? c[q] : 0);
    }
#pragma unroll
    for (int q = 0; q < NC; ++q) ch.scur[q] = (ch.own && ch.live[q]) ? a.cur[c[q] * D + row] : (T)0;
    int32_t rejected[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) rejected[q] = 0;
    for (int t = 0; t < a.n_traj; ++t) {
      const uint32_t n = (uint32_t)(a.traj_offset + t);
      HTA_RTICK(0);
      // ---- gibbs: p = chol(G(theta)) z  (S:183-184), jitter sub-stream 0
      if (a.p_ws) {
        if (ch.own) {
#pragma unroll
          for (int q = 0; q < NC; ++q) {
            ch.spm[q] = ch.live[q] ? a.p_ws[((int64_t)t * a.C + c[q]) * D + row] : (T)0;
            ch.pm[q * F::CHS + row] = ch.spm[q];
          }
        }
      } else if (NC == 1) {                                   // no pre-drawn momenta: factor here (tid-indexed helper phases)
        __syncthreads();
        if (tid < D) ch.ev[tid] = a.has_jitter ? a.jitter * uniform_elem<T>(a.seed, ch.chain[0], n, PURPOSE_JITTER, 0, tid) : (T)0;
        if (a.has_jitter || !have_factor) { ch.factor(); have_factor = true; }
        if (tid < D) ch.w1[tid] = normal_elem<T>(a.seed, ch.chain[0], n, 0, tid);
        __syncthreads();
        if (tid < D) {
          T acc0 = ch.dg[tid] * ch.w1[tid], acc1 = 0;
          const T* rowp = ch.W + tid * ld;
          int k = 0;
          for (; k + 1 < tid; k += 2) { acc0 = fma(rowp[k], ch.w1[k], acc0); acc1 = fma(rowp[k + 1], ch.w1[k + 1], acc1); }
          if (k < tid) acc0 = fma(rowp[k], ch.w1[k], acc0);
          ch.pm[tid] = acc0 + acc1;
        }
        __syncthreads();
        if (ch.own) ch.spm[0] = ch.pm[row];
      }
      // ---- H_old (S:971 -> S:822), sub-stream 1
      T H0[NC], H1[NC], lp0[NC], lp1[NC];
      ch.hamiltonian(n, 1, ch.scur, ch.pm, ch.spm, H0, lp0, ch.y, ch.z);
#pragma unroll
      for (int q = 0; q < NC; ++q) {                                          // S:425-426
        ch.sth[q] = ch.scur[q]; ch.sthc[q] = ch.scur[q]; ch.spmc[q] = ch.spm[q];
        ch.yc[q] = ch.y[q]; ch.zc[q] = ch.z[q];
        if (!TRACK && ch.own) ch.pmc[q * F::CHS + row] = ch.spm[q];
      }
      HTA_RTICK(1);
      // ---- L explicit steps (S:427-461)
      for (int l = 0; l < a.L; ++l) {
        if (l == 0 || ch.jslot == F::NSLOT) ch.refill_jitter(n, l);         // sub-streams 2 + 8 l + {1, 2, 4, 7}, ...
        if (TRACK) ch.pair_tracked(ch.ws, ch.sth, ch.sthc, ch.spm, ch.spmc, ch.y, ch.yc, ch.z, ch.zc, eh);   // phi_A/2, phi_B/2  S:429-433
        else {
          ch.half_step(ch.sth, ch.pmc, ch.sthc, ch.spm, ch.pm, eh);           // phi_A/2  S:429-430
          ch.half_step(ch.sthc, ch.pm, ch.sth, ch.spmc, ch.pmc, eh);          // phi_B/2  S:432-433
        }
        if (a.K == 0) __syncthreads();                                        // slower waves may still stream pm (no refinement barrier)
#pragma unroll
        for (int q = 0; q < NC; ++q) {                                        // phi_C    S:447-450, sequential (Q1)
          T xx = ch.sth[q], b = ch.spm[q], xc = ch.sthc[q], bc = ch.spmc[q];
          const T h = (T)0.5, cc = a.rot_c, ss = a.rot_s;
          xx = h * ((xx + xc) + cc * (xx - xc) + ss * (b - bc));
          b = h * ((b + bc) - ss * (xx - xc) + cc * (b - bc));
          xc = h * ((xx + xc) - cc * (xx - xc) - ss * (b - bc));
          bc = h * ((b + bc) + ss * (xx - xc) - cc * (b - bc));
          ch.sth[q] = xx; ch.spm[q] = b; ch.sthc[q] = xc; ch.spmc[q] = bc;
          if (!TRACK && ch.own) { ch.pm[q * F::CHS + row] = b; ch.pmc[q * F::CHS + row] = bc; }
        }
        if (TRACK) {
          ch.refresh_tracked();
          ch.pair_tracked(ch.ws + 4 * 128, ch.sthc, ch.sth, ch.spmc, ch.spm, ch.yc, ch.y, ch.zc, ch.z, eh);   // phi_B/2, phi_A/2  S:454-458
        } else {
          ch.half_step(ch.sthc, ch.pm, ch.sth, ch.spmc, ch.pmc, eh);          // phi_B/2  S:454-455
          ch.half_step(ch.sth, ch.pmc, ch.sthc, ch.spm, ch.pm, eh);           // phi_A/2  S:457-458
        }
      }
      HTA_RTICK(2);
      if (TRACK) {                                                            // (the tracked half steps keep the momenta in registers)
        if (a.K == 0) __syncthreads();                                        // no solve phase since the refresh phase read pm
        if (ch.own) {
#pragma unroll
          for (int q = 0; q < NC; ++q) ch.pm[q * F::CHS + row] = ch.spm[q];
        }
      }
      // ---- H_new on the un-augmented pair (S:989, Q4), sub-stream 2 + 8L
      T unused1[NC], unused2[NC];
      ch.hamiltonian(n, 2u + 8u * (uint32_t)a.L, ch.sth, ch.pm, ch.spm, H1, lp1, unused1, unused2);
      HTA_RTICK(3);
      // ---- Metropolis test + bookkeeping (S:1000-1026, S:1045-1057), as hmc_pieces.hip:mh_select_kernel
#pragma unroll
      for (int q = 0; q < NC; ++q) {
        const T u = u23<T>(philox_block(a.seed, ch.chain[q], n, PURPOSE_MH, 0, 0).x);
        const bool acc = mh_accept<T>(H0[q], H1[q], lp1[q], u);
        const bool reset = (!acc) && ((int)n == a.burn + 1);                  // Q2
        if (ch.own && ch.live[q]) {
          const T vnew = acc ? ch.sth[q] : (reset ? a.theta_init[c[q] * D + row] : ch.scur[q]);
          ch.scur[q] = vnew;
          if (a.samples && (int)n > a.burn) a.samples[((int64_t)((int)n - a.burn) * a.C + c[q]) * D + row] = vnew;
        }
        if (!acc) ++rejected[q];
        if (tid == 0 && ch.live[q]) {
          if (a.H_old) a.H_old[(int64_t)t * a.C + c[q]] = H0[q];
          if (a.H_new) a.H_new[(int64_t)t * a.C + c[q]] = H1[q];
          if (a.accept) a.accept[(int64_t)t * a.C + c[q]] = acc ? 1 : 0;
        }
      }
    }
    if (ch.own) {
#pragma unroll
      for (int q = 0; q < NC; ++q) if (ch.live[q]) a.cur[c[q] * D + row] = ch.scur[q];
    }
    if (tid == 0) {
#pragma unroll
      for (int q = 0; q < NC; ++q) if (ch.live[q]) a.reject_count[c[q]] += rejected[q];
    }
#if HTA_RM_TIMING
    if (tid == 0 && blockIdx.x == 0) for (int k = 0; k < 8; ++k) hta_rm_dbg[k] = ch.tacc[k];
#endif
  }
}

template <typename T, int KH, int NC, bool TRACK = false>
__global__ __launch_bounds__(FNT, 2) void rmhmc_fused_kernel(FusedArgs<T> a, int ld, int need_w) {
  fused_body<T, KH, NC, TRACK>(a, ld, need_w);
}

// The same kernel for launches of at most one workgroup per CU (chains <= compute units: BASELINE config 3's 256 chains):
// two workgroups per CU cannot be resident then, so the two-workgroup register cap of 256 - under which the KH = 56 / 64
// instances keep 26 / 35 registers in scratch - buys nothing.  HTA_FUSED_WIDE_VGPRS registers instead: together with the
// overlapped momentum waves (rmhmc_momentum_wave_kernel, capped at 512 - HTA_FUSED_WIDE_VGPRS) a SIMD's 512 are exactly used.
#define HTA_FUSED_WIDE_VGPRS 288
template <int KH, bool TRACK = false>
__global__ __launch_bounds__(FNT) __attribute__((amdgpu_num_vgpr(HTA_FUSED_WIDE_VGPRS))) void rmhmc_fused_kernel_wide(FusedArgs<float> a, int ld, int need_w) {
  fused_body<float, KH, 1, TRACK>(a, ld, need_w);
}


// =============================================================================================
// The same run, 16 chains per workgroup, products on the matrix cores (fp32, D <= 112, pre-drawn momenta, log-det series)
//
// S and P are the same for every chain, so a product pass over 16 chains is S X with X a D x 16 matrix:
// v_mfma_f32_16x16x4_f32 tiles (exact fp32 products, fp32 accumulation; 2048 flop per instruction where the one-chain
// kernel issues one v_fmac_f32_dpp per 128).  Wave w owns the row tile 16w..16w+15: its A fragments (S and P, 28 VGPRs
// each: lane (i, g) holds A[16w+i][28g+j], j < 28 - the K index is permuted so that a lane's share of an operand vector is
// contiguous in LDS) stay in registers for the launch; the B operand of chain column n is X[n][28g+j], seven 16-byte
// LDS reads per vector and pass; the accumulator is C[4g+reg][n]: lane (n, g) OWNS elements 16w+4g .. +3 of chain n's vectors
// and does their element-wise work in registers (four consecutive rows = one Philox block of jitter per evaluation).
// Same streams, same update order, same barriers per pass as rmhmc_fused_kernel; sums run in a different order.
// What it buys is bounded by the fp32 matrix rate (64 flop/clk/SIMD, twice the VALU FMA rate, minus 25 % tile padding at
// D = 100) and by its pass latency (7 row tiles on 4 SIMDs: 2 x 56 MFMAs x 32 clk per first pass): measured against one chain
// per workgroup 0.6x / 0.96x / 1.19x / 1.37x at 512 / 1024 / 2048 / 4096 chains (tools/scratch/pair_check.py), so it is taken
// from 2048 chains per GPU on (hta_set_tuning("rmhmc_batch", 0) off, 2 always); there the per-trajectory Cholesky of the
// momentum draw is the next limit.  TRACK ("rmhmc_pair" = 1, the default): the tracked-products schedule derived at
// rmhmc_mfma4x4_kernel (4096 chains: 1.56e8 -> 1.84e8 steps/s, 3072: 1.24e8 -> 1.47e8).
// =============================================================================================
constexpr int BNC = 16, BLD = 116, BWV = 7, BNT = 64 * BWV, BSEG = 28, BBUF = 12;

// BKJ: MFMAs per product and row tile = K / 4 (25 for D <= 100, 28 up to 112).  A vector is stored in four segments of BSEG
// floats (16-byte aligned); element `row` sits in segment row / BKJ at offset row % BKJ.
template <int BKJ, bool TRACK>
__global__ __launch_bounds__(BNT) void rmhmc_batch_kernel(FusedArgs<float> a) {
  typedef float T;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* lds = reinterpret_cast<T*>(smem_raw);
  constexpr int MSZ = BNC * BLD;
  T* PM = lds; T* PMC = PM + MSZ; T* D1 = PMC + MSZ;
  T* DC = D1 + MSZ;                                       // tracked products: theta_c - mu (D1: theta - mu)
  T* D0 = DC + MSZ; T* W0 = D0 + MSZ; T* W1 = W0 + MSZ; T* EV = W1 + MSZ;
  // tracked products: 2 pairs x 2 solves x 2 refinement vectors.  The first pair's four ARE the Hamiltonians' D0 W0 W1 EV
  // (contiguous): a Hamiltonian and the first pair of a step are always separated by barriers (block_sums / the solve phases of
  // the second pair), and the whole set stays at 12 matrices = 88 KB
  T* WS2 = EV + MSZ;
  T* red = WS2 + 4 * MSZ;                                 // [BWV][BNC][4]
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, cl = l & 15, g = l >> 4;
  const int D = a.D;
  const int row0 = 16 * w + 4 * g, arow = 16 * w + cl;
  T Sa[BKJ], Pa[BKJ];
#pragma unroll
  for (int j = 0; j < BKJ; ++j) {
    const int k = BKJ * g + j;
    const bool ok = arow < D && k < D;
    Sa[j] = ok ? a.S[(int64_t)k * D + arow] : 0.f;        // symmetric: column arow, coalesced over the lanes of a tile
    Pa[j] = ok ? a.P[(int64_t)k * D + arow] : 0.f;
  }
  T mu_r[4], sd_r[4];
  bool rok[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int r = row0 + e;
    rok[e] = r < D;
    mu_r[e] = rok[e] ? a.mu[r] : 0.f;
    sd_r[e] = rok[e] ? a.S[(int64_t)r * D + r] : 0.f;
  }
  for (int e = tid; e < BBUF * MSZ + BWV * BNC * 4; e += BNT) lds[e] = 0.f;
  const T eh = 0.5f * a.eps;
  int own_pos[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) own_pos[e] = cl * BLD + ((row0 + e) / BKJ) * BSEG + (row0 + e) % BKJ;
  const int b_off = cl * BLD + BSEG * g;
  int dpar = 0;
  T ev_r[4] = {0.f, 0.f, 0.f, 0.f};
  uint64_t chain = 0;
  bool live = false;

  auto put4 = [&](T* X, const T (&v)[4]) {
    if constexpr (BKJ == BSEG) *reinterpret_cast<bf4*>(X + own_pos[0]) = bf4{v[0], v[1], v[2], v[3]};
    else {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (rok[e]) X[own_pos[e]] = v[e];     // rows >= D have no slot in this layout (and stay 0)
    }
  };
  auto load_b = [&](const T* X, T (&b)[BKJ]) {
#pragma unroll
    for (int q = 0; q < BKJ / 4; ++q) {
      const bf4 v = *reinterpret_cast<const bf4*>(X + b_off + 4 * q);
      b[4 * q] = v[0]; b[4 * q + 1] = v[1]; b[4 * q + 2] = v[2]; b[4 * q + 3] = v[3];
    }
#pragma unroll
    for (int j = (BKJ / 4) * 4; j < BKJ; ++j) b[j] = X[b_off + j];
  };
  auto jitter4 = [&](uint32_t n, uint32_t sub) {            // this lane's four rows are one Philox block (uniform_elem layout)
    if (!a.has_jitter) return;
    const U4 r = philox_block(a.seed, chain, n, PURPOSE_JITTER, sub, (uint32_t)(row0 >> 2));
    const T u[4] = {u23<T>(r.x), u23<T>(r.y), u23<T>(r.z), u23<T>(r.w)};
#pragma unroll
    for (int e = 0; e < 4; ++e) ev_r[e] = (live && rok[e]) ? a.jitter * u[e] : 0.f;
  };
  // x = (P + diag(e))^-1 m continued from x0 = S m (rmhmc_fused_kernel: refine)
  auto refine = [&](const T (&x0)[4], T (&xr)[4]) {
    for (int it = 0; it < a.K; ++it) {
      const T* wr = (it & 1) ? W1 : W0;
      T* ww = (it & 1) ? W0 : W1;
      __syncthreads();
      T bw[BKJ];
      load_b(wr, bw);
      bf4 sx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < BKJ; ++j) sx = __builtin_amdgcn_mfma_f32_16x16x4f32(Sa[j], bw[j], sx, 0, 0, 0);
      T wv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) { xr[e] = x0[e] - sx[e]; wv[e] = ev_r[e] * xr[e]; }
      put4(ww, wv);
    }
  };
  // one half step: upd_x += eh G(X)^-1 m ; upd_g -= eh P (X - mu)   (rmhmc_fused_kernel: half_step)
  auto half_step = [&](uint32_t n, uint32_t sub, const T (&X)[4], const T* m, T (&upd_x)[4], T (&upd_g)[4], T* upd_g_lds) {
    T* d = dpar ? D1 : D0;
    dpar ^= 1;
    T dv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) dv[e] = rok[e] ? X[e] - mu_r[e] : 0.f;
    put4(d, dv);
    __syncthreads();
    T bd[BKJ], bm[BKJ];
    load_b(d, bd);
    load_b(m, bm);
    bf4 Pd = {0.f, 0.f, 0.f, 0.f}, x0v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < BKJ; ++j) {
      Pd = __builtin_amdgcn_mfma_f32_16x16x4f32(Pa[j], bd[j], Pd, 0, 0, 0);
      x0v = __builtin_amdgcn_mfma_f32_16x16x4f32(Sa[j], bm[j], x0v, 0, 0, 0);
    }
    jitter4(n, sub);            // first needed after the products: its Philox rounds issue under the MFMAs
    T x0[4], xr[4], wv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      upd_g[e] -= eh * Pd[e];
      x0[e] = x0v[e]; xr[e] = x0v[e];
      wv[e] = ev_r[e] * x0v[e];
    }
    put4(upd_g_lds, upd_g);
    put4(W0, wv);
    refine(x0, xr);
#pragma unroll
    for (int e = 0; e < 4; ++e) upd_x[e] += eh * xr[e];
  };
  // ---- TRACK: the tracked-products schedule of rmhmc_mfma4x4_kernel (derivation there) in this kernel's layout ----------
  T y[4], yc[4], z[4], zc[4];
  // two independent solves x = (P + E)^-1 m from x_0 = S m, K phases; wa / wb return e . x_(K-1) (zero without jitter)
  auto solve2 = [&](T* WB, const T (&ea)[4], const T (&eb)[4], const T (&x0a)[4], const T (&x0b)[4], T (&xa)[4], T (&xb)[4],
                    T (&wa)[4], T (&wb)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { xa[e] = x0a[e]; xb[e] = x0b[e]; wa[e] = 0.f; wb[e] = 0.f; }
    for (int it = 0; it < a.K; ++it) {
      T* A = WB + (it & 1) * MSZ;                           // read in this phase only; rewritten two phases later
      T* B = A + 2 * MSZ;
#pragma unroll
      for (int e = 0; e < 4; ++e) { wa[e] = ea[e] * xa[e]; wb[e] = eb[e] * xb[e]; }
      put4(A, wa);
      put4(B, wb);
      __syncthreads();
      T ba[BKJ], bb[BKJ];
      load_b(A, ba);
      load_b(B, bb);
      bf4 ra = {0.f, 0.f, 0.f, 0.f}, rb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < BKJ; ++j) {
        ra = __builtin_amdgcn_mfma_f32_16x16x4f32(Sa[j], ba[j], ra, 0, 0, 0);
        rb = __builtin_amdgcn_mfma_f32_16x16x4f32(Sa[j], bb[j], rb, 0, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) { xa[e] = x0a[e] - ra[e]; xb[e] = x0b[e] - rb[e]; }
    }
  };
  // a pair of half steps (S:429-433 and, with the roles of the copies swapped, S:454-458):
  //   a:  g1 -= eh P (X1 - mu)   X2 += eh (P + E_a)^-1 g2        b:  g2 -= eh P (X2 - mu)   X1 += eh (P + E_b)^-1 g1
  // with y1 = P (X1 - mu), y2 = P (X2 - mu), z1 = S g1, z2 = S g2 kept current
  auto pair_tracked = [&](T* WB, uint32_t n, uint32_t suba, uint32_t subb, T (&X1)[4], T (&X2)[4], T (&g1)[4], T (&g2)[4],
                          T (&y1)[4], T (&y2)[4], T (&z1)[4], T (&z2)[4]) {
    T ea[4], eb[4];
    jitter4(n, suba);
#pragma unroll
    for (int e = 0; e < 4; ++e) ea[e] = ev_r[e];
    jitter4(n, subb);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      eb[e] = ev_r[e];
      g1[e] -= eh * y1[e];                                  // a's momentum update ...
      z1[e] -= eh * (rok[e] ? X1[e] - mu_r[e] : 0.f);       // ... and S g1 with it
    }
    T xa[4], xb[4], wa[4], wb[4];
    solve2(WB, ea, eb, z2, z1, xa, xb, wa, wb);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      X2[e] += eh * xa[e];                                  // a's position update; P x_a = g2 - e_a . x_a(K-1)
      y2[e] += eh * (g2[e] - wa[e]);
      g2[e] -= eh * y2[e];                                  // b's momentum update
      z2[e] -= eh * (rok[e] ? X2[e] - mu_r[e] : 0.f);
      X1[e] += eh * xb[e];                                  // b's position update
      y1[e] += eh * (g1[e] - wb[e]);
    }
  };
  // y, y_c, z, z_c of the current state, afresh: two passes of two products (four operand vectors at once do not fit the registers)
  auto refresh_tracked = [&](const T (&th)[4], const T (&thc)[4], const T (&pm_)[4], const T (&pmc_)[4]) {
    T dt[4], dc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { dt[e] = rok[e] ? th[e] - mu_r[e] : 0.f; dc[e] = rok[e] ? thc[e] - mu_r[e] : 0.f; }
    put4(D1, dt);
    put4(DC, dc);
    put4(PM, pm_);
    put4(PMC, pmc_);
    __syncthreads();
    {
      T b1[BKJ], b2[BKJ];
      load_b(D1, b1);
      load_b(DC, b2);
      bf4 p1 = {0.f, 0.f, 0.f, 0.f}, p2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < BKJ; ++j) {
        p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(Pa[j], b1[j], p1, 0, 0, 0);
        p2 = __builtin_amdgcn_mfma_f32_16x16x4f32(Pa[j], b2[j], p2, 0, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) { y[e] = p1[e]; yc[e] = p2[e]; }
    }
    {
      T b1[BKJ], b2[BKJ];
      load_b(PM, b1);
      load_b(PMC, b2);
      bf4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < BKJ; ++j) {
        s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(Sa[j], b1[j], s1, 0, 0, 0);
        s2 = __builtin_amdgcn_mfma_f32_16x16x4f32(Sa[j], b2[j], s2, 0, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) { z[e] = s1[e]; zc[e] = s2[e]; }
    }
  };
  // three sums per chain over the rows, complete in every lane of the chain's column
  auto block_sums = [&](T (&v)[3]) {
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      v[e] += __shfl_xor(v[e], 16, 64);
      v[e] += __shfl_xor(v[e], 32, 64);
    }
    __syncthreads();
    if (g == 0) {
#pragma unroll
      for (int e = 0; e < 3; ++e) red[(w * BNC + cl) * 4 + e] = v[e];
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      T s = 0.f;
#pragma unroll
      for (int i = 0; i < BWV; ++i) s += red[(i * BNC + cl) * 4 + e];
      v[e] = s;
    }
  };
  // H = -log p + D/2 log 2 pi + 1/2 log|G| + 1/2 m^T G^-1 m  (S:731)   (rmhmc_fused_kernel: hamiltonian, series branch)
  auto hamiltonian = [&](uint32_t n, uint32_t sub, const T (&X)[4], const T* m, const T (&mr)[4], T& H, T& logp, T (&Pd_out)[4],
                         T (&Sm_out)[4]) {
    T* d = (dpar && !TRACK) ? D1 : D0;                      // (TRACK: D1 belongs to the refresh phase)
    dpar ^= 1;
    jitter4(n, sub);
    T dr[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) dr[e] = rok[e] ? X[e] - mu_r[e] : 0.f;
    put4(EV, ev_r);
    put4(d, dr);
    __syncthreads();
    T bd[BKJ], bm[BKJ];
    load_b(d, bd);
    load_b(m, bm);
    bf4 Pd = {0.f, 0.f, 0.f, 0.f}, x0v = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < BKJ; ++j) {
      Pd = __builtin_amdgcn_mfma_f32_16x16x4f32(Pa[j], bd[j], Pd, 0, 0, 0);
      x0v = __builtin_amdgcn_mfma_f32_16x16x4f32(Sa[j], bm[j], x0v, 0, 0, 0);
    }
    if (a.has_jitter) {                                     // second-order log-det term: (S . S) e
      T be[BKJ];
      load_b(EV, be);
#pragma unroll
      for (int j = 0; j < BKJ; ++j) s2 = __builtin_amdgcn_mfma_f32_16x16x4f32(Sa[j] * Sa[j], be[j], s2, 0, 0, 0);
    }
    T v[3] = {0.f, 0.f, 0.f}, x0[4], xr[4], wv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      x0[e] = x0v[e]; xr[e] = x0v[e];
      Pd_out[e] = Pd[e]; Sm_out[e] = x0v[e];
      v[0] += dr[e] * Pd[e];
      wv[e] = ev_r[e] * x0v[e];
      if (a.has_jitter) v[2] += ev_r[e] * (sd_r[e] - 0.5f * s2[e]);      // log|P + E| = log|P| + tr(SE) - 1/2 tr((SE)^2) + ...
    }
    put4(W0, wv);
    refine(x0, xr);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[1] += mr[e] * xr[e];
    block_sums(v);
    const float pi_term = (float)D * 1.8378770351409912f;   // S:712
    logp = a.log_norm - 0.5f * v[0];
    H = -logp + 0.5f * pi_term + 0.5f * (a.logdetP + v[2]) + 0.5f * v[1];
  };

  const int64_t ngroup = (a.C + BNC - 1) / BNC;
  for (int64_t cg = blockIdx.x; cg < ngroup; cg += gridDim.x) {
    const int64_t c = BNC * cg + cl;
    live = c < a.C;
    chain = a.chain_offset + (uint64_t)(live ? c : 0);
    T scur[4], sth[4], spm[4], sthc[4], spmc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) scur[e] = (live && rok[e]) ? a.cur[c * D + row0 + e] : 0.f;
    int32_t rejected = 0;
    __syncthreads();                                        // the previous group's last reads of the vector matrices
    for (int t = 0; t < a.n_traj; ++t) {
      const uint32_t n = (uint32_t)(a.traj_offset + t);
      // ---- gibbs: p = chol(G(theta)) z, drawn ahead by rmhmc_momentum_kernel (S:183-184)
#pragma unroll
      for (int e = 0; e < 4; ++e) spm[e] = (live && rok[e]) ? a.p_ws[((int64_t)t * a.C + c) * D + row0 + e] : 0.f;
      put4(PM, spm);
      T H0, H1, lp0, lp1;
      hamiltonian(n, 1, scur, PM, spm, H0, lp0, y, z);      // S:971 -> S:822
#pragma unroll
      for (int e = 0; e < 4; ++e) { sth[e] = scur[e]; sthc[e] = scur[e]; spmc[e] = spm[e]; yc[e] = y[e]; zc[e] = z[e]; }   // S:425-426
      if (!TRACK) put4(PMC, spm);
      for (int lstep = 0; lstep < a.L; ++lstep) {           // S:427-461
        const uint32_t k0 = 2u + 8u * (uint32_t)lstep;
        if (TRACK) pair_tracked(D0, n, k0 + 1, k0 + 2, sth, sthc, spm, spmc, y, yc, z, zc);    // phi_A/2, phi_B/2  S:429-433
        else {
          half_step(n, k0 + 1, sth, PMC, sthc, spm, PM);    // phi_A/2  S:429-430
          half_step(n, k0 + 2, sthc, PM, sth, spmc, PMC);   // phi_B/2  S:432-433
        }
        if (a.K == 0) __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) {                       // phi_C    S:447-450, sequential (Q1)
          T xx = sth[e], b = spm[e], xc = sthc[e], bc = spmc[e];
          const T h = 0.5f, cc = a.rot_c, ss = a.rot_s;
          xx = h * ((xx + xc) + cc * (xx - xc) + ss * (b - bc));
          b = h * ((b + bc) - ss * (xx - xc) + cc * (b - bc));
          xc = h * ((xx + xc) - cc * (xx - xc) - ss * (b - bc));
          bc = h * ((b + bc) + ss * (xx - xc) - cc * (b - bc));
          sth[e] = xx; spm[e] = b; sthc[e] = xc; spmc[e] = bc;
        }
        if (TRACK) {
          refresh_tracked(sth, sthc, spm, spmc);
          pair_tracked(WS2, n, k0 + 4, k0 + 7, sthc, sth, spmc, spm, yc, y, zc, z);   // phi_B/2, phi_A/2  S:454-458
        } else {
          put4(PM, spm);
          put4(PMC, spmc);
          half_step(n, k0 + 4, sthc, PM, sth, spmc, PMC);   // phi_B/2  S:454-455
          half_step(n, k0 + 7, sth, PMC, sthc, spm, PM);    // phi_A/2  S:457-458
        }
      }
      if (TRACK) {                                          // (the tracked half steps keep the momenta in registers)
        if (a.K == 0) __syncthreads();                      // no solve phase since the refresh phase read PM
        put4(PM, spm);
      }
      T unused1[4], unused2[4];
      hamiltonian(n, 2u + 8u * (uint32_t)a.L, sth, PM, spm, H1, lp1, unused1, unused2);   // S:989 (Q4)
      // ---- Metropolis test + bookkeeping (S:1000-1026, S:1045-1057)
      const T u = u23<T>(philox_block(a.seed, chain, n, PURPOSE_MH, 0, 0).x);
      const bool acc = mh_accept<T>(H0, H1, lp1, u);
      const bool reset = (!acc) && ((int)n == a.burn + 1);  // Q2
      if (live) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (rok[e]) {
            const T vnew = acc ? sth[e] : (reset ? a.theta_init[c * D + row0 + e] : scur[e]);
            scur[e] = vnew;
            if (a.samples && (int)n > a.burn) a.samples[((int64_t)((int)n - a.burn) * a.C + c) * D + row0 + e] = vnew;
          }
        }
        if (w == 0 && g == 0) {
          if (a.H_old) a.H_old[(int64_t)t * a.C + c] = H0;
          if (a.H_new) a.H_new[(int64_t)t * a.C + c] = H1;
          if (a.accept) a.accept[(int64_t)t * a.C + c] = acc ? 1 : 0;
        }
      }
      if (!acc) ++rejected;
    }
    if (live) {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (rok[e]) a.cur[c * D + row0 + e] = scur[e];
      if (w == 0 && g == 0) a.reject_count[c] += rejected;
    }
  }
}

// =============================================================================================
// The same run, FOUR chains per workgroup: v_mfma_f32_4x4x1_16b_f32 (16 blocks of 4 x 4, K = 1; fp32, D <= 100)
//
// rmhmc_batch_kernel needs 16 chains per workgroup (the N of its 16 x 16 tile), i.e. C / 16 workgroups: at 1024 chains it
// keeps 64 of the 256 CUs busy.  The 16-block form has N = 4: block b multiplies rows 4b .. 4b+3 of A with the SAME 4
// columns, so one wave covers 64 rows x 4 chains per instruction and two waves a whole product - C / 4 workgroups of two
// waves, one wave per SIMD, the matrix pipes of every CU in use from 512 chains on.  Same flop rate per instruction
// (64 flop / clk / SIMD), same ownership: lane (b, n) = (l >> 2, l & 3) supplies A[64w + l][k] (S and P: 2 x 100 VGPRs for
// the launch) and X[n][k] (LDS, 16-byte reads, the same address for the 16 lanes of a chain) and receives
// C[64w + 4b + r][n], r < 4: four consecutive rows of chain n = one Philox block of jitter, element-wise work in registers.
// Same streams, same update order and barriers as rmhmc_batch_kernel; a product's sum runs over k = 0 .. D-1 in order.
// TRACK ("rmhmc_pair" = 1, the default): the tracked-products schedule derived at rmhmc_mfma4x4_kernel - 12 products in 5 phases
// per step at K = 2 instead of 16 in 8 (2048 chains: 1.36e8 -> 1.59e8 steps/s, 1536: 1.10e8 -> 1.29e8).  This kernel serves
// 1025 .. 2048 chains per GPU, where two of its workgroups fill a CU's four SIMDs.
// =============================================================================================
constexpr int QNC = 4, QLD = 116, QWV = 2, QNT = 64 * QWV, QK = 100, QBUF = 16;

template <bool TRACK>
__global__ __launch_bounds__(QNT) void rmhmc_mfma4_kernel(FusedArgs<float> a) {
  typedef float T;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* lds = reinterpret_cast<T*>(smem_raw);
  constexpr int MSZ = QNC * QLD;
  T* PM = lds; T* PMC = PM + MSZ; T* D0 = PMC + MSZ; T* D1 = D0 + MSZ; T* W0 = D1 + MSZ; T* W1 = W0 + MSZ; T* EV = W1 + MSZ;
  T* DC = EV + MSZ;                                       // tracked products: theta_c - mu (D1: theta - mu)
  T* WS = DC + MSZ;                                       // tracked products: 2 pairs x 2 solves x 2 refinement vectors
  T* red = WS + 8 * MSZ;                                  // [QWV][QNC][4]
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, cl = l & 3, blk = l >> 2;
  const int D = a.D;
  const int row0 = 64 * w + 4 * blk, arow = 64 * w + l;
  T Sa[QK], Pa[QK];
#pragma unroll
  for (int k = 0; k < QK; ++k) {
    const bool ok = arow < D && k < D;
    Sa[k] = ok ? a.S[(int64_t)k * D + arow] : 0.f;        // symmetric: column arow, coalesced over the lanes
    Pa[k] = ok ? a.P[(int64_t)k * D + arow] : 0.f;
  }
  T mu_r[4], sd_r[4];
  bool rok[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int r = row0 + e;
    rok[e] = r < D;
    mu_r[e] = rok[e] ? a.mu[r] : 0.f;
    sd_r[e] = rok[e] ? a.S[(int64_t)r * D + r] : 0.f;
  }
  for (int e = tid; e < QBUF * MSZ + QWV * QNC * 4; e += QNT) lds[e] = 0.f;
  const T eh = 0.5f * a.eps;
  const bool rany = row0 < QLD;                           // rows 116 .. 127 have no slot (and are >= D)
  const int own_off = cl * QLD + row0, b_off = cl * QLD;
  int dpar = 0;
  T ev_r[4] = {0.f, 0.f, 0.f, 0.f};
  uint64_t chain = 0;
  bool live = false;

  auto put4 = [&](T* X, const T (&v)[4]) { if (rany) *reinterpret_cast<bf4*>(X + own_off) = bf4{v[0], v[1], v[2], v[3]}; };
  auto jitter4 = [&](uint32_t n, uint32_t sub) {            // this lane's four rows are one Philox block (uniform_elem layout)
    if (!a.has_jitter) return;
    const U4 r = philox_block(a.seed, chain, n, PURPOSE_JITTER, sub, (uint32_t)(row0 >> 2));
    const T u[4] = {u23<T>(r.x), u23<T>(r.y), u23<T>(r.z), u23<T>(r.w)};
#pragma unroll
    for (int e = 0; e < 4; ++e) ev_r[e] = (live && rok[e]) ? a.jitter * u[e] : 0.f;
  };
  // One wave per SIMD: nothing else hides an LDS round trip, so the operand chunks (4 values of k per 16-byte read) are
  // fetched two chunks (16 MFMAs = 128 clocks) ahead of their use.
  auto chunk = [&](const T* X, int q) { return *reinterpret_cast<const bf4*>(X + b_off + 4 * q); };
  // two products with one pass over k: acc1 = A1 X1, acc2 = A2 X2 (two independent accumulator chains)
  auto prod2 = [&](const T (&A1)[QK], const T* X1, const T (&A2)[QK], const T* X2, bf4& acc1, bf4& acc2) {
    bf4 c1 = chunk(X1, 0), c2 = chunk(X2, 0), n1 = chunk(X1, 1), n2 = chunk(X2, 1);
#pragma unroll
    for (int q = 0; q < QK / 4; ++q) {
      bf4 f1 = n1, f2 = n2;
      if (q + 2 < QK / 4) { f1 = chunk(X1, q + 2); f2 = chunk(X2, q + 2); }
      __builtin_amdgcn_sched_barrier(0);                    // (the scheduler otherwise sinks the reads next to their use)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(A1[4 * q + u], c1[u], acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(A2[4 * q + u], c2[u], acc2, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      c1 = n1; c2 = n2; n1 = f1; n2 = f2;
    }
  };
  // x = (P + diag(e))^-1 m continued from x0 = S m (rmhmc_fused_kernel: refine)
  auto refine = [&](const T (&x0)[4], T (&xr)[4]) {
    for (int it = 0; it < a.K; ++it) {
      const T* wr = (it & 1) ? W1 : W0;
      T* ww = (it & 1) ? W0 : W1;
      __syncthreads();
      bf4 sa = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};      // even / odd k: two chains keep the pipe issuing
      bf4 c = chunk(wr, 0), n1 = chunk(wr, 1), n2 = chunk(wr, 2);
#pragma unroll
      for (int q = 0; q < QK / 4; ++q) {
        bf4 f = n2;
        if (q + 3 < QK / 4) f = chunk(wr, q + 3);
        __builtin_amdgcn_sched_barrier(0);
        sa = __builtin_amdgcn_mfma_f32_4x4x1f32(Sa[4 * q], c[0], sa, 0, 0, 0);
        sb = __builtin_amdgcn_mfma_f32_4x4x1f32(Sa[4 * q + 1], c[1], sb, 0, 0, 0);
        sa = __builtin_amdgcn_mfma_f32_4x4x1f32(Sa[4 * q + 2], c[2], sa, 0, 0, 0);
        sb = __builtin_amdgcn_mfma_f32_4x4x1f32(Sa[4 * q + 3], c[3], sb, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        c = n1; n1 = n2; n2 = f;
      }
      T wv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) { xr[e] = x0[e] - (sa[e] + sb[e]); wv[e] = ev_r[e] * xr[e]; }
      put4(ww, wv);
    }
  };
  // one half step: upd_x += eh G(X)^-1 m ; upd_g -= eh P (X - mu)   (rmhmc_fused_kernel: half_step)
  auto half_step = [&](uint32_t n, uint32_t sub, const T (&X)[4], const T* m, T (&upd_x)[4], T (&upd_g)[4], T* upd_g_lds) {
    T* d = dpar ? D1 : D0;
    dpar ^= 1;
    T dv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) dv[e] = rok[e] ? X[e] - mu_r[e] : 0.f;
    put4(d, dv);
    __syncthreads();
    bf4 Pd = {0.f, 0.f, 0.f, 0.f}, x0v = {0.f, 0.f, 0.f, 0.f};
    prod2(Pa, d, Sa, m, Pd, x0v);
    jitter4(n, sub);            // first needed after the products: its Philox rounds issue under the MFMAs
    T x0[4], xr[4], wv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      upd_g[e] -= eh * Pd[e];
      x0[e] = x0v[e]; xr[e] = x0v[e];
      wv[e] = ev_r[e] * x0v[e];
    }
    put4(upd_g_lds, upd_g);
    put4(W0, wv);
    refine(x0, xr);
#pragma unroll
    for (int e = 0; e < 4; ++e) upd_x[e] += eh * xr[e];
  };
  // ---- TRACK: the tracked-products schedule of rmhmc_mfma4x4_kernel (derivation there) in this kernel's layout ----------
  T y[4], yc[4], z[4], zc[4];
  auto prod4 = [&](const T* X1, const T* X2, const T* X3, const T* X4, bf4& p1, bf4& p2, bf4& s3, bf4& s4) {   // P X1, P X2, S X3, S X4
    bf4 c1 = chunk(X1, 0), c2 = chunk(X2, 0), c3 = chunk(X3, 0), c4 = chunk(X4, 0);
#pragma unroll
    for (int q = 0; q < QK / 4; ++q) {
      bf4 n1 = c1, n2 = c2, n3 = c3, n4 = c4;
      if (q + 1 < QK / 4) { n1 = chunk(X1, q + 1); n2 = chunk(X2, q + 1); n3 = chunk(X3, q + 1); n4 = chunk(X4, q + 1); }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        p1 = __builtin_amdgcn_mfma_f32_4x4x1f32(Pa[4 * q + u], c1[u], p1, 0, 0, 0);
        p2 = __builtin_amdgcn_mfma_f32_4x4x1f32(Pa[4 * q + u], c2[u], p2, 0, 0, 0);
        s3 = __builtin_amdgcn_mfma_f32_4x4x1f32(Sa[4 * q + u], c3[u], s3, 0, 0, 0);
        s4 = __builtin_amdgcn_mfma_f32_4x4x1f32(Sa[4 * q + u], c4[u], s4, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      c1 = n1; c2 = n2; c3 = n3; c4 = n4;
    }
  };
  // two independent solves x = (P + E)^-1 m from x_0 = S m, K phases; wa / wb return e . x_(K-1) (zero without jitter)
  auto solve2 = [&](T* WB, const T (&ea)[4], const T (&eb)[4], const T (&x0a)[4], const T (&x0b)[4], T (&xa)[4], T (&xb)[4],
                    T (&wa)[4], T (&wb)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { xa[e] = x0a[e]; xb[e] = x0b[e]; wa[e] = 0.f; wb[e] = 0.f; }
    for (int it = 0; it < a.K; ++it) {
      T* A = WB + (it & 1) * MSZ;                           // read in this phase only; rewritten two phases later
      T* B = A + 2 * MSZ;
#pragma unroll
      for (int e = 0; e < 4; ++e) { wa[e] = ea[e] * xa[e]; wb[e] = eb[e] * xb[e]; }
      put4(A, wa);
      put4(B, wb);
      __syncthreads();
      bf4 ra = {0.f, 0.f, 0.f, 0.f}, rb = {0.f, 0.f, 0.f, 0.f};
      prod2(Sa, A, Sa, B, ra, rb);
#pragma unroll
      for (int e = 0; e < 4; ++e) { xa[e] = x0a[e] - ra[e]; xb[e] = x0b[e] - rb[e]; }
    }
  };
  // a pair of half steps (S:429-433 and, with the roles of the copies swapped, S:454-458):
  //   a:  g1 -= eh P (X1 - mu)   X2 += eh (P + E_a)^-1 g2        b:  g2 -= eh P (X2 - mu)   X1 += eh (P + E_b)^-1 g1
  // with y1 = P (X1 - mu), y2 = P (X2 - mu), z1 = S g1, z2 = S g2 kept current
  auto pair_tracked = [&](T* WB, uint32_t n, uint32_t suba, uint32_t subb, T (&X1)[4], T (&X2)[4], T (&g1)[4], T (&g2)[4],
                          T (&y1)[4], T (&y2)[4], T (&z1)[4], T (&z2)[4]) {
    T ea[4], eb[4];
    jitter4(n, suba);
#pragma unroll
    for (int e = 0; e < 4; ++e) ea[e] = ev_r[e];
    jitter4(n, subb);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      eb[e] = ev_r[e];
      g1[e] -= eh * y1[e];                                  // a's momentum update ...
      z1[e] -= eh * (rok[e] ? X1[e] - mu_r[e] : 0.f);       // ... and S g1 with it
    }
    T xa[4], xb[4], wa[4], wb[4];
    solve2(WB, ea, eb, z2, z1, xa, xb, wa, wb);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      X2[e] += eh * xa[e];                                  // a's position update; P x_a = g2 - e_a . x_a(K-1)
      y2[e] += eh * (g2[e] - wa[e]);
      g2[e] -= eh * y2[e];                                  // b's momentum update
      z2[e] -= eh * (rok[e] ? X2[e] - mu_r[e] : 0.f);
      X1[e] += eh * xb[e];                                  // b's position update
      y1[e] += eh * (g1[e] - wb[e]);
    }
  };
  // three sums per chain over the rows, complete in every lane of the chain's column
  auto block_sums = [&](T (&v)[3]) {
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      v[e] += __shfl_xor(v[e], 4, 64);
      v[e] += __shfl_xor(v[e], 8, 64);
      v[e] += __shfl_xor(v[e], 16, 64);
      v[e] += __shfl_xor(v[e], 32, 64);
    }
    __syncthreads();
    if (blk == 0) {
#pragma unroll
      for (int e = 0; e < 3; ++e) red[(w * QNC + cl) * 4 + e] = v[e];
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      T s = 0.f;
#pragma unroll
      for (int i = 0; i < QWV; ++i) s += red[(i * QNC + cl) * 4 + e];
      v[e] = s;
    }
  };
  // H = -log p + D/2 log 2 pi + 1/2 log|G| + 1/2 m^T G^-1 m  (S:731)   (rmhmc_fused_kernel: hamiltonian, series branch)
  auto hamiltonian = [&](uint32_t n, uint32_t sub, const T (&X)[4], const T* m, const T (&mr)[4], T& H, T& logp, T (&Pd_out)[4],
                         T (&Sm_out)[4]) {
    T* d = (dpar && !TRACK) ? D1 : D0;                      // (TRACK: D1 belongs to the refresh phase)
    dpar ^= 1;
    jitter4(n, sub);
    T dr[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) dr[e] = rok[e] ? X[e] - mu_r[e] : 0.f;
    put4(EV, ev_r);
    put4(d, dr);
    __syncthreads();
    bf4 Pd = {0.f, 0.f, 0.f, 0.f}, x0v = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    prod2(Pa, d, Sa, m, Pd, x0v);
    if (a.has_jitter) {                                     // second-order log-det term: (S . S) e
      bf4 c = chunk(EV, 0), n1 = chunk(EV, 1), n2 = chunk(EV, 2), s2b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < QK / 4; ++q) {
        bf4 f = n2;
        if (q + 3 < QK / 4) f = chunk(EV, q + 3);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; u += 2) {
          s2 = __builtin_amdgcn_mfma_f32_4x4x1f32(Sa[4 * q + u] * Sa[4 * q + u], c[u], s2, 0, 0, 0);
          s2b = __builtin_amdgcn_mfma_f32_4x4x1f32(Sa[4 * q + u + 1] * Sa[4 * q + u + 1], c[u + 1], s2b, 0, 0, 0);
        }
        c = n1; n1 = n2; n2 = f;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) s2[e] += s2b[e];
    }
    T v[3] = {0.f, 0.f, 0.f}, x0[4], xr[4], wv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      x0[e] = x0v[e]; xr[e] = x0v[e];
      Pd_out[e] = Pd[e]; Sm_out[e] = x0v[e];
      v[0] += dr[e] * Pd[e];
      wv[e] = ev_r[e] * x0v[e];
      if (a.has_jitter) v[2] += ev_r[e] * (sd_r[e] - 0.5f * s2[e]);      // log|P + E| = log|P| + tr(SE) - 1/2 tr((SE)^2) + ...
    }
    put4(W0, wv);
    refine(x0, xr);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[1] += mr[e] * xr[e];
    block_sums(v);
    const float pi_term = (float)D * 1.8378770351409912f;   // S:712
    logp = a.log_norm - 0.5f * v[0];
    H = -logp + 0.5f * pi_term + 0.5f * (a.logdetP + v[2]) + 0.5f * v[1];
  };

  const int64_t ngroup = (a.C + QNC - 1) / QNC;
  for (int64_t cg = blockIdx.x; cg < ngroup; cg += gridDim.x) {
    const int64_t c = QNC * cg + cl;
    live = c < a.C;
    chain = a.chain_offset + (uint64_t)(live ? c : 0);
    T scur[4], sth[4], spm[4], sthc[4], spmc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) scur[e] = (live && rok[e]) ? a.cur[c * D + row0 + e] : 0.f;
    int32_t rejected = 0;
    __syncthreads();                                        // the previous group's last reads of the vector matrices
    for (int t = 0; t < a.n_traj; ++t) {
      const uint32_t n = (uint32_t)(a.traj_offset + t);
      // ---- gibbs: p = chol(G(theta)) z, drawn ahead by the momentum kernel (S:183-184)
#pragma unroll
      for (int e = 0; e < 4; ++e) spm[e] = (live && rok[e]) ? a.p_ws[((int64_t)t * a.C + c) * D + row0 + e] : 0.f;
      put4(PM, spm);
      T H0, H1, lp0, lp1;
      hamiltonian(n, 1, scur, PM, spm, H0, lp0, y, z);      // S:971 -> S:822
#pragma unroll
      for (int e = 0; e < 4; ++e) { sth[e] = scur[e]; sthc[e] = scur[e]; spmc[e] = spm[e]; yc[e] = y[e]; zc[e] = z[e]; }   // S:425-426
      if (!TRACK) put4(PMC, spm);
      for (int lstep = 0; lstep < a.L; ++lstep) {           // S:427-461
        const uint32_t k0 = 2u + 8u * (uint32_t)lstep;
        if (TRACK) pair_tracked(WS, n, k0 + 1, k0 + 2, sth, sthc, spm, spmc, y, yc, z, zc);    // phi_A/2, phi_B/2  S:429-433
        else {
          half_step(n, k0 + 1, sth, PMC, sthc, spm, PM);    // phi_A/2  S:429-430
          half_step(n, k0 + 2, sthc, PM, sth, spmc, PMC);   // phi_B/2  S:432-433
        }
        if (a.K == 0) __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) {                       // phi_C    S:447-450, sequential (Q1)
          T xx = sth[e], b = spm[e], xc = sthc[e], bc = spmc[e];
          const T h = 0.5f, cc = a.rot_c, ss = a.rot_s;
          xx = h * ((xx + xc) + cc * (xx - xc) + ss * (b - bc));
          b = h * ((b + bc) - ss * (xx - xc) + cc * (b - bc));
          xc = h * ((xx + xc) - cc * (xx - xc) - ss * (b - bc));
          bc = h * ((b + bc) + ss * (xx - xc) - cc * (b - bc));
          sth[e] = xx; spm[e] = b; sthc[e] = xc; spmc[e] = bc;
        }
        put4(PM, spm);
        put4(PMC, spmc);
        if (TRACK) {                                        // the four tracked products of the rotated state, afresh
          T dt[4], dc[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) { dt[e] = rok[e] ? sth[e] - mu_r[e] : 0.f; dc[e] = rok[e] ? sthc[e] - mu_r[e] : 0.f; }
          put4(D1, dt);
          put4(DC, dc);
          __syncthreads();
          bf4 p1 = {0.f, 0.f, 0.f, 0.f}, p2 = {0.f, 0.f, 0.f, 0.f}, s3 = {0.f, 0.f, 0.f, 0.f}, s4 = {0.f, 0.f, 0.f, 0.f};
          prod4(D1, DC, PM, PMC, p1, p2, s3, s4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { y[e] = p1[e]; yc[e] = p2[e]; z[e] = s3[e]; zc[e] = s4[e]; }
          pair_tracked(WS + 4 * MSZ, n, k0 + 4, k0 + 7, sthc, sth, spmc, spm, yc, y, zc, z);   // phi_B/2, phi_A/2  S:454-458
        } else {
          half_step(n, k0 + 4, sthc, PM, sth, spmc, PMC);   // phi_B/2  S:454-455
          half_step(n, k0 + 7, sth, PMC, sthc, spm, PM);    // phi_A/2  S:457-458
        }
      }
      if (TRACK) {                                          // (the tracked half steps keep the momenta in registers)
        if (a.K == 0) __syncthreads();                      // no solve phase since the refresh phase read PM
        put4(PM, spm);
      }
      T unused1[4], unused2[4];
      hamiltonian(n, 2u + 8u * (uint32_t)a.L, sth, PM, spm, H1, lp1, unused1, unused2);   // S:989 (Q4)
      // ---- Metropolis test + bookkeeping (S:1000-1026, S:1045-1057)
      const T u = u23<T>(philox_block(a.seed, chain, n, PURPOSE_MH, 0, 0).x);
      const bool acc = mh_accept<T>(H0, H1, lp1, u);
      const bool reset = (!acc) && ((int)n == a.burn + 1);  // Q2
      if (live) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (rok[e]) {
            const T vnew = acc ? sth[e] : (reset ? a.theta_init[c * D + row0 + e] : scur[e]);
            scur[e] = vnew;
            if (a.samples && (int)n > a.burn) a.samples[((int64_t)((int)n - a.burn) * a.C + c) * D + row0 + e] = vnew;
          }
        }
        if (w == 0 && blk == 0) {
          if (a.H_old) a.H_old[(int64_t)t * a.C + c] = H0;
          if (a.H_new) a.H_new[(int64_t)t * a.C + c] = H1;
          if (a.accept) a.accept[(int64_t)t * a.C + c] = acc ? 1 : 0;
        }
      }
      if (!acc) ++rejected;
    }
    if (live) {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (rok[e]) a.cur[c * D + row0 + e] = scur[e];
      if (w == 0 && blk == 0) a.reject_count[c] += rejected;
    }
  }
}

// =============================================================================================
// The four-chain kernel on ALL FOUR SIMDs of a CU (the default from 513 to 1024 chains; "rmhmc_mfma4_waves" = 2 keeps the
// two-wave kernel above, its parity reference).
//
// At 1024 chains there are 256 groups of four chains - one workgroup per CU - and the two-wave kernel leaves two of the four
// matrix pipes of every CU idle.  Splitting the 128 (padded) rows over four waves alone would not help: the 16-block
// instruction covers 64 rows at a time.  But its 16 blocks are INDEPENDENT 4 x 4 outer products, so here eight blocks take a
// wave's 32 rows at the even contraction indices and the other eight the SAME rows at the odd ones: one instruction advances
// k by two, a product is 50 (+2 padding) instructions per wave instead of 100, and the two partial sums of a row sit in lanes
// l and l ^ 8 of one wave - combined by a DPP row rotation, no LDS, no barrier.  Per lane: 2 x 52 matrix operands instead of
// 2 x 100.  The vectors live in LDS with even and odd rows apart (16-byte operand reads per parity).  Both parities run the
// element-wise code (duplicate state, one writer), which frees a trick for the jitter: a half step's Philox block is the
// most expensive scalar piece (quarter-rate integer multiplies), and the two parities draw the blocks of TWO half steps at
// once and exchange them - two Philox passes per step instead of four.
//
// Round 2 added three things (measured steps: profiles/README.md, r02j-r02n):
//  * TRACK ("rmhmc_pair" = 1, default): y = P (theta - mu), S p and their copies' twins are carried along element-wise, so a
//    half step keeps only its K refinement products and the two half steps of a pair solve side by side (see the comment at
//    `T y[4], ...` below): 5 product phases and 12 products per step at K = 2 instead of 8 and 16.
//  * the operand fetch through the B-broadcast modifier of the instruction (blgp 4..7, see mfma_from_group): parity and chain
//    of a lane depend only on its position inside its 16-lane group, so each group fetches a different 16-byte chunk and one
//    LDS read feeds 16 instructions instead of 4.
//  * the non-TRACK path ("rmhmc_pair" = 0) is the schedule of the two-wave kernel: same streams, same update order, same
//    barriers; the sums run over even k then odd k (results agree to rounding).
// =============================================================================================
// (layout constants and the operand helpers: rmhmc_fused_dev.hpp)
// LEAN ("rmhmc_lean"): see rmhmc_uv_kernel - no selects on the padding rows (TRACK paths): 2 897 -> 2 857 instructions per step.
// (The stores keep their lane predicate here: without it the allocator of this 456-register kernel spills 22 values to scratch
//  inside the step loop - tools/isa_of.py on the instance.)
template <bool TRACK, bool LEAN = false>
__global__ __launch_bounds__(XNT) void rmhmc_mfma4x4_kernel(FusedArgs<float> a) {
  typedef float T;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* lds = reinterpret_cast<T*>(smem_raw);
  constexpr int MSZ = XNC * XLD;
  T* PM = lds; T* PMC = PM + MSZ; T* D0 = PMC + MSZ; T* D1 = D0 + MSZ; T* W0 = D1 + MSZ; T* W1 = W0 + MSZ; T* EV = W1 + MSZ;
  T* DC = EV + MSZ;                                       // tracked products: theta_c - mu (D1: theta - mu)
  T* WS = DC + MSZ;                                       // tracked products: 2 pairs x 2 solves x 2 refinement vectors
  T* red = WS + 8 * MSZ;                                  // [XWV][XNC][4]
  // lane bits: [1:0] chain, [2] low bit of the row block, [3] contraction parity, [5:4] 16-lane group = high bits of the row
  // block.  The parity and the chain are functions of the lane's position INSIDE its 16-lane group, so a group's B operand
  // serves all four groups (mfma_from_group): each group fetches a different 16-byte chunk of the vectors and one LDS read
  // feeds 16 matrix instructions instead of 4 - with every lane fetching its own copy of every chunk the four waves' reads
  // (1 KB per instruction) kept the LDS port busy 64 of every 67 clocks of matrix work.
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, cl = l & 3, grp = l >> 4;
  const int kpar = (l >> 3) & 1, rb = 2 * grp + ((l >> 2) & 1);
  const bool upper = kpar != 0, lead = (l >> 2) == 0;
  const int D = a.D;
  const int row0 = 32 * w + 4 * rb, arow = row0 + cl;     // this lane OWNS rows row0..row0+3 of chain cl and SUPPLIES matrix row arow
  T Sa[XKJ], Pa[XKJ];
#pragma unroll
  for (int j = 0; j < XKJ; ++j) {
    const int k = 2 * j + kpar;
    const bool ok = arow < D && k < D;
    Sa[j] = ok ? a.S[(int64_t)k * D + arow] : 0.f;        // symmetric: column arow, coalesced over the lanes
    Pa[j] = ok ? a.P[(int64_t)k * D + arow] : 0.f;
  }
  T mu_r[4], sd_r[4];
  bool rok[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int r = row0 + e;
    rok[e] = r < D;
    mu_r[e] = rok[e] ? a.mu[r] : 0.f;
    sd_r[e] = rok[e] ? a.S[(int64_t)r * D + r] : 0.f;
  }
  for (int e = tid; e < XBUF * MSZ + XWV * XNC * 4; e += XNT) lds[e] = 0.f;
  const T eh = 0.5f * a.eps;
#if HTA_RM_TIMING
  unsigned long long xacc[16] = {0}, xlast = 0;
#endif
  const int own_off = cl * XLD + (row0 >> 1);             // rows row0, row0+2 -> even half; row0+1, row0+3 -> odd half
  const int b_off = cl * XLD + kpar * XHL + 4 * grp;      // this group's chunk of a super-chunk of four
  int dpar = 0;
  T ev_r[4] = {0.f, 0.f, 0.f, 0.f};
  uint64_t chain = 0;
  bool live = false;

  typedef float bf2 __attribute__((ext_vector_type(2)));
  auto put4 = [&](T* X, const T (&v)[4]) {
    if (!upper) {
      *reinterpret_cast<bf2*>(X + own_off) = bf2{v[0], v[2]};
      *reinterpret_cast<bf2*>(X + own_off + XHL) = bf2{v[1], v[3]};
    }
  };
  // the jitter of this lane's four rows for ONE sub-stream (uniform_elem layout: rows 4b..4b+3 are Philox block b)
  auto jitter_raw = [&](uint32_t n, uint32_t sub, T (&out)[4]) {
    const U4 r = philox_block(a.seed, chain, n, PURPOSE_JITTER, sub, (uint32_t)(row0 >> 2));
    const T u[4] = {u23<T>(r.x), u23<T>(r.y), u23<T>(r.z), u23<T>(r.w)};
#pragma unroll
    for (int e = 0; e < 4; ++e) out[e] = (live && rok[e]) ? a.jitter * u[e] : 0.f;
  };
  // two sub-streams with one Philox pass: the lower lane half draws subA, the upper half subB, then they swap
  auto jitter_pair = [&](uint32_t n, uint32_t subA, uint32_t subB, T (&eA)[4], T (&eB)[4]) {
    if (!a.has_jitter) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { eA[e] = 0.f; eB[e] = 0.f; }
      return;
    }
    T mine[4];
    jitter_raw(n, upper ? subB : subA, mine);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const T oth = other_parity(mine[e]);
      eA[e] = upper ? oth : mine[e];
      eB[e] = upper ? mine[e] : oth;
    }
  };
  // super-chunk Q of a vector: 16 contraction indices of this lane's parity, 4 per 16-lane group
  auto fetch = [&](const T* X, bf4 (&c)[XSQ]) {
#pragma unroll
    for (int Q = 0; Q < XSQ; ++Q) c[Q] = *reinterpret_cast<const bf4*>(X + b_off + 16 * Q);
  };
  auto both = [&](bf4& acc) {                                // lanes l and l ^ 8 both end with (even k) + (odd k)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] += other_parity(acc[e]);
  };
  // two products with one pass over k: acc1 = A1 X1, acc2 = A2 X2 (two independent accumulator chains); the four reads of
  // each vector are issued up front (one wave per SIMD: nothing else hides an LDS round trip)
  auto prod2 = [&](const T (&A1)[XKJ], const T* X1, const T (&A2)[XKJ], const T* X2, bf4& acc1, bf4& acc2) {
    bf4 c1[XSQ], c2[XSQ];
    fetch(X1, c1);
    fetch(X2, c2);
    __builtin_amdgcn_sched_barrier(0);                      // (the scheduler otherwise sinks the reads next to their use)
    static_for(std::make_integer_sequence<int, XQ>{}, [&](auto qc) {
      constexpr int q = decltype(qc)::value;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc1 = mfma_from_group<q % 4>(A1[4 * q + u], c1[q / 4][u], acc1);
        acc2 = mfma_from_group<q % 4>(A2[4 * q + u], c2[q / 4][u], acc2);
      }
    });
    both(acc1); both(acc2);
  };
  // one product on two accumulator chains (k in the order 0 2 | 1 3 of every chunk, as the two-wave kernel sums it)
  auto prod1 = [&](const T (&A1)[XKJ], const T* X1, bool squared, bf4& acc) {
    bf4 c1[XSQ], sb = {0.f, 0.f, 0.f, 0.f};
    fetch(X1, c1);
    __builtin_amdgcn_sched_barrier(0);
    static_for(std::make_integer_sequence<int, XQ>{}, [&](auto qc) {
      constexpr int q = decltype(qc)::value;
#pragma unroll
      for (int u = 0; u < 4; u += 2) {
        const T a0 = A1[4 * q + u], a1 = A1[4 * q + u + 1];
        acc = mfma_from_group<q % 4>(squared ? a0 * a0 : a0, c1[q / 4][u], acc);
        sb = mfma_from_group<q % 4>(squared ? a1 * a1 : a1, c1[q / 4][u + 1], sb);
      }
    });
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] += sb[e];
    both(acc);
  };
  // x = (P + diag(e))^-1 m continued from x0 = S m (rmhmc_fused_kernel: refine)
  auto refine = [&](const T (&x0)[4], T (&xr)[4]) {
    for (int it = 0; it < a.K; ++it) {
      const T* wr = (it & 1) ? W1 : W0;
      T* ww = (it & 1) ? W0 : W1;
      __syncthreads();
      bf4 sa = {0.f, 0.f, 0.f, 0.f};
      prod1(Sa, wr, false, sa);
      T wv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) { xr[e] = x0[e] - sa[e]; wv[e] = ev_r[e] * xr[e]; }
      put4(ww, wv);
    }
  };
  // one half step with the jitter in ev_r: upd_x += eh G(X)^-1 m ; upd_g -= eh P (X - mu)   (rmhmc_fused_kernel: half_step)
  auto half_step = [&](const T (&X)[4], const T* m, T (&upd_x)[4], T (&upd_g)[4], T* upd_g_lds) {
    T* d = dpar ? D1 : D0;
    dpar ^= 1;
    T dv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) dv[e] = rok[e] ? X[e] - mu_r[e] : 0.f;
    put4(d, dv);
    __syncthreads();
    bf4 Pd = {0.f, 0.f, 0.f, 0.f}, x0v = {0.f, 0.f, 0.f, 0.f};
    prod2(Pa, d, Sa, m, Pd, x0v);
    T x0[4], xr[4], wv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      upd_g[e] -= eh * Pd[e];
      x0[e] = x0v[e]; xr[e] = x0v[e];
      wv[e] = ev_r[e] * x0v[e];
    }
    put4(upd_g_lds, upd_g);
    put4(W0, wv);
    refine(x0, xr);
#pragma unroll
    for (int e = 0; e < 4; ++e) upd_x[e] += eh * xr[e];
  };
  // ---- TRACK: the half steps without their P d and S m products ------------------------------------------------------
  // S = P^-1 is what makes this path possible in the first place, and it also makes two of a half step's K + 2 products
  // redundant.  Keep y = P (theta - mu), y_c = P (theta_c - mu), z = S p, z_c = S p_c next to the four state vectors:
  //   p -= eh P (theta - mu) = eh y           =>  z -= eh (theta - mu)                          (S P = I)
  //   theta_c += eh x,  x = (P + E)^-1 p_c    =>  y_c += eh P x = eh (p_c - e . x_(K-1))         (x_K = S p_c - S (e . x_(K-1)))
  // element-wise, exact for the iteration as implemented.  The solve starts from x_0 = S p_c = z_c: only its K refinement
  // products remain, and the second half step of a pair (position theta_c, momentum the p just updated) no longer depends on
  // the first one's solve: the two solves run side by side, one accumulator chain each - K product phases per PAIR of half
  // steps instead of 2 (K + 1).  The rotation phi_C mixes all four vectors; right after it the four tracked products are
  // evaluated afresh in one phase (so a tracked vector carries at most four element-wise updates of rounding).  Per step at
  // K = 2: 5 barrier-separated phases and 12 products instead of 8 and 16.
  T y[4], yc[4], z[4], zc[4];
  auto prod4 = [&](const T* X1, const T* X2, const T* X3, const T* X4, bf4& p1, bf4& p2, bf4& s3, bf4& s4) {   // P X1, P X2, S X3, S X4
    bf4 c1[XSQ], c2[XSQ], c3[XSQ], c4[XSQ];
    fetch(X1, c1); fetch(X2, c2); fetch(X3, c3); fetch(X4, c4);
    __builtin_amdgcn_sched_barrier(0);
    static_for(std::make_integer_sequence<int, XQ>{}, [&](auto qc) {
      constexpr int q = decltype(qc)::value;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        p1 = mfma_from_group<q % 4>(Pa[4 * q + u], c1[q / 4][u], p1);
        p2 = mfma_from_group<q % 4>(Pa[4 * q + u], c2[q / 4][u], p2);
        s3 = mfma_from_group<q % 4>(Sa[4 * q + u], c3[q / 4][u], s3);
        s4 = mfma_from_group<q % 4>(Sa[4 * q + u], c4[q / 4][u], s4);
      }
    });
    both(p1); both(p2); both(s3); both(s4);
  };
  // two independent solves x = (P + E)^-1 m from x_0 = S m, K phases; wa / wb return e . x_(K-1) (zero without jitter)
  auto solve2 = [&](T* WB, const T (&ea)[4], const T (&eb)[4], const T (&x0a)[4], const T (&x0b)[4], T (&xa)[4], T (&xb)[4],
                    T (&wa)[4], T (&wb)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { xa[e] = x0a[e]; xb[e] = x0b[e]; wa[e] = 0.f; wb[e] = 0.f; }
    for (int it = 0; it < a.K; ++it) {
      T* A = WB + (it & 1) * MSZ;                           // read in this phase only; rewritten two phases later
      T* B = A + 2 * MSZ;
#pragma unroll
      for (int e = 0; e < 4; ++e) { wa[e] = ea[e] * xa[e]; wb[e] = eb[e] * xb[e]; }
      put4(A, wa);
      put4(B, wb);
      HTA_XTICK(3);
      __syncthreads();
      HTA_XTICK(1);
      bf4 ra = {0.f, 0.f, 0.f, 0.f}, rb = {0.f, 0.f, 0.f, 0.f};
      prod2(Sa, A, Sa, B, ra, rb);
      HTA_XTICK(2);
#pragma unroll
      for (int e = 0; e < 4; ++e) { xa[e] = x0a[e] - ra[e]; xb[e] = x0b[e] - rb[e]; }
    }
  };
  // a pair of half steps (S:429-433 and, with the roles of the copies swapped, S:454-458):
  //   a:  g1 -= eh P (X1 - mu)   X2 += eh (P + E_a)^-1 g2        b:  g2 -= eh P (X2 - mu)   X1 += eh (P + E_b)^-1 g1
  // with y1 = P (X1 - mu), y2 = P (X2 - mu), z1 = S g1, z2 = S g2 kept current
  auto pair_tracked = [&](T* WB, const T (&ea)[4], const T (&eb)[4], T (&X1)[4], T (&X2)[4], T (&g1)[4], T (&g2)[4],
                          T (&y1)[4], T (&y2)[4], T (&z1)[4], T (&z2)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      g1[e] -= eh * y1[e];                                  // a's momentum update ...
      z1[e] -= eh * ((LEAN || rok[e]) ? X1[e] - mu_r[e] : 0.f);       // ... and S g1 with it
    }
    T xa[4], xb[4], wa[4], wb[4];
    solve2(WB, ea, eb, z2, z1, xa, xb, wa, wb);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      X2[e] += eh * xa[e];                                  // a's position update; P x_a = g2 - e_a . x_a(K-1)
      y2[e] += eh * (g2[e] - wa[e]);
      g2[e] -= eh * y2[e];                                  // b's momentum update
      z2[e] -= eh * ((LEAN || rok[e]) ? X2[e] - mu_r[e] : 0.f);
      X1[e] += eh * xb[e];                                  // b's position update
      y1[e] += eh * (g1[e] - wb[e]);
    }
    HTA_XTICK(3);
  };
  // three sums per chain over the rows, complete in every lane of the chain's column (the upper lane half holds duplicates
  // of the lower one: it contributes nothing)
  auto block_sums = [&](T (&v)[3]) {
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      if (upper) v[e] = 0.f;
      v[e] += __shfl_xor(v[e], 4, 64);
      v[e] += __shfl_xor(v[e], 16, 64);
      v[e] += __shfl_xor(v[e], 32, 64);
    }
    __syncthreads();
    if (lead) {
#pragma unroll
      for (int e = 0; e < 3; ++e) red[(w * XNC + cl) * 4 + e] = v[e];
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      T s = 0.f;
#pragma unroll
      for (int i = 0; i < XWV; ++i) s += red[(i * XNC + cl) * 4 + e];
      v[e] = s;
    }
  };
  // H = -log p + D/2 log 2 pi + 1/2 log|G| + 1/2 m^T G^-1 m  (S:731)   (rmhmc_fused_kernel: hamiltonian, series branch)
  auto hamiltonian = [&](uint32_t n, uint32_t sub, const T (&X)[4], const T* m, const T (&mr)[4], T& H, T& logp, T (&Pd_out)[4],
                         T (&Sm_out)[4]) {
    T* d = (dpar && !TRACK) ? D1 : D0;                      // (TRACK: D1 belongs to the refresh phase)
    dpar ^= 1;
    if (a.has_jitter) jitter_raw(n, sub, ev_r);
    T dr[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) dr[e] = rok[e] ? X[e] - mu_r[e] : 0.f;
    put4(EV, ev_r);
    put4(d, dr);
    __syncthreads();
    bf4 Pd = {0.f, 0.f, 0.f, 0.f}, x0v = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    prod2(Pa, d, Sa, m, Pd, x0v);
    if (a.has_jitter) prod1(Sa, EV, true, s2);              // second-order log-det term: (S . S) e
    T v[3] = {0.f, 0.f, 0.f}, x0[4], xr[4], wv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      x0[e] = x0v[e]; xr[e] = x0v[e];
      Pd_out[e] = Pd[e]; Sm_out[e] = x0v[e];
      v[0] += dr[e] * Pd[e];
      wv[e] = ev_r[e] * x0v[e];
      if (a.has_jitter) v[2] += ev_r[e] * (sd_r[e] - 0.5f * s2[e]);      // log|P + E| = log|P| + tr(SE) - 1/2 tr((SE)^2) + ...
    }
    put4(W0, wv);
    refine(x0, xr);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[1] += mr[e] * xr[e];
    block_sums(v);
    const float pi_term = (float)D * 1.8378770351409912f;   // S:712
    logp = a.log_norm - 0.5f * v[0];
    H = -logp + 0.5f * pi_term + 0.5f * (a.logdetP + v[2]) + 0.5f * v[1];
  };

  const int64_t ngroup = (a.C + XNC - 1) / XNC;
  for (int64_t cg = blockIdx.x; cg < ngroup; cg += gridDim.x) {
    const int64_t c = XNC * cg + cl;
    live = c < a.C;
    chain = a.chain_offset + (uint64_t)(live ? c : 0);
    T scur[4], sth[4], spm[4], sthc[4], spmc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) scur[e] = (live && rok[e]) ? a.cur[c * D + row0 + e] : 0.f;
    int32_t rejected = 0;
    // this lane's four rows of the pre-drawn momentum of local trajectory tt as four UNCONDITIONAL loads (a lane without a row
    // reads element 0 of the block and discards it; offsets masked with masks the optimiser cannot trace back to `live` - a
    // select comes back as a branch around each load with its own s_waitcnt vmcnt(0)): issued back to back, and for trajectory
    // t + 1 a whole trajectory before their use (momentum_use pins the first use of the loaded registers where it stands)
    int lmask = -(int)live, rmask[4];
    asm volatile("" : "+v"(lmask));
#pragma unroll
    for (int e = 0; e < 4; ++e) { rmask[e] = -(int)rok[e]; asm volatile("" : "+v"(rmask[e])); }
    auto momentum_raw = [&](int tt, T (&v)[4]) {
      const T* prow = a.p_ws + ((int64_t)tt * a.C + (int64_t)((int)c & lmask)) * D;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = prow[(row0 + e) & rmask[e] & lmask];
    };
    auto momentum_use = [&](T (&v)[4], T (&out)[4]) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { asm volatile("" : "+v"(v[e])); out[e] = (live && rok[e]) ? v[e] : 0.f; }
    };
    T pnext[4];
    if (a.n_traj > 0) momentum_raw(0, pnext);
    __syncthreads();                                        // the previous group's last reads of the vector matrices
#if HTA_RM_TIMING
    xlast = __builtin_readcyclecounter();
#endif
    for (int t = 0; t < a.n_traj; ++t) {
      const uint32_t n = (uint32_t)(a.traj_offset + t);
      // ---- gibbs: p = chol(G(theta)) z, drawn ahead by the momentum kernel (S:183-184)
      momentum_use(pnext, spm);
      if (t + 1 < a.n_traj) momentum_raw(t + 1, pnext);
      put4(PM, spm);
      HTA_XTICK(7);
      T H0, H1, lp0, lp1;
      hamiltonian(n, 1, scur, PM, spm, H0, lp0, y, z);      // S:971 -> S:822
      HTA_XTICK(5);
#pragma unroll
      for (int e = 0; e < 4; ++e) { sth[e] = scur[e]; sthc[e] = scur[e]; spmc[e] = spm[e]; yc[e] = y[e]; zc[e] = z[e]; }   // S:425-426
      if (!TRACK) put4(PMC, spm);
      for (int lstep = 0; lstep < a.L; ++lstep) {           // S:427-461
        const uint32_t k0 = 2u + 8u * (uint32_t)lstep;
        T e1[4], e2[4];
        jitter_pair(n, k0 + 1, k0 + 2, e1, e2);
        HTA_XTICK(0);
        if (TRACK) pair_tracked(WS, e1, e2, sth, sthc, spm, spmc, y, yc, z, zc);     // phi_A/2, phi_B/2  S:429-433
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e) ev_r[e] = e1[e];
          half_step(sth, PMC, sthc, spm, PM);               // phi_A/2  S:429-430
#pragma unroll
          for (int e = 0; e < 4; ++e) ev_r[e] = e2[e];
          half_step(sthc, PM, sth, spmc, PMC);              // phi_B/2  S:432-433
        }
        if (a.K == 0) __syncthreads();
        jitter_pair(n, k0 + 4, k0 + 7, e1, e2);             // (issued before the rotation: independent work for the scheduler)
        HTA_XTICK(0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {                       // phi_C    S:447-450, sequential (Q1)
          T xx = sth[e], b = spm[e], xc = sthc[e], bc = spmc[e];
          const T h = 0.5f, cc = a.rot_c, ss = a.rot_s;
          xx = h * ((xx + xc) + cc * (xx - xc) + ss * (b - bc));
          b = h * ((b + bc) - ss * (xx - xc) + cc * (b - bc));
          xc = h * ((xx + xc) - cc * (xx - xc) - ss * (b - bc));
          bc = h * ((b + bc) + ss * (xx - xc) - cc * (b - bc));
          sth[e] = xx; spm[e] = b; sthc[e] = xc; spmc[e] = bc;
        }
        put4(PM, spm);
        put4(PMC, spmc);
        if (TRACK) {                                        // the four tracked products of the rotated state, afresh
          T dt[4], dc[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) { dt[e] = (LEAN || rok[e]) ? sth[e] - mu_r[e] : 0.f; dc[e] = (LEAN || rok[e]) ? sthc[e] - mu_r[e] : 0.f; }
          put4(D1, dt);
          put4(DC, dc);
          HTA_XTICK(4);
          __syncthreads();
          HTA_XTICK(1);
          bf4 p1 = {0.f, 0.f, 0.f, 0.f}, p2 = {0.f, 0.f, 0.f, 0.f}, s3 = {0.f, 0.f, 0.f, 0.f}, s4 = {0.f, 0.f, 0.f, 0.f};
          prod4(D1, DC, PM, PMC, p1, p2, s3, s4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { y[e] = p1[e]; yc[e] = p2[e]; z[e] = s3[e]; zc[e] = s4[e]; }
          HTA_XTICK(2);
          pair_tracked(WS + 4 * MSZ, e1, e2, sthc, sth, spmc, spm, yc, y, zc, z);    // phi_B/2, phi_A/2  S:454-458
        } else {
          HTA_XTICK(4);
#pragma unroll
          for (int e = 0; e < 4; ++e) ev_r[e] = e1[e];
          half_step(sthc, PM, sth, spmc, PMC);              // phi_B/2  S:454-455
#pragma unroll
          for (int e = 0; e < 4; ++e) ev_r[e] = e2[e];
          half_step(sth, PMC, sthc, spm, PM);               // phi_A/2  S:457-458
        }
      }
      if (TRACK) {                                          // (the tracked half steps keep the momenta in registers)
        if (a.K == 0) __syncthreads();                      // no solve phase since the refresh phase read PM
        put4(PM, spm);
      }
      T unused1[4], unused2[4];
      hamiltonian(n, 2u + 8u * (uint32_t)a.L, sth, PM, spm, H1, lp1, unused1, unused2);   // S:989 (Q4)
      HTA_XTICK(5);
      // ---- Metropolis test + bookkeeping (S:1000-1026, S:1045-1057)
      const T u = u23<T>(philox_block(a.seed, chain, n, PURPOSE_MH, 0, 0).x);
      const bool acc = mh_accept<T>(H0, H1, lp1, u);
      const bool reset = (!acc) && ((int)n == a.burn + 1);  // Q2
      if (live) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (rok[e]) {
            const T vnew = acc ? sth[e] : (reset ? a.theta_init[c * D + row0 + e] : scur[e]);
            scur[e] = vnew;
            if (!upper && a.samples && (int)n > a.burn) a.samples[((int64_t)((int)n - a.burn) * a.C + c) * D + row0 + e] = vnew;
          }
        }
        if (w == 0 && lead) {
          if (a.H_old) a.H_old[(int64_t)t * a.C + c] = H0;
          if (a.H_new) a.H_new[(int64_t)t * a.C + c] = H1;
          if (a.accept) a.accept[(int64_t)t * a.C + c] = acc ? 1 : 0;
        }
      }
      if (!acc) ++rejected;
      HTA_XTICK(6);
    }
    if (live) {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (rok[e] && !upper) a.cur[c * D + row0 + e] = scur[e];
      if (w == 0 && lead) a.reject_count[c] += rejected;
    }
  }
#if HTA_RM_TIMING
  if (tid == 0 && blockIdx.x == 0) for (int k = 0; k < 16; ++k) hta_rm_dbg[k] = xacc[k];
#endif
}

// The momentum draws of a block of trajectories, off the chains' critical path: task (t, c) -> p = chol(P + diag(e)) z
// with the jitter sub-stream 0 and the normals of (chain c, trajectory traj_offset + t)  (S:183-184).  One workgroup per
// task at a time, 3 per CU: the factorisations of different tasks overlap each other's LDS latency, which the chain-
// serial kernel cannot do.  Without jitter the factor is the same for every task and is computed once per workgroup.
template <typename T>
__global__ __launch_bounds__(FNT) void rmhmc_momentum_kernel(const T* __restrict__ P, int has_jitter, T jitter, int64_t C, int D,
                                                             int ld, int n_traj, int traj_offset, uint64_t seed,
                                                             uint64_t chain_offset, T* __restrict__ p_ws) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* W = reinterpret_cast<T*>(smem_raw);
  T* dg = W + D * ld; T* ev = dg + 128; T* z = ev + 128;
  const int tid = threadIdx.x;
  const int64_t ntask = (int64_t)n_traj * C;
  bool have = false;
  for (int64_t task = blockIdx.x; task < ntask; task += gridDim.x) {
    const int t = (int)(task / C);
    const int64_t c = task - (int64_t)t * C;
    const uint64_t chain = chain_offset + (uint64_t)c;
    const uint32_t n = (uint32_t)(traj_offset + t);
    __syncthreads();
    if (tid < D) {
      ev[tid] = has_jitter ? jitter * uniform_elem<T>(seed, chain, n, PURPOSE_JITTER, 0, tid) : (T)0;
      z[tid] = normal_elem<T>(seed, chain, n, 0, tid);
    }
    if (has_jitter || !have) { chol_in_lds<T>(P, ev, W, dg, D, ld, tid); have = true; }
    __syncthreads();
    if (tid < D) {
      T acc0 = dg[tid] * z[tid], acc1 = 0;
      const T* rowp = W + tid * ld;
      int k = 0;
      for (; k + 1 < tid; k += 2) { acc0 = fma(rowp[k], z[k], acc0); acc1 = fma(rowp[k + 1], z[k + 1], acc1); }
      if (k < tid) acc0 = fma(rowp[k], z[k], acc0);
      p_ws[task * D + tid] = acc0 + acc1;
    }
  }
}

// The same draws WITHOUT a factorisation per draw (round 3; tuning key "rmhmc_momsplit", default 1).  On the fused route the
// metric of an evaluation is G = P + diag(e), e = jitter * u (S:113-121 with the soft-abs map the identity), and S:183-184
// asks for p ~ N(0, G).  With L_P = chol(P) - ONE factor per target, computed by the setup on the host in double - and two
// independent standard-normal vectors,
//     p = L_P z1 + sqrt(e) . z2      has covariance  L_P L_P^T + diag(e) = G  exactly,
// the distribution MultivariateNormal(0, G).sample() draws from, given the same u; what changes is the map from the
// uniform / normal draws to p (the reference's, and "rmhmc_momsplit" = 0: chol(G) z), i.e. the realisation, not the law -
// like every other draw of this engine against torch's generator.  z1 is the momentum stream the factorising kernels use
// (normal sub-stream 0), u their jitter stream (sub-stream 0), z2 normal sub-stream 1; oracle/hmc_oracle.py::rm_gibbs_split
// is the same map.  Cost per draw: a D x D triangular matrix-vector product and three Philox passes instead of D^3 / 3
// flops of latency-bound panels: 1.9 ms -> 0.1 ms per 100 trajectories at 1024 chains.
// One wave per task; L_P transposed and zero-filled above the diagonal in LDS ([k][i]: the lanes of a wave read consecutive
// rows i, and the k loop has ONE trip count for every lane - the terms beyond the diagonal add an exact zero, so the sums
// are the ascending-k single-accumulator sums of the oracle); a lane carries its rows i = l and l + 64 through the same loop
// (two accumulators, z1 read as 16-byte broadcasts).  The three Philox streams of a draw go to three groups of lanes (z1:
// lanes 0 .., z2: lanes 32 .., u: lanes 0 .. again behind z1 when D > 128 never happens: D <= 128 means <= 32 blocks per
// stream), so a wave's critical path holds one normal4 and one uniform pass instead of three Philox passes and two normal4.
// Staging walks rows (no integer division).  First version: 0.33 ms per 102 400 draws at D = 100 (rocprofv3, profiles/r03p_*).
template <typename T>
__global__ __launch_bounds__(256) void rmhmc_momentum_split_kernel(const T* __restrict__ LP, T jitter, int64_t C, int D, int n_traj,
                                                                   int traj_offset, uint64_t seed, uint64_t chain_offset,
                                                                   T* __restrict__ p_ws) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* Lt = reinterpret_cast<T*>(smem_raw);                 // [D + 4][ldt] (+ 128): Lt[k * ldt + i] = L_P[i][k]; 0 for k > i and in the rows k >= D
  const int ldt = D | 1;
  const int lt_elems = ((D + 4) * ldt + 128 + 3) & ~3;
  T* zb = Lt + lt_elems;                                  // per wave: z1[128] | sqrt(e) [128] | z2[128], 16-byte aligned
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  T* z1 = zb + w * 384; T* se = z1 + 128; T* z2 = se + 128;
  for (int i = w; i < D; i += 4)
    for (int k = l; k < D; k += 64) Lt[k * ldt + i] = (k <= i) ? LP[(size_t)i * D + k] : (T)0;
  for (int e = D * ldt + tid; e < lt_elems; e += 256) Lt[e] = (T)0;
  for (int e = l; e < 128; e += 64) z1[e] = (T)0;         // the k loop runs to a multiple of 4
  __syncthreads();
  const int64_t ntask = (int64_t)n_traj * C;
  const int nblk = (D + 3) / 4;                           // <= 32
  const int D4 = (D + 3) & ~3;
  const bool row2 = l + 64 < D;
  const T* Lc = Lt + l;
  for (int64_t task = (int64_t)blockIdx.x * 4 + w; task < ntask; task += (int64_t)gridDim.x * 4) {
    const int t = (int)(task / C);
    const int64_t c = task - (int64_t)t * C;
    const uint64_t chain = chain_offset + (uint64_t)c;
    const uint32_t n = (uint32_t)(traj_offset + t);
    // element j of a stream = Philox block j / 4, slot j % 4 (philox.hpp)
    const int b = l & 31;
    if (b < nblk) {
      T a[4];
      if (l < 32) {
        normal4<T>(philox_block(seed, chain, n, PURPOSE_MOMENTUM, 0, (uint32_t)b), a);
#pragma unroll
        for (int q = 0; q < 4; ++q) z1[4 * b + q] = (4 * b + q < D) ? a[q] : (T)0;
        const U4 r = philox_block(seed, chain, n, PURPOSE_JITTER, 0, (uint32_t)b);
        const T u[4] = {u23<T>(r.x), u23<T>(r.y), u23<T>(r.z), u23<T>(r.w)};
#pragma unroll
        for (int q = 0; q < 4; ++q) se[4 * b + q] = sqrt(jitter * u[q]);
      } else {
        normal4<T>(philox_block(seed, chain, n, PURPOSE_MOMENTUM, 1, (uint32_t)b), a);
#pragma unroll
        for (int q = 0; q < 4; ++q) z2[4 * b + q] = a[q];
      }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    T acc0 = 0, acc1 = 0;
#pragma unroll 2
    for (int k = 0; k < D4; k += 4) {                     // ascending k, one accumulator per row: the oracle's order
      T zz[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) zz[q] = z1[k + q];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc0 = fma(Lc[(k + q) * ldt], zz[q], acc0);       // rows k + q >= D: the zeroed slack / z1 = 0
        acc1 = fma(Lc[(k + q) * ldt + 64], zz[q], acc1);
      }
    }
    // (the products are rounded before they are added - the oracle's arithmetic; the empty asm keeps them out of an fma)
    T s0 = se[l] * z2[l], s1 = se[l + 64] * z2[l + 64];
    asm volatile("" : "+v"(s0), "+v"(s1));
    if (l < D) p_ws[task * D + l] = acc0 + s0;
    if (row2) p_ws[task * D + l + 64] = acc1 + s1;
    __builtin_amdgcn_wave_barrier();
  }
}

// The same draws, one WAVE per task, the work matrix in registers (fp32, jitter on, D <= 8 NB).  The workgroup kernel above
// is bound by its LDS round trips: every trailing-update FMA reads its operands from LDS and writes its result back (1.5
// LDS operations per FMA, 2 workgroup barriers per panel, 3 tasks per CU: 53 us per task at D = 100).  Here lane
// (ty, tx) = (l >> 3, l & 7) owns the elements (ty + 8a, tx + 8b), b <= a, of the lower triangle (NB(NB+1)/2 VGPRs); only
// the current panel of 4 columns passes through LDS (one 16-byte row per lane in, one out), the rank-4 update reads 4
// panel values per block row / block column (16-byte loads) for 4 FMAs per owned element, there are no workgroup
// barriers, and the factor is never stored: p = L z is accumulated row by row as the panels finish (row r belongs to
// lane r & 63).  12+ tasks per CU.  Same streams and the same factor as rmhmc_momentum_kernel; the sums of p run in
// panel order.
template <int NB>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(512 - HTA_FUSED_WIDE_VGPRS))) void rmhmc_momentum_wave_kernel(const float* __restrict__ P, float jitter, int64_t C, int D,
                                                                 int n_traj, int traj_offset, uint64_t seed, uint64_t chain_offset,
                                                                 float* __restrict__ p_ws) {
  constexpr int NR = 8 * NB;                             // padded rows
  constexpr int NROW = (NR + 63) / 64;                   // rows per lane in the panel step
  __shared__ __attribute__((aligned(16))) float pan[NR * 4], pan2[NR * 4], zv[NR], evv[NR];
  const int l = threadIdx.x, ty = l >> 3, tx = l & 7;
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  for (int e = l; e < NR * 4; e += 64) { pan[e] = 0.f; pan2[e] = 0.f; }
  const int64_t ntask = (int64_t)n_traj * C;
  for (int64_t task = blockIdx.x; task < ntask; task += gridDim.x) {
    const int t = (int)(task / C);
    const int64_t c = task - (int64_t)t * C;
    const uint64_t chain = chain_offset + (uint64_t)c;
    const uint32_t n = (uint32_t)(traj_offset + t);
    // ---- this task's matrix: W = P + diag(jitter u)  (S:113-116), lower blocks, straight from L2
    // (rows / columns >= D are clamped copies: a factor's leading D x D part does not depend on what lies beyond it, and
    //  nothing of the padding is stored.  P does not depend on the task: the compiler keeps these loads out of the task
    //  loop, i.e. P's share of a lane stays in VGPRs for the launch - 2 waves per SIMD at NB = 13)
    float W[NB][NB];
#pragma unroll
    for (int a = 0; a < NB; ++a) {
      const float* rowp = P + min(ty + 8 * a, D - 1) * D;
#pragma unroll
      for (int b = 0; b <= a; ++b) W[a][b] = rowp[min(tx + 8 * b, D - 1)];
    }
#pragma unroll
    for (int q = 0; q < NROW; ++q) {
      const int r = l + 64 * q;
      if (r < NR) {
        evv[r] = r < D ? jitter * uniform_elem<float>(seed, chain, n, PURPOSE_JITTER, 0, r) : 0.f;
        zv[r] = r < D ? normal_elem<float>(seed, chain, n, 0, r) : 0.f;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (ty == tx) {
#pragma unroll
      for (int a = 0; a < NB; ++a) W[a][a] += evv[ty + 8 * a];
    }
    float pacc[NROW];
#pragma unroll
    for (int q = 0; q < NROW; ++q) pacc[q] = 0.f;

    for (int bp = 0; bp < NB; ++bp) {
      if (8 * bp >= D) break;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int kb = 8 * bp + 4 * half;
        if (kb < D) {
          // ---- the panel's columns kb .. kb+3 (block column bp) leave their owners' registers
          if ((tx >> 2) == half) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
              if (b == bp) {                               // uniform: one of the NB bodies runs, register indices stay static
#pragma unroll
                for (int a = b; a < NB; ++a) pan[(ty + 8 * a) * 4 + (tx & 3)] = W[a][b];
              }
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          // ---- diagonal block factor (every lane, registers), then the panel rows this lane handles
          float Ld[4][4], rinv[4];
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) {
            const f4 rowv = *reinterpret_cast<const f4*>(pan + (kb + cc) * 4);
            const bool in = kb + cc < D;
#pragma unroll
            for (int c2 = 0; c2 < 4; ++c2) Ld[cc][c2] = (in && c2 <= cc) ? rowv[c2] : (cc == c2 ? 1.f : 0.f);
          }
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) {
#pragma unroll
            for (int c2 = 0; c2 < cc; ++c2) {
              float v = Ld[cc][c2];
#pragma unroll
              for (int c3 = 0; c3 < c2; ++c3) v = fmaf(-Ld[cc][c3], Ld[c2][c3], v);
              Ld[cc][c2] = v * rinv[c2];
            }
            float v = Ld[cc][cc];
#pragma unroll
            for (int c3 = 0; c3 < cc; ++c3) v = fmaf(-Ld[cc][c3], Ld[cc][c3], v);
            rinv[cc] = fast_rsqrt<float>(v);
            Ld[cc][cc] = v * rinv[cc];
          }
          const f4 zk = *reinterpret_cast<const f4*>(zv + kb);
#pragma unroll
          for (int q = 0; q < NROW; ++q) {
            const int r = l + 64 * q;
            if (r < NR) {
              f4 out = {0.f, 0.f, 0.f, 0.f};
              if (r >= kb + 4) {
                const f4 wv = *reinterpret_cast<const f4*>(pan + r * 4);
                float lrow[4];
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                  float v = wv[cc];
#pragma unroll
                  for (int c3 = 0; c3 < cc; ++c3) v = fmaf(-lrow[c3], Ld[cc][c3], v);
                  lrow[cc] = v * rinv[cc];
                  pacc[q] = fmaf(lrow[cc], zk[cc], pacc[q]);
                }
                out = f4{lrow[0], lrow[1], lrow[2], lrow[3]};
              } else if (r >= kb) {                        // a row of the diagonal block: its finished entries times z
#pragma unroll
                for (int cc = 0; cc < 4; ++cc)
                  if (r == kb + cc) {
#pragma unroll
                    for (int c2 = 0; c2 <= cc; ++c2) pacc[q] = fmaf(Ld[cc][c2], zk[c2], pacc[q]);
                  }
              }
              if (r >= kb) *reinterpret_cast<f4*>(pan2 + r * 4) = out;      // rows of finished panels hold zeros: no-op updates
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          // ---- rank-4 update of the owned blocks right of / below the panel
          // (block columns in chunks of CH: CH column vectors live at a time instead of NB)
          constexpr int CH = 4;
#pragma unroll
          for (int b0 = 0; b0 < NB; b0 += CH) {
            if (b0 + CH > bp) {
              f4 lj[CH];
#pragma unroll
              for (int u = 0; u < CH; ++u) lj[u] = (b0 + u < NB) ? *reinterpret_cast<const f4*>(pan2 + (tx + 8 * (b0 + u)) * 4) : f4{0.f, 0.f, 0.f, 0.f};
              f2 ljp[CH / 2][4];
#pragma unroll
              for (int u = 0; u < CH; u += 2)
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) ljp[u >> 1][cc] = f2{lj[u][cc], lj[u + 1][cc]};
#pragma unroll
              for (int a = b0; a < NB; ++a) {
                if (a >= bp) {
                  const f4 li = *reinterpret_cast<const f4*>(pan2 + (ty + 8 * a) * 4);
#pragma unroll
                  for (int u = 0; u < CH; u += 2) {        // two columns per v_pk_fma_f32 (columns left of the panel see zeros
                    if (b0 + u + 1 <= a) {                 //  in pan2: no-op updates)
                      f2 v = {W[a][b0 + u], W[a][b0 + u + 1]};
#pragma unroll
                      for (int cc = 0; cc < 4; ++cc) v = __builtin_elementwise_fma(f2{-li[cc], -li[cc]}, ljp[u >> 1][cc], v);
                      W[a][b0 + u] = v[0]; W[a][b0 + u + 1] = v[1];
                    } else if (b0 + u <= a) {
                      float v = W[a][b0 + u];
#pragma unroll
                      for (int cc = 0; cc < 4; ++cc) v = fmaf(-li[cc], lj[u][cc], v);
                      W[a][b0 + u] = v;
                    }
                  }
                }
              }
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
#pragma unroll
    for (int q = 0; q < NROW; ++q) {
      const int r = l + 64 * q;
      if (r < D) p_ws[task * D + r] = pacc[q];
    }
    // pan2 holds zeros again except the last panel's diagonal rows (written as zeros above): nothing to clear
  }
}

// S = V0 diag(1 / lam0) V0^T from the eigen-system of the jitter-free P (one workgroup; once per run)
template <typename T>
__global__ void inverse_from_eigen_kernel(const T* __restrict__ V0, const T* __restrict__ lam0, T* __restrict__ S, int D) {
  for (int e = threadIdx.x; e < D * D; e += blockDim.x) {
    const int i = e / D, j = e - i * D;
    T acc = 0;
    for (int k = 0; k < D; ++k) acc = fma(V0[i * D + k] / lam0[k], V0[j * D + k], acc);
    S[e] = acc;
  }
}

template <typename T> size_t fused_lds_bytes(int D, int* ld_out, int NC = 1, bool need_w = true) {
  const int ld = D | 1;
  if (ld_out) *ld_out = ld;
  return ((size_t)NC * FVC * 128 + 128 + 32 + 8 * 128 + (need_w ? (size_t)D * ld : 0)) * sizeof(T);
}

// Decides whether the identity-soft-abs path applies (lam0: host copy of the jitter-free eigenvalues) and, if so,
// how many refinement products a solve needs (return value K >= 0; -1: the Jacobi path must be used), log |P|, and
// whether the two log-determinants of a trajectory may use log|P + E| = log|P| + tr(SE) - tr((SE)^2)/2, whose
// truncation error is below D rho^3 / 3 (accepted when that is under a quarter ulp of the Hamiltonian's D/2 log 2 pi).
template <typename T>
int fused_plan(const T* lam0_host, int D, int metric, double alpha, int has_jitter, double jitter, double* logdetP, int* series) {
  if (D > 128 || fused_lds_bytes<T>(D, nullptr) > 150 * 1024) return -1;     // 2 row blocks of 64 lanes, slices of <= 64
  double lmin = lam0_host[0], ld = 0;
  for (int i = 0; i < D; ++i) { lmin = lam0_host[i] < lmin ? (double)lam0_host[i] : lmin; ld += log((double)lam0_host[i]); }
  if (!(lmin > 0.0)) return -1;                                     // not positive definite: soft-abs flips signs
  if (metric == HTA_METRIC_SOFTABS && !(alpha * lmin >= 20.0)) return -1;   // coth(alpha lam) != 1 at working precision
  *logdetP = ld; *series = 1;
  if (!has_jitter) return 0;
  if (!(jitter >= 0.0)) return -1;
  const double rho = jitter / lmin;
  if (rho > 0.25) return -1;
  if (rho == 0.0) return 0;
  const double eps = sizeof(T) == 4 ? 6e-8 : 1.1e-16;
  *series = (D * rho * rho * rho / 3.0) <= 0.25 * eps * (0.5 * D);
  int K = (int)ceil(log(eps * 0.25) / log(rho)) - 1;                // relative error after K refinements: rho^(K+1)
  if (K < 1) K = 1;
  return K > 40 ? -1 : K;
}

// side stream + events of the momentum / trajectory overlap, one set per device, created on first use
struct Overlap { hipStream_t side; hipEvent_t start, ready[2], freed[2]; };
static Overlap* overlap_for_current_device() {
  static Overlap pool[16];
  static bool made[16] = {false};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  if (!made[dev]) {
    Overlap& o = pool[dev];
    if (hipStreamCreateWithFlags(&o.side, hipStreamNonBlocking) != hipSuccess) return nullptr;
    bool ok = hipEventCreateWithFlags(&o.start, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < 2 && ok; ++i)
      ok = hipEventCreateWithFlags(&o.ready[i], hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&o.freed[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) return nullptr;
    made[dev] = true;
  }
  return &pool[dev];
}

int g_rmhmc_wide = 1;      // tuning key "rmhmc_wide": spill-free one-workgroup-per-CU instances of the one-chain kernel
static int fused_cu_count() {
  static int cus[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (!cus[dev]) {
    hipDeviceProp_t prop;
    cus[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
  }
  return cus[dev];
}

template <typename T>
int rmhmc_fused_sample(T* cur, const T* theta_init, const T* P, const T* Sinv, const T* mu, double log_norm, double logdetP,
                       int has_jitter, double jitter, int K, int series, int64_t C, int D, int L, double eps, double omega,
                       int n_traj, int traj_offset, int burn, uint64_t seed, uint64_t chain_offset, T* samples,
                       int32_t* reject_count, T* H_old, T* H_new, uint8_t* accept, T* p_ws, int64_t p_ws_elems, const T* LP,
                       hipStream_t s) {
  int ld;
  (void)fused_lds_bytes<T>(D, &ld);
  const float ang = (float)(2.0 * omega * eps);                      // S:435-436: float32 cos / sin whatever the state dtype
  const int grid = (int)(C < 8192 ? C : 8192);
  const int KH = (((D + 1) / 2) + 7) / 8 * 8;                        // register slice: half the contraction range, in eights
  // momenta of a block of trajectories are drawn ahead by rmhmc_momentum_kernel when the workspace has room for them
  const int64_t per_traj = C * (int64_t)D;
  int block = (p_ws && p_ws_elems >= per_traj) ? (int)(p_ws_elems / per_traj < n_traj ? p_ws_elems / per_traj : n_traj) : 0;
  // With room for two blocks the draws of block b+1 (side stream) overlap the trajectories of block b: the momentum kernel
  // is a quarter of the serial time at config 3, and both kernels are latency bound, so they share the CUs well.
  // p_ws is used as two halves; events order "half drawn" -> trajectories and "half consumed" -> next draw.
  Overlap* ov = nullptr;
  // (from 4096 chains on both kernels fill the chip by themselves and the overlap only makes them contend: 4096 chains
  //  50.5 ms overlapped, 46.6 ms serial; 2048 chains 30.7 ms overlapped, 32.3 ms serial)
  if (g_rmhmc_overlap && block >= 16 && n_traj >= 32 && (C < 4096 || g_rmhmc_overlap == 2)) {
    ov = overlap_for_current_device();
    if (ov) {
      int sub = (n_traj + 7) / 8;                      // ~8 blocks: only the first draw is exposed
      if (sub < 8) sub = 8;
      if (sub > block / 2) sub = block / 2;
      block = sub;
      (void)hipEventRecord(ov->start, s);
      (void)hipStreamWaitEvent(ov->side, ov->start, 0);
    }
  }
  static DevOnce done[8];     // per T instantiation
  static DevOnce done2[8];
  static DevOnce done_t[8];   // the tracked-products instances
  static DevOnce done_mom;
  int bidx = 0;
  for (int t0 = 0; t0 < n_traj; t0 += (block > 0 ? block : n_traj), ++bidx) {
    const int nt = block > 0 ? (n_traj - t0 < block ? n_traj - t0 : block) : n_traj;
    T* const p_blk = (ov && (bidx & 1)) ? p_ws + (int64_t)block * per_traj : p_ws;
    if (block > 0) {
      const size_t mlds = ((size_t)D * ld + 3 * 128) * sizeof(T);
      if (!done_mom) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rmhmc_momentum_kernel<T>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { set_error("hta_rmhmc_gaussian_sample: hipFuncSetAttribute: %s", hipGetErrorString(e)); return HTA_ERR_LAUNCH; }
        done_mom = true;
      }
      const int64_t ntask = (int64_t)nt * C;
      const int mgrid = (int)(ntask < 256 * 12 ? ntask : 256 * 12);
      bool wave_done = false;
      if (LP && has_jitter && g_rmhmc_momsplit && D <= 128) {
        // p = chol(P) z1 + sqrt(e) . z2: no factorisation per draw (see rmhmc_momentum_split_kernel)
        static DevOnce done_split;
        if (!done_split) {
          hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rmhmc_momentum_split_kernel<T>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
          if (e != hipSuccess) { set_error("hta_rmhmc_gaussian_sample: hipFuncSetAttribute: %s", hipGetErrorString(e)); return HTA_ERR_LAUNCH; }
          done_split = true;
        }
        hipStream_t ms = ov ? ov->side : s;
        const size_t slds = ((size_t)((((D + 4) * (D | 1)) + 128 + 3) & ~3) + 4 * 384) * sizeof(T);
        const int64_t sg = (ntask + 3) / 4;
        const int64_t sres = (int64_t)fused_cu_count() * (slds <= 52 * 1024 ? 3 : (slds <= 78 * 1024 ? 2 : 1));   // resident workgroups: L_P is staged once each
        const int sgrid = (int)(sg < sres ? sg : sres);
        if (ov) { if (bidx >= 2) (void)hipStreamWaitEvent(ov->side, ov->freed[bidx & 1], 0); }
        else profile_begin(s);
        rmhmc_momentum_split_kernel<T><<<sgrid, 256, slds, ms>>>(LP, (T)jitter, C, D, nt, traj_offset + t0, seed, chain_offset, p_blk);
        if (ov) {
          (void)hipEventRecord(ov->ready[bidx & 1], ov->side);
          (void)hipStreamWaitEvent(s, ov->ready[bidx & 1], 0);
        } else profile_end(s);
        wave_done = true;
      }
      if constexpr (sizeof(T) == 4) {
        // fp32 with jitter (a factorisation per task), D <= 104: one wave per task, work matrix in registers
        if (!wave_done && g_rmhmc_momwave && has_jitter && D <= 104) {
          hipStream_t ms = ov ? ov->side : s;
          const int wgrid = (int)(ntask < 256 * 16 ? ntask : 256 * 16);
          if (ov) { if (bidx >= 2) (void)hipStreamWaitEvent(ov->side, ov->freed[bidx & 1], 0); }
          else profile_begin(s);
          const int nb = (D + 7) / 8;
#define HTA_MOMWAVE(NB) rmhmc_momentum_wave_kernel<NB><<<wgrid, 64, 0, ms>>>(P, (float)jitter, C, D, nt, traj_offset + t0, seed, chain_offset, p_blk)
          if (nb <= 4) HTA_MOMWAVE(4);
          else if (nb <= 7) HTA_MOMWAVE(7);
          else if (nb <= 10) HTA_MOMWAVE(10);
          else HTA_MOMWAVE(13);
#undef HTA_MOMWAVE
          if (ov) {
            (void)hipEventRecord(ov->ready[bidx & 1], ov->side);
            (void)hipStreamWaitEvent(s, ov->ready[bidx & 1], 0);
          } else profile_end(s);
          wave_done = true;
        }
      }
      if (wave_done) {
        // drawn by the wave-per-draw kernel above
      } else if (ov) {
        if (bidx >= 2) (void)hipStreamWaitEvent(ov->side, ov->freed[bidx & 1], 0);     // trajectories of block b-2 are done with it
        rmhmc_momentum_kernel<T><<<mgrid, FNT, mlds, ov->side>>>(P, has_jitter, (T)jitter, C, D, ld, nt, traj_offset + t0, seed,
                                                                 chain_offset, p_blk);
        (void)hipEventRecord(ov->ready[bidx & 1], ov->side);
        (void)hipStreamWaitEvent(s, ov->ready[bidx & 1], 0);
      } else {
        profile_begin(s);
        rmhmc_momentum_kernel<T><<<mgrid, FNT, mlds, s>>>(P, has_jitter, (T)jitter, C, D, ld, nt, traj_offset + t0, seed, chain_offset, p_blk);
        profile_end(s);
      }
    }
    FusedArgs<T> a{cur, theta_init, P, Sinv, mu, (T)log_norm, (T)logdetP, has_jitter, (T)jitter, K, series, C, D, L, (T)eps,
                   (T)cosf(ang), (T)sinf(ang), nt, traj_offset + t0, burn, seed, chain_offset, samples, reject_count,
                   H_old ? H_old + (int64_t)t0 * C : nullptr, H_new ? H_new + (int64_t)t0 * C : nullptr,
                   accept ? accept + (int64_t)t0 * C : nullptr, block > 0 ? p_blk : nullptr};
    // two chains per workgroup (NC = 2; needs pre-drawn momenta and the log-det series): measured 5 % SLOWER than one
    // chain per workgroup at 1024 and 4096 chains (the pass is LDS-bandwidth bound: every wave streams every vector), so it
    // is only taken on request (tuning value 3: parity tests keep the variant alive)
    const bool pair = g_rmhmc_fused == 3 && block > 0 && (series || !has_jitter);
    const bool need_w = !(block > 0 && (series || !has_jitter));      // a Cholesky inside the kernel: work matrix in LDS
    auto launch = [&](auto kern, auto kern2, DevOnce& dn, DevOnce& dn2) -> int {
      if constexpr (sizeof(T) == 4) {
        // one or two chains per workgroup, their state sets as columns of the 16-block matrix instruction (rmhmc_uv.hip): up to
        // 2 x (compute units) chains (tuning key "rmhmc_uv": 0 off, 2 at any chain count)
        const bool uv = g_rmhmc_uv && g_rmhmc_mfma4 != 2 && g_rmhmc_batch != 2 && !pair && block > 0 && (series || !has_jitter) && D <= QK &&
                        (C <= (g_rmhmc_uv_co ? 7 : 2) * (int64_t)fused_cu_count() || g_rmhmc_uv == 2);
        if (uv) {
          profile_begin(s);
          const int rc_uv = rmhmc_uv_launch(a, fused_cu_count(), s);
          profile_end(s);
          return rc_uv;
        }
        // four chains per workgroup on the 16-block matrix instruction: C / 4 two-wave workgroups (tuning key "rmhmc_mfma4")
        const bool quad4 = g_rmhmc_mfma4 && !pair && block > 0 && (series || !has_jitter) && D <= QK &&
                           ((C >= g_rmhmc_mfma4_lo && C < g_rmhmc_mfma4_hi) || g_rmhmc_mfma4 == 2);
        if (quad4) {
          const int64_t ngroup = (C + QNC - 1) / QNC;
          profile_begin(s);
          // four waves per group while the groups fit one per CU (<= 1024 chains on 256 CUs): beyond that two groups share a CU
          // and the two-wave kernel already fills its four SIMDs
          const bool two_wave = g_rmhmc_mfma4_waves == 2 || (g_rmhmc_mfma4_waves != 5 && ngroup > 256);
          note_route("%s<%s>", two_wave ? "rmhmc_mfma4_kernel" : "rmhmc_mfma4x4_kernel", g_rmhmc_pair ? "true" : "false");
          if (two_wave) {
            const size_t qlds = (size_t)(QBUF * QNC * QLD + QWV * QNC * 4) * sizeof(float);
            if (g_rmhmc_pair) rmhmc_mfma4_kernel<true><<<(int)(ngroup < 8192 ? ngroup : 8192), QNT, qlds, s>>>(a);
            else rmhmc_mfma4_kernel<false><<<(int)(ngroup < 8192 ? ngroup : 8192), QNT, qlds, s>>>(a);
          } else {
            const size_t xlds = (size_t)(XBUF * XNC * XLD + XWV * XNC * 4) * sizeof(float);
            if (g_rmhmc_pair && g_rmhmc_lean) {
              note_route("rmhmc_mfma4x4_kernel<true,lean>");
              rmhmc_mfma4x4_kernel<true, true><<<(int)(ngroup < 8192 ? ngroup : 8192), XNT, xlds, s>>>(a);
            }
            else if (g_rmhmc_pair) rmhmc_mfma4x4_kernel<true><<<(int)(ngroup < 8192 ? ngroup : 8192), XNT, xlds, s>>>(a);
            else rmhmc_mfma4x4_kernel<false><<<(int)(ngroup < 8192 ? ngroup : 8192), XNT, xlds, s>>>(a);
          }
          profile_end(s);
          return HTA_OK;
        }
      }
      if constexpr (sizeof(T) == 4) {
        const bool batch = g_rmhmc_batch && !pair && block > 0 && (series || !has_jitter) && D <= 16 * BWV &&
                           (C >= 2048 || g_rmhmc_batch == 2);
        if (batch) {
          static DevOnce dn_b;
          if (!dn_b) {
            hipError_t e = hipSuccess;
            const void* kerns[4] = {reinterpret_cast<const void*>(&rmhmc_batch_kernel<25, true>), reinterpret_cast<const void*>(&rmhmc_batch_kernel<28, true>),
                                    reinterpret_cast<const void*>(&rmhmc_batch_kernel<25, false>), reinterpret_cast<const void*>(&rmhmc_batch_kernel<28, false>)};
            for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipFuncSetAttribute(kerns[i], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) { set_error("hta_rmhmc_gaussian_sample: hipFuncSetAttribute: %s", hipGetErrorString(e)); return HTA_ERR_LAUNCH; }
            dn_b = true;
          }
          const int64_t ngroup = (C + BNC - 1) / BNC;
          const size_t blds = (size_t)(BBUF * BNC * BLD + BWV * BNC * 4) * sizeof(float);      // 88 KB: one workgroup per CU
          const int bgrid = (int)(ngroup < 4096 ? ngroup : 4096);
          profile_begin(s);
          note_route("rmhmc_batch_kernel<%d,%s>", D <= 100 ? 25 : 28, g_rmhmc_pair ? "true" : "false");
          if (g_rmhmc_pair) {
            if (D <= 100) rmhmc_batch_kernel<25, true><<<bgrid, BNT, blds, s>>>(a);
            else rmhmc_batch_kernel<28, true><<<bgrid, BNT, blds, s>>>(a);
          } else {
            if (D <= 100) rmhmc_batch_kernel<25, false><<<bgrid, BNT, blds, s>>>(a);
            else rmhmc_batch_kernel<28, false><<<bgrid, BNT, blds, s>>>(a);
          }
          profile_end(s);
          return HTA_OK;
        }
      }
      if (pair) {
        if (!dn2) {
          hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern2), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
          if (e != hipSuccess) { set_error("hta_rmhmc_gaussian_sample: hipFuncSetAttribute: %s", hipGetErrorString(e)); return HTA_ERR_LAUNCH; }
          dn2 = true;
        }
        const int64_t npair = (C + 1) / 2;
        profile_begin(s);
        note_route("rmhmc_fused_kernel<%s,%d,2>", sizeof(T) == 4 ? "float" : "double", KH);
        kern2<<<(int)(npair < 8192 ? npair : 8192), FNT, fused_lds_bytes<T>(D, nullptr, 2, false), s>>>(a, ld, 0);
        profile_end(s);
        return HTA_OK;
      }
      if (!dn) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { set_error("hta_rmhmc_gaussian_sample: hipFuncSetAttribute: %s", hipGetErrorString(e)); return HTA_ERR_LAUNCH; }
        dn = true;
      }
      if constexpr (sizeof(T) == 4) {
        if (g_rmhmc_wide && (KH == 56 || KH == 64) && C <= fused_cu_count()) {       // one workgroup per CU at most: the spill-free instances
          static DevOnce dw4[4];
          DevOnce& dwf = dw4[(KH == 64) + 2 * (g_rmhmc_pair != 0)];
          auto wide = g_rmhmc_pair ? (KH == 56 ? &rmhmc_fused_kernel_wide<56, true> : &rmhmc_fused_kernel_wide<64, true>)
                                   : (KH == 56 ? &rmhmc_fused_kernel_wide<56, false> : &rmhmc_fused_kernel_wide<64, false>);
          if (!dwf) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wide), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) { set_error("hta_rmhmc_gaussian_sample: hipFuncSetAttribute: %s", hipGetErrorString(e)); return HTA_ERR_LAUNCH; }
            dwf = true;
          }
          profile_begin(s);
          note_route("rmhmc_fused_kernel_wide<%d,%s>", KH, g_rmhmc_pair ? "true" : "false");
          wide<<<grid, FNT, fused_lds_bytes<T>(D, nullptr, 1, need_w), s>>>(a, ld, need_w ? 1 : 0);
          profile_end(s);
          return HTA_OK;
        }
      }
      profile_begin(s);
      note_route("rmhmc_fused_kernel<%s,%d,1,%s>", sizeof(T) == 4 ? "float" : "double", KH, (sizeof(T) == 4 && g_rmhmc_pair) ? "true" : "false");
      kern<<<grid, FNT, fused_lds_bytes<T>(D, nullptr, 1, need_w), s>>>(a, ld, need_w ? 1 : 0);
      profile_end(s);
      return HTA_OK;
    };
    int rc;
    // the tracked-products instances: fp32 only (the fp64 ones would spill twice as much under the 256-register cap)
#define HTA_KH_CASE(KHV, IDX)                                                                                              \
    if constexpr (sizeof(T) == 4) {                                                                                       \
      if (g_rmhmc_pair) { rc = launch(&rmhmc_fused_kernel<T, KHV, 1, true>, &rmhmc_fused_kernel<T, KHV, 2>, done_t[IDX], done2[IDX]); break; } \
    }                                                                                                                     \
    rc = launch(&rmhmc_fused_kernel<T, KHV, 1>, &rmhmc_fused_kernel<T, KHV, 2>, done[IDX], done2[IDX]); break;
    switch (KH) {
      case 8: HTA_KH_CASE(8, 0)
      case 16: HTA_KH_CASE(16, 1)
      case 24: HTA_KH_CASE(24, 2)
      case 32: HTA_KH_CASE(32, 3)
      case 40: HTA_KH_CASE(40, 4)
      case 48: HTA_KH_CASE(48, 5)
      case 56: HTA_KH_CASE(56, 6)
      default: HTA_KH_CASE(64, 7)
    }
#undef HTA_KH_CASE
    if (rc) return rc;
    if (ov) (void)hipEventRecord(ov->freed[bidx & 1], s);
  }
  HTA_CHECK_LAUNCH("hta_rmhmc_gaussian_sample (fused)");
  return HTA_OK;
}

template <typename T> int inverse_from_eigen(const T* V0, const T* lam0, T* S, int D, hipStream_t s) {
  inverse_from_eigen_kernel<T><<<1, 1024, 0, s>>>(V0, lam0, S, D);
  HTA_CHECK_LAUNCH("hta_rmhmc_gaussian_sample (inverse)");
  return HTA_OK;
}

#define HTA_INST(T)                                                                                                   \
  template int fused_plan<T>(const T*, int, int, double, int, double, double*, int*);                                 \
  template int inverse_from_eigen<T>(const T*, const T*, T*, int, hipStream_t);                                       \
  template int rmhmc_fused_sample<T>(T*, const T*, const T*, const T*, const T*, double, double, int, double, int,   \
                                     int, int64_t, int, int, double, double, int, int, int, uint64_t, uint64_t, T*,   \
                                     int32_t*, T*, T*, uint8_t*, T*, int64_t, const T*, hipStream_t);
HTA_INST(float)
HTA_INST(double)

}  // namespace hta
