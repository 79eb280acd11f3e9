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
#include "rmhmc_metric_dev.hpp"

namespace hta {

// VG (round 4): the eigenvector matrix VT lives in a per-workgroup slab of GLOBAL memory (L2-resident: 80 KB at D = 100 fp64)
// instead of LDS - the instance for the sizes whose A + VT exceed the 160 KiB of a CU (fp64 from D = 100, fp32 from D = 141,
// up to the per-thread work lists' D ~ 110 / 156).  Same code, same barriers (a workgroup's waves share one CU: a barrier
// orders its global accesses as it orders its LDS accesses); every round pays L2 latency instead of LDS latency - the slow
// answer the reference also has there (S:108-122 has no size limit), not an error.
// AG (round 5): the work matrix A moves to the slab as well - the instance for everything beyond (fp32 D > ~198, fp64 D > ~140;
// with MAXB = MAXV = 8 work-list entries per thread: up to D = 254 fp32 / 180 fp64).  Slower again (every rotation of a round
// is an L2 round trip), never an error: SURVEY 8(a) a8, VERDICT r04 "missing" #4.
template <typename T, int MAXB, int MAXV, bool VG = false, bool AG = false>
__global__ __launch_bounds__(MT) void metric_eval_kernel(MetricArgsT<T> a, int ne, int lda, int ldv, int v0_lds, T* vws) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int D = a.D, tid = threadIdx.x;
  // LDS layout (16-byte aligned regions first): VT | [V0 copy] | (c,s) pairs | A | 4 vectors | reduction | pair table
  const int cs_len = (ne + 3) & ~3;
  T* V;                                                  // VT[D][ldv], row k = eigenvector k
  T* V0s;
  if constexpr (VG && AG) {                              // (16-byte aligned slabs: metric_geometry)
    const int64_t slab = ((int64_t)D * ldv + (int64_t)ne * lda + 3) & ~(int64_t)3;
    V = vws + (int64_t)blockIdx.x * slab; V0s = reinterpret_cast<T*>(smem_raw);
  } else if constexpr (VG) {
    // (D * ldv is a multiple of 16 bytes.  float32 only since round 6: metric_geometry sends float64 past this instance - see there.)
    V = vws + (int64_t)blockIdx.x * D * ldv; V0s = reinterpret_cast<T*>(smem_raw);
  }
  else { V = reinterpret_cast<T*>(smem_raw); V0s = V + D * ldv; }
  T* cs = V0s + (v0_lds ? D * ldv : 0);
  T* A;
  T* vec0;                        // lam~ (ne)
  if constexpr (AG) { A = V + (int64_t)D * ldv; vec0 = cs + cs_len; }
  else { A = cs + cs_len; vec0 = A + ne * lda; }
  T* vec1 = vec0 + ne;            // y / w / solve vector (ne)
  T* vec2 = vec1 + ne;            // d = X - mu, later z (ne)
  T* vec3 = vec2 + ne;            // Pd (ne)
  T* red = vec3 + ne;             // MT / 64
  int* pq = reinterpret_cast<int*>(red + MT / 64);
  const bool softabs = a.metric == 1;
  // warm start: the shared eigenbasis V0 of the jitter-free Hs (LDS copy when it fits, else L2)
  // (only when the LDS copy fits: the tiled formation below reads 16-byte vectors from padded rows)
  const bool warm = softabs && a.V0 && a.lam0 && a.hs_stride == 0 && v0_lds;
  const T* V0 = V0s;
  const int ld0 = ldv;
  if (warm) {
    for (int e = tid; e < D * ldv; e += MT) { const int i = e / ldv, j = e - i * ldv; V0s[e] = (j < D) ? a.V0[i * D + j] : (T)0; }
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
      // register-tiled: each thread owns a TB x TB tile of B and streams the rows of V0 (16-byte LDS / L2 reads)
      constexpr int TB = 16 / (int)sizeof(T);
      typedef typename Vec16<T>::type V16;
      const int nt = (D + TB - 1) / TB;
      for (int e = tid; e < ne * ne; e += MT) { const int k = e / ne, l = e - k * ne; if (k >= D || l >= D) A[k * lda + l] = 0; }
      for (int e = tid; e < nt * nt; e += MT) {
        const int tk = e / nt, tl = e - tk * nt;
        if (tl < tk) continue;                                         // upper tiles only (Jacobi reads i <= j)
        T acc[TB][TB];
#pragma unroll
        for (int x = 0; x < TB; ++x)
#pragma unroll
          for (int y = 0; y < TB; ++y) acc[x][y] = 0;
        if (a.has_jitter) {
          for (int i = 0; i < D; ++i) {
            const V16 vk = *reinterpret_cast<const V16*>(V0 + i * ld0 + tk * TB);
            const V16 vl = *reinterpret_cast<const V16*>(V0 + i * ld0 + tl * TB);
            const T ei = vec1[i];
#pragma unroll
            for (int x = 0; x < TB; ++x) {
              const T ek = ei * vk[x];
#pragma unroll
              for (int y = 0; y < TB; ++y) acc[x][y] = fma(ek, vl[y], acc[x][y]);
            }
          }
        }
#pragma unroll
        for (int x = 0; x < TB; ++x)
#pragma unroll
          for (int y = 0; y < TB; ++y) {
            const int k = tk * TB + x, l = tl * TB + y;
            if (k < D && l < D && l >= k) A[k * lda + l] = acc[x][y] + ((k == l) ? a.lam0[k] : (T)0);
          }
      }
    }
    T logdet = 0, quad = 0;
    if (softabs) {
      for (int e = tid; e < D * ldv; e += MT) { const int k = e / ldv, i = e - k * ldv; V[e] = (i == k) ? (T)1 : (T)0; }   // VT = I (incl. row padding)
      __syncthreads();
      if constexpr (MAXB == 0) lds_jacobi_dyn<T>(A, V, D, ne, lda, ldv, cs, red, a.max_sweeps);      // (run-time work lists: any D <= MT)
      else lds_jacobi<T, MAXB, MAXV>(A, V, D, ne, lda, ldv, cs, pq, red, a.max_sweeps);
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
      if (warm && (a.V_out || a.G_out || a.p_out || a.L_out || a.dmetric_out)) {
        // eigenvectors in the original basis: V <- V0 J  (through the A region; its eigenvalues are in vec0)
        __syncthreads();
        for (int e = tid; e < D * D; e += MT) {
          const int i = e / D, j = e - i * D;
          T acc = 0;
          for (int k = 0; k < D; ++k) acc += V0[i * ld0 + k] * V[j * ldv + k];     // (V0 J)[i][j], J[k][j] = VT[j][k]
          A[i * lda + j] = acc;
        }
        __syncthreads();
        for (int e = tid; e < D * D; e += MT) { const int i = e / D, j = e - i * D; V[j * ldv + i] = A[i * lda + j]; }
        __syncthreads();
      }
      const bool rotated = warm && !(a.V_out || a.G_out || a.p_out || a.L_out || a.dmetric_out);   // V still holds J
      if (a.V_out) for (int e = tid; e < D * D; e += MT) { const int i = e / D, j = e - i * D; a.V_out[b * D * D + e] = V[j * ldv + i]; }
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
          for (int i = 0; i < D; ++i) acc += V[k * ldv + i] * vec2[i];
          const T w = acc / vec0[k];
          vec1[k] = w;
          qd += acc * w;
        }
        quad = block_sum(qd, red);
        if (rotated) {                 // x = V0 (J w)
          for (int i = tid; i < D; i += MT) {
            T acc = 0;
            for (int k = 0; k < D; ++k) acc += V[k * ldv + i] * vec1[k];
            vec3[i] = acc;
          }
          __syncthreads();
        }
        for (int i = tid; i < D; i += MT) {
          T acc = 0;
          if (rotated) { for (int k = 0; k < D; ++k) acc += V0[i * ld0 + k] * vec3[k]; }
          else { for (int k = 0; k < D; ++k) acc += V[k * ldv + i] * vec1[k]; }
          if (a.x_out) a.x_out[b * D + i] = acc;
          if (a.upd_x) a.upd_x[b * D + i] += (T)a.cx * acc;
        }
      }
      if (a.dmetric_out) {
        // M = Q W Q^T with W_kl = 1/2 [k==l] lam~'_k / lam~_k - 1/2 J_kl u_k u_l  (u = Q^T m / lam~ is in vec1):
        // the derivative of 1/2 log|G| + 1/2 m^T G^-1 m with respect to the entries of Hs (Daleckii-Krein), which the
        // reference reaches by back-propagating S:726-731 through eigh (S:398).
        __syncthreads();
        for (int i = tid; i < D; i += MT) {
          const T lam = A[i * lda + i];
          vec3[i] = lam;
          vec2[i] = softabs_slope<T>((T)a.alpha, lam);
          if (!a.m) vec1[i] = 0;
        }
        __syncthreads();
        const T tol = sizeof(T) == 4 ? (T)1e-3 : (T)1e-6;
        for (int e = tid; e < D * D; e += MT) {
          const int k = e / D, l = e - k * D;
          const T lk = vec3[k], ll = vec3[l], dl = lk - ll;
          T J;
          if (k == l || fabs(dl) <= tol * (fabs(lk) + fabs(ll))) J = softabs_slope<T>((T)a.alpha, (T)0.5 * (lk + ll));
          else J = (vec0[k] - vec0[l]) / dl;
          T w = (T)-0.5 * J * vec1[k] * vec1[l];
          if (k == l) w += (T)0.5 * vec2[k] / vec0[k];
          A[k * lda + l] = w;
        }
        __syncthreads();
        // T = W Q^T in place, a row-aligned batch of rows at a time
        const int RB = MT / D;
        for (int r0 = 0; r0 < D; r0 += RB) {
          const int r = r0 + tid / D, j = tid % D;
          const bool on = tid < RB * D && r < D;
          T acc = 0;
          if (on) for (int l = 0; l < D; ++l) acc += A[r * lda + l] * V[l * ldv + j];
          __syncthreads();
          if (on) A[r * lda + j] = acc;
          __syncthreads();
        }
        for (int e = tid; e < D * D; e += MT) {
          const int i = e / D, j = e - i * D;
          T acc = 0;
          for (int k = 0; k < D; ++k) acc += V[k * ldv + i] * A[k * lda + j];
          a.dmetric_out[b * D * D + e] = acc;
        }
      }
      if (a.G_out || a.p_out || a.L_out) {          // G = Q diag(lam~) Q^T  (S:121), into the A region
        __syncthreads();
        for (int e = tid; e < D * D; e += MT) {
          const int i = e / D, j = e - i * D;
          if (j <= i) {
            T acc = 0;
            for (int k = 0; k < D; ++k) acc += V[k * ldv + i] * vec0[k] * V[k * ldv + j];
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
        if (a.dmetric_out) {
          // Hessian metric: d/dHs [1/2 log|G| + 1/2 m^T G^-1 m] = 1/2 G^-1 - 1/2 v v^T, v = G^-1 m (in vec1, 0 without m).
          // L^-1 column by column (one thread per column, plain forward substitution, no barriers) into the V region,
          // then G^-1 = L^-T L^-1.
          __syncthreads();
          if (!a.m) for (int i = tid; i < D; i += MT) vec1[i] = 0;
          for (int j = tid; j < D; j += MT) {
            for (int i = 0; i < j; ++i) V[i * ldv + j] = 0;
            V[j * ldv + j] = (T)1 / A[j * lda + j];
            for (int i = j + 1; i < D; ++i) {
              T acc = 0;
              for (int k = j; k < i; ++k) acc = fma(A[i * lda + k], V[k * ldv + j], acc);
              V[i * ldv + j] = -acc / A[i * lda + i];
            }
          }
          __syncthreads();
          for (int e = tid; e < D * D; e += MT) {
            const int i = e / D, j = e - i * D;
            T acc = 0;
            for (int k = (i > j ? i : j); k < D; ++k) acc = fma(V[k * ldv + i], V[k * ldv + j], acc);
            a.dmetric_out[b * D * D + e] = (T)0.5 * acc - (T)0.5 * vec1[i] * vec1[j];
          }
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

// Where a system's two matrices live, by size (one function for the launcher and for hta_metric_eval_workspace_bytes):
//   LDS: A[ne][lda] + VT[D][ldv] + 5 vectors + reduction scratch + pair table, the leading dimension padded to an odd stride when that
//   still fits the 160 KiB of one CU (+ a copy of the shared warm basis when there is room);
//   vglobal: VT in a global slab per workgroup (fp64 from D = 100, fp32 from D = 141);  aglobal: A there as well.
struct MetricGeom { int ne, lda, ldv, v0_lds, grid_cap; size_t lds; bool small, vglobal, aglobal, dyn; int64_t slab_elems; };
static MetricGeom metric_geometry(int D, int elem, bool warm) {
  MetricGeom g{};
  const int ne = D + (D & 1);
  auto bytes = [&](int lda, int ldv, bool a_in_lds) {
    return (size_t)((a_in_lds ? ne * lda : 0) + D * ldv + 5 * ne + 4 + MT / 64) * elem + (size_t)ne * sizeof(int) + 64;
  };
  const int vn = 16 / elem;
  g.ne = ne;
  g.ldv = ((D + vn - 1) / vn) * vn;               // VT rows are moved 16 bytes at a time
  g.lda = ne + 1;
  if (bytes(g.lda, g.ldv, true) > 160 * 1024) g.lda = ne;
  g.lds = bytes(g.lda, g.ldv, true);
  g.grid_cap = 65536;
  if (warm && g.lds + (size_t)D * g.ldv * elem <= 160 * 1024) { g.v0_lds = 1; g.lds += (size_t)D * g.ldv * elem; }
  const int NP = ne / 2, nv = g.ldv / vn;
  if (g.lds > 160 * 1024) {                        // the eigenvector matrix moves out
    g.lda = ne + 1;
    if (bytes(g.lda, 0, true) > 160 * 1024) g.lda = ne;
    g.lds = bytes(g.lda, 0, true);
    g.vglobal = true;
    g.grid_cap = 512;
    g.slab_elems = (int64_t)D * g.ldv;               // (a multiple of 16 bytes: ldv is)
  }
  // per-thread work-list lengths of the Jacobi rounds (register arrays): 2/2 up to D ~ 126, 4/3 up to ~156 fp32 / 110 fp64, 8/8 beyond
  g.small = NP * (NP + 1) / 2 <= 2 * MT && NP * nv <= 2 * MT;
  const bool mid = NP * (NP + 1) / 2 <= 4 * MT && NP * nv <= 3 * MT;
  // float64 never runs on the VT-only-in-global instance <4, 3, true, false> (round 6): that instance - 97 VGPR / 215 SGPR spills under its
  // 128-register cap - came out of hipcc 7.2 with wrong G / Cholesky accesses whenever its slab offset was spelled through a rounded
  // variable (round 5, bisected, never root-caused: tools/history/r05d.py).  A kernel whose correctness depends on the spelling of an
  // address expression is not kept in service: the sizes it served (fp64, D = 100 ... 110) take the instance with both matrices in the
  // slab, which tests/test_gpu_rmhmc.py pins at D = 100, 101, 128 and 200 against the float64 oracle.
  const bool retire_vg64 = g.vglobal && elem == 8;
  if (g.lds > 160 * 1024 || !mid || retire_vg64) {  // the work matrix moves out as well; the instance with the long work lists
    g.dyn = !(NP * (NP + 1) / 2 <= 8 * MT && NP * nv <= 8 * MT);      // beyond the longest register lists: the run-time work lists
    if (g.dyn && D > MT) { g.lds = 0; return g; }                      // (the kernel's own row batches: D <= MT)
    g.lda = ne + 1;
    g.lds = bytes(0, 0, false);
    g.vglobal = g.aglobal = true;
    g.small = false;
    g.v0_lds = 0;
    g.grid_cap = 512;
    g.slab_elems = ((int64_t)D * g.ldv + (int64_t)ne * g.lda + 3) & ~(int64_t)3;
  }
  return g;
}

int64_t metric_eval_workspace_bytes(int64_t B, int D, int elem_size) {
  if (B <= 0 || D <= 0 || (elem_size != 4 && elem_size != 8)) return 0;
  const MetricGeom g = metric_geometry(D, elem_size, false);          // (the warm copy only ever shrinks the need)
  if (!g.lds || !g.slab_elems) return 0;
  return (B < g.grid_cap ? B : g.grid_cap) * g.slab_elems * (int64_t)elem_size;
}

template <typename T> int metric_eval(const MetricArgsT<T>& a, hipStream_t s) {
  HTA_REQUIRE(a.B > 0 && a.D > 0 && a.Hs, "hta_metric_eval: bad shape / NULL Hs (B=%lld D=%d)", (long long)a.B, a.D);
  HTA_REQUIRE(a.metric == 0 || a.metric == 1, "hta_metric_eval: metric must be 0 (HESSIAN) or 1 (SOFTABS)");
  HTA_REQUIRE(!a.X || (a.Pm && a.mu), "hta_metric_eval: X given without Pm / mu");
  HTA_REQUIRE(!a.upd_g || a.X, "hta_metric_eval: upd_g needs X");
  if constexpr (sizeof(T) == 4) {
    const MetricArgsT<float>& af = reinterpret_cast<const MetricArgsT<float>&>(a);
    if (metric_warm_mfma_eligible(af)) return metric_warm_mfma(af, s);      // rmhmc_metric_mfma.hip
  }
  const int D = a.D;
  const MetricGeom g = metric_geometry(D, (int)sizeof(T), a.metric == 1 && a.V0 && a.lam0 && a.hs_stride == 0);
  HTA_REQUIRE(g.lds > 0, "hta_metric_eval: D=%d exceeds the kernel's row batches (D <= %d)", D, MT);
  MetricArgsT<T> k = a;
  if (k.max_sweeps <= 0) k.max_sweeps = sizeof(T) == 4 ? 16 : 24;
  const int grid = (int)(a.B < g.grid_cap ? a.B : g.grid_cap);
  T* vws = nullptr;
  if (g.slab_elems) {                              // ABI 10: the slabs are the caller's (rounds 4: a hipMallocAsync per call)
    const int64_t need = (int64_t)grid * g.slab_elems * (int64_t)sizeof(T);
    HTA_REQUIRE(a.workspace && a.workspace_bytes >= need, "hta_metric_eval: D=%d needs a workspace of %lld bytes for the matrices beyond "
                "one CU's LDS (HtaMetricArgs::workspace, hta_metric_eval_workspace_bytes)", D, (long long)need);
    vws = static_cast<T*>(a.workspace);
  }
  auto launch = [&](auto kern, DevOnce& done) -> int {
    if (!done) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) { set_error("hta_metric_eval: hipFuncSetAttribute: %s", hipGetErrorString(e)); return HTA_ERR_LAUNCH; }
      done = true;
    }
    profile_begin(s);
    note_route("metric_eval_kernel<%s%s>", sizeof(T) == 4 ? "float" : "double", g.dyn ? ",aglobal,dyn" : g.aglobal ? ",aglobal" : (g.vglobal ? ",vglobal" : ""));
    kern<<<grid, MT, g.lds, s>>>(k, g.ne, g.lda, g.ldv, g.v0_lds, vws);
    profile_end(s);
    return HTA_OK;
  };
  static DevOnce done_small, done_big, done_vg, done_ag, done_dyn;   // per T instantiation
  // (the float64 instance of the VT-only-in-global kernel is not even BUILT since round 6 - metric_geometry never selects it, and a kernel
  // whose correctness depended on the spelling of an address expression has no business in the code object)
  int rc;
  if (g.dyn) rc = launch(&metric_eval_kernel<T, 0, 0, true, true>, done_dyn);
  else if (g.aglobal) rc = launch(&metric_eval_kernel<T, 8, 8, true, true>, done_ag);
  else if (g.vglobal) {
    if constexpr (sizeof(T) == 4) rc = launch(&metric_eval_kernel<float, 4, 3, true>, done_vg);
    else { set_error("hta_metric_eval: internal error: the float64 vglobal instance was retired (metric_geometry)"); rc = HTA_ERR_INVALID; }
  }
  else rc = g.small ? launch(&metric_eval_kernel<T, 2, 2>, done_small) : launch(&metric_eval_kernel<T, 4, 3>, done_big);
  if (rc) return rc;
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
int64_t hta_metric_eval_workspace_bytes(int64_t B, int D, int elem_size) { return hta::metric_eval_workspace_bytes(B, D, elem_size); }
}
