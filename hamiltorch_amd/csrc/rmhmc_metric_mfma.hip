// Riemannian-metric evaluation on the matrix cores (fp32, D <= 112): the evaluations of a constant-curvature target, which
// all start from ONE shared eigenbasis.
//
// Same contract as metric_eval_kernel (rmhmc_metric.hip): per system b it replaces
//   fisher()            (samplers.py:108-122)  Hs + jitter*U(0,1) on the diagonal -> eigh -> lam/tanh(alpha lam) -> Q diag Q^T
//   cholesky_inverse()  (samplers.py:146-148)  G^-1 p
//   rm_hamiltonian()    (samplers.py:710-731)  -logp + D/2 log 2pi + 1/2 log|G| + 1/2 p^T G^-1 p
//   gibbs(RMHMC)        (samplers.py:183-184)  p = chol(G) z
// and the fused first-order updates of an explicit half step (samplers.py:429-458).
//
// The eigendecomposition.  Every evaluation's matrix is Hs_b = P + diag(e_b), e_b = jitter * U(0,1)^D (S:113-115), and the
// driver has diagonalised the jitter-free P once (V0, lam0).  In that basis A = V0^T Hs_b V0 = diag(lam0) + V0^T diag(e) V0
// is a small perturbation of a diagonal matrix, and its eigenvectors are REFINED from X = I by the iteration of Ogita &
// Aishima (Japan J. Indust. Appl. Math. 35, 2018: "Iterative refinement for symmetric eigenvalue decomposition"):
//     S = X^T A X,  Gm = X^T X,  lam_i = S_ii / Gm_ii,
//     E_ii = (1 - Gm_ii) / 2,   E_ij = (S_ij - lam_j Gm_ij) / (lam_j - lam_i),   X <- X (I + E),
// quadratically convergent, and nothing but dense products: five D^3 GEMMs per evaluation (formation of A, A X, X^T (A X),
// X^T X, X E) as v_mfma_f32_16x16x4_f32 tiles (exact fp32 products, fp32 accumulation) on [DP][LD] buffers in LDS.  The
// iteration needs the coupling to be small against the eigenvalue gaps; max |E_ij| says whether it is: above 0.03 (a
// nearly degenerate spectrum, a large jitter, a stalled iteration) the system falls back - inside the same launch - to
// the cyclic Jacobi solver of rmhmc_metric.hip, which needs no such assumption.  Results agree with that solver to
// rounding (tests/test_gpu_rmhmc.py::test_metric_mfma_kernel_equals_jacobi_kernel); hta_set_tuning("metric_mfma", 0)
// selects it outright.
//
// The momentum draw / fisher() outputs add Q = V0 X, G = Q diag(lam~) Q^T (two more GEMMs) and a right-looking Cholesky in
// 16-column panels whose triangular solve and trailing update are MFMA tiles as well.
// Matrix-vector products (V0^T m, X^T m', X w, V0 x', P d) use all 1024 threads: 8 lanes per row, DPP-free shuffles.
#include "rmhmc_metric_dev.hpp"

namespace hta {

int g_metric_mfma = 1;   // tuning key "metric_mfma": 1 = warm fp32 evaluations run here, 0 = always the Jacobi kernel

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr float kFallbackE = 0.03f;   // max |E_ij| beyond which the refinement is not trusted
constexpr float kConvE = 3e-4f;       // an update with max |E_ij| below this leaves an error of order 1e-7

// tile index -> (I, J), I <= J, row-major over the upper block triangle
__device__ __forceinline__ void upper_tile(int t, int nt, int& I, int& J) {
  I = 0;
  while (t >= nt - I) { t -= nt - I; ++I; }
  J = I + t;
}

// C = Cinit + op(A) op(B) on zero-padded [DP][LD] buffers.  op(A)[m][k] = TA ? A[k][m] : A[m][k]; op(B)[k][n] = TB ? B[n][k]
// : B[k][n], times kscale[k] when given.  SYM: the product is symmetric - upper tiles only, mirrored on store.
// Lane l of a wave feeds A[m = l & 15][k = l >> 4], B[k = l >> 4][n = l & 15] and owns C[4 (l >> 4) + r][l & 15].
template <bool TA, bool TB, bool SYM>
__device__ __forceinline__ void lds_gemm(const float* A, const float* B, float* C, const float* Cinit, const float* kscale,
                                         int nt, int LD) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int ntile = SYM ? nt * (nt + 1) / 2 : nt * nt;
  for (int t = wave; t < ntile; t += MT / 64) {
    int I, J;
    if (SYM) upper_tile(t, nt, I, J);
    else { I = t / nt; J = t - I * nt; }
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    if (Cinit) {
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = Cinit[(16 * I + 4 * lk + r) * LD + 16 * J + li];
    }
    const float* ap = TA ? A + lk * LD + 16 * I + li : A + (16 * I + li) * LD + lk;
    const float* bp = TB ? B + (16 * J + li) * LD + lk : B + lk * LD + 16 * J + li;
    const int as = TA ? 4 * LD : 4, bs = TB ? 4 : 4 * LD;
    // the contraction runs over all DP = 16 nt (zero padded) indices in chunks of four instructions; the operands of the
    // next chunk are fetched from LDS while the current one is on the matrix pipe
    float a0[4], b0[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a0[u] = ap[u * as];
      b0[u] = bp[u * bs];
      if (kscale) b0[u] *= kscale[4 * u + lk];
    }
    for (int c = 0; c < nt; ++c) {
      float a1[4], b1[4];
      const int cn = (c + 1 < nt) ? c + 1 : c;                 // (the last chunk re-reads itself: no branch in the loop body)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a1[u] = ap[(4 * cn + u) * as];
        b1[u] = bp[(4 * cn + u) * bs];
        if (kscale) b1[u] *= kscale[16 * cn + 4 * u + lk];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u], b0[u], acc, 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 4; ++u) { a0[u] = a1[u]; b0[u] = b1[u]; }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      C[(16 * I + 4 * lk + r) * LD + 16 * J + li] = acc[r];
      if (SYM && I != J) C[(16 * J + li) * LD + 16 * I + 4 * lk + r] = acc[r];
    }
  }
}

__device__ __forceinline__ float block_max(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float m = red[0];
#pragma unroll
  for (int i = 1; i < MT / 64; ++i) m = fmaxf(m, red[i]);
  return m;
}

// out[row] = sum_k M(row, k) v[k] with 8 lanes per row (rows 0 .. 127 of the workgroup's 1024 threads); M(row, k) =
// TRANS ? M[k * ld + row] : M[row * ld + k]; n = vector length (rows and columns).  Every lane of a row's group returns the sum.
template <bool TRANS> __device__ __forceinline__ float mv8(const float* M, int ld, const float* v, int n) {
  const int row = threadIdx.x >> 3, seg = threadIdx.x & 7;
  float acc = 0.f;
  if (row < n)
    for (int k = seg; k < n; k += 8) acc = fmaf(TRANS ? M[k * ld + row] : M[row * ld + k], v[k], acc);
  acc += __shfl_xor(acc, 1, 64);
  acc += __shfl_xor(acc, 2, 64);
  acc += __shfl_xor(acc, 4, 64);
  return acc;
}

// the same for a lower-triangular M: out[row] = sum_{k <= row} M[row][k] v[k]  (p = L z)
__device__ __forceinline__ float mv8_lower(const float* M, int ld, const float* v, int n) {
  const int row = threadIdx.x >> 3, seg = threadIdx.x & 7;
  float acc = 0.f;
  if (row < n)
    for (int k = seg; k <= row; k += 8) acc = fmaf(M[row * ld + k], v[k], acc);
  acc += __shfl_xor(acc, 1, 64);
  acc += __shfl_xor(acc, 2, 64);
  acc += __shfl_xor(acc, 4, 64);
  return acc;
}

// a zero-padded [DP][LD] copy of a dense row-major [D][D] matrix in global memory
__device__ __forceinline__ void stage_dense(const float* __restrict__ src, float* dst, int D, int DP, int LD) {
  for (int e = threadIdx.x; e < DP * LD; e += MT) {
    const int i = e / LD, j = e - i * LD;
    dst[e] = (i < D && j < D) ? src[i * D + j] : 0.f;
  }
}

// Right-looking Cholesky of the symmetric [D][D] matrix in G (leading dimension LD, zero padded to DP) in panels of 16
// columns: the 16 x 16 diagonal block is factored and inverted by one wave (lane = row of the block; a column step is a
// broadcast of the pivot row through LDS), the panel below it is L21 = A21 inv(L11)^T and the trailing matrix loses
// L21 L21^T, both as MFMA tiles (K = 16: four instructions per tile).  The factor replaces the lower triangle; W is a
// [16][20] scratch block.  A non-positive pivot yields NaN, as the reference's cholesky raises.
__device__ __attribute__((noinline)) void mfma_cholesky(float* G, int D, int DP, int LD, float* W) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int nt = DP / 16;
  for (int pb = 0; pb < nt; ++pb) {
    const int c0 = 16 * pb;
    __syncthreads();
    if (wave == 0) {
      // --- unblocked factorisation of the diagonal block, lane i (< 16) owns row i; columns past D are the identity
      float row[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) row[j] = (lane < 16) ? G[(c0 + lane) * LD + c0 + j] : 0.f;
      const bool live = lane < 16 && c0 + lane < D;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const bool cj = c0 + j < D;
        float d = __shfl(row[j], j, 64);                       // pivot
        d = cj ? sqrtf(d) : 1.f;
        const float inv = 1.f / d;
        float lij = (lane == j) ? d : row[j] * inv;            // column j of L (rows >= j)
        if (!cj) lij = (lane == j) ? 1.f : 0.f;
        if (lane < j) lij = 0.f;
        row[j] = lij;
#pragma unroll
        for (int k = j + 1; k < 16; ++k) {
          const float lkj = __shfl(lij, k, 64);                // L[k][j]
          if (lane >= k) row[k] -= lij * lkj;
        }
      }
      if (live) {
#pragma unroll
        for (int j = 0; j < 16; ++j) G[(c0 + lane) * LD + c0 + j] = (j <= lane) ? row[j] : 0.f;
      }
      // --- inverse of the triangular block, row by row: inv(L)[i][:] = (e_i - sum_{k<i} L[i][k] inv(L)[k][:]) / L[i][i];
      //     lane = column of the inverse
      float invc[16];                                          // invc[i] = inv(L)[i][lane]
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float acc = (lane == i) ? 1.f : 0.f;
#pragma unroll
        for (int k = 0; k < i; ++k) acc -= __shfl(row[k], i, 64) * invc[k];
        invc[i] = acc / __shfl(row[i], i, 64);
      }
      if (lane < 16) {
#pragma unroll
        for (int i = 0; i < 16; ++i) W[i * 20 + lane] = invc[i];   // W[i][c] = inv(L11)[i][c]
      }
    }
    __syncthreads();
    // --- panel: L21 = A21 inv(L11)^T, i.e. L21[m][n] = sum_k A21[m][k] inv(L11)[n][k]; one wave per 16-row tile
    const int below = nt - pb - 1;
    for (int t = wave; t < below; t += MT / 64) {
      const int r0 = 16 * (pb + 1 + t);
      f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(G[(r0 + li) * LD + c0 + 4 * kk + lk], W[li * 20 + 4 * kk + lk], acc, 0, 0, 0);
      // all of the tile's operands are in registers before any lane stores (the MFMA consumed them)
#pragma unroll
      for (int r = 0; r < 4; ++r) G[(r0 + 4 * lk + r) * LD + c0 + li] = acc[r];
    }
    __syncthreads();
    // --- trailing update: A22 -= L21 L21^T on the lower tiles (I >= J)
    const int ntr = below * (below + 1) / 2;
    for (int t = wave; t < ntr; t += MT / 64) {
      int J, I;
      upper_tile(t, below, J, I);                              // J <= I
      const int r0 = 16 * (pb + 1 + I), q0 = 16 * (pb + 1 + J);
      f4 acc;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = G[(r0 + 4 * lk + r) * LD + q0 + li];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(-G[(r0 + li) * LD + c0 + 4 * kk + lk], G[(q0 + li) * LD + c0 + 4 * kk + lk], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) G[(r0 + 4 * lk + r) * LD + q0 + li] = acc[r];
    }
  }
  __syncthreads();
}

// the rare path, kept out of line so that its registers do not weigh on the refinement loop
__device__ __attribute__((noinline)) void jacobi_fallback(float* A, float* VT, int D, int ne, int LD, float* cs, float* red, int max_sweeps) {
  lds_jacobi<float, 2, 2>(A, VT, D, ne, LD, LD, cs, nullptr, red, max_sweeps);
}

__global__ __launch_bounds__(MT) void metric_warm_mfma_kernel(MetricArgsT<float> a, int DP, int LD) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int D = a.D, tid = threadIdx.x;
  const int nt = DP / 16;
  const int ne = D + (D & 1);
  float* buf0 = reinterpret_cast<float*>(smem_raw);
  float* buf1 = buf0 + DP * LD;
  float* buf2 = buf1 + DP * LD;
  float* vjit = buf2 + DP * LD;      // e = jitter * u                          (later: Jacobi (c, s) pairs)
  float* vlam = vjit + DP;           // eigenvalues of Hs
  float* vlt = vlam + DP;            // soft-abs eigenvalues
  float* vm = vlt + DP;              // m, then m' = V0^T m
  float* vy = vm + DP;               // y = X^T m', then w = y / lam~
  float* vx = vy + DP;               // x' = X w
  float* vd = vx + DP;               // d = X - mu / z
  float* vpd = vd + DP;              // P d / x
  float* red = vpd + DP;             // MT / 64
  float* W = red + MT / 64;          // [16][20] panel scratch
  const bool softabs = a.metric == 1;

  for (int64_t b = blockIdx.x; b < a.B; b += gridDim.x) {
    const uint64_t chain = a.chain_offset + (uint64_t)b;
    __syncthreads();
    // ---- 0. operands: jitter, the solve vector, d = X - mu; V0 into LDS
    for (int i = tid; i < DP; i += MT) {
      vjit[i] = (i < D && a.has_jitter) ? (float)a.jitter * uniform_elem<float>(a.seed, chain, a.draw, PURPOSE_JITTER, a.sub, i) : 0.f;
      vm[i] = (i < D && a.m) ? a.m[b * D + i] : 0.f;
      vd[i] = (i < D && a.X) ? a.X[b * D + i] - a.mu[i] : 0.f;
    }
    if (softabs) stage_dense(a.V0, buf1, D, DP, LD);
    __syncthreads();
    // ---- Gaussian log-prob and P (X - mu)
    float logp = 0.f;
    if (a.X) {
      const float pd = mv8<false>(a.Pm, D, vd, D);
      const int row = tid >> 3;
      float part = 0.f;
      if ((tid & 7) == 0 && row < D) {
        vpd[row] = pd;
        part = vd[row] * pd;
        if (a.upd_g) a.upd_g[b * D + row] += (float)a.cg * pd;
      }
      logp = (float)a.log_norm - 0.5f * block_sum(part, red);
    }
    // ---- m' = V0^T m
    if (a.m && softabs) {
      const float v = mv8<true>(buf1, LD, vm, D);
      __syncthreads();
      if ((tid & 7) == 0 && (tid >> 3) < DP) vm[tid >> 3] = ((tid >> 3) < D) ? v : 0.f;
    }
    // ---- 1. A = diag(lam0) + V0^T diag(e) V0 into buf0 (symmetric, zero padded)
    if (softabs) {
      lds_gemm<true, false, true>(buf1, buf1, buf0, nullptr, vjit, nt, LD);
      __syncthreads();
      for (int i = tid; i < D; i += MT) buf0[i * LD + i] += a.lam0[i];
    }
    __syncthreads();
    // ---- 2. eigenvectors X of A by iterative refinement from X = I; bx: X, by: A / S / E, bz: scratch
    float* bx = buf1; float* by = buf0; float* bz = buf2;
    bool have_x = false, converged = false, fallback = !softabs;     // Metric.HESSIAN: G = A, no decomposition needed
    if (softabs) {
      for (int it = 0; it < 4 && !converged && !fallback; ++it) {
        if (it >= 2) {                                               // rare: A was consumed by the previous pass, form it again
          stage_dense(a.V0, bz, D, DP, LD);
          __syncthreads();
          lds_gemm<true, false, true>(bz, bz, by, nullptr, vjit, nt, LD);
          __syncthreads();
          for (int i = tid; i < D; i += MT) by[i * LD + i] += a.lam0[i];
          __syncthreads();
        }
        if (have_x) {
          lds_gemm<false, false, false>(by, bx, bz, nullptr, nullptr, nt, LD);       // T = A X
          __syncthreads();
          lds_gemm<true, false, true>(bx, bz, by, nullptr, nullptr, nt, LD);         // S = X^T T
          __syncthreads();
          lds_gemm<true, false, true>(bx, bx, bz, nullptr, nullptr, nt, LD);         // Gm = X^T X
          __syncthreads();
        }
        for (int i = tid; i < D; i += MT) vlam[i] = have_x ? by[i * LD + i] / bz[i * LD + i] : by[i * LD + i];
        __syncthreads();
        float emax = 0.f, scale = 0.f;
        for (int i = tid; i < D; i += MT) scale = fmaxf(scale, fabsf(vlam[i]));
        scale = block_max(scale, red);
        const float tiny = 8.f * Eps<float>::v * scale;
        float* const edst = have_x ? by : bx;                         // first pass: X = I + E next to A (still needed for A X)
        for (int e = tid; e < D * D; e += MT) {
          const int i = e / D, j = e - i * D;
          const float gm = have_x ? bz[i * LD + j] : (i == j ? 1.f : 0.f);
          float E;
          if (i == j) E = 0.5f * (1.f - gm) + (have_x ? 0.f : 1.f);
          else {
            const float num = by[i * LD + j] - vlam[j] * gm;
            E = (fabsf(num) <= tiny) ? -0.5f * gm : num / (vlam[j] - vlam[i]);
            emax = fmaxf(emax, fabsf(E));
            if (!(fabsf(E) <= kFallbackE)) emax = 1.f;               // NaN / inf / too large
          }
          edst[i * LD + j] = E;
        }
        emax = block_max(emax, red);
        if (emax > kFallbackE) { fallback = true; break; }
        __syncthreads();
        if (have_x) {
          lds_gemm<false, false, false>(bx, by, bz, bx, nullptr, nt, LD);            // X <- X + X E
          __syncthreads();
          float* t = bx; bx = bz; bz = t;
        } else {
          have_x = true;                                                                 // bx = I + E, by = A still
        }
        converged = emax <= kConvE;
      }
      if (!converged) fallback = true;
      if (fallback) {
        // cyclic Jacobi on A (rmhmc_metric_dev.hpp): no assumption on gaps or perturbation size
        __syncthreads();
        stage_dense(a.V0, bz, D, DP, LD);
        __syncthreads();
        lds_gemm<true, false, true>(bz, bz, by, nullptr, vjit, nt, LD);
        __syncthreads();
        for (int i = tid; i < D; i += MT) by[i * LD + i] += a.lam0[i];
        for (int e = tid; e < DP * LD; e += MT) { const int i = e / LD, j = e - i * LD; bz[e] = (i == j && i < D) ? 1.f : 0.f; }
        __syncthreads();
        jacobi_fallback(by, bz, D, ne, LD, vjit, red, a.max_sweeps);
        for (int i = tid; i < D; i += MT) vlam[i] = by[i * LD + i];
        for (int e = tid; e < DP * LD; e += MT) { const int i = e / LD, j = e - i * LD; bx[e] = (i < D && j < D) ? bz[j * LD + i] : 0.f; }   // X[i][k] = VT[k][i]
        __syncthreads();
      }
    }
    // ---- 3. soft-abs map, log-determinant  (S:120, S:726)
    float logdet = 0.f, quad = 0.f;
    if (softabs) {
      float ld = 0.f;
      for (int i = tid; i < DP; i += MT) {
        float lt = 1.f;
        if (i < D) {
          const float lam = vlam[i];
          lt = (1.f / tanhf((float)a.alpha * lam)) * lam;
          ld += logf(lt);
          if (a.lam_out) a.lam_out[b * D + i] = lt;
          if (a.lamraw_out) a.lamraw_out[b * D + i] = lam;
        }
        vlt[i] = lt;
      }
      logdet = block_sum(ld, red);
      // ---- 4. x = V0 X (X^T m' / lam~)
      if (a.m) {
        const float y = mv8<true>(bx, LD, vm, D);
        const int row = tid >> 3;
        float qd = 0.f;
        if ((tid & 7) == 0 && row < DP) {
          const float w = (row < D) ? y / vlt[row] : 0.f;
          vy[row] = w;
          qd = (row < D) ? y * w : 0.f;
        }
        quad = block_sum(qd, red);
        const float xp = mv8<false>(bx, LD, vy, D);
        if ((tid & 7) == 0 && row < DP) vx[row] = (row < D) ? xp : 0.f;
        __syncthreads();
        const float x = mv8<false>(a.V0, D, vx, D);
        if ((tid & 7) == 0 && row < D) {
          if (a.x_out) a.x_out[b * D + row] = x;
          if (a.upd_x) a.upd_x[b * D + row] += (float)a.cx * x;
        }
      }
    }
    // ---- 5. G = Q diag(lam~) Q^T, Q = V0 X  (S:121) for fisher() / the momentum draw; Metric.HESSIAN: G = Hs itself
    if (a.G_out || a.p_out || !softabs) {
      __syncthreads();
      float* g = bz;
      if (softabs) {
        stage_dense(a.V0, by, D, DP, LD);
        __syncthreads();
        lds_gemm<false, false, false>(by, bx, bz, nullptr, nullptr, nt, LD);         // Q = V0 X
        __syncthreads();
        lds_gemm<false, true, true>(bz, bz, by, nullptr, vlt, nt, LD);               // G = Q (diag(lam~) Q^T)
        g = by;
      } else {
        for (int e = tid; e < DP * LD; e += MT) {
          const int i = e / LD, j = e - i * LD;
          float v = 0.f;
          if (i < D && j < D) { const float* Hs = a.Hs + b * a.hs_stride; v = (i >= j) ? Hs[i * D + j] : Hs[j * D + i]; if (i == j) v += vjit[i]; }
          g[e] = v;
        }
      }
      __syncthreads();
      if (a.G_out) for (int e = tid; e < D * D; e += MT) { const int i = e / D, j = e - i * D; a.G_out[b * D * D + e] = g[i * LD + j]; }
      if (a.p_out || !softabs) {
        mfma_cholesky(g, D, DP, LD, W);
        if (!softabs) {
          float ld = 0.f;
          for (int i = tid; i < D; i += MT) ld += 2.f * logf(g[i * LD + i]);             // slogdet (S:728) for a PD metric
          logdet = block_sum(ld, red);
          if (a.m) {
            for (int i = tid; i < D; i += MT) { vy[i] = a.m[b * D + i]; vx[i] = vy[i]; }
            lds_chol_solve<float>(g, D, LD, vy);
            float qd = 0.f;
            for (int i = tid; i < D; i += MT) {
              qd += vx[i] * vy[i];
              if (a.x_out) a.x_out[b * D + i] = vy[i];
              if (a.upd_x) a.upd_x[b * D + i] += (float)a.cx * vy[i];
            }
            quad = block_sum(qd, red);
          }
        }
        if (a.p_out) {                   // p = L z  (S:184 via MultivariateNormal.rsample)
          __syncthreads();
          for (int i = tid; i < DP; i += MT) vd[i] = (i < D) ? normal_elem<float>(a.seed, chain, a.draw, 0, i) : 0.f;
          __syncthreads();
          const float p = mv8_lower(g, LD, vd, D);
          if ((tid & 7) == 0 && (tid >> 3) < D) a.p_out[b * D + (tid >> 3)] = p;
        }
      }
    }
    if (tid == 0) {
      if (a.logdet_out) a.logdet_out[b] = logdet;
      if (a.quad_out) a.quad_out[b] = quad;
      if (a.logp_out) a.logp_out[b] = logp;
      if (a.H_out) {
        const float pi_term = (float)D * 1.8378770351409912f;                            // S:712 in float32
        a.H_out[b] = -logp + 0.5f * pi_term + 0.5f * logdet + 0.5f * quad;               // S:731
      }
    }
  }
}

bool metric_warm_mfma_eligible(const MetricArgsT<float>& a) {
  if (!g_metric_mfma || a.D < 1 || a.D > 112 || a.dmetric_out || a.V_out || a.L_out) return false;
  if (a.metric == 1) return a.V0 && a.lam0 && a.hs_stride == 0;      // soft-abs: evaluations that share an eigenbasis
  return a.metric == 0;                                               // Metric.HESSIAN: Cholesky + solve, any curvature input
}

int metric_warm_mfma(const MetricArgsT<float>& a, hipStream_t s) {
  const int D = a.D;
  const int DP = (D + 15) / 16 * 16, LD = DP + 4;
  const size_t lds = ((size_t)3 * DP * LD + 8 * DP + MT / 64 + 16 * 20) * sizeof(float);
  HTA_REQUIRE(lds <= 160 * 1024, "hta_metric_eval (mfma): D=%d does not fit the LDS", D);
  MetricArgsT<float> k = a;
  if (k.max_sweeps <= 0) k.max_sweeps = 16;
  static DevOnce done;
  if (!done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&metric_warm_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { set_error("hta_metric_eval: hipFuncSetAttribute: %s", hipGetErrorString(e)); return HTA_ERR_LAUNCH; }
    done = true;
  }
  const int grid = (int)(a.B < 65536 ? a.B : 65536);
  profile_begin(s);
  metric_warm_mfma_kernel<<<grid, MT, lds, s>>>(k, DP, LD);
  profile_end(s);
  HTA_CHECK_LAUNCH("hta_metric_eval (mfma)");
  return HTA_OK;
}

}  // namespace hta
