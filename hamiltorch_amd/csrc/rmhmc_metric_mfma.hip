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
// quadratically convergent, and nothing but dense products (formation of A, A X, X^T (A X), X^T X, X E) as
// v_mfma_f32_16x16x4_f32 tiles (exact fp32 products, fp32 accumulation) on [DP][LD] buffers in LDS.  The
// iteration needs the coupling to be small against the eigenvalue gaps; max |E_ij| says whether it is: above 0.03 (a
// nearly degenerate spectrum, a large jitter, a stalled iteration) the system falls back - inside the same launch - to
// the cyclic Jacobi solver of rmhmc_metric.hip, which needs no such assumption.  Results agree with that solver to
// rounding (tests/test_gpu_rmhmc.py::test_metric_mfma_kernel_equals_jacobi_kernel); hta_set_tuning("metric_mfma", 0)
// selects it outright.
// Round 4: the first pass starts from X = I, so X = I + E1 with E1 antisymmetric; where max |E1_ij| <= 8e-3 (BASELINE config 3:
// 5e-3) the SECOND pass is taken in closed form - second-order perturbation theory, ONE product F E1 (ph_refine_E2) instead of
// A X, X^T A X and X^T X - and applied to the vectors of a solve without being formed: two D^3 products per solve evaluation
// (formation, F E1); hta_set_tuning("metric_second", 0) keeps the three-product pass.  And a TRAJECTORY of the Gaussian-target
// sampler is one launch (metric_traj_mfma_kernel below): the chain's workgroup runs its 4 L + 3 evaluations back to back.
//
// The momentum draw / fisher() outputs add Q = V0 X, G = Q diag(lam~) Q^T (two more GEMMs) and a right-looking Cholesky in
// 16-column panels whose triangular solve and trailing update are MFMA tiles as well.
// Matrix-vector products (V0^T m, X^T m', X w, V0 x', P d) use all 1024 threads: 8 lanes per row, DPP-free shuffles.
#include "rmhmc_metric_dev.hpp"

// The phase functions are real calls (s_swappc) by default: inlined, the trajectory kernel's body no longer fits the instruction cache a
// lone workgroup per CU has to itself and its register allocation degrades.  -DHTA_PH_INLINE=1 builds the all-inlined variant (the A/B of
// profiles/r06e_metric_inline_ab.txt).
#if defined(HTA_PH_INLINE) && HTA_PH_INLINE
#define HTA_PH_ATTR __forceinline__
#else
#define HTA_PH_ATTR __attribute__((noinline))
#endif

namespace hta {

int g_metric_mfma = 1;   // tuning key "metric_mfma": 1 = warm fp32 evaluations run here, 0 = always the Jacobi kernel
int g_metric_second = 1;    // tuning key "metric_second": 1 = the refinement's second pass in closed form (one product: ph_refine_E2) where the first pass's update is small, 0 = always the full pass (three products)
int g_metric_general = 1;   // tuning key "metric_general": 1 = evaluations with per-system curvature AND per-system bases run here too

typedef float f4 __attribute__((ext_vector_type(4)));

#ifndef HTA_TIMING
#define HTA_TIMING 0
#endif
#if HTA_TIMING      // developer builds (tools/scratch/metric_phase.cpp): s_memtime stamps of workgroup 0 at the phase boundaries
__device__ long long hta_metric_dbg[32];
__device__ long long hta_metric_wdbg[16][4];        // per wave of workgroup 0: entry / k loop begins / k loop ends / return of the formation product
#define HTA_WVSTAMP(k) do { if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) hta_metric_wdbg[threadIdx.x >> 6][k] = clock64(); } while (0)
#define HTA_STAMP(k) do { __syncthreads(); if (threadIdx.x == 0 && blockIdx.x == 0) hta_metric_dbg[k] = clock64(); } while (0)
#define HTA_WSTAMP(k) do { if (threadIdx.x == 0 && blockIdx.x == 0) hta_metric_dbg[k] = clock64(); } while (0)      // wave 0's own progress: no barrier
#else
#define HTA_STAMP(k) do { } while (0)
#define HTA_WSTAMP(k) do { } while (0)
#define HTA_WVSTAMP(k) do { } while (0)
#endif

constexpr float kFallbackE = 0.03f;   // max |E_ij| beyond which the refinement is not trusted
constexpr float kConvE = 3e-4f;       // an update with max |E_ij| below this leaves an error of order 1e-7
constexpr float kSecondE = 8e-3f;     // a first pass with max |E_ij| below this is followed by the second-order pass (ph_refine_E2): its
                                      // truncation leaves ||A X - X Lam|| of order |F| d^2 (d = max |E_ij|): < 1e-8 here, below fp32 rounding

// tile index -> (I, J), I <= J, row-major over the upper block triangle
__device__ __forceinline__ void upper_tile(int t, int nt, int& I, int& J) {
  I = 0;
  while (t >= nt - I) { t -= nt - I; ++I; }
  J = I + t;
}

// C = Cinit + op(A) op(B) on zero-padded [DP][LD] buffers.  op(A)[m][k] = TA ? A[k][m] : A[m][k]; op(B)[k][n] = TB ? B[n][k]
// : B[k][n], times kscale[k] when SCALE.  SYM: the product is symmetric - upper macro tiles only, mirrored on store.
// Lane l of a wave feeds A[m = l & 15][k = l >> 4], B[k = l >> 4][n = l & 15] and owns C[4 (l >> 4) + r][l & 15].
//
// A wave owns one MACRO tile of 2 x 2 instruction tiles: an operand fetched from LDS feeds two instructions (LDS traffic is
// what bounds a one-tile-per-wave loop: 2 x 256 bytes per 32-cycle instruction and SIMD) and the four accumulators are
// independent chains (40 cycles of dependent latency against 32 of issue).  nt <= 7 gives at most 16 macro tiles, one per
// wave; they are dealt out by size (full 2 x 2 tiles, then the 1 x 2 / 2 x 1 edges of an odd nt, then the corner) in
// boustrophedon order over the four SIMDs - wave w runs on SIMD w & 3 - so that the matrix pipes get equal shares.
// The contraction covers k4 steps of four indices (k4 = ceil(D / 4)): chunks of four steps, the operands of the next chunk
// in flight while the current one is on the pipe; the steps beyond the last full chunk take indices 4 ks + lk.
// the k loop of one macro tile; R2 / C2: the macro tile has a second tile row / column
template <bool TA, bool TB, bool SCALE, bool R2, bool C2>
__device__ __forceinline__ void gemm_macro(const float* pa0, const float* pb0, const float* kscale, int k4, int LD, int lk, f4 (&acc)[2][2]) {
  const int as = TA ? 4 * LD : 4, bs = TB ? 4 : 4 * LD;
  const float* pa1 = pa0 + (TA ? 16 : 16 * LD);
  const float* pb1 = pb0 + (TB ? 16 * LD : 16);
  float av[2][2][4], bv[2][2][4];                               // [buffer][tile row / column][step]
#define HTA_LOAD_STEP(buf, u, ks)                                                        \
  do {                                                                                   \
    av[buf][0][u] = pa0[(ks) * as];                                                      \
    if (R2) av[buf][1][u] = pa1[(ks) * as];                                              \
    const float sc__ = SCALE ? kscale[4 * (ks) + lk] : 1.f;                              \
    bv[buf][0][u] = SCALE ? pb0[(ks) * bs] * sc__ : pb0[(ks) * bs];                      \
    if (C2) bv[buf][1][u] = SCALE ? pb1[(ks) * bs] * sc__ : pb1[(ks) * bs];              \
  } while (0)
#if defined(HTA_GEMM_ABLATE) && HTA_GEMM_ABLATE == 1      // developer build (tools/scratch/metric_phase.cpp): no matrix instructions
#define HTA_MMA_STEP(buf, u)                                                                                              \
  do {                                                                                                                    \
    acc[0][0][0] = fmaf(av[buf][0][u], bv[buf][0][u], acc[0][0][0]);                                                      \
    if (C2) acc[0][1][0] = fmaf(av[buf][0][u], bv[buf][1][u], acc[0][1][0]);                                              \
    if (R2) acc[1][0][0] = fmaf(av[buf][1][u], bv[buf][0][u], acc[1][0][0]);                                              \
    if (R2 && C2) acc[1][1][0] = fmaf(av[buf][1][u], bv[buf][1][u], acc[1][1][0]);                                        \
  } while (0)
#else
#define HTA_MMA_STEP(buf, u)                                                                                              \
  do {                                                                                                                    \
    acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[buf][0][u], bv[buf][0][u], acc[0][0], 0, 0, 0);                   \
    if (C2) acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[buf][0][u], bv[buf][1][u], acc[0][1], 0, 0, 0);           \
    if (R2) acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[buf][1][u], bv[buf][0][u], acc[1][0], 0, 0, 0);           \
    if (R2 && C2) acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[buf][1][u], bv[buf][1][u], acc[1][1], 0, 0, 0);     \
  } while (0)
#endif
  // A chunk = 16 contraction indices = four instructions.  Inside a chunk, lane group lk takes the indices 16 c + 4 lk + u at
  // step u (not 16 c + 4 u + lk: any assignment serves as long as both operands use it): an operand that is contiguous
  // in k (A[m][k], B[n][k], the scale vector) arrives as ONE 16-byte read per chunk, and an operand that is strided in k
  // (A[k][m], B[k][n]) is read from rows 4 lk + u - with LD = 4 mod 8 the four lane groups of an instruction fall on four
  // different bank quarters.  (Rounds 2-3 read rows 4 u + lk: groups 0 / 1 of LD = 116 overlapped in four banks, and the
  // k-contiguous operand took four 4-byte reads with rows li and li + 8 in the same banks: every operand read of every
  // product was a two-way bank conflict, and LDS time - not the matrix pipe - bounded the products.)
  const float* qa0 = pa0 + (TA ? 3 * lk * LD : 3 * lk);
  const float* qa1 = pa1 + (TA ? 3 * lk * LD : 3 * lk);
  const float* qb0 = pb0 + (TB ? 3 * lk : 3 * lk * LD);
  const float* qb1 = pb1 + (TB ? 3 * lk : 3 * lk * LD);
  const float* qsc = SCALE ? kscale + 4 * lk : nullptr;
#define HTA_LOAD_CHUNK(buf, c)                                                                                            \
  do {                                                                                                                    \
    if (TA) {                                                                                                             \
      _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                                     \
        av[buf][0][u] = qa0[(16 * (c) + u) * LD];                                                                         \
        if (R2) av[buf][1][u] = qa1[(16 * (c) + u) * LD];                                                                 \
      }                                                                                                                   \
    } else {                                                                                                              \
      const f4 t0__ = *reinterpret_cast<const f4*>(qa0 + 16 * (c));                                                       \
      _Pragma("unroll") for (int u = 0; u < 4; ++u) av[buf][0][u] = t0__[u];                                              \
      if (R2) {                                                                                                           \
        const f4 t1__ = *reinterpret_cast<const f4*>(qa1 + 16 * (c));                                                     \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) av[buf][1][u] = t1__[u];                                            \
      }                                                                                                                   \
    }                                                                                                                     \
    f4 sc__ = f4{1.f, 1.f, 1.f, 1.f};                                                                                     \
    if (SCALE) sc__ = *reinterpret_cast<const f4*>(qsc + 16 * (c));                                                       \
    if (TB) {                                                                                                             \
      const f4 t0__ = *reinterpret_cast<const f4*>(qb0 + 16 * (c));                                                       \
      _Pragma("unroll") for (int u = 0; u < 4; ++u) bv[buf][0][u] = SCALE ? t0__[u] * sc__[u] : t0__[u];                  \
      if (C2) {                                                                                                           \
        const f4 t1__ = *reinterpret_cast<const f4*>(qb1 + 16 * (c));                                                     \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) bv[buf][1][u] = SCALE ? t1__[u] * sc__[u] : t1__[u];                \
      }                                                                                                                   \
    } else {                                                                                                              \
      _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                                     \
        bv[buf][0][u] = SCALE ? qb0[(16 * (c) + u) * LD] * sc__[u] : qb0[(16 * (c) + u) * LD];                            \
        if (C2) bv[buf][1][u] = SCALE ? qb1[(16 * (c) + u) * LD] * sc__[u] : qb1[(16 * (c) + u) * LD];                    \
      }                                                                                                                   \
    }                                                                                                                     \
  } while (0)
#if defined(HTA_GEMM_ABLATE) && HTA_GEMM_ABLATE == 2      // developer build: no operand reads (the matrix instructions alone)
#undef HTA_LOAD_CHUNK
#define HTA_LOAD_CHUNK(buf, c)                                                                                            \
  do {                                                                                                                    \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                                       \
      av[buf][0][u] = av[buf][1][u] = __int_as_float(0x3f800000 + (c));                                                   \
      bv[buf][0][u] = bv[buf][1][u] = __int_as_float(0x3f800000 + lk);                                                    \
    }                                                                                                                     \
  } while (0)
#endif
  const int nchunk = k4 >> 2;
  if (nchunk > 0) {
    HTA_LOAD_CHUNK(0, 0);
    int ch = 0;
    for (; ch + 2 < nchunk; ch += 2) {
      HTA_LOAD_CHUNK(1, ch + 1);
#pragma unroll
      for (int u = 0; u < 4; ++u) HTA_MMA_STEP(0, u);
      HTA_LOAD_CHUNK(0, ch + 2);
#pragma unroll
      for (int u = 0; u < 4; ++u) HTA_MMA_STEP(1, u);
    }
    if (ch + 1 < nchunk) {
      HTA_LOAD_CHUNK(1, ch + 1);
#pragma unroll
      for (int u = 0; u < 4; ++u) HTA_MMA_STEP(0, u);
#pragma unroll
      for (int u = 0; u < 4; ++u) HTA_MMA_STEP(1, u);
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) HTA_MMA_STEP(0, u);
    }
  }
  for (int ks = 4 * nchunk; ks < k4; ++ks) {
    HTA_LOAD_STEP(0, 0, ks);
    HTA_MMA_STEP(0, 0);
  }
#undef HTA_LOAD_STEP
#undef HTA_LOAD_CHUNK
#undef HTA_MMA_STEP
}

// (Operands are OFFSETS, in floats, into the kernel's dynamic LDS block: an out-of-line function only sees generic pointers
// in its arguments - flat loads, every wait a full one; and its integer arguments arrive in vector registers: readfirstlane
// makes them scalars again so that the tile bookkeeping compiles to scalar branches.)
// LDC: the leading dimension as a compile-time constant (0: the run-time value) - with it the four rows of a chunk's operand
// reads are immediate offsets of one address register instead of an address computation per read
template <bool TA, bool TB, bool SYM, bool SCALE, int LDC>
__device__ HTA_PH_ATTR void lds_gemm_ld(int offA, int offB, int offC, int offCinit, int offScale, int nt, int k4, int LDr) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* const lds = reinterpret_cast<float*>(smem_raw);
  nt = __builtin_amdgcn_readfirstlane(nt); k4 = __builtin_amdgcn_readfirstlane(k4);
  const int LD = LDC ? LDC : __builtin_amdgcn_readfirstlane(LDr);
  const float* A = lds + __builtin_amdgcn_readfirstlane(offA);
  const float* B = lds + __builtin_amdgcn_readfirstlane(offB);
  float* C = lds + __builtin_amdgcn_readfirstlane(offC);
  offCinit = __builtin_amdgcn_readfirstlane(offCinit);
  const float* Cinit = offCinit >= 0 ? lds + offCinit : nullptr;
  const float* kscale = SCALE ? lds + __builtin_amdgcn_readfirstlane(offScale) : nullptr;
  if (SYM && SCALE && TA) { HTA_WSTAMP(25); HTA_WVSTAMP(0); }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);        // wave-uniform: scalar branches below
  const int li = lane & 15, lk = lane >> 4;
  int I0, J0;
  bool r2, c2;
  if (SYM) {
    // upper block triangle only, as 1 x 2 macro tiles (pairs of neighbours in a tile row) and the odd tile that ends a row of odd
    // length: nt = 7 gives 12 pairs + 4 singles = one item per wave, pairs first - every SIMD (wave & 3) gets three pairs and a
    // single, 7 of the 28 tiles.  (Round 3 dealt 2 x 2 macro tiles here too: 10 items for 16 waves, three full ones - 12 tiles -
    // on one SIMD: a symmetric product took as long as a full one.)
    int npairs = 0;
    for (int I = 0; I < nt; ++I) npairs += (nt - I) >> 1;
    int w = wave, I = 0;
    if (w < npairs) {
      for (;; ++I) { const int pr = (nt - I) >> 1; if (w < pr) break; w -= pr; }
      I0 = I; J0 = I + 2 * w; c2 = true;
    } else {
      w -= npairs;
      for (; I < nt; ++I) if ((nt - I) & 1) { if (w == 0) break; --w; }
      if (I >= nt) return;
      I0 = I; J0 = nt - 1; c2 = false;
    }
    r2 = false;
  } else {
    const int nf = nt >> 1, odd = nt & 1;
    const int s = wave & 3, q = wave >> 2;
    int k = (q & 1) ? 4 * q + 3 - s : 4 * q + s;                  // position in the size-sorted macro list
    int r, c;
    const int nfull = nf * nf;
    if (k < nfull) { r = k / nf; c = k - r * nf; }
    else {
      k -= nfull;
      if (!odd || k > 2 * nf) return;
      if (k < nf) { r = k; c = nf; } else if (k < 2 * nf) { r = nf; c = k - nf; } else { r = nf; c = nf; }
    }
    I0 = 2 * r; J0 = 2 * c;
    r2 = I0 + 1 < nt; c2 = J0 + 1 < nt;
  }
  const float* pa0 = TA ? A + lk * LD + 16 * I0 + li : A + (16 * I0 + li) * LD + lk;
  const float* pb0 = TB ? B + (16 * J0 + li) * LD + lk : B + lk * LD + 16 * J0 + li;
  f4 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      acc[x][y] = f4{0.f, 0.f, 0.f, 0.f};
      if (Cinit && (x == 0 || r2) && (y == 0 || c2)) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[x][y][t] = Cinit[(16 * (I0 + x) + 4 * lk + t) * LD + 16 * (J0 + y) + li];
      }
    }
  if (SYM && SCALE && TA) { HTA_WSTAMP(26); HTA_WVSTAMP(1); }
  if (r2 && c2) gemm_macro<TA, TB, SCALE, true, true>(pa0, pb0, kscale, k4, LD, lk, acc);
  else if (r2) gemm_macro<TA, TB, SCALE, true, false>(pa0, pb0, kscale, k4, LD, lk, acc);
  else if (c2) gemm_macro<TA, TB, SCALE, false, true>(pa0, pb0, kscale, k4, LD, lk, acc);
  else gemm_macro<TA, TB, SCALE, false, false>(pa0, pb0, kscale, k4, LD, lk, acc);
  if (SYM && SCALE && TA) { HTA_WSTAMP(27); HTA_WVSTAMP(2); }
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      if ((x == 1 && !r2) || (y == 1 && !c2)) continue;
      const int I = I0 + x, J = J0 + y;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        C[(16 * I + 4 * lk + t) * LD + 16 * J + li] = acc[x][y][t];
        if (SYM && I != J) C[(16 * J + li) * LD + 16 * I + 4 * lk + t] = acc[x][y][t];
      }
    }
}

constexpr int kLdCfg3 = 116;             // LD of 97 <= D <= 112 (BASELINE config 3's D = 100)
template <bool TA, bool TB, bool SYM, bool SCALE>
__device__ __forceinline__ void lds_gemm(int offA, int offB, int offC, int offCinit, int offScale, int nt, int k4, int LD) {
  if (LD == kLdCfg3) lds_gemm_ld<TA, TB, SYM, SCALE, kLdCfg3>(offA, offB, offC, offCinit, offScale, nt, k4, LD);
  else lds_gemm_ld<TA, TB, SYM, SCALE, 0>(offA, offB, offC, offCinit, offScale, nt, k4, LD);
}

// Cross-lane sums / maxima as DPP operands of the adding instruction (quad permutes, row_half_mirror, row_mirror; the four rows
// of a wave through v_readlane): __shfl_xor compiles to ds_bpermute_b32 - an LDS round trip per butterfly step, six in a row for
// a wave sum (~700 cycles; a block reduction in this file cost ~1.9 k, measured).  Every lane gets the result.
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float lane_f(float v, int lane) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane)); }
__device__ __forceinline__ float sum8_dpp(float v) {            // aligned groups of 8 lanes
  v += dpp_f<0xB1>(v);                                          // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);                                          // quad_perm [2,3,0,1]
  v += dpp_f<0x141>(v);                                         // row_half_mirror: the other quad of the 8
  return v;
}
__device__ __forceinline__ float sum16_dpp(float v) { v = sum8_dpp(v); v += dpp_f<0x140>(v); return v; }      // row_mirror: the other half of the 16
__device__ __forceinline__ float max16_dpp(float v) {
  v = fmaxf(v, dpp_f<0xB1>(v)); v = fmaxf(v, dpp_f<0x4E>(v)); v = fmaxf(v, dpp_f<0x141>(v)); v = fmaxf(v, dpp_f<0x140>(v));
  return v;
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v = sum16_dpp(v);
  return (lane_f(v, 0) + lane_f(v, 16)) + (lane_f(v, 32) + lane_f(v, 48));
}
__device__ __forceinline__ float wave_max_dpp(float v) {
  v = max16_dpp(v);
  return fmaxf(fmaxf(lane_f(v, 0), lane_f(v, 16)), fmaxf(lane_f(v, 32), lane_f(v, 48)));
}
static_assert(MT / 64 == 16, "the block reductions below fold the 16 wave results in one DPP row");
__device__ __forceinline__ float block_sum_dpp(float v, float* red) {
  v = wave_sum_dpp(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return sum16_dpp(red[threadIdx.x & 15]);
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max_dpp(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return max16_dpp(red[threadIdx.x & 15]);
}

// out[row] = sum_k M(row, k) v[k] with 8 lanes per row (rows 0 .. 127 of the workgroup's 1024 threads); M(row, k) =
// TRANS ? M[k * ld + row] : M[row * ld + k]; n = vector length (rows and columns).  Every lane of a row's group returns the sum.
// Row-wise (TRANS = false) the 8 lanes of a row read it as 16-byte quads - quad seg + 8 u of the row and of v, zero padded to a
// multiple of 4 - : 8 LDS instructions per lane instead of 28 (the 4-byte form is bound by the LDS instruction rate: 448
// wave-reads of 2+ clocks for a 40 KB matrix - 3 k cycles per product, measured; the quads: ~1 k).  Column-wise (TRANS) keeps
// 4-byte reads of 8 consecutive rows per k.
template <bool TRANS> __device__ __forceinline__ float mv8(const float* M, int ld, const float* v, int n) {
  const int row = threadIdx.x >> 3, seg = threadIdx.x & 7;
  float acc0 = 0.f, acc1 = 0.f;
  if (!TRANS) {
    if (row < n) {
      const int nq = (n + 3) >> 2;
      const float* mr = M + row * ld;
      f4 mq[4], vq[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = seg + 8 * u;
        const bool on = q < nq;
        mq[u] = on ? *reinterpret_cast<const f4*>(mr + 4 * q) : f4{0.f, 0.f, 0.f, 0.f};
        vq[u] = on ? *reinterpret_cast<const f4*>(v + 4 * q) : f4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc0 = fmaf(mq[u][0], vq[u][0], acc0); acc1 = fmaf(mq[u][1], vq[u][1], acc1);
        acc0 = fmaf(mq[u][2], vq[u][2], acc0); acc1 = fmaf(mq[u][3], vq[u][3], acc1);
      }
    }
  } else if (row < n) {
    // n <= 112: 14 steps of 8, seven operand pairs in flight at a time
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float mv[7], vv[7];
#pragma unroll
      for (int u = 0; u < 7; ++u) {
        const int k = seg + 8 * (7 * h + u);
        const bool on = k < n;
        mv[u] = on ? (TRANS ? M[k * ld + row] : M[row * ld + k]) : 0.f;
        vv[u] = on ? v[k] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 7; ++u) { if (u & 1) acc1 = fmaf(mv[u], vv[u], acc1); else acc0 = fmaf(mv[u], vv[u], acc0); }
    }
  }
  float acc = acc0 + acc1;
  return sum8_dpp(acc);
}

// y = M v for a SYMMETRIC dense row-major [n][n] matrix in global memory (P; L2 resident): thread (column c = tid & 127,
// slice q = tid >> 7) sums M[k][c] v[k] over k = q, q + 8, ... - consecutive lanes read consecutive addresses, all 14 loads
// of a thread are in flight together, no cross-lane reduction.  The 8 slice partials meet in `part` ([8][128] floats of
// LDS); after the barrier inside, y[c] = sum_q part[q][c] is returned to the threads tid < 128 (0 elsewhere).
__device__ __forceinline__ float gmv_sym(const __attribute__((address_space(1))) float* M, const float* v, int n, float* part) {
  const int c = threadIdx.x & 127, q = threadIdx.x >> 7;
  float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float mv[7];
#pragma unroll
    for (int u = 0; u < 7; ++u) {
      const int k = q + 8 * (7 * h + u);
      mv[u] = (c < n && k < n) ? M[k * n + c] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 7; ++u) {
      const int k = q + 8 * (7 * h + u);
      const float vk = k < n ? v[k] : 0.f;
      if (u & 1) acc1 = fmaf(mv[u], vk, acc1); else acc0 = fmaf(mv[u], vk, acc0);
    }
  }
  part[q * 128 + c] = acc0 + acc1;
  __syncthreads();
  float y = 0.f;
  if (threadIdx.x < 128) {
#pragma unroll
    for (int t = 0; t < 8; ++t) y += part[t * 128 + threadIdx.x];
  }
  return y;
}

// the same for a lower-triangular M: out[row] = sum_{k <= row} M[row][k] v[k]  (p = L z)
__device__ __forceinline__ float mv8_lower(const float* M, int ld, const float* v, int n) {
  const int row = threadIdx.x >> 3, seg = threadIdx.x & 7;
  float acc = 0.f;
  if (row < n)
    for (int k = seg; k <= row; k += 8) acc = fmaf(M[row * ld + k], v[k], acc);
  return sum8_dpp(acc);
}

// a zero-padded [DP][LD] copy of a dense row-major [D][D] matrix in global memory
__device__ __forceinline__ void stage_dense(const __attribute__((address_space(1))) float* src, float* dst, int D, int DP, int LD) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int h = 0; h < 2; ++h) {                              // DP <= 112, LD <= 116: seven loads of a thread in flight at a time
    float v[7];
    const int j = lane + 64 * h;
#pragma unroll
    for (int t = 0; t < 7; ++t) { const int i = wave + 16 * t; v[t] = (i < D && j < D) ? src[i * D + j] : 0.f; }
#pragma unroll
    for (int t = 0; t < 7; ++t) { const int i = wave + 16 * t; if (i < DP && j < LD) dst[i * LD + j] = v[t]; }
  }
}

// Right-looking Cholesky of the symmetric [D][D] matrix in G (leading dimension LD, zero padded to DP) in panels of 16
// columns: the 16 x 16 diagonal block is factored and inverted by one wave (lane = row of the block; a column step is a
// broadcast of the pivot row through LDS), the panel below it is L21 = A21 inv(L11)^T and the trailing matrix loses
// L21 L21^T, both as MFMA tiles (K = 16: four instructions per tile).  The factor replaces the lower triangle; W is a
// [16][20] scratch block.  A non-positive pivot yields NaN, as the reference's cholesky raises.
__device__ __forceinline__ float lane_bcast(float x, int lane) {       // v_readlane_b32: `lane` is uniform (a compile-time constant here)
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lane));
}

__device__ HTA_PH_ATTR void mfma_cholesky(int offG, int D, int DP, int LD, int offW) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* const G = reinterpret_cast<float*>(smem_raw) + __builtin_amdgcn_readfirstlane(offG);
  float* const W = reinterpret_cast<float*>(smem_raw) + __builtin_amdgcn_readfirstlane(offW);
  D = __builtin_amdgcn_readfirstlane(D); DP = __builtin_amdgcn_readfirstlane(DP); LD = __builtin_amdgcn_readfirstlane(LD);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
        float d = lane_bcast(row[j], j);                       // pivot
        d = cj ? sqrtf(d) : 1.f;
        const float inv = 1.f / d;
        float lij = (lane == j) ? d : row[j] * inv;            // column j of L (rows >= j)
        if (!cj) lij = (lane == j) ? 1.f : 0.f;
        if (lane < j) lij = 0.f;
        row[j] = lij;
#pragma unroll
        for (int k = j + 1; k < 16; ++k) {
          const float lkj = lane_bcast(lij, k);                // L[k][j]
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
        for (int k = 0; k < i; ++k) acc -= lane_bcast(row[k], i) * invc[k];
        invc[i] = acc / lane_bcast(row[i], i);
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

// ---- the phases of an evaluation, each OUT OF LINE ---------------------------------------------------------------------
// Inlined into one kernel body, the compiler hoists every phase's per-thread address arithmetic over the whole kernel and
// keeps it alive across the others: > 128 live registers at 1024 threads, i.e. scratch traffic inside the loops.  As
// functions nothing is live across a phase but a few scalars.  LDS operands are offsets (see lds_gemm); global operands
// are cast back to the global address space (a generic pointer argument would become flat loads).
typedef const __attribute__((address_space(1))) float* gcf;
#define HTA_LDS_BASE() extern __shared__ __attribute__((aligned(16))) char smem_raw[]; float* const lds = reinterpret_cast<float*>(smem_raw)
#define HTA_U(x) __builtin_amdgcn_readfirstlane(x)

__device__ HTA_PH_ATTR void jacobi_fallback(int offA, int offVT, int D, int ne, int LD, int offCs, int offRed, int max_sweeps) {
  HTA_LDS_BASE();
  D = HTA_U(D); ne = HTA_U(ne); LD = HTA_U(LD); max_sweeps = HTA_U(max_sweeps);
  lds_jacobi<float, 2, 2>(lds + HTA_U(offA), lds + HTA_U(offVT), D, ne, LD, LD, lds + HTA_U(offCs), nullptr, lds + HTA_U(offRed), max_sweeps);
}

__device__ HTA_PH_ATTR void ph_stage(const float* src, int offDst, int D, int DP, int LD) {
  HTA_LDS_BASE();
  stage_dense((gcf)src, lds + HTA_U(offDst), HTA_U(D), HTA_U(DP), HTA_U(LD));
}

// sum_i d_i (P d)_i over the block, upd_g += cg P d  (d in LDS at offVd; P symmetric, global)
__device__ HTA_PH_ATTR float ph_logp(const float* P, int offVd, int D, int offPart, int offRed, float* upd_g, float cg) {
  HTA_LDS_BASE();
  D = HTA_U(D);
  const float* vd = lds + HTA_U(offVd);
  const float pd = gmv_sym((gcf)P, vd, D, lds + HTA_U(offPart));
  float part = 0.f;
  if ((int)threadIdx.x < D) {
    part = vd[threadIdx.x] * pd;
    if (upd_g) { __attribute__((address_space(1))) float* g = (__attribute__((address_space(1))) float*)upd_g; g[threadIdx.x] += cg * pd; }
  }
  return block_sum_dpp(part, lds + HTA_U(offRed));
}

// M (or M^T) v with M [n][ld] and v in LDS; every lane of row (tid >> 3)'s group of 8 returns the row's sum
__device__ HTA_PH_ATTR float ph_mv8(int trans, int offM, int ld, int offV, int n) {
  HTA_LDS_BASE();
  ld = HTA_U(ld); n = HTA_U(n);
  const float* M = lds + HTA_U(offM); const float* v = lds + HTA_U(offV);
  return HTA_U(trans) ? mv8<true>(M, ld, v, n) : mv8<false>(M, ld, v, n);
}

// One pass of the refinement's element-wise step: lam_i = S_ii / Gm_ii, then E from S (at offS) and Gm (at offG; the
// identity when have_x == 0) into offDst (which may be offS).  Returns max |E_ij|, i != j (1 for NaN / inf / > kFallbackE).
__device__ HTA_PH_ATTR float ph_refine_E(int offS, int offG, int offDst, int offLam, int offRed, int have_x, int D, int LD) {
  HTA_LDS_BASE();
  D = HTA_U(D); LD = HTA_U(LD); have_x = HTA_U(have_x);
  const float* by = lds + HTA_U(offS); const float* bz = lds + HTA_U(offG);
  float* edst = lds + HTA_U(offDst); float* vlam = lds + HTA_U(offLam); float* red = lds + HTA_U(offRed);
  const int tid = threadIdx.x;
  float scale = 0.f;
  if (tid < D) {
    const float l = have_x ? by[tid * LD + tid] / bz[tid * LD + tid] : by[tid * LD + tid];
    vlam[tid] = l;
    scale = fabsf(l);
  }
  scale = block_max(scale, red);                                   // (its barriers also publish vlam)
  const float tiny = 8.f * Eps<float>::v * scale;
  float emax = 0.f;
  // The off-diagonal elements in PAIRS (i, j), (j, i), i < j: S and Gm are symmetric (the products mirror their upper tiles), the
  // two quotients share the reciprocal of lam_j - lam_i.  Row r and row D - 1 - r have D - 1 upper elements between them: the
  // pairs form a [ceil(D / 2)][D - 1] rectangle - thread (c = tid & 127, r = tid >> 7 + 8 p) needs no index division.
  // (Rounds 2-3 walked all D^2 elements, a division each: 12.8 k + 10.7 k of an evaluation's 108 k cycles.)
  const int c = tid & 127, H = (D + 1) >> 1;
  if (tid < D) {                                                   // the diagonal: E_ii = (1 - Gm_ii) / 2 (+ 1: the first pass writes X = I + E)
    const float gm = have_x ? bz[tid * LD + tid] : 1.f;
    edst[tid * LD + tid] = 0.5f * (1.f - gm) + (have_x ? 0.f : 1.f);
  }
#pragma unroll
  for (int p8 = 0; p8 < 7; ++p8) {
    const int r = (tid >> 7) + 8 * p8;
    if (r >= H || c >= D - 1) continue;
    const int n1 = D - 1 - r;
    int i, j;
    if (c < n1) { i = r; j = r + 1 + c; }
    else { i = D - 1 - r; j = D - r + (c - n1); if (i == r) continue; }          // (odd D: the middle row is its own partner)
    const float li = vlam[i], lj = vlam[j];
    const float sij = by[i * LD + j];
    const float gm = have_x ? bz[i * LD + j] : 0.f;
    const float rinv = __builtin_amdgcn_rcpf(lj - li);
    const float nu = sij - lj * gm, nl = sij - li * gm;
    const float eu = (fabsf(nu) <= tiny) ? -0.5f * gm : nu * rinv;                // E_ij = (S_ij - lam_j Gm_ij) / (lam_j - lam_i)
    const float el = (fabsf(nl) <= tiny) ? -0.5f * gm : -(nl * rinv);             // E_ji = (S_ij - lam_i Gm_ij) / (lam_i - lam_j)
    const float em = fmaxf(fabsf(eu), fabsf(el));
    emax = fmaxf(emax, em);
    if (!(fabsf(eu) <= kFallbackE) || !(fabsf(el) <= kFallbackE)) emax = 1.f;     // NaN / inf / too large
    edst[i * LD + j] = eu;
    edst[j * LD + i] = el;
  }
  return block_max(emax, red);
}

// The SECOND pass in closed form (round 4).  After the first pass X = I + E1 with E1 antisymmetric (the pairs above) and zero on the
// diagonal, the refinement's quantities are, up to terms of third order in the perturbation F = A - diag(A):
//     S_ij - lam_j Gm_ij = (F E1)_ij =: M_ij  (i != j),    lam_i' = S_ii / Gm_ii = lam_i + M_ii,    Gm_ii = 1 + sum_k E1_ki^2
// (the first-order terms cancel by the choice of E1; (E1^T Lam E1 - lam_j E1^T E1)_ij = -(E1^T F)_ij cancels one of the two
// mixed terms) - i.e. second-order perturbation theory: ONE full product M = F E1 instead of T = A X, S = X^T T and Gm = X^T X
// (2.1 full products), with E2_ij = M_ij / (lam_j' - lam_i'), E2_ii = -1/2 sum_k E1_ki^2.  The caller zeroed the diagonals of A and
// X before the product; this pass reads M (offM) and E1 (offX, whose unit diagonal it restores), writes E2 to offDst, E2^T in M's
// place (a pair owns its two entries of M: the solve then reads both E2 and E2^T row-wise) and the corrected eigenvalues to offLam.  Returns max |E2_ij| (1 for NaN / inf / > kFallbackE).  Truncation: |A X2 - X2 Lam'| ~ |F| d^2.
__device__ HTA_PH_ATTR float ph_refine_E2(int offM, int offX, int offDst, int offLam, int offRed, int D, int LD) {
  HTA_LDS_BASE();
  D = HTA_U(D); LD = HTA_U(LD);
  float* M = lds + HTA_U(offM); float* X = lds + HTA_U(offX);
  float* edst = lds + HTA_U(offDst); float* vlam = lds + HTA_U(offLam); float* red = lds + HTA_U(offRed);
  const int tid = threadIdx.x;
  float scale = 0.f;
  {
    const int row = tid >> 3, seg = tid & 7;                       // sum_k E1_ki^2 = sum_k E1_ik^2: 8 lanes per row
    float c0 = 0.f, c1 = 0.f;
    if (row < D) {
#pragma unroll
      for (int u = 0; u < 14; ++u) {
        const int k = seg + 8 * u;
        const float v = k < D ? X[row * LD + k] : 0.f;
        if (u & 1) c1 = fmaf(v, v, c1); else c0 = fmaf(v, v, c0);
      }
    }
    float cs = c0 + c1;
    cs = sum8_dpp(cs);
    if (seg == 0 && row < D) {
      const float l2 = vlam[row] + M[row * LD + row];
      vlam[row] = l2;
      scale = fabsf(l2);
      edst[row * LD + row] = -0.5f * cs;
      M[row * LD + row] = -0.5f * cs;
      X[row * LD + row] = 1.f;
    }
  }
  scale = block_max(scale, red);                                   // (its barriers also publish vlam)
  // M is a product of small factors, not a difference of large ones: its rounding noise is RELATIVE (eps |F| d per term), far below
  // the full pass's threshold 8 eps |lam| on S_ij - lam_j Gm_ij (which would zero M itself: |M_ij| ~ 1e-6 at BASELINE config 3)
  const float tiny = 8.f * Eps<float>::v * scale * kSecondE;
  float emax = 0.f;
  const int c = tid & 127, H = (D + 1) >> 1;
#pragma unroll
  for (int p8 = 0; p8 < 7; ++p8) {
    const int r = (tid >> 7) + 8 * p8;
    if (r >= H || c >= D - 1) continue;
    const int n1 = D - 1 - r;
    int i, j;
    if (c < n1) { i = r; j = r + 1 + c; }
    else { i = D - 1 - r; j = D - r + (c - n1); if (i == r) continue; }
    const float rinv = __builtin_amdgcn_rcpf(vlam[j] - vlam[i]);
    const float nu = M[i * LD + j], nl = M[j * LD + i];
    const float eu = (fabsf(nu) <= tiny) ? 0.f : nu * rinv;
    const float el = (fabsf(nl) <= tiny) ? 0.f : -(nl * rinv);
    emax = fmaxf(emax, fmaxf(fabsf(eu), fabsf(el)));
    if (!(fabsf(eu) <= kFallbackE) || !(fabsf(el) <= kFallbackE)) emax = 1.f;
    edst[i * LD + j] = eu;
    edst[j * LD + i] = el;
    M[i * LD + j] = el;                                            // E2^T
    M[j * LD + i] = eu;
  }
  return block_max(emax, red);
}

__device__ HTA_PH_ATTR void ph_chol_solve(int offG, int D, int LD, int offV) {
  HTA_LDS_BASE();
  lds_chol_solve<float>(lds + HTA_U(offG), HTA_U(D), HTA_U(LD), lds + HTA_U(offV));
}

__device__ HTA_PH_ATTR float ph_mv8_lower(int offM, int ld, int offV, int n) {
  HTA_LDS_BASE();
  return mv8_lower(lds + HTA_U(offM), HTA_U(ld), lds + HTA_U(offV), HTA_U(n));
}

// a zero-padded [DP][LD] copy of the symmetric matrix whose LOWER triangle is in global memory (eigh UPLO = 'L', S:119), jitter
// (LDS vector at offJit) added on the diagonal
__device__ HTA_PH_ATTR void ph_stage_sym(const float* src, int offDst, int offJit, int D, int DP, int LD) {
  HTA_LDS_BASE();
  D = HTA_U(D); DP = HTA_U(DP); LD = HTA_U(LD);
  float* dst = lds + HTA_U(offDst); const float* vj = lds + HTA_U(offJit);
  gcf g = (gcf)src;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float v[7];
    const int j = lane + 64 * h;
#pragma unroll
    for (int t = 0; t < 7; ++t) {
      const int i = wave + 16 * t;
      v[t] = (i < D && j < D) ? (i >= j ? g[i * D + j] : g[j * D + i]) : 0.f;
    }
#pragma unroll
    for (int t = 0; t < 7; ++t) { const int i = wave + 16 * t; if (i < DP && j < LD) dst[i * LD + j] = v[t] + ((i == j && i < D) ? vj[i] : 0.f); }
  }
}

// [D][D] row-major global <- the leading block of an LDS matrix
__device__ HTA_PH_ATTR void ph_store_dense(float* dstg, int offSrc, int D, int LD) {
  HTA_LDS_BASE();
  D = HTA_U(D); LD = HTA_U(LD);
  const float* src = lds + HTA_U(offSrc);
  __attribute__((address_space(1))) float* g = (__attribute__((address_space(1))) float*)dstg;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = wave; i < D; i += MT / 64)
    for (int j = lane; j < D; j += 64) g[i * D + j] = src[i * LD + j];
}

// W of the derivative matrix M = Q W Q^T (HtaMetricArgs::dmetric_out; the same formula as metric_eval_kernel):
// W_kl = 1/2 [k == l] lam~'_k / lam~_k - 1/2 J_kl u_k u_l,  J = divided differences of lam -> lam~, zero padded
__device__ HTA_PH_ATTR void ph_dmetric_w(int offDst, int offLam, int offLt, int offU, float alpha, int D, int DP, int LD) {
  HTA_LDS_BASE();
  D = HTA_U(D); DP = HTA_U(DP); LD = HTA_U(LD);
  float* dst = lds + HTA_U(offDst);
  const float* vlam = lds + HTA_U(offLam); const float* vlt = lds + HTA_U(offLt); const float* vu = lds + HTA_U(offU);
  const int l = threadIdx.x & 127;
  for (int k = threadIdx.x >> 7; k < DP; k += MT / 128) {
    if (l >= LD) continue;
    float w = 0.f;
    if (k < D && l < D) {
      const float lk = vlam[k], ll = vlam[l], dl = lk - ll;
      float J;
      if (k == l || fabsf(dl) <= 1e-3f * (fabsf(lk) + fabsf(ll))) J = softabs_slope<float>(alpha, 0.5f * (lk + ll));
      else J = (vlt[k] - vlt[l]) / dl;
      w = -0.5f * J * vu[k] * vu[l];
      if (k == l) w += 0.5f * softabs_slope<float>(alpha, lk) / vlt[k];
    }
    dst[k * LD + l] = w;
  }
}

// The thread index, opaque to the optimiser.  Per-lane global addresses (a.m + b D + i, a.upd_x + b D + row, ...) derived from
// the plain index were computed at the top of the kernel and kept across its 30 out-of-line phase calls - i.e. spilled to
// scratch memory at the 128-register cap of a 1024-thread workgroup (14 stores at the top, 13 reloads scattered over the
// phases, each a round trip beyond the L2: profiles/r02zz WRITE_SIZE 20.7 MB per launch against 0.2 MB of outputs).
// Deriving them from an opaque copy at the point of use keeps them out of the calls' live ranges.
__device__ __forceinline__ int opaque_tid() {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));
  return t;
}

// One evaluation of system b (everything of the file's header); the workgroup's 1024 threads, state in the dynamic LDS block.
// `vres`: the matrix buffer (offset) that holds the staged shared basis V0 on entry, or -1; on return, the buffer that holds it
// now (a solve ends with V0 staged for x = V0 x': the next evaluation of a trajectory kernel starts from that copy), or -1.
__device__ __forceinline__ void metric_warm_system(const MetricArgsT<float>& a, int DP, int LD, int64_t b, int& vres, bool second) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int D = a.D, tid = threadIdx.x;
  const int nt = DP / 16, k4 = (D + 3) / 4;
  const int ne = D + (D & 1);
  float* const lds0 = reinterpret_cast<float*>(smem_raw);
  const int BS = DP * LD > 1024 ? DP * LD : 1024;             // (gmv_sym parks 8 x 128 slice partials in a matrix buffer)
  const int oB0 = 0, oB1 = BS, oB2 = 2 * BS;
  const int oJit = 3 * BS;           // e = jitter * u                          (later: Jacobi (c, s) pairs)
  const int oLam = oJit + DP;        // eigenvalues of Hs
  const int oLt = oLam + DP;         // soft-abs eigenvalues
  const int oM = oLt + DP;           // m, then m' = V0^T m
  const int oY = oM + DP;            // w = (X^T m') / lam~
  const int oX = oY + DP;            // x' = X w
  const int oD = oX + DP;            // d = X - mu / z
  const int oRed = oD + 2 * DP;      // MT / 64
  const int oW = oRed + MT / 64;     // [16][20] panel scratch
  float* vjit = lds0 + oJit; float* vlam = lds0 + oLam; float* vlt = lds0 + oLt; float* vm = lds0 + oM; float* vy = lds0 + oY;
  float* vx = lds0 + oX; float* vd = lds0 + oD; float* red = lds0 + oRed;
  const bool softabs = a.metric == 1;
  // GENERAL mode (round 3): every system has its own curvature Hs_b AND its own approximate eigenbasis V0_b (the caller's
  // previous evaluation at that chain): A = V0_b^T (Hs_b + diag(e)) V0_b by two products, the refinement where its coupling
  // test passes, the in-launch Jacobi (on the nearly diagonal A: few sweeps) where it does not; V_out = V0_b X is the
  // basis for the caller's next call.  lam0 is not used.
  const bool general = softabs && a.hs_stride != 0;

  {
    const uint64_t chain = a.chain_offset + (uint64_t)b;
    const float* V0b = a.V0 + (general ? b * a.v0_stride : 0);
    __syncthreads();
    HTA_STAMP(0);
    // ---- 0. operands: jitter, the solve vector, d = X - mu; V0 into LDS
    if (tid < DP) {
      const int i = opaque_tid();
      vjit[i] = (i < D && a.has_jitter) ? (float)a.jitter * uniform_elem<float>(a.seed, chain, a.draw, PURPOSE_JITTER, a.sub, i) : 0.f;
      vm[i] = (i < D && a.m) ? a.m[b * D + i] : 0.f;
      vd[i] = (i < D && a.X) ? a.X[b * D + i] - a.mu[i] : 0.f;
    }
    // buffer roles: bx = V0 (then X), by = A / S / E, bz = scratch
    int bx = oB1, by = oB0, bz = oB2;
    const bool resident = softabs && !general && vres >= 0;
    if (resident) { bx = vres; by = vres == oB0 ? oB1 : oB0; bz = vres == oB2 ? oB1 : oB2; }
    vres = -1;
    if (softabs && !resident) ph_stage(V0b, bx, D, DP, LD);
    __syncthreads();
    HTA_STAMP(1);
    // ---- Gaussian log-prob and P (X - mu)
    float logp = 0.f;
    if (a.X) logp = (float)a.log_norm - 0.5f * ph_logp(a.Pm, oD, D, bz, oRed, a.upd_g ? a.upd_g + b * D : nullptr, (float)a.cg);
    // ---- m' = V0^T m
    if (a.m && softabs) {
      const float v = ph_mv8(1, bx, LD, oM, D);
      __syncthreads();
      if ((tid & 7) == 0 && (tid >> 3) < DP) vm[tid >> 3] = ((tid >> 3) < D) ? v : 0.f;
    }
    HTA_STAMP(2);
    // ---- 1. A = diag(lam0) + V0^T diag(e) V0 into buffer 0 (symmetric, zero padded); general: A = V0^T (Hs + diag(e)) V0 into buffer 2
    if (softabs && !general) {
      lds_gemm<true, false, true, true>(bx, bx, by, -1, oJit, nt, k4, LD);
      HTA_WSTAMP(28); HTA_WVSTAMP(3);
      __syncthreads();
      HTA_WSTAMP(29);
      { const int i = opaque_tid(); if (i < D) lds0[by + i * LD + i] += a.lam0[i]; }
      HTA_WSTAMP(30);
    } else if (general) {
      ph_stage_sym(a.Hs + b * a.hs_stride, oB2, oJit, D, DP, LD);
      __syncthreads();
      lds_gemm<false, false, false, false>(oB2, oB1, oB0, -1, -1, nt, k4, LD);          // T = (Hs + diag e) V0
      __syncthreads();
      lds_gemm<true, false, true, false>(oB1, oB0, oB2, -1, -1, nt, k4, LD);            // A = V0^T T
      by = oB2; bz = oB0;
    }
    __syncthreads();
    HTA_STAMP(3);
    // ---- 2. eigenvectors X of A by iterative refinement from X = I; bx: X, by: A / S / E, bz: scratch
    bool have_x = false, converged = false, fallback = !softabs;     // Metric.HESSIAN: G = A, no decomposition needed
    float emax_prev = 1.f;
    bool implicit_e = false;          // the last update X (I + E) is applied to the vectors of the solve instead of being formed
    bool x_antisym = false;           // X = I + E1 straight from the first pass: X + X^T = 2 I exactly (the pairs), so X^T v = 2 v - X v
    bool et_in_bz = false;            // the closed-form second pass left E2^T in bz (next to E2 in by)
    const bool want_matrix = a.G_out || a.p_out || a.V_out || a.dmetric_out;
    if (softabs) {
      for (int it = 0; it < 4 && !converged && !fallback; ++it) {
        if (it >= 2 && general) { fallback = true; break; }          // (forming A again needs all three buffers: the Jacobi path does it)
        if (it >= 2) {                                               // rare: A was consumed by the previous pass, form it again
          ph_stage(a.V0, bz, D, DP, LD);
          __syncthreads();
          lds_gemm<true, false, true, true>(bz, bz, by, -1, oJit, nt, k4, LD);
          __syncthreads();
          if (tid < D) lds0[by + tid * LD + tid] += a.lam0[tid];
          __syncthreads();
        }
        HTA_STAMP(4 + 4 * it);
        float emax;
        if (it == 1 && have_x && second && emax_prev <= kSecondE) {
          // second pass in closed form: M = F E1 (F = A, E1 = X without their diagonals), then ph_refine_E2
          { const int i = opaque_tid(); if (i < D) { lds0[by + i * LD + i] = 0.f; lds0[bx + i * LD + i] = 0.f; } }
          __syncthreads();
          lds_gemm<false, false, false, false>(by, bx, bz, -1, -1, nt, k4, LD);       // M = F E1
          __syncthreads();
          HTA_STAMP(5 + 4 * it);
          emax = ph_refine_E2(bz, bx, by, oLam, oRed, D, LD);
          et_in_bz = true;
        } else {
          if (have_x) {
            lds_gemm<false, false, false, false>(by, bx, bz, -1, -1, nt, k4, LD);     // T = A X
            __syncthreads();
            lds_gemm<true, false, true, false>(bx, bz, by, -1, -1, nt, k4, LD);       // S = X^T T
            __syncthreads();
            lds_gemm<true, false, true, false>(bx, bx, bz, -1, -1, nt, k4, LD);       // Gm = X^T X
            __syncthreads();
          }
          HTA_STAMP(5 + 4 * it);
          // first pass: X = I + E goes next to A (still needed for A X); later passes: E in place of S
          emax = ph_refine_E(by, bz, have_x ? by : bx, oLam, oRed, have_x ? 1 : 0, D, LD);
        }
        emax_prev = emax;
        HTA_STAMP(6 + 4 * it);
        if (emax > kFallbackE) { fallback = true; break; }
        __syncthreads();
        if (have_x && !want_matrix && emax <= kConvE) {
          // the converging pass of an evaluation that only solves: X2 = X (I + E) enters through X2^T m' = (I + E^T) X^T m'
          // and X2 w = X (w + E w) - four matrix-vector products instead of one D^3 product (14.8 k of 116 k cycles at cfg3)
          implicit_e = true; converged = true;
          break;
        }
        if (have_x) {
          lds_gemm<false, false, false, false>(bx, by, bz, bx, -1, nt, k4, LD);       // X <- X + X E
          __syncthreads();
          const int t = bx; bx = bz; bz = t;
          x_antisym = false; et_in_bz = false;
        } else {
          have_x = true;                                                              // bx = I + E, by = A still
          x_antisym = true;
        }
        converged = emax <= kConvE;
        HTA_STAMP(7 + 4 * it);
      }
      if (!converged) fallback = true;
      if (fallback) {
        // cyclic Jacobi on A (rmhmc_metric_dev.hpp): no assumption on gaps or perturbation size
        __syncthreads();
        if (general) {                                               // A again: V0 -> bx, Hs -> bz, T -> by, A -> bz; then A lives in `by`
          ph_stage(V0b, bx, D, DP, LD);
          ph_stage_sym(a.Hs + b * a.hs_stride, bz, oJit, D, DP, LD);
          __syncthreads();
          lds_gemm<false, false, false, false>(bz, bx, by, -1, -1, nt, k4, LD);
          __syncthreads();
          lds_gemm<true, false, true, false>(bx, by, bz, -1, -1, nt, k4, LD);
          const int t = by; by = bz; bz = t;
        } else {
          ph_stage(a.V0, bz, D, DP, LD);
          __syncthreads();
          lds_gemm<true, false, true, true>(bz, bz, by, -1, oJit, nt, k4, LD);
          __syncthreads();
          if (tid < D) lds0[by + tid * LD + tid] += a.lam0[tid];
        }
        __syncthreads();
        for (int e = tid; e < DP * LD; e += MT) { const int i = e / LD, j = e - i * LD; lds0[bz + e] = (i == j && i < D) ? 1.f : 0.f; }
        __syncthreads();
        jacobi_fallback(by, bz, D, ne, LD, oJit, oRed, a.max_sweeps);
        if (tid < D) vlam[tid] = lds0[by + tid * LD + tid];
        for (int e = tid; e < DP * LD; e += MT) { const int i = e / LD, j = e - i * LD; lds0[bx + e] = (i < D && j < D) ? lds0[bz + j * LD + i] : 0.f; }   // X[i][k] = VT[k][i]
        __syncthreads();
      }
    }
    HTA_STAMP(20);
    // ---- 3. soft-abs map, log-determinant  (S:120, S:726)
    float logdet = 0.f, quad = 0.f;
    if (softabs) {
      float ld = 0.f;
      if (tid < DP) {
        const int i = opaque_tid();
        float lt = 1.f;
        if (i < D) {
          const float lam = vlam[i];
          lt = (1.f / tanhf((float)a.alpha * lam)) * lam;
          ld = logf(lt);
          if (a.lam_out) a.lam_out[b * D + i] = lt;
          if (a.lamraw_out) a.lamraw_out[b * D + i] = lam;
        }
        vlt[i] = lt;
      }
      logdet = block_sum_dpp(ld, red);
      HTA_STAMP(12);                                            // (12 .. 18: the solve's sub-phases; the passes it = 2, 3 reuse the slots when they run)
      // ---- 4. x = V0 X (X^T m' / lam~)
      if (a.m) {
        // y = X^T m': row-wise products only where the structure allows it (mv8: a third of the column-wise product's time)
        float y;
        const int row = opaque_tid() >> 3;
        if (x_antisym && !fallback) { const float xm = ph_mv8(0, bx, LD, oM, D); y = 2.f * vm[row < DP ? row : 0] - xm; }
        else y = ph_mv8(1, bx, LD, oM, D);
        HTA_STAMP(13);
        if (implicit_e) {                                       // y <- (I + E^T) y
          __syncthreads();
          if ((tid & 7) == 0 && row < DP) vy[row] = (row < D) ? y : 0.f;
          __syncthreads();
          y += et_in_bz ? ph_mv8(0, bz, LD, oY, D) : ph_mv8(1, by, LD, oY, D);
          __syncthreads();
        }
        HTA_STAMP(14);
        float qd = 0.f;
        float wreg = 0.f;
        if ((tid & 7) == 0 && row < DP) {
          const float w = (row < D) ? y / vlt[row] : 0.f;
          wreg = w;
          (implicit_e ? vx : vy)[row] = w;
          qd = (row < D) ? y * w : 0.f;
        }
        quad = block_sum_dpp(qd, red);
        HTA_STAMP(15);
        if (implicit_e) {                                       // w <- w + E w
          const float ew = ph_mv8(0, by, LD, oX, D);
          __syncthreads();
          if ((tid & 7) == 0 && row < DP) vy[row] = (row < D) ? wreg + ew : 0.f;
          __syncthreads();
        }
        HTA_STAMP(16);
        const float xp = ph_mv8(0, bx, LD, oY, D);
        __syncthreads();
        if ((tid & 7) == 0 && row < DP) vx[row] = (row < D) ? xp : 0.f;
        __syncthreads();
        HTA_STAMP(17);
        ph_stage(V0b, by, D, DP, LD);                           // (the E buffer is dead)
        __syncthreads();
        HTA_STAMP(18);
        if (!general) vres = by;
        const float x = ph_mv8(0, by, LD, oX, D);
        const int orow = opaque_tid() >> 3;
        if ((tid & 7) == 0 && orow < D) {
          if (a.x_out) a.x_out[b * D + orow] = x;
          if (a.upd_x) a.upd_x[b * D + orow] += (float)a.cx * x;
        }
      }
    }
    // ---- 4b. the eigenbasis itself (V_out: the next call's warm start) and the derivative matrix M = Q W Q^T (dmetric_out)
    bool q_ready = false;
    if (softabs && (a.V_out || a.dmetric_out)) {
      vres = -1;
      __syncthreads();
      if (!a.m && tid < DP) vy[tid] = 0.f;                            // u = Q^T m / lam~ (vy holds it after the solve)
      ph_stage(V0b, by, D, DP, LD);
      __syncthreads();
      lds_gemm<false, false, false, false>(by, bx, bz, -1, -1, nt, k4, LD);             // Q = V0 X
      __syncthreads();
      if (a.V_out) {
        // a system that went non-finite (a diverged chain: NaN curvature, or a NaN basis handed in) must not poison its NEXT
        // evaluation: its basis restarts from the identity
        if (!(fabsf(logdet) <= 3.0e38f)) {
          __attribute__((address_space(1))) float* vg = (__attribute__((address_space(1))) float*)(a.V_out + b * D * D);
          for (int e = tid; e < D * D; e += MT) vg[e] = (e / D == e % D) ? 1.f : 0.f;
        } else {
          ph_store_dense(a.V_out + b * D * D, bz, D, LD);
        }
      }
      if (a.dmetric_out) {
        ph_dmetric_w(by, oLam, oLt, oY, (float)a.alpha, D, DP, LD);
        __syncthreads();
        lds_gemm<false, true, false, false>(by, bz, bx, -1, -1, nt, k4, LD);            // T = W Q^T
        __syncthreads();
        lds_gemm<false, false, true, false>(bz, bx, by, -1, -1, nt, k4, LD);            // M = Q T (symmetric)
        __syncthreads();
        ph_store_dense(a.dmetric_out + b * D * D, by, D, LD);
        __syncthreads();
      } else {
        q_ready = true;                                               // bz = Q, bx = X still: step 5 must not stage V0 again (V_out may alias it)
      }
    }
    HTA_STAMP(21);
    // ---- 5. G = Q diag(lam~) Q^T, Q = V0 X  (S:121) for fisher() / the momentum draw; Metric.HESSIAN: G = Hs itself
    if (a.G_out || a.p_out || !softabs) {
      vres = -1;
      __syncthreads();
      int g = bz;
      if (softabs) {
        if (!q_ready) {
          ph_stage(V0b, by, D, DP, LD);
          __syncthreads();
          lds_gemm<false, false, false, false>(by, bx, bz, -1, -1, nt, k4, LD);       // Q = V0 X
        }
        __syncthreads();
        lds_gemm<false, true, true, true>(bz, bz, by, -1, oLt, nt, k4, LD);           // G = Q (diag(lam~) Q^T)
        g = by;
      } else {
        const float* Hs = a.Hs + b * a.hs_stride;
        for (int e = tid; e < DP * LD; e += MT) {
          const int i = e / LD, j = e - i * LD;
          float v = 0.f;
          if (i < D && j < D) { v = (i >= j) ? Hs[i * D + j] : Hs[j * D + i]; if (i == j) v += vjit[i]; }
          lds0[g + e] = v;
        }
      }
      __syncthreads();
      if (a.G_out) for (int e = tid; e < D * D; e += MT) { const int i = e / D, j = e - i * D; a.G_out[b * D * D + e] = lds0[g + i * LD + j]; }
      HTA_STAMP(22);
      if (a.p_out || !softabs) {
        mfma_cholesky(g, D, DP, LD, oW);
        HTA_STAMP(23);
        if (!softabs) {
          float ld = 0.f;
          if (tid < D) ld = 2.f * logf(lds0[g + tid * LD + tid]);                     // slogdet (S:728) for a PD metric
          logdet = block_sum_dpp(ld, red);
          if (a.m) {
            if (tid < D) { vy[tid] = a.m[b * D + tid]; vx[tid] = vy[tid]; }
            ph_chol_solve(g, D, LD, oY);
            float qd = 0.f;
            if (tid < D) {
              qd = vx[tid] * vy[tid];
              if (a.x_out) a.x_out[b * D + tid] = vy[tid];
              if (a.upd_x) a.upd_x[b * D + tid] += (float)a.cx * vy[tid];
            }
            quad = block_sum_dpp(qd, red);
          }
        }
        if (a.p_out) {                   // p = L z  (S:184 via MultivariateNormal.rsample)
          __syncthreads();
          if (tid < DP) vd[tid] = (tid < D) ? normal_elem<float>(a.seed, chain, a.draw, 0, tid) : 0.f;
          __syncthreads();
          const float p = ph_mv8_lower(g, LD, oD, D);
          const int prow = opaque_tid() >> 3;
          if ((tid & 7) == 0 && prow < D) a.p_out[b * D + prow] = p;
        }
      }
    }
    HTA_STAMP(24);
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

__global__ __launch_bounds__(MT) void metric_warm_mfma_kernel(MetricArgsT<float> a, int DP, int LD, int second) {
  for (int64_t b = blockIdx.x; b < a.B; b += gridDim.x) { int vres = -1; metric_warm_system(a, DP, LD, b, vres, second != 0); }
}

// One explicit-RMHMC trajectory of chain b in ONE launch (S:969-989 with S:425-461 inside): the 4 L + 3 metric evaluations of
// rmhmc_explicit.hip's launch sequence - momentum draw (sub-stream 0), H_old (1), per step the four half steps (2 + 8 l + {1, 2,
// 4, 7}) with the binding rotation between the second and the third, H_new (2 + 8 L) - run back to back by the chain's
// workgroup.  A chain's evaluations depend on nothing but that chain's own rows of th / pm / thc / pmc: the sequence needs no
// grid-wide step, only the workgroup barrier between an evaluation's row updates and the next evaluation's reads.  Same
// evaluation code as metric_warm_mfma_kernel (metric_warm_system), same arithmetic: bit-identical to the launch sequence
// (tests/test_gpu_rmhmc.py::test_trajectory_kernel_equals_the_launch_sequence); what goes away is 57 launches per trajectory
// with their ramps and the gaps between them (profiles/r04q: 12 % of the step).
__global__ __launch_bounds__(MT) void metric_traj_mfma_kernel(MetricArgsT<float> a, MetricTrajArgs t, int DP, int LD, int second) {
  const int D = a.D;
  const int nops = 4 * t.L + 3;
  for (int64_t b = blockIdx.x; b < a.B; b += gridDim.x) {
    int vres = -1;                       // the buffer a solve left the staged V0 in: the next evaluation starts from it
    for (int op = 0; op < nops; ++op) {
      MetricArgsT<float> o = a;
      int j = -1;
      if (op == 0) { o.sub = 0; o.p_out = t.pm; }                                                  // gibbs: p ~ N(0, G(theta))  S:183-184
      else if (op == 1) { o.sub = 1; o.X = t.cur; o.m = t.pm; o.H_out = t.H0; }                    // H_old  S:971
      else if (op == nops - 1) { o.sub = 2u + 8u * (uint32_t)t.L; o.X = t.th; o.m = t.pm; o.H_out = t.H1; o.logp_out = t.lp1; }   // H_new  S:989
      else {
        const int q = op - 2, l = q >> 2;
        j = q & 3;
        const bool fa = j == 0 || j == 3;                                                          // phi_A/2 (S:429-430, S:457-458) : phi_B/2
        o.sub = 2u + 8u * (uint32_t)l + (j == 0 ? 1u : j == 1 ? 2u : j == 2 ? 4u : 7u);
        o.X = fa ? t.th : t.thc; o.m = fa ? t.pmc : t.pm; o.upd_x = fa ? t.thc : t.th; o.upd_g = fa ? t.pm : t.pmc;
        o.cx = t.eh; o.cg = -t.eh;
      }
      metric_warm_system(o, DP, LD, b, vres, second != 0);
      if (op == 1 || j == 1) {
        __syncthreads();
        const int i = opaque_tid();
        if (i < D) {
          const int64_t e = b * D + i;
          if (op == 1) {                                                                           // S:425-426
            const float x = t.cur[e];
            t.th[e] = x; t.thc[e] = x; t.pmc[e] = t.pm[e];
          } else {
            phi_c_elem<float>(t.th[e], t.pm[e], t.thc[e], t.pmc[e], t.c, t.s);                     // phi_C  S:447-450
          }
        }
      }
    }
  }
}

int g_metric_traj = 1;   // tuning key "metric_traj": 1 = a trajectory of the eigendecomposition route is one launch, 0 = one launch per evaluation

bool metric_traj_mfma_eligible(const MetricArgsT<float>& a) {
  return g_metric_traj && a.hs_stride == 0 && !a.V_out && !a.dmetric_out && !a.G_out && metric_warm_mfma_eligible(a);
}

int metric_traj_mfma(const MetricArgsT<float>& a, const MetricTrajArgs& t, hipStream_t s) {
  const int D = a.D;
  const int DP = (D + 15) / 16 * 16, LD = DP + 4;
  const size_t lds = ((size_t)3 * (DP * LD > 1024 ? DP * LD : 1024) + 8 * DP + MT / 64 + 16 * 20) * sizeof(float);
  HTA_REQUIRE(lds <= 160 * 1024, "hta_rmhmc_gaussian_sample (trajectory kernel): D=%d does not fit the LDS", D);
  MetricArgsT<float> k = a;
  if (k.max_sweeps <= 0) k.max_sweeps = 16;
  static DevOnce done;
  if (!done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&metric_traj_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { set_error("hta_rmhmc_gaussian_sample: hipFuncSetAttribute: %s", hipGetErrorString(e)); return HTA_ERR_LAUNCH; }
    done = true;
  }
  const int grid = (int)(a.B < 65536 ? a.B : 65536);
  profile_begin(s);
  note_route("metric_traj_mfma_kernel");
  metric_traj_mfma_kernel<<<grid, MT, lds, s>>>(k, t, DP, LD, g_metric_second);
  profile_end(s);
  HTA_CHECK_LAUNCH("hta_rmhmc_gaussian_sample (trajectory kernel)");
  return HTA_OK;
}

bool metric_warm_mfma_eligible(const MetricArgsT<float>& a) {
  if (!g_metric_mfma || a.D < 1 || a.D > 112 || a.L_out) return false;
  if (a.metric == 1) {
    if (a.hs_stride == 0) return a.V0 && a.lam0 && !a.dmetric_out && !a.V_out;      // soft-abs: evaluations that share an eigenbasis
    // per-system curvature with per-system bases (a general target's chains, each warm-started from its previous evaluation);
    // G_out / p_out together with dmetric_out is not a combination the kernel keeps X for
    return g_metric_general && a.V0 && a.v0_stride != 0 && !(a.dmetric_out && (a.G_out || a.p_out));
  }
  return a.metric == 0 && !a.dmetric_out && !a.V_out;                 // Metric.HESSIAN: Cholesky + solve, any curvature input
}

int metric_warm_mfma(const MetricArgsT<float>& a, hipStream_t s) {
  const int D = a.D;
  const int DP = (D + 15) / 16 * 16, LD = DP + 4;
  const size_t lds = ((size_t)3 * (DP * LD > 1024 ? DP * LD : 1024) + 8 * DP + MT / 64 + 16 * 20) * sizeof(float);   // = oW + 320 floats
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
  note_route("metric_warm_mfma_kernel");
  metric_warm_mfma_kernel<<<grid, MT, lds, s>>>(k, DP, LD, g_metric_second);
  profile_end(s);
  HTA_CHECK_LAUNCH("hta_metric_eval (mfma)");
  return HTA_OK;
}

}  // namespace hta
