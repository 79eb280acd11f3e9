#define HTA_PHILOX_MAD64 1      // philox.hpp: 64-bit products (this file's kernels have the registers for them)
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
// Round 6: the SOLVE evaluations of a Gaussian target on the shared basis - 4 L + 2 of a trajectory's 4 L + 3 - and, by default, its momentum
// draw (p = G^(1/2) z, solve-shaped) run a reorganised sequence of the same mathematics (metric_fast_solve and the ph_fast_* phases below: V0
// resident, the element-wise passes in the products' epilogues, the second-order product as three bfloat16 products on pre-split operands,
// vector phases on 8 waves, the trajectory's state resident in LDS in eigen-coordinates): 35 k cycles per evaluation instead of 71 k.  What
// follows in this header describes the GENERAL sequence (metric_warm_system), which still serves everything else: per-system curvature and
// bases, outputs that need G or Q, Metric.HESSIAN, and whatever the fast sequence declines.
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
int g_metric_bx3 = 2;       // tuning key "metric_bx3": 2 = the fast solve's formation AND its second-pass product F E1 as three bfloat16 products of operands split hi + lo, 1 = F E1 only (the formation in exact fp32), 0 = exact fp32 products
int g_metric_sqrtdraw = 1;  // tuning key "metric_sqrtdraw": 1 = the momentum draw of soft-abs evaluations on a shared basis is p = G^(1/2) z (the symmetric square root: a SOLVE-shaped
                            // evaluation - same law as S:183-184's chol(G) z, no assembly of G, no Cholesky), 0 = the reference's map chol(G) z
int g_metric_general = 1;   // tuning key "metric_general": 1 = evaluations with per-system curvature AND per-system bases run here too

typedef float f4 __attribute__((ext_vector_type(4)));

#ifndef HTA_TIMING
#define HTA_TIMING 0
#endif
#if HTA_TIMING      // developer builds (tools/scratch/metric_phase.cpp): s_memtime stamps of workgroup 0 at the phase boundaries
__device__ long long hta_metric_dbg[32];
__device__ long long hta_metric_wdbg[16][16];       // per wave of workgroup 0: entry / k loop begins / k loop ends / return of the formation product (0 .. 3: general sequence; 4 .. 9 / 10 .. 15: ph_fast_form / ph_fast_second)
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
  float scv[2][4];                                              // the chunk's scale factors: applied where the operand is consumed (applied
                                                                // where it is loaded, the multiply - and with it the wait for the NEXT chunk's reads -
                                                                // sits in front of the CURRENT chunk's matrix instructions: no read was ever in flight)
#define HTA_LOAD_STEP(buf, u, ks)                                                        \
  do {                                                                                   \
    av[buf][0][u] = pa0[(ks) * as];                                                      \
    if (R2) av[buf][1][u] = pa1[(ks) * as];                                              \
    const float sc__ = SCALE ? kscale[4 * (ks) + lk] : 1.f;                              \
    bv[buf][0][u] = SCALE ? pb0[(ks) * bs] * sc__ : pb0[(ks) * bs];                      \
    if (C2) bv[buf][1][u] = SCALE ? pb1[(ks) * bs] * sc__ : pb1[(ks) * bs];              \
    scv[buf][u] = 1.f;                                                                   \
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
    const float b0__ = SCALE ? bv[buf][0][u] * scv[buf][u] : bv[buf][0][u];                                               \
    const float b1__ = (SCALE && C2) ? bv[buf][1][u] * scv[buf][u] : bv[buf][1][u];                                       \
    acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[buf][0][u], b0__, acc[0][0], 0, 0, 0);                            \
    if (C2) acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[buf][0][u], b1__, acc[0][1], 0, 0, 0);                    \
    if (R2) acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[buf][1][u], b0__, acc[1][0], 0, 0, 0);                    \
    if (R2 && C2) acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[buf][1][u], b1__, acc[1][1], 0, 0, 0);              \
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
    if (SCALE) {                                                                                                          \
      const f4 sc__ = *reinterpret_cast<const f4*>(qsc + 16 * (c));                                                       \
      _Pragma("unroll") for (int u = 0; u < 4; ++u) scv[buf][u] = sc__[u];                                                \
    }                                                                                                                     \
    if (TB) {                                                                                                             \
      const f4 t0__ = *reinterpret_cast<const f4*>(qb0 + 16 * (c));                                                       \
      _Pragma("unroll") for (int u = 0; u < 4; ++u) bv[buf][0][u] = t0__[u];                                              \
      if (C2) {                                                                                                           \
        const f4 t1__ = *reinterpret_cast<const f4*>(qb1 + 16 * (c));                                                     \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) bv[buf][1][u] = t1__[u];                                            \
      }                                                                                                                   \
    } else {                                                                                                              \
      _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                                     \
        bv[buf][0][u] = qb0[(16 * (c) + u) * LD];                                                                         \
        if (C2) bv[buf][1][u] = qb1[(16 * (c) + u) * LD];                                                                 \
      }                                                                                                                   \
    }                                                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
  } while (0)
#if defined(HTA_GEMM_ABLATE) && HTA_GEMM_ABLATE == 2      // developer build: no operand reads (the matrix instructions alone)
#undef HTA_LOAD_CHUNK
#define HTA_LOAD_CHUNK(buf, c)                                                                                            \
  do {                                                                                                                    \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                                       \
      av[buf][0][u] = av[buf][1][u] = __int_as_float(0x3f800000 + (c));                                                   \
      bv[buf][0][u] = bv[buf][1][u] = __int_as_float(0x3f800000 + lk);                                                    \
      scv[buf][u] = 1.f;                                                                                                  \
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
      __builtin_amdgcn_sched_barrier(0);
      HTA_LOAD_CHUNK(0, ch + 2);
#pragma unroll
      for (int u = 0; u < 4; ++u) HTA_MMA_STEP(1, u);
      __builtin_amdgcn_sched_barrier(0);
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

// The same macro tile for M = F E1 with the operands split on the fly into two bfloat16 terms each (x = hi + lo + O(2^-16 x)) and three
// products on v_mfma_f32_16x16x32_bf16 (hi hi + hi lo + lo hi, fp32 accumulation): 48 cycles of the matrix pipe per 32 contraction
// indices instead of 256.  M is a second-order CORRECTION - |E2| = |M| / gap <= 1e-4 where this pass is taken (kSecondE) - so its
// relative error 2^-16 is an absolute error below 2e-9 in the eigenvectors: under fp32 rounding of the first-order terms, which stay
// exact fp32 products (formation, and every matrix-vector product of the solve).  Both operands are read along k: A = F[m][k]
// (symmetric), B = E1[k][n] = -E1[n][k] (antisymmetric: the caller negates the result); lane l feeds row / column l & 15 and the
// indices 8 (l >> 4) .. + 7 of a 32-step (16-byte reads, rows 464 bytes apart: conflict free), 4 (l >> 4) .. + 3 of the 16-step that
// ends an odd number of tiles.
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
typedef short s4v __attribute__((ext_vector_type(4)));
typedef int i4v __attribute__((ext_vector_type(4)));
typedef int i2v __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair(float x0, float x1, int& hi, int& lo) {
  const unsigned a = __float_as_uint(x0), b = __float_as_uint(x1);
  hi = (int)__builtin_amdgcn_perm(b, a, 0x07060302);                     // the upper halves: truncation to bfloat16
  const f2v x = {x0, x1}, h = {__uint_as_float(a & 0xffff0000u), __uint_as_float(b & 0xffff0000u)};
  const f2v r = x - h;
  lo = (int)__builtin_amdgcn_perm(__float_as_uint(r[1]), __float_as_uint(r[0]), 0x07060302);
}
// (F arrives already split: the formation's epilogue stores it as two bfloat16 planes [DP][DP] (hi, then lo; rows 2 DP bytes apart - with an odd
// tile count the 16 lanes of a ds_read_b128 group fall on 16 different quads of the bank row) in the buffer fp32 F would take.  Measured
// (HTA_GEMM_ABLATE builds): a k loop's time is the SUM of its matrix instructions' time and its other instructions' time, not the larger -
// every split saved is time saved.)
// A 1 x NB strip of the second product: ONE tile of E1 rows (operand A, fp32, split on the fly ONCE per 32 indices) against NB <= 4 tiles
// of F's planes (operand B) - the 2 x 2 macro tile split its two A tiles for two B tiles each, and the split is what that loop's time is
// made of.  acc[y] = sum_k E1[j][k] F[i_y][k] = -M[i_y][j].
// NT: the tile count as a compile-time constant (the planes exist only where the leading dimension is kLdCfg3, i.e. 7 tiles): the k loop
// unrolls completely - every operand read is an immediate offset of ONE address register per operand and the next step's A tile lands in
// fresh registers (the rolled loop spent 18 address additions and 4 register-pair copies per step next to its 12 matrix instructions).
template <int NB, int NT>
__device__ __forceinline__ void gemm_strip_bx3(const float* pa0, const unsigned short* pb0, f4 (&acc)[4]) {
  constexpr int nt = NT;
  const int kg = (threadIdx.x & 63) >> 4;
  constexpr int DPh = 16 * nt, plane = DPh * DPh;
  const float* pa = pa0 + 8 * kg;
  const unsigned short* pb = pb0 + 8 * kg;
  constexpr int nfull = nt >> 1;
  f4 ra0, ra1;
  if (nfull > 0) { ra0 = *reinterpret_cast<const f4*>(pa); ra1 = *reinterpret_cast<const f4*>(pa + 4); }
#pragma unroll
  for (int s = 0; s < nfull; ++s) {
    // (B tiles two at a time: all four at once put the function beyond the caller-saved registers)
    i4v bh[2], bl[2];
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      if (y >= NB) continue;
      bh[y] = *reinterpret_cast<const i4v*>(pb + y * 16 * DPh + 32 * s);
      bl[y] = *reinterpret_cast<const i4v*>(pb + y * 16 * DPh + plane + 32 * s);
    }
    i4v ah, al;
    {
      int h, l;
      split_pair(ra0[0], ra0[1], h, l); ah[0] = h; al[0] = l;
      split_pair(ra0[2], ra0[3], h, l); ah[1] = h; al[1] = l;
      split_pair(ra1[0], ra1[1], h, l); ah[2] = h; al[2] = l;
      split_pair(ra1[2], ra1[3], h, l); ah[3] = h; al[3] = l;
    }
    __builtin_amdgcn_sched_barrier(0);
    if (s + 1 < nfull) { ra0 = *reinterpret_cast<const f4*>(pa + 32 * (s + 1)); ra1 = *reinterpret_cast<const f4*>(pa + 32 * (s + 1) + 4); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
      for (int y = 0; y < 2; ++y) {
        if (y >= NB) continue;
        acc[y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8v, pr == 2 ? al : ah), __builtin_bit_cast(bf8v, pr == 1 ? bl[y] : bh[y]), acc[y], 0, 0, 0);
      }
    __builtin_amdgcn_sched_barrier(0);
    if (NB > 2) {
#pragma unroll
      for (int y = 2; y < 4; ++y) {
        if (y >= NB) continue;
        bh[y - 2] = *reinterpret_cast<const i4v*>(pb + y * 16 * DPh + 32 * s);
        bl[y - 2] = *reinterpret_cast<const i4v*>(pb + y * 16 * DPh + plane + 32 * s);
      }
#pragma unroll
      for (int pr = 0; pr < 3; ++pr)
#pragma unroll
        for (int y = 2; y < 4; ++y) {
          if (y >= NB) continue;
          acc[y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8v, pr == 2 ? al : ah), __builtin_bit_cast(bf8v, pr == 1 ? bl[y - 2] : bh[y - 2]), acc[y], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (nt & 1) {
    const int k0 = 16 * (nt - 1) + 4 * kg - 8 * kg;
    const f4 v = *reinterpret_cast<const f4*>(pa + k0);
    i2v th, tl;
    { int h, l; split_pair(v[0], v[1], h, l); th[0] = h; tl[0] = l; split_pair(v[2], v[3], h, l); th[1] = h; tl[1] = l; }
#pragma unroll
    for (int y = 0; y < NB; ++y) {
      const i2v uh = *reinterpret_cast<const i2v*>(pb + y * 16 * DPh + k0), ul = *reinterpret_cast<const i2v*>(pb + y * 16 * DPh + plane + k0);
      acc[y] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s4v, th), __builtin_bit_cast(s4v, uh), acc[y], 0, 0, 0);
      acc[y] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s4v, th), __builtin_bit_cast(s4v, ul), acc[y], 0, 0, 0);
      acc[y] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s4v, tl), __builtin_bit_cast(s4v, uh), acc[y], 0, 0, 0);
    }
  }
}

// Both operands pre-split (the formation F = W W^T on the planes of W^T = (diag(sqrt e) V0)^T, round 6): a row tile against one or two
// row tiles, three products per tile and 32 indices, no vector instruction in the loop.  pa / pb: row starts of the hi plane.
template <bool C2, int NT>
__device__ __forceinline__ void gemm_macro_pp(const unsigned short* pa0, const unsigned short* pb0, f4 (&acc)[2][2]) {
  constexpr int nt = NT;                                                  // (compile-time tile count: see gemm_strip_bx3)
  const int kg = (threadIdx.x & 63) >> 4;
  constexpr int DPh = 16 * nt, plane = DPh * DPh;
  const unsigned short* pa = pa0 + 8 * kg;
  const unsigned short* pb[2] = {pb0 + 8 * kg, pb0 + 16 * DPh + 8 * kg};
  constexpr int nfull = nt >> 1;
  i4v ah, al, bh[2], bl[2];
#define HTA_PP_LOAD(s)                                                                                   \
  do {                                                                                                   \
    ah = *reinterpret_cast<const i4v*>(pa + 32 * (s)); al = *reinterpret_cast<const i4v*>(pa + plane + 32 * (s)); \
    bh[0] = *reinterpret_cast<const i4v*>(pb[0] + 32 * (s)); bl[0] = *reinterpret_cast<const i4v*>(pb[0] + plane + 32 * (s)); \
    if (C2) { bh[1] = *reinterpret_cast<const i4v*>(pb[1] + 32 * (s)); bl[1] = *reinterpret_cast<const i4v*>(pb[1] + plane + 32 * (s)); } \
  } while (0)
  if (nfull > 0) HTA_PP_LOAD(0);
#pragma unroll
  for (int s = 0; s < nfull; ++s) {
    const i4v cah = ah, cal = al, cbh0 = bh[0], cbl0 = bl[0], cbh1 = bh[1], cbl1 = bl[1];
    __builtin_amdgcn_sched_barrier(0);
    if (s + 1 < nfull) HTA_PP_LOAD(s + 1);
    __builtin_amdgcn_sched_barrier(0);
    acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8v, cah), __builtin_bit_cast(bf8v, cbh0), acc[0][0], 0, 0, 0);
    if (C2) acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8v, cah), __builtin_bit_cast(bf8v, cbh1), acc[0][1], 0, 0, 0);
    acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8v, cah), __builtin_bit_cast(bf8v, cbl0), acc[0][0], 0, 0, 0);
    if (C2) acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8v, cah), __builtin_bit_cast(bf8v, cbl1), acc[0][1], 0, 0, 0);
    acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8v, cal), __builtin_bit_cast(bf8v, cbh0), acc[0][0], 0, 0, 0);
    if (C2) acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8v, cal), __builtin_bit_cast(bf8v, cbh1), acc[0][1], 0, 0, 0);
  }
#undef HTA_PP_LOAD
  if (nt & 1) {                                                          // the last 16 indices
    const int k0 = 16 * (nt - 1) + 4 * kg - 8 * kg;
    const i2v th = *reinterpret_cast<const i2v*>(pa + k0), tl = *reinterpret_cast<const i2v*>(pa + plane + k0);
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      if (y == 1 && !C2) continue;
      const i2v uh = *reinterpret_cast<const i2v*>(pb[y] + k0), ul = *reinterpret_cast<const i2v*>(pb[y] + plane + k0);
      acc[0][y] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s4v, th), __builtin_bit_cast(s4v, uh), acc[0][y], 0, 0, 0);
      acc[0][y] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s4v, th), __builtin_bit_cast(s4v, ul), acc[0][y], 0, 0, 0);
      acc[0][y] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s4v, tl), __builtin_bit_cast(s4v, uh), acc[0][y], 0, 0, 0);
    }
  }
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
constexpr int kNtCfg3 = (kLdCfg3 - 4) / 16;      // its tile count: 7
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

// Maxima of NON-NEGATIVE values (|x|, or 0) as unsigned maxima of their bit patterns: the DPP operand folds into v_max_u32 - one instruction per
// butterfly step where fmaxf takes three (a v_mov_dpp and a canonicalising v_max of each operand); a NaN orders above every number and survives.
template <int CTRL> __device__ __forceinline__ unsigned dpp_u(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true); }
__device__ __forceinline__ unsigned max16_u(unsigned v) {
  v = max(v, dpp_u<0xB1>(v)); v = max(v, dpp_u<0x4E>(v)); v = max(v, dpp_u<0x141>(v)); v = max(v, dpp_u<0x140>(v));
  return v;
}
__device__ __forceinline__ float wave_max_nn(float x) {
  const unsigned v = max16_u(__float_as_uint(x));
  return __uint_as_float(max(max((unsigned)__builtin_amdgcn_readlane((int)v, 0), (unsigned)__builtin_amdgcn_readlane((int)v, 16)),
                             max((unsigned)__builtin_amdgcn_readlane((int)v, 32), (unsigned)__builtin_amdgcn_readlane((int)v, 48))));
}
// max |v_i| over a vector of n <= 128 LDS floats whose padding entries are zero (no predicate: the index is clamped)
__device__ __forceinline__ float wave_max_abs(const float* v, int n) {
  const int lane = threadIdx.x & 63;
  const unsigned a = __float_as_uint(v[min(lane, n - 1)]) & 0x7fffffffu, b = __float_as_uint(v[min(lane + 64, n - 1)]) & 0x7fffffffu;
  return wave_max_nn(__uint_as_float(max(a, b)));
}
// a block maximum (non-negative values) with ONE barrier, for call sites whose previous use of `red` is already fenced by a barrier every wave has passed
__device__ __forceinline__ float block_max1(float v, float* red) {
  v = wave_max_nn(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return __uint_as_float(max16_u(__float_as_uint(red[threadIdx.x & 15])));
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

// ======== Round 6: the SOLVE evaluation of a Gaussian target on the shared basis - 4 L + 2 of a trajectory's 4 L + 3 =============
// Same mathematics as the general sequence in metric_warm_system (formation, first pass from X = I, second pass in closed form applied
// to the vectors), reorganised around what the phase table of round 5 showed (profiles/r05ad: 71 k cycles, of which the two
// products' k loops are 19.5 k): (1) V0 never leaves its buffer - X = I + E1 is not stored over it, E1 and E2 live in the other two,
// so nothing is staged again (3.0 k + the first staging); (2) the element-wise passes run in the EPILOGUES of the two products on
// the accumulators (no LDS round trip of S / M, no pair bookkeeping, 16-byte mirror stores: 6.5 k + 8.0 k -> ~4 k); (3) log p and
// P d come from the eigenbasis the workgroup already holds - d' = V0^T d rides in the pass that computes m' = V0^T m, d^T P d =
// sum lam0 d'^2, P d = V0 (lam0 d') rides in the pass that computes x = V0 x' - instead of a pass over P in global memory with an
// L2 read-modify-write behind it (8.3 k -> ~2 k); (4) the three block sums (log-det, quadratic form, d^T P d) share the chain's
// barriers.  Anything this path does not cover (per-system curvature, outputs that need G or Q, a first pass above kSecondE, a second
// pass above kConvE) returns false BEFORE any global write and the general sequence runs, V0 still resident.
constexpr int kFastScratch = 320 + 4 * 112;   // floats at oW: the Cholesky's [16][20] panel scratch (the fast solve parks its [10][4] wave partials there), then the
                                              // trajectory kernel's four state vectors in eigen-coordinates (kFastState: theta', p', theta~', p~', DP floats each)
constexpr int kFastState = 320;

// Cost model of the vector phases (measured, round 6: profiles/r06i_metric_fast_phases.txt): with 16 waves on the CU every instruction a
// wave executes costs the workgroup ~16 cycles (4 waves per SIMD x 4 cycles of issue; the ONE scalar unit of the CU serves all 16 waves),
// whatever it does - a matrix-vector product spread over all 1024 threads is bound by its address / predicate / reduction instructions,
// not by its 12.5 k multiply-adds.  So these phases run on the first 8 waves only, 4 lanes per row with 32 multiply-adds each as packed
// FMAs; the other waves go straight to the next barrier.  Lane c of a row takes the 16-byte quads 4c .. 4c+3 and 16+4c .. 16+4c+3: the 16
// lanes of a ds_read_b128 group (4 rows x 4 lanes, rows 29 quads apart) then fall on 16 different quads of the bank row.
__device__ __forceinline__ f2v pk_fma(f2v a, f2v b, f2v c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ float quad_sum(float s) { s += dpp_f<0xB1>(s); s += dpp_f<0x4E>(s); return s; }

// (M v)_row, row = tid >> 2, tid < 512; nq = quads per row (DP / 4, a multiple of 4); every lane of the row's quad returns the sum
__device__ __forceinline__ float mv4(const float* M, int ld, const float* v, int nq, int nrow) {
  const int row = threadIdx.x >> 2, c = threadIdx.x & 3;
  f2v a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
  if (row < nrow) {
    const float* mr = M + row * ld + 16 * c;
    const float* vr = v + 16 * c;
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      if (16 * rd + 4 * c < nq) {
        f4 mq[4], vq[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { mq[t] = *reinterpret_cast<const f4*>(mr + 64 * rd + 4 * t); vq[t] = *reinterpret_cast<const f4*>(vr + 64 * rd + 4 * t); }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          a0 = pk_fma(f2v{mq[t][0], mq[t][1]}, f2v{vq[t][0], vq[t][1]}, a0);
          a1 = pk_fma(f2v{mq[t][2], mq[t][3]}, f2v{vq[t][2], vq[t][3]}, a1);
        }
      }
    }
  }
  return quad_sum((a0[0] + a0[1]) + (a1[0] + a1[1]));
}

// two products on one pass over M: (M v0, M v1)
__device__ __forceinline__ void mv4_dual(const float* M, int ld, const float* v0, const float* v1, int nq, int nrow, float& o0, float& o1) {
  const int row = threadIdx.x >> 2, c = threadIdx.x & 3;
  f2v a0 = {0.f, 0.f}, a1 = {0.f, 0.f}, b0 = {0.f, 0.f}, b1 = {0.f, 0.f};
  if (row < nrow) {
    const float* mr = M + row * ld + 16 * c;
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      if (16 * rd + 4 * c < nq) {
        f4 mq[4], xq[4], yq[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          mq[t] = *reinterpret_cast<const f4*>(mr + 64 * rd + 4 * t);
          xq[t] = *reinterpret_cast<const f4*>(v0 + 16 * c + 64 * rd + 4 * t);
          yq[t] = *reinterpret_cast<const f4*>(v1 + 16 * c + 64 * rd + 4 * t);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          a0 = pk_fma(f2v{mq[t][0], mq[t][1]}, f2v{xq[t][0], xq[t][1]}, a0);
          a1 = pk_fma(f2v{mq[t][2], mq[t][3]}, f2v{xq[t][2], xq[t][3]}, a1);
          b0 = pk_fma(f2v{mq[t][0], mq[t][1]}, f2v{yq[t][0], yq[t][1]}, b0);
          b1 = pk_fma(f2v{mq[t][2], mq[t][3]}, f2v{yq[t][2], yq[t][3]}, b1);
        }
      }
    }
  }
  o0 = quad_sum((a0[0] + a0[1]) + (a1[0] + a1[1]));
  o1 = quad_sum((b0[0] + b0[1]) + (b1[0] + b1[1]));
}

// (M^T v)_k, k = tid >> 2, tid < 512: lane c sums the rows 4u + c (4-byte reads along the row: the 32 lanes of a read group touch
// 4 banks twice); nrow = DP (a multiple of 16).  Every lane of the column's quad returns the sum.
__device__ __forceinline__ float mv4t(const float* M, int ld, const float* v, int nrow) {
  const int k = threadIdx.x >> 2, c = threadIdx.x & 3;
  float a0 = 0.f, a1 = 0.f;
  if (k < nrow) {
    const float* mc = M + c * ld + k;
    const float* vc = v + c;
    for (int u0 = 0; 4 * u0 < nrow; u0 += 8) {                     // rows 4 (u0 + t) + c, eight (or four: nrow / 4 is a multiple of 4) in flight
      float mv[8], vv[8];
      const bool two = 4 * (u0 + 4) < nrow;
#pragma unroll
      for (int t = 0; t < 4; ++t) { mv[t] = mc[4 * (u0 + t) * ld]; vv[t] = vc[4 * (u0 + t)]; }
      if (two) {
#pragma unroll
        for (int t = 4; t < 8; ++t) { mv[t] = mc[4 * (u0 + t) * ld]; vv[t] = vc[4 * (u0 + t)]; }
      } else {
#pragma unroll
        for (int t = 4; t < 8; ++t) { mv[t] = 0.f; vv[t] = 0.f; }
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) { if (t & 1) a1 = fmaf(mv[t], vv[t], a1); else a0 = fmaf(mv[t], vv[t], a0); }
    }
  }
  return quad_sum(a0 + a1);
}

// m' = V0^T m and d' = V0^T d in one pass over V0 (mv4t's layout, (m_i, d_i) interleaved at offMD), waves 0 .. 7
__device__ HTA_PH_ATTR void ph_fast_vt(int offV, int offMD, int offM, int offD, int DP, int LD) {
  HTA_LDS_BASE();
  DP = HTA_U(DP); LD = HTA_U(LD);
  if (threadIdx.x < 512) {
    const float* V = lds + HTA_U(offV);
    const f2v* md = reinterpret_cast<const f2v*>(lds + HTA_U(offMD));
    const int k = threadIdx.x >> 2, c = threadIdx.x & 3;
    f2v a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
    if (k < DP) {
      const float* mc = V + c * LD + k;
      for (int u0 = 0; 4 * u0 < DP; u0 += 8) {
        float mv[8]; f2v w[8];
        const bool two = 4 * (u0 + 4) < DP;
#pragma unroll
        for (int t = 0; t < 4; ++t) { mv[t] = mc[4 * (u0 + t) * LD]; w[t] = md[4 * (u0 + t) + c]; }
        if (two) {
#pragma unroll
          for (int t = 4; t < 8; ++t) { mv[t] = mc[4 * (u0 + t) * LD]; w[t] = md[4 * (u0 + t) + c]; }
        } else {
#pragma unroll
          for (int t = 4; t < 8; ++t) { mv[t] = 0.f; w[t] = f2v{0.f, 0.f}; }
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) { if (t & 1) a1 = pk_fma(f2v{mv[t], mv[t]}, w[t], a1); else a0 = pk_fma(f2v{mv[t], mv[t]}, w[t], a0); }
      }
    }
    const float sm = quad_sum(a0[0] + a1[0]), sd = quad_sum(a0[1] + a1[1]);
    if (c == 0 && k < DP) { (lds + HTA_U(offM))[k] = sm; (lds + HTA_U(offD))[k] = sd; }
  }
  __syncthreads();
}

// The work items of the two products, once per launch (the scalar loops that deal them out cost every wave ~100 instructions per call
// on the CU's one scalar unit: the waves entered their k loops 1 .. 3.4 k cycles apart).  Bits 0 .. 9: the wave's item of the upper block
// triangle as lds_gemm_ld's SYM branch deals them (I0 | J0 << 4 | second tile << 8 | active << 9); bits 16 .. 26: its 2 x 2 macro tile
// of a full product as the other branch does (I0 | J0 << 4 | second row << 8 | second column << 9 | active << 10).
__device__ __forceinline__ int fast_tiles(int nt) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int lo, hi;
  {
    int npairs = 0;
    for (int I = 0; I < nt; ++I) npairs += (nt - I) >> 1;
    int w = wave, I = 0, I0, J0, c2 = 0, active = 1;
    if (w < npairs) {
      for (;; ++I) { const int pr = (nt - I) >> 1; if (w < pr) break; w -= pr; }
      I0 = I; J0 = I + 2 * w; c2 = 1;
    } else {
      w -= npairs;
      for (; I < nt; ++I) if ((nt - I) & 1) { if (w == 0) break; --w; }
      if (I >= nt) { active = 0; I = 0; }
      I0 = I; J0 = nt - 1;
    }
    lo = I0 | (J0 << 4) | (c2 << 8) | (active << 9);
  }
  {
    const int nf = nt >> 1, odd = nt & 1;
    const int s = wave & 3, q = wave >> 2;
    int k = (q & 1) ? 4 * q + 3 - s : 4 * q + s;
    int r = 0, c = 0, active = 1;
    const int nfull = nf * nf;
    if (k < nfull) { r = k / nf; c = k - r * nf; }
    else {
      k -= nfull;
      if (!odd || k > 2 * nf) active = 0;
      else if (k < nf) { r = k; c = nf; } else if (k < 2 * nf) { r = nf; c = k - nf; } else { r = nf; c = nf; }
    }
    const int I0 = 2 * r, J0 = 2 * c;
    hi = I0 | (J0 << 4) | ((I0 + 1 < nt ? 1 : 0) << 8) | ((J0 + 1 < nt ? 1 : 0) << 9) | (active << 10);
  }
  // bits 27 .. 30 / 31: the wave's STRIP of the second product's bfloat16 form (ph_fast_second_strip) and whether it has one.  Strip s < nt:
  // column tile J = s against row tiles 0 .. 3; strip nt + J: against row tiles 4 .. nt - 1.  nt = 7 (BASELINE config 3): 7 strips of four
  // tiles and 7 of three dealt so that the four SIMDs (wave & 3) carry 13 / 12 / 12 / 12 tiles.
  int sid = wave, sact = wave < nt * ((nt + 3) >> 2) ? 1 : 0;
  if (nt == 7) {
    const int big[7] = {0, 1, 5, 9, 2, 6, 10}, small[7] = {4, 8, 12, 3, 7, 11, 15};
    sact = 0;
    for (int k = 0; k < 7; ++k) { if (wave == big[k]) { sid = k; sact = 1; } if (wave == small[k]) { sid = 7 + k; sact = 1; } }
  }
  return (int)((unsigned)(lo | (hi << 16)) | ((unsigned)(sid & 15) << 27) | ((unsigned)sact << 31));
}

// F as two bfloat16 planes (PL: what the bfloat16 form of the second product reads): hi = the upper half of the fp32 word, lo = the upper
// half of (x - hi); a lane's four values are four ROWS of the tile (2-byte stores) and four consecutive columns of its mirror image (8 bytes)
__device__ __forceinline__ void plane_store_col(unsigned short* ph, unsigned short* pl, int stride, const f4& a) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const unsigned b = __float_as_uint(a[t]);
    const float r = a[t] - __uint_as_float(b & 0xffff0000u);
    ph[t * stride] = (unsigned short)(b >> 16);
    pl[t * stride] = (unsigned short)(__float_as_uint(r) >> 16);
  }
}
__device__ __forceinline__ void plane_store_row(unsigned short* ph, unsigned short* pl, const f4& a) {
  int h0, l0, h1, l1;
  split_pair(a[0], a[1], h0, l0); split_pair(a[2], a[3], h1, l1);
  *reinterpret_cast<i2v*>(ph) = i2v{h0, h1};
  *reinterpret_cast<i2v*>(pl) = i2v{l0, l1};
}

// F = V0^T diag(e) V0 (upper tiles, as lds_gemm_ld<true, false, true, true>) with the first pass in the epilogue: the diagonal tiles
// publish lam_i = lam0_i + F_ii, then every tile turns its accumulators into E1_ij = F_ij / (lam_j - lam_i) (ph_refine_E's
// arithmetic with X = I: the same quotients, the same threshold) and stores F (zero diagonal) and E1 (antisymmetric, zero
// diagonal) with their mirror images as 16-byte rows.  Returns max |E1_ij| (1 for NaN / inf / > kFallbackE).
// WB (round 6, "metric_bx3" = 2): the formation itself on bfloat16.  W^T = (diag(sqrt e) V0)^T is written once per evaluation as two bfloat16
// planes (hi, lo) into E1's buffer - idle until the epilogue - by a pass that reads V0 column-wise (4-byte reads, consecutive lanes on
// consecutive columns) and stores 16-byte groups of eight contraction indices; F = W^T W is then three products per tile and 32 indices
// with NO vector instruction in the loop (gemm_macro_pp): 1.2 k cycles of the matrix pipe per SIMD instead of 5.6 k.  Each term of F carries a
// relative error 2^-16; emulated in float64 around it (tools/scratch/bf16_formation_err.py), max |x - x64| / max |x64| of a solve moves from
// 7.4e-8 to 7.8e-8 at BASELINE config 3's jitter (1.5e-7 to 1.9e-7 at three times the jitter) - the closed-form second pass's own truncation
// is that size, the kernel's fp32 vector arithmetic (3.6e-6) is fifty times it.
template <int LDC, bool PL, bool WB = false>
__device__ __forceinline__ float fast_form_body(int offV, int offF, int offE, int offJit, int offLam0, int offLam, int offRed, int tile, int k4, int D, int LDr, int nt) {
  HTA_LDS_BASE();
  k4 = HTA_U(k4); D = HTA_U(D); nt = LDC == kLdCfg3 ? kNtCfg3 : HTA_U(nt);
  const int LD = LDC ? LDC : HTA_U(LDr);
  const float* V = lds + HTA_U(offV);
  float* F = lds + HTA_U(offF); float* E = lds + HTA_U(offE);
  const float* kscale = lds + HTA_U(offJit);
  const float* vlam0 = lds + HTA_U(offLam0);
  float* vlam = lds + HTA_U(offLam); float* red = lds + HTA_U(offRed);
  const int lane = threadIdx.x & 63;
  const int li = lane & 15, lk = lane >> 4;
  tile = HTA_U(tile);                                              // fast_tiles(): this wave's item of the upper block triangle
  const int I0 = tile & 15, J0 = (tile >> 4) & 15;
  const bool c2 = (tile >> 8) & 1, active = (tile >> 9) & 1;
  HTA_WVSTAMP(4);
  f4 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) acc[x][y] = f4{0.f, 0.f, 0.f, 0.f};
  if (WB) {
    // W^T planes: thread (column i = tid & 127 of V0, octet o = tid >> 7 [+ 8]) scales V0[8 o .. 8 o + 7][i] by sqrt(e) and splits
    unsigned short* Wh = reinterpret_cast<unsigned short*>(E);
    const int DPh = 16 * nt, plane = DPh * DPh;
    const int i = threadIdx.x & 127, o0 = HTA_U(threadIdx.x >> 7);
    if (i < DPh) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int o = o0 + 8 * u;
        if (8 * o < DPh) {
          float w[8];
          const f4 s0 = *reinterpret_cast<const f4*>(kscale + 8 * o), s1 = *reinterpret_cast<const f4*>(kscale + 8 * o + 4);
#pragma unroll
          for (int r = 0; r < 8; ++r) w[r] = V[(8 * o + r) * LD + i] * (r < 4 ? s0[r] : s1[r - 4]);      // (the scale vector holds sqrt(e) here: metric_fast_solve)
          int h[4], l[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) split_pair(w[2 * r], w[2 * r + 1], h[r], l[r]);
          *reinterpret_cast<i4v*>(Wh + i * DPh + 8 * o) = i4v{h[0], h[1], h[2], h[3]};
          *reinterpret_cast<i4v*>(Wh + plane + i * DPh + 8 * o) = i4v{l[0], l[1], l[2], l[3]};
        }
      }
    }
    __syncthreads();
  }
  if (active) {
    if (WB) {
      const unsigned short* Wh = reinterpret_cast<const unsigned short*>(E);
      const unsigned short* pa0 = Wh + (16 * I0 + li) * (16 * nt);
      const unsigned short* pb0 = Wh + (16 * J0 + li) * (16 * nt);
      if (c2) gemm_macro_pp<true, kNtCfg3>(pa0, pb0, acc); else gemm_macro_pp<false, kNtCfg3>(pa0, pb0, acc);
    } else {
      const float* pa0 = V + lk * LD + 16 * I0 + li;
      const float* pb0 = V + lk * LD + 16 * J0 + li;
      if (c2) gemm_macro<true, false, true, false, true>(pa0, pb0, kscale, k4, LD, lk, acc);
      else gemm_macro<true, false, true, false, false>(pa0, pb0, kscale, k4, LD, lk, acc);
    }
    HTA_WVSTAMP(5);
    if (I0 == J0) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (4 * lk + t == li) { const int i = 16 * I0 + li; if (i < D) vlam[i] = acc[0][0][t] + vlam0[i]; }
    }
  }
  HTA_WVSTAMP(6);
  __syncthreads();
  HTA_WVSTAMP(7);
  const float sc = wave_max_abs(vlam, 16 * nt);                     // (entries D .. DP - 1 of lam are zero)
  const float tiny = 8.f * Eps<float>::v * sc;
  unsigned ebits = 0u;                                             // max |E1_ij| as a bit pattern: inf and NaN order above every finite value
  if (active) {
    const f4 lamI = *reinterpret_cast<const f4*>(vlam + 16 * I0 + 4 * lk);
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      if (y == 1 && !c2) continue;
      const int J = J0 + y, j = 16 * J + li;
      const float lamJ = vlam[j];
      if (I0 != J) {
        const f4 a = acc[0][y];
        const f2v lamJ2 = {lamJ, lamJ};
        const f2v d01 = lamJ2 - f2v{lamI[0], lamI[1]}, d23 = lamJ2 - f2v{lamI[2], lamI[3]};
        const f2v r01 = {__builtin_amdgcn_rcpf(d01[0]), __builtin_amdgcn_rcpf(d01[1])}, r23 = {__builtin_amdgcn_rcpf(d23[0]), __builtin_amdgcn_rcpf(d23[1])};
        const f2v q01 = f2v{a[0], a[1]} * r01, q23 = f2v{a[2], a[3]} * r23;
        f4 e;
        e[0] = (fabsf(a[0]) <= tiny) ? 0.f : q01[0];
        e[1] = (fabsf(a[1]) <= tiny) ? 0.f : q01[1];
        e[2] = (fabsf(a[2]) <= tiny) ? 0.f : q23[0];
        e[3] = (fabsf(a[3]) <= tiny) ? 0.f : q23[1];
        float* fd = F + (16 * I0 + 4 * lk) * LD + j;
        float* ed = E + (16 * I0 + 4 * lk) * LD + j;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          ebits = max(ebits, __float_as_uint(e[t]) & 0x7fffffffu);
          if (!PL) fd[t * LD] = a[t];
          ed[t * LD] = e[t];
        }
        const f2v n01 = f2v{e[0], e[1]} * f2v{-1.f, -1.f}, n23 = f2v{e[2], e[3]} * f2v{-1.f, -1.f};
        if (PL) {
          unsigned short* Fh = reinterpret_cast<unsigned short*>(F);
          const int DPh = 16 * nt, plane = DPh * DPh;
          plane_store_col(Fh + (16 * I0 + 4 * lk) * DPh + j, Fh + plane + (16 * I0 + 4 * lk) * DPh + j, DPh, a);
          plane_store_row(Fh + j * DPh + 16 * I0 + 4 * lk, Fh + plane + j * DPh + 16 * I0 + 4 * lk, a);
        } else {
          *reinterpret_cast<f4*>(F + j * LD + 16 * I0 + 4 * lk) = a;
        }
        *reinterpret_cast<f4*>(E + j * LD + 16 * I0 + 4 * lk) = f4{n01[0], n01[1], n23[0], n23[1]};
      } else {
        // a diagonal tile holds both triangles of its block: every lane stores its own four elements, no mirror.  (F_ij and F_ji differ
        // in the last bit there - the scale factor enters through the B operand only - so E1 + E1^T is zero to rounding of a 5e-3
        // quantity, not exactly: 1e-10, far below anything the solve resolves.)
        const f4 a = acc[0][y];
        const f2v lamJ2 = {lamJ, lamJ};
        const f2v d01 = lamJ2 - f2v{lamI[0], lamI[1]}, d23 = lamJ2 - f2v{lamI[2], lamI[3]};
        const f2v r01 = {__builtin_amdgcn_rcpf(d01[0]), __builtin_amdgcn_rcpf(d01[1])}, r23 = {__builtin_amdgcn_rcpf(d23[0]), __builtin_amdgcn_rcpf(d23[1])};
        const f2v q01 = f2v{a[0], a[1]} * r01, q23 = f2v{a[2], a[3]} * r23;
        f4 e;
        e[0] = (fabsf(a[0]) <= tiny) ? 0.f : q01[0];
        e[1] = (fabsf(a[1]) <= tiny) ? 0.f : q01[1];
        e[2] = (fabsf(a[2]) <= tiny) ? 0.f : q23[0];
        e[3] = (fabsf(a[3]) <= tiny) ? 0.f : q23[1];
        float* fd = F + (16 * I0 + 4 * lk) * LD + j;
        float* ed = E + (16 * I0 + 4 * lk) * LD + j;
        f4 f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const bool dg = 4 * lk + t == li;
          e[t] = dg ? 0.f : e[t]; f[t] = dg ? 0.f : a[t];
          ebits = max(ebits, __float_as_uint(e[t]) & 0x7fffffffu);
          if (!PL) fd[t * LD] = f[t];
          ed[t * LD] = e[t];
        }
        if (PL) {
          unsigned short* Fh = reinterpret_cast<unsigned short*>(F);
          const int DPh = 16 * nt, plane = DPh * DPh;
          plane_store_col(Fh + (16 * I0 + 4 * lk) * DPh + j, Fh + plane + (16 * I0 + 4 * lk) * DPh + j, DPh, f);
        }
      }
    }
  }
  float emax = (ebits > __float_as_uint(kFallbackE)) ? 1.f : __uint_as_float(ebits);      // NaN / inf / too large
  HTA_WVSTAMP(8);
  emax = block_max1(emax, red);         // (`red` was last read before this function's barrier above)
  HTA_WVSTAMP(9);
  return emax;
}
template <int LDC, bool PL, bool WB = false>
__device__ HTA_PH_ATTR float ph_fast_form(int offV, int offF, int offE, int offJit, int offLam0, int offLam, int offRed, int tile, int k4, int D, int LDr, int nt) {
  return fast_form_body<LDC, PL, WB>(offV, offF, offE, offJit, offLam0, offLam, offRed, tile, k4, D, LDr, nt);
}
// The RESIDENT evaluations of the trajectory kernel at kLdCfg3 (ph_fast_form_r / ph_fast_second_strip_r / ph_fast_chain_r): every LDS offset
// is a compile-time function of the buffer V0 sits in, so a call passes ONE packed word (bits 0 .. 1: V0's buffer, 2 .. 9: D, 10 ..: the
// chain's flags) instead of 12 ... 21 registers - an argument costs the caller a move (often a read of a parked scalar as well) and the
// callee a readfirstlane, per wave, and with 16 waves on the CU every one of those is ~16 cycles: ~1 k cycles per call.
struct ResLayout {
  int bx, by, bz, oJit, oLam, oLt, oRed, D;
  __device__ __forceinline__ explicit ResLayout(int code) {
    constexpr int DP = 16 * kNtCfg3, BS = DP * kLdCfg3;
    code = HTA_U(code);
    const int vb = code & 3;
    bx = vb * BS; by = vb == 0 ? BS : 0; bz = vb == 2 ? BS : 2 * BS;
    oJit = 3 * BS; oLam = oJit + DP; oLt = oLam + DP; oRed = oJit + 8 * DP;
    D = (code >> 2) & 255;
  }
};
__device__ HTA_PH_ATTR float ph_fast_form_r(int code, int tile) {
  const ResLayout r(code);
  return fast_form_body<kLdCfg3, true, true>(r.bx, r.by, r.bz, r.oJit, r.oLt, r.oLam, r.oRed, tile, (r.D + 3) >> 2, r.D, kLdCfg3, kNtCfg3);
}

// M = F E1 (as lds_gemm_ld<false, false, false, false>) with the closed-form second pass in the epilogue (ph_refine_E2's arithmetic
// on the accumulators): lam_i' = lam_i + M_ii from the diagonal tiles, E2_ij = M_ij / (lam_j' - lam_i'), E2_ii = -1/2 sum_k E1_ik^2
// (row sums taken before the product), E2 stored over F once every wave has left its k loop.  Returns max |E2_ij|.
template <int LDC>
// Round 6, later: the vectors' first two products ride along.  The pass over E1 that takes the row sums also takes E1 m' - y0 = m' - E1 m'
// is published before the product - and the epilogue, which holds E2 in registers, accumulates y0^T E2 per column (a DPP row sum over the 16
// rows of a tile, one 16-byte store of four columns per tile and lane group) into one partial vector per macro-tile row (offP0 .. offP3:
// idle vectors); the chain starts from y = y0 + the sum of those partials instead of two matrix-vector products and a barrier.
__device__ HTA_PH_ATTR float ph_fast_second(int offF, int offE, int oVec, int offRed, int nt, int tile, int k4, int D, int LDr) {
  HTA_LDS_BASE();
  nt = HTA_U(nt); k4 = HTA_U(k4); D = HTA_U(D); oVec = HTA_U(oVec);
  const int LD = LDC ? LDC : HTA_U(LDr);
  const int DPv = 16 * nt;
  float* F = lds + HTA_U(offF); const float* E = lds + HTA_U(offE);
  float* vlam = lds + oVec + DPv; const float* vm = lds + oVec + 3 * DPv; float* vy = lds + oVec + 4 * DPv; float* vcs = lds + oVec + 7 * DPv;
  float* red = lds + HTA_U(offRed);
  const int lane = threadIdx.x & 63;
  const int li = lane & 15, lk = lane >> 4;
  {
    const int row = threadIdx.x >> 3, seg = threadIdx.x & 7;
    float c0 = 0.f, c1 = 0.f, u0 = 0.f, u1 = 0.f;
    if (row < 16 * nt) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = seg + 8 * u;
        if (q < 4 * nt) {
          const f4 v = *reinterpret_cast<const f4*>(E + row * LD + 4 * q);
          const f4 mq = *reinterpret_cast<const f4*>(vm + 4 * q);
          // (packed: the same sums in the same order, half the instructions)
          const f2v v01 = {v[0], v[1]}, v23 = {v[2], v[3]};
          f2v cc = pk_fma(v01, v01, f2v{c0, c1}); cc = pk_fma(v23, v23, cc); c0 = cc[0]; c1 = cc[1];
          f2v uu = pk_fma(v01, f2v{mq[0], mq[1]}, f2v{u0, u1}); uu = pk_fma(v23, f2v{mq[2], mq[3]}, uu); u0 = uu[0]; u1 = uu[1];
        }
      }
    }
    const float cs = sum8_dpp(c0 + c1), e1m = sum8_dpp(u0 + u1);
    if (seg == 0 && row < 16 * nt) { vcs[row] = cs; vy[row] = (row < D) ? vm[row] - e1m : 0.f; }
  }
  HTA_WVSTAMP(10);
  tile = HTA_U(tile) >> 16;                                        // fast_tiles(): this wave's 2 x 2 macro tile of the full product
  const int I0 = tile & 15, J0 = (tile >> 4) & 15;
  const bool r2 = (tile >> 8) & 1, c2 = (tile >> 9) & 1, active = (tile >> 10) & 1;
  // The instruction computes the TRANSPOSED tile: its rows run over j (operand A = rows of E1), its columns over i (operand B = rows of
  // the symmetric F), acc[x][y][t] = sum_k E1[j][k] F[i][k] = -M[i][j] at i = 16 (I0 + y) + li, j = 16 (J0 + x) + 4 lk + t - a lane
  // holds four consecutive columns of ONE row of M, so E2 leaves as one 16-byte store per tile and lane and the element-wise step runs
  // on register pairs (v_pk_add / v_pk_mul).  The sign goes into the reciprocal: E2_ij = M_ij / (lam_j - lam_i) = acc / (lam_i - lam_j).
  f4 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) acc[x][y] = f4{0.f, 0.f, 0.f, 0.f};
  if (active) {
    // (exact fp32 products; the bfloat16 form runs on strips: ph_fast_second_strip)
    {
      const float* pa0 = E + (16 * J0 + li) * LD + lk;
      const float* pb0 = F + lk * LD + 16 * I0 + li;
      if (r2 && c2) gemm_macro<false, false, false, true, true>(pa0, pb0, nullptr, k4, LD, lk, acc);
      else if (c2) gemm_macro<false, false, false, true, false>(pa0, pb0, nullptr, k4, LD, lk, acc);
      else if (r2) gemm_macro<false, false, false, false, true>(pa0, pb0, nullptr, k4, LD, lk, acc);
      else gemm_macro<false, false, false, false, false>(pa0, pb0, nullptr, k4, LD, lk, acc);
    }
    HTA_WVSTAMP(11);
    if (I0 == J0) {                                                // lam_i' = lam_i + M_ii
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        if (x == 1 && !r2) continue;
#pragma unroll
        for (int t = 0; t < 4; ++t)
          if (4 * lk + t == li) { const int i = 16 * (I0 + x) + li; if (i < D) vlam[i] = vlam[i] - acc[x][x][t]; }
      }
    }
  }
  HTA_WVSTAMP(12);
  __syncthreads();
  HTA_WVSTAMP(13);
  const float sc = wave_max_abs(vlam, 16 * nt);                     // (entries D .. DP - 1 of lam are zero)
  const float tiny = 8.f * Eps<float>::v * sc * kSecondE;
  unsigned ebits = 0u;                                             // max |E2_ij| as a bit pattern: inf and NaN order above every finite value
  f4 pacc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};   // y0^T E2 over this wave's rows, per column tile
  if (active) {
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      if (y == 1 && !r2) continue;
      const int I = I0 + y, i = 16 * I + li;
      const float lamI = vlam[i], y0i = vy[i];
      const f2v lamI2 = {lamI, lamI};
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        if (x == 1 && !c2) continue;
        const int J = J0 + x;
        const f4 lamJ = *reinterpret_cast<const f4*>(vlam + 16 * J + 4 * lk);
        const f4 a = acc[x][y];
        const f2v d01 = lamI2 - f2v{lamJ[0], lamJ[1]}, d23 = lamI2 - f2v{lamJ[2], lamJ[3]};
        const f2v r01 = {__builtin_amdgcn_rcpf(d01[0]), __builtin_amdgcn_rcpf(d01[1])}, r23 = {__builtin_amdgcn_rcpf(d23[0]), __builtin_amdgcn_rcpf(d23[1])};
        const f2v q01 = f2v{a[0], a[1]} * r01, q23 = f2v{a[2], a[3]} * r23;
        f4 e;
        e[0] = (fabsf(a[0]) <= tiny) ? 0.f : q01[0];
        e[1] = (fabsf(a[1]) <= tiny) ? 0.f : q01[1];
        e[2] = (fabsf(a[2]) <= tiny) ? 0.f : q23[0];
        e[3] = (fabsf(a[3]) <= tiny) ? 0.f : q23[1];
        if (I == J) {                                              // (wave-uniform) the diagonal element: E2_ii = -1/2 sum_k E1_ik^2, not an update
          const float dg = -0.5f * vcs[i];
#pragma unroll
          for (int t = 0; t < 4; ++t) if (4 * lk + t == li) e[t] = 0.f;
#pragma unroll
          for (int t = 0; t < 4; ++t) ebits = max(ebits, __float_as_uint(e[t]) & 0x7fffffffu);
#pragma unroll
          for (int t = 0; t < 4; ++t) if (4 * lk + t == li) e[t] = dg;
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) ebits = max(ebits, __float_as_uint(e[t]) & 0x7fffffffu);
        }
        *reinterpret_cast<f4*>(F + i * LD + 16 * J + 4 * lk) = e;
#pragma unroll
        for (int t = 0; t < 4; ++t) pacc[x][t] = fmaf(e[t], y0i, pacc[x][t]);
      }
    }
    // one partial vector per macro-tile row: vectors that are idle here (e | m' | x | the Cholesky's panel scratch)
    float* part = lds + (I0 == 0 ? oVec : I0 == 2 ? oVec + 3 * DPv : I0 == 4 ? oVec + 5 * DPv : oVec + 8 * DPv + MT / 64 + 64);
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      if (x == 1 && !c2) continue;
      f4 sum;
#pragma unroll
      for (int t = 0; t < 4; ++t) sum[t] = sum16_dpp(pacc[x][t]);
      if (li == 0) *reinterpret_cast<f4*>(part + 16 * (J0 + x) + 4 * lk) = sum;
    }
  }
  float emax = (ebits > __float_as_uint(kFallbackE)) ? 1.f : __uint_as_float(ebits);      // NaN / inf / too large
  HTA_WVSTAMP(14);
  emax = block_max1(emax, red);
  HTA_WVSTAMP(15);
  return emax;
}

// The soft-abs map of rows i (one wave: 64 consecutive rows) - lam~ over lam0 at vlt, lam0 d' over d' at vd, the wave's partials of the
// log-determinant and of sum lam0 d'^2 into slot `wslot` of the block-sum table.  Runs on waves 8 .. 9 of ph_fast_chain, or (round 6, later)
// on the two waves ph_fast_second_strip leaves without a strip, under that phase's element-wise step.
__device__ __forceinline__ void fast_softabs_rows(int i, int wslot, int D, int DP, float alpha, const float* vlam, float* vlt, float* vd, float* red3,
                                                  float* lam_out, float* lamraw_out, bool has_x, bool sync_after_map) {
  typedef __attribute__((address_space(1))) float* gf;
  float ld = 0.f, lq = 0.f, lt = 1.f, l0 = 0.f;
  if (i < D) {
    const float lam = vlam[i], x = alpha * lam;
    // |alpha lam| >= 10: tanh is 1 - 4e-9, i.e. tanhf returns +-1 exactly, lam / tanh = |lam| bit for bit - the whole wave skips the function's body when
    // every eigenvalue is there (the identity soft-abs map of BASELINE config 3: alpha = 1e6)
    const bool sat = fabsf(x) >= 10.f;
    if (__builtin_amdgcn_ballot_w64(!sat) == 0) lt = fabsf(lam);
    else lt = (1.f / tanhf(x)) * lam;
    l0 = vlt[i];                                               // (lam0 is parked where lam~ goes)
    if (lamraw_out) ((gf)lamraw_out)[i] = lam;
  }
  if (i < DP) vlt[i] = lt;
  if (sync_after_map) __syncthreads();                         // (ph_fast_chain without a second pass: the first of its four barriers - y0)
  if (i < D) {
    ld = logf(lt);
    if (lam_out) ((gf)lam_out)[i] = lt;
    if (has_x) { const float dp = vd[i]; lq = l0 * dp * dp; vd[i] = l0 * dp; }
  }
  const float s0 = wave_sum_dpp(ld), s2 = wave_sum_dpp(lq);
  if ((threadIdx.x & 63) == 0) { float* r = red3 + 4 * wslot; r[0] = s0; r[1] = 0.f; r[2] = s2; }
}

// The bfloat16 form of the second product on 1 x 4 / 1 x 3 STRIPS (fast_tiles bits 27 .. 31): the same pre-pass, barrier and element-wise step
// as ph_fast_second<.., true>, one E1 tile per wave split once per 32 indices for up to four tiles of F's planes; the partial vectors of
// E2^T y0 are two (row tiles 0 .. 3 -> the slot of I0 = 0, row tiles 4 .. -> the slot of I0 = 4: ph_fast_chain flag 32).
template <int LDC>
__device__ __forceinline__ float fast_second_strip_body(int offF, int offE, int oVec, int offRed, int nt, int tile, int D, float alpha, int has_x,
                                                        float* lam_out, float* lamraw_out) {
  HTA_LDS_BASE();
  static_assert(LDC == kLdCfg3, "the planes exist at this leading dimension only");
  nt = kNtCfg3; D = HTA_U(D); oVec = HTA_U(oVec);
  constexpr int LD = LDC;
  const int DPv = 16 * nt;
  float* F = lds + HTA_U(offF); const float* E = lds + HTA_U(offE);
  float* vlam = lds + oVec + DPv; const float* vm = lds + oVec + 3 * DPv; float* vy = lds + oVec + 4 * DPv; float* vcs = lds + oVec + 7 * DPv;
  float* red = lds + HTA_U(offRed);
  const int lane = threadIdx.x & 63;
  const int li = lane & 15, lk = lane >> 4;
  {
    const int row = threadIdx.x >> 3, seg = threadIdx.x & 7;
    float c0 = 0.f, c1 = 0.f, u0 = 0.f, u1 = 0.f;
    if (row < 16 * nt) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = seg + 8 * u;
        if (q < 4 * nt) {
          const f4 v = *reinterpret_cast<const f4*>(E + row * LD + 4 * q);
          const f4 mq = *reinterpret_cast<const f4*>(vm + 4 * q);
          // (packed: the same sums in the same order, half the instructions)
          const f2v v01 = {v[0], v[1]}, v23 = {v[2], v[3]};
          f2v cc = pk_fma(v01, v01, f2v{c0, c1}); cc = pk_fma(v23, v23, cc); c0 = cc[0]; c1 = cc[1];
          f2v uu = pk_fma(v01, f2v{mq[0], mq[1]}, f2v{u0, u1}); uu = pk_fma(v23, f2v{mq[2], mq[3]}, uu); u0 = uu[0]; u1 = uu[1];
        }
      }
    }
    const float cs = sum8_dpp(c0 + c1), e1m = sum8_dpp(u0 + u1);
    if (seg == 0 && row < 16 * nt) { vcs[row] = cs; vy[row] = (row < D) ? vm[row] - e1m : 0.f; }
  }
  HTA_WVSTAMP(10);
  const unsigned tl = (unsigned)HTA_U(tile);
  const int sid = (tl >> 27) & 15;
  const bool active = (tl >> 31) != 0;
  const int part = sid >= nt ? 1 : 0, J = sid - part * nt, I0 = 4 * part;
  const int cnt = part ? nt - 4 : (nt < 4 ? nt : 4);
  f4 acc[4];
#pragma unroll
  for (int y = 0; y < 4; ++y) acc[y] = f4{0.f, 0.f, 0.f, 0.f};
  if (active) {
    const float* pa0 = E + (16 * J + li) * LD;
    const unsigned short* pb0 = reinterpret_cast<const unsigned short*>(F) + (16 * I0 + li) * (16 * nt);
    if (cnt == 4) gemm_strip_bx3<4, kNtCfg3>(pa0, pb0, acc);      // (7 tiles: strips of 4 and 3)
    else gemm_strip_bx3<3, kNtCfg3>(pa0, pb0, acc);
    HTA_WVSTAMP(11);
#pragma unroll
    for (int y = 0; y < 4; ++y)
      if (y < cnt && I0 + y == J) {                                // lam_i' = lam_i + M_ii
#pragma unroll
        for (int t = 0; t < 4; ++t)
          if (4 * lk + t == li) { const int i = 16 * J + li; if (i < D) vlam[i] = vlam[i] - acc[y][t]; }
      }
  }
  HTA_WVSTAMP(12);
  __syncthreads();
  HTA_WVSTAMP(13);
  const float sc = wave_max_abs(vlam, 16 * nt);                     // (entries D .. DP - 1 of lam are zero)
  const float tiny = 8.f * Eps<float>::v * sc * kSecondE;
  unsigned ebits = 0u;
  {
    // the two waves without a strip (fast_tiles: 13 and 14 at 7 tiles) take the soft-abs map of the corrected eigenvalues, which ph_fast_chain
    // would otherwise wait for behind a barrier of its own (flag 64 there)
    const int wv = HTA_U(threadIdx.x >> 6);
    if (wv == 13 || wv == 14)
      fast_softabs_rows((int)threadIdx.x - 13 * 64, 8 + wv - 13, D, DPv, alpha, vlam, lds + oVec + 2 * DPv, lds + oVec + 6 * DPv, lds + oVec + 8 * DPv + MT / 64,
                        lam_out, lamraw_out, HTA_U(has_x) != 0, false);
  }
  if (active) {
    const f4 lamJ = *reinterpret_cast<const f4*>(vlam + 16 * J + 4 * lk);
    f4 pacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      if (y >= cnt) continue;
      const int I = I0 + y, i = 16 * I + li;
      const float lamI = vlam[i], y0i = vy[i];
      const f2v lamI2 = {lamI, lamI};
      const f4 a = acc[y];
      const f2v d01 = lamI2 - f2v{lamJ[0], lamJ[1]}, d23 = lamI2 - f2v{lamJ[2], lamJ[3]};
      const f2v r01 = {__builtin_amdgcn_rcpf(d01[0]), __builtin_amdgcn_rcpf(d01[1])}, r23 = {__builtin_amdgcn_rcpf(d23[0]), __builtin_amdgcn_rcpf(d23[1])};
      const f2v q01 = f2v{a[0], a[1]} * r01, q23 = f2v{a[2], a[3]} * r23;
      f4 e;
      e[0] = (fabsf(a[0]) <= tiny) ? 0.f : q01[0];
      e[1] = (fabsf(a[1]) <= tiny) ? 0.f : q01[1];
      e[2] = (fabsf(a[2]) <= tiny) ? 0.f : q23[0];
      e[3] = (fabsf(a[3]) <= tiny) ? 0.f : q23[1];
      if (I == J) {
        const float dg = -0.5f * vcs[i];
#pragma unroll
        for (int t = 0; t < 4; ++t) if (4 * lk + t == li) e[t] = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) ebits = max(ebits, __float_as_uint(e[t]) & 0x7fffffffu);
#pragma unroll
        for (int t = 0; t < 4; ++t) if (4 * lk + t == li) e[t] = dg;
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) ebits = max(ebits, __float_as_uint(e[t]) & 0x7fffffffu);
      }
      *reinterpret_cast<f4*>(F + i * LD + 16 * J + 4 * lk) = e;
#pragma unroll
      for (int t = 0; t < 4; ++t) pacc[t] = fmaf(e[t], y0i, pacc[t]);
    }
    float* pv = lds + (part ? oVec + 5 * DPv : oVec);
    f4 sum;
#pragma unroll
    for (int t = 0; t < 4; ++t) sum[t] = sum16_dpp(pacc[t]);
    if (li == 0) *reinterpret_cast<f4*>(pv + 16 * J + 4 * lk) = sum;
  }
  float emax = (ebits > __float_as_uint(kFallbackE)) ? 1.f : __uint_as_float(ebits);
  HTA_WVSTAMP(14);
  emax = block_max1(emax, red);
  HTA_WVSTAMP(15);
  return emax;
}
template <int LDC>
__device__ HTA_PH_ATTR float ph_fast_second_strip(int offF, int offE, int oVec, int offRed, int nt, int tile, int D, float alpha, int has_x,
                                                  float* lam_out, float* lamraw_out) {
  return fast_second_strip_body<LDC>(offF, offE, oVec, offRed, nt, tile, D, alpha, has_x, lam_out, lamraw_out);
}
__device__ HTA_PH_ATTR float ph_fast_second_strip_r(int code, int tile, float alpha) {
  const ResLayout r(code);
  return fast_second_strip_body<kLdCfg3>(r.by, r.bz, r.oJit, r.oRed, kNtCfg3, tile, r.D, alpha, 1, nullptr, nullptr);
}

// The vectors of the solve: soft-abs map, y = (I + E2^T)(I - E1) m', w = y / lam~, x' = (I + E1)(I + E2) w, x = V0 x', P d = V0 (lam0 d'),
// the two row updates; the block sums (log-det, y^T w, sum lam0 d'^2) as wave partials at oS (waves 0 .. 9, stride 4: the caller adds them).
// Waves 0 .. 7 carry the products (mv4: a row's value stays in its four lanes from one stage to the next), waves 8 .. 9 the soft-abs
// map under the second product; five barriers.  Vector block at oVec: e | lam | lam0 -> lam~ | m' | y | x | d' -> lam0 d' | cs, then 16 floats, then oS.
// flags: 8 = the DRAW p = G^(1/2) z: w = y sqrt(lam~) instead of y / lam~; 16 (with 4) = x' REPLACES the LDS vector at 4 (resoff & 0xffff), no second update;
// 32 = the second product ran on strips (two partial vectors), 64 = it also took the soft-abs map (no first barrier here);
// 1 = no second pass (E2 = 0), 2 = log p / P d wanted, 4 = RESIDENT: the state lives in LDS in eigen-coordinates (the trajectory
// kernel) - the updates are theta~'[row] += cx x'[row] and p'[row] += cg lam0 d'[row] on the LDS vectors at 4 (resoff & 0xffff) / 4 (resoff >> 16)
// (0: none), the last product and its barrier do not exist.
__device__ __forceinline__ void fast_chain_body(int offV, int offE1, int offE2, int oVec, int D, int DP, int LD, int flags, float alpha,
                                                float* lam_out, float* lamraw_out, float* x_out, float* upd_x, float cx, float* upd_g, float cg, int resoff) {
  HTA_LDS_BASE();
  D = HTA_U(D); DP = HTA_U(DP); LD = HTA_U(LD); flags = HTA_U(flags); oVec = HTA_U(oVec); resoff = HTA_U(resoff);
  const bool skip2 = flags & 1, has_x = flags & 2, resident = flags & 4, sdraw = flags & 8, assign = flags & 16, strips = flags & 32;
  const bool mapped = !skip2 && (flags & 64);                    // lam~, lam0 d' and their block sums are in LDS already (ph_fast_second_strip)
  const float* V = lds + HTA_U(offV); const float* E1 = lds + HTA_U(offE1); const float* E2 = lds + HTA_U(offE2);
  float* vlam = lds + oVec + DP; float* vlt = vlam + DP; float* vm = vlt + DP; float* vy = vm + DP; float* vx = vy + DP; float* vd = vx + DP;
  float* red3 = vd + 2 * DP + MT / 64;
  typedef __attribute__((address_space(1))) float* gf;
  const int tid = threadIdx.x, wave = HTA_U(tid >> 6);
  const int nq = DP >> 2;
  // !skip2: y0 = m' - E1 m' and the partials of E2^T y0 are in LDS already (ph_fast_second): ONE barrier (the soft-abs map) before w.
  // skip2 (no second pass: E2 = 0): y0 is computed here, under the map.
  const int nbar = (skip2 ? 4 : 3) + (resident ? 0 : 1) - (mapped ? 1 : 0);      // barriers of this call
  if (wave >= 10) {                                              // nothing to compute
    for (int k = 0; k < nbar; ++k) __syncthreads();
    return;
  }
  if (wave >= 8) {                                               // soft-abs map (S:120), log-determinant (S:726), d^T P d, lam0 d'
    if (mapped) {                                                // (done by ph_fast_second_strip's idle waves, under its element-wise step)
      for (int k = 0; k < nbar; ++k) __syncthreads();
      return;
    }
    fast_softabs_rows(tid - 512, wave, D, DP, alpha, vlam, vlt, vd, red3, lam_out, lamraw_out, has_x, skip2);
    for (int k = skip2 ? 1 : 0; k < nbar; ++k) __syncthreads();
    return;
  }
  const int row = tid >> 2, c = tid & 3;
  const bool wr = c == 0 && row < D;
  float ux = 0.f, ug = 0.f;                                      // the rows the updates add to: requested now, used at the end
  if (!resident) {
    if (wr && upd_x) ux = ((gf)upd_x)[row];
    if (wr && upd_g) ug = ((gf)upd_g)[row];
  }
  HTA_WSTAMP(12);
  float y = 0.f;
  if (skip2) {                                                   // y0 = m' - E1 m'
    const float e1m = mv4(E1, LD, vm, nq, DP);
    if (row < D) y = vm[row] - e1m;
    if (c == 0 && row < DP) vy[row] = y;
    __syncthreads();                                             // 1: y0
    HTA_WSTAMP(13);
  } else if (row < DP) {                                         // y = y0 + E2^T y0 from the second product's partials (one per macro-tile row)
    const float* p0 = lds + oVec; const float* p1 = lds + oVec + 3 * DP; const float* p2 = lds + oVec + 5 * DP; const float* p3 = lds + oVec + 8 * DP + MT / 64 + 64;
    y = vy[row] + p0[row];
    if (strips) { if (DP > 64) y += p2[row]; }                   // (ph_fast_second_strip: row tiles 0 .. 3 | 4 ..)
    else {
      if (DP > 32) y += p1[row];
      if (DP > 64) y += p2[row];
      if (DP > 96) y += p3[row];
    }
    if (row >= D) y = 0.f;
  }
  if (!mapped) __syncthreads();                                  // lam~, lam0 d' (x may now be overwritten: a row's partial is read by the quad that writes the row)
  HTA_WSTAMP(14);
  float w = 0.f, qd = 0.f;
  if (row < D) { w = sdraw ? y * sqrtf(vlt[row]) : y / vlt[row]; if (c == 0) qd = y * w; }
  if (c == 0 && row < DP) vx[row] = w;
  {
    const float s1 = wave_sum_dpp(qd);
    if ((tid & 63) == 0) { float* r = red3 + 4 * wave; r[0] = 0.f; r[1] = s1; r[2] = 0.f; }
  }
  __syncthreads();                                               // 3: w
  HTA_WSTAMP(15);
  const float* vsrc = vx;
  if (!skip2) {                                                  // w <- w + E2 w
    w += mv4(E2, LD, vx, nq, DP);
    if (c == 0 && row < DP) vy[row] = (row < D) ? w : 0.f;
    vsrc = vy;
  }
  __syncthreads();                                               // 4
  HTA_WSTAMP(16);
  const float xp = w + mv4(E1, LD, vsrc, nq, DP);                // x' = w + E1 w
  if (resident) {
    if (wr && resoff) {
      float* sx = lds + 4 * (resoff & 0xffff); float* sg = lds + 4 * (resoff >> 16);
      if (assign) sx[row] = xp;
      else { sx[row] += cx * xp; sg[row] += cg * vd[row]; }
    }
    HTA_WSTAMP(18);
    return;
  }
  float* vdst = skip2 ? vy : vx;
  if (c == 0 && row < DP) vdst[row] = (row < D) ? xp : 0.f;
  __syncthreads();                                               // 5
  HTA_WSTAMP(17);
  float x, g;
  mv4_dual(V, LD, vdst, vd, nq, DP, x, g);                       // x = V0 x', P d = V0 (lam0 d')
  if (wr) {
    if (x_out) ((gf)x_out)[row] = x;
    if (upd_x) ((gf)upd_x)[row] = ux + cx * x;
    if (upd_g) ((gf)upd_g)[row] = ug + cg * g;
  }
  HTA_WSTAMP(18);
}
__device__ HTA_PH_ATTR void ph_fast_chain(int offV, int offE1, int offE2, int oVec, int D, int DP, int LD, int flags, float alpha,
                                          float* lam_out, float* lamraw_out, float* x_out, float* upd_x, float cx, float* upd_g, float cg, int resoff) {
  fast_chain_body(offV, offE1, offE2, oVec, D, DP, LD, flags, alpha, lam_out, lamraw_out, x_out, upd_x, cx, upd_g, cg, resoff);
}
__device__ HTA_PH_ATTR void ph_fast_chain_r(int code, float alpha, float cx, float cg, int resoff) {      // (ResLayout; the flags from bit 10, RESIDENT among them)
  const ResLayout r(code);
  fast_chain_body(r.bx, r.bz, r.by, r.oJit, r.D, 16 * kNtCfg3, kLdCfg3, (HTA_U(code) >> 10) | 4, alpha, nullptr, nullptr, nullptr, nullptr, cx, nullptr, cg, resoff);
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

// the fast solve evaluation of system b (see the block comment above ph_fast_vt); false: nothing written, run the general sequence
// res_xm >= 0: RESIDENT operands - d' at LDS offset 4 (res_xm & 0xffff), m' at 4 (res_xm >> 16) (offsets are multiples of 4: 14 bits each) (eigen-coordinates: no V0 product), the updates
// go to the LDS vectors of res_upd (see ph_fast_chain; 0: none)
// sdraw: the evaluation is the momentum DRAW p = G^(1/2) z - the solve's sequence with m = z (Philox normals of the draw's stream, S:183-184's z)
// and w = y sqrt(lam~); the result goes to a.p_out (resident: it REPLACES the vector of res_upd & 0xffff; res_xm then names theta' and the
// place V0^T z is taken from / put to)
__device__ __forceinline__ bool metric_fast_solve(const MetricArgsT<float>& a, int DP, int LD, int64_t b, int& vres, bool bx3, int tiles,
                                                  int res_xm = -1, int res_upd = 0, bool sdraw = false, bool wb = false) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int D = a.D, tid = threadIdx.x;
  const int nt = DP / 16, k4 = (D + 3) / 4;
  float* const lds0 = reinterpret_cast<float*>(smem_raw);
  const int BS = DP * LD > 1024 ? DP * LD : 1024;
  const int oJit = 3 * BS, oLam = oJit + DP, oLt = oLam + DP, oM = oLt + DP, oY = oM + DP, oD = oY + 2 * DP, oRed = oD + 2 * DP, oS = oRed + MT / 64;
  const int bx = vres >= 0 ? vres : BS;                              // V0: where the previous evaluation left it, else staged into buffer 1
  const int by = bx == 0 ? BS : 0, bz = bx == 2 * BS ? BS : 2 * BS;  // F then E2 | E1
  const uint64_t chain = a.chain_offset + (uint64_t)b;
  __syncthreads();
#if HTA_TIMING
  if (threadIdx.x == 0 && blockIdx.x == 0) hta_metric_dbg[31] = clock64() - hta_metric_dbg[24];      // the gap since the previous evaluation's last stamp
#endif
  HTA_STAMP(0);
  if (tid < DP) {
    const int i = opaque_tid();
    const bool in = i < D;
    const float l0 = in ? a.lam0[i] : 0.f;                           // (requested first: the read's latency runs under the jitter's Philox rounds)
    {
      const float ej = (in && a.has_jitter) ? (float)a.jitter * uniform_elem<float>(a.seed, chain, a.draw, PURPOSE_JITTER, a.sub, i) : 0.f;
      lds0[oJit + i] = (wb && bx3 && LD == kLdCfg3) ? sqrtf(ej) : ej;      // (the bfloat16 formation scales BOTH operands: sqrt(e))
    }
    if (res_xm >= 0) {
      lds0[oM + i] = lds0[4 * (res_xm >> 16) + i];
      lds0[oD + i] = lds0[4 * (res_xm & 0xffff) + i];
    } else {
      lds0[oY + 2 * i] = in ? (sdraw ? normal_elem<float>(a.seed, chain, a.draw, 0, i) : a.m[b * D + i]) : 0.f;
      lds0[oY + 2 * i + 1] = (in && a.X) ? a.X[b * D + i] - a.mu[i] : 0.f;
    }
    lds0[oLt + i] = l0;                                              // lam0 (the soft-abs map overwrites it element by element at the end)
    lds0[oLam + i] = 0.f;
  }
  if (vres < 0) ph_stage(a.V0, bx, D, DP, LD);
  vres = bx;
  __syncthreads();
  HTA_STAMP(1);
  if (res_xm < 0) ph_fast_vt(bx, oY, oM, oD, DP, LD);
  HTA_STAMP(2);
  const bool planes = bx3 && LD == kLdCfg3;                          // F as bfloat16 planes for the bfloat16 form of the second product
  const bool packed = planes && wb && res_xm >= 0;                   // the resident evaluations at kLdCfg3: one packed argument (ResLayout)
  const int rcode = (bx == 0 ? 0 : bx == BS ? 1 : 2) | (D << 2);
  const float e1 = packed ? ph_fast_form_r(rcode, tiles) : (planes && wb) ? ph_fast_form<kLdCfg3, true, true>(bx, by, bz, oJit, oLt, oLam, oRed, tiles, k4, D, LD, nt)
                   : planes ? ph_fast_form<kLdCfg3, true>(bx, by, bz, oJit, oLt, oLam, oRed, tiles, k4, D, LD, nt)
                   : LD == kLdCfg3 ? ph_fast_form<kLdCfg3, false>(bx, by, bz, oJit, oLt, oLam, oRed, tiles, k4, D, LD, nt)
                                   : ph_fast_form<0, false>(bx, by, bz, oJit, oLt, oLam, oRed, tiles, k4, D, LD, nt);
  HTA_STAMP(3);
  if (!(e1 <= kSecondE)) return false;
  const bool skip2 = e1 <= kConvE;                                   // (no jitter: F = 0, the shared basis is the answer)
  if (!skip2) {
    float e2;
    // (the bfloat16 form only where the leading dimension is a compile-time constant: the run-time instance needs four registers beyond
    // the caller-saved set, i.e. a save / restore through scratch memory per call that costs more than the product saves)
    if (packed) e2 = ph_fast_second_strip_r(rcode, tiles, (float)a.alpha);
    else if (planes)
      e2 = ph_fast_second_strip<kLdCfg3>(by, bz, oJit, oRed, nt, tiles, D, (float)a.alpha, (res_xm >= 0 || a.X) ? 1 : 0,
                                         (res_xm < 0 && a.lam_out) ? a.lam_out + b * D : nullptr, (res_xm < 0 && a.lamraw_out) ? a.lamraw_out + b * D : nullptr);
    else e2 = LD == kLdCfg3 ? ph_fast_second<kLdCfg3>(by, bz, oJit, oRed, nt, tiles, k4, D, LD)
                            : ph_fast_second<0>(by, bz, oJit, oRed, nt, tiles, k4, D, LD);
    if (!(e2 <= kConvE)) return false;
  }
  HTA_STAMP(9);
  if (packed)
    ph_fast_chain_r(rcode | (((skip2 ? 1 : 0) | 2 | 4 | (sdraw ? 8 | 16 : 0) | 32 | 64) << 10), (float)a.alpha, (float)a.cx, (float)a.cg, res_upd);
  else if (res_xm >= 0)
    ph_fast_chain(bx, bz, by, oJit, D, DP, LD, (skip2 ? 1 : 0) | 2 | 4 | (sdraw ? 8 | 16 : 0) | (planes ? 32 | 64 : 0), (float)a.alpha, nullptr, nullptr, nullptr, nullptr, (float)a.cx, nullptr,
                  (float)a.cg, res_upd);
  else
    ph_fast_chain(bx, bz, by, oJit, D, DP, LD, (skip2 ? 1 : 0) | (a.X ? 2 : 0) | (sdraw ? 8 : 0) | (planes ? 32 | 64 : 0), (float)a.alpha,
                  a.lam_out ? a.lam_out + b * D : nullptr, a.lamraw_out ? a.lamraw_out + b * D : nullptr,
                  sdraw ? a.p_out + b * D : (a.x_out ? a.x_out + b * D : nullptr),
                  a.upd_x ? a.upd_x + b * D : nullptr, (float)a.cx, a.upd_g ? a.upd_g + b * D : nullptr, (float)a.cg, 0);
  HTA_STAMP(21);
  if (tid == 0 && (a.logdet_out || a.quad_out || a.logp_out || a.H_out)) {      // (the half steps of a trajectory want none of the scalars)
    const float* r = lds0 + oS;
    float logdet = 0.f, quad = 0.f, dpd = 0.f;
#pragma unroll
    for (int w = 0; w < 10; ++w) { logdet += r[4 * w]; quad += r[4 * w + 1]; dpd += r[4 * w + 2]; }
    const float logp = (a.X || res_xm >= 0) ? (float)a.log_norm - 0.5f * dpd : 0.f;
    if (a.logdet_out) a.logdet_out[b] = logdet;
    if (a.quad_out) a.quad_out[b] = quad;
    if (a.logp_out) a.logp_out[b] = logp;
    if (a.H_out) {
      const float pi_term = (float)D * 1.8378770351409912f;                              // S:712 in float32
      a.H_out[b] = -logp + 0.5f * pi_term + 0.5f * logdet + 0.5f * quad;                 // S:731
    }
  }
  HTA_STAMP(24);
  return true;
}

// ---- the trajectory kernel's resident state (round 6) ----------------------------------------------------------------------------
// (ga, gb) = (mu + V0 a', V0 b') for two LDS vectors in eigen-coordinates (mu = nullptr: none): the way back to the caller's coordinates
__device__ HTA_PH_ATTR void ph_res_out(int offV, int offA, int offB, float* ga, float* gb, const float* mu, int D, int DP, int LD) {
  HTA_LDS_BASE();
  D = HTA_U(D); DP = HTA_U(DP); LD = HTA_U(LD);
  typedef __attribute__((address_space(1))) float* gf;
  if (threadIdx.x < 512) {
    float x, g;
    mv4_dual(lds + HTA_U(offV), LD, lds + HTA_U(offA), lds + HTA_U(offB), DP >> 2, DP, x, g);
    const int row = threadIdx.x >> 2;
    if ((threadIdx.x & 3) == 0 && row < D) {
      ((gf)ga)[row] = x + (mu ? ((gcf)mu)[row] : 0.f);
      ((gf)gb)[row] = g;
    }
  }
}
// (a', b') = V0^T (ga - mu, gb): into eigen-coordinates ((m_i, d_i) pairs staged at offMD, then ph_fast_vt)
__device__ __forceinline__ void res_in(int offV, int offMD, int offA, int offB, const float* ga, const float* gb, const float* mu, int D, int DP, int LD) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* const lds0 = reinterpret_cast<float*>(smem_raw);
  __syncthreads();
  if ((int)threadIdx.x < DP) {
    const int i = opaque_tid();
    lds0[offMD + 2 * i] = i < D ? gb[i] : 0.f;
    lds0[offMD + 2 * i + 1] = i < D ? ga[i] - (mu ? mu[i] : 0.f) : 0.f;
  }
  __syncthreads();
  ph_fast_vt(offV, offMD, offB, offA, DP, LD);
}

// One evaluation of system b (everything of the file's header); the workgroup's 1024 threads, state in the dynamic LDS block.
// `vres`: the matrix buffer (offset) that holds the staged shared basis V0 on entry, or -1; on return, the buffer that holds it
// now (a solve ends with V0 staged for x = V0 x': the next evaluation of a trajectory kernel starts from that copy), or -1.
__device__ __forceinline__ void metric_warm_system(const MetricArgsT<float>& a, int DP, int LD, int64_t b, int& vres, int second, int tiles) {      // second: bit 0 = the closed-form second pass, bit 1 = its product as three bfloat16 products (the fast solve)
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
  // the momentum draw as a solve-shaped evaluation (tuning key "metric_sqrtdraw"): p = G^(1/2) z = Q diag(sqrt lam~) Q^T z - same law as chol(G) z
  const bool sdraw = (second & 16) && a.p_out && softabs && !general && !a.m && !a.G_out && !a.V_out && !a.dmetric_out;
  const bool has_m = a.m || sdraw;
  // (second & 32: the caller has tried the fast solve itself - metric_warm_mfma_kernel - and it declined)
  if (!(second & 32) && (second & 1) && softabs && !general && has_m && !(a.G_out || (a.p_out && !sdraw) || a.V_out || a.dmetric_out) && (!a.X || a.Pm == a.Hs)) {
    if (metric_fast_solve(a, DP, LD, b, vres, (second & 2) != 0, tiles, -1, 0, sdraw, (second & 64) != 0)) return;
  }

  {
    const uint64_t chain = a.chain_offset + (uint64_t)b;
    const float* V0b = a.V0 + (general ? b * a.v0_stride : 0);
    __syncthreads();
    HTA_STAMP(0);
    // ---- 0. operands: jitter, the solve vector, d = X - mu; V0 into LDS
    if (tid < DP) {
      const int i = opaque_tid();
      vjit[i] = (i < D && a.has_jitter) ? (float)a.jitter * uniform_elem<float>(a.seed, chain, a.draw, PURPOSE_JITTER, a.sub, i) : 0.f;
      vm[i] = (i < D && has_m) ? (sdraw ? normal_elem<float>(a.seed, a.chain_offset + (uint64_t)b, a.draw, 0, i) : a.m[b * D + i]) : 0.f;
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
    if (has_m && softabs) {
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
    const bool want_matrix = a.G_out || (a.p_out && !sdraw) || a.V_out || a.dmetric_out;
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
        if (it == 1 && have_x && (second & 1) && emax_prev <= kSecondE) {
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
      if (has_m) {
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
          const float w = (row < D) ? (sdraw ? y * sqrtf(vlt[row]) : y / vlt[row]) : 0.f;
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
          if (sdraw) a.p_out[b * D + orow] = x;
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
      if (!has_m && tid < DP) vy[tid] = 0.f;                            // u = Q^T m / lam~ (vy holds it after the solve)
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
    if (a.G_out || (a.p_out && !sdraw) || !softabs) {
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

// one evaluation on the general sequence, out of line, its arguments read again from the kernel-argument segment (see traj_general_eval):
// the kernel's own body keeps only what the fast solve reads.  Defined behind load_kernarg, below.
__device__ __attribute__((noinline)) void warm_general_eval(const __attribute__((address_space(4))) char* ka, int DP, int LD, int64_t b, int vres, int second, int tiles);

__global__ __launch_bounds__(MT) void metric_warm_mfma_kernel(MetricArgsT<float> a, int DP, int LD, int second) {
  const int tiles = fast_tiles(DP / 16);
  const __attribute__((address_space(4))) char* ka = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
  // what metric_warm_system's own hook tests, evaluated once per launch: the fast solve covers solves and solve-shaped draws of a soft-abs
  // metric on a shared basis
  const bool softabs = a.metric == 1, general = softabs && a.hs_stride != 0;
  const bool sdraw = (second & 16) && a.p_out && softabs && !general && !a.m && !a.G_out && !a.V_out && !a.dmetric_out;
  const bool fast = (second & 1) && softabs && !general && (a.m || sdraw) && !(a.G_out || (a.p_out && !sdraw) || a.V_out || a.dmetric_out) &&
                    (!a.X || a.Pm == a.Hs);
  for (int64_t b = blockIdx.x; b < a.B; b += gridDim.x) {
    int vres = -1;
    if (fast && metric_fast_solve(a, DP, LD, b, vres, (second & 2) != 0, tiles, -1, 0, sdraw, (second & 64) != 0)) continue;
    warm_general_eval(ka, DP, LD, b, vres, second | 32, tiles);
  }
}

// The evaluations of a trajectory that are NOT the fast solve - the momentum draw, every evaluation of the form that keeps the state in the
// caller's coordinates, the resident form's rare way out - run out of line and read the kernel's arguments again from the kernel-argument
// segment: inlined into the kernel they kept all ~90 argument dwords live across the resident loop (404 scalar registers parked in vector
// lanes, a v_readlane per use - and with 16 waves on the CU every instruction a wave executes costs the workgroup 16 cycles).
// op: the evaluation's index in the trajectory (0 draw, 1 H_old, 2 .. 4 L + 1 half steps, 4 L + 2 H_new); mode 0: the state in global memory
// (t.th / t.pm / t.thc / t.pmc), the copies / the binding rotation after the evaluation included; mode 1: the state is resident in LDS in
// eigen-coordinates - state out, evaluation, state in.  Returns the buffer that holds V0 afterwards (-1: none).
constexpr int kTrajArgOff = (int)((sizeof(MetricArgsT<float>) + 7) / 8 * 8);
typedef const __attribute__((address_space(4))) char* kernarg_ptr;
// (the segment pointer is read in the KERNEL and handed down: __builtin_amdgcn_kernarg_segment_ptr() in a called function returned null here)
template <typename S> __device__ __forceinline__ S load_kernarg(kernarg_ptr ka, int off) {      // dword by dword from the constant address space: scalar loads
  S v;
#if defined(__HIP_DEVICE_COMPILE__)
  typedef const __attribute__((address_space(4))) uint32_t* kp;
  kp src = (kp)(ka + off);
  uint32_t* dst = reinterpret_cast<uint32_t*>(&v);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(S) / 4); ++i) dst[i] = src[i];
#else
  memset(&v, 0, sizeof(v));
#endif
  return v;
}
static_assert(alignof(MetricTrajArgs) == 8 && alignof(MetricArgsT<float>) == 8, "kernel-argument layout of metric_traj_mfma_kernel");
__device__ __attribute__((noinline)) void warm_general_eval(kernarg_ptr ka, int DP, int LD, int64_t b, int vres, int second, int tiles) {
  {
    const uint64_t u = (uint64_t)ka;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    ka = (kernarg_ptr)(((uint64_t)hi << 32) | lo);
  }
  const MetricArgsT<float> a = load_kernarg<MetricArgsT<float>>(ka, 0);
  DP = HTA_U(DP); LD = HTA_U(LD); vres = HTA_U(vres); second = HTA_U(second);
  metric_warm_system(a, DP, LD, b, vres, second, tiles);
}

__device__ __attribute__((noinline)) int traj_general_eval(kernarg_ptr ka, int DP, int LD, int64_t b, int vres, int op, int second, int tiles, int mode) {
  {                                                                 // (uniform: an argument arrives in vector registers)
    const uint64_t u = (uint64_t)ka;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    ka = (kernarg_ptr)(((uint64_t)hi << 32) | lo);
  }
  const MetricArgsT<float> a = load_kernarg<MetricArgsT<float>>(ka, 0);
  const MetricTrajArgs t = load_kernarg<MetricTrajArgs>(ka, kTrajArgOff);
  DP = HTA_U(DP); LD = HTA_U(LD); vres = HTA_U(vres); op = HTA_U(op); second = HTA_U(second); mode = HTA_U(mode);
  const int D = a.D, nops = 4 * t.L + 3;
  const int BS = DP * LD > 1024 ? DP * LD : 1024;
  const int oY = 3 * BS + 4 * DP, oSt = 3 * BS + 8 * DP + MT / 64 + kFastState;
  const int sTh = oSt, sP = oSt + DP, sThc = oSt + 2 * DP, sPc = oSt + 3 * DP;
  MetricArgsT<float> o = a;
  int j = -1;
  if (op == 0) { o.sub = 0; o.p_out = t.pm; }                                                  // gibbs: p ~ N(0, G(theta))  S:183-184
  else if (op == 1) { o.sub = 1; o.X = mode ? t.th : t.cur; o.m = t.pm; o.H_out = t.H0; }      // H_old  S:971
  else if (op == nops - 1) { o.sub = 2u + 8u * (uint32_t)t.L; o.X = t.th; o.m = t.pm; o.H_out = t.H1; o.logp_out = t.lp1; }   // H_new  S:989
  else {
    const int q = op - 2, l = q >> 2;
    j = q & 3;
    const bool fa = j == 0 || j == 3;                                                          // phi_A/2 (S:429-430, S:457-458) : phi_B/2
    o.sub = 2u + 8u * (uint32_t)l + (j == 0 ? 1u : j == 1 ? 2u : j == 2 ? 4u : 7u);
    o.X = fa ? t.th : t.thc; o.m = fa ? t.pmc : t.pm; o.upd_x = fa ? t.thc : t.th; o.upd_g = fa ? t.pm : t.pmc;
    o.cx = t.eh; o.cg = -t.eh;
  }
  const int64_t e0 = b * D;
  if (mode == 1) {
    __syncthreads();
    if (vres < 0) { ph_stage(a.V0, BS, D, DP, LD); vres = BS; __syncthreads(); }
    ph_res_out(vres, sTh, sP, t.th + e0, t.pm + e0, a.mu, D, DP, LD);
    ph_res_out(vres, sThc, sPc, t.thc + e0, t.pmc + e0, a.mu, D, DP, LD);
    __syncthreads();
    metric_warm_system(o, DP, LD, b, vres, second & 2, tiles);                                 // (bit 0 off: not the fast solve again)
    __syncthreads();
    if (vres < 0) { ph_stage(a.V0, BS, D, DP, LD); vres = BS; }
    res_in(vres, oY, sTh, sP, t.th + e0, t.pm + e0, a.mu, D, DP, LD);
    res_in(vres, oY, sThc, sPc, t.thc + e0, t.pmc + e0, a.mu, D, DP, LD);
    return vres;
  }
  metric_warm_system(o, DP, LD, b, vres, second, tiles);
  if ((op == 1 || j == 1) && !(second & 8)) {                                                  // (second & 8: the caller is resident and does this itself)
    __syncthreads();
    const int i = opaque_tid();
    if (i < D) {
      const int64_t e = e0 + i;
      if (op == 1) {                                                                           // S:425-426
        const float x = t.cur[e];
        t.th[e] = x; t.thc[e] = x; t.pmc[e] = t.pm[e];
      } else {
        phi_c_elem<float>(t.th[e], t.pm[e], t.thc[e], t.pmc[e], t.c, t.s);                     // phi_C  S:447-450
      }
    }
  }
  return vres;
}

// The chain's Metropolis selection at the end of its trajectory (mh_select_kernel's rule, uniform and writes - hmc_pieces.hip - by the chain's own
// workgroup: the proposal t.th and the three scalars were written by this workgroup, visible after the barrier)
__device__ __forceinline__ void traj_select(const MetricArgsT<float>& a, const MetricTrajArgs& t, int64_t b) {
  __syncthreads();
  const int D = a.D, j = opaque_tid();
  const uint64_t chain = a.chain_offset + (uint64_t)b;
  const float u = u23<float>(philox_block(a.seed, chain, (uint32_t)t.n, PURPOSE_MH, 0, 0).x);
  const bool acc = mh_accept<float>(t.H0[b], t.H1[b], t.lp1[b], u);
  const bool reset = (!acc) && (t.n == t.burn + 1);                  // SURVEY Q2 (samplers.py:1018)
  const int64_t e = b * D + j;
  if (j < D) {
    const float v = acc ? t.th[e] : (reset ? t.init[e] : t.cur[e]);
    t.cur[e] = v;
    if (t.row && t.n > t.burn) t.row[e] = v;
  }
  if (j == 0) {
    if (!acc) t.rej[b] += 1;
    if (t.acc) t.acc[b] = acc ? 1 : 0;
  }
}

__global__ __launch_bounds__(MT) void metric_traj_mfma_kernel(MetricArgsT<float> a, MetricTrajArgs t, int DP, int LD, int second) {
  const int D = a.D;
  const int nops = 4 * t.L + 3;
  const int tiles = fast_tiles(DP / 16);
  const kernarg_ptr ka = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
  // RESIDENT mode (round 6, second & 4: tuning key "metric_resident"): after the momentum draw the chain's four state vectors are taken
  // into the eigenbasis ONCE - theta' = V0^T (theta - mu), p' = V0^T p - and live in LDS for the trajectory: a solve evaluation then has
  // no V0 product at either end (m' and d' ARE the state, x' and lam0 d' update it element-wise), no global read or write but its
  // scalars; the binding rotation is element-wise in any orthonormal basis.  theta = mu + V0 theta' once at the end.  The rounding of
  // the 4 L + 2 round trips through V0 is gone, not added: results agree with the launch sequence to fp32 rounding, not bit for bit
  // (second & 4 == 0 keeps the bit-identical form: tests/test_gpu_rmhmc.py::test_trajectory_kernel_equals_the_launch_sequence).
  const bool resident = (second & 4) && (second & 1) && a.metric == 1 && a.hs_stride == 0 && a.Pm == a.Hs && a.V0 && a.lam0;
  const int BS = DP * LD > 1024 ? DP * LD : 1024;
  const int oY = 3 * BS + 4 * DP, oSt = 3 * BS + 8 * DP + MT / 64 + kFastState;
  const int sTh = oSt, sP = oSt + DP, sThc = oSt + 2 * DP, sPc = oSt + 3 * DP;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* const lds0 = reinterpret_cast<float*>(smem_raw);
  for (int64_t b = blockIdx.x; b < a.B; b += gridDim.x) {
    int vres = -1;                       // the buffer a solve left the staged V0 in: the next evaluation starts from it
    if (!resident) {
      for (int op = 0; op < nops; ++op) vres = traj_general_eval(ka, DP, LD, b, vres, op, second, tiles, 0);
      if (t.select) traj_select(a, t, b);
      continue;
    }
    const int64_t e0 = b * D;
    bool drawn = false;
    if (second & 16) {
      // the momentum draw p = G^(1/2) z in eigen-coordinates: (theta', z') = V0^T (cur - mu, z) in one pass, then the solve's sequence with
      // m' = z' and w = y sqrt(lam~); its x' IS p' = V0^T p (no way back through V0)
      __syncthreads();
      ph_stage(a.V0, BS, D, DP, LD); vres = BS;
      if ((int)threadIdx.x < DP) {
        const int i = opaque_tid();
        lds0[oY + 2 * i] = i < D ? normal_elem<float>(a.seed, a.chain_offset + (uint64_t)b, a.draw, 0, i) : 0.f;
        lds0[oY + 2 * i + 1] = i < D ? t.cur[e0 + i] - a.mu[i] : 0.f;
      }
      __syncthreads();
      ph_fast_vt(vres, oY, sP, sTh, DP, LD);
      MetricArgsT<float> o = a;
      o.sub = 0; o.X = nullptr; o.m = nullptr; o.p_out = t.pm;
      drawn = metric_fast_solve(o, DP, LD, b, vres, (second & 2) != 0, tiles, (sTh >> 2) | ((sP >> 2) << 16), sP >> 2, true, (second & 64) != 0);
    }
    if (!drawn) {
      vres = traj_general_eval(ka, DP, LD, b, vres, 0, second, tiles, 0);                      // the draw on the general sequence (p in t.pm)
      // into the eigenbasis: (theta', p') = V0^T (cur - mu, pm)
      __syncthreads();
      if (vres < 0) { ph_stage(a.V0, BS, D, DP, LD); vres = BS; }
      res_in(vres, oY, sTh, sP, t.cur + e0, t.pm + e0, a.mu, D, DP, LD);
    }
    __syncthreads();
    if ((int)threadIdx.x < DP) { const int i = opaque_tid(); lds0[sThc + i] = lds0[sTh + i]; lds0[sPc + i] = lds0[sP + i]; }   // theta~' = theta', p~' = p'
    __syncthreads();
    for (int op = 1; op < nops; ++op) {
      // the fields the fast solve reads (everything else of `a` is dead here)
      MetricArgsT<float> o = a;
      o.X = t.cur; o.m = t.pm;                                                                 // (non-null: the operands themselves are the resident vectors)
      int j = -1;
      bool fa = false;
      if (op == 1) { o.sub = 1; o.H_out = t.H0; }                                              // H_old  S:971
      else if (op == nops - 1) { o.sub = 2u + 8u * (uint32_t)t.L; o.H_out = t.H1; o.logp_out = t.lp1; }   // H_new  S:989
      else {
        const int q = op - 2, l = q >> 2;
        j = q & 3;
        fa = j == 0 || j == 3;                                                                 // phi_A/2 (S:429-430, S:457-458) : phi_B/2
        o.sub = 2u + 8u * (uint32_t)l + (j == 0 ? 1u : j == 1 ? 2u : j == 2 ? 4u : 7u);
        o.cx = t.eh; o.cg = -t.eh;
      }
      const bool step = j >= 0;
      const int rx = (step && !fa) ? sThc : sTh, rm = (step && fa) ? sPc : sP;
      const int upd = step ? (((fa ? sThc : sTh) >> 2) | (((fa ? sP : sPc) >> 2) << 16)) : 0;
      if (!metric_fast_solve(o, DP, LD, b, vres, (second & 2) != 0, tiles, (rx >> 2) | ((rm >> 2) << 16), upd, false, (second & 64) != 0))
        vres = traj_general_eval(ka, DP, LD, b, vres, op, second | 8, tiles, 1);                   // (rare: a first pass above kSecondE, a non-finite state)
      if (j == 1) {                                                                            // phi_C  S:447-450 (element-wise: any orthonormal basis)
        __syncthreads();
        const int i = opaque_tid();
        if (i < D) phi_c_elem<float>(lds0[sTh + i], lds0[sP + i], lds0[sThc + i], lds0[sPc + i], t.c, t.s);
      }
    }
    // back: theta = mu + V0 theta' (and the final momentum, as the other form leaves it)
    __syncthreads();
    ph_res_out(vres, sTh, sP, t.th + e0, t.pm + e0, a.mu, D, DP, LD);
    if (t.select) traj_select(a, t, b);
  }
}

int g_metric_select = 1;     // tuning key "metric_select": 1 = the trajectory kernel ends with the chain's Metropolis selection (one launch per trajectory), 0 = mh_select is a second launch
int g_metric_resident = 1;   // tuning key "metric_resident": 1 = the trajectory kernel keeps the chain's state in LDS in eigen-coordinates, 0 = in the caller's (bit-identical to the launch sequence)
int g_metric_traj = 1;   // tuning key "metric_traj": 1 = a trajectory of the eigendecomposition route is one launch, 0 = one launch per evaluation

bool metric_traj_mfma_eligible(const MetricArgsT<float>& a) {
  return g_metric_traj && a.hs_stride == 0 && !a.V_out && !a.dmetric_out && !a.G_out && metric_warm_mfma_eligible(a);
}

int metric_traj_mfma(const MetricArgsT<float>& a, const MetricTrajArgs& t, hipStream_t s) {
  const int D = a.D;
  const int DP = (D + 15) / 16 * 16, LD = DP + 4;
  const size_t lds = ((size_t)3 * (DP * LD > 1024 ? DP * LD : 1024) + 8 * DP + MT / 64 + kFastScratch) * sizeof(float);
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
  metric_traj_mfma_kernel<<<grid, MT, lds, s>>>(k, t, DP, LD, (g_metric_second ? 1 : 0) | (g_metric_bx3 ? 2 : 0) | (g_metric_resident ? 4 : 0) | (g_metric_sqrtdraw ? 16 : 0) | (g_metric_bx3 >= 2 ? 64 : 0));
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
  const size_t lds = ((size_t)3 * (DP * LD > 1024 ? DP * LD : 1024) + 8 * DP + MT / 64 + kFastScratch) * sizeof(float);    // = oW + kFastScratch floats (the Cholesky's [16][20] panel scratch and the fast solve's partials share them)
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
  metric_warm_mfma_kernel<<<grid, MT, lds, s>>>(k, DP, LD, (g_metric_second ? 1 : 0) | (g_metric_bx3 ? 2 : 0) | (g_metric_sqrtdraw ? 16 : 0) | (g_metric_bx3 >= 2 ? 64 : 0));
  profile_end(s);
  HTA_CHECK_LAUNCH("hta_metric_eval (mfma)");
  return HTA_OK;
}

}  // namespace hta
