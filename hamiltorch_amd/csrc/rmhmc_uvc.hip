#define HTA_PHILOX_MAD64 1      // philox.hpp: 64-bit products (this file's kernels have the registers for them)
// Explicit RMHMC for a Gaussian target on the identity-soft-abs path (rmhmc_fused.hip: why that path exists; the integrator is
// hamiltorch/samplers.py:425-461 inside the sampler loop S:969-1026), ONE chain per four-wave workgroup: the kernel of BASELINE
// config 3 (256 chains on 256 CUs).  Round 4: the third generation of rmhmc_uv_kernel<1>, built from what the ablation builds of
// that kernel measured (profiles/r04b_uv_ablation.txt; per step at 256 chains: 1.01 us in the 312 matrix instructions, 0.65 us in
// the five LDS publish / barrier / fetch round trips, 0.94 us issuing the ~570 other instructions of a lone wave):
//
// (1) COMPACT ELEMENT-WISE LAYOUT.  The product leaves four rows per lane, twice over (both contraction parities) and in four
//     columns of which one chain fills two (its state set U = (theta, p) and the copy V = (theta_c, p_c), S:425-426): the
//     element-wise work of rmhmc_uv_kernel<1> ran every instruction four times for one useful value in four.  Here the eight
//     lanes of a row block (4 columns x 2 parities) split its 4 rows x 2 sets: lane (column cl, parity kp) OWNS row
//     row0 + 2 (cl >> 1) + kp of set cl & 1 - one value per lane, every element-wise instruction once, every lane useful.  Going
//     from the accumulator to that layout costs four selects, two DPP adds (the parity sum, needed anyway) and one select with a
//     quad permutation.
// (2) THREE PRODUCT PHASES PER STEP INSTEAD OF FIVE.  A half step solves x = (P + E)^-1 g, E = diag(jitter u), by K = 2
//     refinements from x0 = z = S g (S = P^-1 shared): x2 = z - S(e . (z - S(e . z))) = z - r1 + c2, r1 = S(e . z), c2 = S(e . r1).
//     c2 is second order in rho = jitter / lambda_min (cfg3: 2e-3): it only has to be ADDED to theta, and products that take a
//     vector carrying c2 as input change by third-order terms - below the rounding the host picked K for (rho^(K+1) <= eps / 4).
//     So a solve is ONE phase (r1), and c2 = S q, q = e . r1, rides in the two idle columns of the NEXT phase's instruction (a
//     one-chain group leaves columns 2, 3 of v_mfma_f32_4x4x1_16b free); when it arrives it is added where theta entered since
//     (z -= eh (theta - mu): a multiple of c2; the rotation S:447-450 is linear: its image of (c2_U, c2_V) is added to the rotated
//     state; P c2 = q needs no product; the momentum part of that image goes through S once more, again in idle columns).
//     Per step: phase 1 (A pair: r1, and c2 of the previous B pair), phase 2 (after the rotation: y = P (theta - mu), z = S p
//     afresh, and c2 of the A pair), phase 3 (B pair: r1, and S of the rotation's momentum correction) - 208 matrix instructions
//     and three barriers instead of 312 and five; one flush phase per trajectory.  The result equals the K = 2 iteration up to
//     third-order terms (oracle/ prototype: identical error against the exact solve, 1.9e-9 in float64 at cfg3's sizes).
// Everything else as rmhmc_uv_kernel: same Philox streams, same update order (Q1 sequential rotation, Q2 reset, Q4 acceptance on
// the un-augmented pair), momenta drawn ahead, the V column of a trajectory's last Hamiltonian evaluates the next trajectory's
// momentum terms.  Only launched for K == 2 with jitter; anything else stays on rmhmc_uv_kernel.
#include <math.h>
#include "rmhmc_fused_dev.hpp"

namespace hta {

constexpr int CBUF = 7;            // LDS vector matrices: DV GV EV W0 W1 WA WB

template <int CTRL> __device__ __forceinline__ float quad_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

template <bool CO>
__global__ __launch_bounds__(XNT, CO ? 2 : 1) void rmhmc_uvc_kernel(FusedArgs<float> a) {
  typedef float T;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* lds = reinterpret_cast<T*>(smem_raw);
  constexpr int MSZ = XNC * XLD;
  T* DV = lds;                     // theta - mu (operand of P: refresh phase, Hamiltonians)
  T* GV = DV + MSZ;                // momenta (operand of S there); columns 2, 3: the A pair's q
  T* EV = GV + MSZ;                // Hamiltonian: the jitter
  T* W0 = EV + MSZ; T* W1 = W0 + MSZ;   // Hamiltonian: refinement vectors
  T* WA = W1 + MSZ;                // A pair: e . z | q of the previous B pair
  T* WB = WA + MSZ;                // B pair: e . z | the rotation's momentum correction
  T* red = WB + MSZ;               // [XWV][2][4]
  // lane bits: [1:0] column, [2] low bit of the row block, [3] contraction parity, [5:4] 16-lane group
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, cl = l & 3, grp = l >> 4;
  const int kpar = (l >> 3) & 1, rb = 2 * grp + ((l >> 2) & 1);
  const bool khi = kpar != 0, hih = (cl >> 1) != 0, setV = (cl & 1) != 0;
  const int D = a.D;
  const int row0 = 32 * w + 4 * rb, arow = row0 + cl;     // this lane SUPPLIES matrix row arow ...
  const int row = row0 + 2 * (cl >> 1) + kpar;            // ... and OWNS row `row` of set cl & 1
  const bool rok = row < D, rok_p = (row ^ 1) < D;         // (row ^ 1: the row of the lane at the other parity)
  T Sa[XKJ], Pa[XKJ];
#pragma unroll
  for (int j = 0; j < XKJ; ++j) {
    const int k = 2 * j + kpar;
    const bool ok = arow < D && k < D;
    Sa[j] = ok ? a.S[(int64_t)k * D + arow] : 0.f;        // symmetric: column arow, coalesced over the lanes
    Pa[j] = ok ? a.P[(int64_t)k * D + arow] : 0.f;
  }
  const T mu_r = rok ? a.mu[row] : 0.f, sd_r = rok ? a.S[(int64_t)row * D + row] : 0.f;
  for (int e = tid; e < CBUF * MSZ + XWV * 2 * 4; e += XNT) lds[e] = 0.f;
  const T eh = 0.5f * a.eps;
  const T ehU = setV ? 0.f : eh, ehV = setV ? eh : 0.f;   // the set that moves first in S:429-433 is U, in S:454-458 V
  const T zc2 = setV ? 0.f : -2.f * eh * eh;              // what z_U misses while c2 of a B pair is in flight (two kicks: S:458, S:429)
  // the rotation S:447-450 as a linear map of (d theta, 0, d theta_c, 0) (sequential, Q1): this set's rows of it
  const T hc = 0.5f, rc = a.rot_c, rs = a.rot_s;
  const T t_u = hc * (1.f + rc), t_v = hc * (1.f - rc);                                   // theta'   = t_u du + t_v dv
  const T p_u = -hc * rs * t_u, p_v = -hc * rs * (t_v - 1.f);                             // p'       = -s/2 (theta' - dv)
  const T c_u = hc * (t_u - rc * t_u - rs * p_u), c_v = hc * ((t_v + 1.f) - rc * (t_v - 1.f) - rs * p_v);   // theta_c'
  const T q_u = hc * (p_u + rs * (t_u - c_u) - rc * p_u), q_v = hc * (p_v + rs * (t_v - c_v) - rc * p_v);   // p_c'
  // this lane's coefficients on (own c2, partner's c2), eh folded in: X and g corrections
  const T kXo = eh * (setV ? c_v : t_u), kXp = eh * (setV ? c_u : t_v);
  const T kGo = eh * (setV ? q_v : p_u), kGp = eh * (setV ? q_u : p_v);
  const int own_off = (cl & 1) * XLD + (row & 1) * XHL + (row >> 1), def_off = own_off + 2 * XLD;
  const int b_off = cl * XLD + kpar * XHL + 4 * grp;      // this group's chunk of a super-chunk of four
  uint64_t chain = 0;
  bool live = false;

  auto partner = [&](T v) { return quad_dpp<0xB1>(v); };    // the other set's value of the same row: quad_perm [1,0,3,2]
  auto of_set_u = [&](T v) { return quad_dpp<0xA0>(v); };   // [0,0,2,2]
  auto of_set_v = [&](T v) { return quad_dpp<0xF5>(v); };   // [1,1,3,3]
  // the owned value of (set cl & 1) out of the instruction's columns 0 / 1, and that of columns 2 / 3 (the deferred products)
  auto extract2 = [&](const bf4& acc, T& main, T& defer) {
    const T m01 = khi ? acc[1] : acc[0], o01 = khi ? acc[0] : acc[1];
    const T m23 = khi ? acc[3] : acc[2], o23 = khi ? acc[2] : acc[3];
    const T s01 = m01 + other_parity(o01), s23 = m23 + other_parity(o23);   // rows row0 + kp, row0 + 2 + kp of THIS lane's column
    // (the cross-lane reads stand OUTSIDE the selects: inside an arm they would run under that arm's lane mask)
    const T lo23 = quad_dpp<0x44>(s23);                    // [0,1,0,1]: columns 2, 3 take the rows 2 + kp of columns 0, 1
    const T hi01 = quad_dpp<0xEE>(s01);                    // [2,3,2,3]: columns 0, 1 take the rows kp of columns 2, 3
    main = hih ? lo23 : s01;
    defer = hih ? s23 : hi01;
  };
  auto extract1 = [&](const bf4& acc) -> T {
    const T m01 = khi ? acc[1] : acc[0], o01 = khi ? acc[0] : acc[1];
    const T m23 = khi ? acc[3] : acc[2], o23 = khi ? acc[2] : acc[3];
    const T s01 = m01 + other_parity(o01), s23 = m23 + other_parity(o23);
    const T lo23 = quad_dpp<0x44>(s23);
    return hih ? lo23 : s01;
  };
  auto fetch = [&](const T* X, bf4 (&c)[XSQ]) {
#pragma unroll
    for (int Q = 0; Q < XSQ; ++Q) c[Q] = *reinterpret_cast<const bf4*>(X + b_off + 16 * Q);
  };
  // A1 X1 and A2 X2 for all four columns (two accumulator chains each product would need s_nops: the two products interleave)
  auto prod2 = [&](const T (&A1)[XKJ], const T* X1, const T (&A2)[XKJ], const T* X2, bf4& acc1, bf4& acc2) {
    bf4 c1[XSQ], c2[XSQ];
    fetch(X1, c1);
    fetch(X2, c2);
    __builtin_amdgcn_sched_barrier(0);
    static_for(std::make_integer_sequence<int, XQ>{}, [&](auto qc) {
      constexpr int q = decltype(qc)::value;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc1 = mfma_from_group<q % 4>(A1[4 * q + u], c1[q / 4][u], acc1);
        acc2 = mfma_from_group<q % 4>(A2[4 * q + u], c2[q / 4][u], acc2);
      }
    });
  };
  // one product on two accumulator chains (k in the order 0 2 | 1 3 of every chunk)
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
  };
  // this lane's element of one jitter vector (uniform_elem layout: rows 4b..4b+3 are Philox block b)
  auto jitter_one = [&](uint32_t n, uint32_t sub) -> T {
    const U4 r = philox_block(a.seed, chain, n, PURPOSE_JITTER, sub, (uint32_t)(row0 >> 2));
    const uint32_t t0 = hih ? r.z : r.x, t1 = hih ? r.w : r.y;
    return (live && rok) ? a.jitter * u23<T>(khi ? t1 : t0) : 0.f;
  };
  // the two jitter vectors of a step's solves for this set (sub-streams subA, subB): the even parity draws the block of the A
  // pair, the odd parity that of the B pair, each hands the other the element of the other's row - one Philox pass per step
  auto jitter_pair = [&](uint32_t n, uint32_t subA, uint32_t subB, T& eA, T& eB) {
    const U4 r = philox_block(a.seed, chain, n, PURPOSE_JITTER, khi ? subB : subA, (uint32_t)(row0 >> 2));
    const uint32_t t0 = hih ? r.z : r.x, t1 = hih ? r.w : r.y;
    const T mine = (live && rok) ? a.jitter * u23<T>(khi ? t1 : t0) : 0.f;
    const T oth = other_parity((live && rok_p) ? a.jitter * u23<T>(khi ? t0 : t1) : 0.f);
    eA = khi ? oth : mine;
    eB = khi ? mine : oth;
  };
  // three sums per set over the rows, complete in every lane of the set
  auto set_sums = [&](T (&v)[3]) {
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      v[e] += quad_dpp<0x4E>(v[e]);                        // [2,3,0,1]: the same set's other row pair
      v[e] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[e]), 0x124 /* row_ror:4 */, 0xf, 0xf, false));
      v[e] += other_parity(v[e]);                          // row_ror:8: with the step before, all four quads of the 16-lane row
      v[e] += __shfl_xor(v[e], 16, 64);
      v[e] += __shfl_xor(v[e], 32, 64);
    }
    __syncthreads();
    if (l < 2) {
#pragma unroll
      for (int e = 0; e < 3; ++e) red[(w * 2 + l) * 4 + e] = v[e];
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      T s = 0.f;
#pragma unroll
      for (int i = 0; i < XWV; ++i) s += red[(i * 2 + (cl & 1)) * 4 + e];
      v[e] = s;
    }
  };
  // H = -log p + D/2 log 2 pi + 1/2 log|G| + 1/2 g^T G^-1 g  (S:731) of this set's (X, g); also returns P (X - mu) and S g
  // (n, sub: per lane - in the Hamiltonian at a trajectory's end the V set evaluates the NEXT trajectory's momentum terms)
  // The kinetic term of the K = 2 solve needs ONE product: with x2 = z - r1 + c2, z = S g, r1 = S (e . z), c2 = S (e . r1) and S
  // symmetric, g . r1 = z . (e . z) and g . c2 = z . (e . r1), so g . x2 = sum_i [g_i z_i - e_i z_i (z_i - r1_i)].
  auto hamiltonian = [&](uint32_t n, uint32_t sub, T X, T g, T& H, T& logp, T& Pd_out, T& Sg_out, T& kin_out, T& ld_out) {
    const T ev = a.has_jitter ? jitter_one(n, sub) : 0.f, dr = X - mu_r;
    EV[own_off] = ev;
    DV[own_off] = dr;
    GV[own_off] = g;
    __syncthreads();
    bf4 Pdv = {0.f, 0.f, 0.f, 0.f}, x0v = {0.f, 0.f, 0.f, 0.f}, s2v = {0.f, 0.f, 0.f, 0.f};
    prod2(Pa, DV, Sa, GV, Pdv, x0v);
    const T Pd = extract1(Pdv), x0 = extract1(x0v);
    T s2 = 0.f;
    if (a.has_jitter) { prod1(Sa, EV, true, s2v); s2 = extract1(s2v); }     // second-order log-det term: (S . S) e
    Pd_out = Pd; Sg_out = x0;
    T v[3];
    v[0] = dr * Pd;
    v[2] = a.has_jitter ? ev * (sd_r - 0.5f * s2) : 0.f;                     // log|P + E| = log|P| + tr(SE) - 1/2 tr((SE)^2) + ...
    W0[own_off] = ev * x0;
    __syncthreads();
    bf4 rv = {0.f, 0.f, 0.f, 0.f};
    prod1(Sa, W0, false, rv);
    const T r1 = extract1(rv);
    v[1] = fmaf(g, x0, -(ev * x0) * (x0 - r1));
    set_sums(v);
    const float pi_term = (float)D * 1.8378770351409912f;   // S:712
    logp = a.log_norm - 0.5f * v[0];
    H = -logp + 0.5f * pi_term + 0.5f * (a.logdetP + v[2]) + 0.5f * v[1];
    kin_out = v[1]; ld_out = v[2];
  };

  for (int64_t c = blockIdx.x; c < a.C; c += gridDim.x) {
    live = true;
    chain = a.chain_offset + (uint64_t)c;
    T scur = rok ? a.cur[c * D + row] : 0.f;
    int rmask = -(int)rok;
    asm volatile("" : "+v"(rmask));
    // this lane's row of the pre-drawn momentum of local trajectory tt (a lane without a row reads element 0 and discards it)
    auto momentum_raw = [&](int tt) -> T { return a.p_ws[((int64_t)tt * a.C + c) * D + (row & rmask)]; };
    auto momentum_use = [&](T v) -> T { asm volatile("" : "+v"(v)); return rok ? v : 0.f; };
    int32_t rejected = 0;
    __syncthreads();                                        // the previous chain's last reads of the vector matrices
    bool have_next = false;
    T gn = 0.f, y_next = 0.f, z_next = 0.f, H0_next = 0.f, lp_next = 0.f;
    for (int t = 0; t < a.n_traj; ++t) {
      const uint32_t n = (uint32_t)(a.traj_offset + t);
      // ---- gibbs: p = chol(G(theta)) z, drawn ahead (S:183-184); theta_c = theta, p_c = p (S:425-426)
      T H0, H1, lp0, lp1, kin, ld, X, g, y, z;
      if (have_next) {
        g = gn; X = scur; y = y_next; z = z_next; H0 = H0_next; lp0 = lp_next;
      } else {
        g = momentum_use(momentum_raw(t));
        X = scur;
        hamiltonian(n, 1, X, g, H0, lp0, y, z, kin, ld);    // S:971 -> S:822
      }
      const bool pre = t + 1 < a.n_traj;
      T gn_raw = 0.f;
      if (pre) gn_raw = momentum_raw(t + 1);                // a whole trajectory ahead of its first use
      const T y_start = y;
      T qB = 0.f;                                           // e . r1 of the last B pair: its c2 = S qB is still owed to theta
      for (int lstep = 0; lstep < a.L; ++lstep) {           // S:427-461
        const uint32_t k0 = 2u + 8u * (uint32_t)lstep;
        // S:429-433: U moves first and solves with sub-stream k0 + 2, V with k0 + 1; S:454-458: V first (k0 + 7), U second (k0 + 4)
        T eA, eB;
        jitter_pair(n, setV ? k0 + 1u : k0 + 2u, setV ? k0 + 7u : k0 + 4u, eA, eB);
        // ---- A pair  (phi_A/2, phi_B/2  S:429-433)
        g = fmaf(-ehU, y, g);
        z = fmaf(-ehU, X - mu_r, z);
        const T aA = eA * z;
        WA[own_off] = aA;
        WA[def_off] = qB;
        __syncthreads();
        bf4 r = {0.f, 0.f, 0.f, 0.f};
        prod1(Sa, WA, false, r);
        T r1, c2;
        extract2(r, r1, c2);
        X = fmaf(eh, c2, X);                                // the previous B pair's second-order term, one phase late
        z = fmaf(zc2, c2, z);
        const T qA = eA * r1;
        X = fmaf(eh, z - r1, X);                            // x1 = z - r1
        y = fmaf(eh, g - (aA - qA), y);                     // P x2 = g - e . x1,  e . x1 = e . z - e . r1
        g = fmaf(-ehV, y, g);                               // (z of the second mover is evaluated afresh below)
        // ---- phi_C  S:447-450, sequential (Q1), both sets compute it; theta still lacks eh c2 of the A pair
        {
          const T pX = partner(X), pg = partner(g);
          T xx = setV ? pX : X, b = setV ? pg : g, xc = setV ? X : pX, bc = setV ? g : pg;
          xx = hc * ((xx + xc) + rc * (xx - xc) + rs * (b - bc));
          b = hc * ((b + bc) - rs * (xx - xc) + rc * (b - bc));
          xc = hc * ((xx + xc) - rc * (xx - xc) - rs * (b - bc));
          bc = hc * ((b + bc) + rs * (xx - xc) - rc * (b - bc));
          X = setV ? xc : xx; g = setV ? bc : b;
        }
        DV[own_off] = X - mu_r;
        GV[own_off] = g;
        GV[def_off] = qA;
        __syncthreads();
        bf4 p1 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
        prod2(Pa, DV, Sa, GV, p1, s1);
        y = extract1(p1);
        T c2a;
        extract2(s1, z, c2a);
        // the rotation's image of (eh c2_U, 0, eh c2_V, 0): theta and momentum parts; P c2 = q (no product)
        const T pc2 = partner(c2a), pq = partner(qA);
        X = fmaf(kXo, c2a, fmaf(kXp, pc2, X));
        y = fmaf(kXo, qA, fmaf(kXp, pq, y));
        const T dg = fmaf(kGo, c2a, kGp * pc2);
        g += dg;                                            // z = S g still lacks S dg: next phase
        // ---- B pair  (phi_B/2, phi_A/2  S:454-458)
        g = fmaf(-ehV, y, g);
        z = fmaf(-ehV, X - mu_r, z);
        const T aB = eB * z;
        WB[own_off] = aB;
        WB[def_off] = dg;
        __syncthreads();
        bf4 r2 = {0.f, 0.f, 0.f, 0.f};
        prod1(Sa, WB, false, r2);
        T r1b, sdg;
        extract2(r2, r1b, sdg);
        z += sdg;
        qB = eB * r1b;
        X = fmaf(eh, z - r1b, X);
        y = fmaf(eh, g - (aB - qB), y);
        g = fmaf(-ehU, y, g);
        z = fmaf(-ehU, X - mu_r, z);
      }
      // ---- flush: c2 of the last B pair.  (It could ride in the idle columns of the closing Hamiltonian's first product, with
      // P (theta - mu) += eh qB through P S = 1 - measured +3 % - but then the P (theta' - mu) a trajectory hands to the next one
      // is not the product a fresh launch computes from theta': results would depend, in the last bit, on how a run is cut
      // into launches.  tests/test_gpu_rmhmc.py::test_fused_workspace_passes_are_equivalent holds that line.)
      {
        WA[own_off] = 0.f;
        WA[def_off] = qB;
        __syncthreads();
        bf4 r = {0.f, 0.f, 0.f, 0.f};
        prod1(Sa, WA, false, r);
        T r1, c2;
        extract2(r, r1, c2);
        X = fmaf(eh, c2, X);
      }
      // ---- H_new on the un-augmented pair = set U (S:989, Q4); in the V set: the next trajectory's momentum terms
      if (pre) gn = momentum_use(gn_raw);
      const T Xh = of_set_u(X);                              // both sets at the proposal theta': the V set's P (theta' - mu) is not used
      const bool nextcol = setV && pre;
      const T gu = of_set_u(g);
      const T gh = nextcol ? gn : gu;
      T Pd1, Sg1;
      hamiltonian(nextcol ? n + 1u : n, nextcol ? 1u : 2u + 8u * (uint32_t)a.L, Xh, gh, H1, lp1, Pd1, Sg1, kin, ld);
      // ---- Metropolis test + bookkeeping (S:1000-1026, S:1045-1057) on the U set's values, mirrored in the V set
      const T H0u = of_set_u(H0), H1u = of_set_u(H1), lp1u = of_set_u(lp1);
      const T u = u23<T>(philox_block(a.seed, chain, n, PURPOSE_MH, 0, 0).x);
      const bool acc = mh_accept<T>(H0u, H1u, lp1u, u);
      const bool reset = (!acc) && ((int)n == a.burn + 1);  // Q2
      have_next = pre && !reset;
      {                                                      // H_old, y, z of trajectory t + 1 (the expression of hamiltonian(), same order)
        // (cross-lane reads outside the branch: they must run with every lane of the quad enabled)
        const T lp0u = of_set_u(lp0), Pd1u = of_set_u(Pd1), ldv = of_set_v(ld), kinv = of_set_v(kin), Sg1v = of_set_v(Sg1);
        if (have_next) {
          const float pi_term = (float)D * 1.8378770351409912f;
          lp_next = acc ? lp1u : lp0u;
          H0_next = -lp_next + 0.5f * pi_term + 0.5f * (a.logdetP + ldv) + 0.5f * kinv;
          y_next = acc ? Pd1u : y_start;                     // (both sets start a trajectory with the same y)
          z_next = Sg1v;
        }
      }
      if (rok) {
        const T vnew = acc ? Xh : (reset ? a.theta_init[c * D + row] : scur);
        scur = vnew;
        if (!setV && a.samples && (int)n > a.burn) a.samples[((int64_t)((int)n - a.burn) * a.C + c) * D + row] = vnew;
      }
      if (tid == 0) {
        if (a.H_old) a.H_old[(int64_t)t * a.C + c] = H0u;
        if (a.H_new) a.H_new[(int64_t)t * a.C + c] = H1u;
        if (a.accept) a.accept[(int64_t)t * a.C + c] = acc ? 1 : 0;
      }
      if (!acc) ++rejected;
    }
    if (rok && !setV) a.cur[c * D + row] = scur;
    if (tid == 0) a.reject_count[c] += rejected;
  }
}

// ---- TWO chains per workgroup (513 ... 1024 chains: two workgroups per CU under the 256-register cap; 257 ... 512 chains: one) ----
// All four columns of the matrix instruction are in use (chain 0's U and V, chain 1's U and V), so there are no idle columns to
// defer into: the schedule is rmhmc_uv_kernel<2>'s (K refinement phases per pair of half steps + one phase after the rotation,
// any K).  What is taken over from the one-chain kernel above is the element-wise layout: the two parities of a row block hold
// the same four rows of a column; here lane (column cl, parity kp) OWNS rows row0 + kp and row0 + 2 + kp - two values per lane
// instead of four duplicated ones, half the element-wise instructions, one 8-byte LDS store per published vector - and the
// branch-free half steps (the set that moves first is selected by a per-lane coefficient, not by divergent control flow).
template <bool CO>
__global__ __launch_bounds__(XNT, CO ? 2 : 1) void rmhmc_uvc2_kernel(FusedArgs<float> a) {
  typedef float T;
  typedef float bf2 __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* lds = reinterpret_cast<T*>(smem_raw);
  constexpr int MSZ = XNC * XLD;
  T* DV = lds; T* GV = DV + MSZ; T* EV = GV + MSZ; T* W0 = EV + MSZ; T* W1 = W0 + MSZ;
  T* WS = W1 + MSZ;                // half steps: refinement vectors [pair of the step][iteration parity]
  T* red = WS + 4 * MSZ;           // [XWV][XNC][4]
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, cl = l & 3, grp = l >> 4;
  const int kpar = (l >> 3) & 1, rb = 2 * grp + ((l >> 2) & 1);
  const bool khi = kpar != 0, setV = (cl & 1) != 0;
  const int cidx = cl >> 1;
  const int D = a.D;
  const int row0 = 32 * w + 4 * rb, arow = row0 + cl;
  const int rowa = row0 + kpar, rowb = rowa + 2;           // the two rows this lane owns (column cl)
  const bool roka = rowa < D, rokb = rowb < D, roka_p = (rowa ^ 1) < D, rokb_p = (rowb ^ 1) < D;
  T Sa[XKJ], Pa[XKJ];
#pragma unroll
  for (int j = 0; j < XKJ; ++j) {
    const int k = 2 * j + kpar;
    const bool ok = arow < D && k < D;
    Sa[j] = ok ? a.S[(int64_t)k * D + arow] : 0.f;
    Pa[j] = ok ? a.P[(int64_t)k * D + arow] : 0.f;
  }
  const T mu_a = roka ? a.mu[rowa] : 0.f, mu_b = rokb ? a.mu[rowb] : 0.f;
  const T sd_a = roka ? a.S[(int64_t)rowa * D + rowa] : 0.f, sd_b = rokb ? a.S[(int64_t)rowb * D + rowb] : 0.f;
  for (int e = tid; e < 9 * MSZ + XWV * XNC * 4; e += XNT) lds[e] = 0.f;
  const T eh = 0.5f * a.eps;
  const T ehU = setV ? 0.f : eh, ehV = setV ? eh : 0.f;
  const T hc = 0.5f, rc = a.rot_c, rs = a.rot_s;
  const int own_off = cl * XLD + kpar * XHL + (row0 >> 1);  // rows rowa, rowb: two consecutive floats of this parity's half
  const int b_off = cl * XLD + kpar * XHL + 4 * grp;
  uint64_t chain = 0;
  bool live = false;

  auto partner = [&](T v) { return quad_dpp<0xB1>(v); };
  auto of_set_u = [&](T v) { return quad_dpp<0xA0>(v); };
  auto of_set_v = [&](T v) { return quad_dpp<0xF5>(v); };
  auto put2 = [&](T* X, T va, T vb) { *reinterpret_cast<bf2*>(X + own_off) = bf2{va, vb}; };
  // rows rowa, rowb of this lane's column: the parity sums
  auto extract = [&](const bf4& acc, T& va, T& vb) {
    const T m01 = khi ? acc[1] : acc[0], o01 = khi ? acc[0] : acc[1];
    const T m23 = khi ? acc[3] : acc[2], o23 = khi ? acc[2] : acc[3];
    va = m01 + other_parity(o01);
    vb = m23 + other_parity(o23);
  };
  auto fetch = [&](const T* X, bf4 (&c)[XSQ]) {
#pragma unroll
    for (int Q = 0; Q < XSQ; ++Q) c[Q] = *reinterpret_cast<const bf4*>(X + b_off + 16 * Q);
  };
  auto prod2 = [&](const T (&A1)[XKJ], const T* X1, const T (&A2)[XKJ], const T* X2, bf4& acc1, bf4& acc2) {
    bf4 c1[XSQ], c2[XSQ];
    fetch(X1, c1);
    fetch(X2, c2);
    __builtin_amdgcn_sched_barrier(0);
    static_for(std::make_integer_sequence<int, XQ>{}, [&](auto qc) {
      constexpr int q = decltype(qc)::value;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc1 = mfma_from_group<q % 4>(A1[4 * q + u], c1[q / 4][u], acc1);
        acc2 = mfma_from_group<q % 4>(A2[4 * q + u], c2[q / 4][u], acc2);
      }
    });
  };
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
  };
  auto jitter_one = [&](uint32_t n, uint32_t sub, T& ea, T& eb) {
    const U4 r = philox_block(a.seed, chain, n, PURPOSE_JITTER, sub, (uint32_t)(row0 >> 2));
    ea = (live && roka) ? a.jitter * u23<T>(khi ? r.y : r.x) : 0.f;
    eb = (live && rokb) ? a.jitter * u23<T>(khi ? r.w : r.z) : 0.f;
  };
  // both jitter vectors of a step for this column: the even parity draws the A pair's block, the odd parity the B pair's, each
  // hands the other the elements of the other's rows
  auto jitter_pair = [&](uint32_t n, uint32_t subA, uint32_t subB, T (&eA)[2], T (&eB)[2]) {
    const U4 r = philox_block(a.seed, chain, n, PURPOSE_JITTER, khi ? subB : subA, (uint32_t)(row0 >> 2));
    const T ma = (live && roka) ? a.jitter * u23<T>(khi ? r.y : r.x) : 0.f;
    const T mb = (live && rokb) ? a.jitter * u23<T>(khi ? r.w : r.z) : 0.f;
    const T oa = other_parity((live && roka_p) ? a.jitter * u23<T>(khi ? r.x : r.y) : 0.f);
    const T ob = other_parity((live && rokb_p) ? a.jitter * u23<T>(khi ? r.z : r.w) : 0.f);
    eA[0] = khi ? oa : ma; eA[1] = khi ? ob : mb;
    eB[0] = khi ? ma : oa; eB[1] = khi ? mb : ob;
  };
  // x = (P + diag(e))^-1 g from x0 = S g: K phases; w returns e . x_(K-1)
  auto solve = [&](T* WB, const T (&e)[2], const T (&x0)[2], T (&x)[2], T (&wv)[2]) {
    x[0] = x0[0]; x[1] = x0[1]; wv[0] = 0.f; wv[1] = 0.f;
    for (int it = 0; it < a.K; ++it) {
      T* A = WB + (it & 1) * MSZ;
      wv[0] = e[0] * x[0]; wv[1] = e[1] * x[1];
      put2(A, wv[0], wv[1]);
      __syncthreads();
      bf4 r = {0.f, 0.f, 0.f, 0.f};
      prod1(Sa, A, false, r);
      T ra, rbv;
      extract(r, ra, rbv);
      x[0] = x0[0] - ra; x[1] = x0[1] - rbv;
    }
  };
  auto col_sums = [&](T (&v)[3]) {
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      v[e] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[e]), 0x124 /* row_ror:4 */, 0xf, 0xf, false));
      v[e] += other_parity(v[e]);
      v[e] += __shfl_xor(v[e], 16, 64);
      v[e] += __shfl_xor(v[e], 32, 64);
    }
    __syncthreads();
    if (l < 4) {
#pragma unroll
      for (int e = 0; e < 3; ++e) red[(w * XNC + l) * 4 + e] = v[e];
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
  auto hamiltonian = [&](uint32_t n, uint32_t sub, const T (&X)[2], const T (&g)[2], T& H, T& logp, T (&Pd_out)[2], T (&Sg_out)[2],
                         T& kin_out, T& ld_out) {
    T ev[2] = {0.f, 0.f};
    if (a.has_jitter) jitter_one(n, sub, ev[0], ev[1]);
    const T dra = X[0] - mu_a, drb = X[1] - mu_b;
    put2(EV, ev[0], ev[1]);
    put2(DV, dra, drb);
    put2(GV, g[0], g[1]);
    __syncthreads();
    bf4 Pdv = {0.f, 0.f, 0.f, 0.f}, x0v = {0.f, 0.f, 0.f, 0.f}, s2v = {0.f, 0.f, 0.f, 0.f};
    prod2(Pa, DV, Sa, GV, Pdv, x0v);
    T x0[2], s2a = 0.f, s2b = 0.f;
    extract(Pdv, Pd_out[0], Pd_out[1]);
    extract(x0v, x0[0], x0[1]);
    if (a.has_jitter) { prod1(Sa, EV, true, s2v); extract(s2v, s2a, s2b); }
    Sg_out[0] = x0[0]; Sg_out[1] = x0[1];
    T v[3];
    v[0] = dra * Pd_out[0] + drb * Pd_out[1];
    v[2] = a.has_jitter ? ev[0] * (sd_a - 0.5f * s2a) + ev[1] * (sd_b - 0.5f * s2b) : 0.f;
    T xr[2], wv[2];
    solve(W0, ev, x0, xr, wv);
    v[1] = g[0] * xr[0] + g[1] * xr[1];
    col_sums(v);
    const float pi_term = (float)D * 1.8378770351409912f;   // S:712
    logp = a.log_norm - 0.5f * v[0];
    H = -logp + 0.5f * pi_term + 0.5f * (a.logdetP + v[2]) + 0.5f * v[1];
    kin_out = v[1]; ld_out = v[2];
  };

  const int64_t ngroup = (a.C + 1) / 2;
  for (int64_t cg = blockIdx.x; cg < ngroup; cg += gridDim.x) {
    const int64_t c = 2 * cg + cidx;
    live = c < a.C;
    chain = a.chain_offset + (uint64_t)(live ? c : 0);
    const int64_t cs = live ? c : 0;                        // (a dead column reads chain 0's rows and discards them)
    T scur[2];
    scur[0] = (live && roka) ? a.cur[cs * D + rowa] : 0.f;
    scur[1] = (live && rokb) ? a.cur[cs * D + rowb] : 0.f;
    int ma = -(int)roka, mb = -(int)rokb;
    asm volatile("" : "+v"(ma));
    asm volatile("" : "+v"(mb));
    auto momentum_raw = [&](int tt, T (&v)[2]) {
      const T* prow = a.p_ws + ((int64_t)tt * a.C + cs) * D;
      v[0] = prow[rowa & ma]; v[1] = prow[rowb & mb];
    };
    auto momentum_use = [&](T (&v)[2], T (&out)[2]) {
      asm volatile("" : "+v"(v[0])); asm volatile("" : "+v"(v[1]));
      out[0] = (live && roka) ? v[0] : 0.f; out[1] = (live && rokb) ? v[1] : 0.f;
    };
    int32_t rejected = 0;
    __syncthreads();
    bool have_next = false;
    T gn[2] = {0.f, 0.f}, y_next[2] = {0.f, 0.f}, z_next[2] = {0.f, 0.f}, H0_next = 0.f, lp_next = 0.f;
    for (int t = 0; t < a.n_traj; ++t) {
      const uint32_t n = (uint32_t)(a.traj_offset + t);
      T H0, H1, lp0, lp1, kin, ld, X[2], g[2], y[2], z[2];
      if (have_next) {
#pragma unroll
        for (int i = 0; i < 2; ++i) { g[i] = gn[i]; X[i] = scur[i]; y[i] = y_next[i]; z[i] = z_next[i]; }
        H0 = H0_next; lp0 = lp_next;
      } else {
        T raw[2];
        momentum_raw(t, raw);
        momentum_use(raw, g);
        X[0] = scur[0]; X[1] = scur[1];
        hamiltonian(n, 1, X, g, H0, lp0, y, z, kin, ld);    // S:971 -> S:822
      }
      const bool pre = t + 1 < a.n_traj;
      T gn_raw[2] = {0.f, 0.f};
      if (pre) momentum_raw(t + 1, gn_raw);
      const T y_start[2] = {y[0], y[1]};
      // one pair of half steps: the set with the non-zero `pre` coefficient moves its momentum first, the other one after its solve
      auto half_pair = [&](T cpre, T cpost, const T (&e)[2], T* WB) {
        g[0] = fmaf(-cpre, y[0], g[0]); g[1] = fmaf(-cpre, y[1], g[1]);
        z[0] = fmaf(-cpre, X[0] - mu_a, z[0]); z[1] = fmaf(-cpre, X[1] - mu_b, z[1]);
        T x[2], wv[2];
        solve(WB, e, z, x, wv);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          X[i] = fmaf(eh, x[i], X[i]);
          y[i] = fmaf(eh, g[i] - wv[i], y[i]);              // P x = g - e . x_(K-1)
        }
        g[0] = fmaf(-cpost, y[0], g[0]); g[1] = fmaf(-cpost, y[1], g[1]);
        z[0] = fmaf(-cpost, X[0] - mu_a, z[0]); z[1] = fmaf(-cpost, X[1] - mu_b, z[1]);
      };
      for (int lstep = 0; lstep < a.L; ++lstep) {           // S:427-461
        const uint32_t k0 = 2u + 8u * (uint32_t)lstep;
        T eA[2] = {0.f, 0.f}, eB[2] = {0.f, 0.f};
        if (a.has_jitter) jitter_pair(n, setV ? k0 + 1u : k0 + 2u, setV ? k0 + 7u : k0 + 4u, eA, eB);
        half_pair(ehU, ehV, eA, WS);                        // phi_A/2, phi_B/2  S:429-433
        if (a.K == 0) __syncthreads();                      // (no solve phase since the last reads of DV / GV)
#pragma unroll
        for (int i = 0; i < 2; ++i) {                       // phi_C  S:447-450, sequential (Q1), both sets compute it
          const T pX = partner(X[i]), pg = partner(g[i]);
          T xx = setV ? pX : X[i], b = setV ? pg : g[i], xc = setV ? X[i] : pX, bc = setV ? g[i] : pg;
          xx = hc * ((xx + xc) + rc * (xx - xc) + rs * (b - bc));
          b = hc * ((b + bc) - rs * (xx - xc) + rc * (b - bc));
          xc = hc * ((xx + xc) - rc * (xx - xc) - rs * (b - bc));
          bc = hc * ((b + bc) + rs * (xx - xc) - rc * (b - bc));
          X[i] = setV ? xc : xx; g[i] = setV ? bc : b;
        }
        put2(DV, X[0] - mu_a, X[1] - mu_b);                 // the tracked products of the rotated state, afresh
        put2(GV, g[0], g[1]);
        __syncthreads();
        bf4 p1 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
        prod2(Pa, DV, Sa, GV, p1, s1);
        extract(p1, y[0], y[1]);
        extract(s1, z[0], z[1]);
        half_pair(ehV, ehU, eB, WS + 2 * MSZ);              // phi_B/2, phi_A/2  S:454-458
      }
      if (a.K == 0) __syncthreads();
      // ---- H_new on the un-augmented pair = set U (S:989, Q4); in the V column: the next trajectory's momentum terms
      T Pd1[2], Sg1[2], Xh[2], gh[2];
      if (pre) momentum_use(gn_raw, gn);
      const bool nextcol = setV && pre;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        Xh[i] = of_set_u(X[i]);
        const T gu = of_set_u(g[i]);
        gh[i] = nextcol ? gn[i] : gu;
      }
      hamiltonian(nextcol ? n + 1u : n, nextcol ? 1u : 2u + 8u * (uint32_t)a.L, Xh, gh, H1, lp1, Pd1, Sg1, kin, ld);
      // ---- Metropolis test + bookkeeping (S:1000-1026, S:1045-1057) on the U column's values, mirrored in the V column
      const T H0u = of_set_u(H0), H1u = of_set_u(H1), lp1u = of_set_u(lp1);
      const T u = u23<T>(philox_block(a.seed, chain, n, PURPOSE_MH, 0, 0).x);
      const bool acc = mh_accept<T>(H0u, H1u, lp1u, u);
      const bool reset = (!acc) && ((int)n == a.burn + 1);  // Q2
      have_next = pre && !reset;
      {
        const T lp0u = of_set_u(lp0), ldv = of_set_v(ld), kinv = of_set_v(kin);
        const T Pa0 = of_set_u(Pd1[0]), Pa1 = of_set_u(Pd1[1]), Sv0 = of_set_v(Sg1[0]), Sv1 = of_set_v(Sg1[1]);
        if (have_next) {
          const float pi_term = (float)D * 1.8378770351409912f;
          lp_next = acc ? lp1u : lp0u;
          H0_next = -lp_next + 0.5f * pi_term + 0.5f * (a.logdetP + ldv) + 0.5f * kinv;
          y_next[0] = acc ? Pa0 : y_start[0]; y_next[1] = acc ? Pa1 : y_start[1];
          z_next[0] = Sv0; z_next[1] = Sv1;
        }
      }
      if (live) {
        if (roka) {
          const T vnew = acc ? Xh[0] : (reset ? a.theta_init[c * D + rowa] : scur[0]);
          scur[0] = vnew;
          if (!setV && a.samples && (int)n > a.burn) a.samples[((int64_t)((int)n - a.burn) * a.C + c) * D + rowa] = vnew;
        }
        if (rokb) {
          const T vnew = acc ? Xh[1] : (reset ? a.theta_init[c * D + rowb] : scur[1]);
          scur[1] = vnew;
          if (!setV && a.samples && (int)n > a.burn) a.samples[((int64_t)((int)n - a.burn) * a.C + c) * D + rowb] = vnew;
        }
      }
      if (live && !setV && w == 0 && (l >> 2) == 0) {
        if (a.H_old) a.H_old[(int64_t)t * a.C + c] = H0u;
        if (a.H_new) a.H_new[(int64_t)t * a.C + c] = H1u;
        if (a.accept) a.accept[(int64_t)t * a.C + c] = acc ? 1 : 0;
      }
      if (!acc) ++rejected;
    }
    if (live && !setV) {
      if (roka) a.cur[c * D + rowa] = scur[0];
      if (rokb) a.cur[c * D + rowb] = scur[1];
      if (w == 0 && (l >> 2) == 0) a.reject_count[c] += rejected;
    }
  }
}

int g_rmhmc_uvc = 1;         // tuning key "rmhmc_uvc" (default 1): one-chain groups with K == 2 and jitter run on rmhmc_uvc_kernel, two-chain groups on rmhmc_uvc2_kernel; 0 = rmhmc_uv_kernel

int rmhmc_uvc2_launch(const FusedArgs<float>& a, bool co, hipStream_t s) {
  const size_t bytes = (size_t)(9 * XNC * XLD + XWV * XNC * 4) * sizeof(float);
  const int64_t ngroup = (a.C + 1) / 2;
  const int grid = (int)(ngroup < 8192 ? ngroup : 8192);
  note_route("rmhmc_uvc2_kernel<%s>", co ? "co" : "solo");
  if (co) rmhmc_uvc2_kernel<true><<<grid, XNT, bytes, s>>>(a);
  else rmhmc_uvc2_kernel<false><<<grid, XNT, bytes, s>>>(a);
  return HTA_OK;
}

int rmhmc_uvc_launch(const FusedArgs<float>& a, bool co, hipStream_t s) {
  const size_t bytes = (size_t)(CBUF * XNC * XLD + XWV * 2 * 4) * sizeof(float);
  const int grid = (int)(a.C < 8192 ? a.C : 8192);
  note_route("rmhmc_uvc_kernel<%s>", co ? "co" : "solo");
  if (co) rmhmc_uvc_kernel<true><<<grid, XNT, bytes, s>>>(a);
  else rmhmc_uvc_kernel<false><<<grid, XNT, bytes, s>>>(a);
  return HTA_OK;
}

}  // namespace hta
