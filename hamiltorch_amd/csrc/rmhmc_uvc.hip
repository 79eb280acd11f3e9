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
  // x = (P + diag(e))^-1 g from x0 = S g by K refinement phases (the Hamiltonians: K as the host chose it)
  auto solve = [&](T e, T x0) -> T {
    T x = x0;
    for (int it = 0; it < a.K; ++it) {
      T* A = (it & 1) ? W1 : W0;
      A[own_off] = e * x;
      __syncthreads();
      bf4 r = {0.f, 0.f, 0.f, 0.f};
      prod1(Sa, A, false, r);
      x = x0 - extract1(r);
    }
    return x;
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
    const T xr = solve(ev, x0);
    v[1] = g * xr;
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
      // ---- flush: c2 of the last B pair
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

int g_rmhmc_uvc = 0;         // tuning key "rmhmc_uvc": 1 = one-chain groups with K == 2 and jitter run on rmhmc_uvc_kernel

int rmhmc_uvc_launch(const FusedArgs<float>& a, bool co, hipStream_t s) {
  const size_t bytes = (size_t)(CBUF * XNC * XLD + XWV * 2 * 4) * sizeof(float);
  const int grid = (int)(a.C < 8192 ? a.C : 8192);
  note_route("rmhmc_uvc_kernel<%s>", co ? "co" : "solo");
  if (co) rmhmc_uvc_kernel<true><<<grid, XNT, bytes, s>>>(a);
  else rmhmc_uvc_kernel<false><<<grid, XNT, bytes, s>>>(a);
  return HTA_OK;
}

}  // namespace hta
