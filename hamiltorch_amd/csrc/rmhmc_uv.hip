// Explicit RMHMC for a Gaussian target on the identity-soft-abs path (see rmhmc_fused.hip for why that path exists and what
// it computes: hamiltorch/samplers.py:969-1026 with the integrator S:425-461), ONE or TWO chains per workgroup on the matrix
// cores: the kernel for BASELINE config 3 (256 chains) and for every chain count up to 2 x (compute units).
//
// The tracked schedule of rmhmc_mfma4x4_kernel (a half step keeps only its K refinement products; the two half steps of a pair
// solve side by side) has, per chain, exactly two vectors in flight in every product phase: one of the state set
// U = (theta, p, y = P (theta - mu), z = S p) and one of the copy V = (theta_c, p_c, y_c, z_c).  Here those two vectors are two
// COLUMNS of the 16-block instruction v_mfma_f32_4x4x1_16b_f32 (N = 4: chain 0's U and V, chain 1's U and V), so a phase is ONE
// product of 52 instructions per wave for both solves of both chains, and lane (row block, column) keeps only ITS set's four
// rows - the element-wise work of a pair of half steps splits over the two columns:
//   first  (the set whose momentum moves first: U in S:429-433, V in S:454-458):  g -= eh y;  z -= eh (X - mu);  solve;  X += eh x;  y += eh (g - w)
//   second:                                                                         solve;  X += eh x;  y += eh (g - w);  g -= eh y;  z -= eh (X - mu)
// with x = (P + E)^-1 g from x_0 = z and w = e . x_(K-1).  The rotation phi_C (S:447-450) needs both sets in one lane: the
// columns exchange their X and g through a DPP quad permutation, both compute the (sequential, Q1) rotation and keep their half;
// right after it y and z are evaluated afresh (P (X - mu) and S g: one phase, two accumulator chains).  Rows x contraction
// parity inside a wave, operand fetch through the B-broadcast modifier, LDS layout: as rmhmc_mfma4x4_kernel
// (rmhmc_fused_dev.hpp).  Per step at K = 2: 5 phases, 312 matrix instructions per wave (the one-chain VALU kernel: 8 phases,
// 896 v_fmac_f32_dpp per lane at about 9 clocks each).  Same Philox streams, same update order, same bookkeeping (Q1, Q2, Q4).
#include <math.h>
#include "rmhmc_fused_dev.hpp"

// developer ablation builds (tools/uv_ablate.sh: wrong results, honest timing): 1 = no Philox (constant jitter), 2 = 4 instead
// of 52 matrix instructions per product, 4 = no workgroup barriers inside the step loop, 8 = operands not fetched from LDS,
// 16 = vectors not published to LDS
#ifndef HTA_UV_ABLATE
#define HTA_UV_ABLATE 0
#endif

namespace hta {

constexpr int UBUF = 9;            // LDS vector matrices: DV GV EV W0 W1 + 4 solve buffers [pair][iteration parity]

// LEAN (tuning key "rmhmc_lean", default 1; measured: profiles/r03y_ab_lines.txt - 256 chains 8.12e7 -> 8.25e7, 1024 chains
// (rmhmc_mfma4x4_kernel) 1.847e8 -> 1.870e8 explicit steps/s): the same arithmetic with fewer instructions around it - these kernels run one
// wave per SIMD: nothing hides an instruction of the element-wise / bookkeeping stretches between their matrix phases
// ("what comes next").  (i) put4 without its lane predicate: the upper lane half holds bit-identical duplicates of
// the lower one's values ("duplicate state, one writer"), so both halves store - same address, same value - and the
// s_and_saveexec / s_cbranch_execz / s_or_b64 around every store go; (ii) the padding rows (>= D) of every state vector are
// exact zeros by construction (zero matrix rows, zero jitter, zero mu_r), so X - mu needs no select there.  A chain that went
// non-finite may carry NaN into its own padding rows; it is rejected, and every vector is rewritten from the clean current
// point before the next trajectory reads it.
// CO (round 4, tuning key "rmhmc_uv_co"): the same code under a 256-register cap, so that TWO workgroups share a CU (one wave of
// each per SIMD): a phase is a latency chain (LDS publish - barrier - operand fetch - 52 dependent matrix instructions - combine)
// that leaves the matrix pipe idle more than half of the time; a second, independent workgroup fills those gaps.  The capped
// build keeps S and P in VGPRs (no accumulation registers at all: the matrix instructions write VGPRs directly) and spills
// nothing inside the trajectory (tests/test_kernel_resources.py).  NACC ("rmhmc_uv_acc"): accumulator chains per product - with
// two, the compiler pads every dependent pair of v_mfma_f32_4x4x1 with an s_nop (22 per phase); four need none.  The sums of a
// row are then taken in another order: not bit-identical to NACC = 2, the same to rounding (tests/test_gpu_rmhmc.py).
template <int G, bool LEAN = false, bool CO = false, int NACC = 2>                   // chains per workgroup
__global__ __launch_bounds__(XNT, CO ? 2 : 1) void rmhmc_uv_kernel(FusedArgs<float> a) {
  typedef float T;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* lds = reinterpret_cast<T*>(smem_raw);
  constexpr int MSZ = XNC * XLD;
  T* DV = lds;                     // X - mu of every column (operand of P: refresh phase, Hamiltonians)
  T* GV = DV + MSZ;                // momenta of every column (operand of S there)
  T* EV = GV + MSZ;                // Hamiltonian: the jitter
  T* W0 = EV + MSZ; T* W1 = W0 + MSZ;   // Hamiltonian: refinement vectors
  T* WS = W1 + MSZ;                // half steps: refinement vectors [pair of the step][iteration parity]
  T* red = WS + 4 * MSZ;           // [XWV][XNC][4]
  // lane bits: [1:0] column = 2 chain + set, [2] low bit of the row block, [3] contraction parity, [5:4] 16-lane group
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, cl = l & 3, grp = l >> 4;
  const int kpar = (l >> 3) & 1, rb = 2 * grp + ((l >> 2) & 1);
  const bool upper = kpar != 0, lead = (l >> 2) == 0;
  const int cidx = cl >> 1;
  const bool setV = (cl & 1) != 0;
  const int D = a.D;
  const int row0 = 32 * w + 4 * rb, arow = row0 + cl;     // this lane OWNS rows row0..row0+3 of its column and SUPPLIES matrix row arow
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
  for (int e = tid; e < UBUF * MSZ + XWV * XNC * 4; e += XNT) lds[e] = 0.f;
  const T eh = 0.5f * a.eps;
  const int own_off = cl * XLD + (row0 >> 1);             // rows row0, row0+2 -> even half; row0+1, row0+3 -> odd half
  const int b_off = cl * XLD + kpar * XHL + 4 * grp;      // this group's chunk of a super-chunk of four
  uint64_t chain = 0;
  bool live = false;

  typedef float bf2 __attribute__((ext_vector_type(2)));
  auto put4 = [&](T* X, const T (&v)[4]) {
    if (HTA_UV_ABLATE & 16) return;
    if (LEAN || !upper) {
      *reinterpret_cast<bf2*>(X + own_off) = bf2{v[0], v[2]};
      *reinterpret_cast<bf2*>(X + own_off + XHL) = bf2{v[1], v[3]};
    }
  };
  auto centred = [&](const T (&X)[4], T (&d)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) d[e] = (LEAN || rok[e]) ? X[e] - mu_r[e] : 0.f;
  };
  // the jitter of this lane's four rows for one sub-stream (uniform_elem layout: rows 4b..4b+3 are Philox block b)
  auto jitter_raw = [&](uint32_t n, uint32_t sub, T (&out)[4]) {
    const U4 r = (HTA_UV_ABLATE & 1) ? U4{n * 2654435761u + sub, sub * 40503u + n, n ^ (sub << 16), (uint32_t)tid * 77u + sub}
                                     : philox_block(a.seed, chain, n, PURPOSE_JITTER, sub, (uint32_t)(row0 >> 2));
    const T u[4] = {u23<T>(r.x), u23<T>(r.y), u23<T>(r.z), u23<T>(r.w)};
#pragma unroll
    for (int e = 0; e < 4; ++e) out[e] = (live && rok[e]) ? a.jitter * u[e] : 0.f;
  };
  auto partner = [&](T v) {                                  // the other set's value: lane l ^ 1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, false));
  };
  auto of_set_u = [&](T v) {                                 // the U column's value of this chain: lane l & ~1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xA0 /* quad_perm:[0,0,2,2] */, 0xf, 0xf, false));
  };
  auto fetch = [&](const T* X, bf4 (&c)[XSQ]) {
#pragma unroll
    for (int Q = 0; Q < XSQ; ++Q) {
      if (HTA_UV_ABLATE & 8) { const float f = (float)(Q + 1) * 1e-3f * mu_r[Q & 3]; c[Q] = bf4{f, f + 1.f, f, f - 1.f}; }
      else c[Q] = *reinterpret_cast<const bf4*>(X + b_off + 16 * Q);
    }
  };
  auto both = [&](bf4& acc) {                                // lanes l and l ^ 8 both end with (even k) + (odd k)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] += other_parity(acc[e]);
  };
  // acc1 = A1 X1, acc2 = A2 X2 for all four columns (two independent accumulator chains)
  auto prod2 = [&](const T (&A1)[XKJ], const T* X1, const T (&A2)[XKJ], const T* X2, bf4& acc1, bf4& acc2) {
    bf4 c1[XSQ], c2[XSQ];
    fetch(X1, c1);
    fetch(X2, c2);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NACC == 4) {
      bf4 b1 = {0.f, 0.f, 0.f, 0.f}, b2 = {0.f, 0.f, 0.f, 0.f};
      static_for(std::make_integer_sequence<int, XQ>{}, [&](auto qc) {
        constexpr int q = decltype(qc)::value;
      if constexpr ((HTA_UV_ABLATE & 2) && q >= 1) return;
#pragma unroll
        for (int u = 0; u < 4; u += 2) {
          acc1 = mfma_from_group<q % 4>(A1[4 * q + u], c1[q / 4][u], acc1);
          acc2 = mfma_from_group<q % 4>(A2[4 * q + u], c2[q / 4][u], acc2);
          b1 = mfma_from_group<q % 4>(A1[4 * q + u + 1], c1[q / 4][u + 1], b1);
          b2 = mfma_from_group<q % 4>(A2[4 * q + u + 1], c2[q / 4][u + 1], b2);
        }
      });
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc1[e] += b1[e]; acc2[e] += b2[e]; }
    } else {
      static_for(std::make_integer_sequence<int, XQ>{}, [&](auto qc) {
        constexpr int q = decltype(qc)::value;
      if constexpr ((HTA_UV_ABLATE & 2) && q >= 1) return;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          acc1 = mfma_from_group<q % 4>(A1[4 * q + u], c1[q / 4][u], acc1);
          acc2 = mfma_from_group<q % 4>(A2[4 * q + u], c2[q / 4][u], acc2);
        }
      });
    }
    both(acc1); both(acc2);
  };
  // one product on two accumulator chains (k in the order 0 2 | 1 3 of every chunk)
  auto prod1 = [&](const T (&A1)[XKJ], const T* X1, bool squared, bf4& acc) {
    bf4 c1[XSQ], sb = {0.f, 0.f, 0.f, 0.f};
    fetch(X1, c1);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NACC == 4) {
      bf4 sc = {0.f, 0.f, 0.f, 0.f}, sd = {0.f, 0.f, 0.f, 0.f};
      static_for(std::make_integer_sequence<int, XQ>{}, [&](auto qc) {
        constexpr int q = decltype(qc)::value;
      if constexpr ((HTA_UV_ABLATE & 2) && q >= 1) return;
        const T a0 = A1[4 * q], a1 = A1[4 * q + 1], a2 = A1[4 * q + 2], a3 = A1[4 * q + 3];
        acc = mfma_from_group<q % 4>(squared ? a0 * a0 : a0, c1[q / 4][0], acc);
        sb = mfma_from_group<q % 4>(squared ? a1 * a1 : a1, c1[q / 4][1], sb);
        sc = mfma_from_group<q % 4>(squared ? a2 * a2 : a2, c1[q / 4][2], sc);
        sd = mfma_from_group<q % 4>(squared ? a3 * a3 : a3, c1[q / 4][3], sd);
      });
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = (acc[e] + sc[e]) + (sb[e] + sd[e]);
    } else {
      static_for(std::make_integer_sequence<int, XQ>{}, [&](auto qc) {
        constexpr int q = decltype(qc)::value;
      if constexpr ((HTA_UV_ABLATE & 2) && q >= 1) return;
#pragma unroll
        for (int u = 0; u < 4; u += 2) {
          const T a0 = A1[4 * q + u], a1 = A1[4 * q + u + 1];
          acc = mfma_from_group<q % 4>(squared ? a0 * a0 : a0, c1[q / 4][u], acc);
          sb = mfma_from_group<q % 4>(squared ? a1 * a1 : a1, c1[q / 4][u + 1], sb);
        }
      });
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += sb[e];
    }
    both(acc);
  };
  // x = (P + diag(e))^-1 g from x0 = S g: K phases (one barrier, one product each); w returns e . x_(K-1) (zero without jitter).
  // WB: two vector matrices, written before the barrier of a phase and read in that phase only.
  auto solve = [&](T* WB, const T (&e)[4], const T (&x0)[4], T (&x)[4], T (&wv)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { x[i] = x0[i]; wv[i] = 0.f; }
    for (int it = 0; it < a.K; ++it) {
      T* A = WB + (it & 1) * MSZ;
#pragma unroll
      for (int i = 0; i < 4; ++i) wv[i] = e[i] * x[i];
      put4(A, wv);
      if (HTA_UV_ABLATE & 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else __syncthreads();
      bf4 r = {0.f, 0.f, 0.f, 0.f};
      prod1(Sa, A, false, r);
#pragma unroll
      for (int i = 0; i < 4; ++i) x[i] = x0[i] - r[i];
    }
  };
  // three sums per column over the rows, complete in every lane of the column (the odd parity holds duplicates: it adds nothing)
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
  // H = -log p + D/2 log 2 pi + 1/2 log|G| + 1/2 g^T G^-1 g  (S:731) of this column's (X, g); also returns P (X - mu) and S g
  // (n, sub: per lane - in the Hamiltonian at a trajectory's end the V column evaluates the NEXT trajectory's momentum terms)
  auto hamiltonian = [&](uint32_t n, uint32_t sub, const T (&X)[4], const T (&g)[4], T& H, T& logp, T (&Pd_out)[4], T (&Sg_out)[4],
                         T& kin_out, T& ld_out) {
    T ev[4] = {0.f, 0.f, 0.f, 0.f}, dr[4];
    if (a.has_jitter) jitter_raw(n, sub, ev);
    centred(X, dr);
    put4(EV, ev);
    put4(DV, dr);
    put4(GV, g);
    __syncthreads();
    bf4 Pd = {0.f, 0.f, 0.f, 0.f}, x0v = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    prod2(Pa, DV, Sa, GV, Pd, x0v);
    if (a.has_jitter) prod1(Sa, EV, true, s2);              // second-order log-det term: (S . S) e
    T v[3] = {0.f, 0.f, 0.f}, x0[4], xr[4], wv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      x0[e] = x0v[e];
      Pd_out[e] = Pd[e]; Sg_out[e] = x0v[e];
      v[0] += dr[e] * Pd[e];
      if (a.has_jitter) v[2] += ev[e] * (sd_r[e] - 0.5f * s2[e]);       // log|P + E| = log|P| + tr(SE) - 1/2 tr((SE)^2) + ...
    }
    solve(W0, ev, x0, xr, wv);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[1] += g[e] * xr[e];
    block_sums(v);
    const float pi_term = (float)D * 1.8378770351409912f;   // S:712
    logp = a.log_norm - 0.5f * v[0];
    H = -logp + 0.5f * pi_term + 0.5f * (a.logdetP + v[2]) + 0.5f * v[1];
    kin_out = v[1]; ld_out = v[2];
  };
  auto of_set_v = [&](T v) {                                 // the V column's value of this chain: lane l | 1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xF5 /* quad_perm:[1,1,3,3] */, 0xf, 0xf, false));
  };

  const int64_t ngroup = (a.C + G - 1) / G;
  for (int64_t cg = blockIdx.x; cg < ngroup; cg += gridDim.x) {
    const int64_t c = G * cg + cidx;
    live = cidx < G && c < a.C;
    chain = a.chain_offset + (uint64_t)(live ? c : 0);
    T scur[4], X[4], g[4], y[4], z[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) scur[e] = (live && rok[e]) ? a.cur[c * D + row0 + e] : 0.f;
    int lmask = -(int)live, rmask[4];
    asm volatile("" : "+v"(lmask));
#pragma unroll
    for (int e = 0; e < 4; ++e) { rmask[e] = -(int)rok[e]; asm volatile("" : "+v"(rmask[e])); }
    // this lane's four rows of the pre-drawn momentum of local trajectory tt: four UNCONDITIONAL loads (a lane without a row
    // reads element 0 of the row block and discards it), so that they issue back to back and are waited for once
    auto momentum_row_raw = [&](int tt, T (&v)[4]) {
      // (offsets masked arithmetically with masks the optimiser cannot see through: a select on the index - or a mask it can
      //  trace back to `live` - comes back as a branch around each load, with a wait inside)
      const T* prow = a.p_ws + ((int64_t)tt * a.C + (int64_t)((int)c & lmask)) * D;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = prow[(row0 + e) & rmask[e] & lmask];
    };
    auto momentum_row_use = [&](T (&v)[4], T (&out)[4]) {     // (the empty asm pins the first use of the loaded registers HERE)
#pragma unroll
      for (int e = 0; e < 4; ++e) { asm volatile("" : "+v"(v[e])); out[e] = (live && rok[e]) ? v[e] : 0.f; }
    };
    auto momentum_row = [&](int tt, T (&out)[4]) { T v[4]; momentum_row_raw(tt, v); momentum_row_use(v, out); };
    int32_t rejected = 0;
    __syncthreads();                                        // the previous group's last reads of the vector matrices
    // H_old of trajectory t + 1 needs, besides -log p of the current point, only terms of the NEW momentum and jitter
    // (1/2 p^T G^-1 p and log |G|: the curvature is constant) - none of them waits for trajectory t's outcome.  So the
    // Hamiltonian at the END of trajectory t evaluates them on the side: the V column, idle there, takes p(t+1) and the jitter
    // of (t+1, sub-stream 1) through the same phases, and trajectory t + 1 starts without a Hamiltonian of its own
    // (log p, y = P (theta - mu) of the accepted / kept point are known; z = S p(t+1) comes out of the V column).  The same
    // products in the same order as the separate evaluation: bit-identical results, four phases less per trajectory.  Not
    // taken for a launch's first trajectory and after the Q2 reset to params_init (its log p is not known).
    bool have_next = false;
    T gn[4] = {0.f, 0.f, 0.f, 0.f}, y_next[4], z_next[4], H0_next = 0.f, lp_next = 0.f;
    for (int t = 0; t < a.n_traj; ++t) {
      const uint32_t n = (uint32_t)(a.traj_offset + t);
      // ---- gibbs: p = chol(G(theta)) z, drawn ahead by the momentum kernel (S:183-184); theta_c = theta, p_c = p (S:425-426)
      T H0, H1, lp0, lp1, kin, ld;
      if (have_next) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { g[e] = gn[e]; X[e] = scur[e]; y[e] = y_next[e]; z[e] = z_next[e]; }
        H0 = H0_next; lp0 = lp_next;
      } else {
        momentum_row(t, g);
#pragma unroll
        for (int e = 0; e < 4; ++e) X[e] = scur[e];
        hamiltonian(n, 1, X, g, H0, lp0, y, z, kin, ld);    // S:971 -> S:822
      }
      // the NEXT trajectory's momentum row is requested here, a whole trajectory before its first use (the V column of this
      // trajectory's last Hamiltonian): requested there, each of its four loads was followed by its own s_waitcnt vmcnt(0) -
      // four exposed memory round trips per trajectory of a wave that has nothing else to issue
      const bool pre = t + 1 < a.n_traj;
      T gn_raw[4] = {0.f, 0.f, 0.f, 0.f};
      if (pre) momentum_row_raw(t + 1, gn_raw);
      T y_start[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) y_start[e] = y[e];
      // one pair of half steps for this column; WB: the pair's two refinement matrices
      auto half_pair = [&](bool first, const T (&e)[4], T* WB) {
        if (first) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            g[i] -= eh * y[i];
            z[i] -= eh * ((LEAN || rok[i]) ? X[i] - mu_r[i] : 0.f);
          }
        }
        T x[4], wv[4];
        solve(WB, e, z, x, wv);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          X[i] += eh * x[i];
          y[i] += eh * (g[i] - wv[i]);                      // P x = g - e . x_(K-1)
        }
        if (!first) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            g[i] -= eh * y[i];
            z[i] -= eh * ((LEAN || rok[i]) ? X[i] - mu_r[i] : 0.f);
          }
        }
      };
      for (int lstep = 0; lstep < a.L; ++lstep) {           // S:427-461
        const uint32_t k0 = 2u + 8u * (uint32_t)lstep;
        // jitter: in S:429-433 the set that moves first (U) solves with sub-stream k0 + 2 and V with k0 + 1; in S:454-458 V moves
        // first (k0 + 7) and U second (k0 + 4).  The even parity draws this column's block of the first pair, the odd parity
        // that of the second pair, then they exchange: one Philox pass per step and lane (about 1000 clocks of a 9000-clock
        // step).  Measured and rejected (profiles/README.md, r02r - r02t): the pass's rounds issued one per chunk of eight matrix
        // instructions of the refresh phase - 1.5x SLOWER (3.08e7 against 4.71e7 steps/s at 256 chains: a wave issues in order,
        // a quarter-rate multiply holds back the matrix instructions behind it); one round in each window where a phase waits for
        // LDS (after the operand fetch, between the stores and the barrier) - 5 % slower; the kernel under a 256-register cap so
        // that two workgroups share a CU's SIMDs - with P's operands in LDS instead of registers (nothing in scratch inside the
        // step loop) 1024 chains as 512 two-chain workgroups reach 9.2e7 where the four-chain kernel does 1.15e8 in the same
        // run, and 512 chains lose 15 %: a second resident workgroup bought no overlap at all.
        T e1[4] = {0.f, 0.f, 0.f, 0.f}, e2[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.has_jitter) {
          const uint32_t sub1 = setV ? k0 + 1u : k0 + 2u, sub2 = setV ? k0 + 7u : k0 + 4u;
          T mine[4];
          jitter_raw(n, upper ? sub2 : sub1, mine);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const T oth = other_parity(mine[e]);
            e1[e] = upper ? oth : mine[e];
            e2[e] = upper ? mine[e] : oth;
          }
        }
        half_pair(!setV, e1, WS);                           // phi_A/2, phi_B/2  S:429-433
        if (a.K == 0) __syncthreads();                      // (no solve phase since the last reads of DV / GV)
        {                                                   // phi_C  S:447-450, sequential (Q1), both columns compute it
          T dv[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const T pX = partner(X[e]), pg = partner(g[e]);
            T xx = setV ? pX : X[e], b = setV ? pg : g[e], xc = setV ? X[e] : pX, bc = setV ? g[e] : pg;
            const T h = 0.5f, cc = a.rot_c, ss = a.rot_s;
            xx = h * ((xx + xc) + cc * (xx - xc) + ss * (b - bc));
            b = h * ((b + bc) - ss * (xx - xc) + cc * (b - bc));
            xc = h * ((xx + xc) - cc * (xx - xc) - ss * (b - bc));
            bc = h * ((b + bc) + ss * (xx - xc) - cc * (b - bc));
            X[e] = setV ? xc : xx; g[e] = setV ? bc : b;
          }
          centred(X, dv);                                   // the tracked products of the rotated state, afresh
          put4(DV, dv);
          put4(GV, g);
          if (HTA_UV_ABLATE & 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else __syncthreads();
          bf4 p1 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
          prod2(Pa, DV, Sa, GV, p1, s1);
#pragma unroll
          for (int e = 0; e < 4; ++e) { y[e] = p1[e]; z[e] = s1[e]; }
        }
        half_pair(setV, e2, WS + 2 * MSZ);                  // phi_B/2, phi_A/2  S:454-458
      }
      if (a.K == 0) __syncthreads();
      // ---- H_new on the un-augmented pair = set U (S:989, Q4); in the V column: the next trajectory's momentum terms
      T Pd1[4], Sg1[4], Xh[4], gh[4];
      if (pre) momentum_row_use(gn_raw, gn);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const T xu = of_set_u(X[e]);
        Xh[e] = xu;                                          // both columns at the proposal theta': the V column's P (theta' - mu) is not used
        gh[e] = (setV && pre) ? gn[e] : (setV ? of_set_u(g[e]) : g[e]);
      }
      const bool nextcol = setV && pre;
      hamiltonian(nextcol ? n + 1u : n, nextcol ? 1u : 2u + 8u * (uint32_t)a.L, Xh, gh, H1, lp1, Pd1, Sg1, kin, ld);
      // ---- Metropolis test + bookkeeping (S:1000-1026, S:1045-1057) on the U column's values, mirrored in the V column
      const T H0u = of_set_u(H0), H1u = of_set_u(H1), lp1u = of_set_u(lp1);
      const T u = u23<T>(philox_block(a.seed, chain, n, PURPOSE_MH, 0, 0).x);
      const bool acc = mh_accept<T>(H0u, H1u, lp1u, u);
      const bool reset = (!acc) && ((int)n == a.burn + 1);  // Q2
      have_next = pre && !reset;
      if (have_next) {                                       // H_old, y, z of trajectory t + 1 (the expression of hamiltonian(), same order)
        const float pi_term = (float)D * 1.8378770351409912f;
        lp_next = acc ? lp1u : of_set_u(lp0);
        H0_next = -lp_next + 0.5f * pi_term + 0.5f * (a.logdetP + of_set_v(ld)) + 0.5f * of_set_v(kin);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          y_next[e] = acc ? of_set_u(Pd1[e]) : y_start[e];
          z_next[e] = of_set_v(Sg1[e]);
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const T xu = of_set_u(X[e]);
        if (live && rok[e]) {
          const T vnew = acc ? xu : (reset ? a.theta_init[c * D + row0 + e] : scur[e]);
          scur[e] = vnew;
          if (!upper && !setV && a.samples && (int)n > a.burn) a.samples[((int64_t)((int)n - a.burn) * a.C + c) * D + row0 + e] = vnew;
        }
      }
      if (live && !setV && w == 0 && lead) {
        if (a.H_old) a.H_old[(int64_t)t * a.C + c] = H0u;
        if (a.H_new) a.H_new[(int64_t)t * a.C + c] = H1u;
        if (a.accept) a.accept[(int64_t)t * a.C + c] = acc ? 1 : 0;
      }
      if (!acc) ++rejected;
    }
    if (live && !setV) {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (rok[e] && !upper) a.cur[c * D + row0 + e] = scur[e];
      if (w == 0 && lead) a.reject_count[c] += rejected;
    }
  }
}

int g_rmhmc_uv_co = 1;       // tuning key "rmhmc_uv_co" (default 1): the 256-register instances, two workgroups per CU; 0 = the uncapped ones
int g_rmhmc_uv_acc = 2;      // tuning key "rmhmc_uv_acc": accumulator chains per product (2 or 4)
int g_rmhmc_uv_g = 0;        // tuning key "rmhmc_uv_g": chains per workgroup (0 = by chain count, 1, 2)

int rmhmc_uv_launch(const FusedArgs<float>& a, int cus, hipStream_t s) {
  const size_t bytes = (size_t)(UBUF * XNC * XLD + XWV * XNC * 4) * sizeof(float);
  const int g = g_rmhmc_uv_g == 1 || g_rmhmc_uv_g == 2 ? g_rmhmc_uv_g : (a.C <= cus ? 1 : 2);
  const int64_t ngroup = (a.C + g - 1) / g;
  const int grid = (int)(ngroup < 8192 ? ngroup : 8192);
  const bool co = g_rmhmc_uv_co != 0, acc4 = g_rmhmc_uv_acc == 4;
  if (g == 1 && g_rmhmc_uvc && g_rmhmc_lean && a.K == 2 && a.has_jitter) return rmhmc_uvc_launch(a, co, s);
  // (round 5, measured and not kept - tools/scratch/rmhmc_uvc2d.hip.rejected, profiles/r05b_uvc2d_ab.txt: the one-chain kernel's three
  //  phases per step for two-chain groups, the deferred second-order products as extra accumulator chains: 364 matrix instructions and 3
  //  barriers per step instead of 312 and 5, equal to rounding - 5 % SLOWER at 1024 chains (4.64 against 4.39 ms), slower at 384 ... 1536:
  //  with two workgroups per CU the phases' latency is already covered, the extra matrix instructions are not)
  if (g == 2 && g_rmhmc_uvc && g_rmhmc_lean) return rmhmc_uvc2_launch(a, co, s);
  if (!g_rmhmc_lean) {
    note_route("rmhmc_uv_kernel<%d>", g);
    if (g == 1) rmhmc_uv_kernel<1><<<grid, XNT, bytes, s>>>(a);
    else rmhmc_uv_kernel<2><<<grid, XNT, bytes, s>>>(a);
    return HTA_OK;
  }
  if (!co && !acc4) note_route("rmhmc_uv_kernel<%d,lean>", g);
  else note_route("rmhmc_uv_kernel<%d,lean,%s,%d>", g, co ? "co" : "solo", acc4 ? 4 : 2);
#define HTA_UV(GG, CO_, NA) rmhmc_uv_kernel<GG, true, CO_, NA><<<grid, XNT, bytes, s>>>(a)
  if (g == 1) {
    if (co) { if (acc4) HTA_UV(1, true, 4); else HTA_UV(1, true, 2); }
    else { if (acc4) HTA_UV(1, false, 4); else HTA_UV(1, false, 2); }
  } else {
    if (co) { if (acc4) HTA_UV(2, true, 4); else HTA_UV(2, true, 2); }
    else { if (acc4) HTA_UV(2, false, 4); else HTA_UV(2, false, 2); }
  }
#undef HTA_UV
  return HTA_OK;
}

}  // namespace hta
