// The fused (split-)HMC kernel of a one-hidden-layer Bayesian MLP on the matrix cores (instantiated by mlp_mfma.hip), written
// generically in the number of chains a workgroup carries (NC).  The product runs NC = 1.
//
// NC = 2 was round 5's experiment (tools/scratch/mlp_mfma2.hip.rejected instantiates it).  The idea: a gradient pass is a chain of
// dependent stages (14 matrix instructions -> activation -> quad reduce -> LDS -> barrier -> residual -> LDS -> barrier -> 28 matrix
// instructions -> cross-group sums -> kick) whose LATENCY is what the pass costs - one workgroup per CU takes 2.97 us per gradient,
// two co-resident workgroups 4.80 us for their two (profiles/r03l_cfg4_phase_ab.txt) - so two chains in the SAME waves, every stage
// issued for both back to back (shared X operands, shared barriers, 256 registers, 12 spills instead of 56), should overlap better
// than the hardware overlaps two workgroups.  Measured (profiles/r05s_mlp_pair_ab.txt): bit-identical per chain, and 10 % SLOWER
// at BASELINE config 4 (6.81 against 6.13 ms per 20 x 10 split steps at 512 chains; 1024 chains: 13.6 against 12.0 ms) - one
// in-order wave issues the two chains' instructions one after the other, two waves issue matrix and vector instructions in the same
// cycle.  What the rewrite did leave behind: the backward pass's four X operands loaded ahead of its four matrix instructions
// (NC = 1: 6.25 -> 6.13 ms, +2 %).
#pragma once
#include "mlp.hpp"
#include "philox.hpp"

#ifndef HTA_TIMING
#define HTA_TIMING 0   // developer cycle counters per phase of a gradient (wave 0 of block 0), read by tools/scratch/mlp_ablate.cpp
#endif
#if HTA_TIMING
extern __device__ unsigned long long hta_dbg[16];
#define HTA_TICK(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); tacc[k] += now_ - tlast; tlast = now_; } while (0)
#else
#define HTA_TICK(k) do {} while (0)
#endif

namespace hta {

typedef float V4f __attribute__((ext_vector_type(4)));

template <int CTRL> __device__ __forceinline__ float dpp_get(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
// DPP controls: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_ror:4, row_ror:8
#define HTA_DPP_X1 0xB1
#define HTA_DPP_X2 0x4E
#define HTA_DPP_ROR4 0x124
#define HTA_DPP_ROR8 0x128

// sum over the 4 lane groups (g); every lane gets the total
__device__ __forceinline__ float groups_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

template <int ACT> __device__ __forceinline__ float mact(float z) {
  // max(z, 0) in ONE VALU op, v_med3_f32(z, 0, FLT_MAX): fmaxf - and med3 against +inf, which the compiler folds into it -
  // compile to v_max(z, z) (a canonicalize) + v_max(0, .): 56 instead of 28 instructions per gradient at BASELINE config 4.
  // (An activation of +inf - a diverged chain - becomes FLT_MAX: the log-probability still overflows, the proposal is rejected.)
  if (ACT == 0) return __builtin_amdgcn_fmed3f(z, 0.0f, 3.4028234663852886e38f);
  if (ACT == 1) return tanhf(z);
  return 1.0f / (1.0f + expf(-z));
}
template <int ACT> __device__ __forceinline__ float mact_deriv(float h) {
  if (ACT == 0) return h > 0.0f ? 1.0f : 0.0f;
  if (ACT == 1) return 1.0f - h * h;
  return h * (1.0f - h);
}

template <int NK, int NPT, int ACT, int NC>
struct MfmaChain {
  static constexpr int INP = 4 * NK;          // padded input width of the LDS copy of X
  static constexpr int CP = 16 * NPT;         // points per chunk
  struct Rec { float w1[NK]; float b1, w2, b2; };
  const MlpArgs<float>& a;
  const float* Xs; const float* Ys; const float* ones; float* fpart; float* rbuf; float* red; float* dump;
  int tid, nthr, NU, t, c, g, kb;
  bool uvalid, wvalid[NK], ones_lane;
#if HTA_TIMING
  unsigned long long tacc[16] = {0}, tlast = 0;
#endif
  __device__ MfmaChain(const MlpArgs<float>& a_) : a(a_) {}

  // per chain: the sum over the workgroup (the same two barriers and the same left-to-right sum over the tiles as with one chain)
  __device__ __forceinline__ void block_sum(float (&v)[NC]) {
#pragma unroll
    for (int k = 0; k < NC; ++k) v[k] = wave_sum(v[k]);
    __syncthreads();
    if ((tid & 63) == 0) {
#pragma unroll
      for (int k = 0; k < NC; ++k) red[k * 16 + t] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      float s = 0;
      for (int i = 0; i < NU; ++i) s += red[k * 16 + i];
      v[k] = s;
    }
  }

  // Likelihood part of split points [lo, hi) for the workgroup's NC chains.  GRAD: its gradient into gr (prior added by the
  // caller); otherwise ret[k] = the sum of squared residuals (every lane).
  template <bool GRAD> __device__ __forceinline__ void pass(const Rec (&q)[NC], int lo, int hi, Rec (&gr)[NC], float (&ret)[NC]) {
    V4f gacc0[NC], gacc1[NC];
    float sdv[NC][4], gw2v[NC][4], ssev[NC][4], gb1v[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      gacc0[k] = V4f{0, 0, 0, 0}; gacc1[k] = V4f{0, 0, 0, 0}; gb1v[k] = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) { sdv[k][e] = 0; gw2v[k][e] = 0; ssev[k][e] = 0; }
    }
    const bool odd = c & 1, bit1 = c & 2;
    for (int c0 = lo; c0 < hi; c0 += CP) {
      const int cnt = min(CP, hi - c0);               // a ragged last chunk runs all NPT tiles; its extra points get delta = 0
      V4f h[NC][NPT];
      HTA_TICK(0);
      // ---- forward: pre-activations of this wave's 16 units at every point of the chunk, then the activations and this
      //      tile's share of f(x_p) = sum_u w2_u h[p, u]: a reduce-scatter over the quad (the 4 point registers end up one
      //      per lane), then rotations by 4 and 8 across the 16 unit lanes.  Straight-line code: tile pt + 1's MFMAs are in
      //      flight under tile pt's VALU work; the chains of a workgroup share the X operand of every instruction.
      const float* xr = Xs + (size_t)(c0 + c) * INP + NK * g;
      // lanes c >= 4 hold copies: they store to a per-lane dump slot instead of branching around the store
      float* fw = (c < 4) ? fpart + (size_t)t * CP + 4 * g + c : dump + (tid & 63);
      const int fstride = (c < 4) ? NU * CP : 0;      // chain k's partials: fpart[k][NU][CP]
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt) {
        V4f acc[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) acc[k] = V4f{q[k].b1, q[k].b1, q[k].b1, q[k].b1};
#pragma unroll
        for (int r = 0; r < NK; ++r) {
          const float xv = xr[16 * pt * INP + r];
#pragma unroll
          for (int k = 0; k < NC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv, q[k].w1[r], acc[k], 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < NC; ++k) h[k][pt] = acc[k];
      }
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt) {
#pragma unroll
        for (int k = 0; k < NC; ++k) {
          float fp[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) { h[k][pt][e] = mact<ACT>(h[k][pt][e]); fp[e] = q[k].w2 * h[k][pt][e]; }
          const float s01 = (odd ? fp[1] : fp[0]) + dpp_get<HTA_DPP_X1>(odd ? fp[0] : fp[1]);
          const float s23 = (odd ? fp[3] : fp[2]) + dpp_get<HTA_DPP_X1>(odd ? fp[2] : fp[3]);
          float sq = (bit1 ? s23 : s01) + dpp_get<HTA_DPP_X2>(bit1 ? s01 : s23);
          sq += dpp_get<HTA_DPP_ROR4>(sq);
          sq += dpp_get<HTA_DPP_ROR8>(sq);
          fw[k * fstride + 16 * pt] = sq;
        }
      }
      HTA_TICK(1);
      __syncthreads();
      HTA_TICK(2);
      // ---- once per point and chain: delta_p = -tau_out r_p (GRAD) or r_p, with r_p = b2 + sum_tiles - y_p; 0 beyond the chunk
      for (int i = tid; i < NC * CP; i += nthr) {
        const int k = (NC > 1 && i >= CP) ? 1 : 0, ip = i - k * CP;
        float f = (NC > 1 && k) ? q[NC - 1].b2 : q[0].b2;
        for (int tt = 0; tt < NU; tt += 4) {          // four independent LDS loads in flight; same left-to-right sum
          const float* fq = fpart + (size_t)k * NU * CP + (size_t)tt * CP + ip;
          const float v0 = fq[0], v1 = fq[CP], v2 = fq[2 * CP], v3 = fq[3 * CP];   // past NU: in-bounds LDS (the next region), masked
          f += v0;
          f += (tt + 1 < NU) ? v1 : 0.0f;
          f += (tt + 2 < NU) ? v2 : 0.0f;
          f += (tt + 3 < NU) ? v3 : 0.0f;
        }
        const float r = f - Ys[c0 + ip];
        rbuf[i] = (ip < cnt) ? (GRAD ? -a.tau_out * r : r) : 0.0f;
      }
      HTA_TICK(3);
      __syncthreads();
      HTA_TICK(4);
      // ---- backward (scalar f32 VALU on purpose: packed f32 ops are slow beside MFMAs)
      const float* xb = ones_lane ? ones : Xs + (size_t)(c0 + 4 * g) * INP + kb;
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt) {
        V4f dl[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) dl[k] = *reinterpret_cast<const V4f*>(rbuf + k * CP + 16 * pt + 4 * g);
        if (GRAD) {
          float bop[NC][4];                              // delta_p act'(h); the common factor w2_u scales the result columns
#pragma unroll
          for (int k = 0; k < NC; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              sdv[k][e] += dl[k][e];
              gw2v[k][e] = fmaf(dl[k][e], h[k][pt][e], gw2v[k][e]);
              if (ACT == 0) bop[k][e] = h[k][pt][e] > 0.0f ? dl[k][e] : 0.0f;
              else bop[k][e] = dl[k][e] * mact_deriv<ACT>(h[k][pt][e]);
              if (NK == 4) gb1v[k] += bop[k][e];
            }
          const float x0 = xb[(16 * pt + 0) * INP], x1 = xb[(16 * pt + 1) * INP], x2 = xb[(16 * pt + 2) * INP], x3 = xb[(16 * pt + 3) * INP];
#pragma unroll
          for (int k = 0; k < NC; ++k) {
            gacc0[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, bop[k][0], gacc0[k], 0, 0, 0);
            gacc1[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1, bop[k][1], gacc1[k], 0, 0, 0);
          }
#pragma unroll
          for (int k = 0; k < NC; ++k) {
            gacc0[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(x2, bop[k][2], gacc0[k], 0, 0, 0);
            gacc1[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(x3, bop[k][3], gacc1[k], 0, 0, 0);
          }
        } else {
#pragma unroll
          for (int k = 0; k < NC; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) ssev[k][e] = fmaf(dl[k][e], dl[k][e], ssev[k][e]);
        }
      }
    }
    HTA_TICK(5);
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      ret[k] = 0;
      if (GRAD) {
        V4f ga;
#pragma unroll
        for (int e = 0; e < 4; ++e) ga[e] = (gacc0[k][e] + gacc1[k][e]) * q[k].w2;
#pragma unroll
        for (int r = 0; r < NK; ++r) gr[k].w1[r] = wvalid[r] ? ga[r] : 0.0f;
        float gb1;
        if (NK < 4) gb1 = __shfl(ga[NK < 4 ? NK : 0], c, 64);          // the all-ones row of X^T: lane (0, c) holds sum_p dh[p, c]
        else gb1 = groups_sum(gb1v[k]) * q[k].w2;
        const float gw2 = groups_sum((gw2v[k][0] + gw2v[k][1]) + (gw2v[k][2] + gw2v[k][3]));
        gr[k].b1 = uvalid ? gb1 : 0.0f; gr[k].w2 = uvalid ? gw2 : 0.0f;
        gr[k].b2 = groups_sum((sdv[k][0] + sdv[k][1]) + (sdv[k][2] + sdv[k][3]));
      } else {
        ret[k] = groups_sum((ssev[k][0] + ssev[k][1]) + (ssev[k][2] + ssev[k][3]));
      }
    }
    HTA_TICK(6);
  }

  // d log p_m / d theta over points [lo, hi) + prior / prior_scale  (S:1156)
  __device__ __forceinline__ void grad_range(const Rec (&q)[NC], int lo, int hi, Rec (&gr)[NC]) {
    float unused[NC];
    pass<true>(q, lo, hi, gr, unused);
    const float ips = 1.0f / a.prior_scale;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
#pragma unroll
      for (int r = 0; r < NK; ++r) gr[k].w1[r] -= ips * a.tau[0] * q[k].w1[r];
      gr[k].b1 -= ips * a.tau[1] * q[k].b1; gr[k].w2 -= ips * a.tau[2] * q[k].w2; gr[k].b2 -= ips * a.tau[3] * q[k].b2;
    }
  }
  // log-likelihood of split points [lo, hi)
  __device__ __forceinline__ void loglik_range(const Rec (&w)[NC], int lo, int hi, float (&ll)[NC]) {
    Rec dummy[NC];
    pass<false>(w, lo, hi, dummy, ll);
#pragma unroll
    for (int k = 0; k < NC; ++k) ll[k] = -0.5f * a.tau_out * ll[k];
  }

  // prior log-density (whole, not divided): sum_l [ -1/2 tau_l sum w^2 + n_l (1/2 log tau_l - 1/2 log 2 pi) ]
  __device__ __forceinline__ void log_prior(const Rec (&w)[NC], float (&lp)[NC]) {
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      float qq = 0;
#pragma unroll
      for (int r = 0; r < NK; ++r) qq = fmaf(w[k].w1[r], w[k].w1[r], qq);      // padding entries are exactly 0
      qq *= a.tau[0];
      if (g == 0) qq += a.tau[1] * w[k].b1 * w[k].b1 + a.tau[2] * w[k].w2 * w[k].w2;
      if (tid == 0) qq += a.tau[3] * w[k].b2 * w[k].b2;
      lp[k] = qq;
    }
    block_sum(lp);
    const float hl2p = 0.9189385332046727f;
    const float n0 = (float)(a.H * a.n_in), n1 = (float)a.H;
#pragma unroll
    for (int k = 0; k < NC; ++k)
      lp[k] = -0.5f * lp[k] + n0 * (0.5f * logf(a.tau[0]) - hl2p) + n1 * (0.5f * logf(a.tau[1]) - hl2p) +
              n1 * (0.5f * logf(a.tau[2]) - hl2p) + (0.5f * logf(a.tau[3]) - hl2p);
  }

  // sum_m log p_m(theta) = full-data log-likelihood + (M / prior_scale) * prior   (S:787-796)
  __device__ __forceinline__ void logp_total(const Rec (&w)[NC], float (&lp)[NC]) {
    float ll[NC], pr[NC];
    loglik_range(w, 0, a.M * a.Nb, ll);
    log_prior(w, pr);
#pragma unroll
    for (int k = 0; k < NC; ++k) lp[k] = ll[k] + ((float)a.M / a.prior_scale) * pr[k];
  }

  __device__ __forceinline__ void kinetic(const Rec (&p)[NC], const Rec& im, float (&kin)[NC]) {
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      float v = 0;
#pragma unroll
      for (int r = 0; r < NK; ++r) v += p[k].w1[r] * im.w1[r] * p[k].w1[r];
      if (g == 0) v += p[k].b1 * im.b1 * p[k].b1 + p[k].w2 * im.w2 * p[k].w2;
      if (tid == 0) v += p[k].b2 * im.b2 * p[k].b2;
      kin[k] = v;
    }
    block_sum(kin);
#pragma unroll
    for (int k = 0; k < NC; ++k) kin[k] *= 0.5f;
  }

  static __device__ __forceinline__ void axpy(Rec& y, float cc, const Rec& x) {       // y += c x
#pragma unroll
    for (int r = 0; r < NK; ++r) y.w1[r] = fmaf(cc, x.w1[r], y.w1[r]);
    y.b1 = fmaf(cc, x.b1, y.b1); y.w2 = fmaf(cc, x.w2, y.w2); y.b2 = fmaf(cc, x.b2, y.b2);
  }
  static __device__ __forceinline__ void drift(Rec& q, float cc, const Rec& im, const Rec& p) {   // q += c M^-1 p
#pragma unroll
    for (int r = 0; r < NK; ++r) q.w1[r] = fmaf(cc * im.w1[r], p.w1[r], q.w1[r]);
    q.b1 = fmaf(cc * im.b1, p.b1, q.b1); q.w2 = fmaf(cc * im.w2, p.w2, q.w2); q.b2 = fmaf(cc * im.b2, p.b2, q.b2);
  }
};

// One workgroup = NC chains (NC = 2: chains 2 b and 2 b + 1 of the launch; a lone last chain runs beside a masked copy of itself).
template <int NK, int NPT, int ACT, int NTMAX, int NC>
__global__ __launch_bounds__(NTMAX, NC == 1 ? 4 : 2) void mlp_mfma_kernel(MlpArgs<float> a, int NU, int Npad) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  typedef MfmaChain<NK, NPT, ACT, NC> Ch;
  typedef typename Ch::Rec Rec;
  constexpr int INP = Ch::INP, CP = Ch::CP;
  Ch ch(a);
  const int tid = threadIdx.x, H = a.H, n_in = a.n_in;
  const int lane = tid & 63;
  ch.tid = tid; ch.nthr = blockDim.x; ch.NU = NU;
  ch.t = __builtin_amdgcn_readfirstlane(tid >> 6);
  ch.c = lane & 15; ch.g = lane >> 4;
  // LDS: X rows padded to INP inputs plus one chunk of zero rows | Y | per-chain, per-tile f partials | residuals | reduction
  // scratch | a block of ones (the X^T row that sums dh over the points, see pass())
  float* Xs = reinterpret_cast<float*>(smem_raw);
  float* Ys = Xs + (size_t)(a.N + CP) * INP;
  ch.Xs = Xs; ch.Ys = Ys;
  ch.fpart = Ys + Npad;
  ch.rbuf = ch.fpart + (size_t)NC * NU * CP;
  ch.red = ch.rbuf + NC * CP;
  float* ones = ch.red + NC * 16;
  ch.ones = ones;
  ch.dump = ones + (size_t)CP * INP + (size_t)ch.t * (64 + 16 * NPT);   // [NU][64 + 16 NPT] scratch behind the ones
  int* perm = reinterpret_cast<int*>(ones + (size_t)CP * INP + (size_t)NU * (64 + 16 * NPT));   // 64 ints: subset order
  for (int e = tid; e < CP * INP; e += ch.nthr) ones[e] = 1.0f;
  for (int e = tid; e < (a.N + CP) * INP; e += ch.nthr) {
    const int i = e / INP, k = e - i * INP;
    Xs[e] = (i < a.N && k < n_in) ? a.X[(size_t)i * n_in + k] : 0.0f;
  }
  for (int e = tid; e < Npad; e += ch.nthr) Ys[e] = e < a.N ? a.Y[e] : 0.0f;
  {  // backward A operand: row rho = c = 4 gamma + r  <->  input NK gamma + r (r < NK); other rows feed ignored outputs
    const int gam = ch.c >> 2, r = ch.c & 3;
    const int k = NK * gam + r;
    ch.kb = k < INP ? k : INP - 1;
    ch.ones_lane = NK < 4 && ch.c == NK;          // rho = NK (gamma 0, r = NK): a free output row, fed with ones
  }
  const int D = H * n_in + 2 * H + 1;
  const int j = 16 * ch.t + ch.c;
  ch.uvalid = j < H;
  const int jj = ch.uvalid ? j : 0;
  int o_w1[NK];
#pragma unroll
  for (int r = 0; r < NK; ++r) {
    const int k = NK * ch.g + r;
    ch.wvalid[r] = ch.uvalid && k < n_in;
    o_w1[r] = ch.wvalid[r] ? jj * n_in + k : 0;
  }
  const int o_b1 = H * n_in + jj, o_w2 = H * n_in + H + jj, o_b2 = H * n_in + 2 * H;
  const bool uwriter = ch.uvalid && ch.g == 0;

  Rec im, mf;       // diagonal M^-1 and sqrt(M) per parameter (1 for the identity)
  const bool dg = a.mass_kind == HTA_MASS_DIAG;
#pragma unroll
  for (int r = 0; r < NK; ++r) {
    const bool ok = dg && ch.wvalid[r];
    im.w1[r] = ok ? a.inv_mass[o_w1[r]] : 1.0f; mf.w1[r] = ok ? a.mass_factor[o_w1[r]] : 1.0f;
  }
  im.b1 = dg ? a.inv_mass[o_b1] : 1.0f; im.w2 = dg ? a.inv_mass[o_w2] : 1.0f; im.b2 = dg ? a.inv_mass[o_b2] : 1.0f;
  mf.b1 = dg ? a.mass_factor[o_b1] : 1.0f; mf.w2 = dg ? a.mass_factor[o_w2] : 1.0f; mf.b2 = dg ? a.mass_factor[o_b2] : 1.0f;

  auto load_rec = [&](const float* th, Rec& w) {
#pragma unroll
    for (int r = 0; r < NK; ++r) w.w1[r] = ch.wvalid[r] ? th[o_w1[r]] : 0.0f;
    w.b1 = ch.uvalid ? th[o_b1] : 0.0f; w.w2 = ch.uvalid ? th[o_w2] : 0.0f; w.b2 = th[o_b2];
  };
  auto store_rec = [&](float* th, const Rec& w) {
#pragma unroll
    for (int r = 0; r < NK; ++r) if (ch.wvalid[r]) th[o_w1[r]] = w.w1[r];
    if (uwriter) { th[o_b1] = w.b1; th[o_w2] = w.w2; }
    if (tid == 0) th[o_b2] = w.b2;
  };

  const int64_t ngroup = (a.C + NC - 1) / NC;
  for (int64_t gidx = blockIdx.x; gidx < ngroup; gidx += gridDim.x) {
    int64_t cidx[NC]; bool live[NC]; uint64_t chain[NC];
    Rec cur[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      const int64_t cc = NC * gidx + k;
      live[k] = cc < a.C;
      cidx[k] = live[k] ? cc : a.C - 1;                 // a dead slot shadows the last chain (its barriers stay matched), no stores
      chain[k] = a.chain_offset + (uint64_t)cidx[k];
      load_rec(a.theta + cidx[k] * D, cur[k]);
    }
    __syncthreads();                                                       // LDS staging (first group) / buffers of the previous group

    if (a.n_traj == 0) {          // evaluation-only: gradient and value of one split closure (parity tests)
      Rec gr[NC];
      float ll[NC], pr[NC];
      const int lo = a.eval_split * a.Nb;
      ch.grad_range(cur, lo, lo + a.Nb, gr);
      ch.loglik_range(cur, lo, lo + a.Nb, ll);
      ch.log_prior(cur, pr);
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        if (!live[k]) continue;
        if (a.grad_out) store_rec(a.grad_out + cidx[k] * D, gr[k]);
        if (a.logp_out && tid == 0) a.logp_out[cidx[k]] = ll[k] + pr[k] / a.prior_scale;
      }
      continue;
    }

#if HTA_TIMING
    ch.tlast = __builtin_readcyclecounter();
#endif
    float lp_cur[NC];
    ch.logp_total(cur, lp_cur);
    int32_t rejected[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) rejected[k] = 0;
    const float eps = a.eps, heps = 0.5f * a.eps;
    const int M = a.M;
    const SplitPlan<float> plan = split_plan<float>(a.integ, M, a.L, eps);
    for (int tr = 0; tr < a.n_traj; ++tr) {
      const int n = a.traj_offset + tr;
      // ---- gibbs (S:185-186 / S:200-201); copies of a parameter draw the same Philox element
      Rec p[NC];
#pragma unroll
      for (int k = 0; k < NC; ++k) {
#pragma unroll
        for (int r = 0; r < NK; ++r)
          p[k].w1[r] = ch.wvalid[r] ? mf.w1[r] * normal_elem<float>(a.seed, chain[k], (uint32_t)n, 0, o_w1[r]) : 0.0f;
        p[k].b1 = ch.uvalid ? mf.b1 * normal_elem<float>(a.seed, chain[k], (uint32_t)n, 0, o_b1) : 0.0f;
        p[k].w2 = ch.uvalid ? mf.w2 * normal_elem<float>(a.seed, chain[k], (uint32_t)n, 0, o_w2) : 0.0f;
        p[k].b2 = mf.b2 * normal_elem<float>(a.seed, chain[k], (uint32_t)n, 0, o_b2);
      }
      float h_old[NC], h_new[NC], lp_new[NC];
      ch.kinetic(p, im, h_old);
#pragma unroll
      for (int k = 0; k < NC; ++k) h_old[k] = -lp_cur[k] + h_old[k];     // S:971
      Rec q[NC], gr[NC];
#pragma unroll
      for (int k = 0; k < NC; ++k) q[k] = cur[k];
      // one stage loop for every integrator; the stage table is split_stage() in mlp.hpp (the same for every chain)
      const int nstage = split_stage_count(a.integ, M, a.L);
      if (a.integ == HTA_SPLIT_RAND) {                                    // S:549: one subset order per trajectory
        __syncthreads();
        if (tid == 0) split_permutation(a.seed, (uint32_t)n, M, perm);
        __syncthreads();
      }
      int prev_m = -1; float prev_dr = 1.0f;
      for (int st = 0, s2 = 0; st < nstage; ++st, s2 = split_next_s2(s2, plan.M2)) {
        int m; float kick, dr;
        split_stage_at<float>(plan, st, s2, perm, m, kick, dr);
        const int lo = m * a.Nb;
        // the same subset at the same parameters as the stage before (no drift since): its gradient is still in `gr` (mlp.hpp)
        if (!split_stage_reuses<float>(prev_m, prev_dr, m)) ch.grad_range(q, lo, lo + a.Nb, gr);
#pragma unroll
        for (int k = 0; k < NC; ++k) {
          Ch::axpy(p[k], kick, gr[k]);
          if (dr != 0.0f) Ch::drift(q[k], dr, im, p[k]);
        }
        prev_m = m; prev_dr = dr;
      }
      if (M == 1 && a.integ == HTA_SPLIT_SYMMETRIC) {
#pragma unroll
        for (int k = 0; k < NC; ++k) Ch::axpy(p[k], -heps, gr[k]);                                            // S:302
      }
      ch.logp_total(q, lp_new);                                           // S:995
      ch.kinetic(p, im, h_new);
      bool acc[NC], any_reset = false;
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        h_new[k] = -lp_new[k] + h_new[k];
        const float u = u23<float>(philox_block(a.seed, chain[k], (uint32_t)n, PURPOSE_MH, 0, 0).x);
        acc[k] = mh_accept<float>(h_old[k], h_new[k], lp_new[k], u);      // S:1000-1004
        if (acc[k]) { cur[k] = q[k]; lp_cur[k] = lp_new[k]; }
        else {
          ++rejected[k];
          any_reset = any_reset || (n == a.burn + 1);
        }
      }
      if (any_reset) {                                                    // Q2 reset to params_init (S:1018): workgroup-uniform
        Rec cand[NC];
        float lpc[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) {
          cand[k] = cur[k];
          if (!acc[k]) load_rec(a.theta_init + cidx[k] * D, cand[k]);
        }
        ch.logp_total(cand, lpc);
#pragma unroll
        for (int k = 0; k < NC; ++k) if (!acc[k]) { cur[k] = cand[k]; lp_cur[k] = lpc[k]; }
      }
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        if (!live[k]) continue;
        if (a.samples && n > a.burn) store_rec(a.samples + ((int64_t)(n - a.burn) * a.C + cidx[k]) * D, cur[k]);
        if (tid == 0) {
          if (a.H_old) a.H_old[(int64_t)tr * a.C + cidx[k]] = h_old[k];
          if (a.H_new) a.H_new[(int64_t)tr * a.C + cidx[k]] = h_new[k];
          if (a.accept) a.accept[(int64_t)tr * a.C + cidx[k]] = acc[k] ? 1 : 0;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      if (!live[k]) continue;
      store_rec(a.theta + cidx[k] * D, cur[k]);
      if (tid == 0 && a.reject_count) a.reject_count[cidx[k]] += rejected[k];
    }
#if HTA_TIMING
    if (tid == 0 && blockIdx.x == 0) for (int k = 0; k < 16; ++k) hta_dbg[k] = ch.tacc[k];
#endif
  }
}

// point tiles per chunk: the whole split when it has at most 128 points (the exact count is a template parameter, so a
// chunk is straight-line code), else chunks of 8 tiles
static inline int mfma_npt(const MlpArgs<float>& a) { const int n = (a.Nb + 15) / 16; return (n < 8 && a.H <= 128) ? n : 8; }

static inline size_t mfma_lds_bytes(const MlpArgs<float>& a, int NK, int NU, int NC, int* npad_out) {
  const int INP = 4 * NK, CP = 16 * mfma_npt(a), Npad = (a.N + CP + 3) / 4 * 4;
  if (npad_out) *npad_out = Npad;
  return ((size_t)(a.N + CP) * INP + Npad + (size_t)NC * NU * CP + NC * CP + NC * 16 + (size_t)CP * INP + (size_t)NU * (64 + CP) + 64 + 4) * sizeof(float);
}

}  // namespace hta
