// Fused (split-)HMC for a one-hidden-layer Bayesian MLP, fp32 on the gfx950 matrix cores
// (v_mfma_f32_16x16x4_f32: f32 in, f32 accumulate, exact - bitwise an fmaf chain).
// Same contract as mlp_hmc.hip (hamiltorch/samplers.py:965-1026 trajectory loop, S:1141-1199 closures,
// S:499-540 split integrator, S:281-302 leapfrog); this is the kernel BASELINE config 4 runs on.
//
// Work layout.  One workgroup per chain, one wave per tile of 16 hidden units; lane l = (g = l >> 4, c = l & 15).
// Wave t owns the parameters of units 16 t .. 16 t + 15 - nothing is replicated across waves but b2:
//   w1[r]   lane (g, c): W1[16 t + c][NK g + r],  r < NK = ceil(n_in / 4)      (one VGPR per r)
//   b1, w2  lane (g, c): unit 16 t + c  (the 4 lane groups hold copies)
// which is at once the B-operand layout of the forward product and the C/D layout of the backward one:
//   forward   Z[p, u] = b1_u + sum_k X[p, k] W1[u, k]      per tile of 16 points: NK MFMAs, A = X (row = point c,
//             K slot g <-> input NK g + r), B = w1[r], C initialised with b1; D: lane (g, c) reg q = point 4 g + q, unit c.
//   f(x_p)    w2_u h[p, u] summed over the 16 units of the tile by 4 DPP row steps, over the tiles through LDS
//             (one float4 per (wave, 4 points)); threads then form the residuals r_p once, in LDS.
//   backward  dW1[u, k] = sum_p dh[p, u] X[p, k],  dh = delta_p w2_u act'(h): 4 MFMAs per point tile accumulating over
//             ALL point tiles in the wave's own accumulator (K slot g of step s <-> point 4 g + s, so B = dh reg s as it
//             lies; A = X^T gathered from LDS).  D rows 4 g + r <-> input NK g + r: exactly the w1[r] layout.
//   db1, dw2, db2, sum r^2: in-lane sums over the tile registers, then two cross-group shuffles.
// The activations of a point chunk (<= 128 points) stay in registers between the passes: no chunk matrix in LDS,
// two barriers per chunk, and per-wave state of 3 + NK registers per record.
#include "mlp.hpp"
#include "philox.hpp"

#ifndef HTA_TIMING
#define HTA_TIMING 0   // developer cycle counters per phase of a gradient (wave 0 of block 0), read by tools/scratch/mlp_ablate.cpp
#endif
#if HTA_TIMING
__device__ unsigned long long hta_dbg[16];
extern "C" void hta_dbg_read(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(hta_dbg), sizeof(hta_dbg)); }
#define HTA_TICK(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); tacc[k] += now_ - tlast; tlast = now_; } while (0)
#else
#define HTA_TICK(k) do {} while (0)
#endif

namespace hta {

void profile_begin(hipStream_t s);
void profile_end(hipStream_t s);

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

template <int NK, int NPT, int ACT>
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

  __device__ __forceinline__ float block_sum(float v) {
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) red[t] = v;
    __syncthreads();
    float s = 0;
    for (int i = 0; i < NU; ++i) s += red[i];
    return s;
  }

  // Likelihood part of split points [lo, hi).  GRAD: its gradient into gr (prior added by the caller), returns 0;
  // otherwise returns the sum of squared residuals (every lane).
  template <bool GRAD> __device__ __forceinline__ float pass(const Rec& q, int lo, int hi, Rec& gr) {
    V4f gacc0 = {0, 0, 0, 0}, gacc1 = {0, 0, 0, 0};
    float sdv[4] = {0, 0, 0, 0}, gw2v[4] = {0, 0, 0, 0}, ssev[4] = {0, 0, 0, 0};
    float gb1v = 0;
    const V4f binit = {q.b1, q.b1, q.b1, q.b1};
    const bool odd = c & 1, bit1 = c & 2;
    for (int c0 = lo; c0 < hi; c0 += CP) {
      const int cnt = min(CP, hi - c0);               // a ragged last chunk runs all NPT tiles; its extra points get delta = 0
      V4f h[NPT];
      HTA_TICK(0);
      // ---- forward: pre-activations of this wave's 16 units at every point of the chunk, then the activations and this
      //      tile's share of f(x_p) = sum_u w2_u h[p, u]: a reduce-scatter over the quad (the 4 point registers end up one
      //      per lane), then rotations by 4 and 8 across the 16 unit lanes.  Straight-line code: tile pt + 1's MFMAs are in
      //      flight under tile pt's VALU work.
      const float* xr = Xs + (size_t)(c0 + c) * INP + NK * g;
      // lanes c >= 4 hold copies: they store to a per-lane dump slot instead of branching around the store
      float* fw = (c < 4) ? fpart + (size_t)t * CP + 4 * g + c : dump + (tid & 63);
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt) {
        V4f acc = binit;
#pragma unroll
        for (int r = 0; r < NK; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xr[16 * pt * INP + r], q.w1[r], acc, 0, 0, 0);
        h[pt] = acc;
      }
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt) {
        float fp[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { h[pt][e] = mact<ACT>(h[pt][e]); fp[e] = q.w2 * h[pt][e]; }
        const float s01 = (odd ? fp[1] : fp[0]) + dpp_get<HTA_DPP_X1>(odd ? fp[0] : fp[1]);
        const float s23 = (odd ? fp[3] : fp[2]) + dpp_get<HTA_DPP_X1>(odd ? fp[2] : fp[3]);
        float sq = (bit1 ? s23 : s01) + dpp_get<HTA_DPP_X2>(bit1 ? s01 : s23);
        sq += dpp_get<HTA_DPP_ROR4>(sq);
        sq += dpp_get<HTA_DPP_ROR8>(sq);
        fw[16 * pt] = sq;
      }
      HTA_TICK(1);
      __syncthreads();
      HTA_TICK(2);
      // ---- once per point: delta_p = -tau_out r_p (GRAD) or r_p, with r_p = b2 + sum_tiles - y_p; 0 beyond the chunk
      for (int i = tid; i < CP; i += nthr) {
        float f = q.b2;
        for (int tt = 0; tt < NU; tt += 4) {          // four independent LDS loads in flight; same left-to-right sum
          const float* fq = fpart + (size_t)tt * CP + i;
          const float v0 = fq[0], v1 = fq[CP], v2 = fq[2 * CP], v3 = fq[3 * CP];   // past NU: in-bounds LDS (rbuf/ones), masked
          f += v0;
          f += (tt + 1 < NU) ? v1 : 0.0f;
          f += (tt + 2 < NU) ? v2 : 0.0f;
          f += (tt + 3 < NU) ? v3 : 0.0f;
        }
        const float r = f - Ys[c0 + i];
        rbuf[i] = (i < cnt) ? (GRAD ? -a.tau_out * r : r) : 0.0f;
      }
      HTA_TICK(3);
      __syncthreads();
      HTA_TICK(4);
      // ---- backward (scalar f32 VALU on purpose: packed f32 ops are slow beside MFMAs)
      const float* xb = ones_lane ? ones : Xs + (size_t)(c0 + 4 * g) * INP + kb;
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt) {
        const V4f dl = *reinterpret_cast<const V4f*>(rbuf + 16 * pt + 4 * g);
        if (GRAD) {
          float bop[4];                                  // delta_p act'(h); the common factor w2_u scales the result columns
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            sdv[e] += dl[e];
            gw2v[e] = fmaf(dl[e], h[pt][e], gw2v[e]);
            if (ACT == 0) bop[e] = h[pt][e] > 0.0f ? dl[e] : 0.0f;
            else bop[e] = dl[e] * mact_deriv<ACT>(h[pt][e]);
            if (NK == 4) gb1v += bop[e];
          }
          gacc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[(16 * pt + 0) * INP], bop[0], gacc0, 0, 0, 0);
          gacc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[(16 * pt + 1) * INP], bop[1], gacc1, 0, 0, 0);
          gacc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[(16 * pt + 2) * INP], bop[2], gacc0, 0, 0, 0);
          gacc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[(16 * pt + 3) * INP], bop[3], gacc1, 0, 0, 0);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) ssev[e] = fmaf(dl[e], dl[e], ssev[e]);
        }
      }
    }
    HTA_TICK(5);
    float ret = 0;
    if (GRAD) {
      V4f ga;
#pragma unroll
      for (int e = 0; e < 4; ++e) ga[e] = (gacc0[e] + gacc1[e]) * q.w2;
#pragma unroll
      for (int r = 0; r < NK; ++r) gr.w1[r] = wvalid[r] ? ga[r] : 0.0f;
      float gb1;
      if (NK < 4) gb1 = __shfl(ga[NK < 4 ? NK : 0], c, 64);          // the all-ones row of X^T: lane (0, c) holds sum_p dh[p, c]
      else gb1 = groups_sum(gb1v) * q.w2;
      const float gw2 = groups_sum((gw2v[0] + gw2v[1]) + (gw2v[2] + gw2v[3]));
      gr.b1 = uvalid ? gb1 : 0.0f; gr.w2 = uvalid ? gw2 : 0.0f;
      gr.b2 = groups_sum((sdv[0] + sdv[1]) + (sdv[2] + sdv[3]));
    } else {
      ret = groups_sum((ssev[0] + ssev[1]) + (ssev[2] + ssev[3]));
    }
    HTA_TICK(6);
    return ret;
  }

  // d log p_m / d theta over points [lo, hi) + prior / prior_scale  (S:1156)
  __device__ __forceinline__ void grad_range(const Rec& q, int lo, int hi, Rec& gr) {
    pass<true>(q, lo, hi, gr);
    const float ips = 1.0f / a.prior_scale;
#pragma unroll
    for (int r = 0; r < NK; ++r) gr.w1[r] -= ips * a.tau[0] * q.w1[r];
    gr.b1 -= ips * a.tau[1] * q.b1; gr.w2 -= ips * a.tau[2] * q.w2; gr.b2 -= ips * a.tau[3] * q.b2;
  }
  // log-likelihood of split points [lo, hi)
  __device__ __forceinline__ float loglik_range(const Rec& w, int lo, int hi) {
    Rec dummy;
    return -0.5f * a.tau_out * pass<false>(w, lo, hi, dummy);
  }

  // prior log-density (whole, not divided): sum_l [ -1/2 tau_l sum w^2 + n_l (1/2 log tau_l - 1/2 log 2 pi) ]
  __device__ __forceinline__ float log_prior(const Rec& w) {
    float qq = 0;
#pragma unroll
    for (int r = 0; r < NK; ++r) qq = fmaf(w.w1[r], w.w1[r], qq);             // padding entries are exactly 0
    qq *= a.tau[0];
    if (g == 0) qq += a.tau[1] * w.b1 * w.b1 + a.tau[2] * w.w2 * w.w2;
    if (tid == 0) qq += a.tau[3] * w.b2 * w.b2;
    qq = block_sum(qq);
    const float hl2p = 0.9189385332046727f;
    const float n0 = (float)(a.H * a.n_in), n1 = (float)a.H;
    return -0.5f * qq + n0 * (0.5f * logf(a.tau[0]) - hl2p) + n1 * (0.5f * logf(a.tau[1]) - hl2p) +
           n1 * (0.5f * logf(a.tau[2]) - hl2p) + (0.5f * logf(a.tau[3]) - hl2p);
  }

  // sum_m log p_m(theta) = full-data log-likelihood + (M / prior_scale) * prior   (S:787-796)
  __device__ __forceinline__ float logp_total(const Rec& w) {
    return loglik_range(w, 0, a.M * a.Nb) + ((float)a.M / a.prior_scale) * log_prior(w);
  }

  __device__ __forceinline__ float kinetic(const Rec& p, const Rec& im) {
    float k = 0;
#pragma unroll
    for (int r = 0; r < NK; ++r) k += p.w1[r] * im.w1[r] * p.w1[r];
    if (g == 0) k += p.b1 * im.b1 * p.b1 + p.w2 * im.w2 * p.w2;
    if (tid == 0) k += p.b2 * im.b2 * p.b2;
    return 0.5f * block_sum(k);
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

template <int NK, int NPT, int ACT, int NTMAX>
__global__ __launch_bounds__(NTMAX, 4) void mlp_mfma_kernel(MlpArgs<float> a, int NU, int Npad) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  typedef MfmaChain<NK, NPT, ACT> Ch;
  typedef typename Ch::Rec Rec;
  constexpr int INP = Ch::INP, CP = Ch::CP;
  Ch ch(a);
  const int tid = threadIdx.x, H = a.H, n_in = a.n_in;
  const int lane = tid & 63;
  ch.tid = tid; ch.nthr = blockDim.x; ch.NU = NU;
  ch.t = __builtin_amdgcn_readfirstlane(tid >> 6);
  ch.c = lane & 15; ch.g = lane >> 4;
  // LDS: X rows padded to INP inputs plus one chunk of zero rows | Y | per-tile f partials | residuals | reduction scratch |
  // a block of ones (the X^T row that sums dh over the points, see pass())
  float* Xs = reinterpret_cast<float*>(smem_raw);
  float* Ys = Xs + (size_t)(a.N + CP) * INP;
  ch.Xs = Xs; ch.Ys = Ys;
  ch.fpart = Ys + Npad;
  ch.rbuf = ch.fpart + (size_t)NU * CP;
  ch.red = ch.rbuf + CP;
  float* ones = ch.red + 16;
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

  for (int64_t cidx = blockIdx.x; cidx < a.C; cidx += gridDim.x) {
    const uint64_t chain = a.chain_offset + (uint64_t)cidx;
    Rec cur;
    load_rec(a.theta + cidx * D, cur);
    __syncthreads();                                                       // LDS staging (first chain) / buffers of the previous chain

    if (a.n_traj == 0) {          // evaluation-only: gradient and value of one split closure (parity tests)
      Rec gr;
      const int lo = a.eval_split * a.Nb;
      ch.grad_range(cur, lo, lo + a.Nb, gr);
      const float lp = ch.loglik_range(cur, lo, lo + a.Nb) + ch.log_prior(cur) / a.prior_scale;
      if (a.grad_out) store_rec(a.grad_out + cidx * D, gr);
      if (a.logp_out && tid == 0) a.logp_out[cidx] = lp;
      continue;
    }

#if HTA_TIMING
    ch.tlast = __builtin_readcyclecounter();
#endif
    float lp_cur = ch.logp_total(cur);
    int32_t rejected = 0;
    const float eps = a.eps, heps = 0.5f * a.eps;
    const int M = a.M;
    for (int tr = 0; tr < a.n_traj; ++tr) {
      const int n = a.traj_offset + tr;
      // ---- gibbs (S:185-186 / S:200-201); copies of a parameter draw the same Philox element
      Rec p;
#pragma unroll
      for (int r = 0; r < NK; ++r)
        p.w1[r] = ch.wvalid[r] ? mf.w1[r] * normal_elem<float>(a.seed, chain, (uint32_t)n, 0, o_w1[r]) : 0.0f;
      p.b1 = ch.uvalid ? mf.b1 * normal_elem<float>(a.seed, chain, (uint32_t)n, 0, o_b1) : 0.0f;
      p.w2 = ch.uvalid ? mf.w2 * normal_elem<float>(a.seed, chain, (uint32_t)n, 0, o_w2) : 0.0f;
      p.b2 = mf.b2 * normal_elem<float>(a.seed, chain, (uint32_t)n, 0, o_b2);
      const float h_old = -lp_cur + ch.kinetic(p, im);                    // S:971
      Rec q = cur, gr;
      // one stage loop for every integrator; the stage table is split_stage() in mlp.hpp
      const int nstage = split_stage_count(a.integ, M, a.L);
      if (a.integ == HTA_SPLIT_RAND) {                                    // S:549: one subset order per trajectory
        __syncthreads();
        if (tid == 0) split_permutation(a.seed, (uint32_t)n, M, perm);
        __syncthreads();
      }
      int prev_m = -1; float prev_dr = 1.0f;
      for (int st = 0; st < nstage; ++st) {
        int m; float kick, dr;
        split_stage<float>(a.integ, M, a.L, st, eps, perm, m, kick, dr);
        const int lo = m * a.Nb;
        // the same subset at the same parameters as the stage before (no drift since): its gradient is still in `gr` (mlp.hpp)
        if (!split_stage_reuses<float>(prev_m, prev_dr, m)) ch.grad_range(q, lo, lo + a.Nb, gr);
        Ch::axpy(p, kick, gr);
        if (dr != 0.0f) Ch::drift(q, dr, im, p);
        prev_m = m; prev_dr = dr;
      }
      if (M == 1 && a.integ == HTA_SPLIT_SYMMETRIC) Ch::axpy(p, -heps, gr);                                 // S:302
      const float lp_new = ch.logp_total(q);                              // S:995
      const float h_new = -lp_new + ch.kinetic(p, im);
      const float u = u23<float>(philox_block(a.seed, chain, (uint32_t)n, PURPOSE_MH, 0, 0).x);
      const bool acc = mh_accept<float>(h_old, h_new, lp_new, u);         // S:1000-1004
      if (acc) { cur = q; lp_cur = lp_new; }
      else {
        ++rejected;
        if (n == a.burn + 1) {                                            // Q2 reset to params_init (S:1018)
          load_rec(a.theta_init + cidx * D, cur);
          lp_cur = ch.logp_total(cur);
        }
      }
      if (a.samples && n > a.burn) store_rec(a.samples + ((int64_t)(n - a.burn) * a.C + cidx) * D, cur);
      if (tid == 0) {
        if (a.H_old) a.H_old[(int64_t)tr * a.C + cidx] = h_old;
        if (a.H_new) a.H_new[(int64_t)tr * a.C + cidx] = h_new;
        if (a.accept) a.accept[(int64_t)tr * a.C + cidx] = acc ? 1 : 0;
      }
    }
    store_rec(a.theta + cidx * D, cur);
    if (tid == 0 && a.reject_count) a.reject_count[cidx] += rejected;
#if HTA_TIMING
    if (tid == 0 && blockIdx.x == 0) for (int k = 0; k < 16; ++k) hta_dbg[k] = ch.tacc[k];
#endif
  }
}

// point tiles per chunk: the whole split when it has at most 128 points (the exact count is a template parameter, so a
// chunk is straight-line code), else chunks of 8 tiles
static int mfma_npt(const MlpArgs<float>& a) { const int n = (a.Nb + 15) / 16; return (n < 8 && a.H <= 128) ? n : 8; }

static size_t mfma_lds_bytes(const MlpArgs<float>& a, int NK, int NU, int* npad_out) {
  const int INP = 4 * NK, CP = 16 * mfma_npt(a), Npad = (a.N + CP + 3) / 4 * 4;
  if (npad_out) *npad_out = Npad;
  return ((size_t)(a.N + CP) * INP + Npad + (size_t)NU * CP + CP + 16 + (size_t)CP * INP + (size_t)NU * (64 + CP) + 64 + 4) * sizeof(float);
}

bool mlp_mfma_eligible(const MlpArgs<float>& a) {
  // Gaussian likelihood only: the other likelihoods run on the VALU kernel (mlp_hmc.hip).  Their exp / log1p sequences - even
  // out of line, even behind a call never taken - cost this kernel 2-4 % on BASELINE config 4 at its 128-register cap (measured).
  if (a.loss != HTA_LOSS_REGRESSION) return false;
  if (a.n_in < 1 || a.n_in > 16 || a.H < 1 || a.H > 256) return false;
  if (!(a.mass_kind == HTA_MASS_NONE || a.mass_kind == HTA_MASS_DIAG)) return false;
  const int NK = (a.n_in + 3) / 4, NU = (a.H + 15) / 16;
  return mfma_lds_bytes(a, NK, NU, nullptr) <= 150 * 1024;
}

template <int NK, int NPT, int ACT, int NTMAX> static int launch_mfma(const MlpArgs<float>& a, hipStream_t s) {
  const int NU = (a.H + 15) / 16;
  int Npad;
  const size_t lds = mfma_lds_bytes(a, NK, NU, &Npad);
  static DevOnce done;
  if (!done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_mfma_kernel<NK, NPT, ACT, NTMAX>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { set_error("hta_mlp_hmc: hipFuncSetAttribute: %s", hipGetErrorString(e)); return HTA_ERR_LAUNCH; }
    done = true;
  }
  const int grid = (int)(a.C < 8192 ? a.C : 8192);
  profile_begin(s);
  note_route("mlp_mfma_kernel<%d,%d,%d,%d>", NK, NPT, ACT, NTMAX);
  mlp_mfma_kernel<NK, NPT, ACT, NTMAX><<<grid, 64 * NU, lds, s>>>(a, NU, Npad);
  profile_end(s);
  HTA_CHECK_LAUNCH("hta_mlp_hmc (mfma)");
  return HTA_OK;
}

template <int NK, int NPT, int NTMAX> static int launch_mfma_act(const MlpArgs<float>& a, hipStream_t s) {
  if (a.act == 0) return launch_mfma<NK, NPT, 0, NTMAX>(a, s);
  if (a.act == 1) return launch_mfma<NK, NPT, 1, NTMAX>(a, s);
  return launch_mfma<NK, NPT, 2, NTMAX>(a, s);
}

template <int NK> int launch_mfma_nk(const MlpArgs<float>& a, hipStream_t s) {
  if (a.H > 128) return launch_mfma_act<NK, 8, 1024>(a, s);      // wide layers: 8-tile chunks only (ragged chunks run padded)
  switch (mfma_npt(a)) {
    case 1: return launch_mfma_act<NK, 1, 512>(a, s);
    case 2: return launch_mfma_act<NK, 2, 512>(a, s);
    case 3: return launch_mfma_act<NK, 3, 512>(a, s);
    case 4: return launch_mfma_act<NK, 4, 512>(a, s);
    case 5: return launch_mfma_act<NK, 5, 512>(a, s);
    case 6: return launch_mfma_act<NK, 6, 512>(a, s);
    case 7: return launch_mfma_act<NK, 7, 512>(a, s);
    default: return launch_mfma_act<NK, 8, 512>(a, s);
  }
}

int mlp_mfma(const MlpArgs<float>& a, hipStream_t s) {
#ifdef HTA_MLP_SINGLE
  return launch_mfma<2, 7, 0, 512>(a, s);
#else
  const int NK = (a.n_in + 3) / 4;
  if (NK == 1) return launch_mfma_nk<1>(a, s);
  if (NK == 2) return launch_mfma_nk<2>(a, s);
  if (NK == 3) return launch_mfma_nk<3>(a, s);
  return launch_mfma_nk<4>(a, s);
#endif
}

}  // namespace hta
