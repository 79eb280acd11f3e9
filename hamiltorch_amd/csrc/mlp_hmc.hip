// Fused (split-)HMC for a one-hidden-layer Bayesian MLP with a Gaussian (regression) likelihood, any dtype and shape
// (n_in <= 32, H <= 1024): the fp64 / wide-layer route, and the fp32 parity partner (tuning key "mlp_valu") of
// mlp_mfma.hip, which runs BASELINE config 4.  One workgroup per chain; the whole trajectory loop of sample()
// (hamiltorch/samplers.py:965-1026) runs on-chip with
//   log p_m(theta) = -1/2 tau_out sum_{i in split m} (f(x_i) - y_i)^2
//                    + (1/prior_scale) sum_layers Normal(0, tau_l^-1/2).log_prob(w).sum()
// (define_model_log_prob, S:1141-1199; one closure per DataLoader batch, S:1251-1255) and either
//   * the symmetric split integrator (S:499-540): 2M gradient half-kicks + 2(M-1) drifts per step, or
//   * plain leapfrog over the full data (S:281-302) when M == 1 (sample_model).
//
// Work layout.  theta = [W1[H,in] | b1[H] | W2[1,H] | b2[1]] (U:121-122 order).  Wave (g, s): lane l is
// hidden unit j = 64 g + l, the wave is point slice s of PS; the unit's weights, momentum and gradient
// live in its registers (replicated over the PS slice waves, which stay bit-identical).  Because a wave
// works on ONE data point at a time, x_i is wave-uniform: it comes through the scalar cache
// (s_load_dwordx8) straight into SGPR operands of v_pk_fma_f32 - no LDS or vector-memory traffic for
// the data set at all.  A gradient of one data chunk is two passes over the chunk's points:
//   1. h_ij = act(b1_j + W1_j . x_i) -> LDS [i][j];  4 lanes per row form f(x_i) = b2 + sum_j W2_j h_ij
//      (float4 reads), the residual and delta_i = -tau_out (f(x_i) - y_i)  (LDS vector);
//   2. every wave re-reads h_ij of its points and accumulates dW2_j, db1_j, dW1_j. in registers
//      (packed FMAs, x_i from SGPRs); one LDS exchange sums the PS slice partials per gradient.
// HBM traffic per trajectory: one [D] sample row per chain.
#include "mlp.hpp"
#include "philox.hpp"

#ifndef HTA_UNR
#define HTA_UNR 1
#endif
#define HTA_STR_(x) #x
#define HTA_UNROLL_(n) _Pragma(HTA_STR_(unroll n))
#define HTA_UNROLL(n) HTA_UNROLL_(n)
#ifndef HTA_ABL
#define HTA_ABL 0      // developer ablation bits (tools/scratch/mlp_ablate.sh); 0 in the product build
#endif

#ifdef HTA_TIMING
#undef HTA_TIMING     // the counters live in mlp_mfma.hip now; -DHTA_TIMING_VALU=1 brings these back (with mlp_valu set)
#endif
#ifdef HTA_TIMING_VALU
#define HTA_TIMING 1
#else
#define HTA_TIMING 0
#endif
#if HTA_TIMING
__device__ unsigned long long hta_dbg[16];
extern "C" void hta_dbg_read(unsigned long long* out) { hipMemcpyFromSymbol(out, HIP_SYMBOL(hta_dbg), sizeof(hta_dbg)); }
#define HTA_TICK(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); tacc[k] += now_ - tlast; tlast = now_; } while (0)
#else
#define HTA_TICK(k) do {} while (0)
#endif

namespace hta {

void profile_begin(hipStream_t s);
void profile_end(hipStream_t s);

// activation from the pre-activation, and its derivative from the activation value itself
template <int ACT, typename T> __device__ __forceinline__ T act_fn(T z) {
  if (ACT == 0) return z > (T)0 ? z : (T)0;
  if (ACT == 1) return tanh(z);
  return (T)1 / ((T)1 + exp(-z));
}
template <int ACT, typename T> __device__ __forceinline__ T act_deriv(T h) {
  if (ACT == 0) return h > (T)0 ? (T)1 : (T)0;
  if (ACT == 1) return (T)1 - h * h;
  return h * ((T)1 - h);
}

template <int G, typename T> __device__ __forceinline__ T slice_sum(T v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <typename T, int INMAX, int NT, int ACT, bool EXACT>
struct MlpChain {
  typedef T P2 __attribute__((ext_vector_type(2)));
  typedef T V4 __attribute__((ext_vector_type(4)));
  typedef const __attribute__((address_space(4))) T* CPtr;       // constant address space: uniform loads go through the scalar cache
  static constexpr int NP = INMAX / 2;
  static constexpr int GS = INMAX + 4;                           // floats per (slice, unit) record of the slice exchange
  // per-thread parameter record of hidden unit j; w1 in pairs for v_pk_fma_f32
  struct Rec { P2 w1[NP]; T b1; T w2; T b2; };
  const MlpArgs<T>& a;
  CPtr Xc;
  T* Ys; T* cm; T* dv; T* red; T* w2s; T* gp; int* perm;
  int nbch, ldc, Hp, j, slice, PS, UGL, tid;
  bool unit;      // j < H and the wave has a slice
#if HTA_TIMING
  unsigned long long tacc[16] = {0}, tlast = 0;
#endif
  __device__ MlpChain(const MlpArgs<T>& a_) : a(a_) {}

  static __device__ __forceinline__ P2 fma2(P2 x, P2 y, P2 z) { return __builtin_elementwise_fma(x, y, z); }

  // x_i (wave-uniform) as NP pairs, zero-padded to INMAX; EXACT (n_in == INMAX): the whole row in one scalar load
  __device__ __forceinline__ void load_x(int i, P2 (&x)[NP]) const {
    if (EXACT) {
      typedef T VX __attribute__((ext_vector_type(INMAX)));
      const VX v = *reinterpret_cast<const __attribute__((address_space(4))) VX*>(Xc + (int64_t)i * INMAX);
#pragma unroll
      for (int k = 0; k < NP; ++k) x[k] = P2{v[2 * k], v[2 * k + 1]};
    } else {
      CPtr r = Xc + (int64_t)i * a.n_in;
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        x[k].x = (2 * k < a.n_in) ? r[2 * k] : (T)0;
        x[k].y = (2 * k + 1 < a.n_in) ? r[2 * k + 1] : (T)0;
      }
    }
  }

  __device__ __forceinline__ T block_sum2(T v, T& v2) {   // two sums at once
    v = wave_sum(v); v2 = wave_sum(v2);
    const int w = tid >> 6;
    __syncthreads();
    if ((tid & 63) == 0) { red[2 * w] = v; red[2 * w + 1] = v2; }
    __syncthreads();
    T t1 = 0, t2 = 0;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) { t1 += red[2 * i]; t2 += red[2 * i + 1]; }
    v2 = t2;
    return t1;
  }

  // pass 1 over points [lo, lo+cnt): h -> cm, delta_i -> dv[i]; returns (sum r_i^2, sum delta_i)
  __device__ __forceinline__ T forward_chunk(const Rec& w, int lo, int cnt, T& sum_delta) {
    HTA_TICK(0);
    __syncthreads();
    HTA_TICK(1);
    if (unit && slice == 0) w2s[j] = w.w2;
    if (slice < PS && !(HTA_ABL & 1)) {
HTA_UNROLL(HTA_UNR)
      for (int i = slice; i < cnt; i += PS) {
        P2 x[NP];
        load_x(lo + i, x);
        P2 acc = P2{w.b1, (T)0};
#pragma unroll
        for (int k = 0; k < NP; ++k) acc = fma2(w.w1[k], x[k], acc);
        cm[i * ldc + j] = unit ? act_fn<ACT, T>(acc.x + acc.y) : (T)0;    // every lane stores (ldc covers all lanes): no exec branch per point
      }
    }
    HTA_TICK(2);
    __syncthreads();
    HTA_TICK(3);
    T sse = 0, sd = 0;
    constexpr int RS = 4;                      // lanes per row of the chunk matrix
    const int part = tid % RS, nq = Hp / 4;
    const V4* wv = reinterpret_cast<const V4*>(w2s);
    if (!(HTA_ABL & 2)) for (int i = tid / RS; i < cnt; i += NT / RS) {
      const V4* row = reinterpret_cast<const V4*>(cm + i * ldc);
      P2 acc = P2{(T)0, (T)0};
      for (int q = part; q < nq; q += RS) {
        const V4 r = row[q], ww = wv[q];
        acc = fma2(P2{ww.x, ww.y}, P2{r.x, r.y}, acc);
        acc = fma2(P2{ww.z, ww.w}, P2{r.z, r.w}, acc);
      }
      const T out = slice_sum<RS>(acc.x + acc.y) + w.b2;
      if (part == 0) {
        T d, e;
        mlp_point_loss<T>(a.loss, out, Ys[lo + i], a.tau_out, d, e);
        dv[i] = d;
        sse += e; sd += d;
      }
    }
    HTA_TICK(4);
    if (!(HTA_ABL & 16)) sse = block_sum2(sse, sd);
    HTA_TICK(5);
    sum_delta = sd;
    return sse;
  }

  // pass 2: accumulate the likelihood gradient of points [lo, lo+cnt) into g (this wave's slice)
  __device__ __forceinline__ void backward_chunk(const Rec& w, int lo, int cnt, Rec& g) {
    if (slice < PS && !(HTA_ABL & 4)) {
HTA_UNROLL(HTA_UNR)
      for (int i = slice; i < cnt; i += PS) {
        P2 x[NP];
        load_x(lo + i, x);
        const T d = dv[i];
        const T h = cm[i * ldc + j];                                       // 0 in lanes without a unit
        g.w2 = fma(d, h, g.w2);
        const T dh = d * w.w2 * act_deriv<ACT, T>(h);
        g.b1 += dh;
        const P2 dh2 = P2{dh, dh};
#pragma unroll
        for (int k = 0; k < NP; ++k) g.w1[k] = fma2(dh2, x[k], g.w1[k]);
      }
    }
  }

  // d log p_m / d theta over points [lo, hi) + prior / prior_scale; returns log-likelihood part
  __device__ __forceinline__ T grad_range(const Rec& w, int lo, int hi, Rec& g) {
#pragma unroll
    for (int k = 0; k < NP; ++k) g.w1[k] = P2{(T)0, (T)0};
    g.b1 = 0; g.w2 = 0; g.b2 = 0;
    T sse = 0;
    for (int c0 = lo; c0 < hi; c0 += nbch) {
      const int cnt = min(nbch, hi - c0);
      T sd;
      sse += forward_chunk(w, c0, cnt, sd);
      g.b2 += sd;
      backward_chunk(w, c0, cnt, g);
      HTA_TICK(6);
    }
    // sum the PS slice partials of every unit (same order in every slice wave: the replicas stay identical)
    if (PS > 1 && !(HTA_ABL & 8)) {
      __syncthreads();                       // the exchange records alias the chunk matrix: every wave is done with h
      if (unit) {
        V4* rec = reinterpret_cast<V4*>(gp + (size_t)tid * GS);            // == (slice * UGL + j) * GS
#pragma unroll
        for (int k = 0; k < NP / 2; ++k) rec[k] = V4{g.w1[2 * k].x, g.w1[2 * k].y, g.w1[2 * k + 1].x, g.w1[2 * k + 1].y};
        rec[NP / 2] = V4{g.b1, g.w2, (T)0, (T)0};
      }
      __syncthreads();
      if (unit) {
#pragma unroll
        for (int k = 0; k < NP; ++k) g.w1[k] = P2{(T)0, (T)0};
        g.b1 = 0; g.w2 = 0;
        for (int ss = 0; ss < PS; ++ss) {
          const V4* rec = reinterpret_cast<const V4*>(gp + ((size_t)ss * UGL + j) * GS);
#pragma unroll
          for (int k = 0; k < NP / 2; ++k) {
            const V4 v = rec[k];
            g.w1[2 * k] += P2{v.x, v.y}; g.w1[2 * k + 1] += P2{v.z, v.w};
          }
          const V4 v = rec[NP / 2];
          g.b1 += v.x; g.w2 += v.y;
        }
      }
    }
    HTA_TICK(7);
    const T ips = (T)1 / a.prior_scale;
    const T t0 = ips * a.tau[0];
#pragma unroll
    for (int k = 0; k < NP; ++k) g.w1[k] = fma2(P2{-t0, -t0}, w.w1[k], g.w1[k]);     // S:1156: d/dw Normal(0, tau^-1/2).log_prob
    g.b1 -= ips * a.tau[1] * w.b1; g.w2 -= ips * a.tau[2] * w.w2; g.b2 -= ips * a.tau[3] * w.b2;
    return (T)-0.5 * a.tau_out * sse;
  }

  // prior log-density (whole, not divided): sum_l [ -1/2 tau_l sum w^2 + n_l (1/2 log tau_l - 1/2 log 2 pi) ]
  __device__ __forceinline__ T log_prior(const Rec& w) {
    T q = 0, dummy = 0;
    if (unit && slice == 0) {
      T sw = 0;
#pragma unroll
      for (int k = 0; k < NP; ++k) sw += w.w1[k].x * w.w1[k].x + w.w1[k].y * w.w1[k].y;   // padding weights are exactly 0
      q = a.tau[0] * sw + a.tau[1] * w.b1 * w.b1 + a.tau[2] * w.w2 * w.w2;
    }
    if (tid == 0) q += a.tau[3] * w.b2 * w.b2;
    q = block_sum2(q, dummy);
    const T hl2p = (T)0.9189385332046727;
    const T n0 = (T)(a.H * a.n_in), n1 = (T)a.H;
    return (T)-0.5 * q + n0 * ((T)0.5 * log(a.tau[0]) - hl2p) + n1 * ((T)0.5 * log(a.tau[1]) - hl2p) +
           n1 * ((T)0.5 * log(a.tau[2]) - hl2p) + ((T)0.5 * log(a.tau[3]) - hl2p);
  }

  // sum_m log p_m(theta) = full-data log-likelihood + (M / prior_scale) * prior   (S:787-796)
  __device__ __forceinline__ T logp_total(const Rec& w) {
    T sse = 0;
    const int used = a.M * a.Nb;
    for (int c0 = 0; c0 < used; c0 += nbch) {
      T sd;
      sse += forward_chunk(w, c0, min(nbch, used - c0), sd);
    }
    return (T)-0.5 * a.tau_out * sse + ((T)a.M / a.prior_scale) * log_prior(w);
  }

  __device__ __forceinline__ T kinetic(const Rec& p, const Rec& im) {
    T k = 0, dummy = 0;
    if (unit && slice == 0) {
#pragma unroll
      for (int q = 0; q < NP; ++q) {                                       // padding momenta are exactly 0
        const P2 t = p.w1[q] * im.w1[q] * p.w1[q];
        k += t.x + t.y;
      }
      k += p.b1 * im.b1 * p.b1 + p.w2 * im.w2 * p.w2;
    }
    if (tid == 0) k += p.b2 * im.b2 * p.b2;
    return (T)0.5 * block_sum2(k, dummy);
  }

  static __device__ __forceinline__ void axpy(Rec& y, T c, const Rec& x) {       // y += c x
#pragma unroll
    for (int k = 0; k < NP; ++k) y.w1[k] = fma2(P2{c, c}, x.w1[k], y.w1[k]);
    y.b1 = fma(c, x.b1, y.b1); y.w2 = fma(c, x.w2, y.w2); y.b2 = fma(c, x.b2, y.b2);
  }
  static __device__ __forceinline__ void drift(Rec& q, T c, const Rec& im, const Rec& p) {   // q += c M^-1 p
#pragma unroll
    for (int k = 0; k < NP; ++k) q.w1[k] = fma2(P2{c, c} * im.w1[k], p.w1[k], q.w1[k]);
    q.b1 = fma(c * im.b1, p.b1, q.b1); q.w2 = fma(c * im.w2, p.w2, q.w2); q.b2 = fma(c * im.b2, p.b2, q.b2);
  }
};

template <typename T, int INMAX, int NT, int ACT, bool EXACT>
__global__ __launch_bounds__(NT, 4) void mlp1_hmc_kernel(MlpArgs<T> a, int nbch, int ldc, int cmsz, int Hp, int PS, int UG) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  typedef MlpChain<T, INMAX, NT, ACT, EXACT> Ch;
  typedef typename Ch::Rec Rec;
  Ch ch(a);
  const int tid = threadIdx.x, H = a.H, n_in = a.n_in;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  ch.tid = tid; ch.PS = PS; ch.UGL = UG * 64; ch.nbch = nbch; ch.ldc = ldc; ch.Hp = Hp;
  ch.slice = wave / UG;                         // waves beyond PS * UG have no slice: they only join the row sums
  ch.j = (wave % UG) * 64 + (tid & 63);
  ch.unit = ch.j < H && ch.slice < PS;
  ch.Xc = (typename Ch::CPtr)(uintptr_t)a.X;
  // LDS (16-byte aligned pieces first): chunk matrix (aliased by the per-thread slice-exchange records) | W2 copy | Y | delta | reduction scratch
  ch.cm = reinterpret_cast<T*>(smem_raw);
  ch.gp = ch.cm;                                // per-thread records, live only between the two passes' use of the chunk matrix
  ch.w2s = ch.cm + cmsz;
  ch.Ys = ch.w2s + Hp;
  ch.dv = ch.Ys + a.N;
  ch.red = ch.dv + nbch;
  ch.perm = reinterpret_cast<int*>(ch.red + 2 * (NT / 64) + 8);
  for (int e = tid; e < Hp; e += NT) ch.w2s[e] = (T)0;                    // W2 padding stays 0 for good; pass 1 rewrites the h padding
  for (int e = tid; e < a.N; e += NT) ch.Ys[e] = a.Y[e];
  const int j = (ch.j < H) ? ch.j : 0;     // lanes without a unit shadow unit 0 (their b2 replica must stay exact for the row sums)
  const int D = H * n_in + 2 * H + 1;
  const int o_b1 = H * n_in + j, o_w2 = H * n_in + H + j, o_b2 = H * n_in + 2 * H;
  const bool writer = ch.unit && ch.slice == 0;
#define HTA_W1(r, k) ((r).w1[(k) / 2][(k) % 2])

  Rec im, mf;       // diagonal M^-1 and sqrt(M) per parameter (1 for the identity)
#pragma unroll
  for (int k = 0; k < INMAX; ++k) {
    const bool ok = k < n_in && a.mass_kind == HTA_MASS_DIAG;
    HTA_W1(im, k) = ok ? a.inv_mass[j * n_in + k] : (T)1; HTA_W1(mf, k) = ok ? a.mass_factor[j * n_in + k] : (T)1;
  }
  const bool dg = a.mass_kind == HTA_MASS_DIAG;
  im.b1 = dg ? a.inv_mass[o_b1] : (T)1; im.w2 = dg ? a.inv_mass[o_w2] : (T)1; im.b2 = dg ? a.inv_mass[o_b2] : (T)1;
  mf.b1 = dg ? a.mass_factor[o_b1] : (T)1; mf.w2 = dg ? a.mass_factor[o_w2] : (T)1; mf.b2 = dg ? a.mass_factor[o_b2] : (T)1;

  for (int64_t c = blockIdx.x; c < a.C; c += gridDim.x) {
    const uint64_t chain = a.chain_offset + (uint64_t)c;
    const T* th0 = a.theta + c * D;
    Rec cur;
#pragma unroll
    for (int k = 0; k < INMAX; ++k) HTA_W1(cur, k) = (k < n_in) ? th0[j * n_in + k] : (T)0;
    cur.b1 = th0[o_b1]; cur.w2 = th0[o_w2]; cur.b2 = th0[o_b2];

    if (a.n_traj == 0) {          // evaluation-only: gradient and value of one split closure (parity tests)
      Rec g;
      const int lo = a.eval_split * a.Nb;
      const T ll = ch.grad_range(cur, lo, lo + a.Nb, g);
      const T lp = ll + ch.log_prior(cur) / a.prior_scale;
      if (a.grad_out && writer) {
        T* go = a.grad_out + c * D;
#pragma unroll
        for (int k = 0; k < INMAX; ++k) if (k < n_in) go[ch.j * n_in + k] = HTA_W1(g, k);
        go[o_b1] = g.b1; go[o_w2] = g.w2;
        if (tid == 0) go[o_b2] = g.b2;
      }
      if (a.logp_out && tid == 0) a.logp_out[c] = lp;
      continue;
    }

    T lp_cur = ch.logp_total(cur);
    int32_t rejected = 0;
    const T eps = a.eps, heps = (T)0.5 * a.eps;
    const int M = a.M;
    for (int t = 0; t < a.n_traj; ++t) {
      const int n = a.traj_offset + t;
      // ---- gibbs (S:185-186 / S:200-201)
      // One Philox/Box-Muller body in a rolled loop (11 inlined copies cost ~100 live VGPRs): each thread
      // parks its draws in its own record of the slice-exchange buffer, then picks them up with static indices.
      Rec p;
      {
        T* rec = ch.gp + (size_t)tid * Ch::GS;
#pragma unroll 1
        for (int k = 0; k < n_in + 3; ++k) {
          const int idx = (k < n_in) ? j * n_in + k : (k == n_in ? o_b1 : (k == n_in + 1 ? o_w2 : o_b2));
          rec[k < n_in ? k : INMAX + (k - n_in)] = normal_elem<T>(a.seed, chain, (uint32_t)n, 0, idx);
        }
#pragma unroll
        for (int k = 0; k < INMAX; ++k) HTA_W1(p, k) = (k < n_in) ? HTA_W1(mf, k) * rec[k] : (T)0;
        p.b1 = mf.b1 * rec[INMAX]; p.w2 = mf.w2 * rec[INMAX + 1]; p.b2 = mf.b2 * rec[INMAX + 2];
      }
      const T h_old = -lp_cur + ch.kinetic(p, im);                        // S:971
      Rec q = cur, g;
      // One stage loop for every integrator (a single gradient call site keeps the kernel's register budget);
      // the stage table is split_stage() in mlp.hpp.
      const int nstage = split_stage_count(a.integ, M, a.L);
      if (a.integ == HTA_SPLIT_RAND) {                                    // S:549: one subset order per trajectory
        __syncthreads();
        if (tid == 0) split_permutation(a.seed, (uint32_t)n, M, ch.perm);
        __syncthreads();
      }
      int prev_m = -1; T prev_dr = (T)1;
      for (int st = 0; st < nstage; ++st) {
        int m; T kick, dr;
        split_stage<T>(a.integ, M, a.L, st, eps, ch.perm, m, kick, dr);
        const int lo = m * a.Nb;
        if (!split_stage_reuses<T>(prev_m, prev_dr, m)) ch.grad_range(q, lo, lo + a.Nb, g);   // else: the gradient of the stage before (mlp.hpp)
        Ch::axpy(p, kick, g);
        if (dr != (T)0) Ch::drift(q, dr, im, p);
        prev_m = m; prev_dr = dr;
      }
      if (M == 1 && a.integ == HTA_SPLIT_SYMMETRIC) Ch::axpy(p, -heps, g);                                  // S:302
      const T lp_new = ch.logp_total(q);                                  // S:995
      const T h_new = -lp_new + ch.kinetic(p, im);
      const T u = u23<T>(philox_block(a.seed, chain, (uint32_t)n, PURPOSE_MH, 0, 0).x);
      const bool acc = mh_accept<T>(h_old, h_new, lp_new, u);             // S:1000-1004
      if (acc) { cur = q; lp_cur = lp_new; }
      else {
        ++rejected;
        if (n == a.burn + 1) {                                            // Q2 reset to params_init (S:1018)
          const T* ti = a.theta_init + c * D;
#pragma unroll
          for (int k = 0; k < INMAX; ++k) HTA_W1(cur, k) = (k < n_in) ? ti[j * n_in + k] : (T)0;
          cur.b1 = ti[o_b1]; cur.w2 = ti[o_w2]; cur.b2 = ti[o_b2];
          lp_cur = ch.logp_total(cur);
        }
      }
      if (a.samples && n > a.burn && writer) {
        T* row = a.samples + ((int64_t)(n - a.burn) * a.C + c) * D;
#pragma unroll
        for (int k = 0; k < INMAX; ++k) if (k < n_in) row[ch.j * n_in + k] = HTA_W1(cur, k);
        row[o_b1] = cur.b1; row[o_w2] = cur.w2;
        if (tid == 0) row[o_b2] = cur.b2;
      }
      if (tid == 0) {
        if (a.H_old) a.H_old[(int64_t)t * a.C + c] = h_old;
        if (a.H_new) a.H_new[(int64_t)t * a.C + c] = h_new;
        if (a.accept) a.accept[(int64_t)t * a.C + c] = acc ? 1 : 0;
      }
    }
    if (writer) {
      T* out = a.theta + c * D;
#pragma unroll
      for (int k = 0; k < INMAX; ++k) if (k < n_in) out[ch.j * n_in + k] = HTA_W1(cur, k);
      out[o_b1] = cur.b1; out[o_w2] = cur.w2;
      if (tid == 0) out[o_b2] = cur.b2;
    }
    if (tid == 0 && a.reject_count) a.reject_count[c] += rejected;
#if HTA_TIMING
    if (tid == 0 && blockIdx.x == 0) for (int k = 0; k < 16; ++k) hta_dbg[k] = ch.tacc[k];
#endif
    __syncthreads();
  }
#undef HTA_W1
}

template <typename T, int INMAX, int NT, int ACT, bool EXACT> int launch_mlp_act(const MlpArgs<T>& a, hipStream_t s) {
  // waves: UG groups of 64 hidden units x PS point slices
  const int UG = (a.H + 63) / 64, PS = (NT / 64) / UG;
  const int Hp = (a.H + 15) / 16 * 16;                 // row-sum lanes read float4 x 4 lanes
  int ldc = UG * 64;                                   // every lane of a unit group owns a column: unconditional stores
  while (ldc % 32 != 16) ldc += 4;                     // 4 consecutive rows land on distinct bank quarters
  typedef MlpChain<T, INMAX, NT, ACT, EXACT> Ch;
  const size_t recs = (size_t)NT * Ch::GS;             // slice-exchange / momentum-draw records, aliasing the chunk matrix
  const size_t fixed = ((size_t)Hp + a.N + 2 * (NT / 64) + 8 + 64) * sizeof(T);        // ... + 64 ints: subset order
  const size_t cap = 150 * 1024;
  HTA_REQUIRE(fixed + (recs + 4) * sizeof(T) <= cap && fixed + (size_t)4 * (ldc + 1) * sizeof(T) <= cap,
              "hta_mlp_hmc: data set (N=%d) / hidden layer (H=%d) do not fit the LDS staging", a.N, a.H);
  int nbch = (int)((cap - fixed) / ((ldc + 1) * sizeof(T)));
  if (nbch > a.Nb) nbch = a.Nb;
  if (nbch > 256) nbch = 256;
  HTA_REQUIRE(nbch >= 1, "hta_mlp_hmc: no LDS left for the chunk matrix");
  size_t cmsz = (size_t)nbch * ldc;
  if (cmsz < recs) cmsz = recs;
  const size_t lds = fixed + (cmsz + nbch) * sizeof(T);
  static DevOnce done;
  if (!done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp1_hmc_kernel<T, INMAX, NT, ACT, EXACT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { set_error("hta_mlp_hmc: hipFuncSetAttribute: %s", hipGetErrorString(e)); return HTA_ERR_LAUNCH; }
    done = true;
  }
  const int grid = (int)(a.C < 4096 ? a.C : 4096);
  profile_begin(s);
  note_route("mlp1_hmc_kernel<%s,%d,%d,%d,%s>", sizeof(T) == 4 ? "float" : "double", INMAX, NT, ACT, EXACT ? "true" : "false");
  mlp1_hmc_kernel<T, INMAX, NT, ACT, EXACT><<<grid, NT, lds, s>>>(a, nbch, ldc, (int)cmsz, Hp, PS, UG);
  profile_end(s);
  HTA_CHECK_LAUNCH("hta_mlp_hmc");
  return HTA_OK;
}

template <typename T, int INMAX, int NT, bool EXACT> int launch_mlp_x(const MlpArgs<T>& a, hipStream_t s) {
  if (a.act == 0) return launch_mlp_act<T, INMAX, NT, 0, EXACT>(a, s);
  if (a.act == 1) return launch_mlp_act<T, INMAX, NT, 1, EXACT>(a, s);
  return launch_mlp_act<T, INMAX, NT, 2, EXACT>(a, s);
}

template <typename T, int INMAX, int NT> int launch_mlp(const MlpArgs<T>& a, hipStream_t s) {
  if (a.n_in == INMAX) return launch_mlp_x<T, INMAX, NT, true>(a, s);
  return launch_mlp_x<T, INMAX, NT, false>(a, s);
}

template <typename T, int INMAX> int dispatch_ps(const MlpArgs<T>& a, hipStream_t s) {
  if (a.H <= 512) return launch_mlp<T, INMAX, 512>(a, s);
  return launch_mlp<T, INMAX, 1024>(a, s);
}

template <typename T> int mlp_hmc(const MlpArgs<T>& a, hipStream_t s) {
  HTA_REQUIRE(a.theta && a.X && a.Y && a.C > 0, "hta_mlp_hmc: NULL pointer / empty batch");
  HTA_REQUIRE(a.n_in >= 1 && a.n_in <= 32, "hta_mlp_hmc: input width %d not in [1, 32]", a.n_in);
  HTA_REQUIRE(a.H >= 1 && a.H <= 1024, "hta_mlp_hmc: hidden width %d not in [1, 1024]", a.H);
  HTA_REQUIRE(a.act >= 0 && a.act <= 2, "hta_mlp_hmc: unknown activation %d", a.act);
  HTA_REQUIRE(a.loss == HTA_LOSS_REGRESSION || a.loss == HTA_LOSS_BINARY_LOGITS, "hta_mlp_hmc: unknown loss kind %d", a.loss);
  HTA_REQUIRE(a.M >= 1 && a.Nb >= 1 && (int64_t)a.M * a.Nb <= a.N, "hta_mlp_hmc: M=%d splits of Nb=%d points exceed N=%d", a.M, a.Nb, a.N);
  HTA_REQUIRE(a.mass_kind == HTA_MASS_NONE || (a.mass_kind == HTA_MASS_DIAG && a.inv_mass && a.mass_factor),
              "hta_mlp_hmc: only identity / diagonal inv_mass are supported natively");
  if (a.n_traj > 0) HTA_REQUIRE(a.theta_init && a.L >= 0, "hta_mlp_hmc: bad trajectory arguments");
  if (a.n_traj == 0) HTA_REQUIRE(a.eval_split >= 0 && a.eval_split < a.M, "hta_mlp_logp_grad: split %d not in [0, %d)", a.eval_split, a.M);
  HTA_REQUIRE(a.integ >= HTA_SPLIT_SYMMETRIC && a.integ <= HTA_SPLIT_KMID, "hta_mlp_hmc: unknown integrator %d", a.integ);
  HTA_REQUIRE(a.integ != HTA_SPLIT_RAND || a.M <= 64, "hta_mlp_hmc: SPLITTING_RAND supports at most 64 subsets natively (M=%d)", a.M);
  HTA_REQUIRE(a.integ != HTA_SPLIT_KMID || a.M >= 2, "hta_mlp_hmc: SPLITTING_KMID needs at least 2 subsets");
  if constexpr (sizeof(T) == 4) {
    if (!g_mlp_valu && mlp_mfma_eligible(a)) return mlp_mfma(a, s);
  }
#ifdef HTA_MLP_SINGLE       // developer builds: one instantiation (the BASELINE config-4 shape), seconds to compile
  return launch_mlp_act<T, 8, 512, 0, true>(a, s);
#else
  if (a.n_in <= 4) return dispatch_ps<T, 4>(a, s);
  if (a.n_in <= 8) return dispatch_ps<T, 8>(a, s);
  if (a.n_in <= 16) return dispatch_ps<T, 16>(a, s);
  return dispatch_ps<T, 32>(a, s);
#endif
}

}  // namespace hta

extern "C" {
#define HTA_DEFINE_MLP(SUF, T)                                                                                    \
  int hta_mlp_hmc_sample_##SUF(T* theta, const T* theta_init, int64_t C, int n_in, int H, int act, int loss_kind, const T* X, \
                               const T* Y, int N, int M, int Nb, const T* tau4, T tau_out, T prior_scale,           \
                               int mass_kind, const T* inv_mass, const T* mass_factor, int integrator, int L,      \
                               T eps, int n_traj, int traj_offset, int burn, uint64_t seed, uint64_t chain_offset, \
                               T* samples,                                                                          \
                               int32_t* reject_count, T* H_old, T* H_new, uint8_t* accept, void* stream) {          \
    if (!tau4) { hta::set_error("hta_mlp_hmc_sample: tau4 is NULL (host pointer to 4 precisions)"); return HTA_ERR_INVALID; } \
    hta::MlpArgs<T> a{theta, theta_init, C, n_in, H, act, X, Y, N, M, Nb, {tau4[0], tau4[1], tau4[2], tau4[3]},     \
                      tau_out, prior_scale, mass_kind, inv_mass, mass_factor, L, eps, n_traj, traj_offset, burn,    \
                      seed, chain_offset, samples, reject_count, H_old, H_new, accept, nullptr, nullptr, 0,         \
                      integrator, loss_kind};                                                                       \
    if (n_traj <= 0) return HTA_OK;                                                                                 \
    return hta::mlp_hmc<T>(a, (hipStream_t)stream);                                                                 \
  }                                                                                                                 \
  int hta_mlp_logp_grad_##SUF(const T* theta, int64_t C, int n_in, int H, int act, int loss_kind, const T* X, const T* Y, int N, \
                              int M, int Nb, int split, const T* tau4, T tau_out, T prior_scale, T* grad_out,       \
                              T* logp_out, void* stream) {                                                          \
    if (!tau4) { hta::set_error("hta_mlp_logp_grad: tau4 is NULL"); return HTA_ERR_INVALID; }                       \
    hta::MlpArgs<T> a{const_cast<T*>(theta), nullptr, C, n_in, H, act, X, Y, N, M, Nb,                              \
                      {tau4[0], tau4[1], tau4[2], tau4[3]}, tau_out, prior_scale, HTA_MASS_NONE, nullptr, nullptr,  \
                      0, (T)0, 0, 0, 0, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, grad_out, logp_out,      \
                      split, HTA_SPLIT_SYMMETRIC, loss_kind};                                                       \
    return hta::mlp_hmc<T>(a, (hipStream_t)stream);                                                                 \
  }
HTA_DEFINE_MLP(f32, float)
HTA_DEFINE_MLP(f64, double)
}
