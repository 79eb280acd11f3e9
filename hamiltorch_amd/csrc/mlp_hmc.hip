// Fused (split-)HMC for a one-hidden-layer Bayesian MLP with a Gaussian (regression) likelihood:
// BASELINE config 4.  One workgroup per chain; the whole trajectory loop of sample()
// (hamiltorch/samplers.py:965-1026) runs on-chip with
//   log p_m(theta) = -1/2 tau_out sum_{i in split m} (f(x_i) - y_i)^2
//                    + (1/prior_scale) sum_layers Normal(0, tau_l^-1/2).log_prob(w).sum()
// (define_model_log_prob, S:1141-1199; one closure per DataLoader batch, S:1251-1255) and either
//   * the symmetric split integrator (S:499-540): 2M gradient half-kicks + 2(M-1) drifts per step, or
//   * plain leapfrog over the full data (S:281-302) when M == 1 (sample_model).
//
// Work layout.  theta = [W1[H,in] | b1[H] | W2[1,H] | b2[1]] (U:121-122 order).  Thread (j, s):
// hidden unit j, point slice s of PS; the unit's weights, momentum and gradient live in its
// registers (replicated over the PS slice lanes, which stay bit-identical).  A gradient of one data
// chunk is two passes over the chunk's points:
//   1. h_ij = act(b1_j + W1_j . x_i),  c_ij = W2_j h_ij  -> LDS [i][j];  row sums give f(x_i), the
//      residual and delta_i = -tau_out (f(x_i) - y_i)  (LDS vector);
//   2. every thread re-forms h_ij for its points and accumulates dW2_j, db1_j, dW1_j. (in-register
//      FMAs, x_i broadcast from LDS), then a 2-step butterfly over the PS slice lanes.
// The data set (shared by all chains) is staged once per workgroup in LDS.  HBM traffic per
// trajectory: one [D] sample row per chain.
#include "common.hpp"
#include "philox.hpp"

namespace hta {

void profile_begin(hipStream_t s);
void profile_end(hipStream_t s);

template <typename T> struct MlpArgs {
  T* theta; const T* theta_init; int64_t C;
  int n_in; int H; int act;
  const T* X; const T* Y; int N;
  int M; int Nb;
  T tau[4]; T tau_out; T prior_scale;
  int mass_kind; const T* inv_mass; const T* mass_factor;
  int L; T eps; int n_traj; int traj_offset; int burn;
  uint64_t seed; uint64_t chain_offset;
  T* samples; int32_t* reject_count; T* H_old; T* H_new; uint8_t* accept;
  T* grad_out; T* logp_out;   // evaluation-only mode (n_traj == 0): d log p_m / d theta [C, D] and log p_m [C] of split `eval_split`
  int eval_split;
};

template <int ACT, typename T> __device__ __forceinline__ T act_fn(T z, T& dz) {
  if (ACT == 0) { dz = z > (T)0 ? (T)1 : (T)0; return z > (T)0 ? z : (T)0; }
  if (ACT == 1) { const T h = tanh(z); dz = (T)1 - h * h; return h; }
  const T h = (T)1 / ((T)1 + exp(-z)); dz = h * ((T)1 - h); return h;
}

template <int G, typename T> __device__ __forceinline__ T slice_sum(T v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <typename T, int INMAX, int PS, int NT, int ACT>
struct MlpChain {
  // per-thread parameter record of hidden unit j
  struct Rec { T w1[INMAX]; T b1; T w2; T b2; };
  const MlpArgs<T>& a;
  T* Xs; T* Ys; T* cm; T* dv; T* red;
  int nbch, ldc, j, s, tid;
  bool unit;      // j < H
  __device__ MlpChain(const MlpArgs<T>& a_) : a(a_) {}

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

  // pass 1 over points [lo, lo+cnt): delta_i -> dv[i]; returns (sum r_i^2, sum delta_i)
  __device__ __forceinline__ T forward_chunk(const Rec& w, int lo, int cnt, T& sum_delta) {
    __syncthreads();
    if (unit) {
      for (int i = s; i < cnt; i += PS) {
        const T* x = Xs + (lo + i) * INMAX;          // rows zero-padded to INMAX: no guards in the hot loops
        T z = w.b1;
#pragma unroll
        for (int k = 0; k < INMAX; ++k) z = fma(w.w1[k], x[k], z);
        T dz;
        const T h = act_fn<ACT, T>(z, dz);
        cm[i * ldc + j] = w.w2 * h;
      }
    }
    __syncthreads();
    T sse = 0, sd = 0;
    constexpr int RS = 4;                      // lanes per row of the chunk matrix
    for (int i = tid / RS; i < cnt; i += NT / RS) {
      const int part = tid % RS;
      T out = 0;
      const T* row = cm + i * ldc;
      for (int jj = part; jj < a.H; jj += RS) out += row[jj];
      out = slice_sum<RS>(out) + w.b2;
      if (part == 0) {
        const T r = out - Ys[lo + i];
        const T d = -a.tau_out * r;
        dv[i] = d;
        sse += r * r; sd += d;
      }
    }
    sse = block_sum2(sse, sd);
    sum_delta = sd;
    return sse;
  }

  // pass 2: accumulate the likelihood gradient of points [lo, lo+cnt) into g (this thread's slice)
  __device__ __forceinline__ void backward_chunk(const Rec& w, int lo, int cnt, Rec& g) {
    if (unit) {
      for (int i = s; i < cnt; i += PS) {
        const T* x = Xs + (lo + i) * INMAX;
        T xv[INMAX];
#pragma unroll
        for (int k = 0; k < INMAX; ++k) xv[k] = x[k];
        T z = w.b1;
#pragma unroll
        for (int k = 0; k < INMAX; ++k) z = fma(w.w1[k], xv[k], z);
        T dz;
        const T h = act_fn<ACT, T>(z, dz);
        const T d = dv[i];
        g.w2 = fma(d, h, g.w2);
        const T dh = d * w.w2 * dz;
        g.b1 += dh;
#pragma unroll
        for (int k = 0; k < INMAX; ++k) g.w1[k] = fma(dh, xv[k], g.w1[k]);
      }
    }
  }

  // d log p_m / d theta over points [lo, hi) + prior / prior_scale; returns log-likelihood part
  __device__ __forceinline__ T grad_range(const Rec& w, int lo, int hi, Rec& g) {
#pragma unroll
    for (int k = 0; k < INMAX; ++k) g.w1[k] = 0;
    g.b1 = 0; g.w2 = 0; g.b2 = 0;
    T sse = 0;
    for (int c0 = lo; c0 < hi; c0 += nbch) {
      const int cnt = min(nbch, hi - c0);
      T sd;
      sse += forward_chunk(w, c0, cnt, sd);
      g.b2 += sd;
      backward_chunk(w, c0, cnt, g);
    }
#pragma unroll
    for (int k = 0; k < INMAX; ++k) g.w1[k] = slice_sum<PS>(g.w1[k]);
    g.b1 = slice_sum<PS>(g.b1); g.w2 = slice_sum<PS>(g.w2);
    const T ips = (T)1 / a.prior_scale;
#pragma unroll
    for (int k = 0; k < INMAX; ++k) g.w1[k] -= ips * a.tau[0] * w.w1[k];     // S:1156: d/dw Normal(0, tau^-1/2).log_prob
    g.b1 -= ips * a.tau[1] * w.b1; g.w2 -= ips * a.tau[2] * w.w2; g.b2 -= ips * a.tau[3] * w.b2;
    return (T)-0.5 * a.tau_out * sse;
  }

  // prior log-density (whole, not divided): sum_l [ -1/2 tau_l sum w^2 + n_l (1/2 log tau_l - 1/2 log 2 pi) ]
  __device__ __forceinline__ T log_prior(const Rec& w) {
    T q = 0, dummy = 0;
    if (unit && s == 0) {
      T sw = 0;
#pragma unroll
      for (int k = 0; k < INMAX; ++k) sw += w.w1[k] * w.w1[k];           // padding weights are exactly 0
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
    if (unit && s == 0) {
#pragma unroll
      for (int q = 0; q < INMAX; ++q) k += p.w1[q] * im.w1[q] * p.w1[q];   // padding momenta are exactly 0
      k += p.b1 * im.b1 * p.b1 + p.w2 * im.w2 * p.w2;
    }
    if (tid == 0) k += p.b2 * im.b2 * p.b2;
    return (T)0.5 * block_sum2(k, dummy);
  }

  static __device__ __forceinline__ void axpy(Rec& y, T c, const Rec& x) {       // y += c x
#pragma unroll
    for (int k = 0; k < INMAX; ++k) y.w1[k] = fma(c, x.w1[k], y.w1[k]);
    y.b1 = fma(c, x.b1, y.b1); y.w2 = fma(c, x.w2, y.w2); y.b2 = fma(c, x.b2, y.b2);
  }
  static __device__ __forceinline__ void drift(Rec& q, T c, const Rec& im, const Rec& p) {   // q += c M^-1 p
#pragma unroll
    for (int k = 0; k < INMAX; ++k) q.w1[k] = fma(c * im.w1[k], p.w1[k], q.w1[k]);
    q.b1 = fma(c * im.b1, p.b1, q.b1); q.w2 = fma(c * im.w2, p.w2, q.w2); q.b2 = fma(c * im.b2, p.b2, q.b2);
  }
};

template <typename T, int INMAX, int PS, int NT, int ACT>
__global__ __launch_bounds__(NT, 4) void mlp1_hmc_kernel(MlpArgs<T> a, int nbch, int ldc) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  typedef MlpChain<T, INMAX, PS, NT, ACT> Ch;
  typedef typename Ch::Rec Rec;
  Ch ch(a);
  const int tid = threadIdx.x, H = a.H, n_in = a.n_in;
  ch.tid = tid; ch.j = tid / PS; ch.s = tid % PS; ch.unit = ch.j < H; ch.nbch = nbch; ch.ldc = ldc;
  ch.Xs = reinterpret_cast<T*>(smem_raw);
  ch.Ys = ch.Xs + a.N * INMAX;
  ch.cm = ch.Ys + a.N;
  ch.dv = ch.cm + nbch * ldc;
  ch.red = ch.dv + nbch;
  for (int e = tid; e < a.N * INMAX; e += NT) {
    const int i = e / INMAX, k = e - i * INMAX;
    ch.Xs[e] = (k < n_in) ? a.X[i * n_in + k] : (T)0;
  }
  for (int e = tid; e < a.N; e += NT) ch.Ys[e] = a.Y[e];
  const int j = ch.unit ? ch.j : 0;
  const int D = H * n_in + 2 * H + 1;
  const int o_b1 = H * n_in + j, o_w2 = H * n_in + H + j, o_b2 = H * n_in + 2 * H;
  const bool writer = ch.unit && ch.s == 0;

  Rec im, mf;       // diagonal M^-1 and sqrt(M) per parameter (1 for the identity)
#pragma unroll
  for (int k = 0; k < INMAX; ++k) {
    const bool ok = k < n_in && a.mass_kind == HTA_MASS_DIAG;
    im.w1[k] = ok ? a.inv_mass[j * n_in + k] : (T)1; mf.w1[k] = ok ? a.mass_factor[j * n_in + k] : (T)1;
  }
  const bool dg = a.mass_kind == HTA_MASS_DIAG;
  im.b1 = dg ? a.inv_mass[o_b1] : (T)1; im.w2 = dg ? a.inv_mass[o_w2] : (T)1; im.b2 = dg ? a.inv_mass[o_b2] : (T)1;
  mf.b1 = dg ? a.mass_factor[o_b1] : (T)1; mf.w2 = dg ? a.mass_factor[o_w2] : (T)1; mf.b2 = dg ? a.mass_factor[o_b2] : (T)1;

  for (int64_t c = blockIdx.x; c < a.C; c += gridDim.x) {
    const uint64_t chain = a.chain_offset + (uint64_t)c;
    const T* th0 = a.theta + c * D;
    Rec cur;
#pragma unroll
    for (int k = 0; k < INMAX; ++k) cur.w1[k] = (k < n_in) ? th0[j * n_in + k] : (T)0;
    cur.b1 = th0[o_b1]; cur.w2 = th0[o_w2]; cur.b2 = th0[o_b2];

    if (a.n_traj == 0) {          // evaluation-only: gradient and value of one split closure (parity tests)
      Rec g;
      const int lo = a.eval_split * a.Nb;
      const T ll = ch.grad_range(cur, lo, lo + a.Nb, g);
      const T lp = ll + ch.log_prior(cur) / a.prior_scale;
      if (a.grad_out && writer) {
        T* go = a.grad_out + c * D;
#pragma unroll
        for (int k = 0; k < INMAX; ++k) if (k < n_in) go[ch.j * n_in + k] = g.w1[k];
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
      Rec p;
#pragma unroll
      for (int k = 0; k < INMAX; ++k)
        p.w1[k] = (k < n_in) ? mf.w1[k] * normal_elem<T>(a.seed, chain, (uint32_t)n, 0, j * n_in + k) : (T)0;
      p.b1 = mf.b1 * normal_elem<T>(a.seed, chain, (uint32_t)n, 0, o_b1);
      p.w2 = mf.w2 * normal_elem<T>(a.seed, chain, (uint32_t)n, 0, o_w2);
      p.b2 = mf.b2 * normal_elem<T>(a.seed, chain, (uint32_t)n, 0, o_b2);
      const T h_old = -lp_cur + ch.kinetic(p, im);                        // S:971
      Rec q = cur, g;
      // One stage loop for both integrators (a single gradient call site keeps the kernel's register budget):
      //   plain leapfrog (M == 1, S:281-302): stage 0 kicks eps/2, stages 1..L kick eps, drifts of eps in between,
      //     and the final half kick is taken back afterwards, in the reference's order;
      //   symmetric split (S:499-540): per step 2M stages m = 0..M-1, M-1..0, each a half kick, with a drift of
      //     eps / (2 (M-1)) after every stage except the two turning points.
      const T dq = (M > 1) ? eps / (T)((M - 1) * 2) : (T)0;
      const int nstage = (M == 1) ? a.L + 1 : a.L * 2 * M;
      for (int st = 0; st < nstage; ++st) {
        int lo; T kick, dr;
        if (M == 1) { lo = 0; kick = (st == 0) ? heps : eps; dr = (st < a.L) ? eps : (T)0; }
        else {
          const int s2 = st % (2 * M);
          const int m = (s2 < M) ? s2 : 2 * M - 1 - s2;
          lo = m * a.Nb; kick = heps; dr = (s2 == M - 1 || s2 == 2 * M - 1) ? (T)0 : dq;
        }
        ch.grad_range(q, lo, lo + a.Nb, g);
        Ch::axpy(p, kick, g);
        if (dr != (T)0) Ch::drift(q, dr, im, p);
      }
      if (M == 1) Ch::axpy(p, -heps, g);                                  // S:302
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
          for (int k = 0; k < INMAX; ++k) cur.w1[k] = (k < n_in) ? ti[j * n_in + k] : (T)0;
          cur.b1 = ti[o_b1]; cur.w2 = ti[o_w2]; cur.b2 = ti[o_b2];
          lp_cur = ch.logp_total(cur);
        }
      }
      if (a.samples && n > a.burn && writer) {
        T* row = a.samples + ((int64_t)(n - a.burn) * a.C + c) * D;
#pragma unroll
        for (int k = 0; k < INMAX; ++k) if (k < n_in) row[ch.j * n_in + k] = cur.w1[k];
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
      for (int k = 0; k < INMAX; ++k) if (k < n_in) out[ch.j * n_in + k] = cur.w1[k];
      out[o_b1] = cur.b1; out[o_w2] = cur.w2;
      if (tid == 0) out[o_b2] = cur.b2;
    }
    if (tid == 0 && a.reject_count) a.reject_count[c] += rejected;
    __syncthreads();
  }
}

template <typename T, int INMAX, int PS, int NT, int ACT> int launch_mlp_act(const MlpArgs<T>& a, hipStream_t s) {
  // LDS: data set + chunk matrix [nbch][ldc] + delta vector + reduction scratch
  const int ldc = a.H | 1;
  const size_t fixed = ((size_t)a.N * (INMAX + 1) + 2 * (NT / 64) + 8) * sizeof(T);
  HTA_REQUIRE(fixed + (size_t)8 * (ldc + 1) * sizeof(T) <= 150 * 1024, "hta_mlp_hmc: data set (N=%d, in=%d) does not fit the LDS staging", a.N, a.n_in);
  int nbch = (int)((150 * 1024 - fixed) / ((ldc + 1) * sizeof(T)));
  const int need = a.Nb;
  if (nbch > need) nbch = need;
  if (nbch > 256) nbch = 256;
  HTA_REQUIRE(nbch >= 1, "hta_mlp_hmc: no LDS left for the chunk matrix");
  const size_t lds = fixed + (size_t)nbch * (ldc + 1) * sizeof(T);
  static bool done = false;
  if (!done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp1_hmc_kernel<T, INMAX, PS, NT, ACT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { set_error("hta_mlp_hmc: hipFuncSetAttribute: %s", hipGetErrorString(e)); return HTA_ERR_LAUNCH; }
    done = true;
  }
  const int grid = (int)(a.C < 4096 ? a.C : 4096);
  profile_begin(s);
  mlp1_hmc_kernel<T, INMAX, PS, NT, ACT><<<grid, NT, lds, s>>>(a, nbch, ldc);
  profile_end(s);
  HTA_CHECK_LAUNCH("hta_mlp_hmc");
  return HTA_OK;
}

template <typename T, int INMAX, int PS, int NT> int launch_mlp(const MlpArgs<T>& a, hipStream_t s) {
  if (a.act == 0) return launch_mlp_act<T, INMAX, PS, NT, 0>(a, s);
  if (a.act == 1) return launch_mlp_act<T, INMAX, PS, NT, 1>(a, s);
  return launch_mlp_act<T, INMAX, PS, NT, 2>(a, s);
}

template <typename T, int INMAX> int dispatch_ps(const MlpArgs<T>& a, hipStream_t s) {
  const int H = a.H;
  if (H * 8 <= 512) return launch_mlp<T, INMAX, 8, 512>(a, s);
  if (H * 4 <= 512) return launch_mlp<T, INMAX, 4, 512>(a, s);
  if (H * 2 <= 512) return launch_mlp<T, INMAX, 2, 512>(a, s);
  if (H <= 512) return launch_mlp<T, INMAX, 1, 512>(a, s);
  return launch_mlp<T, INMAX, 1, 1024>(a, s);
}

template <typename T> int mlp_hmc(const MlpArgs<T>& a, hipStream_t s) {
  HTA_REQUIRE(a.theta && a.X && a.Y && a.C > 0, "hta_mlp_hmc: NULL pointer / empty batch");
  HTA_REQUIRE(a.n_in >= 1 && a.n_in <= 32, "hta_mlp_hmc: input width %d not in [1, 32]", a.n_in);
  HTA_REQUIRE(a.H >= 1 && a.H <= 1024, "hta_mlp_hmc: hidden width %d not in [1, 1024]", a.H);
  HTA_REQUIRE(a.act >= 0 && a.act <= 2, "hta_mlp_hmc: unknown activation %d", a.act);
  HTA_REQUIRE(a.M >= 1 && a.Nb >= 1 && (int64_t)a.M * a.Nb <= a.N, "hta_mlp_hmc: M=%d splits of Nb=%d points exceed N=%d", a.M, a.Nb, a.N);
  HTA_REQUIRE(a.mass_kind == HTA_MASS_NONE || (a.mass_kind == HTA_MASS_DIAG && a.inv_mass && a.mass_factor),
              "hta_mlp_hmc: only identity / diagonal inv_mass are supported natively");
  if (a.n_traj > 0) HTA_REQUIRE(a.theta_init && a.L >= 0, "hta_mlp_hmc: bad trajectory arguments");
  if (a.n_in <= 4) return dispatch_ps<T, 4>(a, s);
  if (a.n_in <= 8) return dispatch_ps<T, 8>(a, s);
  if (a.n_in <= 16) return dispatch_ps<T, 16>(a, s);
  return dispatch_ps<T, 32>(a, s);
}

}  // namespace hta

extern "C" {
#define HTA_DEFINE_MLP(SUF, T)                                                                                    \
  int hta_mlp_hmc_sample_##SUF(T* theta, const T* theta_init, int64_t C, int n_in, int H, int act, const T* X,     \
                               const T* Y, int N, int M, int Nb, const T* tau4, T tau_out, T prior_scale,           \
                               int mass_kind, const T* inv_mass, const T* mass_factor, int L, T eps, int n_traj,    \
                               int traj_offset, int burn, uint64_t seed, uint64_t chain_offset, T* samples,         \
                               int32_t* reject_count, T* H_old, T* H_new, uint8_t* accept, void* stream) {          \
    if (!tau4) { hta::set_error("hta_mlp_hmc_sample: tau4 is NULL (host pointer to 4 precisions)"); return HTA_ERR_INVALID; } \
    hta::MlpArgs<T> a{theta, theta_init, C, n_in, H, act, X, Y, N, M, Nb, {tau4[0], tau4[1], tau4[2], tau4[3]},     \
                      tau_out, prior_scale, mass_kind, inv_mass, mass_factor, L, eps, n_traj, traj_offset, burn,    \
                      seed, chain_offset, samples, reject_count, H_old, H_new, accept, nullptr, nullptr, 0};        \
    if (n_traj <= 0) return HTA_OK;                                                                                 \
    return hta::mlp_hmc<T>(a, (hipStream_t)stream);                                                                 \
  }                                                                                                                 \
  int hta_mlp_logp_grad_##SUF(const T* theta, int64_t C, int n_in, int H, int act, const T* X, const T* Y, int N,   \
                              int M, int Nb, int split, const T* tau4, T tau_out, T prior_scale, T* grad_out,       \
                              T* logp_out, void* stream) {                                                          \
    if (!tau4) { hta::set_error("hta_mlp_logp_grad: tau4 is NULL"); return HTA_ERR_INVALID; }                       \
    hta::MlpArgs<T> a{const_cast<T*>(theta), nullptr, C, n_in, H, act, X, Y, N, M, Nb,                              \
                      {tau4[0], tau4[1], tau4[2], tau4[3]}, tau_out, prior_scale, HTA_MASS_NONE, nullptr, nullptr,  \
                      0, (T)0, 0, 0, 0, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, grad_out, logp_out,      \
                      split};                                                                                       \
    return hta::mlp_hmc<T>(a, (hipStream_t)stream);                                                                 \
  }
HTA_DEFINE_MLP(f32, float)
HTA_DEFINE_MLP(f64, double)
}
