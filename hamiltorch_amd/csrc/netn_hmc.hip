// Fused (split-)HMC for SMALL fully connected Bayesian networks of any depth: 1 .. 4 Linear layers (none to three hidden ones),
// one activation kind between them, Gaussian / Bernoulli-with-logits / softmax-cross-entropy likelihood.
// Same contract as mlp_hmc.hip (hamiltorch/samplers.py:965-1026 trajectory loop, S:1141-1199 closures, S:499-596 split
// integrators, S:281-302 leapfrog); what it adds are the shapes the reference's notebooks actually sample: the two-hidden-layer
// regression net of the split-HMC notebook (1-10-10-1), the softmax regression of the BNN notebook (Linear(4, 3),
// `multi_class_linear_output`, the default `model_loss` of sample_model) and small classifiers with several outputs.
//
// Work layout.  These nets have tens to a few hundred parameters and layers 1 .. 32 wide: nothing for a matrix core to hold on
// to, but a data set of hundreds of points.  ONE WAVE PER CHAIN, lanes = points: a lane carries its point through the layers
// (activations and deltas of the point in this lane's column of two small LDS matrices, [unit][lane]: conflict-free), the
// parameters of the chain are a broadcast operand (one LDS copy per wave, every lane reads the same address).  A weight's
// gradient sum_p delta[p][o] a[p][i] is a sum of outer products over the points: in fp32 it runs on the matrix cores
// (v_mfma_f32_4x4x1_16b_f32: the weight matrices of all layers, bias columns included, cut into 4 x 4 blocks, 16 blocks per
// instruction - one or a few instructions per point, blocks accumulated in registers over a whole pass; pass()), in fp64 it
// is a reduction over the lanes per weight.  Leapfrog state (q, p, gradient, masses) lives in registers, parameter
// 64 k + lane in register k of lane `lane`.  One wave per workgroup: barriers cost nothing, 1024 chains put a wave on every SIMD.
// How it got here (tools/scratch/netn_time.py, profiles/r02z_netn_speed.txt; Net([1,10,10,1]), 400 points, 1024 chains):
// 3.6e6 chain-steps/s with one point per lane and a DPP reduction per weight -> 5.4e6 (interleaved reductions, plain ds_add) ->
// 8.6e6 (several points per lane, blocks of four units) -> 1.46e7 (gradient on the matrix cores); the callback path: 6.8e5.
// Tried and dropped: the forward products and the delta propagation as matrix instructions as well ((4 points) x (4 units)
// blocks, the lane's own activation as the A operand, the result written back transposed inside the quad): 1.16e7 - its
// per-input loop with run-time layer shapes issues more than the blocked FMA loops it replaces.
// Layer shapes are run-time values (uniform loops); limits: D <= 512 parameters, widths <= 64, N <= what L2 holds (X is read
// from global memory, coalesced over the lanes).
#include "mlp.hpp"
#include "netn.hpp"
#include "philox.hpp"

#ifndef NETN_TIMING
#define NETN_TIMING 0   // developer cycle counters per phase of a pass (block 0): tools/scratch/netn_time.py prints them
#endif
#if NETN_TIMING
__device__ unsigned long long hta_netn_dbg[8];
extern "C" void hta_netn_dbg_read(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(hta_netn_dbg), sizeof(hta_netn_dbg)); }
#define NETN_TICK(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); tacc[k] += now_ - tlast; tlast = now_; } while (0)
#else
#define NETN_TICK(k) do {} while (0)
#endif

namespace hta {

void profile_begin(hipStream_t s);
void profile_end(hipStream_t s);
extern int g_netn_waves;                // tuning key "netn_waves" (default 1)

template <int CTRL, int ROWMASK> __device__ __forceinline__ float dpp_add(float v) {
  const int o = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xF, false);
  return v + __builtin_bit_cast(float, o);
}
// sum over the 64 lanes, uniform result (SGPR): quad butterflies, two row rotations, then the row totals hop to lane 63
__device__ __forceinline__ float wave_total(float v) {
  v = dpp_add<0xB1, 0xF>(v);           // quad_perm:[1,0,3,2]
  v = dpp_add<0x4E, 0xF>(v);           // quad_perm:[2,3,0,1]
  v = dpp_add<0x124, 0xF>(v);          // row_ror:4
  v = dpp_add<0x128, 0xF>(v);          // row_ror:8   -> every lane holds its row's total
  v = dpp_add<0x142, 0xA>(v);          // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xC>(v);          // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ double wave_total(double v) { return wave_sum(v); }
// four sums at once, step by step: the four chains of dependent DPP adds fill each other's wait states
__device__ __forceinline__ void wave_total4(float (&v)[4]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = dpp_add<0xB1, 0xF>(v[k]);
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = dpp_add<0x4E, 0xF>(v[k]);
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = dpp_add<0x124, 0xF>(v[k]);
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = dpp_add<0x128, 0xF>(v[k]);
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = dpp_add<0x142, 0xA>(v[k]);
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = dpp_add<0x143, 0xC>(v[k]);
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[k]), 63));
}
__device__ __forceinline__ void wave_total4(double (&v)[4]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = wave_sum(v[k]);
}
// *p += v by the calling lane(s), p in LDS: the plain LDS add (atomicAdd on a wave-uniform value makes the compiler count the
// active lanes and scale the value first - a dozen instructions where one is meant)
__device__ __forceinline__ void lds_add(float* p, float v) {
  const uint32_t off = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)p;
  asm volatile("ds_add_f32 %0, %1" : : "v"(off), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_add(double* p, double v) {
  const uint32_t off = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double*)p;
  asm volatile("ds_add_f64 %0, %1" : : "v"(off), "v"(v) : "memory");
}

template <typename T> __device__ __forceinline__ T netn_act(int act, T z) {
  if (act == 0) return z > (T)0 ? z : (T)0;
  if (act == 1) return tanh(z);
  return (T)1 / ((T)1 + exp(-z));
}
template <typename T> __device__ __forceinline__ T netn_act_deriv(int act, T h) {      // in terms of the activation's value
  if (act == 0) return h > (T)0 ? (T)1 : (T)0;
  if (act == 1) return (T)1 - h * h;
  return h * ((T)1 - h);
}

// PB: points per lane and sweep (a sweep covers 64 PB points).  More points per lane = PB independent FMA chains that share
// every weight operand, and PB points folded into a lane's partial product BEFORE the reduction over the lanes: a weight costs
// one reduction per 64 PB points.  The launcher takes the smallest PB that covers a pass's points in one or two sweeps.
// WV: waves per chain (workgroup = 64 WV threads).  Every wave keeps the whole leapfrog state (duplicates, as cheap as
// idle lanes) and takes every WV-th sweep of a pass with its own activation / delta matrices; the waves share the parameter copy
// and the gradient vector (LDS adds) and exchange their likelihood sums through LDS.  At 1024 chains one wave per chain is one
// wave per SIMD with nothing to hide its LDS round trips behind; WV = 2 doubles the waves on the chip.
template <typename T, int PB, int WV>
struct NetChain {
  struct Rec { T v[NETN_KMAX]; };
  // floats per row of the activation / delta matrices [unit][point column]: odd, so that a lane's own column (stride 1 over
  // the lanes) AND one column read across the rows (the matrix-core gradient below) are both free of bank conflicts
  static constexpr int RS = 64 * PB + 1;
  static constexpr int SWEEP = 64 * PB;  // points per sweep
  static constexpr bool MATRIX_GRAD = sizeof(T) == 4;
  const NetArgs<T>& a;
  int lane, wave, D, nl, n_out, out_row;
  T* esh;                                // [2] likelihood sums of the workgroup's waves (alternating slots)
  int eslot;
  int woff[NETN_MAX_LAYERS], boff[NETN_MAX_LAYERS], aoff[NETN_MAX_LAYERS + 1];
  T *th, *gacc, *act, *dmat;             // dmat: the deltas of EVERY layer, same row numbering as act (+ one row of zeros)
  int* perm;
  // the matrix-core gradient: per instruction set, this lane's operand rows and what its accumulator block is (see grad_blocks)
  int n_set, rowA[NETN_NSET], rowB[NETN_NSET], desc[NETN_NSET], ones_row, zero_row;
  typedef float acc4 __attribute__((ext_vector_type(4)));
  acc4 gblk[NETN_NSET];
  Rec tauv;                              // prior precision of every parameter this lane owns (0 beyond D)
  T prior_const;                         // sum_t n_t (1/2 log tau_t - 1/2 log 2 pi)
#if NETN_TIMING
  unsigned long long tacc[8] = {0}, tlast = 0;
#endif
  __device__ NetChain(const NetArgs<T>& a_) : a(a_) {}

  // LDS traffic between the lanes of ONE wave (its own matrices): ordered by waiting for the wave's outstanding LDS operations;
  // no workgroup barrier - the waves of a chain run different numbers of sweeps
  static __device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  __device__ __forceinline__ void publish(const Rec& q) {     // the chain's copy of the parameters; gradient vector cleared
    __syncthreads();                                           // (every wave holds the same q: wave 0 writes)
    if (wave == 0) {
#pragma unroll
      for (int k = 0; k < NETN_KMAX; ++k) {
        const int pid = 64 * k + lane;
        if (pid < D) { th[pid] = q.v[k]; gacc[pid] = (T)0; }
      }
      if (lane == 0) esh[eslot] = (T)0;                        // this pass's slot of the likelihood sums (slots alternate)
    }
    __syncthreads();
  }

  // this lane's PB points (p0 + 64 b + lane) through the layers; the outputs sit in rows out_row .. of `act`
  __device__ __forceinline__ void forward(int p0, int hi) {
    const int n_in = a.dims[0];
    T* Al = act + lane;
    for (int i = 0; i < n_in; ++i)
#pragma unroll
      for (int b = 0; b < PB; ++b) {
        const int p = p0 + 64 * b + lane;
        Al[i * RS + 64 * b] = p < hi ? a.X[(int64_t)p * n_in + i] : (T)0;
      }
    NETN_TICK(1);
#pragma unroll
    for (int l = 0; l < NETN_MAX_LAYERS; ++l) {             // (unrolled with a guard: the per-layer tables stay in scalar registers)
      if (l >= nl) break;
      const int I = a.dims[l], O = a.dims[l + 1];
      const T* W = th + woff[l];
      const T* Ar = Al + aoff[l] * RS;
      T* Ao = Al + aoff[l + 1] * RS;
      // One wave per SIMD (at 1024 chains) and chains of dependent FMAs: an iteration costs an LDS round trip whatever it
      // computes, so an iteration carries a block of FOUR output units x PB points = 4 PB independent chains fed by 4 + PB
      // operand reads (measured before the blocking: 76 clocks per FMA - tools/scratch/netn_time.py).  A unit's sum keeps its
      // order i = 0, 1, 2, ...
      int o = 0;
      for (; o + 4 <= O; o += 4) {
        T acc[4][PB];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const T bias = th[boff[l] + o + u];
#pragma unroll
          for (int b = 0; b < PB; ++b) acc[u][b] = bias;
        }
        const T* Wr = W + o * I;
        for (int i = 0; i < I; ++i) {
          T w[4], av[PB];
#pragma unroll
          for (int u = 0; u < 4; ++u) w[u] = Wr[u * I + i];
#pragma unroll
          for (int b = 0; b < PB; ++b) av[b] = Ar[i * RS + 64 * b];
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int b = 0; b < PB; ++b) acc[u][b] = fma(w[u], av[b], acc[u][b]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int b = 0; b < PB; ++b) Ao[(o + u) * RS + 64 * b] = (l + 1 < nl) ? netn_act<T>(a.act, acc[u][b]) : acc[u][b];
      }
      for (; o < O; ++o) {
        T acc[PB];
        const T bias = th[boff[l] + o];
#pragma unroll
        for (int b = 0; b < PB; ++b) acc[b] = bias;
        const T* Wr = W + o * I;
        int i = 0;
        for (; i + 2 <= I; i += 2) {
          const T w0 = Wr[i], w1 = Wr[i + 1];
          T a0[PB], a1[PB];
#pragma unroll
          for (int b = 0; b < PB; ++b) { a0[b] = Ar[i * RS + 64 * b]; a1[b] = Ar[(i + 1) * RS + 64 * b]; }
#pragma unroll
          for (int b = 0; b < PB; ++b) { acc[b] = fma(w0, a0[b], acc[b]); acc[b] = fma(w1, a1[b], acc[b]); }
        }
        for (; i < I; ++i) {
          const T w0 = Wr[i];
#pragma unroll
          for (int b = 0; b < PB; ++b) acc[b] = fma(w0, Ar[i * RS + 64 * b], acc[b]);
        }
#pragma unroll
        for (int b = 0; b < PB; ++b) Ao[o * RS + 64 * b] = (l + 1 < nl) ? netn_act<T>(a.act, acc[b]) : acc[b];
      }
    }
  }

  // likelihood of one of this lane's points (S:1170-1184): e with log-lik = -1/2 tau_out e, and d log-lik / d output into dl
  __device__ __forceinline__ T point_loss(int p, bool valid, const T* Ao, T* dl, bool want_delta) {
    const int O = n_out;
    T e = 0;
    if (a.loss == HTA_LOSS_SOFTMAX_CE) {                     // S:1173-1178: CrossEntropyLoss(reduction='sum') on logits, integer labels
      const int y = valid ? (int)a.Y[p] : 0;
      T m = Ao[0];
      for (int o = 1; o < O; ++o) m = fmax(m, Ao[o * RS]);
      T s = 0;
      for (int o = 0; o < O; ++o) s += exp(Ao[o * RS] - m);
      const T lse = m + log(s);
      e = (T)2 * (lse - Ao[(y >= 0 && y < O ? y : 0) * RS]);
      if (want_delta)
        for (int o = 0; o < O; ++o) dl[o * RS] = valid ? -a.tau_out * (exp(Ao[o * RS] - lse) - (o == y ? (T)1 : (T)0)) : (T)0;
    } else {
      for (int o = 0; o < O; ++o) {
        const T f = Ao[o * RS], y = valid ? a.Y[(int64_t)p * O + o] : (T)0;
        T d, ee;
        mlp_point_loss<T>(a.loss, f, y, a.tau_out, d, ee);
        e += ee;
        if (want_delta) dl[o * RS] = valid ? d : (T)0;
      }
    }
    return valid ? e : (T)0;
  }

  // d log-lik / d theta of split points [lo, hi) summed into gacc (through publish()'s zero), returns sum of e over the points
  __device__ __forceinline__ T pass(const Rec& q, int lo, int hi, bool grad) {
    NETN_TICK(6);
    publish(q);
    NETN_TICK(0);
    T esum = 0;
    for (int p0 = lo + wave * SWEEP; p0 < hi; p0 += WV * SWEEP) {
      forward(p0, hi);
      NETN_TICK(2);
      T* Dl = dmat + lane;
#pragma unroll
      for (int b = 0; b < PB; ++b) {
        const int p = p0 + 64 * b + lane;
        esum += point_loss(p, p < hi, act + lane + out_row * RS + 64 * b, Dl + out_row * RS + 64 * b, grad);
      }
      NETN_TICK(3);
      if (!grad) continue;
      // ---- deltas of every layer (rows as in act: delta of layer l's input unit i sits in row aoff[l] + i)
#pragma unroll
      for (int l = NETN_MAX_LAYERS - 1; l >= 0; --l) {
        if (l >= nl) continue;
        const int I = a.dims[l], O = a.dims[l + 1];
        const T* W = th + woff[l];
        const T* Ar = act + lane + aoff[l] * RS;
        const T* dcur = Dl + aoff[l + 1] * RS;
        if constexpr (!MATRIX_GRAD) {                       // fp64: a weight's gradient = a reduction over the lanes
          for (int o = 0; o < O; ++o) {
            T d[PB], ds = 0;
#pragma unroll
            for (int b = 0; b < PB; ++b) { d[b] = dcur[o * RS + 64 * b]; ds += d[b]; }
            const T gb = wave_total(ds);
            if (lane == 0) lds_add(&gacc[boff[l] + o], gb);
            T* grow = gacc + woff[l] + o * I;
            int i = 0;
            for (; i + 4 <= I; i += 4) {
              T gq[4] = {0, 0, 0, 0};
#pragma unroll
              for (int b = 0; b < PB; ++b) {
#pragma unroll
                for (int u = 0; u < 4; ++u) gq[u] = fma(d[b], Ar[(i + u) * RS + 64 * b], gq[u]);
              }
              wave_total4(gq);
              if (lane < 4) lds_add(&grow[i + lane], lane == 0 ? gq[0] : lane == 1 ? gq[1] : lane == 2 ? gq[2] : gq[3]);
            }
            for (; i < I; ++i) {
              T gw = 0;
#pragma unroll
              for (int b = 0; b < PB; ++b) gw = fma(d[b], Ar[i * RS + 64 * b], gw);
              gw = wave_total(gw);
              if (lane == 0) lds_add(&grow[i], gw);
            }
          }
          NETN_TICK(4);
        }
        if (l > 0) {
          T* dprev = Dl + aoff[l] * RS;
          int i = 0;
          for (; i + 4 <= I; i += 4) {                      // (blocks of four input units: see forward())
            T sacc[4][PB];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
              for (int b = 0; b < PB; ++b) sacc[u][b] = 0;
            for (int o = 0; o < O; ++o) {
              T w[4], dv[PB];
#pragma unroll
              for (int u = 0; u < 4; ++u) w[u] = W[o * I + i + u];
#pragma unroll
              for (int b = 0; b < PB; ++b) dv[b] = dcur[o * RS + 64 * b];
#pragma unroll
              for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int b = 0; b < PB; ++b) sacc[u][b] = fma(dv[b], w[u], sacc[u][b]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
              for (int b = 0; b < PB; ++b) dprev[(i + u) * RS + 64 * b] = sacc[u][b] * netn_act_deriv<T>(a.act, Ar[(i + u) * RS + 64 * b]);
          }
          for (; i < I; ++i) {
            T sacc[PB];
#pragma unroll
            for (int b = 0; b < PB; ++b) sacc[b] = 0;
            for (int o = 0; o < O; ++o) {
              const T w0 = W[o * I + i];
#pragma unroll
              for (int b = 0; b < PB; ++b) sacc[b] = fma(dcur[o * RS + 64 * b], w0, sacc[b]);
            }
#pragma unroll
            for (int b = 0; b < PB; ++b) dprev[i * RS + 64 * b] = sacc[b] * netn_act_deriv<T>(a.act, Ar[i * RS + 64 * b]);
          }
          NETN_TICK(5);
        }
      }
      // ---- fp32: the gradient on the matrix cores.  G_l[o][i] = sum_p delta_l[p][o] a_(l-1)[p][i] is a sum of outer products
      // over the points; v_mfma_f32_4x4x1_16b_f32 forms 16 independent 4 x 4 outer products per instruction, so the weight
      // matrices of ALL layers (bias = an input column of ones) are cut into 4 x 4 blocks, 16 blocks per instruction, and one
      // point costs n_set instructions + 2 n_set LDS reads per lane (lane (block, k) reads row rowA of the deltas and row rowB of
      // the activations at the point's column) - where a reduction over the lanes per weight cost ~200 clocks each.  The
      // blocks accumulate in registers over all the points of the pass; scatter_gradient() adds them into gacc once.
      if constexpr (MATRIX_GRAD) {
        wave_sync();                                        // the columns written above are read across the lanes (of THIS wave)
        const int cnt = min(SWEEP, hi - p0);                 // (columns beyond cnt hold zero deltas: whole groups of 8 are safe)
        for (int c = 0; c < cnt; c += 8) {                  // 16 operand reads in flight per LDS round trip, then 8 instructions
#pragma unroll
          for (int s_ = 0; s_ < NETN_NSET; ++s_) {
            if (s_ >= n_set) break;
            const T* pa = dmat + rowA[s_] * RS + c;
            const T* pb = act + rowB[s_] * RS + c;
            float av[8], bv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { av[u] = pa[u]; bv[u] = pb[u]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) gblk[s_] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[u], bv[u], gblk[s_], 0, 0, 0);
          }
        }
        wave_sync();                                        // before the next sweep overwrites the columns
        NETN_TICK(4);
      }
    }
    if constexpr (MATRIX_GRAD) { if (grad) scatter_gradient(); }
    T etot = wave_total(esum);
    if constexpr (WV > 1) {                                   // the waves' sums through LDS: every wave reads the same total
      if (lane == 0) lds_add(&esh[eslot], etot);
      __syncthreads();
      etot = esh[eslot];
      eslot ^= 1;
    }
    return etot;
  }

  // which 4 x 4 block of which layer's [O x (I + 1)] gradient (weights | bias) every (instruction set, 4-lane block) holds
  __device__ __forceinline__ void grad_blocks() {
    const int blk = lane >> 2, idx = lane & 3;
#pragma unroll
    for (int s_ = 0; s_ < NETN_NSET; ++s_) { rowA[s_] = zero_row; rowB[s_] = zero_row; desc[s_] = -1; gblk[s_] = acc4{0.f, 0.f, 0.f, 0.f}; }
    int k = 0;
#pragma unroll
    for (int l = 0; l < NETN_MAX_LAYERS; ++l) {
      if (l >= nl) break;
      const int I = a.dims[l], O = a.dims[l + 1];
      const int OB = (O + 3) >> 2, IB = (I + 4) >> 2;      // I + 1 columns: the weights and the bias
      for (int ob = 0; ob < OB; ++ob)
        for (int ib = 0; ib < IB; ++ib, ++k) {
          if ((k & 15) != blk) continue;
          const int o = 4 * ob + idx, i = 4 * ib + idx;
          const int ra = o < O ? aoff[l + 1] + o : zero_row;
          const int rbb = i < I ? aoff[l] + i : (i == I ? ones_row : zero_row);
#pragma unroll
          for (int s_ = 0; s_ < NETN_NSET; ++s_)
            if (s_ == (k >> 4)) { rowA[s_] = ra; rowB[s_] = rbb; desc[s_] = l | (ob << 4) | (ib << 12); }
        }
    }
    n_set = (k + 15) >> 4;
  }

  // after the last sweep of a gradient pass: the accumulated blocks into the wave's gradient vector; accumulators cleared
  __device__ __forceinline__ void scatter_gradient() {
    const int n = lane & 3;
#pragma unroll
    for (int s_ = 0; s_ < NETN_NSET; ++s_) {
      if (s_ >= n_set) break;
      const int dsc = desc[s_];
      if (dsc >= 0) {
        const int l = dsc & 15, ob = (dsc >> 4) & 255, ib = dsc >> 12;
        int I = a.dims[0], O = a.dims[1], wo = woff[0], bo = boff[0];
#pragma unroll
        for (int ll = 1; ll < NETN_MAX_LAYERS; ++ll) if (l == ll) { I = a.dims[ll]; O = a.dims[ll + 1]; wo = woff[ll]; bo = boff[ll]; }
        const int i = 4 * ib + n;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = 4 * ob + r;
          if (o < O && i <= I) lds_add(i < I ? &gacc[wo + o * I + i] : &gacc[bo + o], (T)gblk[s_][r]);
        }
      }
      gblk[s_] = acc4{0.f, 0.f, 0.f, 0.f};
    }
  }

  // after a gradient pass: d log p_m / d theta = the summed likelihood gradient + prior gradient / prior_scale  (S:1156)
  __device__ __forceinline__ void collect_gradient(const Rec& q, Rec& g) {
    __syncthreads();                                        // lane 0's LDS adds
    const T ips = (T)1 / a.prior_scale;
#pragma unroll
    for (int k = 0; k < NETN_KMAX; ++k) {
      const int pid = 64 * k + lane;
      g.v[k] = pid < D ? gacc[pid] - ips * tauv.v[k] * q.v[k] : (T)0;
    }
  }

  // prior log-density (whole, not divided): sum_t [ -1/2 tau_t sum w^2 + n_t (1/2 log tau_t - 1/2 log 2 pi) ]
  __device__ __forceinline__ T log_prior(const Rec& w) {
    T qq = 0;
#pragma unroll
    for (int k = 0; k < NETN_KMAX; ++k) qq = fma(tauv.v[k] * w.v[k], w.v[k], qq);       // entries beyond D are exactly 0
    return (T)-0.5 * wave_total(qq) + prior_const;
  }
  __device__ __forceinline__ T kinetic(const Rec& p, const Rec& im) {
    T k = 0;
#pragma unroll
    for (int q = 0; q < NETN_KMAX; ++q) k = fma(p.v[q] * im.v[q], p.v[q], k);
    return (T)0.5 * wave_total(k);
  }
  static __device__ __forceinline__ void axpy(Rec& y, T c, const Rec& x) {
#pragma unroll
    for (int k = 0; k < NETN_KMAX; ++k) y.v[k] = fma(c, x.v[k], y.v[k]);
  }
  static __device__ __forceinline__ void drift(Rec& q, T c, const Rec& im, const Rec& p) {   // q += c M^-1 p
#pragma unroll
    for (int k = 0; k < NETN_KMAX; ++k) q.v[k] = fma(c * im.v[k], p.v[k], q.v[k]);
  }
};

template <typename T, int PB, int WV>
__global__ __launch_bounds__(64 * WV) void netn_hmc_kernel(NetArgs<T> a, int D, int SW, int WM) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  typedef NetChain<T, PB, WV> Ch;
  typedef typename Ch::Rec Rec;
  Ch ch(a);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  ch.lane = lane; ch.wave = wave; ch.D = D; ch.nl = a.n_layers; ch.eslot = 0;
  const int Dp = (D + 63) & ~63;
  ch.th = reinterpret_cast<T*>(smem_raw);
  ch.gacc = ch.th + Dp;
  ch.esh = ch.gacc + Dp;
  T* mats = ch.esh + 4;
  ch.act = mats + (size_t)wave * 2 * (SW + 2) * Ch::RS;    // per wave: act and dmat, SW rows + two special rows each (act: ones, zeros; dmat: zeros, zeros)
  ch.dmat = ch.act + (size_t)(SW + 2) * Ch::RS;
  ch.perm = reinterpret_cast<int*>(mats + (size_t)WV * 2 * (SW + 2) * Ch::RS);
  ch.ones_row = SW; ch.zero_row = SW + 1;
  for (int e = lane; e < Ch::RS; e += 64) { ch.act[(size_t)SW * Ch::RS + e] = (T)1; ch.act[(size_t)(SW + 1) * Ch::RS + e] = (T)0; }
  for (int e = lane; e < (SW + 2) * Ch::RS; e += 64) ch.dmat[e] = (T)0;
  {
    int off = 0, ao = 0;
    T pc = 0;
    const T hl2p = (T)0.9189385332046727;
    ch.aoff[0] = 0;
#pragma unroll
    for (int l = 0; l < NETN_MAX_LAYERS; ++l) {
      if (l >= a.n_layers) { ch.woff[l] = ch.boff[l] = 0; ch.aoff[l + 1] = ao; continue; }
      const int I = a.dims[l], O = a.dims[l + 1];
      ch.woff[l] = off; off += I * O;
      ch.boff[l] = off; off += O;
      ao += I; ch.aoff[l + 1] = ao;
      pc += (T)(I * O) * ((T)0.5 * log(a.tau[2 * l]) - hl2p) + (T)O * ((T)0.5 * log(a.tau[2 * l + 1]) - hl2p);
    }
    ch.prior_const = pc;
    ch.out_row = ao;
    int no = a.dims[1];
#pragma unroll
    for (int l = 1; l < NETN_MAX_LAYERS; ++l) if (l < a.n_layers) no = a.dims[l + 1];
    ch.n_out = no;
  }
  if constexpr (Ch::MATRIX_GRAD) ch.grad_blocks();
  __syncthreads();
  Rec im, mf;
  const bool dg = a.mass_kind == HTA_MASS_DIAG;
#pragma unroll
  for (int k = 0; k < NETN_KMAX; ++k) {
    const int pid = 64 * k + lane;
    const bool ok = pid < D;
    T tv = 0;
    if (ok) {
#pragma unroll
      for (int l = 0; l < NETN_MAX_LAYERS; ++l) {
        if (l >= a.n_layers) break;
        if (pid >= ch.woff[l] && pid < ch.boff[l]) tv = a.tau[2 * l];
        if (pid >= ch.boff[l] && pid < ch.boff[l] + a.dims[l + 1]) tv = a.tau[2 * l + 1];
      }
    }
    ch.tauv.v[k] = tv;
    im.v[k] = (ok && dg) ? a.inv_mass[pid] : (T)1;
    mf.v[k] = (ok && dg) ? a.mass_factor[pid] : (T)1;
  }
  auto load_rec = [&](const T* src, Rec& w) {
#pragma unroll
    for (int k = 0; k < NETN_KMAX; ++k) { const int pid = 64 * k + lane; w.v[k] = pid < D ? src[pid] : (T)0; }
  };
  auto store_rec = [&](T* dst, const Rec& w) {
#pragma unroll
    for (int k = 0; k < NETN_KMAX; ++k) { const int pid = 64 * k + lane; if (pid < D) dst[pid] = w.v[k]; }
  };

  // The likelihood pass (forward, loss, backward: the bulk of the kernel's code, unrolled over the layers) has ONE call site:
  // a chain's run is a sequence of passes - [log p of the current point when it is not known] then, per trajectory, the
  // integrator's gradient stages and the log p of the proposal - driven by a small state machine around that call.  (Four
  // inlined copies - initial log p, stages, proposal, reset - tripled the code and spilled 226 scalar registers.)
  const T eps = a.eps, heps = (T)0.5 * a.eps;
  const int M = a.M;
  const int nstage = a.n_traj > 0 ? split_stage_count(a.integ, M, a.L) : 0;
  const int all_pts = M * a.Nb;
  for (int64_t c = blockIdx.x; c < a.C; c += gridDim.x) {
    const uint64_t chain = a.chain_offset + (uint64_t)c;
    Rec cur, q, p, g;
    load_rec(a.theta + c * D, cur);
    q = cur;
#pragma unroll
    for (int k = 0; k < NETN_KMAX; ++k) { p.v[k] = 0; g.v[k] = 0; }
    T lp_cur = 0, h_old = 0;
    bool lp_known = false;
    int32_t rejected = 0;
    int t = 0, st = -1;                // st: -1 = before the trajectory's first pass, 0 .. nstage-1 = stages, nstage = proposal's log p
    for (;;) {
      // ---- what the next pass evaluates
      int lo, hi; bool grad;
      int m = 0; T kick = 0, dr = 0;
      if (a.n_traj == 0) { lo = a.eval_split * a.Nb; hi = lo + a.Nb; grad = true; }          // evaluation-only (parity tests)
      else {
        if (t >= a.n_traj) break;
        if (st < 0) {                  // trajectory start: gibbs (S:185-186 / S:200-201), the subset order of SPLITTING_RAND (S:549)
          const int n = a.traj_offset + t;
#pragma unroll
          for (int k = 0; k < NETN_KMAX; ++k) {
            const int pid = 64 * k + lane;
            p.v[k] = pid < D ? mf.v[k] * normal_elem<T>(a.seed, chain, (uint32_t)n, 0, pid) : (T)0;
          }
          q = cur;
          if (a.integ == HTA_SPLIT_RAND) {
            __syncthreads();
            if (threadIdx.x == 0) split_permutation(a.seed, (uint32_t)n, M, ch.perm);
            __syncthreads();
          }
          if (lp_known) st = 0;        // else: this pass is log p of the current point (first trajectory / after the Q2 reset)
        }
        if (st < 0 || st >= nstage) { lo = 0; hi = all_pts; grad = false; }
        else {
          split_stage<T>(a.integ, M, a.L, st, eps, ch.perm, m, kick, dr);
          lo = m * a.Nb; hi = lo + a.Nb; grad = true;
        }
      }
      // ---- the pass
      const T esum = ch.pass(q, lo, hi, grad);
      const T ll = (T)-0.5 * a.tau_out * esum;
      if (grad) ch.collect_gradient(q, g);
      // ---- what it was for
      if (a.n_traj == 0) {
        const T lp = ll + ch.log_prior(q) / a.prior_scale;
        if (a.grad_out && wave == 0) store_rec(a.grad_out + c * D, g);
        if (a.logp_out && threadIdx.x == 0) a.logp_out[c] = lp;
        break;
      }
      if (st < 0) {                    // log p of the current point (q == cur)
        lp_cur = ll + ((T)M / a.prior_scale) * ch.log_prior(q);                              // S:787-796
        lp_known = true;
        st = 0;
        if (nstage > 0) continue;
      }
      if (st == 0) h_old = -lp_cur + ch.kinetic(p, im);                                        // S:971 (before the first stage's kick)
      if (st < nstage) {
        if (grad) {
          Ch::axpy(p, kick, g);
          if (dr != (T)0) Ch::drift(q, dr, im, p);
          ++st;
          // the stages that evaluate the same subset at the same parameters (no drift since) take this gradient (mlp.hpp)
          while (st < nstage) {
            int m2; T k2, d2;
            split_stage<T>(a.integ, M, a.L, st, eps, ch.perm, m2, k2, d2);
            if (!split_stage_reuses<T>(m, dr, m2)) break;
            Ch::axpy(p, k2, g);
            if (d2 != (T)0) Ch::drift(q, d2, im, p);
            dr = d2; ++st;
          }
          if (st == nstage && M == 1 && a.integ == HTA_SPLIT_SYMMETRIC) Ch::axpy(p, -heps, g);   // S:302
          continue;
        }
      }
      // ---- the proposal's log p (S:995), the MH test (S:1000-1004), bookkeeping (S:1006-1026)
      {
        const int n = a.traj_offset + t;
        const T lp_new = ll + ((T)M / a.prior_scale) * ch.log_prior(q);
        const T h_new = -lp_new + ch.kinetic(p, im);
        const T u = u23<T>(philox_block(a.seed, chain, (uint32_t)n, PURPOSE_MH, 0, 0).x);
        const bool acc = mh_accept<T>(h_old, h_new, lp_new, u);
        if (acc) { cur = q; lp_cur = lp_new; }
        else {
          ++rejected;
          if (n == a.burn + 1) {                                            // Q2 reset to params_init (S:1018)
            load_rec(a.theta_init + c * D, cur);
            lp_known = false;                                               // its log p: the next trajectory's first pass
          }
        }
        if (a.samples && n > a.burn && wave == 0) store_rec(a.samples + ((int64_t)(n - a.burn) * a.C + c) * D, cur);
        if (threadIdx.x == 0) {
          if (a.H_old) a.H_old[(int64_t)t * a.C + c] = h_old;
          if (a.H_new) a.H_new[(int64_t)t * a.C + c] = h_new;
          if (a.accept) a.accept[(int64_t)t * a.C + c] = acc ? 1 : 0;
        }
        ++t; st = -1;
      }
    }
    if (a.n_traj > 0) {
      if (wave == 0) store_rec(a.theta + c * D, cur);
      if (threadIdx.x == 0 && a.reject_count) a.reject_count[c] += rejected;
    }
#if NETN_TIMING
    if (threadIdx.x == 0 && blockIdx.x == 0) for (int k = 0; k < 8; ++k) hta_netn_dbg[k] = ch.tacc[k];
#endif
  }
}

template <typename T> int netn_hmc(const NetArgs<T>& a, hipStream_t s) {
  HTA_REQUIRE(a.theta && a.X && a.Y && a.C > 0, "hta_netn_hmc: NULL pointer / empty batch");
  if constexpr (sizeof(T) == 4) {          // two wide hidden layers, one output, Gaussian likelihood: the matrix-core kernel (mlp3_mfma.hip)
    if (a.n_layers == 3 && mlp3_eligible(a)) return mlp3_mfma(a, s);
  }
  HTA_REQUIRE(a.n_layers >= 1 && a.n_layers <= NETN_MAX_LAYERS, "hta_netn_hmc: %d Linear layers not in [1, %d]", a.n_layers, NETN_MAX_LAYERS);
  int D = 0, SW = a.dims[0], WM = 1;
  for (int l = 0; l <= a.n_layers; ++l)
    HTA_REQUIRE(a.dims[l] >= 1 && a.dims[l] <= NETN_MAX_WIDTH, "hta_netn_hmc: layer width %d not in [1, %d]", a.dims[l], NETN_MAX_WIDTH);
  for (int l = 0; l < a.n_layers; ++l) {
    D += a.dims[l] * a.dims[l + 1] + a.dims[l + 1];
    SW += a.dims[l + 1];
    if (a.dims[l + 1] > WM) WM = a.dims[l + 1];
    if (a.dims[l] > WM) WM = a.dims[l];
  }
  HTA_REQUIRE(D <= 64 * NETN_KMAX, "hta_netn_hmc: %d parameters exceed the native limit of %d", D, 64 * NETN_KMAX);
  HTA_REQUIRE(a.act >= 0 && a.act <= 2, "hta_netn_hmc: unknown activation %d", a.act);
  HTA_REQUIRE(a.loss == HTA_LOSS_REGRESSION || a.loss == HTA_LOSS_BINARY_LOGITS || a.loss == HTA_LOSS_SOFTMAX_CE,
              "hta_netn_hmc: unknown loss kind %d", a.loss);
  HTA_REQUIRE(a.M >= 1 && a.Nb >= 1 && (int64_t)a.M * a.Nb <= a.N, "hta_netn_hmc: M=%d splits of Nb=%d points exceed N=%d", a.M, a.Nb, a.N);
  HTA_REQUIRE(a.mass_kind == HTA_MASS_NONE || (a.mass_kind == HTA_MASS_DIAG && a.inv_mass && a.mass_factor),
              "hta_netn_hmc: only identity / diagonal inv_mass are supported natively");
  if (a.n_traj > 0) HTA_REQUIRE(a.theta_init && a.L >= 0, "hta_netn_hmc: bad trajectory arguments");
  if (a.n_traj == 0) HTA_REQUIRE(a.eval_split >= 0 && a.eval_split < a.M, "hta_netn_logp_grad: split %d not in [0, %d)", a.eval_split, a.M);
  HTA_REQUIRE(a.integ >= HTA_SPLIT_SYMMETRIC && a.integ <= HTA_SPLIT_KMID, "hta_netn_hmc: unknown integrator %d", a.integ);
  HTA_REQUIRE(a.integ != HTA_SPLIT_RAND || a.M <= 64, "hta_netn_hmc: SPLITTING_RAND supports at most 64 subsets natively (M=%d)", a.M);
  HTA_REQUIRE(a.integ != HTA_SPLIT_KMID || a.M >= 2, "hta_netn_hmc: SPLITTING_KMID needs at least 2 subsets");
  // Waves per chain (WV) and points per lane (PB), from the size of a GRADIENT pass (Nb points; the two full-data log p passes
  // of a trajectory are the minority).  A sweep of a wave covers 64 PB points.  Both: the largest of 4, 2, 1 that a pass keeps
  // more than half busy - while the LDS footprint still lets every chain of the launch be resident (a second round of
  // workgroups costs more than either buys: at 1024 chains four workgroups share a CU).
  const int Dp = (D + 63) & ~63;
  auto lds_for = [&](int pb, int wv) {
    return ((size_t)2 * Dp + 4 + (size_t)wv * 2 * (SW + 2) * (64 * pb + 1)) * sizeof(T) + 64 * sizeof(int);
  };
  if (sizeof(T) == 4) {                                     // the matrix-core gradient holds at most 16 NETN_NSET blocks of 4 x 4
    int nb = 0;
    for (int l = 0; l < a.n_layers; ++l) nb += ((a.dims[l + 1] + 3) / 4) * ((a.dims[l] + 4) / 4);
    HTA_REQUIRE(nb <= 16 * NETN_NSET, "hta_netn_hmc: %d blocks of 4 x 4 weights exceed the native limit of %d", nb, 16 * NETN_NSET);
  }
  int cus = 256;
  {
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
  }
  auto resident = [&](int pb, int wv) {
    const size_t per_cu = (size_t)(150 * 1024) / lds_for(pb, wv);
    const size_t by_waves = (size_t)32 / wv;                // 8 waves per SIMD at most
    return (int64_t)cus * (int64_t)(per_cu < by_waves ? per_cu : by_waves);
  };
  // Measured (profiles/r02z_netn_speed.txt): more waves per chain LOSE at 1024 chains - 1.46e7 -> 1.11e7 chain-steps/s on
  // Net([1,10,10,1]) with two waves and one point per lane instead of one wave and two points, 7.9e7 -> 4.3e7 on the softmax
  // regression with four: the real barriers and the LDS exchange cost more than the second wave's latency hiding buys (the
  // RMHMC kernels found the same).  So one wave per chain unless the tuning key "netn_waves" asks for more.
  int WV = g_netn_waves >= 4 ? 4 : (g_netn_waves >= 2 ? 2 : 1), PB = 4;
  while (WV > 1 && (a.Nb <= 32 * WV || resident(1, WV) < a.C)) WV >>= 1;
  while (PB > 1 && (a.Nb <= 32 * PB * WV || resident(PB, WV) < a.C)) PB >>= 1;
  const size_t lds = lds_for(PB, WV);
  HTA_REQUIRE(lds <= 150 * 1024, "hta_netn_hmc: the layer widths need %zu bytes of LDS", lds);
  const int grid = (int)(a.C < 65536 ? a.C : 65536);
  auto launch = [&](auto kern, DevOnce& done, int threads) -> int {
    if (!done) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) { set_error("hta_netn_hmc: hipFuncSetAttribute: %s", hipGetErrorString(e)); return HTA_ERR_LAUNCH; }
      done = true;
    }
    profile_begin(s);
    note_route("netn_hmc_kernel<%s,%d,%d>", sizeof(T) == 4 ? "float" : "double", PB, WV);
    kern<<<grid, threads, lds, s>>>(a, D, SW, WM);
    profile_end(s);
    return HTA_OK;
  };
  static DevOnce done[9];
  int rc = HTA_OK;
#define NETN_CASE(P, W, IDX) if (PB == P && WV == W) rc = launch(&netn_hmc_kernel<T, P, W>, done[IDX], 64 * W);
  NETN_CASE(1, 1, 0) NETN_CASE(1, 2, 1) NETN_CASE(1, 4, 2)
  NETN_CASE(2, 1, 3) NETN_CASE(2, 2, 4) NETN_CASE(2, 4, 5)
  NETN_CASE(4, 1, 6) NETN_CASE(4, 2, 7) NETN_CASE(4, 4, 8)
#undef NETN_CASE
  if (rc) return rc;
  HTA_CHECK_LAUNCH("hta_netn_hmc");
  return HTA_OK;
}

template <typename T>
static int netn_fill(NetArgs<T>& a, int n_layers, const int* dims, const T* taus) {
  if (!dims || !taus) { set_error("hta_netn: dims / taus is NULL (host pointers)"); return HTA_ERR_INVALID; }
  if (n_layers < 1 || n_layers > NETN_MAX_LAYERS) { set_error("hta_netn: %d Linear layers not in [1, %d]", n_layers, NETN_MAX_LAYERS); return HTA_ERR_INVALID; }
  a.n_layers = n_layers;
  for (int l = 0; l <= NETN_MAX_LAYERS; ++l) a.dims[l] = l <= n_layers ? dims[l] : 0;
  for (int l = 0; l < 2 * NETN_MAX_LAYERS; ++l) a.tau[l] = l < 2 * n_layers ? taus[l] : (T)1;
  return HTA_OK;
}

}  // namespace hta

extern "C" {
#define HTA_DEFINE_NETN(SUF, T)                                                                                          \
  int hta_netn_hmc_sample_##SUF(T* theta, const T* theta_init, int64_t C, int n_layers, const int* dims, int act,        \
                                int loss_kind, const T* X, const T* Y, int N, int M, int Nb, const T* taus, T tau_out,   \
                                T prior_scale, int mass_kind, const T* inv_mass, const T* mass_factor, int integrator,   \
                                int L, T eps, int n_traj, int traj_offset, int burn, uint64_t seed, uint64_t chain_offset, \
                                T* samples, int32_t* reject_count, T* H_old, T* H_new, uint8_t* accept, void* workspace, \
                                int64_t workspace_bytes, void* stream) {                                                 \
    hta::NetArgs<T> a{};                                                                                                 \
    if (int rc = hta::netn_fill<T>(a, n_layers, dims, taus)) return rc;                                                  \
    a.theta = theta; a.theta_init = theta_init; a.C = C; a.act = act; a.loss = loss_kind; a.X = X; a.Y = Y; a.N = N;      \
    a.M = M; a.Nb = Nb; a.tau_out = tau_out; a.prior_scale = prior_scale; a.mass_kind = mass_kind; a.inv_mass = inv_mass; \
    a.mass_factor = mass_factor; a.L = L; a.eps = eps; a.n_traj = n_traj; a.traj_offset = traj_offset; a.burn = burn;    \
    a.seed = seed; a.chain_offset = chain_offset; a.samples = samples; a.reject_count = reject_count; a.H_old = H_old;   \
    a.H_new = H_new; a.accept = accept; a.integ = integrator; a.workspace = workspace; a.workspace_bytes = workspace_bytes; \
    if (n_traj <= 0) return HTA_OK;                                                                                      \
    return hta::netn_hmc<T>(a, (hipStream_t)stream);                                                                     \
  }                                                                                                                      \
  int hta_netn_logp_grad_##SUF(const T* theta, int64_t C, int n_layers, const int* dims, int act, int loss_kind, const T* X, \
                               const T* Y, int N, int M, int Nb, int split, const T* taus, T tau_out, T prior_scale,     \
                               T* grad_out, T* logp_out, void* workspace, int64_t workspace_bytes, void* stream) {       \
    hta::NetArgs<T> a{};                                                                                                 \
    if (int rc = hta::netn_fill<T>(a, n_layers, dims, taus)) return rc;                                                  \
    a.theta = const_cast<T*>(theta); a.C = C; a.act = act; a.loss = loss_kind; a.X = X; a.Y = Y; a.N = N; a.M = M;        \
    a.Nb = Nb; a.tau_out = tau_out; a.prior_scale = prior_scale; a.mass_kind = HTA_MASS_NONE; a.grad_out = grad_out;     \
    a.logp_out = logp_out; a.eval_split = split; a.integ = HTA_SPLIT_SYMMETRIC; a.workspace = workspace;                 \
    a.workspace_bytes = workspace_bytes;                                                                                 \
    return hta::netn_hmc<T>(a, (hipStream_t)stream);                                                                     \
  }
HTA_DEFINE_NETN(f32, float)
HTA_DEFINE_NETN(f64, double)
#undef HTA_DEFINE_NETN

int64_t hta_netn_hmc_workspace_bytes(int64_t C, int n_layers, const int* dims, int elem_size) {
  return elem_size == 4 ? hta::mlp3_workspace_bytes(C, n_layers, dims) : 0;      /* the one-wave-per-chain kernels need none */
}
}
