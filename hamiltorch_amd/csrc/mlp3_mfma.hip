// Fused (split-)HMC for a Bayesian MLP with TWO wide hidden layers, fp32 on the gfx950 matrix cores:
//     Linear(n_in, H1) - act - Linear(H1, H2) - act - Linear(H2, 1),  Gaussian likelihood,  n_in <= 4,  H1, H2 <= 104.
// This is the one model the reference publishes a GPU number for - notebooks/hamiltorch_split_HMC_BNN_example.ipynb cell 9:
// Linear(1,100)-ReLU-Linear(100,100)-ReLU-Linear(100,1), D = 10401, 400 points in M = 4 splits, L = 30, eps = 5e-4
// (13.47 samples/s full HMC, 1.83 samples/s symmetric split HMC on an RTX 2080 Max-Q, one chain).
// Same contract as mlp_mfma.hip / netn_hmc.hip: hamiltorch/samplers.py:965-1026 (trajectory loop), S:1141-1199 (closures of
// define_model_log_prob / define_split_model_log_prob), S:499-596 (split integrators), S:281-302 (leapfrog).
//
// Cost of a gradient: three (points x H) x (H x H) products - Z2 = A1 W2^T, dW2 = delta2^T A1, delta1 = delta2 W2 -
// 6 N_b H1 H2 flop, 100 x the element-wise work of the thin first and last layers: a GEMM problem, v_mfma_f32_16x16x4_f32
// (exact fp32).  ONE WORKGROUP OF 7 WAVES PER CHAIN; wave w owns the 16 second-layer units 16 w .. 16 w + 15 and the 16
// first-layer units of the same indices; lane l = (g = l >> 4, c = l & 15).
//
// State in registers.  theta, p and the gradient of a chain live in VGPRs for the whole launch.  The 100 x 100 matrix W2 is
// held as 7 x 4 registers per lane in the OPERAND layout of the forward product: lane (g, c) of wave w keeps row
// u = 16 w + c, register [s][t] <-> column j = 16 s + 4 g + t (s < 6; the last block is j = 96 + 4 t + g).  A matrix
// instruction sums over its 4 K slots in any order, so a block of 16 contraction indices is consumed by 4 instructions whose
// K slot g at step t is index 16 s + 4 g + t: BOTH operands of a step are then one 16-byte LDS read (ds_read_b128, rows
// padded to 104 / 120 floats: conflict free) per FOUR instructions, instead of one 4-byte read per instruction.
//   forward  Z2[p, u]  = b2_u + sum_j A1[p, j] W2[u, j]    A = A1[p][.] from LDS, B = the theta registers themselves
//   dW2^T[j, u]        = sum_p A1[p, j] delta2[p, u]        A = A1T[j][.] from LDS, B = delta2 AS IT LIES in the forward
//                        accumulator (C layout: lane (g, c) register r = point 4 g + r, unit c = K slot g, step r); the
//                        result's C layout (row j = 16 s + 4 g + r, column u = 16 w + c) IS the theta register layout, so kick
//                        and drift of W2 are element-wise register operations
//   delta1[p, j]       = sum_u delta2[p, u] W2[u, j]        A = delta2[p][.] from LDS, B = W2 re-read column-wise from a
//                        staging copy (7 x 4 more registers, refreshed once per gradient)
// f(x_p) = b3 + sum_u w3_u a2[p, u]: reduce-scatter over the 16 unit lanes (DPP), over the waves through LDS.  The thin layers
// (W1, b1, b2, w3, b3: 401 of 10401 parameters) are per-unit registers with copies in the 4 lane groups.
// LDS (151 KB): A1 [112][104] | D2 [112][104] (delta2; doubles as the W2 staging copy) | A1T [112][120] | X, Y of the chunk | partials.
#include <map>
#include <mutex>
#include <utility>
#include "netn.hpp"
#include "mlp.hpp"
#include "philox.hpp"

#ifndef M3_TIMING
#define M3_TIMING 0   // developer cycle counters per phase of a pass (wave 0 of workgroup 0): tools/scratch/m3_time.py prints them
#endif
#if M3_TIMING
__device__ unsigned long long hta_m3_dbg[20];
extern "C" void hta_m3_dbg_read(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(hta_m3_dbg), sizeof(hta_m3_dbg)); }
#define M3_TICK(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); tacc[k] += now_ - tacc[19]; tacc[19] = now_; } while (0)
#else
#define M3_TICK(k) do {} while (0)
#endif

namespace hta {

void profile_begin(hipStream_t s);
void profile_end(hipStream_t s);

typedef float V4f __attribute__((ext_vector_type(4)));

constexpr int M3_NW = 7;                 // waves per chain = tiles of 16 units
constexpr int M3_NT = 64 * M3_NW;        // 448 threads
constexpr int M3_HMAX = 104;             // widest hidden layer
constexpr int M3_LDJ = 104;              // row stride of A1 [p][j], D2 [p][u] and the W2 staging copy [u][j]
constexpr int M3_LDP = 120;              // row stride of A1T [j][p]
constexpr int M3_CP = 112;               // points per chunk (7 tiles)
constexpr int M3_NIN = 4;                // widest input layer

template <int CTRL> __device__ __forceinline__ float m3_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float m3_groups_sum(float v) {       // over the 4 lane groups g; every lane gets the total
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
template <int ACT> __device__ __forceinline__ float m3_act(float z) {
  if (ACT == 0) return __builtin_amdgcn_fmed3f(z, 0.0f, __builtin_inff());
  if (ACT == 1) return tanhf(z);
  return 1.0f / (1.0f + expf(-z));
}
template <int ACT> __device__ __forceinline__ float m3_dact(float h) {      // act' in terms of the activation value
  if (ACT == 0) return h > 0.0f ? 1.0f : 0.0f;
  if (ACT == 1) return 1.0f - h * h;
  return h * (1.0f - h);
}

// A lane's share of one parameter-shaped vector (theta, momentum, gradient, masses)
struct M3State {
  V4f w2[7];                  // W2[u = 16 w + c][j]: [s][t] <-> j = 16 s + 4 g + t (s < 6), j = 96 + 4 t + g (s = 6)
  float w1[M3_NIN];           // W1[16 w + c][k]   (copies in the 4 lane groups, like b1, b2, w3; b3 everywhere)
  float b1, b2, w3, b3;
};

// The per-lane description of where a state vector's elements live in the flat parameter vector (passed BY VALUE to the
// out-of-line helpers below: the rarely executed code - loads, stores, momentum draws, hundreds of instructions with their own
// register needs - must not share the register allocation of the gradient pass, or the allocator parks the momentum in scratch
// memory across every pass: measured, 12 000 of a pass's 74 000 cycles were serial scratch round trips in the kick).
struct M3Lay {
  int w, g, c, n_in, H1, H2, rq1;
  int o_w1, o_b1, o_w2, o_b2, o_w3, o_b3;       // flat offsets (torch parameter order: W1, b1, W2, b2, W3, b3)
  bool jval, uval, vecd;                         // layer-1 unit / layer-2 unit 16 w + c exists; rows of W2 start on a multiple of 4 elements
  // column of register [s][t] and whether the element exists
  __device__ __forceinline__ int col(int s, int t) const { return s < 6 ? 16 * s + 4 * g + t : 96 + 4 * t + g; }
  __device__ __forceinline__ bool ok(int s, int t) const { return uval && col(s, t) < H1 && (s < 6 || t < rq1); }
};

// element idx of `base` if it exists, else 0 - an unconditional load from a clamped address and a select (no branch per element)
static __device__ __forceinline__ float m3_pick(const float* base, int idx, bool exists) {
  const float v = base[exists ? idx : 0];
  return exists ? v : 0.0f;
}

// vec4: th + o_w2 + u H1 is 16-byte aligned for every row (the caller knows the base's alignment)
static __device__ __forceinline__ void m3_load_inline(const M3Lay& L, const float* th, bool vec4, M3State& q) {
  const int u = 16 * L.w + L.c;
  const float* row = th + L.o_w2 + (size_t)(L.uval ? u : 0) * L.H1;
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    const int j0 = 16 * s + 4 * L.g;
    if (vec4 && L.uval && j0 + 4 <= L.H1) q.w2[s] = *reinterpret_cast<const V4f*>(row + j0);
    else
#pragma unroll
      for (int t = 0; t < 4; ++t) q.w2[s][t] = m3_pick(row, j0 + t, L.ok(s, t));
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) q.w2[6][t] = m3_pick(row, L.col(6, t), L.ok(6, t));
  const int j1 = L.jval ? 16 * L.w + L.c : 0, u2 = L.uval ? u : 0;
#pragma unroll
  for (int k = 0; k < M3_NIN; ++k) q.w1[k] = m3_pick(th + L.o_w1 + j1 * L.n_in, k, L.jval && k < L.n_in);
  q.b1 = m3_pick(th + L.o_b1, j1, L.jval);
  q.b2 = m3_pick(th + L.o_b2, u2, L.uval);
  q.w3 = m3_pick(th + L.o_w3, u2, L.uval);
  q.b3 = th[L.o_b3];
}
__device__ __attribute__((noinline)) void m3_load(M3Lay L, const float* th, bool vec4, M3State* qp) {
  M3State q;
  m3_load_inline(L, th, vec4, q);
  *qp = q;
}
// every copy lane stores (same value, same address): a lane later reloads exactly what it stored itself
__device__ __attribute__((noinline)) void m3_store(M3Lay L, float* th, bool vec4, const M3State* qp) {
  const M3State q = *qp;
  const int u = 16 * L.w + L.c;
  float* row = th + L.o_w2 + (size_t)(L.uval ? u : 0) * L.H1;
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    const int j0 = 16 * s + 4 * L.g;
    if (vec4 && L.uval && j0 + 4 <= L.H1) *reinterpret_cast<V4f*>(row + j0) = q.w2[s];
    else
#pragma unroll
      for (int t = 0; t < 4; ++t) if (L.ok(s, t)) row[j0 + t] = q.w2[s][t];
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) if (L.ok(6, t)) row[L.col(6, t)] = q.w2[6][t];
#pragma unroll
  for (int k = 0; k < M3_NIN; ++k) if (L.jval && k < L.n_in) th[L.o_w1 + (16 * L.w + L.c) * L.n_in + k] = q.w1[k];
  if (L.jval) th[L.o_b1 + 16 * L.w + L.c] = q.b1;
  if (L.uval) { th[L.o_b2 + u] = q.b2; th[L.o_w3 + u] = q.w3; }
  th[L.o_b3] = q.b3;
}
// p = sqrt(M) z, z a standard-normal draw in the state layout (element i of the D-vector = Philox block i / 4, slot i % 4: the
// oracle's stream; S:185-186 / S:200-201); mass_factor NULL = identity mass
__device__ __attribute__((noinline)) void m3_draw(M3Lay L, uint64_t seed, uint64_t chain, uint32_t n, const float* mass_factor, M3State* zp) {
  M3State z;
  const int u = 16 * L.w + L.c;
  const int rowoff = L.o_w2 + (L.uval ? u : 0) * L.H1;
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    const int j0 = 16 * s + 4 * L.g;
    if (L.vecd && L.uval && j0 + 4 <= L.H1) {                   // four consecutive elements = one Philox block
      float zz[4];
      normal4<float>(philox_block(seed, chain, n, PURPOSE_MOMENTUM, 0, (uint32_t)((rowoff + j0) >> 2)), zz);
#pragma unroll
      for (int t = 0; t < 4; ++t) z.w2[s][t] = zz[t];
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) { const float v = normal_elem<float>(seed, chain, n, 0, rowoff + j0 + t); z.w2[s][t] = L.ok(s, t) ? v : 0.0f; }
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) { const float v = normal_elem<float>(seed, chain, n, 0, rowoff + L.col(6, t)); z.w2[6][t] = L.ok(6, t) ? v : 0.0f; }
  const int j1 = L.jval ? 16 * L.w + L.c : 0, u2 = L.uval ? u : 0;
#pragma unroll
  for (int k = 0; k < M3_NIN; ++k) {
    float v = 0.0f;
    if (k < L.n_in) v = normal_elem<float>(seed, chain, n, 0, L.o_w1 + j1 * L.n_in + k);      // (uniform branch)
    z.w1[k] = L.jval ? v : 0.0f;
  }
  { const float v = normal_elem<float>(seed, chain, n, 0, L.o_b1 + j1); z.b1 = L.jval ? v : 0.0f; }
  { const float v = normal_elem<float>(seed, chain, n, 0, L.o_b2 + u2); z.b2 = L.uval ? v : 0.0f; }
  { const float v = normal_elem<float>(seed, chain, n, 0, L.o_w3 + u2); z.w3 = L.uval ? v : 0.0f; }
  z.b3 = normal_elem<float>(seed, chain, n, 0, L.o_b3);
  if (mass_factor) {
    M3State mf;
    m3_load(L, mass_factor, L.vecd, &mf);
#pragma unroll
    for (int s = 0; s < 7; ++s)
#pragma unroll
      for (int t = 0; t < 4; ++t) z.w2[s][t] *= mf.w2[s][t];
#pragma unroll
    for (int k = 0; k < M3_NIN; ++k) z.w1[k] *= mf.w1[k];
    z.b1 *= mf.b1; z.b2 *= mf.b2; z.w3 *= mf.w3; z.b3 *= mf.b3;
  }
  *zp = z;
}

template <int ACT>
struct M3Chain {
  const NetArgs<float>& a;
  float *A1, *D2, *A1T, *XS, *YS, *fpart, *rbuf, *red;
  const float *XALL, *YALL;                      // the whole data set in LDS ([k][NP] and [NP], zero padded) when it fits: NP > 0
  int NP;
  int tid, w, g, c, n_in, H1, H2, rq1, rq2;
#if M3_TIMING
  unsigned long long* tacc;                      // [20] in the kernel's frame; [19] = the last stamp
#endif
  int o_w1, o_b1, o_w2, o_b2, o_w3, o_b3;       // flat offsets (torch parameter order: W1, b1, W2, b2, W3, b3)
  bool jval, uval, vecd;                         // layer-1 unit / layer-2 unit 16 w + c exists; rows of W2 start on a multiple of 4 elements
  __device__ M3Chain(const NetArgs<float>& a_) : a(a_) {}

  M3Lay lay;
  // The W2 share of the MOMENTUM lives in global memory (this workgroup's slot of a workspace, [7][448] float4: one 16-byte
  // access per lane and block, L2 resident) except between its prefetch - issued under GEMM2 - and the end of the pass
  // (kick, drift, write back).  Left to the register allocator, those 28 registers were "spilled" across the pass anyway, to
  // scratch memory, reloaded in the kick by six SERIAL round trips to HBM (12 000 of a pass's 74 000 cycles, measured).
  V4f* pw;                                       // (uniform) this workgroup's slot
  // this lane's index into a block of the slot.  Opaque to the optimiser on purpose: seven loop-invariant 64-bit addresses
  // were hoisted out of the stage loop and then SPILLED - each access became "reload the address from scratch, wait, load".
  __device__ __forceinline__ int pw_lane() const {
    int t = tid;
    asm volatile("" : "+v"(t));
    return t;
  }
  __device__ __forceinline__ void park(const M3State& x) const {
    const int t = pw_lane();
#pragma unroll
    for (int s = 0; s < 7; ++s) pw[s * M3_NT + t] = x.w2[s];
  }
  __device__ __forceinline__ void unpark(M3State& x) const {
    const int t = pw_lane();
#pragma unroll
    for (int s = 0; s < 7; ++s) x.w2[s] = pw[s * M3_NT + t];
  }
  __device__ __forceinline__ bool ok(int s, int t) const { return lay.ok(s, t); }
  // loads, stores and draws run out of line (see M3Lay) and hand the state over through memory: once per trajectory
  __device__ __forceinline__ void load(const float* th, M3State& q, bool vec4) const { M3State t; m3_load(lay, th, vec4, &t); q = t; }
  __device__ __forceinline__ void store(float* th, const M3State& q, bool vec4) const { M3State t = q; m3_store(lay, th, vec4, &t); }
  // the momentum draw: its W2 share goes straight to the workspace, the thin layers' share into z (z.w2 is not used)
  __device__ __forceinline__ void draw(uint64_t chain, uint32_t n, M3State& z) const {
    M3State t;
    m3_draw(lay, a.seed, chain, n, a.mass_kind == HTA_MASS_DIAG ? a.mass_factor : nullptr, &t);
    park(t);
#pragma unroll
    for (int k = 0; k < M3_NIN; ++k) z.w1[k] = t.w1[k];
    z.b1 = t.b1; z.b2 = t.b2; z.w3 = t.w3; z.b3 = t.b3;
  }

  __device__ __forceinline__ float block_sum(float v) {
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) red[w] = v;
    __syncthreads();
    float s = 0;
#pragma unroll
    for (int i = 0; i < M3_NW; ++i) s += red[i];
    return s;
  }
  // sum over all parameters of f(x, y) with the copies of the thin layers counted once
  template <typename F> __device__ __forceinline__ float dot_like(const M3State& x, const M3State& y, F f) {
    float k = 0;
#pragma unroll
    for (int s = 0; s < 7; ++s)
#pragma unroll
      for (int t = 0; t < 4; ++t) k += f(x.w2[s][t], y.w2[s][t], 2);
    if (g == 0) {
#pragma unroll
      for (int i = 0; i < M3_NIN; ++i) k += f(x.w1[i], y.w1[i], 0);
      k += f(x.b1, y.b1, 1) + f(x.b2, y.b2, 3) + f(x.w3, y.w3, 4);
    }
    if (tid == 0) k += f(x.b3, y.b3, 5);
    return block_sum(k);
  }

  // Likelihood part over the points [lo, hi).  grad (uniform): with gg = d log p_m / d theta = likelihood gradient -
  // (tau / prior_scale) q (S:1156), pm = (pm + k1 gg) + k2 gg (the kick of the integrator, fused: the gradient never exists as
  // a second state vector - 36 registers; k2 is the half kick S:302 takes back).  Returns the sum of squared residuals of the
  // points (every thread) when !grad.  ONE call site in the kernel (a state machine drives it): the body is 560 matrix
  // instructions of straight-line code.
  __device__ __forceinline__ float pass(M3State& q, int lo, int hi, M3State& pm, float k1, float k2, float dr, bool grad) {
    V4f pv[7];                                // the momentum's W2 share between its load and its write-back
    V4f G[7];
    float gw1[M3_NIN] = {0, 0, 0, 0};
    float gb1 = 0, gb2 = 0, gw3 = 0, gb3 = 0, sse = 0;
#pragma unroll
    for (int s = 0; s < 7; ++s) G[s] = V4f{0, 0, 0, 0};
    const int u = 16 * w + c;                 // this lane's unit (layer 2) / unit (layer 1) / column of the delta1 tile
    const bool odd = c & 1, bit1 = c & 2;
    for (int c0 = lo; c0 < hi; c0 += M3_CP) {
      const int cnt = min(M3_CP, hi - c0);
      M3_TICK(0);
      __syncthreads();                        // everyone is done with A1 / D2 / XS / fpart of the previous chunk or pass
      M3_TICK(1);
      const float* xs = XS; const float* ys = YS;
      int xst = M3_CP;                        // x of input k, point i of the chunk: xs[k * xst + i]
      if (NP > 0) {                           // (uniform) the data set lives in LDS: points beyond the chunk are other points / zeros,
        xs = XALL + c0; ys = YALL + c0; xst = NP;        // finite either way, and their residuals are masked below
      } else {
        for (int i = tid; i < M3_CP; i += M3_NT) {
          const bool in = i < cnt;
#pragma unroll
          for (int k = 0; k < M3_NIN; ++k) XS[k * M3_CP + i] = (in && k < n_in) ? a.X[(size_t)(c0 + i) * n_in + k] : 0.0f;
          YS[i] = in ? a.Y[c0 + i] : 0.0f;
        }
        __syncthreads();
      }
      M3_TICK(2);
      // ---- layer 1 (element-wise): a1[p][j] for this lane's unit j = u and 4 consecutive points per tile, stored both ways
#pragma unroll
      for (int tp = 0; tp < 7; ++tp) {
        V4f z = {q.b1, q.b1, q.b1, q.b1};
#pragma unroll
        for (int k = 0; k < M3_NIN; ++k)
          if (k < n_in) {
            const V4f x4 = *reinterpret_cast<const V4f*>(xs + k * xst + 16 * tp + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) z[r] = fmaf(q.w1[k], x4[r], z[r]);
          }
#pragma unroll
        for (int r = 0; r < 4; ++r) z[r] = jval ? m3_act<ACT>(z[r]) : 0.0f;
        *reinterpret_cast<V4f*>(A1T + (size_t)u * M3_LDP + 16 * tp + 4 * g) = z;
        if (u < M3_LDJ) {
#pragma unroll
          for (int r = 0; r < 4; ++r) A1[(size_t)(16 * tp + 4 * g + r) * M3_LDJ + u] = z[r];
        }
      }
      M3_TICK(3);
      __syncthreads();
      M3_TICK(4);
      // ---- forward: Z[tp] (C layout: register r = point 16 tp + 4 g + r, unit u), bias in the accumulator
      V4f Z[7];
#pragma unroll
      for (int tp = 0; tp < 7; ++tp) Z[tp] = V4f{q.b2, q.b2, q.b2, q.b2};
      {
        const float* arow = A1 + (size_t)c * M3_LDJ + 4 * g;
#pragma unroll
        for (int s = 0; s < 6; ++s) {
          V4f af[7];
#pragma unroll
          for (int tp = 0; tp < 7; ++tp) af[tp] = *reinterpret_cast<const V4f*>(arow + (size_t)16 * tp * M3_LDJ + 16 * s);
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int tp = 0; tp < 7; ++tp) Z[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[tp][t], q.w2[s][t], Z[tp], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);          // keep the operand reads of block s + 1 out of block s - 1 (register pressure)
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
          if (t < rq1) {
#pragma unroll
            for (int tp = 0; tp < 7; ++tp)
              Z[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[(size_t)(16 * tp + c) * M3_LDJ + 96 + 4 * t + g], q.w2[6][t], Z[tp], 0, 0, 0);
          }
      }
      M3_TICK(5);
      // activations and this tile's share of f(x_p) = sum_u w3_u a2[p][u]: reduce-scatter over the quad, rotations by 4 and 8
      {
        float* fw = fpart + (size_t)w * M3_CP + 4 * g + (c & 3);
#pragma unroll
        for (int tp = 0; tp < 7; ++tp) {
          float fp[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) { Z[tp][r] = uval ? m3_act<ACT>(Z[tp][r]) : 0.0f; fp[r] = q.w3 * Z[tp][r]; }
          const float s01 = (odd ? fp[1] : fp[0]) + m3_dpp<0xB1>(odd ? fp[0] : fp[1]);
          const float s23 = (odd ? fp[3] : fp[2]) + m3_dpp<0xB1>(odd ? fp[2] : fp[3]);
          float sq = (bit1 ? s23 : s01) + m3_dpp<0x4E>(bit1 ? s01 : s23);
          sq += m3_dpp<0x124>(sq);
          sq += m3_dpp<0x128>(sq);
          if (c < 4) fw[16 * tp] = sq;
        }
      }
      M3_TICK(6);
      __syncthreads();                        // every wave is past its forward product: A1 is free
      M3_TICK(7);
      // once per point: delta_p = -tau_out r_p (grad) or r_p, r_p = b3 + sum over the unit tiles - y_p; 0 beyond the chunk
      for (int i = tid; i < M3_CP; i += M3_NT) {
        float f = q.b3;
#pragma unroll
        for (int tt = 0; tt < M3_NW; ++tt) f += fpart[tt * M3_CP + i];
        const float r = f - ys[i];
        rbuf[i] = (i < cnt) ? (grad ? -a.tau_out * r : r) : 0.0f;
        if (!grad && i < cnt) sse = fmaf(r, r, sse);
      }
      if (!grad) continue;                    // (uniform) the next chunk starts with a barrier
      // W2 -> staging copy [u][j] in the A1 region (rows u >= H2 and columns >= H1 are zeros): the B operand of delta1 = delta2 W2
      {
        float* srow = A1 + (size_t)u * M3_LDJ;
#pragma unroll
        for (int s = 0; s < 6; ++s) *reinterpret_cast<V4f*>(srow + 16 * s + 4 * g) = q.w2[s];
#pragma unroll
        for (int t = 0; t < 2; ++t) srow[96 + 4 * t + g] = q.w2[6][t];
      }
      M3_TICK(8);
      __syncthreads();
      M3_TICK(9);
      // ---- delta2 (in the forward accumulator's registers), the thin last layer's gradient, delta2 -> LDS for delta1
#pragma unroll
      for (int tp = 0; tp < 7; ++tp) {
        const V4f dl = *reinterpret_cast<const V4f*>(rbuf + 16 * tp + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float h = Z[tp][r];
          gb3 += dl[r];
          gw3 = fmaf(dl[r], h, gw3);
          const float d = (ACT == 0) ? (h > 0.0f ? dl[r] * q.w3 : 0.0f) : dl[r] * q.w3 * m3_dact<ACT>(h);
          gb2 += d;
          Z[tp][r] = d;
          if (u < M3_LDJ) D2[(size_t)(16 * tp + 4 * g + r) * M3_LDJ + u] = d;
        }
      }
      M3_TICK(10);
      // ---- dW2^T[j][u] += sum_p A1T[j][p] delta2[p][u]: the accumulators come out in the theta register layout
      {
        // rows of the last tile follow the remainder block's register order: row 4 g' + r' <-> j = 96 + 4 r' + g'
        const int jrow6 = 96 + 4 * (c & 3) + (c >> 2);
        const float* trow = A1T + (size_t)c * M3_LDP + 4 * g;
        const float* trow6 = A1T + (size_t)jrow6 * M3_LDP + 4 * g;
#pragma unroll
        for (int tp = 0; tp < 7; ++tp) {
          V4f af[7];
#pragma unroll
          for (int s = 0; s < 6; ++s) af[s] = *reinterpret_cast<const V4f*>(trow + (size_t)16 * s * M3_LDP + 16 * tp);
          af[6] = *reinterpret_cast<const V4f*>(trow6 + 16 * tp);
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int s = 0; s < 7; ++s) G[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s][t], Z[tp][t], G[s], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      M3_TICK(11);
      __syncthreads();                        // delta2 of every unit tile is in LDS
      M3_TICK(12);
      // ---- delta1[p][j = u] = sum_u' delta2[p][u'] W2[u'][j]  (before the activation derivative); the B operand streams from
      //      the staging copy, one block of 4 registers at a time (column reads, 4 per 28 matrix instructions)
      V4f E[7];
#pragma unroll
      for (int tp = 0; tp < 7; ++tp) E[tp] = V4f{0, 0, 0, 0};
      {
        const float* drow = D2 + (size_t)c * M3_LDJ + 4 * g;
        const float* scol = A1 + (size_t)(4 * g) * M3_LDJ + (u < M3_LDJ ? u : 0);
        const bool cj = u < H1;
#pragma unroll
        for (int s = 0; s < 6; ++s) {
          V4f af[7], bf;
#pragma unroll
          for (int t = 0; t < 4; ++t) { const float v = scol[(size_t)(16 * s + t) * M3_LDJ]; bf[t] = cj ? v : 0.0f; }
#pragma unroll
          for (int tp = 0; tp < 7; ++tp) af[tp] = *reinterpret_cast<const V4f*>(drow + (size_t)16 * tp * M3_LDJ + 16 * s);
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int tp = 0; tp < 7; ++tp) E[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[tp][t], bf[t], E[tp], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
          if (t < rq2) {
            const float v = A1[(size_t)(96 + 4 * t + g) * M3_LDJ + (u < M3_LDJ ? u : 0)];
            const float bf = cj ? v : 0.0f;
#pragma unroll
            for (int tp = 0; tp < 7; ++tp)
              E[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(D2[(size_t)(16 * tp + c) * M3_LDJ + 96 + 4 * t + g], bf, E[tp], 0, 0, 0);
          }
      }
      M3_TICK(13);
      // the thin first layer's gradient: db1[j] = sum_p delta1, dW1[j][k] = sum_p delta1 x[p][k]
#pragma unroll
      for (int tp = 0; tp < 7; ++tp) {
        const V4f a4 = *reinterpret_cast<const V4f*>(A1T + (size_t)u * M3_LDP + 16 * tp + 4 * g);
        V4f d1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          d1[r] = (ACT == 0) ? (a4[r] > 0.0f ? E[tp][r] : 0.0f) : E[tp][r] * m3_dact<ACT>(a4[r]);
          gb1 += d1[r];
        }
#pragma unroll
        for (int k = 0; k < M3_NIN; ++k)
          if (k < n_in) {
            const V4f x4 = *reinterpret_cast<const V4f*>(xs + k * xst + 16 * tp + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) gw1[k] = fmaf(d1[r], x4[r], gw1[k]);
          }
      }
    }
    M3_TICK(14);
    float ret = 0;
    if (grad) {
      // the momentum's W2 share comes back from the workspace: seven independent 16-byte loads, ONE round trip to the L2
      // (prefetching them under GEMM2 was tried: the register allocator answers by spilling the prefetched values at once)
      const int pwl = pw_lane();
#pragma unroll
      for (int s = 0; s < 7; ++s) pv[s] = pw[s * M3_NT + pwl];
      // W2 comes back from its staging copy (same rows, same columns as the registers it was written from): between the
      // staging write and here - delta2, GEMM3, GEMM2: the registers' busiest stretch - theta's 28 W2 registers are free
      {
        const float* srow = A1 + (size_t)u * M3_LDJ;
#pragma unroll
        for (int s = 0; s < 6; ++s) q.w2[s] = *reinterpret_cast<const V4f*>(srow + 16 * s + 4 * g);
#pragma unroll
        for (int t = 0; t < 2; ++t) q.w2[6][t] = srow[96 + 4 * t + g];
        q.w2[6][2] = 0.0f; q.w2[6][3] = 0.0f;
      }
      const float ips = 1.0f / a.prior_scale;
      const float t0 = ips * a.tau[0], t1 = ips * a.tau[1], t2 = ips * a.tau[2], t3 = ips * a.tau[3], t4 = ips * a.tau[4], t5 = ips * a.tau[5];
      auto kick = [&](float& pv, float gg) { pv = fmaf(k1, gg, pv); pv = fmaf(k2, gg, pv); };
#pragma unroll
      for (int s = 0; s < 7; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          float v = pv[s][t];
          kick(v, ok(s, t) ? fmaf(-t2, q.w2[s][t], G[s][t]) : 0.0f);
          pv[s][t] = v;
        }
#pragma unroll
      for (int k = 0; k < M3_NIN; ++k) { const float v = m3_groups_sum(gw1[k]); kick(pm.w1[k], (jval && k < n_in) ? fmaf(-t0, q.w1[k], v) : 0.0f); }
      const float vb1 = m3_groups_sum(gb1), vb2 = m3_groups_sum(gb2), vw3 = m3_groups_sum(gw3);
      kick(pm.b1, jval ? fmaf(-t1, q.b1, vb1) : 0.0f);
      kick(pm.b2, uval ? fmaf(-t3, q.b2, vb2) : 0.0f);
      kick(pm.w3, uval ? fmaf(-t4, q.w3, vw3) : 0.0f);
      kick(pm.b3, fmaf(-t5, q.b3, m3_groups_sum(gb3)));          // every lane group saw every point once: all lanes hold the same sum
      // the drift that follows this kick (q += dr M^-1 p), while the momentum is in registers; then the W2 share goes back
      if (dr != 0.0f) {
        if (a.mass_kind == HTA_MASS_DIAG) {
          M3State im;
          m3_load_inline(lay, a.inv_mass, vecd, im);
#pragma unroll
          for (int s = 0; s < 7; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t) q.w2[s][t] = fmaf(dr * im.w2[s][t], pv[s][t], q.w2[s][t]);
#pragma unroll
          for (int k = 0; k < M3_NIN; ++k) q.w1[k] = fmaf(dr * im.w1[k], pm.w1[k], q.w1[k]);
          q.b1 = fmaf(dr * im.b1, pm.b1, q.b1); q.b2 = fmaf(dr * im.b2, pm.b2, q.b2);
          q.w3 = fmaf(dr * im.w3, pm.w3, q.w3); q.b3 = fmaf(dr * im.b3, pm.b3, q.b3);
        } else {
#pragma unroll
          for (int s = 0; s < 7; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t) q.w2[s][t] = fmaf(dr, pv[s][t], q.w2[s][t]);
#pragma unroll
          for (int k = 0; k < M3_NIN; ++k) q.w1[k] = fmaf(dr, pm.w1[k], q.w1[k]);
          q.b1 = fmaf(dr, pm.b1, q.b1); q.b2 = fmaf(dr, pm.b2, q.b2); q.w3 = fmaf(dr, pm.w3, q.w3); q.b3 = fmaf(dr, pm.b3, q.b3);
        }
      }
#pragma unroll
      for (int s = 0; s < 7; ++s) pw[s * M3_NT + pwl] = pv[s];
    } else {
      ret = block_sum(sse);
    }
    M3_TICK(15);
    return ret;
  }

  // prior log-density (whole): sum over the six tensors of -1/2 tau sum w^2 + n (1/2 log tau - 1/2 log 2 pi)   (S:1143, S:1156)
  __device__ __forceinline__ float log_prior(const M3State& q) {
    const float qq = dot_like(q, q, [&](float x, float, int which) { return a.tau[which] * x * x; });
    const float hl2p = 0.9189385332046727f;
    const float n[6] = {(float)(H1 * n_in), (float)H1, (float)(H1 * H2), (float)H2, (float)H2, 1.0f};
    float cst = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) cst += n[i] * (0.5f * logf(a.tau[i]) - hl2p);
    return -0.5f * qq + cst;
  }
  // 1/2 p^T M^-1 p; the momentum's W2 share comes from the workspace
  __device__ __forceinline__ float kinetic(const M3State& psmall) {
    M3State p = psmall;
    unpark(p);
    if (a.mass_kind == HTA_MASS_DIAG) {
      M3State im;
      load(a.inv_mass, im, vecd);
      return 0.5f * dot_like(p, im, [](float x, float m, int) { return x * m * x; });
    }
    return 0.5f * dot_like(p, p, [](float x, float, int) { return x * x; });
  }
  static __device__ __forceinline__ void zero(M3State& y) {
#pragma unroll
    for (int s = 0; s < 7; ++s) y.w2[s] = V4f{0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < M3_NIN; ++k) y.w1[k] = 0.0f;
    y.b1 = y.b2 = y.w3 = y.b3 = 0.0f;
  }
  static __device__ __forceinline__ void axpy(M3State& y, float cc, const M3State& x) {       // y += c x
#pragma unroll
    for (int s = 0; s < 7; ++s)
#pragma unroll
      for (int t = 0; t < 4; ++t) y.w2[s][t] = fmaf(cc, x.w2[s][t], y.w2[s][t]);
#pragma unroll
    for (int k = 0; k < M3_NIN; ++k) y.w1[k] = fmaf(cc, x.w1[k], y.w1[k]);
    y.b1 = fmaf(cc, x.b1, y.b1); y.b2 = fmaf(cc, x.b2, y.b2); y.w3 = fmaf(cc, x.w3, y.w3); y.b3 = fmaf(cc, x.b3, y.b3);
  }
};

template <int ACT>
__global__ __launch_bounds__(M3_NT) void mlp3_mfma_kernel(NetArgs<float> a, int NP, float* pws) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  typedef M3Chain<ACT> Ch;
  Ch ch(a);
  const int tid = threadIdx.x, lane = tid & 63;
  ch.tid = tid;
  ch.w = __builtin_amdgcn_readfirstlane(tid >> 6);
  ch.g = lane >> 4; ch.c = lane & 15;
  ch.n_in = a.dims[0]; ch.H1 = a.dims[1]; ch.H2 = a.dims[2];
  const int H1 = ch.H1, H2 = ch.H2, n_in = ch.n_in;
  ch.rq1 = H1 > 100 ? 2 : (H1 > 96 ? 1 : 0);
  ch.rq2 = H2 > 100 ? 2 : (H2 > 96 ? 1 : 0);
  ch.o_w1 = 0; ch.o_b1 = H1 * n_in; ch.o_w2 = ch.o_b1 + H1; ch.o_b2 = ch.o_w2 + H2 * H1; ch.o_w3 = ch.o_b2 + H2; ch.o_b3 = ch.o_w3 + H2;
  const int D = ch.o_b3 + 1;
  ch.jval = 16 * ch.w + ch.c < H1;
  ch.uval = 16 * ch.w + ch.c < H2;
  ch.vecd = (H1 % 4 == 0) && (ch.o_w2 % 4 == 0);        // a row of W2 starts on a multiple of four ELEMENTS of the flat vector
  ch.lay = M3Lay{ch.w, ch.g, ch.c, n_in, H1, H2, ch.rq1, ch.o_w1, ch.o_b1, ch.o_w2, ch.o_b2, ch.o_w3, ch.o_b3, ch.jval, ch.uval, ch.vecd};
  ch.pw = reinterpret_cast<V4f*>(pws) + (size_t)blockIdx.x * 7 * M3_NT;
  float* base = reinterpret_cast<float*>(smem_raw);
  ch.A1 = base;
  ch.D2 = ch.A1 + (size_t)M3_CP * M3_LDJ;
  ch.A1T = ch.D2 + (size_t)M3_CP * M3_LDJ;
  ch.XS = ch.A1T + (size_t)M3_CP * M3_LDP;
  ch.YS = ch.XS + M3_NIN * M3_CP;
  ch.fpart = ch.YS + M3_CP;
  ch.rbuf = ch.fpart + M3_NW * M3_CP;
  ch.red = ch.rbuf + M3_CP;
  int* perm = reinterpret_cast<int*>(ch.red + 16);
  float* xall = ch.red + 16 + 64;
  ch.NP = NP; ch.XALL = xall; ch.YALL = xall + (size_t)n_in * NP;
  if (NP > 0) {                            // the data set, once per workgroup: [k][NP] inputs and [NP] targets, zeros beyond N
    for (int i = tid; i < NP; i += M3_NT) {
      const bool in = i < a.N;
      for (int k = 0; k < n_in; ++k) xall[(size_t)k * NP + i] = in ? a.X[(size_t)i * n_in + k] : 0.0f;
      xall[(size_t)n_in * NP + i] = in ? a.Y[i] : 0.0f;
    }
  }

  enum { EV_GRAD, EV_LOGP, LOGP_INIT, LOGP_END, LOGP_RESET };
  const float eps = a.eps, heps = 0.5f * a.eps;
  const int M = a.M;
  const int nstage = split_stage_count(a.integ, M, a.L);
  const bool plain = M == 1 && a.integ == HTA_SPLIT_SYMMETRIC;

  for (int64_t cidx = blockIdx.x; cidx < a.C; cidx += gridDim.x) {
    const uint64_t chain = a.chain_offset + (uint64_t)cidx;
    // 16-byte global accesses to this chain's row need (cidx D) % 4 == 0 on top of vecd (D = 10401 is odd for the notebook
    // model: three chains of four go element by element - once per trajectory, against hundreds of gradient passes)
    const bool vrow = ch.vecd && ((cidx * (int64_t)D) % 4 == 0);
    float* const cur_row = a.theta + cidx * D;          // the chain's current state lives HERE between trajectories, not in registers
    M3State q, p;
    ch.load(cur_row, q, vrow);
    Ch::zero(p);
    ch.park(p);
    __syncthreads();
#if M3_TIMING
    unsigned long long tacc_store[20] = {0};
    ch.tacc = tacc_store;
    tacc_store[19] = __builtin_readcyclecounter();
#endif

    // TWO call sites of the likelihood pass.  The rare one, driven by a small state machine: evaluation-only (gradient, then
    // value, of one split closure: the parity tests), the full-data log p of a trajectory's start / end point (S:971, S:995)
    // and of params_init after a Q2 reset (S:1018).  The hot one: the stage loop of a trajectory (S:499-596 / S:281-302), whose
    // live state is (q, p) and nothing else.
    int mode = a.n_traj == 0 ? EV_GRAD : LOGP_INIT;
    int tr = 0, n = a.traj_offset;
    float lp_cur = 0, h_old = 0, h_new = 0;
    bool acc = false;
    int32_t rejected = 0;
    for (bool done = false; !done;) {
      const bool ev = mode == EV_GRAD || mode == EV_LOGP;
      const int lo = ev ? a.eval_split * a.Nb : 0, hi = ev ? lo + a.Nb : M * a.Nb;
      const float sse = ch.pass(q, lo, hi, p, 1.0f, 0.0f, 0.0f, mode == EV_GRAD);
      bool finish = false, begin = false;
      if (mode == EV_GRAD) {
        ch.unpark(p);                                                        // (the gradient: 0 + 1 * gg, W2 share in the workspace)
        if (a.grad_out) ch.store(a.grad_out + cidx * D, p, vrow);
        mode = EV_LOGP;
      } else if (mode == EV_LOGP) {
        const float lp = -0.5f * a.tau_out * sse + ch.log_prior(q) / a.prior_scale;
        if (a.logp_out && tid == 0) a.logp_out[cidx] = lp;
        done = true;
      } else {
        // sum_m log p_m(theta) = full-data log-likelihood + (M / prior_scale) * prior   (S:787-796)
        const float lp = -0.5f * a.tau_out * sse + ((float)M / a.prior_scale) * ch.log_prior(q);
        if (mode == LOGP_INIT) { lp_cur = lp; begin = true; }
        else if (mode == LOGP_RESET) { lp_cur = lp; ch.store(cur_row, q, vrow); finish = true; }
        else {                                                               // LOGP_END: S:995-1026
          h_new = -lp + ch.kinetic(p);
          const float uu = u23<float>(philox_block(a.seed, chain, (uint32_t)n, PURPOSE_MH, 0, 0).x);
          acc = mh_accept<float>(h_old, h_new, lp, uu);                      // S:1000-1004
          finish = true;
          if (acc) { lp_cur = lp; ch.store(cur_row, q, vrow); }
          else {
            ++rejected;
            if (n == a.burn + 1) {                                           // Q2 reset to params_init (S:1018): its log p first
              ch.load(a.theta_init + cidx * D, q, vrow);
              mode = LOGP_RESET;
              finish = false;
            } else {
              ch.load(cur_row, q, vrow);
            }
          }
        }
      }
      if (finish) {
        if (a.samples && n > a.burn) {
          const int64_t off = ((int64_t)(n - a.burn) * a.C + cidx) * D;
          ch.store(a.samples + off, q, ch.vecd && (off % 4 == 0));
        }
        if (tid == 0) {
          if (a.H_old) a.H_old[(int64_t)tr * a.C + cidx] = h_old;
          if (a.H_new) a.H_new[(int64_t)tr * a.C + cidx] = h_new;
          if (a.accept) a.accept[(int64_t)tr * a.C + cidx] = acc ? 1 : 0;
        }
        if (++tr == a.n_traj) done = true; else begin = true;
      }
      if (begin) {
        n = a.traj_offset + tr;
        ch.draw(chain, (uint32_t)n, p);                                      // gibbs (S:185-186 / S:200-201)
        h_old = -lp_cur + ch.kinetic(p);                                     // S:971
        if (a.integ == HTA_SPLIT_RAND) {                                     // S:549: one subset order per trajectory
          __syncthreads();
          if (tid == 0) split_permutation(a.seed, (uint32_t)n, M, perm);
          __syncthreads();
        }
#if M3_TIMING
        { const unsigned long long now_ = __builtin_readcyclecounter(); ch.tacc[16] += now_ - ch.tacc[19]; ch.tacc[19] = now_; }
#endif
        {
          // ---- the stage loop (S:499-596 / S:281-302) on LOCAL copies of the state: q and p are live across the out-of-line
          // calls of the state machine, which makes the register allocator give parts of them a home in scratch memory and
          // touch it in every pass; the copies' live ranges cross no call
          M3State ql = q, pl = p;
          for (int st = 0; st < nstage; ++st) {
            int m; float kick, dr;
            split_stage<float>(a.integ, M, a.L, st, eps, perm, m, kick, dr);
            // plain leapfrog: a full kick at the last step, half of it taken back (S:298, S:302); the drift runs inside the pass
            float k2 = (plain && st == nstage - 1) ? -heps : 0.0f;
            if (!plain && st + 1 < nstage) {
              // the next stage evaluates the same subset at the same parameters (this one does not drift): both kicks from this
              // gradient, (p + k1 g) + k2 g, and the next stage's drift (mlp.hpp: split_stage_reuses)
              int m2; float kick2, dr2;
              split_stage<float>(a.integ, M, a.L, st + 1, eps, perm, m2, kick2, dr2);
              if (split_stage_reuses<float>(m, dr, m2)) { k2 = kick2; dr = dr2; ++st; }
            }
            ch.pass(ql, m * a.Nb, m * a.Nb + a.Nb, pl, kick, k2, dr, true);
#if M3_TIMING
            { const unsigned long long now_ = __builtin_readcyclecounter(); ch.tacc[17] += now_ - ch.tacc[19]; ch.tacc[19] = now_; }
#endif
          }
          q = ql; p = pl;
        }
        mode = LOGP_END;
      }
    }
    if (tid == 0 && a.reject_count && a.n_traj > 0) a.reject_count[cidx] += rejected;
#if M3_TIMING
    if (tid == 0 && blockIdx.x == 0) for (int k = 0; k < 20; ++k) hta_m3_dbg[k] = tacc_store[k];
#endif
  }
}

static size_t mlp3_lds_floats() {          // without the data set
  return (size_t)2 * M3_CP * M3_LDJ + (size_t)M3_CP * M3_LDP + M3_NIN * M3_CP + M3_CP + M3_NW * M3_CP + M3_CP + 16 + 64;
}
// padded length of the LDS copy of the data set, or 0 when it does not fit (or the 16-byte reads of a chunk would be
// misaligned: every chunk starts at m Nb + 112 i): then every pass stages its chunk from global memory
static int mlp3_data_np(const NetArgs<float>& a) {
  const int NP = (a.N + 3) / 4 * 4 + M3_CP;
  if (a.Nb % 4 != 0) return 0;
  return (mlp3_lds_floats() + (size_t)(a.dims[0] + 1) * NP) * sizeof(float) <= 160 * 1024 ? NP : 0;
}

bool mlp3_eligible(const NetArgs<float>& a) {
  if (!g_mlp3_route) return false;
  if (a.n_layers != 3 || a.dims[3] != 1 || a.loss != HTA_LOSS_REGRESSION) return false;
  if (a.dims[0] < 1 || a.dims[0] > M3_NIN || a.dims[1] < 1 || a.dims[1] > M3_HMAX || a.dims[2] < 1 || a.dims[2] > M3_HMAX) return false;
  if (!(a.mass_kind == HTA_MASS_NONE || a.mass_kind == HTA_MASS_DIAG)) return false;
  // the small-net kernel (one wave per chain) keeps what it can hold: widths <= 64 and <= 512 parameters
  const int D = a.dims[0] * a.dims[1] + a.dims[1] + a.dims[1] * a.dims[2] + 2 * a.dims[2] + 1;
  return a.dims[1] > NETN_MAX_WIDTH || a.dims[2] > NETN_MAX_WIDTH || D > 64 * NETN_KMAX || g_mlp3_route == 2;
}

// The momentum workspace (see M3Chain::pw): one slot of 7 x 448 float4 per workgroup.  ABI 10: it is the CALLER's memory
// (hta_netn_hmc_workspace_bytes -> the `workspace` argument of hta_netn_hmc_sample / hta_netn_logp_grad); rounds 3-4 kept a
// hipMalloc'ed buffer per (device, stream) behind the ABI, grown with a hipStreamSynchronize + hipFree - an allocation and a
// synchronisation the boundary promises not to make, and a launch that could not be captured in a HIP graph.
static int mlp3_grid(int64_t C) {
  int cus = 256;
  int dev = 0;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  // one workgroup per CU is resident (151 KB of LDS); twice that many in the grid keeps the tail short, each workgroup walks
  // its chains in a grid-stride loop
  const int64_t gmax = 2 * (int64_t)cus;
  return (int)(C < gmax ? C : gmax);
}
int64_t mlp3_workspace_bytes(int64_t C, int n_layers, const int* dims) {
  if (n_layers != 3 || !dims || C <= 0 || dims[3] != 1) return 0;
  if (dims[0] < 1 || dims[0] > M3_NIN || dims[1] < 1 || dims[1] > M3_HMAX || dims[2] < 1 || dims[2] > M3_HMAX) return 0;
  const int D = dims[0] * dims[1] + dims[1] + dims[1] * dims[2] + 2 * dims[2] + 1;
  if (!(dims[1] > NETN_MAX_WIDTH || dims[2] > NETN_MAX_WIDTH || D > 64 * NETN_KMAX || g_mlp3_route == 2)) return 0;   // mlp3_eligible's size test
  return (int64_t)mlp3_grid(C) * 7 * M3_NT * 4 * (int64_t)sizeof(float);
}

template <int ACT> static int launch_mlp3(const NetArgs<float>& a, hipStream_t s) {
  static DevOnce done;
  const int NP = mlp3_data_np(a);
  const size_t lds = (mlp3_lds_floats() + (size_t)(a.dims[0] + 1) * NP) * sizeof(float);
  if (!done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp3_mfma_kernel<ACT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { set_error("hta_netn_hmc (mlp3): hipFuncSetAttribute: %s", hipGetErrorString(e)); return HTA_ERR_LAUNCH; }
    done = true;
  }
  const int grid = mlp3_grid(a.C);
  const int64_t need = (int64_t)grid * 7 * M3_NT * 4 * (int64_t)sizeof(float);
  HTA_REQUIRE(a.workspace && a.workspace_bytes >= need, "hta_netn_hmc (mlp3): workspace of %lld bytes required (hta_netn_hmc_workspace_bytes)",
              (long long)need);
  float* pws = static_cast<float*>(a.workspace);
  profile_begin(s);
  note_route("mlp3_mfma_kernel<%d>", ACT);
  mlp3_mfma_kernel<ACT><<<grid, M3_NT, lds, s>>>(a, NP, pws);
  profile_end(s);
  HTA_CHECK_LAUNCH("hta_netn_hmc (mlp3)");
  return HTA_OK;
}

int mlp3_mfma(const NetArgs<float>& a, hipStream_t s) {
  HTA_REQUIRE(a.theta && a.X && a.Y && a.C > 0, "hta_netn_hmc: NULL pointer / empty batch");
  HTA_REQUIRE(a.act >= 0 && a.act <= 2, "hta_netn_hmc: unknown activation %d", a.act);
  HTA_REQUIRE(a.M >= 1 && a.Nb >= 1 && (int64_t)a.M * a.Nb <= a.N, "hta_netn_hmc: M=%d splits of Nb=%d points exceed N=%d", a.M, a.Nb, a.N);
  HTA_REQUIRE(a.mass_kind == HTA_MASS_NONE || (a.inv_mass && a.mass_factor), "hta_netn_hmc: diagonal mass needs inv_mass and mass_factor");
  if (a.n_traj > 0) HTA_REQUIRE(a.theta_init && a.L >= 0, "hta_netn_hmc: bad trajectory arguments");
  if (a.n_traj == 0) HTA_REQUIRE(a.eval_split >= 0 && a.eval_split < a.M, "hta_netn_logp_grad: split %d not in [0, %d)", a.eval_split, a.M);
  HTA_REQUIRE(a.integ >= HTA_SPLIT_SYMMETRIC && a.integ <= HTA_SPLIT_KMID, "hta_netn_hmc: unknown integrator %d", a.integ);
  HTA_REQUIRE(a.integ != HTA_SPLIT_RAND || a.M <= 64, "hta_netn_hmc: SPLITTING_RAND supports at most 64 subsets natively (M=%d)", a.M);
  HTA_REQUIRE(a.integ != HTA_SPLIT_KMID || a.M >= 2, "hta_netn_hmc: SPLITTING_KMID needs at least 2 subsets");
  if (a.act == 0) return launch_mlp3<0>(a, s);
  if (a.act == 1) return launch_mlp3<1>(a, s);
  return launch_mlp3<2>(a, s);
}

}  // namespace hta
