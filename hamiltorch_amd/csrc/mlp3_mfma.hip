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
#include "netn.hpp"
#include "mlp.hpp"
#include "philox.hpp"

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

template <int ACT>
struct M3Chain {
  const NetArgs<float>& a;
  float *A1, *D2, *A1T, *XS, *YS, *fpart, *rbuf, *red, *dump;
  int tid, w, g, c, n_in, H1, H2, rq1, rq2;
  int o_w1, o_b1, o_w2, o_b2, o_w3, o_b3;       // flat offsets (torch parameter order: W1, b1, W2, b2, W3, b3)
  bool jval, uval, vecd;                         // layer-1 unit / layer-2 unit 16 w + c exists; rows of W2 start on a multiple of 4 elements
  __device__ M3Chain(const NetArgs<float>& a_) : a(a_) {}

  // column of register [s][t] and whether the element exists
  __device__ __forceinline__ int col(int s, int t) const { return s < 6 ? 16 * s + 4 * g + t : 96 + 4 * t + g; }
  __device__ __forceinline__ bool ok(int s, int t) const { return uval && col(s, t) < H1 && (s < 6 || t < rq1); }

  // element idx of `base` if it exists, else 0 - an unconditional load from a clamped address and a select (no branch per element)
  static __device__ __forceinline__ float pick(const float* base, int idx, bool exists) {
    const float v = base[exists ? idx : 0];
    return exists ? v : 0.0f;
  }
  // vec4: th + o_w2 + u H1 is 16-byte aligned for every row (the caller knows the base's alignment)
  __device__ __forceinline__ void load(const float* th, M3State& q, bool vec4) const {
    const int u = 16 * w + c;
    const float* row = th + o_w2 + (size_t)(uval ? u : 0) * H1;
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      const int j0 = 16 * s + 4 * g;
      if (vec4 && uval && j0 + 4 <= H1) q.w2[s] = *reinterpret_cast<const V4f*>(row + j0);
      else
#pragma unroll
        for (int t = 0; t < 4; ++t) q.w2[s][t] = pick(row, j0 + t, ok(s, t));
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) q.w2[6][t] = pick(row, col(6, t), ok(6, t));
    const int j1 = jval ? 16 * w + c : 0, u2 = uval ? u : 0;
#pragma unroll
    for (int k = 0; k < M3_NIN; ++k) q.w1[k] = pick(th + o_w1 + j1 * n_in, k, jval && k < n_in);
    q.b1 = pick(th + o_b1, j1, jval);
    q.b2 = pick(th + o_b2, u2, uval);
    q.w3 = pick(th + o_w3, u2, uval);
    q.b3 = th[o_b3];
  }
  // every copy lane stores (same value, same address): a lane later reloads exactly what it stored itself
  __device__ __forceinline__ void store(float* th, const M3State& q, bool vec4) const {
    const int u = 16 * w + c;
    float* row = th + o_w2 + (size_t)(uval ? u : 0) * H1;
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      const int j0 = 16 * s + 4 * g;
      if (vec4 && uval && j0 + 4 <= H1) *reinterpret_cast<V4f*>(row + j0) = q.w2[s];
      else
#pragma unroll
        for (int t = 0; t < 4; ++t) if (ok(s, t)) row[j0 + t] = q.w2[s][t];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) if (ok(6, t)) row[col(6, t)] = q.w2[6][t];
#pragma unroll
    for (int k = 0; k < M3_NIN; ++k) if (jval && k < n_in) th[o_w1 + (16 * w + c) * n_in + k] = q.w1[k];
    if (jval) th[o_b1 + 16 * w + c] = q.b1;
    if (uval) { th[o_b2 + u] = q.b2; th[o_w3 + u] = q.w3; }
    th[o_b3] = q.b3;
  }
  // standard-normal draw in the state layout (element i of the D-vector = Philox block i / 4, slot i % 4: the oracle's stream)
  __device__ __forceinline__ void draw(uint64_t chain, uint32_t n, M3State& z) const {
    const int u = 16 * w + c;
    const int rowoff = o_w2 + (uval ? u : 0) * H1;
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      const int j0 = 16 * s + 4 * g;
      if (vecd && uval && j0 + 4 <= H1) {                       // four consecutive elements = one Philox block
        float zz[4];
        normal4<float>(philox_block(a.seed, chain, n, PURPOSE_MOMENTUM, 0, (uint32_t)((rowoff + j0) >> 2)), zz);
#pragma unroll
        for (int t = 0; t < 4; ++t) z.w2[s][t] = zz[t];
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) { const float v = normal_elem<float>(a.seed, chain, n, 0, rowoff + j0 + t); z.w2[s][t] = ok(s, t) ? v : 0.0f; }
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) { const float v = normal_elem<float>(a.seed, chain, n, 0, rowoff + col(6, t)); z.w2[6][t] = ok(6, t) ? v : 0.0f; }
    const int j1 = jval ? 16 * w + c : 0, u2 = uval ? u : 0;
#pragma unroll
    for (int k = 0; k < M3_NIN; ++k) {
      float v = 0.0f;
      if (k < n_in) v = normal_elem<float>(a.seed, chain, n, 0, o_w1 + j1 * n_in + k);      // (uniform branch)
      z.w1[k] = jval ? v : 0.0f;
    }
    { const float v = normal_elem<float>(a.seed, chain, n, 0, o_b1 + j1); z.b1 = jval ? v : 0.0f; }
    { const float v = normal_elem<float>(a.seed, chain, n, 0, o_b2 + u2); z.b2 = uval ? v : 0.0f; }
    { const float v = normal_elem<float>(a.seed, chain, n, 0, o_w3 + u2); z.w3 = uval ? v : 0.0f; }
    z.b3 = normal_elem<float>(a.seed, chain, n, 0, o_b3);
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
  __device__ __forceinline__ float pass(const M3State& q, int lo, int hi, M3State& pm, float k1, float k2, bool grad) {
    V4f G[7];
    float gw1[M3_NIN] = {0, 0, 0, 0};
    float gb1 = 0, gb2 = 0, gw3 = 0, gb3 = 0, sse = 0;
#pragma unroll
    for (int s = 0; s < 7; ++s) G[s] = V4f{0, 0, 0, 0};
    const int u = 16 * w + c;                 // this lane's unit (layer 2) / unit (layer 1) / column of the delta1 tile
    const bool odd = c & 1, bit1 = c & 2;
    for (int c0 = lo; c0 < hi; c0 += M3_CP) {
      const int cnt = min(M3_CP, hi - c0);
      __syncthreads();                        // everyone is done with A1 / D2 / XS / fpart of the previous chunk or pass
      for (int i = tid; i < M3_CP; i += M3_NT) {
        const bool in = i < cnt;
#pragma unroll
        for (int k = 0; k < M3_NIN; ++k) XS[k * M3_CP + i] = (in && k < n_in) ? a.X[(size_t)(c0 + i) * n_in + k] : 0.0f;
        YS[i] = in ? a.Y[c0 + i] : 0.0f;
      }
      __syncthreads();
      // ---- layer 1 (element-wise): a1[p][j] for this lane's unit j = u and 4 consecutive points per tile, stored both ways
#pragma unroll
      for (int tp = 0; tp < 7; ++tp) {
        V4f z = {q.b1, q.b1, q.b1, q.b1};
#pragma unroll
        for (int k = 0; k < M3_NIN; ++k)
          if (k < n_in) {
            const V4f x4 = *reinterpret_cast<const V4f*>(XS + k * M3_CP + 16 * tp + 4 * g);
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
      __syncthreads();
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
      // activations and this tile's share of f(x_p) = sum_u w3_u a2[p][u]: reduce-scatter over the quad, rotations by 4 and 8
      {
        float* fw = (c < 4) ? fpart + (size_t)w * M3_CP + 4 * g + c : dump + (tid & 63);
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
          fw[16 * tp] = sq;
        }
      }
      __syncthreads();                        // every wave is past its forward product: A1 is free
      // once per point: delta_p = -tau_out r_p (grad) or r_p, r_p = b3 + sum over the unit tiles - y_p; 0 beyond the chunk
      for (int i = tid; i < M3_CP; i += M3_NT) {
        float f = q.b3;
#pragma unroll
        for (int tt = 0; tt < M3_NW; ++tt) f += fpart[tt * M3_CP + i];
        const float r = f - YS[i];
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
      __syncthreads();
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
      __syncthreads();                        // delta2 of every unit tile is in LDS
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
            const V4f x4 = *reinterpret_cast<const V4f*>(XS + k * M3_CP + 16 * tp + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) gw1[k] = fmaf(d1[r], x4[r], gw1[k]);
          }
      }
    }
    float ret = 0;
    if (grad) {
      const float ips = 1.0f / a.prior_scale;
      const float t0 = ips * a.tau[0], t1 = ips * a.tau[1], t2 = ips * a.tau[2], t3 = ips * a.tau[3], t4 = ips * a.tau[4], t5 = ips * a.tau[5];
      auto kick = [&](float& pv, float gg) { pv = fmaf(k1, gg, pv); pv = fmaf(k2, gg, pv); };
#pragma unroll
      for (int s = 0; s < 7; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          float pv = pm.w2[s][t];
          kick(pv, ok(s, t) ? fmaf(-t2, q.w2[s][t], G[s][t]) : 0.0f);
          pm.w2[s][t] = pv;
        }
#pragma unroll
      for (int k = 0; k < M3_NIN; ++k) { const float v = m3_groups_sum(gw1[k]); kick(pm.w1[k], (jval && k < n_in) ? fmaf(-t0, q.w1[k], v) : 0.0f); }
      const float vb1 = m3_groups_sum(gb1), vb2 = m3_groups_sum(gb2), vw3 = m3_groups_sum(gw3);
      kick(pm.b1, jval ? fmaf(-t1, q.b1, vb1) : 0.0f);
      kick(pm.b2, uval ? fmaf(-t3, q.b2, vb2) : 0.0f);
      kick(pm.w3, uval ? fmaf(-t4, q.w3, vw3) : 0.0f);
      kick(pm.b3, fmaf(-t5, q.b3, m3_groups_sum(gb3)));          // every lane group saw every point once: all lanes hold the same sum
    } else {
      ret = block_sum(sse);
    }
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
  __device__ __forceinline__ float kinetic(const M3State& p) {
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
  __device__ __forceinline__ void drift(M3State& q, float cc, const M3State& p) {              // q += c M^-1 p
    if (a.mass_kind == HTA_MASS_DIAG) {
      M3State im;
      load(a.inv_mass, im, vecd);
#pragma unroll
      for (int s = 0; s < 7; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) q.w2[s][t] = fmaf(cc * im.w2[s][t], p.w2[s][t], q.w2[s][t]);
#pragma unroll
      for (int k = 0; k < M3_NIN; ++k) q.w1[k] = fmaf(cc * im.w1[k], p.w1[k], q.w1[k]);
      q.b1 = fmaf(cc * im.b1, p.b1, q.b1); q.b2 = fmaf(cc * im.b2, p.b2, q.b2);
      q.w3 = fmaf(cc * im.w3, p.w3, q.w3); q.b3 = fmaf(cc * im.b3, p.b3, q.b3);
    } else {
      axpy(q, cc, p);
    }
  }
};

template <int ACT>
__global__ __launch_bounds__(M3_NT) void mlp3_mfma_kernel(NetArgs<float> a) {
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
  float* base = reinterpret_cast<float*>(smem_raw);
  ch.A1 = base;
  ch.D2 = ch.A1 + (size_t)M3_CP * M3_LDJ;
  ch.A1T = ch.D2 + (size_t)M3_CP * M3_LDJ;
  ch.XS = ch.A1T + (size_t)M3_CP * M3_LDP;
  ch.YS = ch.XS + M3_NIN * M3_CP;
  ch.fpart = ch.YS + M3_CP;
  ch.rbuf = ch.fpart + M3_NW * M3_CP;
  ch.red = ch.rbuf + M3_CP;
  ch.dump = ch.red + 16 + (size_t)ch.w * (64 + M3_CP);
  int* perm = reinterpret_cast<int*>(ch.red + 16 + (size_t)M3_NW * (64 + M3_CP));

  enum { EV_GRAD, EV_LOGP, LOGP_INIT, GRAD, LOGP_END, LOGP_RESET };
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
    __syncthreads();

    // One call site of the likelihood pass, driven by a state machine: evaluation-only (gradient, then value, of one split
    // closure: the parity tests), or per trajectory nstage gradient passes (S:499-596 / S:281-302) and a full-data log p pass
    // (S:995), plus one more after a Q2 reset (S:1018).
    int mode = a.n_traj == 0 ? EV_GRAD : LOGP_INIT;
    int tr = 0, st = 0, n = a.traj_offset;
    float lp_cur = 0, h_old = 0, h_new = 0;
    bool acc = false;
    int32_t rejected = 0;
    for (bool done = false; !done;) {
      int lo = 0, hi = M * a.Nb;
      float k1 = 0, k2 = 0, dr = 0;
      bool grad = false;
      if (mode == EV_GRAD || mode == EV_LOGP) {
        lo = a.eval_split * a.Nb; hi = lo + a.Nb;
        grad = mode == EV_GRAD; k1 = 1.0f;
      } else if (mode == GRAD) {
        int m;
        split_stage<float>(a.integ, M, a.L, st, eps, perm, m, k1, dr);
        lo = m * a.Nb; hi = lo + a.Nb;
        grad = true;
        k2 = (plain && st == nstage - 1) ? -heps : 0.0f;    // plain leapfrog: a full kick at the last step, half of it taken back (S:298, S:302)
      }
      const float sse = ch.pass(q, lo, hi, p, k1, k2, grad);
      bool finish = false, begin = false;
      if (mode == EV_GRAD) {
        if (a.grad_out) ch.store(a.grad_out + cidx * D, p, vrow);
        mode = EV_LOGP;
      } else if (mode == EV_LOGP) {
        const float lp = -0.5f * a.tau_out * sse + ch.log_prior(q) / a.prior_scale;
        if (a.logp_out && tid == 0) a.logp_out[cidx] = lp;
        done = true;
      } else if (mode == GRAD) {
        if (dr != 0.0f) ch.drift(q, dr, p);
        if (++st == nstage) mode = LOGP_END;
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
        // ---- gibbs (S:185-186 / S:200-201): p = sqrt(M) z; the Philox element index is the flat parameter index
        ch.draw(chain, (uint32_t)n, p);
        if (a.mass_kind == HTA_MASS_DIAG) {
          M3State mf;
          ch.load(a.mass_factor, mf, ch.vecd);
#pragma unroll
          for (int s = 0; s < 7; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t) p.w2[s][t] *= mf.w2[s][t];
#pragma unroll
          for (int k = 0; k < M3_NIN; ++k) p.w1[k] *= mf.w1[k];
          p.b1 *= mf.b1; p.b2 *= mf.b2; p.w3 *= mf.w3; p.b3 *= mf.b3;
        }
        h_old = -lp_cur + ch.kinetic(p);                                     // S:971
        if (a.integ == HTA_SPLIT_RAND) {                                     // S:549: one subset order per trajectory
          __syncthreads();
          if (tid == 0) split_permutation(a.seed, (uint32_t)n, M, perm);
          __syncthreads();
        }
        st = 0;
        mode = nstage > 0 ? GRAD : LOGP_END;
      }
    }
    if (tid == 0 && a.reject_count && a.n_traj > 0) a.reject_count[cidx] += rejected;
  }
}

static size_t mlp3_lds_bytes() {
  return ((size_t)2 * M3_CP * M3_LDJ + (size_t)M3_CP * M3_LDP + M3_NIN * M3_CP + M3_CP + M3_NW * M3_CP + M3_CP + 16 +
          (size_t)M3_NW * (64 + M3_CP) + 64) * sizeof(float);
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

template <int ACT> static int launch_mlp3(const NetArgs<float>& a, hipStream_t s) {
  static DevOnce done;
  const size_t lds = mlp3_lds_bytes();
  if (!done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp3_mfma_kernel<ACT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { set_error("hta_netn_hmc (mlp3): hipFuncSetAttribute: %s", hipGetErrorString(e)); return HTA_ERR_LAUNCH; }
    done = true;
  }
  const int grid = (int)(a.C < 8192 ? a.C : 8192);
  profile_begin(s);
  note_route("mlp3_mfma_kernel<%d>", ACT);
  mlp3_mfma_kernel<ACT><<<grid, M3_NT, lds, s>>>(a);
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
