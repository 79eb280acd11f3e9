// Posterior predictive of a fully connected net for MANY parameter vectors at once: out[s, p, :] = f(x_p; theta_s).
//
// Replaces the forward passes of hamiltorch.predict_model (hamiltorch/samplers.py:1468-1562: a Python loop over the samples,
// one functional forward each, S:1530-1552) - SURVEY 8(f) N2, "the step right after sampling in every BNN notebook".  theta_s
// is a row of the samples sample_model / sample_split_model returned: the flattened parameters in `model.parameters()` order
// (U:121-122): for every Linear(in, out) its weight [out, in] row-major, then its bias [out].
//
// One wave per (sample, 64 points): lane = point.  The weights of a sample are wave-uniform (scalar 16-byte loads through the
// constant cache, broadcast into the FMA as SGPR operands), a lane's activations of the current and the next layer sit in
// LDS as [unit / 4][lane][4] (one conflict-free 16-byte read = four consecutive units), eight output units accumulate in
// registers per pass over the inputs.  Forward only: 2 P flops per (sample, point), P weights - the whole posterior predictive
// of the reference's notebook model (1000 samples x 400 points x 10 401 weights) is 8 GFLOP: 2.2 ms for predict_model end to
// end (forward + log-probs + host), against 4.5 ms on the torch path (vmap of the closure) and the reference's Python loop over
// the samples.  Widths <= 256; a net with ONE hidden layer wider than that streams it (the output accumulates while the hidden
// units are produced) and has no width limit.
#include <math.h>
#include "common.hpp"

namespace hta {

constexpr int FW_TPB = 64, FW_MAXL = 8, FW_MAXW = 256, FW_MAXO = 16, FW_JB = 8;
struct FwDims { int n[FW_MAXL + 1]; };

template <typename T> __device__ __forceinline__ T fw_act(T z, int act) {
  if (act == 0) return z > (T)0 ? z : (T)0;                 // relu
  if (act == 1) return tanh(z);
  return (T)1 / ((T)1 + exp(-z));                           // sigmoid
}

template <typename T>
__global__ __launch_bounds__(FW_TPB) void net_forward_kernel(const T* __restrict__ theta, int64_t S, int D, int nl, FwDims d, int act,
                                                             const T* __restrict__ X, int N, T* __restrict__ out, int wmax) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* hA = reinterpret_cast<T*>(smem_raw);
  T* hB = hA + (size_t)wmax * FW_TPB;
  const int tid = threadIdx.x;
  const int p = blockIdx.x * FW_TPB + tid;
  const bool on = p < N;
  const int n_in = d.n[0], O = d.n[nl];
  for (int64_t s = blockIdx.y; s < S; s += gridDim.y) {
    const T* th = theta + s * D;
    if (nl == 2 && O <= FW_MAXO && d.n[1] > FW_MAXW) {
      // one hidden layer: out_o = b2_o + sum_j W2[o][j] act(b1_j + W1[j] . x), hidden units streamed
      const int H = d.n[1];
      const T* W1 = th; const T* b1 = W1 + (size_t)H * n_in; const T* W2 = b1 + H; const T* b2 = W2 + (size_t)O * H;
      for (int i = 0; i < n_in; ++i) hA[i * FW_TPB + tid] = on ? X[(size_t)p * n_in + i] : (T)0;
      T acc[FW_MAXO];
#pragma unroll
      for (int o = 0; o < FW_MAXO; ++o) acc[o] = o < O ? b2[o] : (T)0;
      for (int j = 0; j < H; ++j) {
        T z = b1[j];
        for (int i = 0; i < n_in; ++i) z = fma(W1[(size_t)j * n_in + i], hA[i * FW_TPB + tid], z);
        const T h = fw_act(z, act);
#pragma unroll
        for (int o = 0; o < FW_MAXO; ++o) if (o < O) acc[o] = fma(W2[(size_t)o * H + j], h, acc[o]);
      }
      if (on) {
#pragma unroll
        for (int o = 0; o < FW_MAXO; ++o) if (o < O) out[((size_t)s * N + p) * O + o] = acc[o];
      }
      continue;
    }
    // general case, register-blocked: activations live in LDS as [unit / 4][lane][4] (one conflict-free 16-byte read gives a
    // lane four consecutive units), a block of FW_JB output units accumulates in registers while four inputs at a time stream
    // in: per (FW_JB x 4) block one LDS read and FW_JB scalar 16-byte loads feed 4 FW_JB FMAs with SGPR operands
    {
      const int q_in = (n_in + 3) >> 2;
      for (int i = tid; i < q_in * 4 * FW_TPB; i += FW_TPB) hA[i] = (T)0;                     // (one wave: no barrier needed, lgkmcnt orders it)
      for (int i = 0; i < n_in; ++i) hA[((i >> 2) * FW_TPB + tid) * 4 + (i & 3)] = on ? X[(size_t)p * n_in + i] : (T)0;
    }
    T* cur = hA; T* nxt = hB;
    size_t off = 0;
    typedef T V4 __attribute__((ext_vector_type(4)));
    for (int l = 0; l < nl; ++l) {
      const int in = d.n[l], ow = d.n[l + 1];
      const T* W = th + off; const T* b = W + (size_t)ow * in;
      const bool last = l == nl - 1;
      const int in4 = in & ~3;
      for (int j0 = 0; j0 < ow; j0 += FW_JB) {
        T acc[FW_JB];
#pragma unroll
        for (int jj = 0; jj < FW_JB; ++jj) acc[jj] = (j0 + jj < ow) ? b[j0 + jj] : (T)0;
        for (int i = 0; i < in4; i += 4) {
          const V4 hv = *reinterpret_cast<const V4*>(cur + ((size_t)(i >> 2) * FW_TPB + tid) * 4);
#pragma unroll
          for (int jj = 0; jj < FW_JB; ++jj) {
            const int j = (j0 + jj < ow) ? j0 + jj : ow - 1;                                // (a clamped row: its sum is discarded)
            const T* wr = W + (size_t)j * in + i;
            acc[jj] = fma(wr[0], hv[0], acc[jj]); acc[jj] = fma(wr[1], hv[1], acc[jj]);
            acc[jj] = fma(wr[2], hv[2], acc[jj]); acc[jj] = fma(wr[3], hv[3], acc[jj]);
          }
        }
        if (in4 < in) {
          const V4 hv = *reinterpret_cast<const V4*>(cur + ((size_t)(in4 >> 2) * FW_TPB + tid) * 4);
#pragma unroll
          for (int jj = 0; jj < FW_JB; ++jj) {
            const int j = (j0 + jj < ow) ? j0 + jj : ow - 1;
            const T* wr = W + (size_t)j * in + in4;
            for (int k = 0; k < in - in4; ++k) acc[jj] = fma(wr[k], hv[k], acc[jj]);
          }
        }
#pragma unroll
        for (int jj = 0; jj < FW_JB; ++jj) {
          const int j = j0 + jj;
          if (j < ow) {
            if (last) { if (on) out[((size_t)s * N + p) * O + j] = acc[jj]; }
            else nxt[((size_t)(j >> 2) * FW_TPB + tid) * 4 + (j & 3)] = fw_act(acc[jj], act);
          }
        }
      }
      if (!last && (ow & 3)) {                                                                // the padding units of the last group read as zeros
        for (int j = ow; j < ((ow + 3) & ~3); ++j) nxt[((size_t)(j >> 2) * FW_TPB + tid) * 4 + (j & 3)] = (T)0;
      }
      off += (size_t)ow * in + ow;
      T* t = cur; cur = nxt; nxt = t;
    }
  }
}

template <typename T>
int net_forward(const T* theta, int64_t S, int n_layers, const int* dims, int act, const T* X, int N, T* out, hipStream_t s) {
  const char* who = "hta_net_forward";
  HTA_REQUIRE(theta && dims && X && out && S > 0 && N > 0 && n_layers >= 1 && n_layers <= FW_MAXL, "%s: bad arguments", who);
  HTA_REQUIRE(act >= 0 && act <= 2, "%s: activation must be 0 (relu), 1 (tanh) or 2 (sigmoid)", who);
  FwDims d;
  int64_t D = 0;
  int wmax = dims[0];
  for (int l = 0; l <= n_layers; ++l) {
    HTA_REQUIRE(dims[l] >= 1, "%s: layer width %d", who, dims[l]);
    d.n[l] = dims[l];
    if (l < n_layers) D += (int64_t)dims[l] * dims[l + 1] + dims[l + 1];
  }
  const bool streamed = n_layers == 2 && dims[2] <= FW_MAXO && dims[1] > FW_MAXW;
  if (streamed) wmax = dims[0];
  else for (int l = 0; l < n_layers; ++l) wmax = dims[l] > wmax ? dims[l] : wmax;          // inputs and hidden widths
  wmax = (wmax + 3) & ~3;
  HTA_REQUIRE(wmax <= FW_MAXW || streamed, "%s: width %d beyond %d (nets with one hidden layer and <= %d outputs have no width limit)", who, wmax,
              FW_MAXW, FW_MAXO);
  const size_t lds = (size_t)2 * wmax * FW_TPB * sizeof(T);
  static DevOnce done;
  if (!done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&net_forward_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { set_error("%s: hipFuncSetAttribute: %s", who, hipGetErrorString(e)); return HTA_ERR_LAUNCH; }
    done = true;
  }
  const dim3 grid((unsigned)((N + FW_TPB - 1) / FW_TPB), (unsigned)(S < 32768 ? S : 32768));
  note_route("net_forward_kernel<%s>", sizeof(T) == 4 ? "float" : "double");
  net_forward_kernel<T><<<grid, FW_TPB, lds, s>>>(theta, S, (int)D, n_layers, d, act, X, N, out, wmax);
  HTA_CHECK_LAUNCH(who);
  return HTA_OK;
}

}  // namespace hta

extern "C" {
int hta_net_forward_f32(const float* theta, int64_t S, int n_layers, const int* dims, int act, const float* X, int N, float* out,
                        void* stream) {
  return hta::net_forward<float>(theta, S, n_layers, dims, act, X, N, out, (hipStream_t)stream);
}
int hta_net_forward_f64(const double* theta, int64_t S, int n_layers, const int* dims, int act, const double* X, int N, double* out,
                        void* stream) {
  return hta::net_forward<double>(theta, S, n_layers, dims, act, X, N, out, (hipStream_t)stream);
}
}
