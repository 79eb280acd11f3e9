// Scalar helpers the generated callback code calls (hamiltorch_amd/jit/emit.py names them); overloaded on float / double.
// Compiled by hipRTC only (no system headers: the integer types come from here).
#pragma once

#ifdef __HIPCC_RTC__
typedef unsigned char uint8_t;
typedef int int32_t;
typedef unsigned int uint32_t;
typedef long long int64_t;
typedef unsigned long long uint64_t;
#endif

namespace hta_cb {

__device__ __forceinline__ float cb_exp(float x) { return expf(x); }
__device__ __forceinline__ double cb_exp(double x) { return exp(x); }
__device__ __forceinline__ float cb_log(float x) { return logf(x); }
__device__ __forceinline__ double cb_log(double x) { return log(x); }
__device__ __forceinline__ float cb_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double cb_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float cb_rsqrt(float x) { return 1.0f / sqrtf(x); }
__device__ __forceinline__ double cb_rsqrt(double x) { return 1.0 / sqrt(x); }
__device__ __forceinline__ float cb_tanh(float x) { return tanhf(x); }
__device__ __forceinline__ double cb_tanh(double x) { return tanh(x); }
__device__ __forceinline__ float cb_log1p(float x) { return log1pf(x); }
__device__ __forceinline__ double cb_log1p(double x) { return log1p(x); }
__device__ __forceinline__ float cb_expm1(float x) { return expm1f(x); }
__device__ __forceinline__ double cb_expm1(double x) { return expm1(x); }
__device__ __forceinline__ float cb_sin(float x) { return sinf(x); }
__device__ __forceinline__ double cb_sin(double x) { return sin(x); }
__device__ __forceinline__ float cb_cos(float x) { return cosf(x); }
__device__ __forceinline__ double cb_cos(double x) { return cos(x); }
__device__ __forceinline__ float cb_atan(float x) { return atanf(x); }
__device__ __forceinline__ double cb_atan(double x) { return atan(x); }
__device__ __forceinline__ float cb_erf(float x) { return erff(x); }
__device__ __forceinline__ double cb_erf(double x) { return erf(x); }
__device__ __forceinline__ float cb_lgamma(float x) { return lgammaf(x); }
__device__ __forceinline__ double cb_lgamma(double x) { return lgamma(x); }
__device__ __forceinline__ float cb_abs(float x) { return fabsf(x); }
__device__ __forceinline__ double cb_abs(double x) { return fabs(x); }
__device__ __forceinline__ float cb_pow(float x, float y) { return powf(x, y); }
__device__ __forceinline__ double cb_pow(double x, double y) { return pow(x, y); }
__device__ __forceinline__ float cb_max(float x, float y) { return (x != x || y != y) ? x + y : fmaxf(x, y); }     // NaN propagates, as torch.maximum
__device__ __forceinline__ double cb_max(double x, double y) { return (x != x || y != y) ? x + y : fmax(x, y); }
__device__ __forceinline__ float cb_min(float x, float y) { return (x != x || y != y) ? x + y : fminf(x, y); }
__device__ __forceinline__ double cb_min(double x, double y) { return (x != x || y != y) ? x + y : fmin(x, y); }
__device__ __forceinline__ float cb_floor(float x) { return floorf(x); }
__device__ __forceinline__ double cb_floor(double x) { return floor(x); }
__device__ __forceinline__ float cb_ceil(float x) { return ceilf(x); }
__device__ __forceinline__ double cb_ceil(double x) { return ceil(x); }
__device__ __forceinline__ float cb_round(float x) { return rintf(x); }      // half to even, as torch.round
__device__ __forceinline__ double cb_round(double x) { return rint(x); }
__device__ __forceinline__ float cb_trunc(float x) { return truncf(x); }
__device__ __forceinline__ double cb_trunc(double x) { return trunc(x); }

template <typename T> __device__ __forceinline__ T cb_sign(T x) { return (T)((x > (T)0) - (x < (T)0)); }
template <typename T> __device__ __forceinline__ T cb_sigmoid(T x) { return (T)1 / ((T)1 + cb_exp(-x)); }
// log(1 + exp(x)) without overflow: max(x, 0) + log1p(exp(-|x|))
template <typename T> __device__ __forceinline__ T cb_softplus(T x) { return fmax(x, (T)0) + cb_log1p(cb_exp(-cb_abs(x))); }
template <typename T> __device__ __forceinline__ bool cb_isnan(T x) { return x != x; }
template <typename T> __device__ __forceinline__ bool cb_isinf(T x) { return cb_abs(x) == (T)__builtin_huge_val(); }

// digamma: the recurrence up to x >= 6, then the asymptotic series (relative error < 1e-8 in double); reflection for x < 0.
template <typename T> __device__ inline T cb_digamma(T x) {
  double v = (double)x, r = 0.0;
  if (v <= 0.0) {
    if (v == floor(v)) return (T)__builtin_nan("");
    const double pi = 3.14159265358979323846;
    r = -pi / tan(pi * v);          // psi(1 - x) - psi(x) = pi cot(pi x)
    v = 1.0 - v;
  }
  while (v < 6.0) { r -= 1.0 / v; v += 1.0; }
  const double f = 1.0 / (v * v);
  r += log(v) - 0.5 / v - f * (1.0 / 12 - f * (1.0 / 120 - f * (1.0 / 252 - f * (1.0 / 240 - f * (1.0 / 132)))));
  return (T)r;
}

}  // namespace hta_cb
