// Metropolis rules shared by the library's kernels (common.hpp) and the run-time compiled callback kernels
// (csrc/jit/hmc_callback.hip.in): device code only, no includes - hipRTC has no system headers.
#pragma once

namespace hta {

template <typename T> __device__ __forceinline__ bool finite_(T v) { return isfinite(v); }

// accept rule of samplers.py:626 + 1000-1004: rho = min(0, H0 - H1); accept iff rho >= log(u);
// a non-finite proposed energy or log-prob is a rejection (LogProbError path, samplers.py:1045-1057).
template <typename T> __device__ __forceinline__ bool mh_accept(T h_old, T h_new, T logp_new, T u) {
  const T rho = fmin((T)0, h_old - h_new);
  return finite_(h_old) && finite_(h_new) && finite_(logp_new) && (rho >= log(u));
}

// same rule with log(u) already taken (pre-drawn workspace)
template <typename T> __device__ __forceinline__ bool mh_accept_logu(T h_old, T h_new, T logp_new, T logu) {
  const T rho = fmin((T)0, h_old - h_new);
  // one class test: the sum is finite iff all three are (inf - inf = NaN; realistic magnitudes cannot overflow)
  return finite_(h_old + h_new + logp_new) && (rho >= logu);
}

}  // namespace hta
