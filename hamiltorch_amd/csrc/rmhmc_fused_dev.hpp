// Device-side pieces shared by the whole-trajectory RMHMC kernels of the Gaussian-target fast path
// (rmhmc_fused.hip: one / four / sixteen chains per workgroup; rmhmc_uv.hip: one or two chains per workgroup on the matrix cores).
#pragma once
#include <utility>
#include "common.hpp"
#include "philox.hpp"
#include "rmhmc.hpp"

namespace hta {

template <typename T> struct FusedArgs {
  T* cur; const T* theta_init; const T* P; const T* S; const T* mu;
  T log_norm; T logdetP; int has_jitter; T jitter; int K; int series;
  int64_t C; int D; int L; T eps; T rot_c; T rot_s;
  int n_traj; int traj_offset; int burn; uint64_t seed; uint64_t chain_offset;
  T* samples; int32_t* reject_count; T* H_old; T* H_new; uint8_t* accept;
  const T* p_ws;                // pre-drawn momenta [n_traj, C, D] (rmhmc_momentum_kernel) or NULL: factor in the kernel
};

typedef float bf4 __attribute__((ext_vector_type(4)));

// ---- the 16-block fp32 matrix instruction with rows x contraction parity inside a wave (rmhmc_mfma4x4_kernel, rmhmc_uv_kernel)
// (XHL = 68, XLD = 140: the 8 operand segments a group of 8 lanes reads at once - 4 columns x 16 bytes, two such groups per
//  parity - start at banks 0, 12, 24, 36 (+4 for the odd parity): no two share a bank; 64 / 128 would put them on the same four)
constexpr int XNC = 4, XHL = 68, XLD = 2 * XHL + 4, XWV = 4, XNT = 64 * XWV, XKJ = 52, XQ = XKJ / 4, XSQ = (XQ + 3) / 4, XBUF = 16;

// lane l ^ 8's value (the same rows at the other contraction parity): a DPP rotation inside the 16-lane row
__device__ __forceinline__ float other_parity(float h) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, h), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
}
// the 16-block product with the B operand of ALL four 16-lane groups taken from group S (blgp 4 + S; semantics probed on
// gfx950 by tools/scratch/blgp_probe.cpp)
template <int S> __device__ __forceinline__ bf4 mfma_from_group(float av, float bv, bf4 c) {
  return __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, c, 0, 0, 4 + S);
}
template <int... I, typename F> __device__ __forceinline__ void static_for(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}


// rmhmc_uv.hip: the same run for at most 2 x (compute units) chains; cus = compute units of the current device
int rmhmc_uv_launch(const FusedArgs<float>& a, int cus, hipStream_t s);
// rmhmc_uvc.hip: one chain per workgroup, compact element-wise layout, three product phases per step (K == 2 with jitter only)
int rmhmc_uvc_launch(const FusedArgs<float>& a, bool co, hipStream_t s);
int rmhmc_uvc2_launch(const FusedArgs<float>& a, bool co, hipStream_t s);    // two chains per workgroup, two values per lane, any K
extern int g_rmhmc_uvc;

}  // namespace hta
