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
#include "mlp_mfma_dev.hpp"

#if HTA_TIMING
__device__ unsigned long long hta_dbg[16];
extern "C" void hta_dbg_read(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(hta_dbg), sizeof(hta_dbg)); }
#endif

namespace hta {

void profile_begin(hipStream_t s);
void profile_end(hipStream_t s);

bool mlp_mfma_eligible(const MlpArgs<float>& a) {
  // Gaussian likelihood only: the other likelihoods run on the VALU kernel (mlp_hmc.hip).  Their exp / log1p sequences - even
  // out of line, even behind a call never taken - cost this kernel 2-4 % on BASELINE config 4 at its 128-register cap (measured).
  if (a.loss != HTA_LOSS_REGRESSION) return false;
  if (a.n_in < 1 || a.n_in > 16 || a.H < 1 || a.H > 256) return false;
  if (!(a.mass_kind == HTA_MASS_NONE || a.mass_kind == HTA_MASS_DIAG)) return false;
  const int NK = (a.n_in + 3) / 4, NU = (a.H + 15) / 16;
  return mfma_lds_bytes(a, NK, NU, 1, nullptr) <= 150 * 1024;
}

template <int NK, int NPT, int ACT, int NTMAX> static int launch_mfma(const MlpArgs<float>& a, hipStream_t s) {
  const int NU = (a.H + 15) / 16;
  int Npad;
  const size_t lds = mfma_lds_bytes(a, NK, NU, 1, &Npad);
  static DevOnce done;
  if (!done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_mfma_kernel<NK, NPT, ACT, NTMAX, 1>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { set_error("hta_mlp_hmc: hipFuncSetAttribute: %s", hipGetErrorString(e)); return HTA_ERR_LAUNCH; }
    done = true;
  }
  const int grid = (int)(a.C < 8192 ? a.C : 8192);
  profile_begin(s);
  note_route("mlp_mfma_kernel<%d,%d,%d,%d>", NK, NPT, ACT, NTMAX);
  mlp_mfma_kernel<NK, NPT, ACT, NTMAX, 1><<<grid, 64 * NU, lds, s>>>(a, NU, Npad);
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
