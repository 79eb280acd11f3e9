// Counter-based RNG contract of the sampling engine (gfx950 device code).
//
// Replaces the reference's use of the global torch generator for the momentum
// draw (hamiltorch/samplers.py:185-202), the Metropolis uniform
// (samplers.py:1004) and the Fisher-metric jitter (samplers.py:115).
// Stream = (64-bit seed, global chain id, trajectory index, purpose, sub-stream),
// so results do not depend on how chains are sharded over GPUs.
// oracle/hmc_oracle.py implements the identical bit stream on the CPU.
#pragma once
#ifndef __HIPCC_RTC__      // (hipRTC has the device runtime built in and no system headers: csrc/jit/ supplies the integer types)
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

namespace hta {

enum : uint32_t { PURPOSE_MOMENTUM = 0, PURPOSE_MH = 1, PURPOSE_JITTER = 2, PURPOSE_INIT = 3, PURPOSE_PERM = 4 };

struct U4 { uint32_t x, y, z, w; };

__device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                            uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
#if defined(HTA_PHILOX_MAD64)
    // One 32 x 32 -> 64 product per multiplier (v_mad_u64_u32) instead of a v_mul_hi_u32 and a v_mul_lo_u32: integer multiplies run at a
    // quarter of the VALU rate and are most of a Philox round.  Per translation unit (defined ahead of the first include): the 64-bit
    // register pairs cost the 1-chain fused kernel 18 more spills, so the sources whose kernels have the registers opt in
    // (rmhmc_uvc.hip: BASELINE config 3 + 1.3 ... 1.7 %; rmhmc_metric_mfma.hip).  Tried and not kept: mlp_mfma.hip / mlp3_mfma.hip (their kernels
    // spill: the resource tests' budgets), hmc_gaussian.hip (BASELINE config 2 is bound by its consumer waves: 1.68e11 either way).  Same bit stream.
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
#else
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
#endif
    const uint32_t n0 = hi1 ^ c1 ^ k0;
    const uint32_t n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}

// counter = (block, draw, chain, purpose + 16*sub); key = seed
__device__ __forceinline__ U4 philox_block(uint64_t seed, uint64_t chain, uint32_t draw, uint32_t purpose,
                                           uint32_t sub, uint32_t block) {
  return philox4x32_10(block, draw, (uint32_t)chain, purpose + 16u * sub, (uint32_t)seed, (uint32_t)(seed >> 32));
}

// ((x >> 9) + 0.5) * 2^-23 : 24-bit significand, exact in fp32, never 0 or 1.
template <typename T> __device__ __forceinline__ T u23(uint32_t x) {
  return ((T)(x >> 9) + (T)0.5) * (T)1.1920928955078125e-07;
}

__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& z0, float& z1) {
  const float u1 = u23<float>(a), u2 = u23<float>(b);
  const float r = sqrtf(-2.0f * logf(u1));
  float s, c;
  sincospif(2.0f * u2, &s, &c);
  z0 = r * c; z1 = r * s;
}
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, double& z0, double& z1) {
  const double u1 = u23<double>(a), u2 = u23<double>(b);
  const double r = sqrt(-2.0 * log(u1));
  double s, c;
  sincospi(2.0 * u2, &s, &c);
  z0 = r * c; z1 = r * s;
}

// Four standard normals of block `block`: (x,y)->(z0,z1), (z,w)->(z2,z3).
template <typename T> __device__ __forceinline__ void normal4(const U4& r, T (&z)[4]) {
  box_muller(r.x, r.y, z[0], z[1]);
  box_muller(r.z, r.w, z[2], z[3]);
}

// element j of the D-vector draw: block j/4, slot j%4
template <typename T>
__device__ __forceinline__ T normal_elem(uint64_t seed, uint64_t chain, uint32_t draw, uint32_t sub, int j) {
  const U4 r = philox_block(seed, chain, draw, PURPOSE_MOMENTUM, sub, (uint32_t)(j >> 2));
  T z0, z1;
  if (j & 2) box_muller(r.z, r.w, z0, z1); else box_muller(r.x, r.y, z0, z1);
  return (j & 1) ? z1 : z0;
}

template <typename T>
__device__ __forceinline__ T uniform_elem(uint64_t seed, uint64_t chain, uint32_t draw, uint32_t purpose,
                                          uint32_t sub, int j) {
  const U4 r = philox_block(seed, chain, draw, purpose, sub, (uint32_t)(j >> 2));
  const uint32_t v = (j & 2) ? ((j & 1) ? r.w : r.z) : ((j & 1) ? r.y : r.x);
  return u23<T>(v);
}

// The subset order of Integrator.SPLITTING_RAND (torch.randperm(M) once per trajectory, samplers.py:549): a
// Fisher-Yates shuffle on integer draws, one order per (seed, trajectory) shared by every chain of the batch
// (chain key 0xFFFFFFFF).  Same integers in hamiltorch_amd/util.py and oracle/hmc_oracle.py.
__device__ inline void split_permutation(uint64_t seed, uint32_t draw, int M, int* perm) {
  for (int i = 0; i < M; ++i) perm[i] = i;
  for (int i = M - 1; i >= 1; --i) {
    const U4 r = philox_block(seed, 0xFFFFFFFFull, draw, PURPOSE_PERM, 0, (uint32_t)(i >> 2));
    const uint32_t v = (i & 2) ? ((i & 1) ? r.w : r.z) : ((i & 1) ? r.y : r.x);
    const int j = (int)(((uint64_t)v * (uint64_t)(i + 1)) >> 32);
    const int t = perm[i]; perm[i] = perm[j]; perm[j] = t;
  }
}

}  // namespace hta
