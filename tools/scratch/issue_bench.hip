// single-wave issue-rate microbenchmark: ns per loop iteration for different loop bodies
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t err__ = (x); if (err__ != hipSuccess) { printf("err %s\n", hipGetErrorString(err__)); return 1; } } while (0)
template <int K, bool FENCE> __device__ __forceinline__ void qfma(float& acc, float coef, float x) {
  if (FENCE)
    asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 quad_perm:[%3,%3,%3,%3] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(x), "v"(coef), "i"(K));
  else
    asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[%3,%3,%3,%3] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(x), "v"(coef), "i"(K));
}
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int MODE, int UNROLL>
__global__ void k(float* out, const float* in, int L, float eps) {
  int t = threadIdx.x + blockIdx.x * blockDim.x;
  float d0 = in[t], d1 = in[t + 1], d2 = in[t + 2], p0 = in[t + 3], p1 = in[t + 4], p2 = in[t + 5];
  float a0 = in[t + 6] * 1e-3f, a1 = in[t + 7] * 1e-3f, a2 = in[t + 8] * 1e-3f;
  float b0 = in[t + 9] * 1e-3f, b1 = in[t + 10] * 1e-3f, b2 = in[t + 11] * 1e-3f, c0 = in[t + 12] * 1e-3f, c1 = in[t + 13] * 1e-3f, c2 = in[t + 14] * 1e-3f;
#pragma unroll UNROLL
  for (int l = 0; l < L; ++l) {
    if (MODE == 0) {          // thread-per-chain: 12 scalar FMAs
      d0 = fmaf(eps, p0, d0); d1 = fmaf(eps, p1, d1); d2 = fmaf(eps, p2, d2);
      p0 = fmaf(a0, d0, p0); p1 = fmaf(b0, d0, p1); p2 = fmaf(c0, d0, p2);
      p0 = fmaf(a1, d1, p0); p1 = fmaf(b1, d1, p1); p2 = fmaf(c1, d1, p2);
      p0 = fmaf(a2, d2, p0); p1 = fmaf(b2, d2, p1); p2 = fmaf(c2, d2, p2);
    } else if (MODE == 1) {   // quad: fmac + nop + 3 dpp fmac (asm)
      d0 = fmaf(eps, p0, d0);
      qfma<0, true>(p0, a0, d0); qfma<1, false>(p0, a1, d0); qfma<2, false>(p0, a2, d0);
    } else if (MODE == 2) {   // quad without the s_nop (UNSAFE -- timing only)
      d0 = fmaf(eps, p0, d0);
      qfma<0, false>(p0, a0, d0); qfma<1, false>(p0, a1, d0); qfma<2, false>(p0, a2, d0);
    } else if (MODE == 3) {   // quad via compiler intrinsics (mov_dpp + fmac)
      d0 = fmaf(eps, p0, d0);
      p0 = fmaf(a0, dpp_mov<0x00>(d0), p0); p0 = fmaf(a1, dpp_mov<0x55>(d0), p0); p0 = fmaf(a2, dpp_mov<0xAA>(d0), p0);
    } else if (MODE == 4) {   // 4 dependent plain FMAs (no dpp)
      d0 = fmaf(eps, p0, d0);
      p0 = fmaf(a0, d0, p0); p0 = fmaf(a1, d0, p0); p0 = fmaf(a2, d0, p0);
    } else if (MODE == 5) {   // 1 FMA
      d0 = fmaf(eps, p0, d0);
    } else if (MODE == 7) {   // explicit packed: (d0,d1) float2 + d2 scalar
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 d01 = {d0, d1}, p01 = {p0, p1};
      f2 e2 = {eps, eps};
      d01 = __builtin_elementwise_fma(e2, p01, d01); d2 = fmaf(eps, p2, d2);
      f2 A0 = {a0, b0}, A1 = {a1, b1}, A2 = {a2, b2};
      f2 x0 = {d01.x, d01.x}, x1 = {d01.y, d01.y}, x2 = {d2, d2};
      p01 = __builtin_elementwise_fma(A0, x0, p01); p2 = fmaf(c0, d01.x, p2);
      p01 = __builtin_elementwise_fma(A1, x1, p01); p2 = fmaf(c1, d01.y, p2);
      p01 = __builtin_elementwise_fma(A2, x2, p01); p2 = fmaf(c2, d2, p2);
      d0 = d01.x; d1 = d01.y; p0 = p01.x; p1 = p01.y;
    } else if (MODE == 8) {   // two independent chains per lane, scalar FMAs (24 per iteration)
      d0 = fmaf(eps, p0, d0); d1 = fmaf(eps, p1, d1); d2 = fmaf(eps, p2, d2);
      a0 = fmaf(eps, b0, a0); a1 = fmaf(eps, b1, a1); a2 = fmaf(eps, b2, a2);
      p0 = fmaf(c0, d0, p0); p1 = fmaf(c1, d0, p1); p2 = fmaf(c2, d0, p2);
      b0 = fmaf(c0, a0, b0); b1 = fmaf(c1, a0, b1); b2 = fmaf(c2, a0, b2);
      p0 = fmaf(c1, d1, p0); p1 = fmaf(c2, d1, p1); p2 = fmaf(c0, d1, p2);
      b0 = fmaf(c1, a1, b0); b1 = fmaf(c2, a1, b1); b2 = fmaf(c0, a1, b2);
      p0 = fmaf(c2, d2, p0); p1 = fmaf(c0, d2, p1); p2 = fmaf(c1, d2, p2);
      b0 = fmaf(c2, a2, b0); b1 = fmaf(c0, a2, b1); b2 = fmaf(c1, a2, b2);
    } else if (MODE == 6) {   // 2 interleaved quad chains (asm, nop)
      d0 = fmaf(eps, p0, d0); d1 = fmaf(eps, p1, d1);
      qfma<0, true>(p0, a0, d0); qfma<0, false>(p1, b0, d1); qfma<1, false>(p0, a1, d0); qfma<1, false>(p1, b1, d1);
      qfma<2, false>(p0, a2, d0); qfma<2, false>(p1, b2, d1);
    }
  }
  out[t] = d0 + d1 + d2 + p0 + p1 + p2 + a0 + a1 + a2 + b0 + b1 + b2;
}
template <int MODE, int UNROLL> int run(const char* name, float* out, float* in, int blocks, int threads) {
  const int L = 200000;
  hipEvent_t s, e; CK(hipEventCreate(&s)); CK(hipEventCreate(&e));
  k<MODE, UNROLL><<<blocks, threads>>>(out, in, 1000, 1e-3f);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(s)); k<MODE, UNROLL><<<blocks, threads>>>(out, in, L, 1e-3f); CK(hipEventRecord(e)); CK(hipEventSynchronize(e));
  float ms; CK(hipEventElapsedTime(&ms, s, e));
  printf("%-44s blocks=%4d thr=%3d : %7.2f ns/iter\n", name, blocks, threads, ms * 1e6 / L);
  return 0;
}
int main() {
  float *in, *out; CK(hipMalloc(&in, 1 << 22)); CK(hipMalloc(&out, 1 << 22)); CK(hipMemset(in, 0, 1 << 22));
  for (int blocks : {16}) {
    run<7, 1>("explicit float2 pk (no unroll)", out, in, blocks, 64);
    run<7, 5>("explicit float2 pk unroll5", out, in, blocks, 64);
    run<7, 25>("explicit float2 pk unroll25", out, in, blocks, 64);
    run<0, 25>("12 scalar FMA unroll25", out, in, blocks, 64);
    run<8, 5>("2 chains/lane 24 scalar FMA unroll5", out, in, blocks, 64);
    run<0, 5>("12 scalar FMA unroll5, 32 threads", out, in, blocks, 32);
    run<0, 5>("12 scalar FMA unroll5, 16 threads", out, in, blocks, 16);
    run<0, 1>("12 scalar FMA (thread/chain)", out, in, blocks, 64);
    run<0, 5>("12 scalar FMA unroll5", out, in, blocks, 64);
    run<1, 1>("quad asm dpp + s_nop", out, in, blocks, 64);
    run<1, 5>("quad asm dpp + s_nop unroll5", out, in, blocks, 64);
    run<2, 5>("quad asm dpp no nop unroll5", out, in, blocks, 64);
    run<3, 5>("quad intrinsic mov_dpp unroll5", out, in, blocks, 64);
    run<4, 5>("4 dependent plain FMA unroll5", out, in, blocks, 64);
    run<5, 5>("1 FMA unroll5", out, in, blocks, 64);
    run<6, 5>("2 interleaved quad chains unroll5", out, in, blocks, 64);
  }
  run<1, 5>("quad asm dpp + s_nop unroll5 (256 thr)", out, in, 256, 256);
  run<0, 5>("12 scalar FMA unroll5 (256 thr)", out, in, 256, 256);
  return 0;
}
