#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int K, bool FENCE> __device__ __forceinline__ void qfma(float& acc, float coef, float x) {
  if (FENCE)
    asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 quad_perm:[%3,%3,%3,%3] row_mask:0xf bank_mask:0xf bound_ctrl:1"
                 : "+v"(acc) : "v"(x), "v"(coef), "i"(K));
  else
    asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[%3,%3,%3,%3] row_mask:0xf bank_mask:0xf bound_ctrl:1"
                 : "+v"(acc) : "v"(x), "v"(coef), "i"(K));
}
__global__ void k(float* out, const float* x, const float* coef) {
  int t = threadIdx.x;
  float xv = x[t] * 1.0f + 0.0f;
  float acc = 100.0f;
  qfma<0, true>(acc, coef[t], xv);
  qfma<1, false>(acc, coef[t + 64], xv);
  qfma<2, false>(acc, coef[t + 128], xv);
  out[t] = acc;
  float s = xv;
  s += dpp_mov<0xB1>(s);
  s += dpp_mov<0x4E>(s);
  out[64 + t] = s;
}
int main() {
  float hx[64], hc[192], ho[128];
  for (int i = 0; i < 64; ++i) hx[i] = i + 1;
  for (int i = 0; i < 192; ++i) hc[i] = 0.001f * (i + 1);
  float *dx, *dc, *dout;
  hipMalloc(&dx, sizeof hx); hipMalloc(&dc, sizeof hc); hipMalloc(&dout, sizeof ho);
  hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice); hipMemcpy(dc, hc, sizeof hc, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dout, dx, dc);
  hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < 64; ++t) {
    int b = t & ~3;
    float want = 100.0f + hc[t] * hx[b] + hc[t + 64] * hx[b + 1] + hc[t + 128] * hx[b + 2];
    float ws = hx[b] + hx[b + 1] + hx[b + 2] + hx[b + 3];
    if (fabsf(ho[t] - want) > 1e-3f || fabsf(ho[64 + t] - ws) > 1e-3f) { if (bad < 8) printf("lane %d: fma %f want %f  sum %f want %f\n", t, ho[t], want, ho[64 + t], ws); ++bad; }
  }
  printf("dpp_check: %d bad lanes\n", bad);
  return bad != 0;
}
