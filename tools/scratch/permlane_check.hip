// Developer check (round 4): v + shfl_xor(v, 16 / 32) as ONE VALU instruction each through gfx950's v_permlane16_swap_b32 /
// v_permlane32_swap_b32 (a = b = v; after the swap a + b is the pair sum in every lane).  Verified correct on the GPU (inline
// assembly; the builtin's second result is folded onto the first when both operands are the same value).  Tried in the cross-group
// sums of mlp_mfma_kernel / mlp3_mfma_kernel instead of the two ds_bpermute_b32 of __shfl_xor: bit-identical, NOT faster (cfg4
// 1.63e7 -> 1.60e7, nbmlp unchanged) - those sums sit at the end of a gradient pass where the LDS pipe is idle.  Not used.
//   hipcc -O3 --offload-arch=gfx950 tools/scratch/permlane_check.hip -o tools/scratch/permlane_check.bin
#include <hip/hip_runtime.h>
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float xor32_sum(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return a + b;
}
__device__ __forceinline__ float xor16_sum(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return a + b;
}
__global__ void k(float* out, const float* in) {
  float v = in[threadIdx.x];
  out[threadIdx.x] = xor32_sum(xor16_sum(v));
  out[64 + threadIdx.x] = xor16_sum(v);
  out[128 + threadIdx.x] = xor32_sum(v);
}
int main() {
  float h[64], o[192]; for (int i = 0; i < 64; ++i) h[i] = (float)(1 << (i / 16)) * 1000 + i;   // row r: 1000 * 2^r + lane
  float *di, *dout; hipMalloc(&di, 256); hipMalloc(&dout, 768);
  hipMemcpy(di, h, 256, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dout, di); hipMemcpy(o, dout, 768, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; ++i) {
    const float w4 = h[i % 16] + h[16 + i % 16] + h[32 + i % 16] + h[48 + i % 16];
    const float w16 = h[i] + h[i ^ 16], w32 = h[i] + h[i ^ 32];
    if (o[i] != w4 || o[64 + i] != w16 || o[128 + i] != w32) { if (bad < 5) printf("lane %d: %g %g | %g %g | %g %g\n", i, o[i], w4, o[64 + i], w16, o[128 + i], w32); ++bad; }
  }
  printf("permlane swap sums: %s (%d bad)\n", bad ? "WRONG" : "ok", bad);
  return bad != 0;
}
