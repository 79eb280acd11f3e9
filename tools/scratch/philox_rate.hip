// Developer micro-benchmark: Philox4x32-10 with separate v_mul_hi_u32 / v_mul_lo_u32 (form a) against the 64-bit product
// form (b: one v_mad_u64_u32 per multiplier) - blocks per second per wave at 1, 2 and 4 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 philox_rate.hip -o philox_rate.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
struct U4 { uint32_t x, y, z, w; };
__device__ __forceinline__ U4 philox_a(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0;
    const uint32_t n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}
__device__ __forceinline__ U4 philox_b(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}
template <int F> __global__ void kern(uint32_t* o, uint32_t k0, uint32_t k1, int n) {
  uint32_t acc = 0;
#pragma unroll 1
  for (int i = 0; i < n; ++i) {
    U4 r = F ? philox_b(threadIdx.x, i, blockIdx.x, acc, k0, k1) : philox_a(threadIdx.x, i, blockIdx.x, acc, k0, k1);   // dependent chain, as a trajectory's
    acc ^= r.x ^ r.y ^ r.z ^ r.w;
  }
  o[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
  uint32_t* o; hipMalloc(&o, 1024 * 1024 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int n = 20000;
  for (int wps = 1; wps <= 4; wps *= 2)
    for (int f = 0; f < 2; ++f) {
      float best = 1e9;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (f) kern<1><<<256, 256 * wps>>>(o, 1, 2, n); else kern<0><<<256, 256 * wps>>>(o, 1, 2, n);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      printf("waves/SIMD %d form %c: %.3f ms for %d dependent blocks per lane = %.1f ns per block per wave\n", wps, f ? 'b' : 'a', best, n, best * 1e6 / n);
    }
  uint32_t h[4]; hipMemcpy(h, o, 16, hipMemcpyDeviceToHost); printf("%u\n", h[0]);
  return 0;
}
