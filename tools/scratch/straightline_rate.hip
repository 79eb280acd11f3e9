// Does a lone wave issue straight-line code slower than a short loop?  Cycles per dependent v_fmac_f32 for a loop body of 50 x U
// FMAs (U = 1, 8, 32, 64: 200 B ... 12.8 KB of code per iteration; the cfg2 kernel's pass is 2347 instructions = 10.8 KB).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=on tools/scratch/straightline_rate.hip -o tools/scratch/straightline_rate.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
template <int U>
__global__ void k(float* out, long long* stamps, int T, float eps, float nel) {
  float y = threadIdx.x * 1e-3f, r = 1.f - y;
  const long long c0 = clock64();
  for (int t = 0; t < T; t += U) {
    asm volatile("" : "+v"(y), "+v"(r));
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int l = 0; l < 25; ++l) { y = fmaf(eps, r, y); r = fmaf(nel, y, r); }
      asm volatile("" : "+v"(y), "+v"(r));
    }
  }
  const long long c1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = y + r;
  if (threadIdx.x == 0) stamps[blockIdx.x] = c1 - c0;
}
template <int U> void run(int blocks, int threads, int T) {
  float* out; long long* st;
  (void)hipMalloc(&out, blocks * threads * 4); (void)hipMalloc(&st, blocks * 8);
  for (int rep = 0; rep < 3; ++rep) { k<U><<<blocks, threads>>>(out, st, T, 1e-3f, -1e-3f); (void)hipDeviceSynchronize(); }
  std::vector<long long> h(blocks); (void)hipMemcpy(h.data(), st, blocks * 8, hipMemcpyDeviceToHost);
  printf("blocks %3d x %3d threads, %2d x 50 FMAs per loop body (%5d B of code): %.3f cycles per FMA (wave 0 of block 0), %.3f (last block)\n", blocks, threads, U, U * 200, (double)h[0] / T / 50, (double)h[blocks - 1] / T / 50);
  (void)hipFree(out); (void)hipFree(st);
}
int main() {
  const int T = 64 * 300;
  run<1>(16, 256, T); run<8>(16, 256, T); run<32>(16, 256, T); run<64>(16, 256, T);
  run<1>(1, 64, T); run<32>(1, 64, T); run<64>(1, 64, T);
  return 0;
}
