// Shader clock and issue cadence of a lone wave per SIMD (the occupancy of BASELINE config 2's consumers): cycles (s_memtime) and
// nanoseconds (s_memrealtime, 100 MHz) of T x 50 DEPENDENT v_fmac_f32 (y = fma(eps, r, y); r = fma(nel, y, r): the leapfrog chain
// of csrc/hmc_gaussian.hip) - alone, and with the 23 other instructions of a trajectory's bookkeeping approximated by independent
// VALU work.
//   hipcc -O3 --offload-arch=gfx950 tools/scratch/dep_fma_clock.hip -o tools/scratch/dep_fma_clock.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
template <int EXTRA>
__global__ void k(float* out, long long* stamps, int T, float eps, float nel) {
  float y = threadIdx.x * 1e-3f, r = 1.f - y, e0 = 0.f, e1 = 1.f, e2 = 2.f;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int t = 0; t < T; ++t) {
    asm volatile("" : "+v"(y), "+v"(r), "+v"(e0), "+v"(e1), "+v"(e2));      // (no closed form across trajectories)
#pragma unroll
    for (int l = 0; l < 25; ++l) {
      y = fmaf(eps, r, y);
      r = fmaf(nel, y, r);
    }
#pragma unroll
    for (int x = 0; x < EXTRA; ++x) {                 // independent of the chain and of each other (three accumulators)
      if (x % 3 == 0) e0 = fmaf(eps, e0, nel);
      if (x % 3 == 1) e1 = fmaf(eps, e1, nel);
      if (x % 3 == 2) e2 = fmaf(eps, e2, nel);
    }
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = y + r + e0 + e1 + e2;
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = w1 - w0; }
}
template <int EXTRA> void run(int blocks, int threads, int T) {
  float* out; long long* st;
  hipMalloc(&out, blocks * threads * 4); hipMalloc(&st, blocks * 16);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a, 0);
    k<EXTRA><<<blocks, threads>>>(out, st, T, 1e-3f, -1e-3f);
    hipEventRecord(b, 0); hipEventSynchronize(b);
  }
  float ms; hipEventElapsedTime(&ms, a, b);
  std::vector<long long> h(2 * blocks); hipMemcpy(h.data(), st, blocks * 16, hipMemcpyDeviceToHost);
  const double cyc = (double)h[0], ns = h[1] * 10.0;     // s_memrealtime: 100 MHz
  const int n = 50 + EXTRA;
  printf("blocks %4d x %4d threads, %2d extra: %.1f s_memtime ticks and %.1f ns per trajectory of %d instructions: %.3f ticks / %.3f ns per instruction; s_memtime rate %.1f MHz; event %.3f ms\n",
         blocks, threads, EXTRA, cyc / T, ns / T, n, cyc / T / n, ns / T / n, cyc / ns * 1e3, ms);
  hipFree(out); hipFree(st);
}
int main() {
  const int T = 20000;
  run<0>(16, 256, T); run<0>(1, 64, T); run<0>(256, 256, T); run<0>(1024, 256, T);
  run<23>(16, 256, T); run<23>(1, 64, T);
  return 0;
}
