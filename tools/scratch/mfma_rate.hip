// Issue rate of the fp32 matrix instructions on one wave: cycles per instruction with 1 / 2 / 4 independent accumulators.
//   hipcc --offload-arch=gfx950 -O3 tools/scratch/mfma_rate.hip -o tools/scratch/mfma_rate && tools/scratch/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int KIND, int NACC>
__global__ void rate(float* out, long long* cyc, int iters) {
  f4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
        else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int KIND, int NACC> void run(const char* name, int waves) {
  float* out; long long* cyc; long long h = 0;
  hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8);
  const int iters = 2000;
  rate<KIND, NACC><<<256, 64 * waves>>>(out, cyc, iters);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  rate<KIND, NACC><<<256, 64 * waves>>>(out, cyc, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  const double n = (double)iters * 16 * NACC;
  printf("%-10s acc=%d waves/WG=%d: %.2f counter ticks per instruction (wave 0), %.3f ms -> %.2f ns per instruction per wave\n", name, NACC, waves,
         (double)h / n, ms, ms * 1e6 / n);
}
int main() {
  run<0, 1>("4x4x1_16b", 1); run<0, 2>("4x4x1_16b", 1); run<0, 4>("4x4x1_16b", 1); run<0, 4>("4x4x1_16b", 4); run<0, 4>("4x4x1_16b", 8);
  run<1, 1>("16x16x4", 1); run<1, 2>("16x16x4", 1); run<1, 4>("16x16x4", 1); run<1, 4>("16x16x4", 4);
  return 0;
}
