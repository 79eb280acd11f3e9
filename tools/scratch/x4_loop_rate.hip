// How fast does the product loop of rmhmc_mfma4x4_kernel issue when nothing but the loop runs?  104 instructions of
// v_mfma_f32_4x4x1_16b_f32 (two accumulator chains, operands as in the kernel: 2 x 52 A registers, B from 4 x 2 chunk registers
// with blgp 4..7), then the parity combine; the result feeds the next round's B values.  Prints counter ticks per product pair.
//   hipcc --offload-arch=gfx950 -O3 tools/scratch/x4_loop_rate.hip -o tools/scratch/_abl/x4_loop_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <utility>
typedef float bf4 __attribute__((ext_vector_type(4)));
template <int S> __device__ __forceinline__ bf4 mfma_g(float av, float bv, bf4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, c, 0, 0, 4 + S); }
template <int... I, typename F> __device__ __forceinline__ void static_for(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
__device__ __forceinline__ float other_parity(float h) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, h), 0x128, 0xf, 0xf, false));
}
template <int MODE>
__global__ __launch_bounds__(256) void loop_rate(const float* in, float* out, long long* cyc, int rounds) {
  float Sa[52], Pa[52];
#pragma unroll
  for (int j = 0; j < 52; ++j) { Sa[j] = in[j * 64 + (threadIdx.x & 63)]; Pa[j] = in[(52 + j) * 64 + (threadIdx.x & 63)]; }
  bf4 c1[4], c2[4];
#pragma unroll
  for (int Q = 0; Q < 4; ++Q) { c1[Q] = bf4{1e-3f, 2e-3f, 3e-3f, 4e-3f}; c2[Q] = bf4{4e-3f, 3e-3f, 2e-3f, 1e-3f}; }
  const long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < rounds; ++r) {
    bf4 acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
    static_for(std::make_integer_sequence<int, 13>{}, [&](auto qc) {
      constexpr int q = decltype(qc)::value;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc1 = mfma_g<q % 4>(Pa[4 * q + u], c1[q / 4][u], acc1);
        acc2 = mfma_g<q % 4>(Sa[4 * q + u], c2[q / 4][u], acc2);
      }
    });
    if (MODE >= 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc1[e] += other_parity(acc1[e]); acc2[e] += other_parity(acc2[e]); }
    }
#pragma unroll
    for (int Q = 0; Q < 4; ++Q) { c1[Q] = c1[Q] * 0.5f + acc2 * 1e-3f; c2[Q] = c2[Q] * 0.5f + acc1 * 1e-3f; }
    if (MODE >= 2) __syncthreads();
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = c1[0][0] + c2[1][1] + c1[2][2] + c2[3][3];
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE> void run(const char* name, float* in, float* out, long long* cyc) {
  const int rounds = 4000; long long h = 0;
  loop_rate<MODE><<<256, 256>>>(in, out, cyc, rounds);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  loop_rate<MODE><<<256, 256>>>(in, out, cyc, rounds);
  (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-44s %8.1f ticks per 104-instruction product pair (%.2f per instruction), %.3f us per pair\n", name, (double)h / rounds,
         (double)h / rounds / 104, ms * 1e3 / rounds);
}
int main() {
  float *in, *out; long long* cyc;
  (void)hipMalloc(&in, 104 * 64 * 4); (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&cyc, 8);
  (void)hipMemset(in, 0, 104 * 64 * 4);
  run<0>("products only", in, out, cyc);
  run<1>("products + parity combine", in, out, cyc);
  run<2>("products + parity combine + barrier", in, out, cyc);
  return 0;
}
