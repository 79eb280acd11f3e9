// Issue cost of a lone wave's DEPENDENT vector instructions by encoding / kind (cycles per instruction, 64 of them in one asm block):
// the cfg2 kernel's 19 instructions per trajectory that are not the FMA chain cost ~7 cycles each, the chain's 4.1 (profiles/r05y).
//   hipcc -O3 --offload-arch=gfx950 tools/scratch/valu_kinds_rate.hip -o tools/scratch/valu_kinds_rate.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)
#define R64(x) R16(x) R16(x) R16(x) R16(x)
template <int KIND>
__global__ void k(float* out, long long* stamps, int T, float a, float b) {
  float y = threadIdx.x * 1e-3f, r = 1.f - y, z = 0.5f;
  unsigned long long m = (threadIdx.x & 1) ? 0x5555555555555555ull : 0xAAAAAAAAAAAAAAAAull;
  m = __builtin_amdgcn_readfirstlane((unsigned)m) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(m >> 32)) << 32);
  const long long c0 = clock64();
  for (int t = 0; t < T; ++t) {
    if (KIND == 0) asm volatile(R64("v_fmac_f32_e32 %0, %1, %0\n") : "+v"(y) : "v"(a));
    if (KIND == 1) asm volatile(R64("v_fma_f32 %0, %1, %0, %2\n") : "+v"(y) : "v"(a), "v"(b));
    if (KIND == 2) asm volatile(R64("v_cndmask_b32_e64 %0, %0, %1, %2\n") : "+v"(y) : "v"(a), "s"(m));
    if (KIND == 3) asm volatile("s_mov_b64 vcc, %2\n" R64("v_cndmask_b32_e32 %0, %0, %1, vcc\n") : "+v"(y) : "v"(a), "s"(m) : "vcc");
    if (KIND == 4) asm volatile(R64("v_mul_f32_e32 %0, %1, %0\n") : "+v"(y) : "v"(a));
    if (KIND == 5) asm volatile(R64("v_fmac_f32_e32 %0, %2, %1\nv_fmac_f32_e32 %1, %2, %0\n") : "+v"(y), "+v"(r) : "v"(a));       // 128: alternating two registers (the leapfrog chain)
    if (KIND == 6) asm volatile(R64("v_fma_f32 %0, -%2, %1, %0\nv_fma_f32 %1, %2, %0, %1\n") : "+v"(y), "+v"(r) : "v"(a));         // 128, VOP3 with a negated source
    if (KIND == 7) asm volatile(R64("v_cmp_ge_f32_e64 %1, %0, %2\nv_cndmask_b32_e64 %0, %0, %2, %1\n") : "+v"(y), "=&s"(m) : "v"(a)); // 128: compare into an SGPR pair, select on it
    if (KIND == 8) asm volatile(R64("v_cmp_ge_f32_e32 vcc, %0, %1\nv_cndmask_b32_e32 %0, %0, %1, vcc\n") : "+v"(y) : "v"(a) : "vcc"); // 128: the VCC forms
    if (KIND == 9) asm volatile(R64("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\ns_nop 1\n") : "+v"(y));  // 64 DPP + 64 nops
    if (KIND == 10) asm volatile(R64("v_fmac_f32_e32 %0, %1, %0\nv_mov_b32_e32 %2, %0\n") : "+v"(y), "=v"(z) : "v"(a));            // 128: chain + an independent consumer
  }
  const long long c1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = y + r + z + (float)(m & 1);
  if (threadIdx.x == 0) stamps[blockIdx.x] = c1 - c0;
}
template <int KIND> void run(const char* name, int n) {
  const int T = 2000, blocks = 16, threads = 256;
  float* out; long long* st;
  (void)hipMalloc(&out, blocks * threads * 4); (void)hipMalloc(&st, blocks * 8);
  for (int rep = 0; rep < 3; ++rep) { k<KIND><<<blocks, threads>>>(out, st, T, 1e-3f, -1e-3f); (void)hipDeviceSynchronize(); }
  std::vector<long long> h(blocks); (void)hipMemcpy(h.data(), st, blocks * 8, hipMemcpyDeviceToHost);
  printf("%-72s %.2f cycles per instruction\n", name, (double)h[0] / T / n);
  (void)hipFree(out); (void)hipFree(st);
}
int main() {
  run<0>("v_fmac_f32_e32 (VOP2), dependent", 64);
  run<1>("v_fma_f32 (VOP3), dependent", 64);
  run<2>("v_cndmask_b32_e64, mask in an SGPR pair, dependent", 64);
  run<3>("v_cndmask_b32_e32, mask in VCC, dependent", 64);
  run<4>("v_mul_f32_e32, dependent", 64);
  run<5>("v_fmac_f32_e32 alternating y / r (the leapfrog chain)", 128);
  run<6>("v_fma_f32 alternating y / r, VOP3 with a negated source", 128);
  run<7>("v_cmp_ge_f32_e64 -> SGPR pair -> v_cndmask_b32_e64", 128);
  run<8>("v_cmp_ge_f32_e32 -> VCC -> v_cndmask_b32_e32", 128);
  run<9>("v_add_f32_dpp on its own result + s_nop 1 (per pair)", 64);
  run<10>("v_fmac_f32_e32 + independent v_mov of its result", 128);
  return 0;
}
