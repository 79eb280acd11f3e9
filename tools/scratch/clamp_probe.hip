// Device check of the one-instruction ReLU mask used by csrc/mlp_mfma.hip: v_mul_f32(h, inf) with the clamp output modifier
// must give 1 for every h > 0 (denormals included), 0 for h = 0 (0 * inf = NaN, clamped to 0 in DX10_CLAMP mode).
//   hipcc -O3 --offload-arch=gfx950 clamp_probe.hip -o clamp_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
__global__ void k(const float* in, float* out, int n) {
  int i = threadIdx.x;
  if (i < n) { float m; asm("v_mul_f32_e64 %0, %1, %2 clamp" : "=v"(m) : "v"(in[i]), "s"(__builtin_inff())); out[i] = m; }
}
int main() {
  unsigned bits[8] = {0x00000000u, 0x00000001u, 0x007fffffu, 0x00800000u, 0x3f800000u, 0x7f7fffffu, 0x7f800000u, 0x33800000u};
  float h[8], o[8]; memcpy(h, bits, sizeof(h));
  float *di, *dout; hipMalloc(&di, 32); hipMalloc(&dout, 32);
  hipMemcpy(di, h, 32, hipMemcpyHostToDevice);
  k<<<1, 64>>>(di, dout, 8);
  hipMemcpy(o, dout, 32, hipMemcpyDeviceToHost);
  int ok = 1;
  for (int i = 0; i < 8; ++i) { float want = i == 0 ? 0.f : 1.f; printf("h = %-14g (0x%08x) -> %g %s\n", h[i], bits[i], o[i], o[i] == want ? "" : "MISMATCH"); ok &= o[i] == want; }
  printf(ok ? "clamp mask: OK\n" : "clamp mask: FAILED\n");
  return !ok;
}
