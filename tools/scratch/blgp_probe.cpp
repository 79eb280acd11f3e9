// Which lanes supply the B (blgp) / A (cbsz, abid) operand of v_mfma_f32_4x4x1_16b_f32 on gfx950?  Prints, per modifier, the
// source lane every output lane saw.   hipcc --offload-arch=gfx950 -O2 blgp_probe.cpp -o _abl/blgp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float bf4 __attribute__((ext_vector_type(4)));
template <int CBSZ, int ABID, int BLGP>
__global__ void probe(float* o, int which) {
  const int l = threadIdx.x;
  const float av = which == 0 ? 1.f : (float)l, bv = which == 0 ? (float)l : 1.f;
  bf4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, acc, CBSZ, ABID, BLGP);
  for (int e = 0; e < 4; ++e) o[l * 4 + e] = acc[e];
}
template <int CBSZ, int ABID, int BLGP> void run(float* d, int which, const char* name) {
  float h[256];
  probe<CBSZ, ABID, BLGP><<<1, 64>>>(d, which);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("%s (%s source lane; one line per 16 lanes, per lane the 4 accumulator registers)\n", name, which == 0 ? "B" : "A");
  for (int l = 0; l < 64; ++l) {
    printf(" [%2d:%2.0f %2.0f %2.0f %2.0f]", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    if (l % 8 == 7) printf("\n");
  }
}
int main() {
  float* d; hipMalloc(&d, 256 * sizeof(float));
  run<0, 0, 0>(d, 0, "plain");
  run<0, 0, 1>(d, 0, "blgp 1"); run<0, 0, 2>(d, 0, "blgp 2"); run<0, 0, 3>(d, 0, "blgp 3");
  run<0, 0, 4>(d, 0, "blgp 4"); run<0, 0, 5>(d, 0, "blgp 5"); run<0, 0, 6>(d, 0, "blgp 6"); run<0, 0, 7>(d, 0, "blgp 7");
  run<0, 0, 0>(d, 1, "plain");
  run<3, 0, 0>(d, 1, "cbsz 3 abid 0"); run<3, 5, 0>(d, 1, "cbsz 3 abid 5"); run<4, 9, 0>(d, 1, "cbsz 4 abid 9"); run<2, 1, 0>(d, 1, "cbsz 2 abid 1");
  return 0;
}
