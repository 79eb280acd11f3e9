// probe (round 6): can a hipRTC-built module kernel be launched with > 64 KB of LDS (static or dynamic) through hipModuleLaunchKernel?
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <stdio.h>
#include <vector>
static const char* src = R"(
extern "C" __global__ void __launch_bounds__(64) dyn(float* out, int n) {
  extern __shared__ float s[];
  for (int i = threadIdx.x; i < n; i += 64) s[i] = (float)i;
  __syncthreads();
  float a = 0; for (int i = threadIdx.x; i < n; i += 64) a += s[n - 1 - i];
  out[blockIdx.x * 64 + threadIdx.x] = a;
}
extern "C" __global__ void __launch_bounds__(64) stat(float* out) {
  __shared__ float s[36 * 1024];            // 144 KB
  for (int i = threadIdx.x; i < 36 * 1024; i += 64) s[i] = (float)i;
  __syncthreads();
  float a = 0; for (int i = threadIdx.x; i < 36 * 1024; i += 64) a += s[36 * 1024 - 1 - i];
  out[blockIdx.x * 64 + threadIdx.x] = a;
}
)";
int main() {
  hiprtcProgram p; hiprtcCreateProgram(&p, src, "t.hip", 0, nullptr, nullptr);
  const char* o[] = {"--offload-arch=gfx950", "-O3"};
  int rc = hiprtcCompileProgram(p, 2, o);
  size_t ls; hiprtcGetProgramLogSize(p, &ls); if (ls > 1) { std::vector<char> l(ls); hiprtcGetProgramLog(p, l.data()); printf("log: %s\n", l.data()); }
  printf("compile rc %d\n", rc); if (rc) return 1;
  size_t cs; hiprtcGetCodeSize(p, &cs); std::vector<char> code(cs); hiprtcGetCode(p, code.data());
  hipModule_t m; printf("load %d\n", hipModuleLoadData(&m, code.data()));
  hipFunction_t fd, fs; printf("get %d %d\n", hipModuleGetFunction(&fd, m, "dyn"), hipModuleGetFunction(&fs, m, "stat"));
  float* out; hipMalloc(&out, 256 * 64 * 4);
  for (int kb : {32, 64, 96, 128, 160}) {
    int n = kb * 256; struct { float* o; int n; } a{out, n}; size_t sz = sizeof(a);
    void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
    hipError_t e = hipModuleLaunchKernel(fd, 256, 1, 1, 64, 1, 1, kb * 1024, 0, nullptr, cfg);
    hipError_t e2 = hipDeviceSynchronize();
    printf("dynamic %3d KB: launch %s sync %s\n", kb, hipGetErrorString(e), hipGetErrorString(e2));
    if (e != hipSuccess) { (void)hipGetLastError(); e = hipFuncSetAttribute((const void*)fd, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024); printf("   setattr: %s\n", hipGetErrorString(e)); (void)hipGetLastError(); }
  }
  struct { float* o; } b{out}; size_t sz = sizeof(b);
  void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &b, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
  hipError_t e = hipModuleLaunchKernel(fs, 256, 1, 1, 64, 1, 1, 0, 0, nullptr, cfg);
  printf("static 144 KB: launch %s sync %s\n", hipGetErrorString(e), hipGetErrorString(hipDeviceSynchronize()));
  return 0;
}
