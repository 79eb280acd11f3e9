// Run-time half of the callback compiler: hipRTC -> gfx950 code object -> hipModule -> launches.
//
// hamiltorch_amd/jit/ traces a user log_prob_func (the callback contract of hamiltorch/samplers.py:272-274), writes its
// value / derivatives as straight-line device code and hands the SOURCE here; the hand-written kernels it is compiled into
// live under csrc/jit/ (hmc_callback.hip.in: the reference's sample() loop for plain HMC, samplers.py:965-1026, around that
// function; derivs_callback.hip.in: the derivatives the Riemannian samplers ask torch.func for, samplers.py:108, :397-398).
//
// Boundary rules as everywhere else: device pointers are the caller's, launches are enqueued on the caller's stream, nothing is
// synchronised on the launch path.  Compiling (hta_jit_compile) is host work and returns a malloc'ed code object; loading
// (hta_jit_load) creates a hipModule on the current device and reads the module's 32-byte info block back once - a
// preparation step like hta_*_prepare, outside every timed or captured region.
#include <dlfcn.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include "common.hpp"
#include "jit/jit_args.h"

namespace hta {
void profile_begin(hipStream_t s);      // abi.cpp: optional HIP-event bracket of a call's dominant kernel
void profile_end(hipStream_t s);
namespace {

// ---- hipRTC through dlopen: the library loads (and everything that is not the callback compiler works) without it ----
typedef struct _hiprtcProgram* rtcProgram;
struct Rtc {
  void* h = nullptr;
  int (*create)(rtcProgram*, const char*, const char*, int, const char* const*, const char* const*) = nullptr;
  int (*compile)(rtcProgram, int, const char* const*) = nullptr;
  int (*destroy)(rtcProgram*) = nullptr;
  int (*log_size)(rtcProgram, size_t*) = nullptr;
  int (*get_log)(rtcProgram, char*) = nullptr;
  int (*code_size)(rtcProgram, size_t*) = nullptr;
  int (*get_code)(rtcProgram, char*) = nullptr;
  const char* (*err_string)(int) = nullptr;
  bool ok = false;
};

Rtc& rtc() {
  static Rtc r;
  static bool tried = false;
  if (tried) return r;
  tried = true;
  const char* names[] = {"libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"};
  for (const char* n : names) {
    r.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (r.h) break;
  }
  if (!r.h) return r;
#define HTA_RTC_SYM(field, name) *(void**)(&r.field) = dlsym(r.h, name)
  HTA_RTC_SYM(create, "hiprtcCreateProgram");
  HTA_RTC_SYM(compile, "hiprtcCompileProgram");
  HTA_RTC_SYM(destroy, "hiprtcDestroyProgram");
  HTA_RTC_SYM(log_size, "hiprtcGetProgramLogSize");
  HTA_RTC_SYM(get_log, "hiprtcGetProgramLog");
  HTA_RTC_SYM(code_size, "hiprtcGetCodeSize");
  HTA_RTC_SYM(get_code, "hiprtcGetCode");
  HTA_RTC_SYM(err_string, "hiprtcGetErrorString");
#undef HTA_RTC_SYM
  r.ok = r.create && r.compile && r.destroy && r.log_size && r.get_log && r.code_size && r.get_code;
  return r;
}

thread_local std::string g_jit_log;

struct Module {
  hipModule_t mod = nullptr;
  hipFunction_t hmc = nullptr, predraw = nullptr, derivs = nullptr, contract = nullptr, rmhmc = nullptr;
  int info[HTA_CB_INFO_WORDS] = {};
  int device = -1;
};

int check_module(const Module* m, const char* who, int D, int itemsize, int mass_kind, int set) {
  HTA_REQUIRE(m && m->mod, "%s: module is NULL", who);
  int dev = -1;
  (void)hipGetDevice(&dev);
  HTA_REQUIRE(dev == m->device, "%s: the module was loaded on device %d, the call runs on device %d", who, m->device, dev);
  HTA_REQUIRE(m->info[1] == D && m->info[2] == itemsize,
              "%s: the module was compiled for D = %d, %d-byte elements; the call has D = %d, %d-byte elements", who, m->info[1],
              m->info[2], D, itemsize);
  HTA_REQUIRE(mass_kind < 0 || m->info[3] == mass_kind, "%s: the module was compiled for mass kind %d, the call has %d", who,
              m->info[3], mass_kind);
  HTA_REQUIRE(m->info[4] == set, "%s: the module holds kernel set %d, not %d", who, m->info[4], set);
  return HTA_OK;
}

int launch(hipFunction_t fn, const char* who, int64_t C, void* args, size_t bytes, hipStream_t s, unsigned block = 64) {
  void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &bytes, HIP_LAUNCH_PARAM_END};
  const unsigned grid = (unsigned)((C + block - 1) / block);
  hipError_t e = hipModuleLaunchKernel(fn, grid, 1, 1, block, 1, 1, 0, s, nullptr, config);
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", who, hipGetErrorString(e));
    return HTA_ERR_LAUNCH;
  }
  return HTA_OK;
}

}  // namespace
}  // namespace hta

extern "C" {

int hta_jit_available(void) { return hta::rtc().ok ? 1 : 0; }

const char* hta_jit_last_log(void) { return hta::g_jit_log.c_str(); }

int hta_jit_note_fallback(const char* reason) {
  hta::note_route("torch-evaluated callback + hmc_pieces kernels (not compiled: %s)", reason ? reason : "?");
  return HTA_OK;
}

int hta_jit_compile(const char* source, const char* name, int n_headers, const char* const* header_names,
                    const char* const* header_sources, int n_options, const char* const* options, void** code_out,
                    int64_t* code_bytes) {
  using namespace hta;
  g_jit_log.clear();
  HTA_REQUIRE(source && code_out && code_bytes && n_headers >= 0 && n_options >= 0, "hta_jit_compile: bad arguments");
  HTA_REQUIRE(n_headers == 0 || (header_names && header_sources), "hta_jit_compile: headers are NULL");
  *code_out = nullptr;
  *code_bytes = 0;
  Rtc& r = rtc();
  if (!r.ok) {
    set_error("hta_jit_compile: libhiprtc.so could not be loaded (%s)", r.h ? "symbols missing" : dlerror());
    return HTA_ERR_UNSUPPORTED;
  }
  rtcProgram prog = nullptr;
  int rc = r.create(&prog, source, name ? name : "hta_callback.hip", n_headers, header_sources, header_names);
  if (rc != 0) {
    set_error("hta_jit_compile: hiprtcCreateProgram failed (%d: %s)", rc, r.err_string ? r.err_string(rc) : "?");
    return HTA_ERR_LAUNCH;
  }
  rc = r.compile(prog, n_options, options);
  size_t ls = 0;
  if (r.log_size(prog, &ls) == 0 && ls > 1) {
    g_jit_log.resize(ls);
    (void)r.get_log(prog, &g_jit_log[0]);
  }
  if (rc != 0) {
    set_error("hta_jit_compile: hiprtcCompileProgram failed (%d: %s); hta_jit_last_log() has the compiler's messages", rc,
              r.err_string ? r.err_string(rc) : "?");
    (void)r.destroy(&prog);
    return HTA_ERR_INVALID;
  }
  size_t cs = 0;
  rc = r.code_size(prog, &cs);
  void* buf = (rc == 0 && cs > 0) ? malloc(cs) : nullptr;
  if (!buf || r.get_code(prog, (char*)buf) != 0) {
    free(buf);
    (void)r.destroy(&prog);
    set_error("hta_jit_compile: no code object (%zu bytes)", cs);
    return HTA_ERR_LAUNCH;
  }
  (void)r.destroy(&prog);
  *code_out = buf;
  *code_bytes = (int64_t)cs;
  return HTA_OK;
}

void hta_jit_free(void* code) { free(code); }

int hta_jit_load(const void* code, int64_t bytes, void** module_out) {
  using namespace hta;
  HTA_REQUIRE(code && bytes > 0 && module_out, "hta_jit_load: bad arguments");
  *module_out = nullptr;
  Module* m = new Module();
  hipError_t e = hipGetDevice(&m->device);
  if (e == hipSuccess) e = hipModuleLoadData(&m->mod, code);
  if (e != hipSuccess) {
    set_error("hta_jit_load: hipModuleLoadData: %s", hipGetErrorString(e));
    delete m;
    return HTA_ERR_LAUNCH;
  }
  hipDeviceptr_t ip = nullptr;
  size_t ib = 0;
  e = hipModuleGetGlobal(&ip, &ib, m->mod, "hta_cb_info");
  if (e == hipSuccess && ib == sizeof(m->info)) e = hipMemcpyDtoH(m->info, ip, sizeof(m->info));
  if (e != hipSuccess || ib != sizeof(m->info) || m->info[0] != HTA_CB_MAGIC) {
    set_error("hta_jit_load: the code object has no hta_cb_info block (%s)", hipGetErrorString(e));
    (void)hipModuleUnload(m->mod);
    delete m;
    return HTA_ERR_INVALID;
  }
  if (m->info[4] == HTA_CB_SET_HMC) {
    e = hipModuleGetFunction(&m->hmc, m->mod, "hta_cb_hmc_kernel");
    if (e == hipSuccess) e = hipModuleGetFunction(&m->predraw, m->mod, "hta_cb_predraw_kernel");
  } else if (m->info[4] == HTA_CB_SET_DERIVS) {
    e = hipModuleGetFunction(&m->derivs, m->mod, "hta_cb_derivs_kernel");
    if (e == hipSuccess) e = hipModuleGetFunction(&m->contract, m->mod, "hta_cb_contract_kernel");
  } else if (m->info[4] == HTA_CB_SET_RMHMC) {
    e = hipModuleGetFunction(&m->rmhmc, m->mod, "hta_cb_rmhmc_kernel");
  } else {
    e = hipErrorInvalidValue;
  }
  if (e != hipSuccess) {
    set_error("hta_jit_load: kernel set %d: %s", m->info[4], hipGetErrorString(e));
    (void)hipModuleUnload(m->mod);
    delete m;
    return HTA_ERR_INVALID;
  }
  *module_out = m;
  return HTA_OK;
}

int hta_jit_unload(void* module) {
  hta::Module* m = (hta::Module*)module;
  if (!m) return HTA_OK;
  if (m->mod) (void)hipModuleUnload(m->mod);
  delete m;
  return HTA_OK;
}

int hta_jit_module_info(void* module, int* info_out) {
  hta::Module* m = (hta::Module*)module;
  if (!m || !info_out) { hta::set_error("hta_jit_module_info: bad arguments"); return HTA_ERR_INVALID; }
  memcpy(info_out, m->info, sizeof(m->info));
  return HTA_OK;
}

int64_t hta_jit_hmc_workspace_bytes(int64_t C, int D, int itemsize) {
  if (C <= 0 || D <= 0 || (itemsize != 4 && itemsize != 8)) return -1;
  return C * D * itemsize + C * itemsize;       // gcur[C, D] + lp_out[C]
}

int64_t hta_jit_hmc_predraw_bytes(int64_t C, int D, int n_traj, int itemsize) {
  if (C <= 0 || D <= 0 || n_traj < 0 || (itemsize != 4 && itemsize != 8)) return -1;
  return (int64_t)n_traj * (D + 1) * C * itemsize;      // [n_traj, D + 1, C]
}

int hta_jit_hmc_sample(void* module, const HtaCbHmcArgs* args, int D, int itemsize, int mass_kind, void* workspace,
                       int64_t workspace_bytes, void* stream) {
  using namespace hta;
  Module* m = (Module*)module;
  if (int rc = check_module(m, "hta_jit_hmc_sample", D, itemsize, mass_kind, HTA_CB_SET_HMC)) return rc;
  HTA_REQUIRE(args && args->cur && args->init && args->reject_count && args->C > 0 && args->L >= 0 && args->n_traj >= 0,
              "hta_jit_hmc_sample: bad arguments");
  HTA_REQUIRE(mass_kind == HTA_MASS_NONE || (args->inv_mass && args->mass_factor), "hta_jit_hmc_sample: mass operands are NULL");
  HTA_REQUIRE(workspace && workspace_bytes >= hta_jit_hmc_workspace_bytes(args->C, D, itemsize),
              "hta_jit_hmc_sample: workspace of %lld bytes, %lld needed (hta_jit_hmc_workspace_bytes)", (long long)workspace_bytes,
              (long long)hta_jit_hmc_workspace_bytes(args->C, D, itemsize));
  HTA_REQUIRE(!args->pre || args->pre_bytes >= hta_jit_hmc_predraw_bytes(args->C, D, args->n_traj, itemsize),
              "hta_jit_hmc_sample: pre-draw buffer of %lld bytes, %lld needed (hta_jit_hmc_predraw_bytes)", (long long)args->pre_bytes,
              (long long)hta_jit_hmc_predraw_bytes(args->C, D, args->n_traj, itemsize));
  if (args->n_traj == 0) return HTA_OK;
  HtaCbHmcArgs a = *args;
  a.resume = args->resume ? 1 : 0;
  a.gcur = workspace;
  a.lp_out = (char*)workspace + args->C * D * itemsize;
  note_route("hta_cb_hmc_kernel<D=%d,%s,mass=%d,nodes=%d%s>", D, itemsize == 4 ? "f32" : "f64", mass_kind, m->info[5],
             a.pre ? ",predrawn" : "");
  profile_begin((hipStream_t)stream);
  int rc = HTA_OK;
  if (a.pre) rc = launch(m->predraw, "hta_jit_hmc_sample (pre-draw)", a.C * (int64_t)a.n_traj, &a, sizeof(a), (hipStream_t)stream, 256);
  if (rc == HTA_OK) rc = launch(m->hmc, "hta_jit_hmc_sample", a.C, &a, sizeof(a), (hipStream_t)stream);
  profile_end((hipStream_t)stream);
  return rc;
}

int64_t hta_jit_rmhmc_workspace_bytes(int64_t C, int D, int itemsize) {
  if (C <= 0 || D <= 0 || (itemsize != 4 && itemsize != 8)) return -1;
  return C * itemsize;                          // lp_out[C]
}

int hta_jit_rmhmc_sample(void* module, const HtaCbRmhmcArgs* args, int D, int itemsize, int has_jitter, void* workspace,
                         int64_t workspace_bytes, void* stream) {
  using namespace hta;
  Module* m = (Module*)module;
  if (int rc = check_module(m, "hta_jit_rmhmc_sample", D, itemsize, has_jitter ? 1 : 0, HTA_CB_SET_RMHMC)) return rc;
  HTA_REQUIRE(args && args->cur && args->init && args->reject_count && args->C > 0 && args->L >= 0 && args->n_traj >= 0,
              "hta_jit_rmhmc_sample: bad arguments");
  HTA_REQUIRE(workspace && workspace_bytes >= hta_jit_rmhmc_workspace_bytes(args->C, D, itemsize),
              "hta_jit_rmhmc_sample: workspace of %lld bytes, %lld needed (hta_jit_rmhmc_workspace_bytes)", (long long)workspace_bytes,
              (long long)hta_jit_rmhmc_workspace_bytes(args->C, D, itemsize));
  if (args->n_traj == 0) return HTA_OK;
  HtaCbRmhmcArgs a = *args;
  a.lp_out = workspace;
  note_route("hta_cb_rmhmc_kernel<D=%d,%s,jitter=%d,nodes=%d+%d>", D, itemsize == 4 ? "f32" : "f64", has_jitter ? 1 : 0, m->info[5],
             m->info[6]);
  profile_begin((hipStream_t)stream);
  const int rc = launch(m->rmhmc, "hta_jit_rmhmc_sample", a.C, &a, sizeof(a), (hipStream_t)stream);
  profile_end((hipStream_t)stream);
  return rc;
}

/* which: 0 = derivatives (logp / grad / neg_hess, each optional), 1 = third-order contraction (M, contract) */
int hta_jit_derivs(void* module, const HtaCbDerivArgs* args, int which, int D, int itemsize, void* stream) {
  using namespace hta;
  Module* m = (Module*)module;
  if (int rc = check_module(m, "hta_jit_derivs", D, itemsize, -1, HTA_CB_SET_DERIVS)) return rc;
  HTA_REQUIRE(args && args->theta && args->C > 0, "hta_jit_derivs: bad arguments");
  HTA_REQUIRE(which == 0 || (args->M && (args->contract || (args->upd && args->grad_in))), "hta_jit_derivs: M / contract / upd are NULL");
  HtaCbDerivArgs a = *args;
  note_route("%s<D=%d,%s,nodes=%d>", which ? "hta_cb_contract_kernel" : "hta_cb_derivs_kernel", D, itemsize == 4 ? "f32" : "f64",
             m->info[5]);
  return launch(which ? m->contract : m->derivs, "hta_jit_derivs", a.C, &a, sizeof(a), (hipStream_t)stream);
}

}  // extern "C"
