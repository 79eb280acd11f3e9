// libhamiltorch_amd.so: ABI version, error channel, device query, tuning knobs.
#include <stdarg.h>
#include "common.hpp"

namespace hta {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
// ---- which kernel a C-ABI call dispatched to (hta_last_route): the dominant kernel of the calling thread's last call ----
static thread_local char g_route[160] = "";
void note_route(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_route, sizeof(g_route), fmt, ap);
  va_end(ap);
}
int g_small_chains_per_block = 0;  // 0 = default (64)
int g_force_general = 0;
int g_rmhmc_batch = 1;     // fused RMHMC: 16 chains per workgroup on the matrix cores from 2048 chains on (0 off, 2 always)
int g_rmhmc_momwave = 1;   // fused RMHMC: momentum draws by the wave-per-task kernel (fp32, jitter, D <= 104); 0: workgroup-per-task kernel
int g_rmhmc_mfma4 = 1;     // fused RMHMC: four chains per workgroup on v_mfma_f32_4x4x1_16b (0 off, 1 for lo <= chains < hi, 2 always)
int g_rmhmc_mfma4_lo = 1793, g_rmhmc_mfma4_hi = 2049;   // round 4: up to 7 x CUs chains the two-chain kernel of rmhmc_uvc.hip, two workgroups per CU (1536 chains: 2.25e8 against 1.97e8); before:   // round 2 (tracked products): 544 / 640 / 700 chains 1.40x / 1.37x / 1.38x the one-chain kernel, below 513 rmhmc_uv_kernel; round 1: 3072: 0.87x, 4096: 0.88x the 16-chain kernel (a third workgroup per CU doubles a SIMD's load)
int g_netn_waves = 1;        // csrc/netn_hmc.hip: waves per chain (2 / 4 where they fit; measured slower at 1024 chains)
int g_rmhmc_uv = 1;          // rmhmc_uv.hip (one or two chains per workgroup as columns of the matrix instruction): 1 up to 2 x CUs chains, 0 off, 2 always
int g_rmhmc_pair = 1;        // four-chain kernels: consecutive half steps share product phases (0: one half step at a time)
int g_rmhmc_mfma4_waves = 4; // fused RMHMC, four chains per workgroup: 4 = four waves (rows x k parity inside a wave), 2 = two waves
int g_rmhmc_overlap = 0;   // fused RMHMC: 1 = momentum draws of the next block of trajectories on a side stream (round 3: off -
                           // run beside the trajectory kernel the draws slow it by a third; serial is 3-8 % faster at every chain count)
int g_rmhmc_momsplit = 1;  // fused RMHMC with jitter: p = chol(P) z1 + sqrt(jitter u) . z2 (exactly N(0, P + diag(jitter u)) like
                           // chol(P + diag(jitter u)) z, without a Cholesky per draw); 0 = the per-draw factorisation
int g_rmhmc_lean = 1;      // rmhmc_uv / rmhmc_mfma4x4: 1 = the instances without lane-predicated stores and padding selects (bit-identical; 0 = the round-2 instances)
int g_gauss_eig = 1;       // small-D identity-mass Gaussian HMC integrates in the eigenbasis of P (0: direct kernel, 2: chain per lane only)
int g_fill_blocks = 4096;        // grid cap of the pre-draw pass (256-thread blocks, grid-stride)
int g_quad_max_chains = 65536;   // up to here a chain takes a DPP quad (one eigen-coordinate per lane), beyond a lane
int g_quad_variant = 7;          // instance of the quad kernel (hmc_gaussian.hip: VAR); 0 = the round-1 instance, 3 = uniform-base addressing + no NaN guard, 7 = + fused row / butterfly block
int g_rmhmc_fused = 1;           // 0 = per-evaluation Jacobi path, 3 = fused with two chains per workgroup (parity tests)
extern int g_metric_mfma, g_metric_general, g_metric_traj, g_metric_second, g_metric_bx3, g_metric_resident, g_metric_select, g_metric_sqrtdraw, g_rmhmc_wide, g_rmhmc_uv_co, g_rmhmc_uv_acc, g_rmhmc_uv_g, g_rmhmc_uvc, g_quad_fused, g_quad_producers, g_quad_chunks, g_quad_starve;         // rmhmc_metric_mfma.hip, rmhmc_fused.hip
int g_mlp_valu = 0;               // 1 = keep the Bayesian-MLP sampler on the VALU kernel (parity tests of both)
int g_mlp3_route = 1;             // csrc/mlp3_mfma.hip (two wide hidden layers on the matrix cores); 0 = such models stay on the callback path

// ---- optional HIP-event timing of the dominant kernel of each call (measurement only) -----------
// hta_set_tuning("profile", N) arms it; every N-th bracketed launch then records a start/stop event pair
// on the launch stream (no synchronisation); hta_profile_collect() waits for them and returns the sum and the
// number of pairs.  (An event record is a barrier packet: ~3 us of bubble each, which is why N > 1 exists.)
static int g_profile = 0;
static int g_prof_seen = 0;
static bool g_prof_armed = false;
static const int kMaxPairs = 8192;
static hipEvent_t g_ev[kMaxPairs][2];
static int g_ev_created = 0, g_ev_used = 0;
void profile_begin(hipStream_t s) {
  g_prof_armed = g_profile > 0 && g_ev_used < kMaxPairs && (g_prof_seen++ % g_profile) == 0;
  if (!g_prof_armed) return;
  if (g_ev_used >= g_ev_created) {
    (void)hipEventCreate(&g_ev[g_ev_created][0]);
    (void)hipEventCreate(&g_ev[g_ev_created][1]);
    ++g_ev_created;
  }
  (void)hipEventRecord(g_ev[g_ev_used][0], s);
}
void profile_end(hipStream_t s) {
  if (!g_prof_armed) return;
  g_prof_armed = false;
  (void)hipEventRecord(g_ev[g_ev_used][1], s);
  ++g_ev_used;
}
}  // namespace hta

extern "C" {

int hta_abi_version(void) { return HTA_ABI_VERSION; }
const char* hta_last_error(void) { return hta::g_err; }

int hta_device_info(int device, HtaDeviceInfo* out) {
  if (!out) { hta::set_error("hta_device_info: out is NULL"); return HTA_ERR_INVALID; }
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) { hta::set_error("hipGetDeviceProperties: %s", hipGetErrorString(e)); return HTA_ERR_LAUNCH; }
  memset(out, 0, sizeof(*out));
  out->abi_version = HTA_ABI_VERSION;
  out->device = device;
  out->compute_units = prop.multiProcessorCount;
  out->wavefront_size = prop.warpSize;
  out->lds_bytes_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
  out->clock_khz = prop.clockRate;
  out->hbm_bytes = (int64_t)prop.totalGlobalMem;
  strncpy(out->arch, prop.gcnArchName, sizeof(out->arch) - 1);
  return HTA_OK;
}

namespace {
struct TuneKey { const char* key; int* var; int dflt; };
// every route key with its default: hta_reset_tuning() restores the column on the right
const TuneKey kTune[] = {
    {"small_chains_per_block", &hta::g_small_chains_per_block, 0}, {"force_general", &hta::g_force_general, 0},
    {"gauss_eig", &hta::g_gauss_eig, 1}, {"rmhmc_momwave", &hta::g_rmhmc_momwave, 1}, {"rmhmc_mfma4", &hta::g_rmhmc_mfma4, 1},
    {"rmhmc_mfma4_lo", &hta::g_rmhmc_mfma4_lo, 1793}, {"rmhmc_mfma4_hi", &hta::g_rmhmc_mfma4_hi, 2049},
    {"rmhmc_mfma4_waves", &hta::g_rmhmc_mfma4_waves, 4}, {"netn_waves", &hta::g_netn_waves, 1}, {"rmhmc_uv", &hta::g_rmhmc_uv, 1},
    {"rmhmc_pair", &hta::g_rmhmc_pair, 1}, {"rmhmc_wide", &hta::g_rmhmc_wide, 1}, {"rmhmc_overlap", &hta::g_rmhmc_overlap, 0},
    {"rmhmc_momsplit", &hta::g_rmhmc_momsplit, 1},
    {"rmhmc_batch", &hta::g_rmhmc_batch, 1}, {"quad_max_chains", &hta::g_quad_max_chains, 65536}, {"fill_blocks", &hta::g_fill_blocks, 4096},
    {"mlp_valu", &hta::g_mlp_valu, 0}, {"metric_mfma", &hta::g_metric_mfma, 1}, {"metric_general", &hta::g_metric_general, 1}, {"metric_traj", &hta::g_metric_traj, 1}, {"metric_second", &hta::g_metric_second, 1}, {"metric_bx3", &hta::g_metric_bx3, 2}, {"metric_resident", &hta::g_metric_resident, 1}, {"metric_select", &hta::g_metric_select, 1}, {"metric_sqrtdraw", &hta::g_metric_sqrtdraw, 1}, {"rmhmc_fused", &hta::g_rmhmc_fused, 1},
    {"mlp3_route", &hta::g_mlp3_route, 1}, {"quad_variant", &hta::g_quad_variant, 7}, {"rmhmc_lean", &hta::g_rmhmc_lean, 1},
    {"rmhmc_uv_co", &hta::g_rmhmc_uv_co, 1}, {"rmhmc_uv_acc", &hta::g_rmhmc_uv_acc, 2}, {"rmhmc_uv_g", &hta::g_rmhmc_uv_g, 0}, {"rmhmc_uvc", &hta::g_rmhmc_uvc, 1}, {"quad_fused", &hta::g_quad_fused, 1}, {"quad_producers", &hta::g_quad_producers, 64}, {"quad_chunks", &hta::g_quad_chunks, 8}, {"quad_starve", &hta::g_quad_starve, 0},
};
// HTA_TUNING_DEFAULTS=key=value,...: moves the DEFAULT of route keys for this process (applied when the library is loaded and by
// hta_reset_tuning) - A/B runs of whole test files under another route without touching the tests' own set / reset calls
int env_default(const TuneKey& t) {
  const char* e = getenv("HTA_TUNING_DEFAULTS");
  if (!e) return t.dflt;
  const size_t n = strlen(t.key);
  for (const char* p = e; *p;) {
    if (!strncmp(p, t.key, n) && p[n] == '=') return atoi(p + n + 1);
    const char* c = strchr(p, ',');
    if (!c) break;
    p = c + 1;
  }
  return t.dflt;
}
struct ApplyEnvDefaults { ApplyEnvDefaults() { for (const TuneKey& t : kTune) *t.var = env_default(t); } } g_apply_env_defaults;
}  // namespace

int hta_set_tuning(const char* key, int value) {
  if (!key) return HTA_ERR_INVALID;
  if (!strcmp(key, "profile")) { hta::g_profile = value; hta::g_ev_used = 0; hta::g_prof_seen = 0; return HTA_OK; }
  for (const TuneKey& t : kTune)
    if (!strcmp(key, t.key)) {
      *t.var = (t.var == &hta::g_fill_blocks && value <= 0) ? 4096 : value;
      return HTA_OK;
    }
  hta::set_error("hta_set_tuning: unknown key %s", key);
  return HTA_ERR_INVALID;
}

int hta_get_tuning(const char* key, int* value) {
  if (!key || !value) return HTA_ERR_INVALID;
  for (const TuneKey& t : kTune)
    if (!strcmp(key, t.key)) { *value = *t.var; return HTA_OK; }
  hta::set_error("hta_get_tuning: unknown key %s", key);
  return HTA_ERR_INVALID;
}

int hta_reset_tuning(void) {
  for (const TuneKey& t : kTune) *t.var = env_default(t);
  return HTA_OK;
}

const char* hta_last_route(void) { return hta::g_route; }

int hta_profile_collect(double* total_ms, int* launches) {
  double tot = 0;
  for (int i = 0; i < hta::g_ev_used; ++i) {
    float ms = 0;
    hipError_t e = hipEventSynchronize(hta::g_ev[i][1]);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, hta::g_ev[i][0], hta::g_ev[i][1]);
    if (e != hipSuccess) { hta::set_error("hta_profile_collect: %s", hipGetErrorString(e)); return HTA_ERR_LAUNCH; }
    tot += ms;
  }
  if (total_ms) *total_ms = tot;
  if (launches) *launches = hta::g_ev_used;
  hta::g_ev_used = 0;
  return HTA_OK;
}

}  // extern "C"
