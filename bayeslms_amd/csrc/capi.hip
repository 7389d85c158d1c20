// Library info + error state of libbayeslm_hip.so.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "blm_host.h"

static thread_local char g_err[512] = "";

int blm_fail(int status, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return status;
}

extern "C" uint32_t blm_abi_version(void) { return BLM_ABI_VERSION; }
extern "C" const char* blm_last_error(void) { return g_err; }

extern "C" int blm_query(int device, char* arch32, int* n_cu, int* lds_bytes) {
  hipDeviceProp_t prop;
  BLM_HIP(hipGetDeviceProperties(&prop, device));
  if (arch32) {
    strncpy(arch32, prop.gcnArchName, 31);
    arch32[31] = 0;
  }
  if (n_cu) *n_cu = prop.multiProcessorCount;
  if (lds_bytes) *lds_bytes = (int)prop.maxSharedMemoryPerMultiProcessor;
  return BLM_OK;
}


// ---- kernel-selection options --------------------------------------------------------------------------------------
namespace {
struct OptEntry { const char* name; const char* env; int def, lo, hi, value; bool inited; };
OptEntry g_opts[blm::OPT_COUNT] = {
    {"attn_hpw", "BLM_ATTN_HPW", 0, 0, 2, 0, false},      // heads per workgroup of the T <= 128 attention kernels: 0 = by head count
    {"attn_short", "BLM_ATTN_SHORT", 1, 0, 1, 1, false},  // 1: one-wave-per-head forward for T <= 32; 0: the 128-row forward there too
    {"attn_valu", "BLM_ATTN_VALU", 0, 0, 1, 0, false},    // 1: the vector-ALU attention kernels also at head_dim 64 (cross-check of the MFMA ones)
    {"lstm_gemv", "BLM_LSTM_GEMV", 1, 0, 1, 1, false},    // 1: one-wave-per-unit step kernel for B <= 4; 0: the matrix-core step kernel there too
    {"lstm_pipe", "BLM_LSTM_PIPE", 1, 0, 1, 1, false},    // 1: software-pipelined K loop of the step kernels where a lane walks >= 4 chunks
    {"lstm_tail", "BLM_LSTM_TAIL", 0, 0, 1, 0, false},    // 1: the general (K tail) form of the pipelined step kernels also for whole chunks
    // 1: every reduction in a fixed order -- no K slices in the GEMM family (no float atomics into C), column sums / GP coefficient
    // gradients in one row chunk, KL sums through block partials added by one block, the embedding gradient by one wave per
    // vocabulary row in position order: two runs from one seed give bit-identical parameters (tests/test_gpu_deterministic.py)
    {"deterministic", "BLM_DETERMINISTIC", 0, 0, 1, 0, false},
    {"lstm_mb2", "BLM_LSTM_MB2", 1, 0, 1, 1, false},     // 1: the search cell's forward step takes two batch tiles per workgroup at B > 32 (W streamed once, one round)
};
std::mutex g_opt_mu;
void opt_init(OptEntry& e) {
  if (e.inited) return;
  const char* v = getenv(e.env);
  e.value = e.def;
  if (v && *v) { const int x = atoi(v); if (x >= e.lo && x <= e.hi) e.value = x; }
  e.inited = true;
}
}  // namespace

int blm::option(blm::Opt o) {
  std::lock_guard<std::mutex> lk(g_opt_mu);
  opt_init(g_opts[o]);
  return g_opts[o].value;
}

extern "C" int blm_set_option(const char* name, int value) {
  if (!name) return blm_fail(BLM_ERR_INVALID, "blm_set_option: null name");
  std::lock_guard<std::mutex> lk(g_opt_mu);
  for (OptEntry& e : g_opts)
    if (!strcmp(e.name, name)) {
      if (value < e.lo || value > e.hi) return blm_fail(BLM_ERR_INVALID, "blm_set_option: %s takes %d..%d", name, e.lo, e.hi);
      e.value = value; e.inited = true;
      return BLM_OK;
    }
  return blm_fail(BLM_ERR_INVALID, "blm_set_option: unknown option '%s'", name);
}

extern "C" int blm_get_option(const char* name, int* value) {
  if (!name || !value) return blm_fail(BLM_ERR_INVALID, "blm_get_option: null argument");
  std::lock_guard<std::mutex> lk(g_opt_mu);
  for (OptEntry& e : g_opts)
    if (!strcmp(e.name, name)) { opt_init(e); *value = e.value; return BLM_OK; }
  return blm_fail(BLM_ERR_INVALID, "blm_get_option: unknown option '%s'", name);
}


// ---- what THIS chip sustains: a bare v_mfma_f32_32x32x2_f32 loop, no memory traffic --------------------------------
// bench.py times it (HIP events) right after the headline run: boxes of the pool hold different clocks under matrix load (round 4:
// 134.8 TFLOP/s = 2.05 GHz on one, 143-147 TFLOP/s = 2.2-2.3 GHz on the others), and the step follows the clock.  512 workgroups of
// 4 waves = two waves per SIMD on every CU, four independent accumulator chains per wave, `iters` x 4 MFMAs per wave.
namespace {
using probe_f32x16 = __attribute__((ext_vector_type(16))) float;
__global__ __launch_bounds__(256) void mfma_probe_kernel(float* out, int iters, float seed) {
  probe_f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  const float x = seed + threadIdx.x * 1e-3f, y = seed * 0.5f + threadIdx.x * 2e-3f;
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
}  // namespace

extern "C" int64_t blm_mfma_probe_ws_floats(void) { return 512 * 256; }

extern "C" int blm_mfma_probe(float* ws, int iters, double* flops, void* stream) {
  if (!ws || iters < 1 || iters > (1 << 22)) return blm_fail(BLM_ERR_INVALID, "blm_mfma_probe: bad arguments");
  hipLaunchKernelGGL(mfma_probe_kernel, dim3(512), dim3(256), 0, static_cast<hipStream_t>(stream), ws, iters, 0.37f);
  BLM_HIP(hipGetLastError());
  if (flops) *flops = 512.0 * 4.0 * (double)iters * 4.0 * 4096.0;  // workgroups x waves x iterations x MFMAs x 2*32*32*2 flops
  return BLM_OK;
}
