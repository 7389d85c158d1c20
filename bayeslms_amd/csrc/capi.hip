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
