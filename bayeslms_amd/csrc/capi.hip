// Library info + error state of libbayeslm_hip.so.
#include <cstdarg>
#include <cstdio>
#include <cstring>

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
