// Host-side error plumbing of the C ABI: status codes + thread-local message.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/bayeslm.h"

int blm_fail(int status, const char* fmt, ...);

#define BLM_HIP(expr)                                                                        \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) return blm_fail(BLM_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)
