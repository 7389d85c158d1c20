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

// Extent checks shared by the entry points (found by the host-side UBSan build, tests/test_sanitizer_cpu.py): a launcher's size
// arithmetic -- ceil divisions on int extents, products of three extents, byte counts -- must not overflow whatever the caller
// passes.  An entry point accepts extents in [0, 2^30] whose product is at most 2^40 elements (4 TB of fp32: far beyond the
// 288 GB of HBM), and says BLM_ERR_INVALID otherwise.
#include <initializer_list>
namespace blm {
constexpr long kMaxExtent = 1L << 30, kMaxElems = 1L << 40;
inline bool extents_ok(std::initializer_list<long> dims) {
  long prod = 1;
  for (long d : dims) {
    if (d < 0 || d > kMaxExtent) return false;
    if (d > 1) {
      if (prod > kMaxElems / d) return false;
      prod *= d;
    }
  }
  return true;
}
}  // namespace blm

// Kernel-selection options (blm_set_option / blm_get_option, include/bayeslm.h): every switch that picks between two BUILT
// forms of a kernel lives here -- one registry, settable at run time (so the GPU tests run both forms in one process), each
// initialised from its BLM_* environment variable on first use.  INTEGRATION.md lists them with the test that covers each.
namespace blm {
enum Opt { OPT_ATTN_HPW = 0, OPT_ATTN_SHORT, OPT_ATTN_VALU, OPT_LSTM_GEMV, OPT_LSTM_PIPE, OPT_LSTM_TAIL, OPT_DETERMINISTIC, OPT_LSTM_MB2, OPT_COUNT };
int option(Opt o);
}  // namespace blm
