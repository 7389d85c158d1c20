// Instantiations of the fp32 MFMA GEMM for one operand layout (separate TU: parallel build).
#include "gemm_f32_mfma.h"

namespace blm {
template int launch_op<BLM_GEMM_NN, false>(const GemmP&, hipStream_t);
template int launch_op<BLM_GEMM_NN, true>(const GemmP&, hipStream_t);
}  // namespace blm

#ifdef BLM_GEMM_PROF
extern "C" int blm_debug_prof_nn(unsigned long long* out, int reset) {
  if (out) (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(blm::blm_prof), 4 * sizeof(unsigned long long));
  if (reset) { unsigned long long z[4] = {0, 0, 0, 0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(blm::blm_prof), z, sizeof(z)); }
  return 0;
}
#endif
#ifdef BLM_GEMM_LIFE
extern "C" int blm_debug_store_mode_nn(int mode) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(blm::blm_dbg_store), &mode, sizeof(int)); }
extern "C" int blm_debug_wg_life_nn(long long* out, int nwg) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(blm::blm_wg_life), (size_t)4 * nwg * sizeof(long long));
}
#endif
