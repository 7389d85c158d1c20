// Instantiations of the fp32 MFMA GEMM for one operand layout (separate TU: parallel build).
#include "gemm_f32_mfma.h"

namespace blm {
template int launch_op<BLM_GEMM_TN, false>(const GemmP&, hipStream_t);
}  // namespace blm
