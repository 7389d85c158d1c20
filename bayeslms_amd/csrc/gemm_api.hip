// C-ABI entry point of the fp32 MFMA GEMM family: argument validation and dispatch.
#include <cstdlib>
#include <cstring>

#include "gemm_f32_mfma.h"
#include "gemm_plan.h"

namespace blm {
extern template int launch_op<BLM_GEMM_NT, false>(const GemmP&, hipStream_t);
extern template int launch_op<BLM_GEMM_NT, true>(const GemmP&, hipStream_t);
extern template int launch_op<BLM_GEMM_NN, false>(const GemmP&, hipStream_t);
extern template int launch_op<BLM_GEMM_NN, true>(const GemmP&, hipStream_t);
extern template int launch_op<BLM_GEMM_TN, false>(const GemmP&, hipStream_t);
}  // namespace blm

using namespace blm;

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static int g_gemm_mode = -1;  // -1: not read yet (BLM_GEMM_MODE env: "bf16x3" or "1")

extern "C" int blm_get_gemm_mode(void) {
  if (g_gemm_mode < 0) {
    const char* e = getenv("BLM_GEMM_MODE");
    g_gemm_mode = !e ? BLM_GEMM_MODE_F32 : (!strcmp(e, "bf16x3") || !strcmp(e, "1")) ? BLM_GEMM_MODE_BF16X3
                  : (!strcmp(e, "bf16x6") || !strcmp(e, "2")) ? BLM_GEMM_MODE_BF16X6 : BLM_GEMM_MODE_F32;
  }
  return g_gemm_mode;
}

extern "C" int blm_set_gemm_mode(int mode) {
  if (mode != BLM_GEMM_MODE_F32 && mode != BLM_GEMM_MODE_BF16X3 && mode != BLM_GEMM_MODE_BF16X6) return blm_fail(BLM_ERR_INVALID, "blm_set_gemm_mode: unknown mode");
  g_gemm_mode = mode;
  return BLM_OK;
}

extern "C" int blm_gemm(const blm_gemm_args* a, void* stream) {
  if (!a) return blm_fail(BLM_ERR_INVALID, "blm_gemm: null args");
  if (a->abi_version != BLM_ABI_VERSION) return blm_fail(BLM_ERR_ABI, "blm_gemm: abi_version mismatch");
  if (a->M < 0 || a->N < 0 || a->K < 0) return blm_fail(BLM_ERR_INVALID, "blm_gemm: negative dimension");
  if (a->M == 0 || a->N == 0) return BLM_OK;
  if (!a->A || !a->B || !a->C) return blm_fail(BLM_ERR_INVALID, "blm_gemm: null operand");
  if (a->op < BLM_GEMM_NT || a->op > BLM_GEMM_TN) return blm_fail(BLM_ERR_INVALID, "blm_gemm: bad op");
  const int amin = a->op == BLM_GEMM_TN ? a->M : a->K, bmin = a->op == BLM_GEMM_NT ? a->K : a->N;
  if (a->lda < amin || a->ldb < bmin || a->ldc < a->N) return blm_fail(BLM_ERR_INVALID, "blm_gemm: leading dimension too small");
  GemmP p{};
  p.M = a->M; p.N = a->N; p.K = a->K;
  p.A = a->A; p.lda = a->lda; p.B = a->B; p.ldb = a->ldb; p.C = a->C; p.ldc = a->ldc;
  p.alpha = a->alpha; p.flags = a->flags; p.epi = a->epilogue;
  p.bias = a->bias; p.aux = a->aux; p.coef = a->coef;
  p.vb = a->var_b; p.C2 = a->C2; p.wg_mu = a->wg_mu; p.vc = a->var_c;
  p.kl_lambda = a->kl_lambda; p.kl_inv_n = a->kl_inv_n;
  p.drop_on = a->drop_p > 0.f;
  if (p.drop_on) {
    if (a->drop_B <= 0 || a->M % a->drop_B != 0) return blm_fail(BLM_ERR_INVALID, "blm_gemm: drop_B must divide M");
    const double t = (double)a->drop_p * 4294967296.0;
    p.drop_thr = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
    p.drop_inv_keep = a->drop_p < 1.f ? 1.f / (1.f - a->drop_p) : 0.f;
    p.drop_rng = a->drop_rng;
    p.drop_B = a->drop_B; p.drop_col_offset = a->drop_col_offset;
    p.drop_global_cols = a->drop_global_cols > 0 ? a->drop_global_cols : a->drop_B;
  }
  p.colsum_a = a->colsum_a;
  if (a->colsum_a && a->op != BLM_GEMM_TN) return blm_fail(BLM_ERR_INVALID, "blm_gemm: colsum_a needs op TN");
  p.drop_quad = (a->N % 4 == 0);
  p.eps_quad = (a->epilogue == BLM_EPI_BAYES_WGRAD && a->N % 4 == 0 && !blm::INTERLEAVE && !a->var_c.eps) ? 1 : 0;
  p.split = blm_get_gemm_mode() == BLM_GEMM_MODE_BF16X3 ? 3 : (blm_get_gemm_mode() == BLM_GEMM_MODE_BF16X6 ? 6 : 0);
  p.a_vec = aligned16(a->A) && (a->lda % 4 == 0);
  p.b_vec = aligned16(a->B) && (a->ldb % 4 == 0);
  switch (a->epilogue) {
    case BLM_EPI_NONE: break;
    case BLM_EPI_BIAS: case BLM_EPI_BIAS_GELU:
      if (!a->bias) return blm_fail(BLM_ERR_INVALID, "blm_gemm: epilogue needs bias"); break;
    case BLM_EPI_MUL_DGELU:
      if (!a->aux) return blm_fail(BLM_ERR_INVALID, "blm_gemm: epilogue needs aux"); break;
    case BLM_EPI_GP_MIX:
      if (!a->bias || !a->coef) return blm_fail(BLM_ERR_INVALID, "blm_gemm: GP epilogue needs bias and coef"); break;
    case BLM_EPI_MUL_DGP_MIX:
      if (!a->aux || !a->coef) return blm_fail(BLM_ERR_INVALID, "blm_gemm: GP epilogue needs aux and coef"); break;
    case BLM_EPI_BAYES_WGRAD:
      if (a->op != BLM_GEMM_TN) return blm_fail(BLM_ERR_INVALID, "blm_gemm: BAYES_WGRAD needs op TN");
      if (!a->C2 || !a->wg_mu || !a->var_c.lgstd) return blm_fail(BLM_ERR_INVALID, "blm_gemm: BAYES_WGRAD needs C2, wg_mu, var_c.lgstd");
      if (a->var_c.row_lo < 0 || a->var_c.srows < 0 || a->var_c.row_lo + a->var_c.srows > a->M)
        return blm_fail(BLM_ERR_INVALID, "blm_gemm: var_c row window outside W");
      break;
    default: return blm_fail(BLM_ERR_INVALID, "blm_gemm: unknown epilogue");
  }
  const bool samp = a->var_b.lgstd != nullptr;
  if (samp) {
    if (a->op == BLM_GEMM_TN) return blm_fail(BLM_ERR_INVALID, "blm_gemm: var_b not valid for TN");
    // B source matrix is W: NT -> (N x K), NN -> (K x N)
    const int wrows = a->op == BLM_GEMM_NT ? a->N : a->K;
    p.vb_cols = a->op == BLM_GEMM_NT ? a->K : a->N;
    if (p.vb_cols % 4 != 0 || !aligned16(a->var_b.lgstd) || (a->var_b.eps && !aligned16(a->var_b.eps)) || !p.b_vec)
      return blm_fail(BLM_ERR_INVALID, "blm_gemm: fused sampling needs cols % 4 == 0 and 16-byte aligned mu/lgstd/eps");
    if (a->var_b.row_lo < 0 || a->var_b.srows < 0 || a->var_b.row_lo + a->var_b.srows > wrows)
      return blm_fail(BLM_ERR_INVALID, "blm_gemm: var_b row window outside W");
  }
  // fast path: aligned operands whose contiguous extents are multiples of 4 (and >= 4)
  {
    const int ac = a->op == BLM_GEMM_TN ? a->M : a->K, bc = a->op == BLM_GEMM_NT ? a->K : a->N;
    p.fast = p.a_vec && p.b_vec && ac % 4 == 0 && bc % 4 == 0 && ac >= 4 && bc >= 4;
    // the LDS-DMA loaders address an operand with 32-bit byte offsets from a scalar base
    const long arows = a->op == BLM_GEMM_TN ? a->K : a->M, brows = a->op == BLM_GEMM_NT ? a->N : a->K;
    if (arows * (long)a->lda * 4 >= (1L << 32) || brows * (long)a->ldb * 4 >= (1L << 32)) p.fast = false;
  }
  {
    const PlanKey key = plan_key(a);
    if ((key.fast != 0) != (p.fast != 0)) return blm_fail(BLM_ERR_INVALID, "blm_gemm: planner and launcher disagree on the fast path");
    const Plan pl = choose_plan(key);
    p.plan_tile = pl.tile; p.plan_splits = pl.splits;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (p.colsum_a && !p.fast) {  // odd shapes: the guarded-loader kernel does not fuse it
    const int rc = blm_colsum(a->A, a->lda, a->colsum_a, a->K, a->M, 1, stream);
    if (rc) return rc;
    p.colsum_a = nullptr;
  }
  switch (a->op) {
    case BLM_GEMM_NT: return samp ? launch_op<BLM_GEMM_NT, true>(p, st) : launch_op<BLM_GEMM_NT, false>(p, st);
    case BLM_GEMM_NN: return samp ? launch_op<BLM_GEMM_NN, true>(p, st) : launch_op<BLM_GEMM_NN, false>(p, st);
    default: return launch_op<BLM_GEMM_TN, false>(p, st);
  }
}

