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
    case BLM_EPI_CE_PART: return blm_fail(BLM_ERR_INVALID, "blm_gemm: BLM_EPI_CE_PART is internal to blm_linear_nll");
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
    p.plan_tile = pl.tile; p.plan_splits = pl.splits; p.plan_cus = pl.cus;
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



// ---------------------------------------------------------------------------------------------------------------------
// Inference: per-row negative log-likelihood of a linear decoder WITHOUT materialising the logits (the
// reference's evaluate() / scorer compute decoder(x) (M x V floats) and log_softmax over it, train.py:452-455,
// compute_sentence_scores...py:157-170).  The decoder GEMM's epilogue leaves, per (row, column tile), the maximum and the sum
// of exp(logit - maximum), and the target's logit; this kernel folds the tiles of a row: nll = max + log(sum) - logit[target].
namespace blm {
__global__ __launch_bounds__(256) void ce_part_finish_kernel(const float* __restrict__ part, const float* __restrict__ tlogit,
                                                             float* __restrict__ nll, float* __restrict__ lse_out, int M, int gn) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* p = part + (long)row * gn * 2;
  float m = -INFINITY;
  for (int j = lane; j < gn; j += 64) m = fmaxf(m, p[2 * j]);
  m = wave_max(m);
  float s = 0.f;
  for (int j = lane; j < gn; j += 64) s += p[2 * j + 1] * __expf(p[2 * j] - m);
  s = wave_sum(s);
  if (lane == 0) {
    const float lse = m + __logf(s);
    nll[row] = lse - tlogit[row];
    if (lse_out) lse_out[row] = lse;
  }
}
}  // namespace blm

extern "C" int64_t blm_linear_nll_ws_floats(int M, int N) {
  if (!blm::extents_ok({M, N})) return 0;
  return (int64_t)2 * M * ((N + 63) / 64) + M;  // [M][column tiles][2] partials (64-column tiles at most) + the target logits
}

extern "C" int blm_linear_nll(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, const int64_t* tgt,
                              float* nll, float* lse, float* ws, int M, int N, int K, void* stream) {
  if (M < 0 || N <= 0 || K <= 0) return blm_fail(BLM_ERR_INVALID, "blm_linear_nll: bad shape");
  if (M == 0) return BLM_OK;
  if (!x || !w || !tgt || !nll || !ws || ldx < K || ldw < K) return blm_fail(BLM_ERR_INVALID, "blm_linear_nll: bad arguments");
  if (N % 4 != 0 || !aligned16(ws) || (bias && !aligned16(bias)))
    return blm_fail(BLM_ERR_UNSUPPORTED, "blm_linear_nll: needs N %% 4 == 0 and 16-byte aligned bias / workspace");
  blm_gemm_args a{};
  a.abi_version = BLM_ABI_VERSION;
  a.op = BLM_GEMM_NT; a.M = M; a.N = N; a.K = K;
  a.A = x; a.lda = (int)ldx; a.B = w; a.ldb = (int)ldw; a.C = ws; a.ldc = N;
  a.alpha = 1.f; a.epilogue = BLM_EPI_BIAS_GELU;  // planning key of an epilogue that cannot take K slices
  GemmP p{};
  p.M = M; p.N = N; p.K = K;
  p.A = x; p.lda = (int)ldx; p.B = w; p.ldb = (int)ldw; p.C = ws; p.ldc = N;
  p.alpha = 1.f; p.epi = BLM_EPI_CE_PART;
  p.bias = bias; p.aux = ws;
  p.ce_tgt = reinterpret_cast<const long long*>(tgt);
  p.ce_tlogit = ws + (int64_t)2 * M * ((N + 63) / 64);
  p.split = blm_get_gemm_mode() == BLM_GEMM_MODE_BF16X3 ? 3 : (blm_get_gemm_mode() == BLM_GEMM_MODE_BF16X6 ? 6 : 0);
  p.a_vec = aligned16(x) && (ldx % 4 == 0);
  p.b_vec = aligned16(w) && (ldw % 4 == 0);
  p.fast = p.a_vec && p.b_vec && K % 4 == 0 && K >= 4 &&
           (long)M * ldx * 4 < (1L << 32) && (long)N * ldw * 4 < (1L << 32);
  const PlanKey key = plan_key(&a);
  const Plan pl = choose_plan(key);
  p.plan_tile = (key.fast != 0) == (p.fast != 0) ? pl.tile : 11;
  p.plan_splits = 1;
  p.plan_cus = pl.cus;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // the target's logit is written by the one lane whose column window holds it: a target outside [0, N) (a padding id, -1)
  // would leave its slot uninitialised -- every slot starts as NaN, so such a row's NLL is NaN, not garbage
  BLM_HIP(hipMemsetAsync(p.ce_tlogit, 0xFF, (size_t)M * sizeof(float), st));
  const int rc = launch_op<BLM_GEMM_NT, false>(p, st);
  if (rc) return rc;
  // column tiles of the launch: 64 columns on tiles 11 / 21 (and on the guarded 64x64 kernel), 128 otherwise
  const int tile = p.fast ? p.plan_tile : 11;
  const int bn = (tile == 11 || tile == 21) ? 64 : 128;
  const int gn = (N + bn - 1) / bn;
  hipLaunchKernelGGL(ce_part_finish_kernel, dim3((M + 3) / 4), dim3(256), 0, st, ws, p.ce_tlogit, nll, lse, M, gn);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// Two-model scoring (compute_sentence_scores_bayes_jianwei.py:157-168): the reference interpolates the LOGITS of two language
// models, alpha * (x1 W1^T + b1) + (1 - alpha) * (x2 W2^T + b2), then takes log_softmax -- two (M x V) matrices written, mixed
// and read back.  The mixture is ONE product: [alpha x1 | (1 - alpha) x2] . [W1 | W2]^T + (alpha b1 + (1 - alpha) b2), so the
// decoder + cross-entropy launch of blm_linear_nll takes it with K = K1 + K2 and the logits of neither model are ever stored.
// The operands are packed into caller-owned workspace: the activations per call (M x (K1 + K2), small beside the product's
// 2 M V (K1 + K2) flops), the weights and the bias only when asked (pack_w: the first call of a scoring run).
namespace blm {
// dst[r] = [sa * a[r] | sb * b[r]] for r < rows, zeros for rows <= r < rows_out (vocabulary padding)
__global__ __launch_bounds__(256) void pack2_kernel(float* __restrict__ dst, long ldd, const float* __restrict__ a, long lda, int ka,
                                                    float sa, const float* __restrict__ b, long ldb, int kb, float sb, long rows,
                                                    long rows_out) {
  const long kq = (long)(ka + kb) / 4, n = rows_out * kq;  // float4 columns: ka and kb are multiples of 4
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long r = i / kq;
    const int c = (int)(i - r * kq) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < rows) {
      if (c < ka) { v = *reinterpret_cast<const float4*>(a + r * lda + c); v.x *= sa; v.y *= sa; v.z *= sa; v.w *= sa; }
      else { v = *reinterpret_cast<const float4*>(b + r * ldb + (c - ka)); v.x *= sb; v.y *= sb; v.z *= sb; v.w *= sb; }
    }
    *reinterpret_cast<float4*>(dst + r * ldd + c) = v;
  }
}
// mixed bias; the padding columns get -inf: exp(-inf - max) = 0, they never reach the soft-max sum
__global__ __launch_bounds__(256) void bias_mix_kernel(float* __restrict__ dst, const float* __restrict__ b1, float s1,
                                                       const float* __restrict__ b2, float s2, int n, int n_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (b1 ? s1 * b1[i] : 0.f) + (b2 ? s2 * b2[i] : 0.f);
  else if (i < n_out) dst[i] = -INFINITY;
}
}  // namespace blm

extern "C" int64_t blm_linear_nll2_wcat_floats(int N, int K1, int K2) {
  if (!blm::extents_ok({N, K1}) || !blm::extents_ok({N, K2})) return 0;
  const int64_t Np = ((int64_t)N + 3) / 4 * 4;  // vocabulary padded to the vectorised epilogue's multiple of 4
  return Np * (K1 + K2) + Np;                  // [W1 | W2] and the mixed bias
}

extern "C" int64_t blm_linear_nll2_ws_floats(int M, int N, int K1, int K2) {
  if (!blm::extents_ok({M, N}) || !blm::extents_ok({M, K1}) || !blm::extents_ok({M, K2})) return 0;
  return (int64_t)M * (K1 + K2) + blm_linear_nll_ws_floats(M, (N + 3) / 4 * 4);  // packed activations, then blm_linear_nll's own workspace
}

extern "C" int blm_linear_nll2(const float* x1, int64_t ldx1, const float* w1, int64_t ldw1, const float* b1, int K1,
                               const float* x2, int64_t ldx2, const float* w2, int64_t ldw2, const float* b2, int K2, float alpha,
                               const int64_t* tgt, float* nll, float* lse, float* wcat, int pack_w, float* ws, int M, int N,
                               void* stream) {
  if (M < 0 || N <= 0 || K1 <= 0 || K2 <= 0) return blm_fail(BLM_ERR_INVALID, "blm_linear_nll2: bad shape");
  if (M == 0) return BLM_OK;
  if (!x1 || !x2 || !w1 || !w2 || !tgt || !nll || !wcat || !ws || ldx1 < K1 || ldx2 < K2 || ldw1 < K1 || ldw2 < K2)
    return blm_fail(BLM_ERR_INVALID, "blm_linear_nll2: bad arguments");
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (K1 % 4 || K2 % 4 || ldx1 % 4 || ldx2 % 4 || ldw1 % 4 || ldw2 % 4 || !al(x1) || !al(x2) || !al(w1) || !al(w2) || !al(wcat) || !al(ws))
    return blm_fail(BLM_ERR_UNSUPPORTED, "blm_linear_nll2: needs K1, K2 and the row strides to be multiples of 4 and 16-byte aligned operands");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int K = K1 + K2, Np = (N + 3) / 4 * 4;
  float* bias = wcat + (int64_t)Np * K;
  auto blocks = [](long n) { long b = (n + 255) / 256; return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b)); };
  if (pack_w) {
    hipLaunchKernelGGL(pack2_kernel, dim3(blocks((long)Np * K / 4)), dim3(256), 0, st, wcat, (long)K, w1, (long)ldw1, K1, 1.f, w2, (long)ldw2,
                       K2, 1.f, (long)N, (long)Np);
    hipLaunchKernelGGL(bias_mix_kernel, dim3((Np + 255) / 256), dim3(256), 0, st, bias, b1, alpha, b2, 1.f - alpha, N, Np);
  }
  hipLaunchKernelGGL(pack2_kernel, dim3(blocks((long)M * K / 4)), dim3(256), 0, st, ws, (long)K, x1, (long)ldx1, K1, alpha, x2, (long)ldx2, K2,
                     1.f - alpha, (long)M, (long)M);
  BLM_HIP(hipGetLastError());
  return blm_linear_nll(ws, K, wcat, K, bias, tgt, nll, lse, ws + (int64_t)M * K, M, Np, K, stream);
}
