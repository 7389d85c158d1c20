/*
 * bayeslm.h -- C ABI of libbayeslm_hip.so (MI355X / gfx950 only).
 *
 * The reference (AmourWaltz/BayesLMs) has no FFI of its own: its hot path is
 * Python calling ATen/cuDNN ops.  Each entry point below
 * replaces the device work of one group of those call sites; the citation on
 * each declaration is the reference line(s) it stands in for
 * (paths relative to steps/pytorchnn/).
 *
 * Contract:
 *   - plain pointers + sizes, no torch types; every pointer is DEVICE memory
 *     owned by the caller (PyTorch's caching allocator in the Python host);
 *   - the library allocates nothing persistent and keeps no references;
 *   - every call is asynchronous on the hipStream_t passed in (void* here so
 *     the header needs no HIP include); no internal synchronisation;
 *   - returns 0 (BLM_OK) or a negative blm_status; blm_last_error() gives a
 *     thread-local message; nothing throws or aborts;
 *   - all tensors fp32 row-major unless stated; token ids int64.
 */
#ifndef BAYESLM_H
#define BAYESLM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BLM_ABI_VERSION 1u

typedef enum blm_status {
  BLM_OK = 0,
  BLM_ERR_INVALID = -1,   /* bad shape / null pointer / misaligned operand   */
  BLM_ERR_ABI = -2,       /* abi_version mismatch                            */
  BLM_ERR_HIP = -3,       /* a HIP runtime call failed (message has detail)  */
  BLM_ERR_UNSUPPORTED = -4
} blm_status;

/* Philox4x32-10 stream ids: counter = (block_lo, block_hi, stream, step),
 * key = seed.  Must match oracle/philox.py. */
#define BLM_STREAM_WEIGHT  0x10000000u  /* class in the top 4 bits + tensor id (low 28 bits) */
#define BLM_STREAM_DROPOUT 0x20000000u  /* + site id */

typedef struct blm_rng {
  uint64_t seed;    /* Philox key                                             */
  uint32_t stream;  /* BLM_STREAM_* + id                                      */
  uint32_t step;    /* optimisation step: same (seed,stream,step) => same eps */
} blm_rng;

/* --------------------------------------------------------------------------
 * Library info
 * ------------------------------------------------------------------------ */
uint32_t    blm_abi_version(void);
const char* blm_last_error(void);
/* Fills gfx arch name (<= 31 chars), CU count and LDS bytes/CU of device
 * `device`; used by the host to refuse anything that is not gfx950. */
int blm_query(int device, char* arch32, int* n_cu, int* lds_bytes);

/* --------------------------------------------------------------------------
 * Variational weights: W = mu + exp(lgstd) * eps, and the KL term
 * ------------------------------------------------------------------------ */

/* Describes the variational part of a weight matrix W (rows x cols, leading
 * dimension ld): rows [row_lo, row_lo + srows) carry noise, lgstd is
 * (srows x cols) with leading dimension cols.
 *   BayesLinear: row_lo = 0, srows = rows        (model.py:1083-1107)
 *   Bayes2LSTM : row_lo = (pos-1)*H, srows = H   (model.py:716-725)
 * eps != NULL  -> injected noise (srows x cols), used for parity tests;
 * eps == NULL  -> Philox noise from `rng`, element index = r*cols + c with
 *                 r relative to row_lo.
 * lgstd == NULL -> no noise at all (eval mode / mean weights). */
typedef struct blm_variational {
  const float* lgstd;
  const float* eps;
  int32_t row_lo;
  int32_t srows;
  blm_rng rng;
} blm_variational;

/* Materialise W (rows x cols, contiguous) from mu in one pass and, if
 * kl_out != NULL, add  kl_weight * mean(mu_s^2 - 2 lgstd + exp(2 lgstd))/2
 * (mu_s = the noisy rows of mu) to *kl_out.
 * Replaces model.py:1083-1107 (BayesLinear.sample_weight_diff/_flat_weights),
 * :668-732 (Bayes2LSTM.sample_weight_diff/flat_parameters), :1243-1249.
 * Bias vectors: pass rows = len, cols = 1. */
int blm_sample_weight(const float* mu, int64_t rows, int64_t cols, const blm_variational* v,
                      float* w_out, float* kl_out, float kl_weight, void* stream);

/* Backward of blm_sample_weight: dmu[r,c] += dW[r,c] for all rows;
 * dlgstd[r',c] += dW[row_lo+r',c] * eps * exp(lgstd) on the noisy rows (eps
 * injected or regenerated from the counter).  SURVEY.md Appendix C. */
int blm_sample_weight_bwd(const float* dw, int64_t rows, int64_t cols, const blm_variational* v,
                          float* dmu, float* dlgstd, void* stream);

/* Write the N(0,1) stream itself (n floats): test/debug visibility of the
 * generator the fused paths use. */
int blm_philox_normal(float* out, int64_t n, const blm_rng* rng, void* stream);

/* *out += weight * mean(mu^2 - 2 lg + exp(2 lg) [- 1]) / 2 over an
 * (rows x cols) window; mu has leading dimension ld_mu, lg is contiguous.
 * Replaces model.py:1109-1125, :734-765, :1251-1256, :1816-1826. */
int blm_kl_mean_fwd(const float* mu, int64_t ld_mu, const float* lgstd, int64_t rows, int64_t cols,
                    int minus_one, float weight, float* out, void* stream);
/* dmu += g * mu / n ; dlg += g * (exp(2 lg) - 1) / n   (n = rows*cols; g is
 * the upstream scalar gradient times the same weight, read from device).
 * SURVEY.md Appendix C. */
int blm_kl_mean_bwd(const float* mu, int64_t ld_mu, const float* lgstd, int64_t rows, int64_t cols,
                    const float* g_dev, float weight, float* dmu, int64_t ld_dmu, float* dlgstd, void* stream);

/* All variational tensors of a module in ONE launch per direction (Bayes2LSTM samples 8 tensors and takes the KL
 * of 4 of them every step, model.py:668-732 + :734-765: 8 + 4 + 4 + 8 launches of the single-tensor entry points).
 * Item i:  W_i = mu_i (+ exp(lgstd_i) * eps on the rows [row_lo, row_lo + srows))      -> w_out (NULL: KL only)
 *          *kl_out += kl_weight_i * sum(mu_s^2 - 2 lgstd + exp(2 lgstd) - kl_minus_i) / (2 * srows * cols)
 *                     (kl_weight 0: no KL from this item; kl_weight = n / count re-bases the mean, model.py:737-740).
 * blm_variational_group_fwd sets *kl_out = 0 first (kl_out may be NULL when no item has a KL weight).
 * Backward, g = *kl_grad (device scalar, NULL = no KL gradient), n = srows * cols:
 *          dmu_i    += dw_i                          (all rows; dw NULL: skipped)
 *          dmu_i    += g * kl_weight_i * mu / n      (noisy rows)
 *          dlgstd_i += dw_i * eps * exp(lgstd) + g * kl_weight_i * (exp(2 lgstd) - 1) / n
 * = blm_sample_weight_bwd + blm_kl_mean_bwd of every item.  At most BLM_VAR_GROUP_MAX items. */
#define BLM_VAR_GROUP_MAX 16
typedef struct blm_var_item {
  const float* mu;
  int64_t rows, cols;
  blm_variational v;
  float* w_out;
  float kl_weight;
  float kl_minus;
  const float* dw;
  float* dmu;
  float* dlgstd;
} blm_var_item;
int blm_variational_group_fwd(const blm_var_item* items, int32_t n, float* kl_out, void* stream);
int blm_variational_group_bwd(const blm_var_item* items, int32_t n, const float* kl_grad, void* stream);

/* --------------------------------------------------------------------------
 * fp32 MFMA GEMM family (v_mfma_f32_32x32x2_f32, LDS tiled)
 * ------------------------------------------------------------------------ */
typedef enum blm_gemm_op {
  BLM_GEMM_NT = 0, /* C[M,N] = A[M,K] * B[N,K]^T   forward  F.linear      */
  BLM_GEMM_NN = 1, /* C[M,N] = A[M,K] * B[K,N]     dgrad    dX = dY W     */
  BLM_GEMM_TN = 2  /* C[M,N] = A[K,M]^T * B[K,N]   wgrad    dW = dY^T X   */
} blm_gemm_op;

typedef enum blm_epilogue {
  BLM_EPI_NONE = 0,
  BLM_EPI_BIAS = 1,        /* C = acc + bias[n]                                        */
  BLM_EPI_BIAS_GELU = 2,   /* z = acc + bias[n]; C = gelu_erf(z)*keep; aux[m,n] = gelu_erf'(z)*keep (if aux) */
  BLM_EPI_MUL_DGELU = 3,   /* C = acc * aux[m,n]   (aux as written by BIAS_GELU: GELU' and dropout in one factor) */
  BLM_EPI_BAYES_WGRAD = 4, /* TN only, C = dmu, C2 = dlgstd, see below                 */
  BLM_EPI_GP_MIX = 5,      /* z = acc + bias; aux = z; C = sum_i act_i(z) coef[i,n]     */
  BLM_EPI_MUL_DGP_MIX = 6, /* C = acc * sum_i act_i'(aux) coef[i,n]; C2 (optional) = acc (after dropout) */
  BLM_EPI_CE_PART = 7      /* internal to blm_linear_nll (blm_gemm refuses it): no C, per-tile softmax partials instead */
} blm_epilogue;

#define BLM_GEMM_ACCUMULATE 1u /* C (+= C2) accumulate into existing contents */

typedef struct blm_gemm_args {
  uint32_t abi_version; /* BLM_ABI_VERSION */
  int32_t op;           /* blm_gemm_op */
  int32_t M, N, K;
  const float* A; int64_t lda;
  const float* B; int64_t ldb;
  float* C;       int64_t ldc;
  float alpha;          /* C = alpha * acc (+ epilogue)                        */
  uint32_t flags;       /* BLM_GEMM_ACCUMULATE                                 */
  int32_t epilogue;     /* blm_epilogue */
  const float* bias;    /* [N]                                                 */
  float* aux;           /* [M,N] ld = ldc: pre-activation (written or read)    */
  const float* coef;    /* GP mixture coefficients [4,N]: tanh,sigmoid,relu,gelu */
  /* Variational B operand (NT, NN): B is the *mean* matrix mu and the tile
   * loader forms W = mu + exp(lgstd)*eps on the fly (fused sampling, no W in
   * HBM).  var_b.lgstd == NULL -> B used as is.
   * model.py:1127-1129 fwd; autograd dX of the same. */
  blm_variational var_b;
  /* BLM_EPI_BAYES_WGRAD (TN; C has the shape of W):
   *   dW  = alpha*acc
   *   C [n,k] (+)= dW + kl_lambda * mu/n_kl      (KL part on noisy rows only)
   *   C2[r,k] (+)= dW*eps*exp(lgstd) + kl_lambda*(exp(2 lgstd)-1)/n_kl
   *               for rows r = n - row_lo in [0, srows)
   * with mu = wg_mu (ld = ldc), var_c describing lgstd/eps/rng, and
   * 1/n_kl = kl_inv_n (the element count of the mean the KL term belongs to:
   * srows*N for BayesLinear, H*(H+E) for a Bayes2LSTM matrix, model.py:737-740).
   * The KL part only touches the noisy rows of mu.
   * (autograd of model.py:1083-1107 + :1109-1125). */
  float* C2;
  const float* wg_mu;
  blm_variational var_c;
  float kl_lambda;
  float kl_inv_n;
  /* Dropout fused into the activation epilogues (drop_p > 0): C is seen as a
   * (rows, drop_B, N) activation, m = row*drop_B + b, mask keyed by the global
   * element (row, drop_col_offset + b, n) of a tensor with drop_global_cols
   * columns.  BIAS_GELU / GP_MIX: C = act(z) * keep/(1-p) (BIAS_GELU also folds keep
   * into aux);  MUL_DGP_MIX: C = acc * keep/(1-p) * act'(aux).  (model.py:1043 dropout) */
  float drop_p;
  blm_rng drop_rng;
  int32_t drop_B, drop_col_offset, drop_global_cols;
  /* TN only (wgrad dW = dY^T X, A = dY): colsum_a[m] += alpha * sum_k A[k,m], i.e. the bias
   * gradient of the same layer, taken from the A tiles the kernel stages anyway. */
  float* colsum_a;
} blm_gemm_args;

int blm_gemm(const blm_gemm_args* a, void* stream);

/* Arithmetic of the GEMM family's matrix instruction (process-wide; default BLM_GEMM_MODE_F32, or the
 * BLM_GEMM_MODE=bf16x3 environment variable at first use).
 *   F32:    v_mfma_f32_32x32x2_f32 on the fp32 operands -- the parity mode, what every reported number uses.
 *   BF16X3: OPT-IN.  Each fp32 operand value is split into bf16 hi + lo when a wave reads its fragment and
 *           A.B is formed as hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_bf16 (fp32 accumulate): same
 *           memory traffic, 5.3x less matrix-pipe time, 4.5e-6 max relative error at K = 4096 against
 *           3.6e-7.  Lower precision than the reference's fp32: never on by default.  Applies to the
 *           aligned fast path (LDS-DMA loaders) without fused sampling; other launches stay F32.
 *   BF16X6: OPT-IN.  Three parts (hi + mid + lo = the full 24-bit mantissa, an exact representation) and the six
 *           largest part products: an fp32-accurate multiply, 2.7x less matrix-pipe time than F32. */
#define BLM_GEMM_MODE_F32 0
#define BLM_GEMM_MODE_BF16X3 1
#define BLM_GEMM_MODE_BF16X6 2
int blm_set_gemm_mode(int mode);
int blm_get_gemm_mode(void);

/* Inference only: nll[m] = logsumexp_n(x[m,:] . w[n,:] + bias[n]) - (x[m,:] . w[tgt[m],:] + bias[tgt[m]]) without storing the
 * M x N logits -- the decoder product's epilogue keeps one (max, sum of exp) pair per row and column tile, a second small
 * kernel folds them (csrc/gemm_api.hip).  Replaces decoder + log_softmax + gather of train.py:452-455 (evaluate) and
 * compute_sentence_scores_bayes_jianwei.py:157-170 (one model; two models: blm_linear_nll2 below).  bias may be NULL; lse
 * (optional) receives the log-sum-exp per row; a target outside [0, N) gives NaN for its row;
 * ws: blm_linear_nll_ws_floats(M, N) floats, 16-byte aligned, owned by the caller.  N % 4 == 0. */
int64_t blm_linear_nll_ws_floats(int M, int N);
int blm_linear_nll(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, const int64_t* tgt,
                   float* nll, float* lse, float* ws, int M, int N, int K, void* stream);
/* Two-model scoring, reference compute_sentence_scores_bayes_jianwei.py:157-168: per-row NLL of the INTERPOLATED logits
 * alpha * (x1 w1^T + b1) + (1 - alpha) * (x2 w2^T + b2) against tgt, as ONE decoder + cross-entropy launch over the packed
 * operands [alpha x1 | (1 - alpha) x2] (M x (K1 + K2)) and [w1 | w2] (N x (K1 + K2)): neither model's (M x N) logits are stored.
 * wcat: blm_linear_nll2_wcat_floats(N, K1, K2) floats, caller-owned; pack_w != 0 (re)builds [w1 | w2] and the mixed bias in it
 * (the first call of a scoring run; later calls with the same weights and alpha pass 0).  ws: blm_linear_nll2_ws_floats floats
 * per call.  K1, K2 and all row strides multiples of 4, operands 16-byte aligned; any N (the packed vocabulary is padded to a
 * multiple of 4 with zero rows whose bias is -inf); b1 / b2 may be NULL.  A target outside [0, N) gives NaN for its row. */
int64_t blm_linear_nll2_wcat_floats(int N, int K1, int K2);
int64_t blm_linear_nll2_ws_floats(int M, int N, int K1, int K2);
int blm_linear_nll2(const float* x1, int64_t ldx1, const float* w1, int64_t ldw1, const float* b1, int K1,
                    const float* x2, int64_t ldx2, const float* w2, int64_t ldw2, const float* b2, int K2, float alpha,
                    const int64_t* tgt, float* nll, float* lse, float* wcat, int pack_w, float* ws, int M, int N, void* stream);

/* Kernel-selection options: switches that pick between two BUILT forms of a kernel (each form is parity-tested; the defaults are
 * the measured winners).  Settable at run time; each is initialised from its environment variable on first use.
 *   "attn_hpw"   (BLM_ATTN_HPW,   0) heads per workgroup of the T <= 128 attention kernels: 0 = by head count, 1, 2
 *   "attn_short" (BLM_ATTN_SHORT, 1) one-wave-per-head attention forward for T <= 32 (0: the 128-row forward there too)
 *   "attn_valu"  (BLM_ATTN_VALU,  0) 1: the vector-ALU attention kernels also at head_dim 64
 *   "lstm_gemv"  (BLM_LSTM_GEMV,  1) one-wave-per-unit LSTM step kernel for B <= 4 (0: the matrix-core step kernel)
 *   "lstm_pipe"  (BLM_LSTM_PIPE,  1) software-pipelined K loop of the LSTM step kernels
 *   "lstm_tail"  (BLM_LSTM_TAIL,  0) 1: the general (K tail) form of the pipelined LSTM step kernels also for whole chunks
 *   "lstm_mb2"   (BLM_LSTM_MB2,   1) the architecture-search cell's forward step (blm_lstm_search_step_fwd) takes two batch tiles per
 *                                    workgroup at B > 32: the stacked 8H x H weight streams once, one round of workgroups (0: one tile)
 * and one MODE, off by default:
 *   "deterministic" (BLM_DETERMINISTIC, 0) 1: every reduction of the library in a fixed order -- blm_gemm / blm_linear_nll* plans are
 *                                    legalised to ONE K slice (no float atomics into C; colsum_a then has one writer per element),
 *                                    blm_colsum* / blm_gp_coef_grad use one row chunk, the KL sums of blm_sample_weight /
 *                                    blm_variational_group_fwd / blm_kl_mean_fwd go through block partials added by one block (a
 *                                    fixed array of the code object: these calls must then be issued on one stream at a time),
 *                                    blm_embed_bwd runs one wave per vocabulary row in position order.  Same arithmetic, another
 *                                    order of additions: results equal the default mode's to rounding and are bit-identical run to
 *                                    run (tests/test_gpu_deterministic.py); 1.26-1.33 x the default step time.
 * No reference counterpart (the reference leaves kernel selection to the vendor libraries behind torch). */
/* Diagnostic, no reference counterpart: a bare v_mfma_f32_32x32x2_f32 loop (two waves per SIMD on every CU, no memory traffic),
 * `iters` x 4 MFMAs per wave; *flops receives the work of the launch.  The caller times it (bench.py: "chip.bare_mfma_tflops") --
 * what THIS chip sustains under matrix load, which differs between boxes by up to 10 %.  ws: blm_mfma_probe_ws_floats() floats. */
int64_t blm_mfma_probe_ws_floats(void);
int blm_mfma_probe(float* ws, int iters, double* flops, void* stream);
int blm_set_option(const char* name, int value);
int blm_get_option(const char* name, int* value);

/* Launch plan of a blm_gemm call: block tile (11 = 64x64, 12 = 64x128, 21 = 128x64, 22 = 128x128 rows x cols on four waves;
 * 28 = 128x128 on EIGHT waves: two waves per SIMD in one barrier domain, aligned operands and K % 32 == 0 only) and
 * number of K slices (> 1: partial sums meet in C through float atomics; legal for the plain, bias and Bayesian-wgrad
 * epilogues.  splits <= -2 = TAIL slicing: the tiles of the whole rounds of workgroup slots are computed in one piece with
 * plain stores, only the tiles of the last, partly filled round are cut |splits| ways).  The reference leaves this to the vendor
 * BLAS behind F.linear (model.py:1127-1129); here it is one explicit rule for every shape (csrc/gemm_plan.hip):
 * override > plan table (exact-shape entries measured inside the benchmark / recipe steps, csrc/gemm_plans.inc,
 * written by tools/gemm_tune.py) > cost model (workgroup-slot rounds x per-tile matrix-pipe efficiency).
 * All of this is host code: none of the functions below touches the GPU. */
typedef struct blm_gemm_plan {
  int32_t tile;
  int32_t splits;   /* >= 1: K slices of every tile; <= -2: only the tail round's tiles are sliced, |splits| ways */
  int32_t source;   /* 0 cost model, 1 plan table, 2 override, 3 plan measured beside a collective (comm window open) */
  float model_us;   /* the cost model's estimate for this plan */
} blm_gemm_plan;
/* The plan blm_gemm would use for `a` now (pointers are only inspected for alignment).  A query is not a launch: an open comm
 * window (below) keeps its time. */
int blm_gemm_plan_query(const blm_gemm_args* a, blm_gemm_plan* out);
/* The planning step of a launch on its own: the same plan, AND its modelled time is taken off an open comm window, exactly as
 * blm_gemm / blm_linear_nll do in front of their launch.  For callers that account for matrix work they issue outside the
 * library, and for the host-side tests of the window (tests/test_gemm_plan_cpu.py). */
int blm_gemm_plan_launch(const blm_gemm_args* a, blm_gemm_plan* out);
/* The cost model's estimate (microseconds) of `a` under a given tile / slice count. */
int blm_gemm_plan_model_us(const blm_gemm_args* a, int tile, int splits, float* us);
/* Tuning tools: force a tile and / or a slice count for every call of this process (0 = no override; also read once
 * from BLM_GEMM_TILE / BLM_GEMM_SPLITK). */
int blm_gemm_plan_override(int tile, int splits);
/* Add (or supersede) a plan-table entry at run time; blm_gemm_plan_clear drops the run-time entries and, with
 * keep_builtin == 0, switches the built-in table off as well (cost model only). */
int blm_gemm_plan_set(int op, int M, int N, int K, int epilogue, int accumulate, int tile, int splits);
int blm_gemm_plan_clear(int keep_builtin);
/* Compute units the planner may count on: 0 = the whole chip (256, the default), or 8..256.  Data-parallel training narrows it
 * while gradient buckets are in flight: RCCL's channel workgroups (256 threads, ~280 registers per lane, 19.7 KB LDS each on
 * gfx950) hold CUs beside the backward GEMMs, and a plan that is exactly ONE round of 256 one-per-CU workgroups would spill a
 * second round.  The plan table (measured on the whole chip) applies at 256 only; below it the cost model plans for `cus`.
 * No reference counterpart (single GPU, run_nnlm_ami_tm.sh:44).  blm_gemm_plan_get_cus returns the current value. */
int blm_gemm_plan_set_cus(int cus);
int blm_gemm_plan_get_cus(void);
/* "A gradient bucket is in flight": adds `us` microseconds (the bucket's expected time on the links) to the comm window; every
 * blm_gemm planned while the window is open takes its own modelled time off it, so the window is kept in DEVICE time although
 * the host enqueues far ahead.  Inside the window a launch uses the plan measured beside a resident collective stand-in where
 * the table has one (csrc/gemm_plans_comm.inc, tools/gemm_tune_comm.py; blm_gemm_plan_query reports source 3), its usual plan
 * otherwise.  us = 0 closes the window (engine.GradReducer.finish).  Only launches (and blm_gemm_plan_launch) debit the window.  blm_gemm_plan_set_comm adds a run-time entry to that
 * table (blm_gemm_plan_clear drops the run-time entries of both tables). */
int blm_gemm_plan_comm_window(float us);
float blm_gemm_plan_comm_window_left(void);
int blm_gemm_plan_set_comm(int op, int M, int N, int K, int epilogue, int accumulate, int tile, int splits);

/* --------------------------------------------------------------------------
 * Surrounding Transformer / LSTM ops (HBM-bound unless stated)
 * ------------------------------------------------------------------------ */

/* out[t,b,:] = drop(enc[ids[t,b],:] * scale + pe[t,:])   (model.py:1284,116-117)
 * pe may be NULL (LSTM: plain lookup + dropout, model.py:218).  Dropout keep
 * mask is keyed by the GLOBAL element (t, col_offset + b, j) of a tensor with
 * global_cols columns so that a W-rank run equals a 1-rank run.
 * p == 0 -> no dropout. */
int blm_embed_fwd(const int64_t* ids, const float* enc, const float* pe, float* out, int T, int B, int D,
                  int64_t vocab, float scale, float p, const blm_rng* rng, int col_offset, int global_cols,
                  void* stream);
/* denc[ids[t,b],:] += scale * keep * dy[t,b,:]  (float atomics). */
int blm_embed_bwd(const int64_t* ids, const float* dy, float* denc, int T, int B, int D, int64_t vocab, float scale,
                  float p, const blm_rng* rng, int col_offset, int global_cols, void* stream);

/* out[t,b,:] = drop(x[t,b,:] + pe[t,:])   (PositionalEncoding.forward, model.py:116-117);
 * backward is blm_dropout with the same key. */
int blm_add_pe_dropout(const float* x, const float* pe, float* out, int T, int B, int D, float p, const blm_rng* rng,
                       int col_offset, int global_cols, void* stream);

/* y = x * keep / (1-p)  (nn.Dropout; same call for backward with x = dy).
 * x is (rows, B, D) with the same global-column keying as above. */
int blm_dropout(const float* x, float* y, int rows, int B, int D, float p, const blm_rng* rng, int col_offset,
                int global_cols, void* stream);
/* The same on rows [row0, row0 + rows) of a longer (T, B, D) tensor: x / y point at the block, the keep mask is the one
 * blm_dropout would give those rows (the layer wavefront of a 2-layer LSTM applies the inter-layer dropout chunk by chunk). */
int blm_dropout_rows(const float* x, float* y, int rows, int row0, int B, int D, float p, const blm_rng* rng, int col_offset,
                     int global_cols, void* stream);

/* Post-LN residual block tail (model.py:1041-1042,1044-1045):
 *   s = x + drop(y);  out = LayerNorm(s) * gamma + beta
 * Saves s_out (optional), mean, rstd per row for backward.  D <= 8192. */
int blm_add_dropout_ln_fwd(const float* x, const float* y, const float* gamma, const float* beta, float* out,
                           float* s_out, float* mean, float* rstd, int rows, int B, int D, float eps_ln, float p,
                           const blm_rng* rng, int col_offset, int global_cols, void* stream);
/* Given dout: ds = LN backward wrt s; dx = ds; dy = ds*keep/(1-p);
 * dgamma/dbeta accumulated (+=) via per-block partials in ws
 * (ws >= 2*ceil(M/32)*D floats... see blm_ln_bwd_ws_floats). */
int64_t blm_ln_bwd_ws_floats(int M, int D);
int blm_add_dropout_ln_bwd(const float* dout, const float* s, const float* gamma, const float* mean, const float* rstd,
                           float* dx, float* dy, float* dgamma, float* dbeta, float* ws, int rows, int B, int D,
                           float p, const blm_rng* rng, int col_offset, int global_cols, void* stream);

/* Fused causal self-attention for qkv packed (T, B, 3*d) [q|k|v], head index
 * = b*nhead + head, head_dim = d/nhead (model.py:876-920).  Scaling
 * head_dim^-0.5, additive causal mask, softmax, dropout p on the
 * probabilities, P.V.  Saves lse (B*nhead, T) for backward.  Any T and any
 * head_dim <= 512 are taken; which kernels run: head_dim 64 -- matrix cores
 * (T <= 128 one tile, longer sequences chunked); 4 / 8 / 16 / 32 at T <= 128 --
 * vector ALU, a query per lane; any other head_dim <= 128 at T <= 128 -- vector
 * ALU, two lanes per row; everything else -- one wave per query, untiled.
 * q/k/v may also be three separate (T,B,d) tensors (BayesMultiheadAttention,
 * model.py:975-977): pass ld_qkv = d. */
int blm_attn_fwd(const float* q, const float* k, const float* v, int64_t ld_qkv, float* out, float* lse, int T, int B,
                 int nhead, int head_dim, float p, const blm_rng* rng, int col_offset, int global_cols, void* stream);
int blm_attn_bwd(const float* q, const float* k, const float* v, int64_t ld_qkv, const float* out, const float* dout,
                 const float* lse, float* dq, float* dk, float* dv, int64_t ld_dqkv, int T, int B, int nhead,
                 int head_dim, float p, const blm_rng* rng, int col_offset, int global_cols, void* stream);
/* Inference on PACKED rows (the n-best scorer keeps only the real tokens of a padded (T, B) batch of hypotheses through the whole
 * Transformer stack): q / k / v are (R, ld_qkv) and out (R, nhead * head_dim) matrices of the R real tokens, rowmap (T * B) int32
 * gives the row of the token at padded position t * B + b or -1 for padding (the real tokens of a column are a prefix of it).
 * Same arithmetic as blm_attn_fwd without dropout; nothing is scattered to or gathered from a padded buffer.  head_dim 64 and
 * T <= 32 (one wave per head); anything else returns BLM_ERR_UNSUPPORTED and the caller scatters / gathers around blm_attn_fwd. */
int blm_attn_fwd_rows(const float* q, const float* k, const float* v, int64_t ld_qkv, float* out, const int32_t* rowmap, int T,
                      int B, int nhead, int head_dim, void* stream);
/* The same backward with a caller-owned scratch buffer of blm_attn_bwd_ws_floats() floats (0: this shape has no use
 * for one): the dK/dV kernel leaves dS (B*nhead, T, T) there and dQ = dS K is one small product instead of a second
 * recomputation of the probabilities and their dropout masks.  ws == NULL is blm_attn_bwd.  Same results. */
int64_t blm_attn_bwd_ws_floats(int T, int B, int nhead, int head_dim);
int blm_attn_bwd_ws(const float* q, const float* k, const float* v, int64_t ld_qkv, const float* out, const float* dout,
                    const float* lse, float* dq, float* dk, float* dv, int64_t ld_dqkv, int T, int B, int nhead,
                    int head_dim, float p, const blm_rng* rng, int col_offset, int global_cols, float* ws,
                    int64_t ws_floats, void* stream);
/* The same attention with the dropout factors of the probabilities HANDED OVER instead of generated from the Philox stream: keep is
 * (global_cols * nhead, T, T) floats, 0 or 1 / (1 - p), indexed by the head ((col_offset + b) * nhead + head) as the reference's
 * (B * h, T, T) probability tensor is (model.py:905-914: softmax -> nn.Dropout -> bmm).  A parity path -- the caller passes the mask
 * torch's CPU dropout drew (NoiseState.source "torch"); always the vector-ALU kernels, any head size / length. */
int blm_attn_fwd_keep(const float* q, const float* k, const float* v, int64_t ld_qkv, float* out, float* lse, int T, int B, int nhead,
                      int head_dim, const float* keep, int col_offset, int global_cols, void* stream);
int blm_attn_bwd_keep(const float* q, const float* k, const float* v, int64_t ld_qkv, const float* out, const float* dout,
                      const float* lse, float* dq, float* dk, float* dv, int64_t ld_dqkv, int T, int B, int nhead, int head_dim,
                      const float* keep, int col_offset, int global_cols, void* stream);

/* Cross entropy over materialised logits (M, V) (train.py:233,332;
 * compute_sentence_scores_bayes_jianwei.py:168):
 *   nll[m] = logsumexp(logits[m,:]) - logits[m,tgt[m]]
 * lse (optional) receives logsumexp per row.  If dlogits != NULL also writes (in place allowed) the gradient of
 * mean-NLL: (softmax - onehot) * grad_scale.  loss_sum (+=) sum of nll. */
int blm_ce_fwd_bwd(const float* logits, int64_t ld, const int64_t* tgt, float* nll, float* lse, float* loss_sum,
                   float* dlogits, float grad_scale, int M, int V, void* stream);
/* Deferred gradient: dlogits = (exp(logits - lse[m]) - onehot) * g_dev[0] * scale
 * (in place allowed), for callers that do not know the upstream gradient at
 * forward time. */
int blm_ce_bwd(const float* logits, int64_t ld, const int64_t* tgt, const float* lse, const float* g_dev, float scale,
               float* dlogits, int M, int V, void* stream);

/* Scoring with two interpolated models (compute_sentence_scores_bayes_jianwei.py:157-168: the
 * LOGITS are mixed, alpha*a + (1-alpha)*b, before the log-softmax): per-row NLL in one pass over
 * both logit matrices (M, V), the mixture is never stored.  Forward only. */
int blm_ce_interp_fwd(const float* logits_a, const float* logits_b, int64_t ld, float alpha, const int64_t* tgt,
                      float* nll, int M, int V, void* stream);


/* dcoef[i,n] += sum_m g[m,n] * act_i(z[m,n]), i = tanh, sigmoid, relu, gelu: gradient of the GPNN
 * mixture coefficients (autograd of model.py:1885-1899). */
int blm_gp_coef_grad(const float* g, const float* z, float* dcoef, int M, int N, void* stream);

/* out[n] (+)= sum_m x[m,n]   (bias gradients). */
int blm_colsum(const float* x, int64_t ld, float* out, int M, int N, int accumulate, void* stream);
/* ... and the same sums into a second vector out2 (NULL: none) from the same pass: nn.LSTM's b_ih and b_hh receive the same
 * gradient (model.py:35 / _VF.lstm :812).  For callers that bind the ABI directly: the Python host of this repository takes the
 * bias gradients out of the weight-gradient GEMM (blm_gemm_args.colsum_a) and adds them to both leaves with blm_init_multi, so
 * nothing under bayeslms_amd/ calls this entry point (it is covered at kernel level, tests/test_gpu_kernels.py).  In
 * deterministic mode (blm_set_option("deterministic", 1)) both column-sum entry points use one row chunk: one writer per sum. */
int blm_colsum2(const float* x, int64_t ld, float* out, float* out2, int M, int N, int accumulate, void* stream);
/* Up to 8 small vectors set by ONE launch: dst[i][0..n[i]) = (src[i] ? src[i][j] : 0) + (src2[i] ? src2[i][j] : 0); src / src2
 * may be NULL altogether.  dst, src, src2, n are HOST arrays (copied into the launch).  The set-up of a recurrent layer -- initial
 * states into row 0 of the state histories, b_ih + b_hh, zeroed gradient carries -- is 8-12 launches of ~5 us without it. */
int blm_init_multi(int count, float* const* dst, const float* const* src, const float* const* src2, const int64_t* n,
                   void* stream);

/* Global-norm clip + SGD momentum over a list of tensors (train.py:419-420,466):
 *   norm = sqrt(sum_i |g_i|^2); c = min(1, clip/(norm+1e-6));
 *   buf = first ? c*g : mom*buf + c*g;  p -= lr*buf
 * Two calls: blm_sqnorm_multi adds the squared norm to *sq (zeroed by the
 * caller) through per-block partials in ws (>= blm_sqnorm_ws_floats(n)) summed
 * in a fixed order -- deterministic, so data-parallel replicas stay bit-identical;
 * blm_clip_sgd_multi consumes it.  Pointer tables are DEVICE arrays of n
 * pointers/sizes. */
int64_t blm_sqnorm_ws_floats(int n);
int blm_sqnorm_multi(const float* const* grads, const int64_t* sizes, int n, float* sq, float* ws, void* stream);
int blm_clip_sgd_multi(float* const* params, const float* const* grads, float* const* bufs, const int64_t* sizes,
                       int n, const float* sq, float clip, float lr, float momentum, int first, float grad_scale,
                       void* stream);

/* One whole LSTM time step in a single launch (recurrent product on the matrix
 * cores + the cell below fused behind it; model.py:812 `_VF.lstm` per step):
 *   gates = xw_t[b,4H] + h_prev[b,H] . w_hh[4H,H]^T ; then the cell update;
 *   h_noise (H floats, may be NULL) is added to every row of h afterwards (the Variational
 *   LSTM's h += eps*exp(hidden_lgstd), model.py:2523-2527).
 * Requires H % 32 == 0 and 16-byte aligned h_prev / w_hh; otherwise returns
 * BLM_ERR_UNSUPPORTED and the caller composes blm_gemm + blm_lstm_cell_fwd.
 * Deterministic (fixed summation order, no atomics). */
int blm_lstm_step_fwd(const float* xw_t, const float* w_hh, const float* h_prev, const float* c_prev, float* h,
                      float* c, float* gates_act, const float* h_noise, int B, int H, void* stream);

/* The T forward steps of one layer from ONE call (T launches of blm_lstm_step_fwd issued by the library: a scoring
 * pass has nothing to hide the caller's per-call cost behind).  xw (T,B,4H); hs, cs (T+1,B,H) with row 0 = the
 * initial state, rows 1..T written; gates_act (T,B,4H) or NULL; noise_rows (T,H) or NULL.  Same shape rule and
 * status codes as blm_lstm_step_fwd (nothing is launched when the first step is unsupported). */
int blm_lstm_seq_fwd(const float* xw, const float* w_hh, float* hs, float* cs, float* gates_act, const float* noise_rows,
                     int T, int B, int H, void* stream);
/* Two independent recurrences side by side: step t of sequence A (n_a steps) and step t of sequence B (n_b steps) per launch for
 * B <= 4 (the scorer's B = 1 carry chain: layer 1 over chunk c beside layer 2 over chunk c - 1 -- one launch per step pair
 * instead of two; larger batches and the unpaired remainder run as blm_lstm_step_fwd launches).  Buffers as blm_lstm_seq_fwd,
 * per sequence; n_a or n_b may be 0. */
int blm_lstm_seq_pair_fwd(const float* xw_a, const float* w_hh_a, float* hs_a, float* cs_a, float* ga_a, int n_a,
                          const float* xw_b, const float* w_hh_b, float* hs_b, float* cs_b, float* ga_b, int n_b,
                          int B, int H, void* stream);

/* Backward of one LSTM time step in a single launch:
 *   dh = dgates_t[b,4H] . w_hh[4H,H]   (w_hh passed TRANSPOSED: w_hh_t (H,4H), see blm_transpose)
 * then, when dgates_out != NULL, the cell backward of the previous step (blm_lstm_cell_bwd2 with
 * dh = this product, dh2 = dy_prev): dgates_out (B,4H) and dc_prev (B,H) are written and dh itself
 * never reaches memory unless dh_out != NULL.  dgates_out == NULL: only dh_out is written.
 * Same shape/alignment rule and status codes as blm_lstm_step_fwd. */
int blm_lstm_step_bwd(const float* dgates_t, const float* w_hh_t, const float* dy_prev, const float* dc_next,
                      const float* c_prev, const float* c, const float* gates_act, float* dgates_out, float* dc_prev,
                      float* dh_out, int B, int H, void* stream);
/* The backward steps t_hi-1 .. t_lo (descending) of one layer from ONE call -- blm_lstm_seq_fwd's counterpart: step T-1 is
 * the plain cell backward (blm_lstm_cell_bwd2 with dh = dh_T, dh2 = dy[T-1]), every earlier step t one blm_lstm_step_bwd
 * launch (dh = dgates[t+1] . w_hh, then the cell of step t).  dy (T,B,H); cs (T+1,B,H) as blm_lstm_seq_fwd leaves them;
 * gates_act, dgates (T,B,4H); dc_pair (2,B,H): slot k holds the dc arriving from above on entry, the slots alternate per
 * step, so the caller's next k is k ^ ((t_hi - t_lo) & 1); dh_rows (T,B,H) or NULL: row t receives the product of step t's
 * launch (what h_t gets from step t+1).  dh_T is read only when t_hi == T.  A chain may be walked in several calls
 * (t_hi of one = t_lo of the one before): the layer wavefront does, chunk by chunk. */
int blm_lstm_seq_bwd(const float* dh_T, const float* dy, const float* cs, const float* gates_act, const float* w_hh_t,
                     float* dgates, float* dc_pair, int k, float* dh_rows, int T, int t_hi, int t_lo, int B, int H,
                     void* stream);
/* The same two step kernels for the GP-LSTM cell with a GPNN on one gate (GPLSTMCell gate types 1-4,
 * model.py:1754-1771): gate `gate_ovr` (0 i, 1 f, 2 g, 3 o; -1 = plain LSTM) takes the activation
 * mixture sum_i act_i(z) coef4[i] of its pre-activation z instead of its sigmoid/tanh.  The caller
 * places the GPNN's weight rows into that gate's row block of w_hh / xw_t.  coef4 is (4,H) in the slot
 * order tanh, sigmoid, relu, gelu.  Forward keeps z (z_out, (B,H)) for the backward pass; backward
 * multiplies that gate's gradient by the mixture's derivative at z_prev and returns the gradient
 * w.r.t. the mixture value in dact_out (B,H) (for blm_gp_coef_grad).
 * gate_ovr = 4 is GPLSTMCell gate type 6 (model.py:1744-1752): the whole hidden projection
 * h_prev . w_hh^T + rbias (4H) passes through the mixture (coef4 (4,4H), z (B,4H)) before it is added
 * to xw_t; backward then also writes dz_out = dgates_out * mixture'(z_prev) (B,4H), which is the
 * dgates_t operand of the next (earlier) step's launch.
 * gate_ovr = 5 is GPLSTMCell gate type 5 (model.py:1759-1760, `cx = self.gpnn(cx)` in front of the cell update): a second
 * recurrent product per step, c_{t-1} . Wg^T.  The caller computes it with blm_lstm_step_dh(c_prev, Wg, z_out, B, H, H)
 * right before this launch; the step kernel adds rbias (H), stores z back into z_out and lets the mixture of z
 * (coef4 (4,H)) take c_prev's place in  c' = f * c_in + i * g.  Backward (gate_ovr = 5, z_prev = that z): the cell
 * backward uses mixture(z_prev) as its incoming cell state, writes dact_out = d c_in (B,H) (for blm_gp_coef_grad) and
 * dz_out = d c_in * mixture'(z_prev) (B,H); the raw cell-state gradient of the earlier step is then
 * blm_lstm_step_dh(dz_out, Wg^T, dc, B, H, H). */
int blm_lstm_step_fwd_gp(const float* xw_t, const float* w_hh, const float* h_prev, const float* c_prev, float* h,
                         float* c, float* gates_act, const float* h_noise, int gate_ovr, const float* coef4,
                         const float* rbias, float* z_out, int B, int H, void* stream);
int blm_lstm_step_bwd_gp(const float* dgates_t, const float* w_hh_t, const float* dy_prev, const float* dc_next,
                         const float* c_prev, const float* c, const float* gates_act, float* dgates_out, float* dc_prev,
                         float* dh_out, int gate_ovr, const float* coef4, const float* z_prev, float* dact_out,
                         float* dz_out, int B, int H, void* stream);
/* out (cols,rows) = in (rows,cols)^T */
int blm_transpose(const float* in, float* out, int rows, int cols, void* stream);

/* LSTM cell pointwise part (what _VF.lstm fuses, model.py:812): gates =
 * xw[b,4H] + hw[b,4H] (biases already inside xw), order i,f,g,o.
 *   c' = sig(f) c + sig(i) tanh(g);  h' = sig(o) tanh(c')
 * Saves activated gates (B,4H) for backward. */
int blm_lstm_cell_fwd(const float* xw, const float* hw, const float* c_prev, float* h, float* c, float* gates_act,
                      int B, int H, void* stream);
/* In: dh (sum of grad from above and from next step), dc_next; out: dgates (B,4H), dc_prev. */
int blm_lstm_cell_bwd(const float* dh, const float* dc_next, const float* c_prev, const float* c, const float* gates_act,
                      float* dgates, float* dc_prev, int B, int H, void* stream);
/* Same with the incoming hidden gradient given as two addends (recurrent part + this step's dy),
 * so the host needs no separate accumulation pass per time step. */
int blm_lstm_cell_bwd2(const float* dh, const float* dh2, const float* dc_next, const float* c_prev, const float* c,
                       const float* gates_act, float* dgates, float* dc_prev, int B, int H, void* stream);
/* GP-LSTM cell (GPLSTMCell.Gplstm, model.py:1745-1777): same cell, but gate `gate_idx` (0 i, 1 f,
 * 2 g, 3 o) takes its ACTIVATED value from gate_ovr (B,H) -- the output of a GPNN -- instead of
 * sigmoid/tanh of its pre-activation.  Backward returns, in slot gate_idx of dgates, the gradient
 * w.r.t. that activated value (d_ovr, (B,H)) and zero pre-activation gradient for it. */
int blm_lstm_cell_ovr_fwd(const float* xw, const float* hw, const float* c_prev, const float* gate_ovr, int gate_idx,
                          float* h, float* c, float* gates_act, int B, int H, void* stream);
int blm_lstm_cell_ovr_bwd(const float* dh, const float* dc_next, const float* c_prev, const float* c,
                          const float* gates_act, int gate_idx, float* dgates, float* d_ovr, float* dc_prev, int B,
                          int H, void* stream);
/* Same with the incoming hidden gradient as two addends (dh2 may be NULL), like blm_lstm_cell_bwd2. */
int blm_lstm_cell_ovr_bwd2(const float* dh, const float* dh2, const float* dc_next, const float* c_prev, const float* c,
                           const float* gates_act, int gate_idx, float* dgates, float* d_ovr, float* dc_prev, int B, int H,
                           void* stream);
/* Elementwise GPNN mixture on pre-activations z (M,N): out = sum_i act_i(z) coef[i,n];
 * backward dz = dout * sum_i act_i'(z) coef[i,n]   (model.py:1885-1899). */
int blm_gp_mix_fwd(const float* z, const float* coef, float* out, int M, int N, void* stream);
int blm_gp_mix_bwd(const float* dout, const float* z, const float* coef, float* dz, int M, int N, void* stream);
/* x[b,:] += v[:]  (VNN hidden-state noise, model.py:2571-2577); rows B, cols H. */
int blm_add_rowvec(float* x, const float* v, int B, int H, void* stream);

/* y (+)= a*x elementwise helpers used by the host glue. */
int blm_axpy(const float* x, float* y, int64_t n, float a, void* stream);
/* dst[v,:] += src[slot[v],:] for every v in [0,V) with 0 <= slot[v] < n_src (rows of D floats).  Data-parallel
 * training reduces the embedding half of the tied encoder/decoder gradient (model.py:1240; the last tensor backward
 * finishes) as a compact matrix of the rows this step touched; this adds it back (engine.LateRows). */
int blm_rows_gather_add(float* dst, const int64_t* slot, const float* src, int64_t V, int D, int64_t n_src, void* stream);

/* --------------------------------------------------------------------------
 * Architecture search (model_search_bayes.py, architect.py,
 * train_search_bayes.py).  HBM-bound streaming kernels.
 * ------------------------------------------------------------------------ */

/* Mix of two candidate branches by the softmax'd architecture weights
 * (model_search_bayes.py:234-236, :77-78):
 *   out = (probs[0]*a + probs[1]*b) * keep        probs: 2 floats on the DEVICE
 * a, b, out are (rows, B, N) activations; dropout (model_search_bayes.py:236
 * `self.dropout(src1)`) as blm_dropout keys it; drop_p == 0 -> none. */
int blm_mix2_fwd(const float* a, const float* b, const float* probs, float* out, int rows, int B, int N, float drop_p,
                 const blm_rng* rng, int col_offset, int global_cols, void* stream);
/* Backward: g = dout*keep; da = probs[0]*g*(mul_a ? mul_a : 1); db = probs[1]*g (either may be NULL);
 * partial[2*j+{0,1}] = block j's share of sum(g*a), sum(g*b) -- blm_mix2_partials(rows,B,N) floats,
 * written (not accumulated), fixed order inside a block; reduce with blm_colsum(partial, 2, dprobs, n/2, 2).
 * mul_a lets the GELU branch fold its derivative (aux of BLM_EPI_BIAS_GELU) into the same pass. */
int64_t blm_mix2_partials(int rows, int B, int N);
int blm_mix2_bwd(const float* dout, const float* a, const float* b, const float* probs, const float* mul_a, float* da,
                 float* db, float* partial, int rows, int B, int N, float drop_p, const blm_rng* rng, int col_offset,
                 int global_cols, void* stream);

/* blm_mix2_bwd for a GP branch b = sum_i act_i(z_b) coef[i] (GaussTransSearchEncoderLayer, model_search_bayes.py:
 * 234-236): dz_b = probs[1]*g*mixture'(z_b) directly (no separate blm_gp_mix_bwd pass over the activations);
 * dhk (may be NULL) = probs[1]*g, the gradient w.r.t. the mixture value that blm_gp_coef_grad needs. */
int blm_mix2_gp_bwd(const float* dout, const float* a, const float* b, const float* probs, const float* mul_a,
                    const float* z_b, const float* coef, float* da, float* dz_b, float* dhk, float* partial, int rows, int B,
                    int N, float drop_p, const blm_rng* rng, int col_offset, int global_cols, void* stream);

/* BayesLSTMSearchCell.bayeslstm pointwise part (model_search_bayes.py:686-710).  z8 = xw8 + hw8 is
 * (B, 8H) with row layout [i f g o | i' f' g' o'] (standard gates, then the four `Bayes` maps);
 * probs (4,2) on the device, rows i,f,g,o:
 *   gate_k = act_k(z_k)*probs[k][0] + act_k(z'_k)*probs[k][1];  c = f*c_prev + i*g;  h = o*tanh(c)
 * acts8 (B,8H, may be NULL) keeps the eight activations for the backward. */
int blm_lstm_search_cell_fwd(const float* xw8, const float* hw8, const float* c_prev, const float* probs, float* h,
                             float* c, float* acts8, int B, int H, void* stream);
/* Backward: dh (+ dh2, may be NULL), dc_next (may be NULL) -> dz8 (B,8H), dc_prev (B,H) and
 * partial[8*j + 2k + s] = block j's share of d probs[k][s]; blm_lstm_search_cell_partials(B,H) floats. */
int64_t blm_lstm_search_cell_partials(int B, int H);
int blm_lstm_search_cell_bwd(const float* dh, const float* dh2, const float* dc_next, const float* c_prev, const float* c,
                             const float* acts8, const float* probs, float* dz8, float* dc_prev, float* partial, int B,
                             int H, void* stream);

/* One whole search-cell time step in a single launch (blm_lstm_step_fwd with eight gate streams):
 *   z8 = xw8_t[b,8H] + h_prev[b,H] . w8_hh[8H,H]^T, then blm_lstm_search_cell_fwd's pointwise part.
 * Same shape rule and status codes as blm_lstm_step_fwd (BLM_ERR_UNSUPPORTED: compose blm_gemm +
 * blm_lstm_search_cell_fwd). */
int blm_lstm_search_step_fwd(const float* xw8_t, const float* w8_hh, const float* h_prev, const float* c_prev,
                             const float* probs, float* h, float* c, float* acts8, int B, int H, void* stream);

/* Backward of one search-cell time step in a single launch (blm_lstm_step_bwd with the search cell fused behind the
 * product): dh = dz8_t (B,8H) . w8_t (H,8H)^T, then blm_lstm_search_cell_bwd of the previous step with dh + dy_prev:
 * dz8_out (B,8H), dc_prev (B,H) and blm_lstm_search_step_partials(B,H) per-block partials of d probs
 * (partial[8*j + 2k + s]); dh never reaches memory.  H % 16 == 0, 16-byte aligned operands. */
int64_t blm_lstm_search_step_partials(int B, int H);
int blm_lstm_search_step_bwd(const float* dz8_t, const float* w8_t, const float* dy_prev, const float* dc_next,
                             const float* c_prev, const float* c, const float* acts8, const float* probs, float* dz8_out,
                             float* dc_prev, float* partial, int B, int H, void* stream);

/* The skinny recurrent dgrad of one search-cell step on the blm_lstm_step_bwd kernel (no cell fused):
 *   dh_out (B,H) = dz (B,G) . w_t (H,G)^T      w_t = the stacked recurrent weight (G,H) TRANSPOSED
 * One launch, fixed summation order, no memset / atomics.  Needs H % 16 == 0, G % 64 == 0, 16-byte
 * aligned operands (BLM_ERR_UNSUPPORTED otherwise: use blm_gemm). */
int blm_lstm_step_dh(const float* dz, const float* w_t, float* dh_out, int B, int H, int G, void* stream);
/* The same product written into a column window of a wider matrix: dh_out has row stride ldo >= H. */
int blm_lstm_step_dh_ld(const float* dz, const float* w_t, float* dh_out, int64_t ldo, int B, int H, int G, void* stream);
/* ... with the GPNN2 activation sum on the way out (feat has row stride ld_f; the product has H output columns):
 *   act_mode 1: the product is the feature matrix f:  feat = f  and  out[:, m] = (f + sum_a a(f)) * scale for m < M,
 *               out[:, M] = 1, 0 for M < m < H                       (blm_gpnn2_actsum_fwd fused behind the product)
 *   act_mode 2: the product is d s:  out[:, m] = d s * (1 + sum_a a'(feat)) * scale for m < M, 0 beyond   (..._bwd) */
int blm_lstm_step_dh_act(const float* dz, const float* w_t, float* out, int64_t ldo, int B, int H, int G, int act_mode, float* feat,
                         int ld_f, int M, float scale, int acts, void* stream);

/* GPNN2 (random-feature GP, model.py:2036-2076) between its two products: features f (rows, ld_f) -> s (rows, ld_s),
 *   s[:, m] = (f + sum_{a in acts} a(f)) * scale   for m < M   (acts: bit set in the slot order 1 tanh, 2 sigmoid, 4 relu,
 *   s[:, M] = 1, zeros beyond                                   8 gelu; model.py:2069-2075 with skip_act)
 * The 1-column lets the caller fold coef.bias into a coefficient matrix padded to ld_s columns.  Backward:
 *   df[:, m] = ds[:, m] * (1 + sum a'(f)) * scale for m < M, 0 on the padding; ds and df have row stride ld_s. */
int blm_gpnn2_actsum_fwd(const float* f, float* s, int64_t rows, int M, int ld_f, int ld_s, float scale, int acts, void* stream);
int blm_gpnn2_actsum_bwd(const float* ds, const float* f, float* df, int64_t rows, int M, int ld_f, int ld_s, float scale,
                         int acts, void* stream);
/* out[r, c] = a[r, c] + b[r, c] for r < rows, c < cols on column windows of wider matrices (row strides lda / ldb / ldo):
 * the pre-activation of ONE gate, xw_t[:, gH:(g+1)H] + (h W_hh^T)[:, gH:(g+1)H], as a contiguous operand. */
int blm_add_cols(const float* a, int64_t lda, const float* b, int64_t ldb, float* out, int64_t ldo, int64_t rows, int cols,
                 void* stream);
/* The T frequency matrices of one forward of a GPNN2 that is called at every time step (GPLSTMCell with type digit 4,
 * model.py:1763-1770 -> :2062-2065): F_t = mean + exp(lgstd) * eps_t (H x M), eps_t = eps_all[t] (T,H,M) or the Philox
 * stream (rng0->seed, rng0->stream, rng0->step + t) with blm_sample_weight's element order.  Outputs, zero padded:
 * FT (T, MP, H) = F_t^T and Fp (T, H, GP) = F_t -- the operands of the per-step feature product and of its input gradient. */
int blm_gpnn2_sample_steps(const float* mean, const float* lgstd, const float* eps_all, const blm_rng* rng0, int T, int H, int M,
                           int MP, int GP, float* FT, float* Fp, void* stream);
/* Their gradient in one launch: with G_t = pre_t^T df_t (pre (T,B,H) the GPNN2 inputs, df (T,B,GP) the feature gradients),
 *   dmean += sum_t G_t,   dlgstd += exp(lgstd) * sum_t eps_t * G_t      (either may be NULL). */
int blm_gpnn2_freq_grad(const float* pre, const float* df, const float* eps_all, const blm_rng* rng0, const float* lgstd,
                        float* dmean, float* dlgstd, int T, int B, int H, int M, int GP, void* stream);
/* The time loops of a GP-LSTM layer with a GPNN2 that draws fresh frequencies at every step (GPLSTMCell, type digit 4;
 * model.py:1744-1771), ONE call per direction.  GPNN2_t(x) = (actsum(x F_t) | 1) cwp^T with cwp = [coef.weight | coef.bias | 0]
 * (rows = outputs, GP columns), FT (nF,MP,H) / Fp (nF,H,GP) from blm_gpnn2_sample_steps, nF = T or 1 (mean frequencies).
 *   mode 0  gate types 1-4: z4_t = h_{t-1} w_hh^T; pre_t = xw_t[:, gate] + z4_t[:, gate]; gout_t = GPNN2_t(pre_t) (B,H) is
 *           that gate's activation in the cell update (xw carries both bias_ih)
 *   mode 1  gate type 5:    gout_t = c_in = GPNN2_t(c_{t-1}) (B,H) takes c_{t-1}'s place in the fused step (blm_lstm_step_fwd)
 *   mode 2  gate type 6:    gout_t = GPNN2_t(h_{t-1}) (B,4H; cwp is (4H,GP)) is the hidden projection of all four gates
 * Buffers (T rows each unless noted): hs, cs (T+1,B,H) with row 0 = initial state; ga (B,4H); feat (B,MP); sact (B,GP);
 * mode 0 also z4 (B,4H) and pre (B,H).  Backward walks t = T-1..0: dh (B,H) holds dh_T on entry and dh_0 on exit, dcs2
 * (2,B,H) ping-pongs dc ([0] = dc_T on entry, dc_0 ends in [T & 1]); it fills dgates (T,B,4H) (= d xw; mode 0: the gate's
 * slot holds d pre), df (T,B,GP), da (T,B,H) (modes 0 / 1: the gradient of gout); ds is unused (the activation sum and its
 * derivative ride in the epilogues of the feature / d s products, blm_lstm_step_dh_act; sact's padding columns >= MP must be
 * zero on entry of the forward call); w_hh_t = w_hh^T (H,4H), cwt = cwp^T.  Needs H % 64 == 0, MP % 16 == 0, GP % 64 == 0, M < MP <= GP (BLM_ERR_UNSUPPORTED from the
 * products otherwise). */
typedef struct blm_gpnn2_seq {
  uint32_t abi_version; /* BLM_ABI_VERSION */
  int32_t mode, gate, acts;
  int32_t T, B, H, M, MP, GP, nF;
  const float *xw, *w_hh, *w_hh_t, *FT, *Fp, *cwp, *cwt;
  float *hs, *cs, *ga, *z4, *pre, *feat, *sact, *gout;
  const float* dy;
  float *dh, *dcs2, *dgates, *da, *ds, *df;
} blm_gpnn2_seq;
int blm_lstm_gpnn2_seq_fwd(const blm_gpnn2_seq* q, void* stream);
int blm_lstm_gpnn2_seq_bwd(const blm_gpnn2_seq* q, void* stream);

/* torch.optim.Adam(lr, betas, eps, weight_decay) on one tensor (architect.py:33): g += wd*p;
 * m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr * (m/(1-b1^step)) / (sqrt(v/(1-b2^step)) + eps).
 * step counts from 1. */
int blm_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int step, void* stream);

/* blm_clip_sgd_multi with torch.optim.SGD's weight decay: the clipped gradient becomes
 * c*g + weight_decay*p before the momentum update (train_search_bayes.py:330,391-392). */
int blm_clip_sgd_multi_wd(float* const* params, const float* const* grads, float* const* bufs, const int64_t* sizes, int n,
                          const float* sq, float clip, float lr, float momentum, int first, float grad_scale,
                          float weight_decay, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BAYESLM_H */
