/*
 * bayes_linear_step.c -- a host with NO Python and NO torch above the C ABI (include/bayeslm.h): plain C, the HIP
 * runtime for device memory and a stream, libbayeslm_hip.so for the device work.
 *
 * One training step of the reference's BayesLinear (steps/pytorchnn/model.py:1083-1129: sample W = mu + exp(lgstd) eps,
 * y = x W^T, KL = mean(mu^2 - 2 lgstd + exp(2 lgstd)) / 2) and of its autograd (dx = dy W; dmu = dy^T x + lambda dKL/dmu;
 * dlgstd = (dy^T x) eps exp(lgstd) + lambda dKL/dlgstd), through exactly the entry points the Python host uses:
 *   blm_sample_weight (Philox noise, KL on the way)  ->  blm_gemm NT  ->  blm_gemm NN  ->  blm_gemm TN + BLM_EPI_BAYES_WGRAD
 * (eps regenerated from the counter in the epilogue, KL gradient folded in: nothing but W itself is ever stored).
 * The result is checked against a double-precision loop in this file that reads the SAME noise through
 * blm_philox_normal.  Exit code 0 = every tensor within 1e-4 of the loop (north_star's bar is 1e-3).
 *
 * Build (tests/test_cabi_cpu.py compiles it, tests/test_gpu_kernels.py runs it on the GPU):
 *   gcc -O1 -std=c11 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_host/bayes_linear_step.c \
 *       -Lbayeslms_amd -lbayeslm_hip -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/bayeslms_amd -Wl,-rpath,/opt/rocm/lib
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "bayeslm.h"

#define HIPCHECK(e)                                                                        \
  do {                                                                                     \
    hipError_t err_ = (e);                                                                 \
    if (err_ != hipSuccess) {                                                              \
      fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(err_));         \
      return 2;                                                                            \
    }                                                                                      \
  } while (0)

#define BLMCHECK(e)                                                                        \
  do {                                                                                     \
    int st_ = (e);                                                                         \
    if (st_ != BLM_OK) {                                                                   \
      fprintf(stderr, "%s:%d: status %d: %s\n", __FILE__, __LINE__, st_, blm_last_error()); \
      return 3;                                                                            \
    }                                                                                      \
  } while (0)

static unsigned long long lcg_state = 0x9E3779B97F4A7C15ull;
static float uniform_pm1(void) { /* deterministic host data, [-1, 1) */
  lcg_state = lcg_state * 6364136223846793005ull + 1442695040888963407ull;
  return (float)((double)(lcg_state >> 11) / 9007199254740992.0 * 2.0 - 1.0);
}

static float* host_fill(size_t n, float scale, float shift) {
  float* p = (float*)malloc(n * sizeof(float));
  for (size_t i = 0; i < n; ++i) p[i] = shift + scale * uniform_pm1();
  return p;
}

static int to_device(float** d, const float* h, size_t n) {
  HIPCHECK(hipMalloc((void**)d, n * sizeof(float)));
  if (h != NULL) HIPCHECK(hipMemcpy(*d, h, n * sizeof(float), hipMemcpyHostToDevice));
  else HIPCHECK(hipMemset(*d, 0, n * sizeof(float)));
  return 0;
}

static double worst(const char* name, const float* got, const double* want, size_t n) {
  double scale = 1e-30, err = 0.0;
  for (size_t i = 0; i < n; ++i) {
    if (fabs(want[i]) > scale) scale = fabs(want[i]);
    if (fabs((double)got[i] - want[i]) > err) err = fabs((double)got[i] - want[i]);
  }
  printf("  %-7s max |diff| / max |ref| = %.3g\n", name, err / scale);
  return err / scale;
}

int main(int argc, char** argv) {
  /* tokens x in -> out; any sizes are legal for the library (odd ones take its tail paths): try `./a.out 250 100 75` */
  const int M = argc > 1 ? atoi(argv[1]) : 256, K = argc > 2 ? atoi(argv[2]) : 128, N = argc > 3 ? atoi(argv[3]) : 192;
  const float lambda = 0.25f; /* the weight train.py:334-399 gives the KL term, any value here */
  if (M <= 0 || K <= 0 || N <= 0) return 1;

  char arch[32] = {0};
  int n_cu = 0, lds = 0;
  if (blm_abi_version() != BLM_ABI_VERSION) {
    fprintf(stderr, "header / library ABI mismatch\n");
    return 1;
  }
  BLMCHECK(blm_query(0, arch, &n_cu, &lds));
  printf("device 0: %s, %d CUs, %d bytes of LDS per CU\n", arch, n_cu, lds);
  if (strncmp(arch, "gfx950", 6) != 0) {
    fprintf(stderr, "this library is written for gfx950 only\n");
    return 1;
  }
  HIPCHECK(hipSetDevice(0));
  hipStream_t st;
  HIPCHECK(hipStreamCreate(&st));

  const size_t nW = (size_t)N * K, nX = (size_t)M * K, nY = (size_t)M * N;
  float* h_mu = host_fill(nW, 0.1f, 0.f);
  float* h_lg = host_fill(nW, 0.5f, -3.f); /* log sigma around -3 as the reference initialises it */
  float* h_x = host_fill(nX, 1.f, 0.f);
  float* h_dy = host_fill(nY, 0.05f, 0.f);
  float *mu, *lg, *x, *dy, *W, *y, *dx, *dmu, *dlg, *kl, *eps;
  if (to_device(&mu, h_mu, nW) || to_device(&lg, h_lg, nW) || to_device(&x, h_x, nX) || to_device(&dy, h_dy, nY) ||
      to_device(&W, NULL, nW) || to_device(&y, NULL, nY) || to_device(&dx, NULL, nX) || to_device(&dmu, NULL, nW) ||
      to_device(&dlg, NULL, nW) || to_device(&kl, NULL, 1) || to_device(&eps, NULL, nW))
    return 2;

  /* the noise of this step: key = seed, counter = (element block, weight stream | tensor id, step) */
  blm_variational v;
  memset(&v, 0, sizeof v);
  v.lgstd = lg;
  v.eps = NULL; /* Philox, not injected */
  v.row_lo = 0;
  v.srows = N;
  v.rng.seed = 1111;
  v.rng.stream = BLM_STREAM_WEIGHT | 7u;
  v.rng.step = 42;

  /* forward: W and KL in one pass, then y = x W^T */
  BLMCHECK(blm_sample_weight(mu, N, K, &v, W, kl, 1.0f, st));
  blm_gemm_args g;
  memset(&g, 0, sizeof g);
  g.abi_version = BLM_ABI_VERSION;
  g.alpha = 1.0f;
  g.op = BLM_GEMM_NT;
  g.M = M, g.N = N, g.K = K;
  g.A = x, g.lda = K;
  g.B = W, g.ldb = K;
  g.C = y, g.ldc = N;
  BLMCHECK(blm_gemm(&g, st));
  /* backward: dx = dy W */
  g.op = BLM_GEMM_NN;
  g.M = M, g.N = K, g.K = N;
  g.A = dy, g.lda = N;
  g.B = W, g.ldb = K;
  g.C = dx, g.ldc = K;
  BLMCHECK(blm_gemm(&g, st));
  /* backward: dW = dy^T x, turned into dmu / dlgstd (+ lambda dKL) by the epilogue, eps regenerated */
  g.op = BLM_GEMM_TN;
  g.M = N, g.N = K, g.K = M;
  g.A = dy, g.lda = N;
  g.B = x, g.ldb = K;
  g.C = dmu, g.ldc = K;
  g.epilogue = BLM_EPI_BAYES_WGRAD;
  g.C2 = dlg;
  g.wg_mu = mu;
  g.var_c = v;
  g.kl_lambda = lambda;
  g.kl_inv_n = 1.0f / (float)nW;
  g.flags = BLM_GEMM_ACCUMULATE; /* gradients accumulate, as into .grad */
  BLMCHECK(blm_gemm(&g, st));
  /* the same noise, made visible for the check below */
  BLMCHECK(blm_philox_normal(eps, (int64_t)nW, &v.rng, st));
  HIPCHECK(hipStreamSynchronize(st));

  float* o_y = (float*)malloc(nY * 4);
  float* o_dx = (float*)malloc(nX * 4);
  float* o_dmu = (float*)malloc(nW * 4);
  float* o_dlg = (float*)malloc(nW * 4);
  float* o_eps = (float*)malloc(nW * 4);
  float o_kl = 0.f;
  HIPCHECK(hipMemcpy(o_y, y, nY * 4, hipMemcpyDeviceToHost));
  HIPCHECK(hipMemcpy(o_dx, dx, nX * 4, hipMemcpyDeviceToHost));
  HIPCHECK(hipMemcpy(o_dmu, dmu, nW * 4, hipMemcpyDeviceToHost));
  HIPCHECK(hipMemcpy(o_dlg, dlg, nW * 4, hipMemcpyDeviceToHost));
  HIPCHECK(hipMemcpy(o_eps, eps, nW * 4, hipMemcpyDeviceToHost));
  HIPCHECK(hipMemcpy(&o_kl, kl, 4, hipMemcpyDeviceToHost));

  /* the reference arithmetic as a double-precision loop */
  double* r_W = (double*)malloc(nW * 8);
  double* r_y = (double*)calloc(nY, 8);
  double* r_dx = (double*)calloc(nX, 8);
  double* r_dmu = (double*)calloc(nW, 8);
  double* r_dlg = (double*)calloc(nW, 8);
  double r_kl = 0.0, e_mean = 0.0, e_var = 0.0;
  for (size_t i = 0; i < nW; ++i) {
    r_W[i] = (double)h_mu[i] + exp((double)h_lg[i]) * (double)o_eps[i];
    r_kl += (double)h_mu[i] * h_mu[i] - 2.0 * h_lg[i] + exp(2.0 * h_lg[i]);
    e_mean += o_eps[i];
    e_var += (double)o_eps[i] * o_eps[i];
  }
  r_kl = r_kl / (double)nW / 2.0;
  e_mean /= (double)nW;
  e_var = e_var / (double)nW - e_mean * e_mean;
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      double acc = 0.0;
      for (int k = 0; k < K; ++k) acc += (double)h_x[(size_t)m * K + k] * r_W[(size_t)n * K + k];
      r_y[(size_t)m * N + n] = acc;
    }
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      const double d = h_dy[(size_t)m * N + n];
      for (int k = 0; k < K; ++k) {
        r_dx[(size_t)m * K + k] += d * r_W[(size_t)n * K + k];
        r_dmu[(size_t)n * K + k] += d * h_x[(size_t)m * K + k];
      }
    }
  for (size_t i = 0; i < nW; ++i) {
    const double dW = r_dmu[i];
    r_dlg[i] = dW * o_eps[i] * exp((double)h_lg[i]) + lambda * (exp(2.0 * h_lg[i]) - 1.0) / (double)nW;
    r_dmu[i] = dW + lambda * h_mu[i] / (double)nW;
  }

  printf("BayesLinear step, %d tokens, %d -> %d, through the C ABI against a double-precision loop:\n", M, K, N);
  printf("  noise   mean %.4f variance %.4f over %zu draws\n", e_mean, e_var, nW);
  double bad = 0.0, e;
  if ((e = worst("y", o_y, r_y, nY)) > bad) bad = e;
  if ((e = worst("dx", o_dx, r_dx, nX)) > bad) bad = e;
  if ((e = worst("dmu", o_dmu, r_dmu, nW)) > bad) bad = e;
  if ((e = worst("dlgstd", o_dlg, r_dlg, nW)) > bad) bad = e;
  const double ekl = fabs((double)o_kl - r_kl) / fabs(r_kl);
  printf("  %-7s %.6f against %.6f (relative %.3g)\n", "KL", o_kl, r_kl, ekl);
  if (ekl > bad) bad = ekl;
  const int noise_ok = fabs(e_mean) < 0.05 && fabs(e_var - 1.0) < 0.1;

  /* error behaviour: a bad call returns a status and a message, it never aborts the host */
  g.abi_version = 0;
  const int st_abi = blm_gemm(&g, st);
  printf("  a call with abi_version 0 returns %d: %s\n", st_abi, blm_last_error());

  HIPCHECK(hipStreamDestroy(st));
  hipFree(mu), hipFree(lg), hipFree(x), hipFree(dy), hipFree(W), hipFree(y), hipFree(dx), hipFree(dmu), hipFree(dlg),
      hipFree(kl), hipFree(eps);
  if (bad > 1e-4 || !noise_ok || st_abi != BLM_ERR_ABI) {
    printf("FAILED (worst relative difference %.3g)\n", bad);
    return 4;
  }
  printf("OK (worst relative difference %.3g)\n", bad);
  return 0;
}
