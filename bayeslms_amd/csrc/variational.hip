// Variational-weight kernels: one-pass weight materialisation W = mu + exp(lgstd)*eps with the
// Philox eps generated in registers (no eps tensor in HBM) and the KL partial sums fused into the
// same pass; standalone KL forward/backward; the raw N(0,1) stream for tests.  HBM-bound.
//
// Replaces model.py:1083-1107 (BayesLinear), :668-732 (Bayes2LSTM), :1243-1249 (EMB),
// KL: :1109-1125, :734-765, :1251-1256, :1816-1826.
#include "blm_device.h"
#include "blm_host.h"

namespace blm {

constexpr int TPB = 256;

// rows x cols, cols % 4 == 0, all pointers 16-B aligned.
__global__ __launch_bounds__(TPB) void sample_weight_vec4(const float* __restrict__ mu, long rows, long cols,
                                                          blm_variational v, float* __restrict__ w,
                                                          float* kl_out, float kl_scale) {
  __shared__ float red[TPB / 64];
  const long c4n = cols >> 2, total = rows * c4n;
  float klp = 0.f;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
    const long r = i / c4n, c = (i - r * c4n) << 2;
    float4 m = *reinterpret_cast<const float4*>(mu + r * cols + c);
    const long rel = r - v.row_lo;
    if (v.lgstd && (unsigned long)rel < (unsigned long)v.srows) {
      const long idx = rel * cols + c;
      const float4 lg = *reinterpret_cast<const float4*>(v.lgstd + idx);
      const float sx = __expf(lg.x), sy = __expf(lg.y), sz = __expf(lg.z), sw = __expf(lg.w);
      if (kl_out)
        klp += (m.x * m.x - 2.f * lg.x + sx * sx) + (m.y * m.y - 2.f * lg.y + sy * sy) +
               (m.z * m.z - 2.f * lg.z + sz * sz) + (m.w * m.w - 2.f * lg.w + sw * sw);
      if (w) {
        float4 z;
        if (v.eps) z = *reinterpret_cast<const float4*>(v.eps + idx);
        else z = philox_normal4(v.rng, (uint64_t)idx >> 2);
        m.x += sx * z.x; m.y += sy * z.y; m.z += sz * z.z; m.w += sw * z.w;
      }
    }
    if (w) *reinterpret_cast<float4*>(w + r * cols + c) = m;
  }
  if (kl_out) {
    const float t = block_sum<TPB / 64>(klp, red);
    if (threadIdx.x == 0 && t != 0.f) atomicAdd(kl_out, t * kl_scale);
  }
}

// Any shape / alignment (bias vectors, tiny test layers).
__global__ __launch_bounds__(TPB) void sample_weight_scalar(const float* __restrict__ mu, long rows, long cols,
                                                            blm_variational v, float* __restrict__ w,
                                                            float* kl_out, float kl_scale) {
  __shared__ float red[TPB / 64];
  const long total = rows * cols;
  float klp = 0.f;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
    const long r = i / cols;
    float m = mu[i];
    const long rel = r - v.row_lo;
    if (v.lgstd && (unsigned long)rel < (unsigned long)v.srows) {
      const long idx = rel * cols + (i - r * cols);
      const float lg = v.lgstd[idx], s = __expf(lg);
      if (kl_out) klp += m * m - 2.f * lg + s * s;
      if (w) {
        float e;
        if (v.eps) e = v.eps[idx];
        else {
          const float4 z = philox_normal4(v.rng, (uint64_t)idx >> 2);
          const int c = (int)(idx & 3);
          e = c == 0 ? z.x : (c == 1 ? z.y : (c == 2 ? z.z : z.w));
        }
        m += s * e;
      }
    }
    if (w) w[i] = m;
  }
  if (kl_out) {
    const float t = block_sum<TPB / 64>(klp, red);
    if (threadIdx.x == 0 && t != 0.f) atomicAdd(kl_out, t * kl_scale);
  }
}

__global__ __launch_bounds__(TPB) void sample_weight_bwd_kernel(const float* __restrict__ dw, long rows, long cols,
                                                               blm_variational v, float* __restrict__ dmu,
                                                               float* __restrict__ dlg) {
  const long total = rows * cols;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
    const long r = i / cols;
    const float g = dw[i];
    if (dmu) dmu[i] += g;
    const long rel = r - v.row_lo;
    if (dlg && v.lgstd && (unsigned long)rel < (unsigned long)v.srows) {
      const long idx = rel * cols + (i - r * cols);
      float e;
      if (v.eps) e = v.eps[idx];
      else {
        const float4 z = philox_normal4(v.rng, (uint64_t)idx >> 2);
        const int c = (int)(idx & 3);
        e = c == 0 ? z.x : (c == 1 ? z.y : (c == 2 ? z.z : z.w));
      }
      dlg[idx] += g * e * __expf(v.lgstd[idx]);
    }
  }
}

__global__ __launch_bounds__(TPB) void philox_normal_kernel(float* out, long n, blm_rng rng) {
  const long nblk = (n + 3) >> 2;
  for (long b = (long)blockIdx.x * TPB + threadIdx.x; b < nblk; b += (long)gridDim.x * TPB) {
    const float4 z = philox_normal4(rng, (uint64_t)b);
    const long i = b << 2;
    if (i < n) out[i] = z.x;
    if (i + 1 < n) out[i + 1] = z.y;
    if (i + 2 < n) out[i + 2] = z.z;
    if (i + 3 < n) out[i + 3] = z.w;
  }
}

__global__ __launch_bounds__(TPB) void kl_fwd_kernel(const float* __restrict__ mu, long ld, const float* __restrict__ lg,
                                                     long rows, long cols, float minus, float scale, float* out) {
  __shared__ float red[TPB / 64];
  const long total = rows * cols;
  float acc = 0.f;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
    const long r = i / cols, c = i - r * cols;
    const float m = mu[r * ld + c], l = lg[i], s = __expf(l);
    acc += m * m - 2.f * l + s * s - minus;
  }
  const float t = block_sum<TPB / 64>(acc, red);
  if (threadIdx.x == 0) atomicAdd(out, t * scale);
}

// Contiguous case (ld == cols: the whole tensor, or a window of full rows): no index arithmetic, 16-byte loads, four
// independent load pairs in flight per thread (the scalar kernel above spends its time in 64-bit divisions and one
// dependent load per iteration: 16.8 MB took 24-32 us = 6-9 % of the HBM rate).
__global__ __launch_bounds__(TPB) void kl_fwd_flat4_kernel(const float4* __restrict__ mu, const float4* __restrict__ lg, long n4,
                                                           float minus, float scale, float* out) {
  __shared__ float red[TPB / 64];
  float acc = 0.f;
  const long stride = (long)gridDim.x * TPB;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n4; i += 4 * stride) {
    float4 m[4], l[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long j = i + u * stride;
      const bool in = j < n4;
      m[u] = in ? mu[j] : make_float4(0.f, 0.f, 0.f, 0.f);
      l[u] = in ? lg[j] : make_float4(0.f, 0.f, 0.f, 0.f);  // exp(0)^2 - 2*0 + 0 - minus: corrected below
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (i + u * stride < n4) {
        const float s0 = __expf(l[u].x), s1 = __expf(l[u].y), s2 = __expf(l[u].z), s3 = __expf(l[u].w);
        acc += (m[u].x * m[u].x - 2.f * l[u].x + s0 * s0 - minus) + (m[u].y * m[u].y - 2.f * l[u].y + s1 * s1 - minus) +
               (m[u].z * m[u].z - 2.f * l[u].z + s2 * s2 - minus) + (m[u].w * m[u].w - 2.f * l[u].w + s3 * s3 - minus);
      }
    }
  }
  const float t = block_sum<TPB / 64>(acc, red);
  if (threadIdx.x == 0) atomicAdd(out, t * scale);
}

__global__ __launch_bounds__(TPB) void kl_bwd_kernel(const float* __restrict__ mu, long ld, const float* __restrict__ lg,
                                                     long rows, long cols, const float* g_dev, float scale,
                                                     float* dmu, long ld_dmu, float* dlg) {
  const long total = rows * cols;
  const float g = g_dev[0] * scale;  // scale = weight / n
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
    const long r = i / cols, c = i - r * cols;
    const float s = __expf(lg[i]);
    dmu[r * ld_dmu + c] += g * mu[r * ld + c];
    dlg[i] += g * (s * s - 1.0f);
  }
}

static int grid_for(long work_items) {
  long g = (work_items + TPB - 1) / TPB;
  if (g > 2048) g = 2048;  // 256 CUs x 8 blocks, grid-stride the rest
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace blm

using namespace blm;

static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int blm_sample_weight(const float* mu, int64_t rows, int64_t cols, const blm_variational* v, float* w_out,
                                 float* kl_out, float kl_weight, void* stream) {
  if (!mu || !v || rows < 0 || cols < 0) return blm_fail(BLM_ERR_INVALID, "blm_sample_weight: bad arguments");
  if (!w_out && !kl_out) return BLM_OK;
  if (rows == 0 || cols == 0) return BLM_OK;
  if (v->lgstd && (v->row_lo < 0 || v->srows < 0 || v->row_lo + v->srows > rows))
    return blm_fail(BLM_ERR_INVALID, "blm_sample_weight: noisy row window outside W");
  if (kl_out && !v->lgstd) return blm_fail(BLM_ERR_INVALID, "blm_sample_weight: KL requested without lgstd");
  const float kl_scale = v->srows > 0 ? 0.5f * kl_weight / ((float)v->srows * (float)cols) : 0.f;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool vec = (cols % 4 == 0) && al16(mu) && (!w_out || al16(w_out)) && (!v->lgstd || al16(v->lgstd)) &&
                   (!v->eps || al16(v->eps));
  if (vec)
    hipLaunchKernelGGL(sample_weight_vec4, dim3(grid_for(rows * cols / 4)), dim3(TPB), 0, st, mu, (long)rows,
                       (long)cols, *v, w_out, kl_out, kl_scale);
  else
    hipLaunchKernelGGL(sample_weight_scalar, dim3(grid_for(rows * cols)), dim3(TPB), 0, st, mu, (long)rows, (long)cols,
                       *v, w_out, kl_out, kl_scale);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int blm_sample_weight_bwd(const float* dw, int64_t rows, int64_t cols, const blm_variational* v, float* dmu,
                                     float* dlgstd, void* stream) {
  if (!dw || !v || rows < 0 || cols < 0) return blm_fail(BLM_ERR_INVALID, "blm_sample_weight_bwd: bad arguments");
  if (rows == 0 || cols == 0 || (!dmu && !dlgstd)) return BLM_OK;
  if (dlgstd && (!v->lgstd || v->row_lo < 0 || v->srows < 0 || v->row_lo + v->srows > rows))
    return blm_fail(BLM_ERR_INVALID, "blm_sample_weight_bwd: noisy row window outside W");
  hipLaunchKernelGGL(sample_weight_bwd_kernel, dim3(grid_for(rows * cols)), dim3(TPB), 0, static_cast<hipStream_t>(stream),
                     dw, (long)rows, (long)cols, *v, dmu, dlgstd);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int blm_philox_normal(float* out, int64_t n, const blm_rng* rng, void* stream) {
  if (!out || !rng || n < 0) return blm_fail(BLM_ERR_INVALID, "blm_philox_normal: bad arguments");
  if (n == 0) return BLM_OK;
  hipLaunchKernelGGL(philox_normal_kernel, dim3(grid_for((n + 3) / 4)), dim3(TPB), 0, static_cast<hipStream_t>(stream),
                     out, (long)n, *rng);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int blm_kl_mean_fwd(const float* mu, int64_t ld_mu, const float* lgstd, int64_t rows, int64_t cols,
                               int minus_one, float weight, float* out, void* stream) {
  if (!mu || !lgstd || !out || rows <= 0 || cols <= 0 || ld_mu < cols)
    return blm_fail(BLM_ERR_INVALID, "blm_kl_mean_fwd: bad arguments");
  const float scale = 0.5f * weight / ((float)rows * (float)cols);
  // every block ends in ONE float atomic on the same address: 256 blocks (one per CU) instead of 2048 cut the
  // launch from 32 to a few microseconds (the atomics serialise in L2)
  const int kl_grid = grid_for(rows * cols) < 256 ? grid_for(rows * cols) : 256;
  if (ld_mu == cols && (rows * cols) % 4 == 0 && al16(mu) && al16(lgstd)) {
    const long n4 = rows * cols / 4;
    const int g4 = grid_for(n4 / 4 + 1) < 512 ? grid_for(n4 / 4 + 1) : 512;
    hipLaunchKernelGGL(kl_fwd_flat4_kernel, dim3(g4), dim3(TPB), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const float4*>(mu), reinterpret_cast<const float4*>(lgstd), n4,
                       minus_one ? 1.0f : 0.0f, scale, out);
    BLM_HIP(hipGetLastError());
    return BLM_OK;
  }
  hipLaunchKernelGGL(kl_fwd_kernel, dim3(kl_grid), dim3(TPB), 0, static_cast<hipStream_t>(stream), mu,
                     (long)ld_mu, lgstd, (long)rows, (long)cols, minus_one ? 1.0f : 0.0f, scale, out);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int blm_kl_mean_bwd(const float* mu, int64_t ld_mu, const float* lgstd, int64_t rows, int64_t cols,
                               const float* g_dev, float weight, float* dmu, int64_t ld_dmu, float* dlgstd,
                               void* stream) {
  if (!mu || !lgstd || !g_dev || !dmu || !dlgstd || rows <= 0 || cols <= 0 || ld_mu < cols || ld_dmu < cols)
    return blm_fail(BLM_ERR_INVALID, "blm_kl_mean_bwd: bad arguments");
  const float scale = weight / ((float)rows * (float)cols);
  hipLaunchKernelGGL(kl_bwd_kernel, dim3(grid_for(rows * cols)), dim3(TPB), 0, static_cast<hipStream_t>(stream), mu,
                     (long)ld_mu, lgstd, (long)rows, (long)cols, g_dev, scale, dmu, (long)ld_dmu, dlgstd);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}
