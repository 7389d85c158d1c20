// Architecture-search (super-net) operators: the softmax(alpha)-weighted mix of two candidate
// branches, forward and backward with the gradient of the mixing weights; the LSTM search cell whose
// four gates are each such a mix (standard gate | `Bayes` gate); Adam for the handful of
// architecture logits.  All HBM-bound streaming kernels: float4 accesses, one pass, the mixing-weight
// gradients leave as per-block partial sums (fixed order inside a block, no float atomics) that the
// caller reduces with blm_colsum.
//
// Replaces model_search_bayes.py:234-236 (GaussTransSearchEncoderLayer), :77-78
// (BayesTransSearchEncoderLayer), :686-710 (BayesLSTMSearchCell.bayeslstm) and the torch.optim.Adam
// of architect.py:33.
#include "blm_device.h"
#include "blm_host.h"
#include "blm_dropkey.h"

namespace blm {

constexpr int TPB = 256;
constexpr int MIX_MAX_GRID = 2048;

static int grid_for(long items) {
  long g = (items + TPB - 1) / TPB;
  if (g > MIX_MAX_GRID) g = MIX_MAX_GRID;
  if (g < 1) g = 1;
  return (int)g;
}

// ------------------------------------------------------------------ out = (p0 a + p1 b) * keep
__global__ __launch_bounds__(TPB) void mix2_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                       const float* __restrict__ probs, float* __restrict__ out,
                                                       long rows, DropKey dk) {
  const float p0 = probs[0], p1 = probs[1];
  const int B = dk.B, D = dk.D;
  if ((D & 3) == 0) {
    const long d4 = D >> 2, total = rows * B * d4;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
      const long rb = i / d4;
      const int j = (int)(i - rb * d4) << 2;
      const float4 kp = keep4(dk, (int)(rb / B), (int)(rb % B), j);
      const float4 x = *reinterpret_cast<const float4*>(a + rb * D + j);
      const float4 y = *reinterpret_cast<const float4*>(b + rb * D + j);
      float4 o;
      o.x = (p0 * x.x + p1 * y.x) * kp.x; o.y = (p0 * x.y + p1 * y.y) * kp.y;
      o.z = (p0 * x.z + p1 * y.z) * kp.z; o.w = (p0 * x.w + p1 * y.w) * kp.w;
      *reinterpret_cast<float4*>(out + rb * D + j) = o;
    }
  } else {
    const long total = rows * B * D;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
      const long rb = i / D;
      out[i] = (p0 * a[i] + p1 * b[i]) * keep1(dk, (int)(rb / B), (int)(rb % B), (int)(i - rb * D));
    }
  }
}

// g = dout * keep;  da = p0 g (* mul_a);  db = p1 g;  partial[2*block + {0,1}] = sum g a, sum g b.
// GP: branch b is a GPNN mixture of its saved pre-activation z_b (coef (4,N)): db = p1 g mixture'(z_b) is what the
// next GEMMs need, and dhk (optional) keeps p1 g for the coefficient gradient -- the separate GP backward pass
// (another read of 2 and write of 1 activation-sized tensors) disappears.
template <bool GP>
__global__ __launch_bounds__(TPB) void mix2_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ a,
                                                       const float* __restrict__ b, const float* __restrict__ probs,
                                                       const float* __restrict__ mul_a, float* __restrict__ da,
                                                       float* __restrict__ db, float* __restrict__ partial, long rows,
                                                       DropKey dk, const float* __restrict__ z_b,
                                                       const float* __restrict__ coef, float* __restrict__ dhk) {
  __shared__ float red[TPB / 64];
  const float p0 = probs[0], p1 = probs[1];
  const int B = dk.B, D = dk.D;
  float sa = 0.f, sb = 0.f;
  if ((D & 3) == 0) {
    const long d4 = D >> 2, total = rows * B * d4;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
      const long rb = i / d4;
      const int j = (int)(i - rb * d4) << 2;
      const long o = rb * D + j;
      const float4 kp = keep4(dk, (int)(rb / B), (int)(rb % B), j);
      float4 g = *reinterpret_cast<const float4*>(dout + o);
      g.x *= kp.x; g.y *= kp.y; g.z *= kp.z; g.w *= kp.w;
      const float4 x = *reinterpret_cast<const float4*>(a + o);
      const float4 y = *reinterpret_cast<const float4*>(b + o);
      sa += g.x * x.x + g.y * x.y + g.z * x.z + g.w * x.w;
      sb += g.x * y.x + g.y * y.y + g.z * y.z + g.w * y.w;
      if (da) {
        float4 m = mul_a ? *reinterpret_cast<const float4*>(mul_a + o) : make_float4(1.f, 1.f, 1.f, 1.f);
        m.x *= p0 * g.x; m.y *= p0 * g.y; m.z *= p0 * g.z; m.w *= p0 * g.w;
        *reinterpret_cast<float4*>(da + o) = m;
      }
      float4 gb = make_float4(p1 * g.x, p1 * g.y, p1 * g.z, p1 * g.w);
      if constexpr (GP) {
        if (dhk) *reinterpret_cast<float4*>(dhk + o) = gb;
        const float4 z = *reinterpret_cast<const float4*>(z_b + o);
        gb.x *= dgp_mix(z.x, coef, D, j); gb.y *= dgp_mix(z.y, coef, D, j + 1);
        gb.z *= dgp_mix(z.z, coef, D, j + 2); gb.w *= dgp_mix(z.w, coef, D, j + 3);
      }
      if (db) *reinterpret_cast<float4*>(db + o) = gb;
    }
  } else {
    const long total = rows * B * D;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
      const long rb = i / D;
      const float g = dout[i] * keep1(dk, (int)(rb / B), (int)(rb % B), (int)(i - rb * D));
      sa += g * a[i];
      sb += g * b[i];
      if (da) da[i] = p0 * g * (mul_a ? mul_a[i] : 1.f);
      float gb = p1 * g;
      if constexpr (GP) {
        if (dhk) dhk[i] = gb;
        gb *= dgp_mix(z_b[i], coef, D, (int)(i - rb * D));
      }
      if (db) db[i] = gb;
    }
  }
  const float ta = block_sum<TPB / 64>(sa, red);
  const float tb = block_sum<TPB / 64>(sb, red);
  if (threadIdx.x == 0) { partial[2 * blockIdx.x] = ta; partial[2 * blockIdx.x + 1] = tb; }
}

// ------------------------------------------------------------------ LSTM search cell
// z8 = xw8 + hw8, row layout [i f g o | i' f' g' o'] (H each); probs[k][2], k = i,f,g,o
__global__ __launch_bounds__(TPB) void search_cell_fwd_kernel(const float* __restrict__ xw, const float* __restrict__ hw,
                                                              const float* __restrict__ c_prev,
                                                              const float* __restrict__ probs, float* __restrict__ h,
                                                              float* __restrict__ c, float* __restrict__ acts, int B,
                                                              int H) {
  float p[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) p[k] = probs[k];
  const long total = (long)B * H;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
    const long b = i / H, j = i - b * H, o = b * 8 * H + j;
    float a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float z = xw[o + (long)k * H] + hw[o + (long)k * H];
      a[k] = (k & 3) == 2 ? tanhf(z) : sigmoidf_(z);
    }
    const float gi = a[0] * p[0] + a[4] * p[1];
    const float gf = a[1] * p[2] + a[5] * p[3];
    const float gg = a[2] * p[4] + a[6] * p[5];
    const float go = a[3] * p[6] + a[7] * p[7];
    const float cn = gf * c_prev[i] + gi * gg;
    c[i] = cn;
    h[i] = go * tanhf(cn);
    if (acts) {
#pragma unroll
      for (int k = 0; k < 8; ++k) acts[o + (long)k * H] = a[k];
    }
  }
}

// partial[8*block + 2k + s] = sum over this block's elements of dgate_k * act_{k,s}
__global__ __launch_bounds__(TPB) void search_cell_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ dh2,
                                                              const float* __restrict__ dc_next,
                                                              const float* __restrict__ c_prev, const float* __restrict__ c,
                                                              const float* __restrict__ acts, const float* __restrict__ probs,
                                                              float* __restrict__ dz, float* __restrict__ dc_prev,
                                                              float* __restrict__ partial, int B, int H) {
  __shared__ float red[TPB / 64];
  float p[8], s[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { p[k] = probs[k]; s[k] = 0.f; }
  const long total = (long)B * H;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
    const long b = i / H, j = i - b * H, o = b * 8 * H + j;
    float a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = acts[o + (long)k * H];
    const float gi = a[0] * p[0] + a[4] * p[1];
    const float gf = a[1] * p[2] + a[5] * p[3];
    const float gg = a[2] * p[4] + a[6] * p[5];
    const float go = a[3] * p[6] + a[7] * p[7];
    const float tc = tanhf(c[i]);
    const float dhv = dh[i] + (dh2 ? dh2[i] : 0.f);
    const float dc = (dc_next ? dc_next[i] : 0.f) + dhv * go * (1.f - tc * tc);
    float dg[4];
    dg[0] = dc * gg;
    dg[1] = dc * c_prev[i];
    dg[2] = dc * gi;
    dg[3] = dhv * tc;
    dc_prev[i] = dc * gf;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float av = a[k + 4 * q];
        const float dact = k == 2 ? 1.f - av * av : av * (1.f - av);
        dz[o + (long)(k + 4 * q) * H] = dg[k] * p[2 * k + q] * dact;
        s[2 * k + q] += dg[k] * av;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float t = block_sum<TPB / 64>(s[k], red);
    if (threadIdx.x == 0) partial[8 * blockIdx.x + k] = t;
  }
}

// ------------------------------------------------------------------ Adam (L2 weight decay)
__global__ __launch_bounds__(TPB) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                                   float wd, float bc1, float bc2) {
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long)gridDim.x * TPB) {
    const float gv = g[i] + wd * p[i];
    const float mv = b1 * m[i] + (1.f - b1) * gv;
    const float vv = b2 * v[i] + (1.f - b2) * gv * gv;
    m[i] = mv;
    v[i] = vv;
    p[i] -= lr * (mv / bc1) / (sqrtf(vv / bc2) + eps);
  }
}

}  // namespace blm

using namespace blm;
#define ST static_cast<hipStream_t>(stream)

static int mix_grid(long rows, int B, int N) { return grid_for(((N & 3) == 0 ? (rows * B * N) >> 2 : rows * B * N)); }

extern "C" int64_t blm_mix2_partials(int rows, int B, int N) {
  if (!blm::extents_ok({rows, B, N})) return 0;
  return 2 * (int64_t)mix_grid(rows, B, N);
}

extern "C" int blm_mix2_fwd(const float* a, const float* b, const float* probs, float* out, int rows, int B, int N,
                            float drop_p, const blm_rng* rng, int col_offset, int global_cols, void* stream) {
  if (!a || !b || !probs || !out || !blm::extents_ok({rows, B, N})) return blm_fail(BLM_ERR_INVALID, "blm_mix2_fwd: bad arguments");
  if (drop_p > 0.f && !rng) return blm_fail(BLM_ERR_INVALID, "blm_mix2_fwd: dropout needs rng");
  if ((long)rows * B * N == 0) return BLM_OK;
  hipLaunchKernelGGL(mix2_fwd_kernel, dim3(mix_grid(rows, B, N)), dim3(TPB), 0, ST, a, b, probs, out, (long)rows,
                     make_key(drop_p, rng, B, N, col_offset, global_cols));
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int blm_mix2_bwd(const float* dout, const float* a, const float* b, const float* probs, const float* mul_a,
                            float* da, float* db, float* partial, int rows, int B, int N, float drop_p, const blm_rng* rng,
                            int col_offset, int global_cols, void* stream) {
  if (!dout || !a || !b || !probs || !partial || !blm::extents_ok({rows, B, N}))
    return blm_fail(BLM_ERR_INVALID, "blm_mix2_bwd: bad arguments");
  if (drop_p > 0.f && !rng) return blm_fail(BLM_ERR_INVALID, "blm_mix2_bwd: dropout needs rng");
  hipLaunchKernelGGL(mix2_bwd_kernel<false>, dim3(mix_grid(rows, B, N)), dim3(TPB), 0, ST, dout, a, b, probs, mul_a, da, db,
                     partial, (long)rows, make_key(drop_p, rng, B, N, col_offset, global_cols), nullptr, nullptr, nullptr);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int blm_mix2_gp_bwd(const float* dout, const float* a, const float* b, const float* probs, const float* mul_a,
                               const float* z_b, const float* coef, float* da, float* dz_b, float* dhk, float* partial, int rows,
                               int B, int N, float drop_p, const blm_rng* rng, int col_offset, int global_cols, void* stream) {
  if (!dout || !a || !b || !probs || !z_b || !coef || !partial || !blm::extents_ok({rows, B, N}))
    return blm_fail(BLM_ERR_INVALID, "blm_mix2_gp_bwd: bad arguments");
  if (drop_p > 0.f && !rng) return blm_fail(BLM_ERR_INVALID, "blm_mix2_gp_bwd: dropout needs rng");
  hipLaunchKernelGGL(mix2_bwd_kernel<true>, dim3(mix_grid(rows, B, N)), dim3(TPB), 0, ST, dout, a, b, probs, mul_a, da, dz_b,
                     partial, (long)rows, make_key(drop_p, rng, B, N, col_offset, global_cols), z_b, coef, dhk);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int64_t blm_lstm_search_cell_partials(int B, int H) {
  if (B < 0 || H < 0) return 0;
  return 8 * (int64_t)grid_for((long)B * H);
}

extern "C" int blm_lstm_search_cell_fwd(const float* xw8, const float* hw8, const float* c_prev, const float* probs,
                                        float* h, float* c, float* acts8, int B, int H, void* stream) {
  if (!xw8 || !hw8 || !c_prev || !probs || !h || !c || B < 0 || H < 0)
    return blm_fail(BLM_ERR_INVALID, "blm_lstm_search_cell_fwd: bad arguments");
  if ((long)B * H == 0) return BLM_OK;
  hipLaunchKernelGGL(search_cell_fwd_kernel, dim3(grid_for((long)B * H)), dim3(TPB), 0, ST, xw8, hw8, c_prev, probs, h, c,
                     acts8, B, H);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int blm_lstm_search_cell_bwd(const float* dh, const float* dh2, const float* dc_next, const float* c_prev,
                                        const float* c, const float* acts8, const float* probs, float* dz8, float* dc_prev,
                                        float* partial, int B, int H, void* stream) {
  if (!dh || !c_prev || !c || !acts8 || !probs || !dz8 || !dc_prev || !partial || B < 0 || H < 0)
    return blm_fail(BLM_ERR_INVALID, "blm_lstm_search_cell_bwd: bad arguments");
  hipLaunchKernelGGL(search_cell_bwd_kernel, dim3(grid_for((long)B * H)), dim3(TPB), 0, ST, dh, dh2, dc_next, c_prev, c, acts8,
                     probs, dz8, dc_prev, partial, B, H);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int blm_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                             float eps, float weight_decay, int step, void* stream) {
  if (!p || !g || !m || !v || n < 0 || step < 1) return blm_fail(BLM_ERR_INVALID, "blm_adam_step: bad arguments");
  if (n == 0) return BLM_OK;
  const float bc1 = (float)(1.0 - pow((double)beta1, step)), bc2 = (float)(1.0 - pow((double)beta2, step));
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(TPB), 0, ST, p, g, m, v, (long)n, lr, beta1, beta2, eps, weight_decay,
                     bc1, bc2);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}
