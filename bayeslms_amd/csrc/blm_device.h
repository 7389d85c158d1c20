// Device-side helpers shared by the gfx950 kernels: Philox4x32-10, Box-Muller,
// wave/block reductions, exact-erf GELU.  64-lane wavefronts throughout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bayeslm.h"

#define BLM_WAVE 64

namespace blm {

// ---------------------------------------------------------------- Philox4x32-10
// Same generator as oracle/philox.py (Random123 philox4x32_R(10)).
struct u32x4 {
  uint32_t x, y, z, w;
};

__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                               uint32_t k1) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;  // one v_mad_u64_u32 each (hi and lo together)
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += W0; k1 += W1;
  }
  return {c0, c1, c2, c3};
}

// Same function with the round loop kept rolled: for epilogues where the call sits inside a fully
// unrolled accumulator walk and code size, not latency, is what matters.
__device__ __noinline__ u32x4 philox4x32_10_rolled(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                   uint32_t k1) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll 1
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;  // one v_mad_u64_u32 each (hi and lo together)
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += W0; k1 += W1;
  }
  return {c0, c1, c2, c3};
}

__device__ __forceinline__ u32x4 philox_block(const blm_rng& r, uint64_t block) {
  return philox4x32_10((uint32_t)block, (uint32_t)(block >> 32), r.stream, r.step, (uint32_t)r.seed,
                       (uint32_t)(r.seed >> 32));
}

// u1 = ((ra>>9)+1)*2^-23 in (0,1], u2 = (rb>>8)*2^-24 in [0,1).
// v_sin_f32 / v_cos_f32 take revolutions, so cos(2*pi*u2) is one instruction.
__device__ __forceinline__ void box_muller(uint32_t ra, uint32_t rb, float& z0, float& z1) {
  const float u1 = ((float)(ra >> 9) + 1.0f) * 1.1920928955078125e-07f;
  const float u2 = (float)(rb >> 8) * 5.9604644775390625e-08f;
  const float r = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));  // -2 ln u = -2 ln2 log2 u
  z0 = r * __builtin_amdgcn_cosf(u2);
  z1 = r * __builtin_amdgcn_sinf(u2);
}

// Four N(0,1) values of counter block `block` (elements 4*block .. 4*block+3).
__device__ __forceinline__ float4 philox_normal4(const blm_rng& r, uint64_t block) {
  const u32x4 u = philox_block(r, block);
  float4 z;
  box_muller(u.x, u.y, z.x, z.y);
  box_muller(u.z, u.w, z.z, z.w);
  return z;
}

// One N(0,1) value (element `idx` of the stream), compact code.
__device__ __forceinline__ float philox_normal1_rolled(const blm_rng& r, uint64_t idx) {
  const uint64_t block = idx >> 2;
  const u32x4 u = philox4x32_10_rolled((uint32_t)block, (uint32_t)(block >> 32), r.stream, r.step, (uint32_t)r.seed,
                                       (uint32_t)(r.seed >> 32));
  const bool hi = idx & 2;
  float z0, z1;
  box_muller(hi ? u.z : u.x, hi ? u.w : u.y, z0, z1);
  return (idx & 1) ? z1 : z0;
}
__device__ __forceinline__ uint32_t philox_bits1_rolled(const blm_rng& r, uint64_t idx) {
  const uint64_t block = idx >> 2;
  const u32x4 u = philox4x32_10_rolled((uint32_t)block, (uint32_t)(block >> 32), r.stream, r.step, (uint32_t)r.seed,
                                       (uint32_t)(r.seed >> 32));
  const int c = (int)(idx & 3);
  return c == 0 ? u.x : (c == 1 ? u.y : (c == 2 ? u.z : u.w));
}

__device__ __forceinline__ uint32_t dropout_threshold(float p) {
  const double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
}

// ---------------------------------------------------------------- reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block reductions for blocks of NW waves; `red` is >= NW floats of LDS.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) t += red[i];
  return t;
}
template <int NW>
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = red[0];
#pragma unroll
  for (int i = 1; i < NW; ++i) t = fmaxf(t, red[i]);
  return t;
}

// ---------------------------------------------------------------- activations
// Exact-erf GELU (nn.GELU() default, model.py:1035).  erf by Abramowitz-Stegun 7.1.26
// (|abs err| <= 1.5e-7, i.e. fp32 rounding level) sharing its exp(-z^2/2) with the Gaussian pdf
// that the derivative needs: one v_exp + one v_rcp per element instead of a libm erff call.
// The reciprocal is the hardware's v_rcp_f32 (1 ulp), not an IEEE division (10 instructions): the polynomial's own error is
// 1.5e-7.  gelu_parts2 is the same arithmetic on two values with the packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32:
// one issue slot for two lanes' worth of work) -- in a GEMM epilogue every vector instruction is paid in matrix time.
typedef float blm_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_parts2(blm_f2 z, blm_f2& cdf, blm_f2& pdf_exp) {
  const blm_f2 x = __builtin_elementwise_abs(z) * 0.70710678118654752f;
  const blm_f2 den = __builtin_elementwise_fma(x, (blm_f2)(0.3275911f), (blm_f2)(1.0f));
  const blm_f2 t = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
  const blm_f2 xx = -x * x;
  pdf_exp = (blm_f2){__expf(xx.x), __expf(xx.y)};
  blm_f2 poly = __builtin_elementwise_fma(t, (blm_f2)(1.061405429f), (blm_f2)(-1.453152027f));
  poly = __builtin_elementwise_fma(t, poly, (blm_f2)(1.421413741f));
  poly = __builtin_elementwise_fma(t, poly, (blm_f2)(-0.284496736f));
  poly = __builtin_elementwise_fma(t, poly, (blm_f2)(0.254829592f));
  const blm_f2 tail = (poly * t) * (0.5f * pdf_exp);
  cdf = (blm_f2){z.x >= 0.f ? 1.0f - tail.x : tail.x, z.y >= 0.f ? 1.0f - tail.y : tail.y};
}
__device__ __forceinline__ void gelu_parts(float z, float& cdf, float& pdf_exp) {
  const float x = fabsf(z) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * x);
  pdf_exp = __expf(-x * x);  // = exp(-z^2/2)
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float tail = 0.5f * poly * pdf_exp;  // = 0.5 erfc(|z|/sqrt2): no cancellation in the tails
  cdf = z >= 0.f ? 1.0f - tail : tail;
}
__device__ __forceinline__ float gelu_erf(float z) {
  float cdf, e;
  gelu_parts(z, cdf, e);
  return z * cdf;
}
__device__ __forceinline__ float dgelu_erf(float z) {
  float cdf, e;
  gelu_parts(z, cdf, e);
  return cdf + z * 0.3989422804014327f * e;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// GPNN activation mixture sum_i act_i(z) coef[i][n] in the fixed slot order tanh, sigmoid, relu, gelu
// (model.py:1885-1899) and its derivative in z; coef is (4, N).
__device__ __forceinline__ float gp_mix(float z, const float* coef, int N, int n) {
  return tanhf(z) * coef[n] + sigmoidf_(z) * coef[N + n] + fmaxf(z, 0.f) * coef[2 * N + n] +
         gelu_erf(z) * coef[3 * N + n];
}
__device__ __forceinline__ float dgp_mix(float z, const float* coef, int N, int n) {
  const float th = tanhf(z), sg = sigmoidf_(z);
  return (1.f - th * th) * coef[n] + sg * (1.f - sg) * coef[N + n] + (z > 0.f ? coef[2 * N + n] : 0.f) +
         dgelu_erf(z) * coef[3 * N + n];
}

}  // namespace blm
