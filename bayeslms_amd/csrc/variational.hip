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

// Deterministic mode (blm_set_option("deterministic", 1)): the KL block sums do not meet in a float atomic; every block leaves
// its (scaled) partial here and ONE block adds them in index order (kl_finish_kernel).  A fixed array in the code object's
// data segment -- the C ABI hands these entry points no workspace -- used in stream order: the mode assumes that the library's
// KL reductions of a process are issued on one stream at a time (they are: the forward's own stream).
constexpr int KL_DET_SLOTS = 4096;
__device__ float g_kl_partial[KL_DET_SLOTS];

__global__ __launch_bounds__(TPB) void kl_finish_kernel(int n, float* __restrict__ out) {
  __shared__ float red[TPB / 64];
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += TPB) a += g_kl_partial[i];
  const float t = block_sum<TPB / 64>(a, red);
  if (threadIdx.x == 0) out[0] += t;
}

__device__ __forceinline__ void kl_commit(float t, float* out, bool det, int slot) {
  if (det) g_kl_partial[slot] = t;
  else if (t != 0.f) atomicAdd(out, t);
}

// rows x cols, cols % 4 == 0, all pointers 16-B aligned.
__global__ __launch_bounds__(TPB) void sample_weight_vec4(const float* __restrict__ mu, long rows, long cols,
                                                          blm_variational v, float* __restrict__ w,
                                                          float* kl_out, float kl_scale, bool det) {
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
    if (threadIdx.x == 0) kl_commit(t * kl_scale, kl_out, det, blockIdx.x);
  }
}

// Any shape / alignment (bias vectors, tiny test layers).
__global__ __launch_bounds__(TPB) void sample_weight_scalar(const float* __restrict__ mu, long rows, long cols,
                                                            blm_variational v, float* __restrict__ w,
                                                            float* kl_out, float kl_scale, bool det) {
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
    if (threadIdx.x == 0) kl_commit(t * kl_scale, kl_out, det, blockIdx.x);
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
                                                     long rows, long cols, float minus, float scale, float* out, bool det) {
  __shared__ float red[TPB / 64];
  const long total = rows * cols;
  float acc = 0.f;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
    const long r = i / cols, c = i - r * cols;
    const float m = mu[r * ld + c], l = lg[i], s = __expf(l);
    acc += m * m - 2.f * l + s * s - minus;
  }
  const float t = block_sum<TPB / 64>(acc, red);
  if (threadIdx.x == 0) { if (det) g_kl_partial[blockIdx.x] = t * scale; else atomicAdd(out, t * scale); }
}

// Contiguous case (ld == cols: the whole tensor, or a window of full rows): no index arithmetic, 16-byte loads, four
// independent load pairs in flight per thread (the scalar kernel above spends its time in 64-bit divisions and one
// dependent load per iteration: 16.8 MB took 24-32 us = 6-9 % of the HBM rate).
__global__ __launch_bounds__(TPB) void kl_fwd_flat4_kernel(const float4* __restrict__ mu, const float4* __restrict__ lg, long n4,
                                                           float minus, float scale, float* out, bool det) {
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
  if (threadIdx.x == 0) { if (det) g_kl_partial[blockIdx.x] = t * scale; else atomicAdd(out, t * scale); }
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

// ---- all variational tensors of a module in one launch (blm_variational_group_fwd / _bwd) -----------------
struct VarGroupP {
  blm_var_item it[BLM_VAR_GROUP_MAX];
  int vec[BLM_VAR_GROUP_MAX];  // float4 path: cols % 4 == 0 and every pointer of the item 16-byte aligned
  float* kl_out;
  const float* g;
  int det;  // forward: KL block sums into g_kl_partial[y * gridDim.x + x] (items without a KL term leave 0)
};

__device__ __forceinline__ float eps_at(const blm_variational& v, unsigned idx) {
  if (v.eps) return v.eps[idx];
  const float4 z = philox_normal4(v.rng, (uint64_t)(idx >> 2));
  const int c = (int)(idx & 3);
  return c == 0 ? z.x : (c == 1 ? z.y : (c == 2 ? z.z : z.w));
}

// grid (GX, items): blockIdx.y picks the item (block-uniform branches only), blockIdx.x strides over its elements.
__global__ __launch_bounds__(TPB) void var_group_fwd_kernel(const VarGroupP p) {
  __shared__ float red[TPB / 64];
  const blm_var_item& it = p.it[blockIdx.y];
  const blm_variational& v = it.v;
  const unsigned cols = (unsigned)it.cols, stride = gridDim.x * TPB;
  const bool kl_on = it.kl_weight != 0.f && v.lgstd;
  float* __restrict__ w = it.w_out;
  float klp = 0.f;
  if (p.vec[blockIdx.y]) {
    const unsigned c4n = cols >> 2, total = (unsigned)it.rows * c4n;
    for (unsigned i = blockIdx.x * TPB + threadIdx.x; i < total; i += stride) {
      const unsigned r = i / c4n, c = (i - r * c4n) << 2;
      float4 m = *reinterpret_cast<const float4*>(it.mu + (size_t)r * cols + c);
      const unsigned rel = r - (unsigned)v.row_lo;
      if (v.lgstd && rel < (unsigned)v.srows) {
        const unsigned idx = rel * cols + c;
        const float4 lg = *reinterpret_cast<const float4*>(v.lgstd + idx);
        const float sx = __expf(lg.x), sy = __expf(lg.y), sz = __expf(lg.z), sw = __expf(lg.w);
        if (kl_on)
          klp += (m.x * m.x - 2.f * lg.x + sx * sx - it.kl_minus) + (m.y * m.y - 2.f * lg.y + sy * sy - it.kl_minus) +
                 (m.z * m.z - 2.f * lg.z + sz * sz - it.kl_minus) + (m.w * m.w - 2.f * lg.w + sw * sw - it.kl_minus);
        if (w) {
          float4 z;
          if (v.eps) z = *reinterpret_cast<const float4*>(v.eps + idx);
          else z = philox_normal4(v.rng, (uint64_t)(idx >> 2));
          m.x += sx * z.x; m.y += sy * z.y; m.z += sz * z.z; m.w += sw * z.w;
        }
      }
      if (w) *reinterpret_cast<float4*>(w + (size_t)r * cols + c) = m;
    }
  } else {
    const unsigned total = (unsigned)it.rows * cols;
    for (unsigned i = blockIdx.x * TPB + threadIdx.x; i < total; i += stride) {
      const unsigned r = i / cols;
      float m = it.mu[i];
      const unsigned rel = r - (unsigned)v.row_lo;
      if (v.lgstd && rel < (unsigned)v.srows) {
        const unsigned idx = rel * cols + (i - r * cols);
        const float lg = v.lgstd[idx], sg = __expf(lg);
        if (kl_on) klp += m * m - 2.f * lg + sg * sg - it.kl_minus;
        if (w) m += sg * eps_at(v, idx);
      }
      if (w) w[i] = m;
    }
  }
  if (kl_on) {  // block-uniform
    const float t = block_sum<TPB / 64>(klp, red);
    if (threadIdx.x == 0)
      kl_commit(t * (0.5f * it.kl_weight / ((float)v.srows * (float)cols)), p.kl_out, p.det != 0, blockIdx.y * gridDim.x + blockIdx.x);
  } else if (p.det && p.kl_out && threadIdx.x == 0) {
    g_kl_partial[blockIdx.y * gridDim.x + blockIdx.x] = 0.f;
  }
}

__global__ __launch_bounds__(TPB) void var_group_bwd_kernel(const VarGroupP p) {
  const blm_var_item& it = p.it[blockIdx.y];
  const blm_variational& v = it.v;
  const unsigned cols = (unsigned)it.cols, stride = gridDim.x * TPB;
  const float* __restrict__ dw = it.dw;
  float* __restrict__ dmu = it.dmu;
  float* __restrict__ dlg = it.dlgstd;
  // KL part: g * kl_weight / n on the noisy rows (0 when there is no KL gradient for this item)
  const float gk = (p.g && it.kl_weight != 0.f && v.lgstd && v.srows > 0) ? p.g[0] * it.kl_weight / ((float)v.srows * (float)cols) : 0.f;
  if (!dw && gk == 0.f) return;
  // without dW only the noisy rows have anything to add
  const unsigned r_lo = dw ? 0u : (unsigned)v.row_lo, nrows = dw ? (unsigned)it.rows : (unsigned)v.srows;
  if (p.vec[blockIdx.y]) {
    const unsigned c4n = cols >> 2, total = nrows * c4n;
    for (unsigned i = blockIdx.x * TPB + threadIdx.x; i < total; i += stride) {
      const unsigned r = r_lo + i / c4n, c = (i % c4n) << 2;
      const size_t o = (size_t)r * cols + c;
      float4 g = dw ? *reinterpret_cast<const float4*>(dw + o) : make_float4(0.f, 0.f, 0.f, 0.f);
      const unsigned rel = r - (unsigned)v.row_lo;
      const bool noisy = v.lgstd && rel < (unsigned)v.srows;
      if (noisy && dlg) {
        const unsigned idx = rel * cols + c;
        const float4 lg = *reinterpret_cast<const float4*>(v.lgstd + idx);
        const float sx = __expf(lg.x), sy = __expf(lg.y), sz = __expf(lg.z), sw = __expf(lg.w);
        float4 d = *reinterpret_cast<float4*>(dlg + idx);
        if (dw) {
          float4 z;
          if (v.eps) z = *reinterpret_cast<const float4*>(v.eps + idx);
          else z = philox_normal4(v.rng, (uint64_t)(idx >> 2));
          d.x += g.x * z.x * sx; d.y += g.y * z.y * sy; d.z += g.z * z.z * sz; d.w += g.w * z.w * sw;
        }
        if (gk != 0.f) {
          d.x += gk * (sx * sx - 1.0f); d.y += gk * (sy * sy - 1.0f); d.z += gk * (sz * sz - 1.0f); d.w += gk * (sw * sw - 1.0f);
        }
        *reinterpret_cast<float4*>(dlg + idx) = d;
      }
      if (dmu) {
        float4 a = *reinterpret_cast<float4*>(dmu + o);
        a.x += g.x; a.y += g.y; a.z += g.z; a.w += g.w;
        if (noisy && gk != 0.f) {
          const float4 m = *reinterpret_cast<const float4*>(it.mu + o);
          a.x += gk * m.x; a.y += gk * m.y; a.z += gk * m.z; a.w += gk * m.w;
        }
        *reinterpret_cast<float4*>(dmu + o) = a;
      }
    }
  } else {
    const unsigned total = nrows * cols;
    for (unsigned i = blockIdx.x * TPB + threadIdx.x; i < total; i += stride) {
      const unsigned r = r_lo + i / cols, c = i % cols;
      const size_t o = (size_t)r * cols + c;
      const float g = dw ? dw[o] : 0.f;
      const unsigned rel = r - (unsigned)v.row_lo;
      const bool noisy = v.lgstd && rel < (unsigned)v.srows;
      if (noisy && dlg) {
        const unsigned idx = rel * cols + c;
        const float sg = __expf(v.lgstd[idx]);
        float d = dlg[idx];
        if (dw) d += g * eps_at(v, idx) * sg;
        if (gk != 0.f) d += gk * (sg * sg - 1.0f);
        dlg[idx] = d;
      }
      if (dmu) {
        float a = dmu[o] + g;
        if (noisy && gk != 0.f) a += gk * it.mu[o];
        dmu[o] = a;
      }
    }
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
  if (!mu || !v || !blm::extents_ok({rows, cols})) return blm_fail(BLM_ERR_INVALID, "blm_sample_weight: bad arguments");
  if (!w_out && !kl_out) return BLM_OK;
  if (rows == 0 || cols == 0) return BLM_OK;
  if (v->lgstd && (v->row_lo < 0 || v->srows < 0 || (int64_t)v->row_lo + v->srows > rows))
    return blm_fail(BLM_ERR_INVALID, "blm_sample_weight: noisy row window outside W");
  if (kl_out && !v->lgstd) return blm_fail(BLM_ERR_INVALID, "blm_sample_weight: KL requested without lgstd");
  const float kl_scale = v->srows > 0 ? 0.5f * kl_weight / ((float)v->srows * (float)cols) : 0.f;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool vec = (cols % 4 == 0) && al16(mu) && (!w_out || al16(w_out)) && (!v->lgstd || al16(v->lgstd)) &&
                   (!v->eps || al16(v->eps));
  const bool det = kl_out && blm::option(blm::OPT_DETERMINISTIC);
  const int g = grid_for(vec ? rows * cols / 4 : rows * cols);  // <= 2048 <= KL_DET_SLOTS
  if (vec)
    hipLaunchKernelGGL(sample_weight_vec4, dim3(g), dim3(TPB), 0, st, mu, (long)rows, (long)cols, *v, w_out, kl_out, kl_scale, det);
  else
    hipLaunchKernelGGL(sample_weight_scalar, dim3(g), dim3(TPB), 0, st, mu, (long)rows, (long)cols, *v, w_out, kl_out, kl_scale, det);
  BLM_HIP(hipGetLastError());
  if (det) {
    hipLaunchKernelGGL(kl_finish_kernel, dim3(1), dim3(TPB), 0, st, g, kl_out);
    BLM_HIP(hipGetLastError());
  }
  return BLM_OK;
}

static int var_group_pack(const blm_var_item* items, int32_t n, bool bwd, VarGroupP& p, long& most, const char* who) {
  if (!items || n < 0 || n > BLM_VAR_GROUP_MAX) return blm_fail(BLM_ERR_INVALID, "%s: 0..%d items", who, BLM_VAR_GROUP_MAX);
  most = 0;
  for (int i = 0; i < n; ++i) {
    const blm_var_item& it = items[i];
    if (!it.mu || it.rows <= 0 || it.cols <= 0 || it.rows * it.cols >= (1LL << 31))
      return blm_fail(BLM_ERR_INVALID, "%s: item %d: bad tensor (mu, rows, cols; at most 2^31 - 1 elements)", who, i);
    if (it.v.lgstd && (it.v.row_lo < 0 || it.v.srows < 0 || (int64_t)it.v.row_lo + it.v.srows > it.rows))
      return blm_fail(BLM_ERR_INVALID, "%s: item %d: noisy row window outside W", who, i);
    if (it.kl_weight != 0.f && !it.v.lgstd) return blm_fail(BLM_ERR_INVALID, "%s: item %d: KL requested without lgstd", who, i);
    if (bwd && it.dlgstd && !it.v.lgstd) return blm_fail(BLM_ERR_INVALID, "%s: item %d: dlgstd without lgstd", who, i);
    p.it[i] = it;
    bool vec = it.cols % 4 == 0 && al16(it.mu) && (!it.v.lgstd || al16(it.v.lgstd)) && (!it.v.eps || al16(it.v.eps));
    if (bwd) vec = vec && (!it.dw || al16(it.dw)) && (!it.dmu || al16(it.dmu)) && (!it.dlgstd || al16(it.dlgstd));
    else vec = vec && (!it.w_out || al16(it.w_out));
    p.vec[i] = vec ? 1 : 0;
    const long work = it.rows * it.cols / (vec ? 4 : 1);
    if (work > most) most = work;
  }
  return BLM_OK;
}

extern "C" int blm_variational_group_fwd(const blm_var_item* items, int32_t n, float* kl_out, void* stream) {
  VarGroupP p;
  long most = 0;
  const int rc = var_group_pack(items, n, false, p, most, "blm_variational_group_fwd");
  if (rc != BLM_OK) return rc;
  bool any_kl = false;
  for (int i = 0; i < n; ++i) any_kl = any_kl || items[i].kl_weight != 0.f;
  if (any_kl && !kl_out) return blm_fail(BLM_ERR_INVALID, "blm_variational_group_fwd: an item has a KL weight but kl_out is NULL");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (kl_out) BLM_HIP(hipMemsetAsync(kl_out, 0, sizeof(float), st));
  if (n == 0) return BLM_OK;
  p.kl_out = kl_out;
  p.g = nullptr;
  // every KL block ends in one float atomic on the same word (they serialise in L2): <= 256 blocks per item
  long gx = (most + 4 * TPB - 1) / (4 * TPB);
  gx = gx < 1 ? 1 : (gx > 256 ? 256 : gx);
  p.det = (any_kl && blm::option(blm::OPT_DETERMINISTIC)) ? 1 : 0;
  if (p.det) while (gx * n > KL_DET_SLOTS) gx = (gx + 1) / 2;  // n <= BLM_VAR_GROUP_MAX: gx stays >= 1
  hipLaunchKernelGGL(var_group_fwd_kernel, dim3((unsigned)gx, (unsigned)n), dim3(TPB), 0, st, p);
  BLM_HIP(hipGetLastError());
  if (p.det) {
    hipLaunchKernelGGL(kl_finish_kernel, dim3(1), dim3(TPB), 0, st, (int)(gx * n), kl_out);
    BLM_HIP(hipGetLastError());
  }
  return BLM_OK;
}

extern "C" int blm_variational_group_bwd(const blm_var_item* items, int32_t n, const float* kl_grad, void* stream) {
  VarGroupP p;
  long most = 0;
  const int rc = var_group_pack(items, n, true, p, most, "blm_variational_group_bwd");
  if (rc != BLM_OK) return rc;
  if (n == 0) return BLM_OK;
  p.kl_out = nullptr;
  p.g = kl_grad;
  p.det = 0;
  long gx = (most + 4 * TPB - 1) / (4 * TPB);
  gx = gx < 1 ? 1 : (gx > 256 ? 256 : gx);
  hipLaunchKernelGGL(var_group_bwd_kernel, dim3((unsigned)gx, (unsigned)n), dim3(TPB), 0, static_cast<hipStream_t>(stream), p);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int blm_sample_weight_bwd(const float* dw, int64_t rows, int64_t cols, const blm_variational* v, float* dmu,
                                     float* dlgstd, void* stream) {
  if (!dw || !v || !blm::extents_ok({rows, cols})) return blm_fail(BLM_ERR_INVALID, "blm_sample_weight_bwd: bad arguments");
  if (rows == 0 || cols == 0 || (!dmu && !dlgstd)) return BLM_OK;
  if (dlgstd && (!v->lgstd || v->row_lo < 0 || v->srows < 0 || (int64_t)v->row_lo + v->srows > rows))
    return blm_fail(BLM_ERR_INVALID, "blm_sample_weight_bwd: noisy row window outside W");
  hipLaunchKernelGGL(sample_weight_bwd_kernel, dim3(grid_for(rows * cols)), dim3(TPB), 0, static_cast<hipStream_t>(stream),
                     dw, (long)rows, (long)cols, *v, dmu, dlgstd);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

// ---- GPNN2 inside the GP-LSTM cells: T frequency matrices per forward (model.py:2062-2065 at every call of the time loop)
// F_t = mean + exp(lgstd) * eps_t, eps_t from the Philox stream (seed, stream, step0 + t) with the element index u * M + m
// that blm_sample_weight uses, or from eps_all (T,H,M).  Written in the two operand layouts of the per-step products:
// FT (T, MP, H) = F_t^T with zero rows m >= M, and Fp (T, H, GP) = F_t with zero columns m >= M.
__global__ __launch_bounds__(TPB) void gpnn2_sample_steps_kernel(const float* __restrict__ mean, const float* __restrict__ lgstd,
                                                                 const float* __restrict__ eps_all, blm_rng rng, int T, int H, int M,
                                                                 int MP, int GP, float* __restrict__ FT, float* __restrict__ Fp) {
  const long per = (long)H * GP, total = (long)T * per;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
    const int t = (int)(i / per);
    const long r = i - (long)t * per;
    const int u = (int)(r / GP), m = (int)(r - (long)u * GP);
    float f = 0.f;
    if (m < M) {
      const unsigned idx = (unsigned)u * (unsigned)M + (unsigned)m;
      float e;
      if (eps_all) e = eps_all[((long)t * H + u) * M + m];
      else {
        blm_variational v{};
        v.rng = rng;
        v.rng.step = rng.step + (uint32_t)t;
        e = eps_at(v, idx);
      }
      f = mean[idx] + __expf(lgstd[idx]) * e;
    }
    Fp[i] = f;
    if (m < MP) FT[((long)t * MP + m) * H + u] = f;
  }
}

// d mean[u,m] += sum_t G_t[u,m],  d lgstd[u,m] += exp(lgstd[u,m]) * sum_t eps_t[u,m] * G_t[u,m],  G_t = pre_t^T d f_t
// (pre (T,B,H): the GPNN2's inputs; df (T,B,GP): the gradients of its features).  A block owns a 16 x 16 tile of (u, m).
__global__ __launch_bounds__(256) void gpnn2_freq_grad_kernel(const float* __restrict__ pre, const float* __restrict__ df,
                                                              const float* __restrict__ eps_all, blm_rng rng,
                                                              const float* __restrict__ lgstd, float* __restrict__ dmean,
                                                              float* __restrict__ dlgstd, int T, int B, int H, int M, int GP) {
  __shared__ float sp[64][17], sd[64][17];
  const int ui = threadIdx.x >> 4, mi = threadIdx.x & 15;
  const int u0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
  const int u = u0 + ui, m = m0 + mi;
  const bool ok = u < H && m < M;
  float am = 0.f, al = 0.f;
  for (int t = 0; t < T; ++t) {
    float dot = 0.f;
    for (int b0 = 0; b0 < B; b0 += 64) {
      __syncthreads();
      for (int e = threadIdx.x; e < 64 * 16; e += 256) {
        const int b = e >> 4, c = e & 15;
        const bool bin = b0 + b < B;
        sp[b][c] = (bin && u0 + c < H) ? pre[((long)t * B + b0 + b) * H + u0 + c] : 0.f;
        sd[b][c] = (bin && m0 + c < GP) ? df[((long)t * B + b0 + b) * GP + m0 + c] : 0.f;
      }
      __syncthreads();
#pragma unroll 8
      for (int b = 0; b < 64; ++b) dot += sp[b][ui] * sd[b][mi];
    }
    if (ok) {
      const unsigned idx = (unsigned)u * (unsigned)M + (unsigned)m;
      float e;
      if (eps_all) e = eps_all[((long)t * H + u) * M + m];
      else {
        blm_variational v{};
        v.rng = rng;
        v.rng.step = rng.step + (uint32_t)t;
        e = eps_at(v, idx);
      }
      am += dot;
      al += dot * e;
    }
  }
  if (ok) {
    const long idx = (long)u * M + m;
    if (dmean) dmean[idx] += am;
    if (dlgstd) dlgstd[idx] += al * __expf(lgstd[idx]);
  }
}

extern "C" int blm_gpnn2_sample_steps(const float* mean, const float* lgstd, const float* eps_all, const blm_rng* rng0, int T,
                                      int H, int M, int MP, int GP, float* FT, float* Fp, void* stream) {
  if (!mean || !lgstd || (!eps_all && !rng0) || !FT || !Fp || H <= 0 || M <= 0 || MP < M || GP < MP || !blm::extents_ok({T, H, GP}))
    return blm_fail(BLM_ERR_INVALID, "blm_gpnn2_sample_steps: bad arguments");
  if (T == 0) return BLM_OK;
  blm_rng r{};
  if (rng0) r = *rng0;
  hipLaunchKernelGGL(gpnn2_sample_steps_kernel, dim3(grid_for((long)T * H * GP)), dim3(TPB), 0, static_cast<hipStream_t>(stream),
                     mean, lgstd, eps_all, r, T, H, M, MP, GP, FT, Fp);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int blm_gpnn2_freq_grad(const float* pre, const float* df, const float* eps_all, const blm_rng* rng0,
                                   const float* lgstd, float* dmean, float* dlgstd, int T, int B, int H, int M, int GP,
                                   void* stream) {
  if (!pre || !df || (!eps_all && !rng0) || !lgstd || (!dmean && !dlgstd) || H <= 0 || M <= 0 || GP < M || !blm::extents_ok({T, B, H}) ||
      !blm::extents_ok({T, B, GP}))
    return blm_fail(BLM_ERR_INVALID, "blm_gpnn2_freq_grad: bad arguments");
  if (T == 0 || B == 0) return BLM_OK;
  blm_rng r{};
  if (rng0) r = *rng0;
  hipLaunchKernelGGL(gpnn2_freq_grad_kernel, dim3((H + 15) / 16, (M + 15) / 16), dim3(256), 0, static_cast<hipStream_t>(stream),
                     pre, df, eps_all, r, lgstd, dmean, dlgstd, T, B, H, M, GP);
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
  if (!mu || !lgstd || !out || rows <= 0 || cols <= 0 || ld_mu < cols || !blm::extents_ok({rows, cols}) || !blm::extents_ok({rows, ld_mu}))
    return blm_fail(BLM_ERR_INVALID, "blm_kl_mean_fwd: bad arguments");
  const float scale = 0.5f * weight / ((float)rows * (float)cols);
  // every block ends in ONE float atomic on the same address: 256 blocks (one per CU) instead of 2048 cut the
  // launch from 32 to a few microseconds (the atomics serialise in L2)
  const int kl_grid = grid_for(rows * cols) < 256 ? grid_for(rows * cols) : 256;
  const bool det = blm::option(blm::OPT_DETERMINISTIC) != 0;  // block partials + one fixed-order finishing block instead of atomics
  hipStream_t st = static_cast<hipStream_t>(stream);
  int blocks;
  if (ld_mu == cols && (rows * cols) % 4 == 0 && al16(mu) && al16(lgstd)) {
    const long n4 = rows * cols / 4;
    blocks = grid_for(n4 / 4 + 1) < 512 ? grid_for(n4 / 4 + 1) : 512;
    hipLaunchKernelGGL(kl_fwd_flat4_kernel, dim3(blocks), dim3(TPB), 0, st, reinterpret_cast<const float4*>(mu),
                       reinterpret_cast<const float4*>(lgstd), n4, minus_one ? 1.0f : 0.0f, scale, out, det);
  } else {
    blocks = kl_grid;
    hipLaunchKernelGGL(kl_fwd_kernel, dim3(blocks), dim3(TPB), 0, st, mu, (long)ld_mu, lgstd, (long)rows, (long)cols,
                       minus_one ? 1.0f : 0.0f, scale, out, det);
  }
  BLM_HIP(hipGetLastError());
  if (det) {
    hipLaunchKernelGGL(kl_finish_kernel, dim3(1), dim3(TPB), 0, st, blocks, out);
    BLM_HIP(hipGetLastError());
  }
  return BLM_OK;
}

extern "C" int blm_kl_mean_bwd(const float* mu, int64_t ld_mu, const float* lgstd, int64_t rows, int64_t cols,
                               const float* g_dev, float weight, float* dmu, int64_t ld_dmu, float* dlgstd,
                               void* stream) {
  if (!mu || !lgstd || !g_dev || !dmu || !dlgstd || rows <= 0 || cols <= 0 || ld_mu < cols || ld_dmu < cols || !blm::extents_ok({rows, cols}) ||
      !blm::extents_ok({rows, ld_mu}) || !blm::extents_ok({rows, ld_dmu}))
    return blm_fail(BLM_ERR_INVALID, "blm_kl_mean_bwd: bad arguments");
  const float scale = weight / ((float)rows * (float)cols);
  hipLaunchKernelGGL(kl_bwd_kernel, dim3(grid_for(rows * cols)), dim3(TPB), 0, static_cast<hipStream_t>(stream), mu,
                     (long)ld_mu, lgstd, (long)rows, (long)cols, g_dev, scale, dmu, (long)ld_dmu, dlgstd);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}
