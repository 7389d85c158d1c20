// Fused causal self-attention, forward and backward, one workgroup per (batch column, head).
// Scores, softmax, probability dropout and P.V never leave the CU: K/V (forward) and then Q/dO
// (backward) tiles sit in LDS, each lane owns one query (or key) row in registers, LDS reads are
// wave-uniform broadcasts.  T <= 128, head_dim in {4,8,16,32,64} (any other size up to 128: two lanes per row); every other shape (head_dim up to 512, any T) goes
// through the untiled one-wave-per-row kernels at the end of this file.
//
// Replaces model.py:889-920 (MultiheadAttention.forward core: scale, bmm, +mask, softmax, dropout,
// bmm) and the same lines of BayesMultiheadAttention (:990-1011), plus their autograd.
// These f32 VALU kernels serve head_dim in {4,8,16,32} (and 64 when BLM_ATTN_VALU=1); head_dim 64,
// the size of every recipe, runs on the matrix cores in attention_mfma.hip.
#include <cstdlib>

#include "blm_device.h"
#include "blm_host.h"

namespace blm {

constexpr int ATT_T = 128;  // threads per block = max sequence length

struct AttnP {
  const float *q, *k, *v;
  long ld;
  float* out;
  float* lse;
  const float *o_in, *dout;
  float *dq, *dk, *dv;
  long ldd;
  int T, B, nhead;
  int hd;  // head_dim (the templated kernels know it at compile time; the two-lanes-per-row kernels read it here)
  float scale;
  blm_rng rng;
  uint32_t thr;
  float inv_keep;
  int col_offset;
  bool drop;
  const float* keep;  // blm_attn_*_keep: the dropout factors themselves, (B_global * nhead, T, T) floats (0 or 1 / (1 - p)), instead of the Philox stream
};

// keep factor for probability element g (global index into (B_global*nhead, T, T))
__device__ __forceinline__ float keep_at(const AttnP& p, uint64_t g) {
  if (p.keep) return p.keep[g];  // wave-uniform branch
  const u32x4 u = philox_block(p.rng, g >> 2);
  const int c = (int)(g & 3);
  const uint32_t bits = c == 0 ? u.x : (c == 1 ? u.y : (c == 2 ? u.z : u.w));
  return bits >= p.thr ? p.inv_keep : 0.f;
}

template <int HD>
__device__ __forceinline__ void load_tile(float* dst, const float* src, long ld, int T, int B, int b, int off, float mul) {
  // dst[t][c] = src[(t*B+b)*ld + off + c] * mul
  constexpr int Q = HD / 4;
  for (int i = threadIdx.x; i < T * Q; i += ATT_T) {
    const int t = i / Q, c = (i - t * Q) * 4;
    const float* s = src + ((long)t * B + b) * ld + off + c;
    float4 v;
    if (((reinterpret_cast<uintptr_t>(s)) & 15) == 0) v = *reinterpret_cast<const float4*>(s);
    else v = make_float4(s[0], s[1], s[2], s[3]);
    v.x *= mul; v.y *= mul; v.z *= mul; v.w *= mul;
    *reinterpret_cast<float4*>(dst + t * HD + c) = v;
  }
}

template <int HD>
__global__ __launch_bounds__(ATT_T) void attn_fwd_kernel(const AttnP p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Ks = sm;
  float* Vs = sm + p.T * HD;
  const int b = blockIdx.x / p.nhead, head = blockIdx.x % p.nhead, off = head * HD;
  const int T = p.T, i = threadIdx.x;
  load_tile<HD>(Ks, p.k, p.ld, T, p.B, b, off, 1.f);
  load_tile<HD>(Vs, p.v, p.ld, T, p.B, b, off, 1.f);
  float q[HD], o[HD];
  const bool act = i < T;
  if (act) {
    const float* qs = p.q + ((long)i * p.B + b) * p.ld + off;
#pragma unroll
    for (int c = 0; c < HD; ++c) { q[c] = qs[c] * p.scale; o[c] = 0.f; }
  } else {
#pragma unroll
    for (int c = 0; c < HD; ++c) { q[c] = 0.f; o[c] = 0.f; }
  }
  __syncthreads();
  float m = -INFINITY, l = 0.f;
  const uint64_t gbase = (((uint64_t)(p.col_offset + b) * p.nhead + head) * T + i) * (uint64_t)T;
  const int wave_last = min(T - 1, (int)(threadIdx.x | 63));
  for (int j = 0; j <= wave_last; ++j) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
      const float4 kk = *reinterpret_cast<const float4*>(Ks + j * HD + c);
      s += q[c] * kk.x + q[c + 1] * kk.y + q[c + 2] * kk.z + q[c + 3] * kk.w;
    }
    if (act && j <= i) {
      const float mn = fmaxf(m, s);
      const float corr = __expf(m - mn), pe = __expf(s - mn);
      l = l * corr + pe;
      const float pd = p.drop ? pe * keep_at(p, gbase + j) : pe;
#pragma unroll
      for (int c = 0; c < HD; c += 4) {
        const float4 vv = *reinterpret_cast<const float4*>(Vs + j * HD + c);
        o[c] = o[c] * corr + pd * vv.x; o[c + 1] = o[c + 1] * corr + pd * vv.y;
        o[c + 2] = o[c + 2] * corr + pd * vv.z; o[c + 3] = o[c + 3] * corr + pd * vv.w;
      }
      m = mn;
    }
  }
  if (act) {
    const float inv = 1.f / l;
    float* os = p.out + ((long)i * p.B + b) * ((long)p.nhead * HD) + off;
#pragma unroll
    for (int c = 0; c < HD; ++c) os[c] = o[c] * inv;
    if (p.lse) p.lse[(long)blockIdx.x * T + i] = m + __logf(l);
  }
}

template <int HD>
__global__ __launch_bounds__(ATT_T) void attn_bwd_kernel(const AttnP p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int T = p.T;
  float* A = sm;                // K, then Q*scale
  float* Bf = sm + T * HD;      // V, then dO
  float* lse_s = sm + 2 * T * HD;
  float* del_s = lse_s + T;
  const int b = blockIdx.x / p.nhead, head = blockIdx.x % p.nhead, off = head * HD;
  const int i = threadIdx.x;
  const bool act = i < T;
  const long dmodel = (long)p.nhead * HD;
  const uint64_t bh = (uint64_t)(p.col_offset + b) * p.nhead + head;

  // ---------------- phase 1: lane = query row -> dQ
  load_tile<HD>(A, p.k, p.ld, T, p.B, b, off, 1.f);
  load_tile<HD>(Bf, p.v, p.ld, T, p.B, b, off, 1.f);
  {
    float q[HD], dO[HD], dq[HD];
    float lse = 0.f, delta = 0.f;
    if (act) {
      const float* qs = p.q + ((long)i * p.B + b) * p.ld + off;
      const float* ds = p.dout + ((long)i * p.B + b) * dmodel + off;
      const float* os = p.o_in + ((long)i * p.B + b) * dmodel + off;
#pragma unroll
      for (int c = 0; c < HD; ++c) { q[c] = qs[c] * p.scale; dO[c] = ds[c]; dq[c] = 0.f; delta += ds[c] * os[c]; }
      lse = p.lse[(long)blockIdx.x * T + i];
      lse_s[i] = lse;
      del_s[i] = delta;
    } else {
#pragma unroll
      for (int c = 0; c < HD; ++c) { q[c] = 0.f; dO[c] = 0.f; dq[c] = 0.f; }
    }
    __syncthreads();
    const int wave_last = min(T - 1, (int)(threadIdx.x | 63));
    for (int j = 0; j <= wave_last; ++j) {
      float s = 0.f, dpd = 0.f;
#pragma unroll
      for (int c = 0; c < HD; c += 4) {
        const float4 kk = *reinterpret_cast<const float4*>(A + j * HD + c);
        const float4 vv = *reinterpret_cast<const float4*>(Bf + j * HD + c);
        s += q[c] * kk.x + q[c + 1] * kk.y + q[c + 2] * kk.z + q[c + 3] * kk.w;
        dpd += dO[c] * vv.x + dO[c + 1] * vv.y + dO[c + 2] * vv.z + dO[c + 3] * vv.w;
      }
      if (act && j <= i) {
        const float pr = __expf(s - lse);
        const float kp = p.drop ? keep_at(p, (bh * T + i) * (uint64_t)T + j) : 1.f;
        const float dS = pr * (dpd * kp - delta);
#pragma unroll
        for (int c = 0; c < HD; c += 4) {
          const float4 kk = *reinterpret_cast<const float4*>(A + j * HD + c);
          dq[c] += dS * kk.x; dq[c + 1] += dS * kk.y; dq[c + 2] += dS * kk.z; dq[c + 3] += dS * kk.w;
        }
      }
    }
    if (act) {
      float* o = p.dq + ((long)i * p.B + b) * p.ldd + off;
#pragma unroll
      for (int c = 0; c < HD; ++c) o[c] = dq[c] * p.scale;
    }
  }
  __syncthreads();
  // ---------------- phase 2: lane = key row -> dK, dV
  load_tile<HD>(A, p.q, p.ld, T, p.B, b, off, p.scale);
  load_tile<HD>(Bf, p.dout, dmodel, T, p.B, b, off, 1.f);
  float k[HD], v[HD];
  const int j = threadIdx.x;
  if (act) {
    const float* ks = p.k + ((long)j * p.B + b) * p.ld + off;
    const float* vs = p.v + ((long)j * p.B + b) * p.ld + off;
#pragma unroll
    for (int c = 0; c < HD; ++c) { k[c] = ks[c]; v[c] = vs[c]; }
  } else {
#pragma unroll
    for (int c = 0; c < HD; ++c) { k[c] = 0.f; v[c] = 0.f; }
  }
  __syncthreads();
  constexpr int HALVES = HD >= 16 ? 2 : 1;
  constexpr int HH = HD / HALVES;
  const int wave_first = threadIdx.x & ~63;
#pragma unroll
  for (int half = 0; half < HALVES; ++half) {
    float dk[HH], dv[HH];
#pragma unroll
    for (int c = 0; c < HH; ++c) { dk[c] = 0.f; dv[c] = 0.f; }
    for (int qi = wave_first; qi < T; ++qi) {
      float s = 0.f, dpd = 0.f;
#pragma unroll
      for (int c = 0; c < HD; c += 4) {
        const float4 qq = *reinterpret_cast<const float4*>(A + qi * HD + c);
        const float4 dd = *reinterpret_cast<const float4*>(Bf + qi * HD + c);
        s += qq.x * k[c] + qq.y * k[c + 1] + qq.z * k[c + 2] + qq.w * k[c + 3];
        dpd += dd.x * v[c] + dd.y * v[c + 1] + dd.z * v[c + 2] + dd.w * v[c + 3];
      }
      if (act && j <= qi) {
        const float pr = __expf(s - lse_s[qi]);
        const float kp = p.drop ? keep_at(p, (bh * T + qi) * (uint64_t)T + j) : 1.f;
        const float dS = pr * (dpd * kp - del_s[qi]);
        const float pd = pr * kp;
#pragma unroll
        for (int c = 0; c < HH; c += 4) {
          const float4 qq = *reinterpret_cast<const float4*>(A + qi * HD + half * HH + c);
          const float4 dd = *reinterpret_cast<const float4*>(Bf + qi * HD + half * HH + c);
          dk[c] += dS * qq.x; dk[c + 1] += dS * qq.y; dk[c + 2] += dS * qq.z; dk[c + 3] += dS * qq.w;
          dv[c] += pd * dd.x; dv[c + 1] += pd * dd.y; dv[c + 2] += pd * dd.z; dv[c + 3] += pd * dd.w;
        }
      }
    }
    if (act) {
      float* ok = p.dk + ((long)j * p.B + b) * p.ldd + off + half * HH;
      float* ov = p.dv + ((long)j * p.B + b) * p.ldd + off + half * HH;
#pragma unroll
      for (int c = 0; c < HH; ++c) { ok[c] = dk[c]; ov[c] = dv[c]; }
    }
  }
}


// ------------------------------------------------------------------ head_dim <= 128 (T <= 128): two lanes per row
// The tiled kernels above hold a row's whole head in registers (q / o, k / v: 2 x head_dim values per lane), which stops at 64.
// Here a row is shared by a lane PAIR -- lane 2 i + part holds features [64 part, 64 part + 64) of row i -- so the register
// picture per lane is the head_dim-64 kernels', a score is the sum of the pair's two half dot products (one DPP exchange), and
// each lane updates its own half of o / dq / dk / dv.  K / V (then Q / dO) rows sit in LDS with the two halves 68 floats apart
// (row stride 136): the pair's two broadcast addresses of a ds_read_b128 fall into different banks.  256 threads per (column, head).
// Any head size up to 128 that has no kernel of its own runs here (p.hd = the real size: features beyond it are zeros in LDS and in
// the registers, train.py's defaults give 100).  Measured (tools/attn_head_dim_probe.py, T 128, B 64, 4 heads x 128): 130 / 407 us
// forward / backward against 776 / 1680 us on the one-wave-per-query kernels.
constexpr int WIDE_HS = 64;               // features per lane
constexpr int WIDE_HD = 2 * WIDE_HS;      // head_dim served
constexpr int WIDE_PS = WIDE_HS + 4;      // distance of the two halves of a row in LDS
constexpr int WIDE_RS = 2 * WIDE_PS;      // LDS row stride

__device__ __forceinline__ void load_tile_wide(float* dst, const float* src, long ld, int T, int B, int b, int off, float mul, int hd) {
  // dst[t][68 * (c / 64) + c % 64] = c < hd ? src[(t*B+b)*ld + off + c] * mul : 0
  constexpr int Q = WIDE_HD / 4;
  for (int i = threadIdx.x; i < T * Q; i += 2 * ATT_T) {
    const int t = i / Q, c = (i - t * Q) * 4;
    const float* s = src + ((long)t * B + b) * ld + off + c;
    float4 v;
    if (c + 3 < hd && ((reinterpret_cast<uintptr_t>(s)) & 15) == 0) v = *reinterpret_cast<const float4*>(s);
    else v = make_float4(c < hd ? s[0] : 0.f, c + 1 < hd ? s[1] : 0.f, c + 2 < hd ? s[2] : 0.f, c + 3 < hd ? s[3] : 0.f);
    v.x *= mul; v.y *= mul; v.z *= mul; v.w *= mul;
    *reinterpret_cast<float4*>(dst + t * WIDE_RS + (c >= WIDE_HS ? c + (WIDE_PS - WIDE_HS) : c)) = v;
  }
}

__device__ __forceinline__ float pair_sum(float x) { return x + __shfl_xor(x, 1, 64); }

template <bool FULL>  // FULL: head_dim 128 exactly (no tail guards: the loads and stores stay vectorised)
__global__ __launch_bounds__(2 * ATT_T) void attn_fwd_wide_kernel(const AttnP p) {
  constexpr int HS = WIDE_HS;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int T = p.T;
  float* Ks = sm;
  float* Vs = sm + T * WIDE_RS;
  const int b = blockIdx.x / p.nhead, head = blockIdx.x % p.nhead, off = head * (FULL ? WIDE_HD : p.hd);
  const int i = threadIdx.x >> 1, part = threadIdx.x & 1, po = part * WIDE_PS;
  const int nloc = FULL ? HS : min(HS, max(0, p.hd - part * HS));  // this lane's real features
  load_tile_wide(Ks, p.k, p.ld, T, p.B, b, off, 1.f, p.hd);
  load_tile_wide(Vs, p.v, p.ld, T, p.B, b, off, 1.f, p.hd);
  float q[HS], o[HS];
  const bool act = i < T;
  if (act) {
    const float* qs = p.q + ((long)i * p.B + b) * p.ld + off + part * HS;
#pragma unroll
    for (int c = 0; c < HS; ++c) { q[c] = (FULL || c < nloc) ? qs[c] * p.scale : 0.f; o[c] = 0.f; }
  } else {
#pragma unroll
    for (int c = 0; c < HS; ++c) { q[c] = 0.f; o[c] = 0.f; }
  }
  __syncthreads();
  float m = -INFINITY, l = 0.f;
  const uint64_t gbase = (((uint64_t)(p.col_offset + b) * p.nhead + head) * T + i) * (uint64_t)T;
  const int wave_last = min(T - 1, (int)(threadIdx.x | 63) >> 1);
  for (int j = 0; j <= wave_last; ++j) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < HS; c += 4) {
      const float4 kk = *reinterpret_cast<const float4*>(Ks + j * WIDE_RS + po + c);
      s += q[c] * kk.x + q[c + 1] * kk.y + q[c + 2] * kk.z + q[c + 3] * kk.w;
    }
    s = pair_sum(s);  // every lane of the wave takes part: the exchange sits outside the causal branch
    if (act && j <= i) {
      const float mn = fmaxf(m, s);
      const float corr = __expf(m - mn), pe = __expf(s - mn);
      l = l * corr + pe;
      const float pd = p.drop ? pe * keep_at(p, gbase + j) : pe;
#pragma unroll
      for (int c = 0; c < HS; c += 4) {
        const float4 vv = *reinterpret_cast<const float4*>(Vs + j * WIDE_RS + po + c);
        o[c] = o[c] * corr + pd * vv.x; o[c + 1] = o[c + 1] * corr + pd * vv.y;
        o[c + 2] = o[c + 2] * corr + pd * vv.z; o[c + 3] = o[c + 3] * corr + pd * vv.w;
      }
      m = mn;
    }
  }
  if (act) {
    const float inv = 1.f / l;
    float* os = p.out + ((long)i * p.B + b) * ((long)p.nhead * (FULL ? WIDE_HD : p.hd)) + off + part * HS;
#pragma unroll
    for (int c = 0; c < HS; ++c)
      if (FULL || c < nloc) os[c] = o[c] * inv;
    if (p.lse && part == 0) p.lse[(long)blockIdx.x * T + i] = m + __logf(l);
  }
}

template <bool FULL>
__global__ __launch_bounds__(2 * ATT_T) void attn_bwd_wide_kernel(const AttnP p) {
  constexpr int HS = WIDE_HS;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int T = p.T;
  float* A = sm;                     // K, then Q*scale
  float* Bf = sm + T * WIDE_RS;      // V, then dO
  float* lse_s = sm + 2 * T * WIDE_RS;
  float* del_s = lse_s + T;
  const int b = blockIdx.x / p.nhead, head = blockIdx.x % p.nhead, off = head * (FULL ? WIDE_HD : p.hd);
  const int i = threadIdx.x >> 1, part = threadIdx.x & 1, po = part * WIDE_PS;
  const int nloc = FULL ? HS : min(HS, max(0, p.hd - part * HS));  // this lane's real features
  const bool act = i < T;
  const long dmodel = (long)p.nhead * (FULL ? WIDE_HD : p.hd);
  const uint64_t bh = (uint64_t)(p.col_offset + b) * p.nhead + head;

  // ---------------- phase 1: lane pair = query row -> dQ
  load_tile_wide(A, p.k, p.ld, T, p.B, b, off, 1.f, p.hd);
  load_tile_wide(Bf, p.v, p.ld, T, p.B, b, off, 1.f, p.hd);
  {
    float q[HS], dO[HS], dq[HS];
    float lse = 0.f, delta = 0.f;
    if (act) {
      const float* qs = p.q + ((long)i * p.B + b) * p.ld + off + part * HS;
      const float* ds = p.dout + ((long)i * p.B + b) * dmodel + off + part * HS;
      const float* os = p.o_in + ((long)i * p.B + b) * dmodel + off + part * HS;
#pragma unroll
      for (int c = 0; c < HS; ++c) {
        const bool in = FULL || c < nloc;
        q[c] = in ? qs[c] * p.scale : 0.f; dO[c] = in ? ds[c] : 0.f; dq[c] = 0.f;
        if (in) delta += ds[c] * os[c];
      }
      lse = p.lse[(long)blockIdx.x * T + i];
    } else {
#pragma unroll
      for (int c = 0; c < HS; ++c) { q[c] = 0.f; dO[c] = 0.f; dq[c] = 0.f; }
    }
    delta = pair_sum(delta);
    if (act && part == 0) { lse_s[i] = lse; del_s[i] = delta; }
    __syncthreads();
    const int wave_last = min(T - 1, (int)(threadIdx.x | 63) >> 1);
    for (int j = 0; j <= wave_last; ++j) {
      float s = 0.f, dpd = 0.f;
#pragma unroll
      for (int c = 0; c < HS; c += 4) {
        const float4 kk = *reinterpret_cast<const float4*>(A + j * WIDE_RS + po + c);
        const float4 vv = *reinterpret_cast<const float4*>(Bf + j * WIDE_RS + po + c);
        s += q[c] * kk.x + q[c + 1] * kk.y + q[c + 2] * kk.z + q[c + 3] * kk.w;
        dpd += dO[c] * vv.x + dO[c + 1] * vv.y + dO[c + 2] * vv.z + dO[c + 3] * vv.w;
      }
      s = pair_sum(s);
      dpd = pair_sum(dpd);
      if (act && j <= i) {
        const float pr = __expf(s - lse);
        const float kp = p.drop ? keep_at(p, (bh * T + i) * (uint64_t)T + j) : 1.f;
        const float dS = pr * (dpd * kp - delta);
#pragma unroll
        for (int c = 0; c < HS; c += 4) {
          const float4 kk = *reinterpret_cast<const float4*>(A + j * WIDE_RS + po + c);
          dq[c] += dS * kk.x; dq[c + 1] += dS * kk.y; dq[c + 2] += dS * kk.z; dq[c + 3] += dS * kk.w;
        }
      }
    }
    if (act) {
      float* o = p.dq + ((long)i * p.B + b) * p.ldd + off + part * HS;
#pragma unroll
      for (int c = 0; c < HS; ++c)
        if (FULL || c < nloc) o[c] = dq[c] * p.scale;
    }
  }
  __syncthreads();
  // ---------------- phase 2: lane pair = key row -> dK, dV
  load_tile_wide(A, p.q, p.ld, T, p.B, b, off, p.scale, p.hd);
  load_tile_wide(Bf, p.dout, dmodel, T, p.B, b, off, 1.f, p.hd);
  float k[HS], v[HS];
  const int j = i;
  if (act) {
    const float* ks = p.k + ((long)j * p.B + b) * p.ld + off + part * HS;
    const float* vs = p.v + ((long)j * p.B + b) * p.ld + off + part * HS;
#pragma unroll
    for (int c = 0; c < HS; ++c) { k[c] = (FULL || c < nloc) ? ks[c] : 0.f; v[c] = (FULL || c < nloc) ? vs[c] : 0.f; }
  } else {
#pragma unroll
    for (int c = 0; c < HS; ++c) { k[c] = 0.f; v[c] = 0.f; }
  }
  __syncthreads();
  constexpr int HH = HS / 2;  // dk / dv in two passes of 32 features: the register budget of the head_dim-64 kernel
  const int wave_first = (int)(threadIdx.x & ~63) >> 1;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    float dk[HH], dv[HH];
#pragma unroll
    for (int c = 0; c < HH; ++c) { dk[c] = 0.f; dv[c] = 0.f; }
    for (int qi = wave_first; qi < T; ++qi) {
      float s = 0.f, dpd = 0.f;
#pragma unroll
      for (int c = 0; c < HS; c += 4) {
        const float4 qq = *reinterpret_cast<const float4*>(A + qi * WIDE_RS + po + c);
        const float4 dd = *reinterpret_cast<const float4*>(Bf + qi * WIDE_RS + po + c);
        s += qq.x * k[c] + qq.y * k[c + 1] + qq.z * k[c + 2] + qq.w * k[c + 3];
        dpd += dd.x * v[c] + dd.y * v[c + 1] + dd.z * v[c + 2] + dd.w * v[c + 3];
      }
      s = pair_sum(s);
      dpd = pair_sum(dpd);
      if (act && j <= qi) {
        const float pr = __expf(s - lse_s[qi]);
        const float kp = p.drop ? keep_at(p, (bh * T + qi) * (uint64_t)T + j) : 1.f;
        const float dS = pr * (dpd * kp - del_s[qi]);
        const float pd = pr * kp;
#pragma unroll
        for (int c = 0; c < HH; c += 4) {
          const float4 qq = *reinterpret_cast<const float4*>(A + qi * WIDE_RS + po + half * HH + c);
          const float4 dd = *reinterpret_cast<const float4*>(Bf + qi * WIDE_RS + po + half * HH + c);
          dk[c] += dS * qq.x; dk[c + 1] += dS * qq.y; dk[c + 2] += dS * qq.z; dk[c + 3] += dS * qq.w;
          dv[c] += pd * dd.x; dv[c + 1] += pd * dd.y; dv[c + 2] += pd * dd.z; dv[c + 3] += pd * dd.w;
        }
      }
    }
    if (act) {
      float* ok = p.dk + ((long)j * p.B + b) * p.ldd + off + part * HS + half * HH;
      float* ov = p.dv + ((long)j * p.B + b) * p.ldd + off + part * HS + half * HH;
#pragma unroll
      for (int c = 0; c < HH; ++c)
        if (FULL || half * HH + c < nloc) { ok[c] = dk[c]; ov[c] = dv[c]; }
    }
  }
}


// ------------------------------------------------------------------ generic fallback: any head_dim <= 512, any T
// One wave per query (forward, dQ) or per key (dK/dV): the lanes share the head's features (feature f = lane + 64 j),
// every score is a wave reduction, softmax is online.  No LDS, no tiling: a correctness path for the shapes the
// tiled kernels do not take (train.py's own defaults give head_dim 100: --emsize 200 --nhead 2, train.py:36-44),
// not a fast one.  The recipes' head_dim 64 never comes here.
constexpr int GEN_MAXJ = 8;  // features per lane: head_dim <= 512

__global__ __launch_bounds__(64) void attn_fwd_generic_kernel(const AttnP p, int hd) {
  const int q = blockIdx.x, bhl = blockIdx.y, b = bhl / p.nhead, head = bhl % p.nhead, off = head * hd;
  const int lane = threadIdx.x, T = p.T;
  const uint64_t bh = (uint64_t)(p.col_offset + b) * p.nhead + head;
  const int nj = (hd + 63) >> 6;
  float qv[GEN_MAXJ], acc[GEN_MAXJ];
#pragma unroll
  for (int j = 0; j < GEN_MAXJ; ++j) {
    const int f = lane + 64 * j;
    qv[j] = (j < nj && f < hd) ? p.q[((long)q * p.B + b) * p.ld + off + f] * p.scale : 0.f;
    acc[j] = 0.f;
  }
  float m = -INFINITY, l = 0.f;
  for (int key = 0; key <= q; ++key) {
    const float* kr = p.k + ((long)key * p.B + b) * p.ld + off;
    const float* vr = p.v + ((long)key * p.B + b) * p.ld + off;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < GEN_MAXJ; ++j) {
      const int f = lane + 64 * j;
      if (j < nj && f < hd) s += qv[j] * kr[f];
    }
    s = wave_sum(s);
    const float mn = fmaxf(m, s), alpha = __expf(m - mn), e = __expf(s - mn);
    l = l * alpha + e;
    m = mn;
    const float w = e * (p.drop ? keep_at(p, (bh * T + q) * (uint64_t)T + key) : 1.f);
#pragma unroll
    for (int j = 0; j < GEN_MAXJ; ++j) {
      const int f = lane + 64 * j;
      if (j < nj && f < hd) acc[j] = acc[j] * alpha + w * vr[f];
    }
  }
  const float inv = 1.f / l;
#pragma unroll
  for (int j = 0; j < GEN_MAXJ; ++j) {
    const int f = lane + 64 * j;
    if (j < nj && f < hd) p.out[((long)q * p.B + b) * ((long)p.nhead * hd) + off + f] = acc[j] * inv;
  }
  if (p.lse && lane == 0) p.lse[(long)bhl * T + q] = m + __logf(l);
}

__global__ __launch_bounds__(64) void attn_bwd_dq_generic_kernel(const AttnP p, int hd) {
  const int q = blockIdx.x, bhl = blockIdx.y, b = bhl / p.nhead, head = bhl % p.nhead, off = head * hd;
  const int lane = threadIdx.x, T = p.T;
  const long dmodel = (long)p.nhead * hd;
  const uint64_t bh = (uint64_t)(p.col_offset + b) * p.nhead + head;
  const int nj = (hd + 63) >> 6;
  float qv[GEN_MAXJ], dov[GEN_MAXJ], acc[GEN_MAXJ];
  float delta = 0.f;
#pragma unroll
  for (int j = 0; j < GEN_MAXJ; ++j) {
    const int f = lane + 64 * j;
    const bool ok = j < nj && f < hd;
    qv[j] = ok ? p.q[((long)q * p.B + b) * p.ld + off + f] * p.scale : 0.f;
    dov[j] = ok ? p.dout[((long)q * p.B + b) * dmodel + off + f] : 0.f;
    if (ok) delta += dov[j] * p.o_in[((long)q * p.B + b) * dmodel + off + f];
    acc[j] = 0.f;
  }
  delta = wave_sum(delta);
  const float lse = p.lse[(long)bhl * T + q];
  for (int key = 0; key <= q; ++key) {
    const float* kr = p.k + ((long)key * p.B + b) * p.ld + off;
    const float* vr = p.v + ((long)key * p.B + b) * p.ld + off;
    float s = 0.f, dp = 0.f;
#pragma unroll
    for (int j = 0; j < GEN_MAXJ; ++j) {
      const int f = lane + 64 * j;
      if (j < nj && f < hd) { s += qv[j] * kr[f]; dp += dov[j] * vr[f]; }
    }
    s = wave_sum(s);
    dp = wave_sum(dp);
    const float keep = p.drop ? keep_at(p, (bh * T + q) * (uint64_t)T + key) : 1.f;
    const float ds = __expf(s - lse) * (dp * keep - delta);
#pragma unroll
    for (int j = 0; j < GEN_MAXJ; ++j) {
      const int f = lane + 64 * j;
      if (j < nj && f < hd) acc[j] += ds * kr[f];
    }
  }
#pragma unroll
  for (int j = 0; j < GEN_MAXJ; ++j) {
    const int f = lane + 64 * j;
    if (j < nj && f < hd) p.dq[((long)q * p.B + b) * p.ldd + off + f] = acc[j] * p.scale;
  }
}

__global__ __launch_bounds__(64) void attn_bwd_dkv_generic_kernel(const AttnP p, int hd) {
  const int key = blockIdx.x, bhl = blockIdx.y, b = bhl / p.nhead, head = bhl % p.nhead, off = head * hd;
  const int lane = threadIdx.x, T = p.T;
  const long dmodel = (long)p.nhead * hd;
  const uint64_t bh = (uint64_t)(p.col_offset + b) * p.nhead + head;
  const int nj = (hd + 63) >> 6;
  float kv[GEN_MAXJ], vv[GEN_MAXJ], dk[GEN_MAXJ], dv[GEN_MAXJ];
#pragma unroll
  for (int j = 0; j < GEN_MAXJ; ++j) {
    const int f = lane + 64 * j;
    const bool ok = j < nj && f < hd;
    kv[j] = ok ? p.k[((long)key * p.B + b) * p.ld + off + f] : 0.f;
    vv[j] = ok ? p.v[((long)key * p.B + b) * p.ld + off + f] : 0.f;
    dk[j] = dv[j] = 0.f;
  }
  for (int q = key; q < T; ++q) {
    const float* qr = p.q + ((long)q * p.B + b) * p.ld + off;
    const float* dor = p.dout + ((long)q * p.B + b) * dmodel + off;
    const float* orow = p.o_in + ((long)q * p.B + b) * dmodel + off;
    float s = 0.f, dp = 0.f, delta = 0.f;
    float qs[GEN_MAXJ], dof[GEN_MAXJ];
#pragma unroll
    for (int j = 0; j < GEN_MAXJ; ++j) {
      const int f = lane + 64 * j;
      const bool ok = j < nj && f < hd;
      qs[j] = ok ? qr[f] * p.scale : 0.f;
      dof[j] = ok ? dor[f] : 0.f;
      s += qs[j] * kv[j];
      dp += dof[j] * vv[j];
      if (ok) delta += dof[j] * orow[f];
    }
    s = wave_sum(s);
    dp = wave_sum(dp);
    delta = wave_sum(delta);
    const float keep = p.drop ? keep_at(p, (bh * T + q) * (uint64_t)T + key) : 1.f;
    const float pr = __expf(s - p.lse[(long)bhl * T + q]);
    const float pk = pr * keep, ds = pr * (dp * keep - delta);
#pragma unroll
    for (int j = 0; j < GEN_MAXJ; ++j) {
      dv[j] += pk * dof[j];
      dk[j] += ds * qs[j];
    }
  }
#pragma unroll
  for (int j = 0; j < GEN_MAXJ; ++j) {
    const int f = lane + 64 * j;
    if (j < nj && f < hd) {
      p.dk[((long)key * p.B + b) * p.ldd + off + f] = dk[j];
      p.dv[((long)key * p.B + b) * p.ldd + off + f] = dv[j];
    }
  }
}

static bool own_kernel(int head_dim) { return head_dim == 4 || head_dim == 8 || head_dim == 16 || head_dim == 32 || head_dim == 64; }
static bool tiled_ok(int T, int head_dim) {  // the LDS-tiled VALU kernels: T <= 128, head_dim a power of two up to 64 -- or anything up to 128 on lane pairs
  return T <= ATT_T && head_dim <= WIDE_HD;
}

static int fill(AttnP& p, int T, int B, int nhead, int head_dim, float pdrop, const blm_rng* rng, int col_offset,
                const char* who) {
  if (T < 0 || B < 0 || nhead <= 0) return blm_fail(BLM_ERR_INVALID, "%s: bad shape", who);
  if (head_dim <= 0 || head_dim > 64 * GEN_MAXJ)
    return blm_fail(BLM_ERR_UNSUPPORTED, "%s: head_dim %d not in 1..%d", who, head_dim, 64 * GEN_MAXJ);
  if (pdrop > 0.f && !rng) return blm_fail(BLM_ERR_INVALID, "%s: dropout needs rng", who);
  p.T = T; p.B = B; p.nhead = nhead; p.hd = head_dim;
  p.scale = 1.0f / sqrtf((float)head_dim);
  p.drop = pdrop > 0.f;
  if (p.drop) p.rng = *rng;
  const double t = (double)pdrop * 4294967296.0;
  p.thr = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
  p.inv_keep = pdrop < 1.f ? 1.f / (1.f - pdrop) : 0.f;
  p.col_offset = col_offset;
  return BLM_OK;
}

}  // namespace blm

using namespace blm;

// head_dim 64 runs on the matrix cores (attention_mfma.hip); other head sizes on the VALU kernels above
int blm_attn_fwd_mfma(const float* q, const float* k, const float* v, int64_t ld, float* out, float* lse, int T, int B,
                      int nhead, float pdrop, const blm_rng* rng, int col_offset, hipStream_t st);
int blm_attn_bwd_mfma(const float* q, const float* k, const float* v, int64_t ld, const float* out, const float* dout,
                      const float* lse, float* dq, float* dk, float* dv, int64_t ldd, int T, int B, int nhead,
                      float pdrop, const blm_rng* rng, int col_offset, float* ws, hipStream_t st);
int64_t blm_attn_bwd_mfma_ws_floats(int T, int B, int nhead);
static bool use_mfma(int head_dim) {  // option "attn_valu" = 1: the vector-ALU kernels at head_dim 64 too (cross-check of the MFMA ones)
  return head_dim == 64 && !blm::option(blm::OPT_ATTN_VALU);
}

#define DISPATCH_HD(KERN, LDS)                                                                                   \
  switch (head_dim) {                                                                                            \
    case 4: hipLaunchKernelGGL(KERN<4>, dim3(B * nhead), dim3(ATT_T), LDS, st, p); break;                         \
    case 8: hipLaunchKernelGGL(KERN<8>, dim3(B * nhead), dim3(ATT_T), LDS, st, p); break;                         \
    case 16: hipLaunchKernelGGL(KERN<16>, dim3(B * nhead), dim3(ATT_T), LDS, st, p); break;                       \
    case 32: hipLaunchKernelGGL(KERN<32>, dim3(B * nhead), dim3(ATT_T), LDS, st, p); break;                       \
    default: hipLaunchKernelGGL(KERN<64>, dim3(B * nhead), dim3(ATT_T), LDS, st, p); break;                       \
  }

// the vector-ALU kernels: LDS-tiled for T <= 128 and power-of-two heads, one wave per query otherwise
static int launch_fwd_valu(const AttnP& p, int T, int B, int nhead, int head_dim, hipStream_t st) {
  if (!tiled_ok(T, head_dim)) {  // any other head size / longer sequences: one wave per query
    hipLaunchKernelGGL(attn_fwd_generic_kernel, dim3(T, B * nhead), dim3(64), 0, st, p, head_dim);
    BLM_HIP(hipGetLastError());
    return BLM_OK;
  }
  if (!own_kernel(head_dim)) {  // two lanes per row; up to 136 KB of LDS
    const size_t ldsw = (size_t)2 * T * WIDE_RS * sizeof(float);
    const bool full = head_dim == WIDE_HD;
    auto kern = full ? attn_fwd_wide_kernel<true> : attn_fwd_wide_kernel<false>;
    static bool attr_f[2] = {false, false};  // benign race (idempotent)
    if (!attr_f[full]) {
      BLM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)((size_t)2 * ATT_T * WIDE_RS * sizeof(float))));
      attr_f[full] = true;
    }
    hipLaunchKernelGGL(kern, dim3(B * nhead), dim3(2 * ATT_T), ldsw, st, p);
    BLM_HIP(hipGetLastError());
    return BLM_OK;
  }
  const size_t lds = (size_t)2 * T * head_dim * sizeof(float);
  DISPATCH_HD(attn_fwd_kernel, lds)
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

static int launch_bwd_valu(const AttnP& p, int T, int B, int nhead, int head_dim, hipStream_t st) {
  if (!tiled_ok(T, head_dim)) {
    hipLaunchKernelGGL(attn_bwd_dq_generic_kernel, dim3(T, B * nhead), dim3(64), 0, st, p, head_dim);
    BLM_HIP(hipGetLastError());
    hipLaunchKernelGGL(attn_bwd_dkv_generic_kernel, dim3(T, B * nhead), dim3(64), 0, st, p, head_dim);
    BLM_HIP(hipGetLastError());
    return BLM_OK;
  }
  if (!own_kernel(head_dim)) {
    const size_t ldsw = ((size_t)2 * T * WIDE_RS + 2 * T) * sizeof(float);
    const bool full = head_dim == WIDE_HD;
    auto kern = full ? attn_bwd_wide_kernel<true> : attn_bwd_wide_kernel<false>;
    static bool attr_b[2] = {false, false};
    if (!attr_b[full]) {
      BLM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(((size_t)2 * ATT_T * WIDE_RS + 2 * ATT_T) * sizeof(float))));
      attr_b[full] = true;
    }
    hipLaunchKernelGGL(kern, dim3(B * nhead), dim3(2 * ATT_T), ldsw, st, p);
    BLM_HIP(hipGetLastError());
    return BLM_OK;
  }
  const size_t lds = ((size_t)2 * T * head_dim + 2 * T) * sizeof(float);
  DISPATCH_HD(attn_bwd_kernel, lds)
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int blm_attn_fwd(const float* q, const float* k, const float* v, int64_t ld_qkv, float* out, float* lse,
                            int T, int B, int nhead, int head_dim, float pdrop, const blm_rng* rng, int col_offset,
                            int global_cols, void* stream) {
  (void)global_cols;
  if (!q || !k || !v || !out) return blm_fail(BLM_ERR_INVALID, "blm_attn_fwd: null operand");
  AttnP p{};
  int rc = fill(p, T, B, nhead, head_dim, pdrop, rng, col_offset, "blm_attn_fwd");
  if (rc) return rc;
  if (ld_qkv < (int64_t)nhead * head_dim) return blm_fail(BLM_ERR_INVALID, "blm_attn_fwd: ld_qkv too small");
  if ((long)T * B == 0) return BLM_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (use_mfma(head_dim)) return blm_attn_fwd_mfma(q, k, v, ld_qkv, out, lse, T, B, nhead, pdrop, rng, col_offset, st);
  p.q = q; p.k = k; p.v = v; p.ld = ld_qkv; p.out = out; p.lse = lse;
  return launch_fwd_valu(p, T, B, nhead, head_dim, st);
}

// The dropout factors of the probabilities handed over instead of generated: keep (global_cols * nhead, T, T), 0 or 1 / (1 - p), head
// index (col_offset + b) * nhead + head as in the reference's (B * h, T, T) probabilities (model.py:905-914).  A parity path
// (NoiseState.source "torch": the mask torch's CPU dropout drew); always the vector-ALU kernels.
extern "C" int blm_attn_fwd_keep(const float* q, const float* k, const float* v, int64_t ld_qkv, float* out, float* lse, int T, int B,
                                 int nhead, int head_dim, const float* keep, int col_offset, int global_cols, void* stream) {
  if (!q || !k || !v || !out || !keep || col_offset < 0 || B < 0 || (long)global_cols < (long)col_offset + B)
    return blm_fail(BLM_ERR_INVALID, "blm_attn_fwd_keep: bad arguments");
  AttnP p{};
  int rc = fill(p, T, B, nhead, head_dim, 0.f, nullptr, col_offset, "blm_attn_fwd_keep");
  if (rc) return rc;
  if (ld_qkv < (int64_t)nhead * head_dim) return blm_fail(BLM_ERR_INVALID, "blm_attn_fwd_keep: ld_qkv too small");
  if (!blm::extents_ok({global_cols, nhead, T, T})) return blm_fail(BLM_ERR_INVALID, "blm_attn_fwd_keep: extents");
  if ((long)T * B == 0) return BLM_OK;
  p.drop = true; p.keep = keep;
  p.q = q; p.k = k; p.v = v; p.ld = ld_qkv; p.out = out; p.lse = lse;
  return launch_fwd_valu(p, T, B, nhead, head_dim, static_cast<hipStream_t>(stream));
}

int blm_attn_fwd_rows_mfma(const float* q, const float* k, const float* v, int64_t ld, float* out, const int* rowmap, int T, int B,
                           int nhead, hipStream_t st);

extern "C" int blm_attn_fwd_rows(const float* q, const float* k, const float* v, int64_t ld_qkv, float* out, const int32_t* rowmap,
                                 int T, int B, int nhead, int head_dim, void* stream) {
  if (!q || !k || !v || !out || !rowmap || T < 0 || B < 0 || nhead <= 0) return blm_fail(BLM_ERR_INVALID, "blm_attn_fwd_rows: bad arguments");
  if (ld_qkv < (int64_t)nhead * head_dim) return blm_fail(BLM_ERR_INVALID, "blm_attn_fwd_rows: ld_qkv too small");
  if ((long)T * B == 0) return BLM_OK;
  if (!use_mfma(head_dim)) return blm_fail(BLM_ERR_UNSUPPORTED, "blm_attn_fwd_rows: head_dim 64 only");
  return blm_attn_fwd_rows_mfma(q, k, v, ld_qkv, out, rowmap, T, B, nhead, static_cast<hipStream_t>(stream));
}

extern "C" int64_t blm_attn_bwd_ws_floats(int T, int B, int nhead, int head_dim) {
  if (T <= 0 || B <= 0 || nhead <= 0 || !use_mfma(head_dim)) return 0;
  return blm_attn_bwd_mfma_ws_floats(T, B, nhead);
}

extern "C" int blm_attn_bwd(const float* q, const float* k, const float* v, int64_t ld_qkv, const float* out,
                            const float* dout, const float* lse, float* dq, float* dk, float* dv, int64_t ld_dqkv, int T,
                            int B, int nhead, int head_dim, float pdrop, const blm_rng* rng, int col_offset,
                            int global_cols, void* stream) {
  return blm_attn_bwd_ws(q, k, v, ld_qkv, out, dout, lse, dq, dk, dv, ld_dqkv, T, B, nhead, head_dim, pdrop, rng, col_offset,
                         global_cols, nullptr, 0, stream);
}

extern "C" int blm_attn_bwd_ws(const float* q, const float* k, const float* v, int64_t ld_qkv, const float* out,
                               const float* dout, const float* lse, float* dq, float* dk, float* dv, int64_t ld_dqkv, int T,
                               int B, int nhead, int head_dim, float pdrop, const blm_rng* rng, int col_offset,
                               int global_cols, float* ws, int64_t ws_floats, void* stream) {
  (void)global_cols;
  if (!q || !k || !v || !out || !dout || !lse || !dq || !dk || !dv) return blm_fail(BLM_ERR_INVALID, "blm_attn_bwd: null operand");
  AttnP p{};
  int rc = fill(p, T, B, nhead, head_dim, pdrop, rng, col_offset, "blm_attn_bwd");
  if (rc) return rc;
  if (ld_qkv < (int64_t)nhead * head_dim || ld_dqkv < (int64_t)nhead * head_dim)
    return blm_fail(BLM_ERR_INVALID, "blm_attn_bwd: leading dimension too small");
  if ((long)T * B == 0) return BLM_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (use_mfma(head_dim)) {
    if (ws && ws_floats < blm_attn_bwd_mfma_ws_floats(T, B, nhead)) return blm_fail(BLM_ERR_INVALID, "blm_attn_bwd_ws: workspace smaller than blm_attn_bwd_ws_floats()");
    return blm_attn_bwd_mfma(q, k, v, ld_qkv, out, dout, lse, dq, dk, dv, ld_dqkv, T, B, nhead, pdrop, rng, col_offset, ws, st);
  }
  p.q = q; p.k = k; p.v = v; p.ld = ld_qkv; p.o_in = out; p.dout = dout; p.lse = const_cast<float*>(lse);
  p.dq = dq; p.dk = dk; p.dv = dv; p.ldd = ld_dqkv;
  return launch_bwd_valu(p, T, B, nhead, head_dim, st);
}

extern "C" int blm_attn_bwd_keep(const float* q, const float* k, const float* v, int64_t ld_qkv, const float* out, const float* dout,
                                 const float* lse, float* dq, float* dk, float* dv, int64_t ld_dqkv, int T, int B, int nhead,
                                 int head_dim, const float* keep, int col_offset, int global_cols, void* stream) {
  if (!q || !k || !v || !out || !dout || !lse || !dq || !dk || !dv || !keep || col_offset < 0 || B < 0 || (long)global_cols < (long)col_offset + B)
    return blm_fail(BLM_ERR_INVALID, "blm_attn_bwd_keep: bad arguments");
  AttnP p{};
  int rc = fill(p, T, B, nhead, head_dim, 0.f, nullptr, col_offset, "blm_attn_bwd_keep");
  if (rc) return rc;
  if (ld_qkv < (int64_t)nhead * head_dim || ld_dqkv < (int64_t)nhead * head_dim)
    return blm_fail(BLM_ERR_INVALID, "blm_attn_bwd_keep: leading dimension too small");
  if (!blm::extents_ok({global_cols, nhead, T, T})) return blm_fail(BLM_ERR_INVALID, "blm_attn_bwd_keep: extents");
  if ((long)T * B == 0) return BLM_OK;
  p.drop = true; p.keep = keep;
  p.q = q; p.k = k; p.v = v; p.ld = ld_qkv; p.o_in = out; p.dout = dout; p.lse = const_cast<float*>(lse);
  p.dq = dq; p.dk = dk; p.dv = dv; p.ldd = ld_dqkv;
  return launch_bwd_valu(p, T, B, nhead, head_dim, static_cast<hipStream_t>(stream));
}
