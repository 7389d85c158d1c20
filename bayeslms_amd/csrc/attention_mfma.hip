// Causal self-attention on the f32 matrix cores (v_mfma_f32_32x32x2_f32), head_dim 64, T <= 128:
// forward, dQ and dK/dV kernels, flash-style (no T x T matrix in HBM, probabilities recomputed in
// backward from the saved log-sum-exp).
//
// Design (one workgroup of 4 waves per (batch column, head); K,V or Q,dO rows of the head sit in
// LDS once, row-major with a 65-float stride so the same copy serves both operand shapes:
// "lane = row, k = 2s + half" and "lane = feature, row fixed" are both conflict-free ds_read_b32):
//  * The score tile is computed TRANSPOSED, S^T[key][query] = K Q^T, so the MFMA result has one
//    query per lane and its keys in the 16 accumulator registers.  Row max / row sum of the softmax
//    are then register reductions plus ONE cross-half shuffle, and the probabilities are already in
//    the B-operand layout of the next product (O^T = V^T P^T, dQ^T = K^T dS^T): P never leaves the
//    register file (cdna_hip_programming.md section 3, "accumulator tile as the next MFMA's operand",
//    here for the one-float-per-lane f32 operand map).
//  * dK/dV use the other orientation (lane = key, queries in registers) for the same reason.
//  * One 32-query (32-key) tile per wave, the longest causal tile on the first wave: the kernels are
//    latency- not MFMA-bound, so 4 unbalanced waves beat 2 waves with the balanced tile pairs {w, 3-w}.
//  * Dropout on the probabilities: Philox bits keyed by the global element ((b*nhead+h)*T+q)*T+key;
//    4 consecutive keys = one Philox block = 4 accumulator registers of a lane (query orientation)
//    or a quad exchange (key orientation).
//
// Replaces model.py:889-920 / :990-1011 and their autograd.
#include <cstdlib>

#include "blm_device.h"
#include "blm_host.h"

namespace blm {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int AT = 128;   // max sequence length
constexpr int HD = 64;    // head dim of the MFMA path
constexpr int LS = 65;    // LDS row stride (floats)

struct AttnM {
  const float *q, *k, *v;
  long ld;
  float* out;
  float* lse;
  const float *o_in, *dout;
  float *dq, *dk, *dv;
  float* ds;  // optional (B*nhead, T, T) workspace: the dK/dV kernel leaves dS there and dQ = dS K needs no recomputation
  long ldd;
  int T, B, nhead;
  float scale;
  blm_rng rng;
  uint32_t thr;
  float inv_keep;
  int col_offset;
  int drop;
  // optional (T * B): packed row of the token at padded position t * B + b, or -1 for padding -- q / k / v / out are then (R, ld)
  // matrices of the REAL tokens only (the n-best scorer's layout, ops.packed_tokens); T <= 32 forward only
  const int* rowmap;
#ifdef BLM_ATTN_PROF
  long long* prof;  // tools/attn_prof.hip only: 8 wall-clock stamps (10 ns) per wave
#endif
};

#ifdef BLM_ATTN_PROF
__device__ long long* g_attn_stamps;  // per-thread pointer is awkward across helpers: stamps live in registers of lane 0
#define ATTN_STAMP(arr, i) do { if ((threadIdx.x & 63) == 0) (arr)[i] = wall_clock64(); } while (0)
#else
#define ATTN_STAMP(arr, i) do { } while (0)
#endif

// rows [0,T) x 64 floats of a (T,B,*) tensor -> dst[row*LS + c] (* mul); rows [T,128) zeroed.
// Split in an issue half (16 independent float4 loads per thread) and an LDS-write half so a kernel
// can put ALL of its prologue loads in flight before it waits on any of them: a 2-wave workgroup has
// nothing else to hide the memory latency behind (PMC: waves of the first version were parked on
// s_waitcnt for half of their lifetime).
template <int NT>
__device__ __forceinline__ void fetch_rows(float4 (&v)[2048 / NT], const float* src, long ld, int T, int B, int b, int off, int tid) {
  const bool al = ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)(ld * 4) | (uintptr_t)(off * 4)) & 15) == 0;
#pragma unroll
  for (int u = 0; u < 2048 / NT; ++u) {
    const int i = tid + NT * u, row = i >> 4, c = (i & 15) << 2;
    const float* s = src + ((long)min(row, T - 1) * B + b) * ld + off + c;
    if (al) v[u] = *reinterpret_cast<const float4*>(s);
    else v[u] = make_float4(s[0], s[1], s[2], s[3]);
  }
}
template <int NT>
__device__ __forceinline__ void put_rows(float* dst, const float4 (&v)[2048 / NT], int T, float mul, int tid) {
#pragma unroll
  for (int u = 0; u < 2048 / NT; ++u) {
    const int i = tid + NT * u, row = i >> 4, c = (i & 15) << 2;
    const float m = row < T ? mul : 0.f;
    float* d = dst + row * LS + c;
    d[0] = v[u].x * m; d[1] = v[u].y * m; d[2] = v[u].z * m; d[3] = v[u].w * m;
  }
}
// the 32 B-operand values of one lane.  v_mfma_f32_32x32x2_f32 only needs A and B to agree on which
// feature a (step, lane half) pair means, so half 0 takes features 0..31 and half 1 features 32..63 of
// the lane's row: one contiguous 128-B line per lane, 8 float4 loads (base = row + 32 * half).
__device__ __forceinline__ void fetch_op(float (&r)[32], const float* base) {
  if ((reinterpret_cast<uintptr_t>(base) & 15) == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float4 v = reinterpret_cast<const float4*>(base)[j];
      r[4 * j] = v.x; r[4 * j + 1] = v.y; r[4 * j + 2] = v.z; r[4 * j + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int s = 0; s < 32; ++s) r[s] = base[s];
  }
}

// MFMA row index held in accumulator register r by a lane of half h
__device__ __forceinline__ constexpr int mrow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// keep factors of the 4 consecutive columns c0..c0+3 (c0 % 4 == 0) of probability row `grow`
__device__ __forceinline__ void keep_row4(const AttnM& p, uint64_t grow, int c0, float (&k)[4]) {
  const uint64_t g = grow * (uint64_t)p.T + (uint64_t)c0;
  if ((p.T & 3) == 0) {
    const u32x4 u = philox4x32_10((uint32_t)(g >> 2), (uint32_t)(g >> 34), p.rng.stream, p.rng.step,
                                         (uint32_t)p.rng.seed, (uint32_t)(p.rng.seed >> 32));
    k[0] = u.x >= p.thr ? p.inv_keep : 0.f; k[1] = u.y >= p.thr ? p.inv_keep : 0.f;
    k[2] = u.z >= p.thr ? p.inv_keep : 0.f; k[3] = u.w >= p.thr ? p.inv_keep : 0.f;
  } else {  // rows do not start on a block boundary: per element
#pragma unroll
    for (int e = 0; e < 4; ++e) k[e] = philox_bits1_rolled(p.rng, g + e) >= p.thr ? p.inv_keep : 0.f;
  }
}

// S^T tile: acc[r] = sum_k X[row0 + (lane&31)][k] * breg[k-th]  with A from LDS rows, B from registers
__device__ __forceinline__ f32x16 tile_rows_x_regs(const float* X, int row0, const float (&breg)[32], int li, int lh) {
  f32x16 acc = (f32x16)(0.f);
  const float* a = X + (row0 + li) * LS + 32 * lh;
  // all 32 LDS operand reads are issued first (independent, one VGPR each) so the matrix pipe is not
  // stalled on a ds_read latency in front of every MFMA
  float av[32];
#pragma unroll
  for (int s = 0; s < 32; ++s) av[s] = a[s];
#pragma unroll
  for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], breg[s], acc, 0, 0, 0);
  return acc;
}

// two independent S^T-style tiles with their MFMA chains interleaved (a chain of dependent
// v_mfma_f32_32x32x2_f32 on one accumulator leaves issue slots empty between links)
__device__ __forceinline__ void tile2_rows_x_regs(f32x16& acc1, f32x16& acc2, const float* X1, const float* X2, int row0,
                                                  const float (&b1)[32], const float (&b2)[32], int li, int lh) {
  acc1 = (f32x16)(0.f);
  acc2 = (f32x16)(0.f);
  const float* a1 = X1 + (row0 + li) * LS + 32 * lh;
  const float* a2 = X2 + (row0 + li) * LS + 32 * lh;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float av1[16], av2[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) { av1[s] = a1[16 * h + s]; av2[s] = a2[16 * h + s]; }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[s], b1[16 * h + s], acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av2[s], b2[16 * h + s], acc2, 0, 0, 0);
    }
  }
}

// Y^T[d][lane col] += sum over the 32 rows of the register tile:  A = X[row0 + mrow(s,h)][32*dt + (lane&31)], B = regs
__device__ __forceinline__ void acc_xt_regs(f32x16 (&acc)[2], const float* X, int row0, const f32x16& breg, int li, int lh) {
  float a0[16], a1[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const float* a = X + (row0 + mrow(s, lh)) * LS + li;
    a0[s] = a[0];
    a1[s] = a[32];
  }
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], breg[s], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], breg[s], acc[1], 0, 0, 0);
  }
}

// write Y^T accumulators (features in registers, one row per lane) as row-major floats
__device__ __forceinline__ void store_t(float* dst_row, const f32x16 (&acc)[2], int lh, float mul) {
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float* d = dst_row + 32 * dt + 8 * g + 4 * lh;
      const float4 v = make_float4(acc[dt][4 * g] * mul, acc[dt][4 * g + 1] * mul, acc[dt][4 * g + 2] * mul, acc[dt][4 * g + 3] * mul);
      if ((reinterpret_cast<uintptr_t>(d) & 15) == 0) *reinterpret_cast<float4*>(d) = v;
      else { d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; }
    }
}

// ------------------------------------------------------------------ forward
__device__ __forceinline__ void attn_fwd_pass(const AttnM& p, const float* Ks, const float* Vs, int qt, float (&qreg)[32],
                                              int b, int off, uint64_t bh, int li, int lh, int bhid
#ifdef BLM_ATTN_PROF
                                              , long long (&stamps)[8]
#endif
                                              ) {
  const int T = p.T;
  const int q = 32 * qt + li;
  const bool qok = q < T;
#pragma unroll
  for (int s = 0; s < 32; ++s) qreg[s] *= p.scale;
  f32x16 st[4];
  float m = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    if (kt <= qt) {
      st[kt] = tile_rows_x_regs(Ks, 32 * kt, qreg, li, lh);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = 32 * kt + mrow(r, lh);
        if (key > q) st[kt][r] = -INFINITY;  // causal (also hides keys >= T for valid queries)
        m = fmaxf(m, st[kt][r]);
      }
    }
  }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  ATTN_STAMP(stamps, 3);
  float l = 0.f;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    if (kt <= qt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        st[kt][r] = __expf(st[kt][r] - m);
        l += st[kt][r];
      }
    }
  }
  l += __shfl_xor(l, 32, 64);
  const float inv = 1.f / l;
  ATTN_STAMP(stamps, 4);
  f32x16 ot[2] = {(f32x16)(0.f), (f32x16)(0.f)};
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    if (kt <= qt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float kp[4] = {1.f, 1.f, 1.f, 1.f};
        if (p.drop) keep_row4(p, bh * T + min(q, T - 1), 32 * kt + 8 * g + 4 * lh, kp);
#pragma unroll
        for (int e = 0; e < 4; ++e) st[kt][4 * g + e] *= inv * kp[e];
      }
      acc_xt_regs(ot, Vs, 32 * kt, st[kt], li, lh);
    }
  }
  ATTN_STAMP(stamps, 5);
  if (qok) {
    long orow = (long)q * p.B + b;
    if (p.rowmap) orow = p.rowmap[orow];  // packed rows: a padding query has no row to write
    if (orow >= 0) store_t(p.out + orow * ((long)p.nhead * HD) + off, ot, lh, 1.f);
    if (p.lse && lh == 0) p.lse[(long)bhid * T + q] = m + __logf(l);
  }
  ATTN_STAMP(stamps, 6);
}

// 4 waves, one query tile each: the kernel is bound by its load -> compute -> store latency chain, not
// by the matrix pipe (9 us of MFMA in 40), so twice the waves per workgroup (twice the loads in flight,
// half the serial work per wave) beat the balanced two-tiles-per-wave split of the causal triangle.
// HPW = heads per workgroup.  HPW = 2 (even B * nhead): 8 waves, the second head's waves take the query tiles in the
// opposite order, so the two waves that share a SIMD (w and w + 4) hold causal tiles {3 - w, w} = 5 tile-steps on every
// SIMD.  With one head per workgroup the two co-resident workgroups of a CU put BOTH 4-step waves on SIMD 0 and both
// 1-step waves on SIMD 3: 8 tile-steps of MFMA + softmax on the critical SIMD against 5 here.
template <int HPW>
__global__ __launch_bounds__(256 * HPW) void attn_fwd_mfma_kernel(const AttnM p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int hsel = threadIdx.x >> 8, tid = threadIdx.x & 255;
  const int bhid = blockIdx.x * HPW + hsel;
  float* Ks = sm + hsel * 2 * AT * LS;
  float* Vs = Ks + AT * LS;
  const int b = bhid / p.nhead, head = bhid % p.nhead, off = head * HD;
  const int T = p.T, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int ntile = (T + 31) >> 5;
  const int qt = hsel ? wave : 3 - wave;  // the longest tile on the first wave (second head: on the last)
  float qa[32];
#ifdef BLM_ATTN_PROF
  long long stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  ATTN_STAMP(stamps, 0);
  {
    float4 kk[8], vv[8];
    fetch_rows<256>(kk, p.k, p.ld, T, p.B, b, off, tid);
    fetch_rows<256>(vv, p.v, p.ld, T, p.B, b, off, tid);
    fetch_op(qa, p.q + ((long)min(32 * qt + li, T - 1) * p.B + b) * p.ld + off + 32 * lh);
    put_rows<256>(Ks, kk, T, 1.f, tid);
    put_rows<256>(Vs, vv, T, 1.f, tid);
  }
  ATTN_STAMP(stamps, 1);
  __syncthreads();
  ATTN_STAMP(stamps, 2);
  const uint64_t bh = (uint64_t)(p.col_offset + b) * p.nhead + head;
#ifdef BLM_ATTN_PROF
  if (qt < ntile) attn_fwd_pass(p, Ks, Vs, qt, qa, b, off, bh, li, lh, bhid, stamps);
  if (p.prof && lane == 0) {
    long long* o = p.prof + ((long)blockIdx.x * 4 * HPW + (threadIdx.x >> 6)) * 8;
    for (int i = 0; i < 8; ++i) o[i] = stamps[i];
    o[7] = qt;
  }
#else
  if (qt < ntile) attn_fwd_pass(p, Ks, Vs, qt, qa, b, off, bh, li, lh, bhid);
#endif
}

// ------------------------------------------------------------------ forward, T <= 32 (n-best rescoring: hypotheses of 5-30 tokens)
// One causal tile per head, so one WAVE per head: it loads its 32 K and 32 V rows (8 float4 per lane and matrix), keeps them in
// its own 16.6 KB of LDS and runs attn_fwd_pass on query tile 0.  Four heads per workgroup, two workgroups per CU: eight heads in
// flight per CU against two in the 128-row form above, whose other three waves per head have nothing to do at this length
// (packed scoring batches launch thousands of heads: 16 rounds of a latency chain).
__global__ __launch_bounds__(256) void attn_fwd_short_kernel(const AttnM p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const int bhid = blockIdx.x * 4 + wave, nbh = p.B * p.nhead;
  const int hb = min(bhid, nbh - 1);  // a surplus wave of the last workgroup repeats the last head's loads and stores nothing
  float* Ks = sm + wave * 2 * 32 * LS;
  float* Vs = Ks + 32 * LS;
  const int b = hb / p.nhead, head = hb % p.nhead, off = head * HD;
  const int T = p.T;
  float qa[32];
  {
    float4 kk[8], vv[8];
    const bool al = ((reinterpret_cast<uintptr_t>(p.k) | reinterpret_cast<uintptr_t>(p.v) | (uintptr_t)(p.ld * 4) | (uintptr_t)(off * 4)) & 15) == 0;
    bool real[8];
    auto rowof = [&](int row, bool& ok) -> long {  // row of the operand matrices that holds (row, b); ok = false: zero row
      long r = (long)min(row, T - 1) * p.B + b;
      ok = row < T;
      if (p.rowmap) {
        const int pr = p.rowmap[r];
        ok = ok && pr >= 0;
        r = pr >= 0 ? pr : 0;
      }
      return r;
    };
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = lane + 64 * u, row = i >> 4, c = (i & 15) << 2;
      const long o = rowof(row, real[u]) * p.ld + off + c;
      if (al) { kk[u] = *reinterpret_cast<const float4*>(p.k + o); vv[u] = *reinterpret_cast<const float4*>(p.v + o); }
      else { kk[u] = make_float4(p.k[o], p.k[o + 1], p.k[o + 2], p.k[o + 3]); vv[u] = make_float4(p.v[o], p.v[o + 1], p.v[o + 2], p.v[o + 3]); }
    }
    bool qreal;
    fetch_op(qa, p.q + rowof(li, qreal) * p.ld + off + 32 * lh);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = lane + 64 * u, row = i >> 4, c = (i & 15) << 2;
      const float m = real[u] ? 1.f : 0.f;
      float* dk = Ks + row * LS + c;
      float* dv = Vs + row * LS + c;
      dk[0] = kk[u].x * m; dk[1] = kk[u].y * m; dk[2] = kk[u].z * m; dk[3] = kk[u].w * m;
      dv[0] = vv[u].x * m; dv[1] = vv[u].y * m; dv[2] = vv[u].z * m; dv[3] = vv[u].w * m;
    }
  }
  __syncthreads();
  if (bhid >= nbh) return;
  const uint64_t bh = (uint64_t)(p.col_offset + b) * p.nhead + head;
#ifdef BLM_ATTN_PROF
  long long stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  attn_fwd_pass(p, Ks, Vs, 0, qa, b, off, bh, li, lh, bhid, stamps);
#else
  attn_fwd_pass(p, Ks, Vs, 0, qa, b, off, bh, li, lh, bhid);
#endif
}

// ------------------------------------------------------------------ backward: dQ (lane = query)
__device__ __forceinline__ void attn_dq_pass(const AttnM& p, const float* Ks, const float* Vs, int qt, float (&qreg)[32],
                                             const float (&doreg)[32], float delta, int b, int off, uint64_t bh, int li, int lh,
                                             int bhid) {
  const int T = p.T;
  const int q = 32 * qt + li, qc = min(q, T - 1);
#pragma unroll
  for (int s = 0; s < 32; ++s) qreg[s] *= p.scale;
  delta += __shfl_xor(delta, 32, 64);
  const float lse = p.lse[(long)bhid * T + qc];
  f32x16 dqt[2] = {(f32x16)(0.f), (f32x16)(0.f)};
#pragma unroll 1
  for (int kt = 0; kt <= qt; ++kt) {
    f32x16 st, dp;
    tile2_rows_x_regs(st, dp, Ks, Vs, 32 * kt, qreg, doreg, li, lh);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float kp[4] = {1.f, 1.f, 1.f, 1.f};
      if (p.drop) keep_row4(p, bh * T + qc, 32 * kt + 8 * g + 4 * lh, kp);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g + e;
        const int key = 32 * kt + mrow(r, lh);
        const float pr = key <= q ? __expf(st[r] - lse) : 0.f;
        st[r] = pr * (dp[r] * kp[e] - delta);  // dS^T
      }
    }
    acc_xt_regs(dqt, Ks, 32 * kt, st, li, lh);
  }
  if (q < T) store_t(p.dq + ((long)q * p.B + b) * p.ldd + off, dqt, lh, p.scale);
}

template <int HPW>
__global__ __launch_bounds__(256 * HPW) void attn_bwd_dq_mfma_kernel(const AttnM p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int hsel = threadIdx.x >> 8, tid = threadIdx.x & 255;
  const int bhid = blockIdx.x * HPW + hsel;
  float* Ks = sm + hsel * 2 * AT * LS;
  float* Vs = Ks + AT * LS;
  const int b = bhid / p.nhead, head = bhid % p.nhead, off = head * HD;
  const int T = p.T, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const long dmodel = (long)p.nhead * HD;
  const int ntile = (T + 31) >> 5;
  const int qt = hsel ? wave : 3 - wave;  // one query tile per wave; SIMD-balanced pairs with HPW = 2 (forward kernel)
  float qa[32], da[32];
  float delta = 0.f;
  {
    const long r0 = (long)min(32 * qt + li, T - 1) * p.B + b;
    float4 kk[8], vv[8];
    float oa[32];
    fetch_rows<256>(kk, p.k, p.ld, T, p.B, b, off, tid);
    fetch_rows<256>(vv, p.v, p.ld, T, p.B, b, off, tid);
    fetch_op(qa, p.q + r0 * p.ld + off + 32 * lh);
    fetch_op(da, p.dout + r0 * dmodel + off + 32 * lh);
    fetch_op(oa, p.o_in + r0 * dmodel + off + 32 * lh);
    put_rows<256>(Ks, kk, T, 1.f, tid);
    put_rows<256>(Vs, vv, T, 1.f, tid);
#pragma unroll
    for (int s = 0; s < 32; ++s) delta += da[s] * oa[s];
  }
  __syncthreads();
  const uint64_t bh = (uint64_t)(p.col_offset + b) * p.nhead + head;
  if (qt < ntile) attn_dq_pass(p, Ks, Vs, qt, qa, da, delta, b, off, bh, li, lh, bhid);
}

// ------------------------------------------------------------------ backward: dQ from the saved dS (lane = query)
// dQ[q][:] = scale * sum_key dS[q][key] K[key][:] with dS read back from the workspace the dK/dV pass filled: no
// second recomputation of S, dP, the probabilities and their dropout masks (the stand-alone dQ kernel above spends
// 64 of its 96 MFMAs per tile pair and all of its vector work on exactly that).  Same tile ownership as the forward.
__device__ __forceinline__ void load_ds_rows(const AttnM& p, float4 (&dsr)[4][4], int bhid, int qt, int ntile, int li, int lh) {
  const int T = p.T;
  const int qc = min(32 * qt + li, T - 1);
  const float* drow = p.ds + ((long)bhid * T + qc) * T;
  const bool vec = (T & 3) == 0 && (reinterpret_cast<uintptr_t>(p.ds) & 15) == 0;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int k0 = 32 * kt + 8 * g + 4 * lh;  // this lane's accumulator registers 4g..4g+3 = keys k0..k0+3
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kt <= qt && qt < ntile) {  // wave-uniform
        if (vec) {
          if (k0 < T) v = *reinterpret_cast<const float4*>(drow + k0);
        } else {
          if (k0 < T) v.x = drow[k0];
          if (k0 + 1 < T) v.y = drow[k0 + 1];
          if (k0 + 2 < T) v.z = drow[k0 + 2];
          if (k0 + 3 < T) v.w = drow[k0 + 3];
        }
      }
      dsr[kt][g] = v;
    }
  }
}
__device__ __forceinline__ void dq_from_ds(const AttnM& p, const float* Ks, const float4 (&dsr)[4][4], int qt, int b, int off, int li, int lh) {
  f32x16 dqt[2] = {(f32x16)(0.f), (f32x16)(0.f)};
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    if (kt <= qt) {
      f32x16 st;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        st[4 * g] = dsr[kt][g].x; st[4 * g + 1] = dsr[kt][g].y; st[4 * g + 2] = dsr[kt][g].z; st[4 * g + 3] = dsr[kt][g].w;
      }
      acc_xt_regs(dqt, Ks, 32 * kt, st, li, lh);
    }
  }
  const int q = 32 * qt + li;
  if (q < p.T) store_t(p.dq + ((long)q * p.B + b) * p.ldd + off, dqt, lh, p.scale);
}

// ------------------------------------------------------------------ backward: dK, dV (lane = key)
__device__ __forceinline__ void attn_dkv_pass(const AttnM& p, const float* Qs, const float* Os, const float* lse_s,
                                              const float* del_s, int kt, int ntile, const float (&kreg)[32],
                                              const float (&vreg)[32], int b, int off, uint64_t bh, int lane, int li, int lh,
                                              int bhid) {
  const int T = p.T;
  const int key = 32 * kt + li, kc = min(key, T - 1);
  f32x16 dkt[2] = {(f32x16)(0.f), (f32x16)(0.f)}, dvt[2] = {(f32x16)(0.f), (f32x16)(0.f)};
#pragma unroll 1
  for (int qt = kt; qt < ntile; ++qt) {
    f32x16 sc, dp;  // S[q][key], dP[q][key]: q in registers
    tile2_rows_x_regs(sc, dp, Qs, Os, 32 * qt, kreg, vreg, li, lh);
    f32x16 pd;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float kp[4] = {1.f, 1.f, 1.f, 1.f};
      if (p.drop) {  // rows q0..q0+3 of the probability matrix, 4 adjacent lanes = 4 adjacent keys
        const int q0 = 32 * qt + 8 * g + 4 * lh;
        const int kq = lane & 3;
        const int qrow = min(q0 + kq, T - 1);
        if ((T & 3) == 0) {
          const uint64_t gidx = (bh * T + qrow) * (uint64_t)T + (uint64_t)(min(key, T - 1) & ~3);
          const u32x4 u = philox4x32_10((uint32_t)(gidx >> 2), (uint32_t)(gidx >> 34), p.rng.stream, p.rng.step,
                                               (uint32_t)p.rng.seed, (uint32_t)(p.rng.seed >> 32));
#define BLM_QB(x, j) (uint32_t) __builtin_amdgcn_mov_dpp((int)(x), (j) * 0x55, 0xF, 0xF, true)
#define BLM_KP(j)                                                                                         \
{                                                                                                       \
  const uint32_t w0 = BLM_QB(u.x, j), w1 = BLM_QB(u.y, j), w2 = BLM_QB(u.z, j), w3 = BLM_QB(u.w, j);     \
  const uint32_t w = kq == 0 ? w0 : (kq == 1 ? w1 : (kq == 2 ? w2 : w3));                               \
  kp[j] = w >= p.thr ? p.inv_keep : 0.f;                                                                \
}
          BLM_KP(0) BLM_KP(1) BLM_KP(2) BLM_KP(3)
#undef BLM_KP
#undef BLM_QB
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            kp[e] = philox_bits1_rolled(p.rng, (bh * T + min(q0 + e, T - 1)) * (uint64_t)T + kc) >= p.thr ? p.inv_keep : 0.f;
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g + e;
        const int qr = 32 * qt + mrow(r, lh);
        const float pr = (qr >= key && qr < T) ? __expf(sc[r] - lse_s[qr]) : 0.f;
        pd[r] = pr * kp[e];
        sc[r] = pr * (dp[r] * kp[e] - del_s[qr]);  // dS[q][key]
      }
    }
    if (p.ds && key < T) {  // dS[q][key], row-major per head: 32 lanes = 32 consecutive keys of one query row
      // 32-bit element offsets off the uniform base (the host checks that the workspace is under 4 GB): one VGPR per
      // address instead of two -- the HPW = 2 kernel sits at its 256-register budget
      const uint32_t o0 = ((uint32_t)bhid * T + 32 * qt + 4 * lh) * T + key;
      if (32 * qt + 32 <= T) {
#pragma unroll
        for (int r = 0; r < 16; ++r) p.ds[o0 + (uint32_t)(((r & 3) + 8 * (r >> 2)) * T)] = sc[r];
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (32 * qt + mrow(r, lh) < T) p.ds[o0 + (uint32_t)(((r & 3) + 8 * (r >> 2)) * T)] = sc[r];
      }
    }
    acc_xt_regs(dvt, Os, 32 * qt, pd, li, lh);
    acc_xt_regs(dkt, Qs, 32 * qt, sc, li, lh);
  }
  if (key < T) {
    store_t(p.dk + ((long)key * p.B + b) * p.ldd + off, dkt, lh, 1.f);
    store_t(p.dv + ((long)key * p.B + b) * p.ldd + off, dvt, lh, 1.f);
  }
}

// DQ: the workgroup also produces dQ of its heads from the dS tiles its own waves have just written (they are in this
// XCD's L2): no third launch, no second load of K (the wave's key tile goes from its operand registers to LDS).
template <int HPW, bool DQ = false>
__global__ __launch_bounds__(256 * HPW) void attn_bwd_dkv_mfma_kernel(const AttnM p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int hsel = threadIdx.x >> 8, tid = threadIdx.x & 255;
  const int bhid = blockIdx.x * HPW + hsel;
  float* Qs = sm + hsel * (2 * AT * LS + 2 * AT);  // Q * scale
  float* Os = Qs + AT * LS;                        // dO
  float* lse_s = Qs + 2 * AT * LS;
  float* del_s = lse_s + AT;
  const int b = bhid / p.nhead, head = bhid % p.nhead, off = head * HD;
  const int T = p.T, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const long dmodel = (long)p.nhead * HD;
  const int ntile = (T + 31) >> 5;
  const int kt = hsel ? 3 - wave : wave;  // key tile kt meets query tiles kt..ntile-1: one per wave; SIMD-balanced with HPW = 2
  float ka[32], va[32];
  {
    const long r0 = (long)min(32 * kt + li, T - 1) * p.B + b;
    float4 qq[8], dd[8];
    fetch_rows<256>(qq, p.q, p.ld, T, p.B, b, off, tid);
    fetch_rows<256>(dd, p.dout, dmodel, T, p.B, b, off, tid);
    fetch_op(ka, p.k + r0 * p.ld + off + 32 * lh);
    fetch_op(va, p.v + r0 * p.ld + off + 32 * lh);
    put_rows<256>(Qs, qq, T, p.scale, tid);
    put_rows<256>(Os, dd, T, 1.f, tid);
  }
  {  // delta[q] = rowsum(dO * O), lse[q]: two threads per row (half a row each, 2 x 8 float4 in flight)
    const int row = tid >> 1, half = tid & 1, rc = min(row, T - 1);
    const float* ds = p.dout + ((long)rc * p.B + b) * dmodel + off + (HD / 2) * half;
    const float* os = p.o_in + ((long)rc * p.B + b) * dmodel + off + (HD / 2) * half;
    const float ls = p.lse[(long)bhid * T + rc];
    float d = 0.f;
    if ((((uintptr_t)ds | (uintptr_t)os) & 15) == 0) {
      float4 dv4[HD / 8], ov4[HD / 8];
#pragma unroll
      for (int c = 0; c < HD / 8; ++c) { dv4[c] = reinterpret_cast<const float4*>(ds)[c]; ov4[c] = reinterpret_cast<const float4*>(os)[c]; }
#pragma unroll
      for (int c = 0; c < HD / 8; ++c) d += dv4[c].x * ov4[c].x + dv4[c].y * ov4[c].y + dv4[c].z * ov4[c].z + dv4[c].w * ov4[c].w;
    } else {
      for (int c = 0; c < HD / 2; ++c) d += ds[c] * os[c];
    }
    d += __shfl_xor(d, 1);
    if (half == 0) {
      lse_s[row] = row < T ? ls : 0.f;
      del_s[row] = row < T ? d : 0.f;
    }
  }
  __syncthreads();
  const uint64_t bh = (uint64_t)(p.col_offset + b) * p.nhead + head;
  if (kt < ntile) attn_dkv_pass(p, Qs, Os, lse_s, del_s, kt, ntile, ka, va, b, off, bh, lane, li, lh, bhid);
  if constexpr (DQ) {
    // __syncthreads() carries workgroup-scope release / acquire fences: the dS tiles of every wave of this workgroup
    // (same CU, same write-through L1) are visible to the loads below.  An agent-scope fence here would write the
    // XCD's whole L2 back (measured: 245 us per launch instead of 75).
    __syncthreads();   // ... and nobody reads Qs / dOs any more
    float* Ks = Qs;    // K rows of the head, row-major like the staged tiles (stride LS)
#pragma unroll
    for (int s2 = 0; s2 < 32; ++s2) Ks[(32 * kt + li) * LS + 32 * lh + s2] = ka[s2];
    const int qt = hsel ? wave : 3 - wave;  // query tile of the forward: 5 tile-steps per SIMD again
    float4 dsr[4][4];
    load_ds_rows(p, dsr, bhid, qt, ntile, li, lh);
    __syncthreads();
    if (qt < ntile) dq_from_ds(p, Ks, dsr, qt, b, off, li, lh);
  }
}

// ------------------------------------------------------------------ sequences longer than 128 tokens
// Same tiles and operand orientation, with a flash-style outer loop: a workgroup owns 128 queries (keys) = one
// 32-row tile per wave and walks the 128-row chunks of the other side through LDS -- the forward with an online
// softmax (running max / sum per lane = per query, accumulators rescaled when the max moves), the backward passes
// with the saved log-sum-exp.  Not tuned like the T <= 128 kernels above (one chunk in LDS at a time, two barriers
// per chunk): it exists so that long hypotheses and --seq_len > 128 run on the same path (the reference takes any
// length up to its 5000-row positional table, model.py:97-103).
template <int NT>
__device__ __forceinline__ void fetch_rows_at(float4 (&v)[2048 / NT], const float* src, long ld, int T, int B, int b, int off, int row0) {
  const bool al = ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)(ld * 4) | (uintptr_t)(off * 4)) & 15) == 0;
#pragma unroll
  for (int u = 0; u < 2048 / NT; ++u) {
    const int i = threadIdx.x + NT * u, row = i >> 4, c = (i & 15) << 2;
    const float* s = src + ((long)min(row0 + row, T - 1) * B + b) * ld + off + c;
    if (al) v[u] = *reinterpret_cast<const float4*>(s);
    else v[u] = make_float4(s[0], s[1], s[2], s[3]);
  }
}

__global__ __launch_bounds__(256) void attn_fwd_long_kernel(const AttnM p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Ks = sm;
  float* Vs = sm + AT * LS;
  const int b = blockIdx.x / p.nhead, head = blockIdx.x % p.nhead, off = head * HD;
  const int T = p.T, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
  const int qb = gridDim.y - 1 - blockIdx.y;  // the longest query blocks first
  const int qt = 4 * qb + wave, q = 32 * qt + li, qc = min(q, T - 1);
  const uint64_t bh = (uint64_t)(p.col_offset + b) * p.nhead + head;
  float qa[32];
  fetch_op(qa, p.q + ((long)qc * p.B + b) * p.ld + off + 32 * lh);
#pragma unroll
  for (int s = 0; s < 32; ++s) qa[s] *= p.scale;
  float m = -INFINITY, l = 0.f;
  f32x16 ot[2] = {(f32x16)(0.f), (f32x16)(0.f)};
#pragma unroll 1
  for (int kc = 0; kc <= qb; ++kc) {
    __syncthreads();  // every wave is done with the previous chunk
    {
      float4 kk[8], vv[8];
      fetch_rows_at<256>(kk, p.k, p.ld, T, p.B, b, off, 128 * kc);
      fetch_rows_at<256>(vv, p.v, p.ld, T, p.B, b, off, 128 * kc);
      put_rows<256>(Ks, kk, T - 128 * kc, 1.f, threadIdx.x);
      put_rows<256>(Vs, vv, T - 128 * kc, 1.f, threadIdx.x);
    }
    __syncthreads();
#pragma unroll 1
    for (int kt = 0; kt < 4; ++kt) {
      const int gk0 = 128 * kc + 32 * kt;       // first key of the tile
      if (gk0 > 32 * qt + 31 || gk0 >= T) break;  // wave-uniform: above the diagonal / past the sequence
      f32x16 st = tile_rows_x_regs(Ks, 32 * kt, qa, li, lh);
      float tm = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (gk0 + mrow(r, lh) > q) st[r] = -INFINITY;  // causal (also hides keys >= T for valid queries)
        tm = fmaxf(tm, st[r]);
      }
      tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
      const float mn = fmaxf(m, tm);  // finite: key gk0 <= q of some lane half is always unmasked in a processed tile
      const float alpha = __expf(m - mn);
      float ts = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        st[r] = __expf(st[r] - mn);
        ts += st[r];
      }
      ts += __shfl_xor(ts, 32, 64);
      l = l * alpha + ts;
      m = mn;
      ot[0] *= alpha;
      ot[1] *= alpha;
      if (p.drop) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float kp[4];
          keep_row4(p, bh * T + qc, gk0 + 8 * g + 4 * lh, kp);
#pragma unroll
          for (int e = 0; e < 4; ++e) st[4 * g + e] *= kp[e];
        }
      }
      acc_xt_regs(ot, Vs, 32 * kt, st, li, lh);
    }
  }
  if (q < T) {
    store_t(p.out + ((long)q * p.B + b) * ((long)p.nhead * HD) + off, ot, lh, 1.f / l);
    if (p.lse && lh == 0) p.lse[(long)blockIdx.x * T + q] = m + __logf(l);
  }
}

__global__ __launch_bounds__(256) void attn_bwd_dq_long_kernel(const AttnM p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Ks = sm;
  float* Vs = sm + AT * LS;
  const int b = blockIdx.x / p.nhead, head = blockIdx.x % p.nhead, off = head * HD;
  const int T = p.T, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
  const long dmodel = (long)p.nhead * HD;
  const int qb = gridDim.y - 1 - blockIdx.y;
  const int qt = 4 * qb + wave, q = 32 * qt + li, qc = min(q, T - 1);
  const uint64_t bh = (uint64_t)(p.col_offset + b) * p.nhead + head;
  float qa[32], da[32];
  float delta = 0.f;
  {
    const long r0 = (long)qc * p.B + b;
    float oa[32];
    fetch_op(qa, p.q + r0 * p.ld + off + 32 * lh);
    fetch_op(da, p.dout + r0 * dmodel + off + 32 * lh);
    fetch_op(oa, p.o_in + r0 * dmodel + off + 32 * lh);
#pragma unroll
    for (int s = 0; s < 32; ++s) { delta += da[s] * oa[s]; qa[s] *= p.scale; }
  }
  delta += __shfl_xor(delta, 32, 64);
  const float lse = p.lse[(long)blockIdx.x * T + qc];
  f32x16 dqt[2] = {(f32x16)(0.f), (f32x16)(0.f)};
#pragma unroll 1
  for (int kc = 0; kc <= qb; ++kc) {
    __syncthreads();
    {
      float4 kk[8], vv[8];
      fetch_rows_at<256>(kk, p.k, p.ld, T, p.B, b, off, 128 * kc);
      fetch_rows_at<256>(vv, p.v, p.ld, T, p.B, b, off, 128 * kc);
      put_rows<256>(Ks, kk, T - 128 * kc, 1.f, threadIdx.x);
      put_rows<256>(Vs, vv, T - 128 * kc, 1.f, threadIdx.x);
    }
    __syncthreads();
#pragma unroll 1
    for (int kt = 0; kt < 4; ++kt) {
      const int gk0 = 128 * kc + 32 * kt;
      if (gk0 > 32 * qt + 31 || gk0 >= T) break;
      f32x16 st, dp;
      tile2_rows_x_regs(st, dp, Ks, Vs, 32 * kt, qa, da, li, lh);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float kp[4] = {1.f, 1.f, 1.f, 1.f};
        if (p.drop) keep_row4(p, bh * T + qc, gk0 + 8 * g + 4 * lh, kp);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          const float pr = gk0 + mrow(r, lh) <= q ? __expf(st[r] - lse) : 0.f;
          st[r] = pr * (dp[r] * kp[e] - delta);  // dS^T
        }
      }
      acc_xt_regs(dqt, Ks, 32 * kt, st, li, lh);
    }
  }
  if (q < T) store_t(p.dq + ((long)q * p.B + b) * p.ldd + off, dqt, lh, p.scale);
}

// lane = key; walks the query chunks kb .. last (queries >= the key)
__global__ __launch_bounds__(256) void attn_bwd_dkv_long_kernel(const AttnM p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Qs = sm;            // Q * scale
  float* Os = sm + AT * LS;  // dO
  float* lse_s = sm + 2 * AT * LS;
  float* del_s = lse_s + AT;
  const int b = blockIdx.x / p.nhead, head = blockIdx.x % p.nhead, off = head * HD;
  const int T = p.T, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
  const long dmodel = (long)p.nhead * HD;
  const int kb = blockIdx.y;  // key block 0 meets every query chunk: the longest first
  const int nqc = (T + 127) >> 7;
  const int kt = 4 * kb + wave, key = 32 * kt + li, kc = min(key, T - 1);
  const uint64_t bh = (uint64_t)(p.col_offset + b) * p.nhead + head;
  float ka[32], va[32];
  fetch_op(ka, p.k + ((long)kc * p.B + b) * p.ld + off + 32 * lh);
  fetch_op(va, p.v + ((long)kc * p.B + b) * p.ld + off + 32 * lh);
  f32x16 dkt[2] = {(f32x16)(0.f), (f32x16)(0.f)}, dvt[2] = {(f32x16)(0.f), (f32x16)(0.f)};
#pragma unroll 1
  for (int qcn = kb; qcn < nqc; ++qcn) {
    __syncthreads();
    {
      float4 qq[8], dd[8];
      fetch_rows_at<256>(qq, p.q, p.ld, T, p.B, b, off, 128 * qcn);
      fetch_rows_at<256>(dd, p.dout, dmodel, T, p.B, b, off, 128 * qcn);
      put_rows<256>(Qs, qq, T - 128 * qcn, p.scale, threadIdx.x);
      put_rows<256>(Os, dd, T - 128 * qcn, 1.f, threadIdx.x);
    }
    {  // delta[q] = rowsum(dO * O), lse[q] of this chunk's rows: two threads per row
      const int row = threadIdx.x >> 1, half = threadIdx.x & 1, grow = 128 * qcn + row, rc = min(grow, T - 1);
      const float* ds = p.dout + ((long)rc * p.B + b) * dmodel + off + (HD / 2) * half;
      const float* os = p.o_in + ((long)rc * p.B + b) * dmodel + off + (HD / 2) * half;
      float d = 0.f;
      for (int c = 0; c < HD / 2; ++c) d += ds[c] * os[c];
      d += __shfl_xor(d, 1);
      if (half == 0) {
        lse_s[row] = grow < T ? p.lse[(long)blockIdx.x * T + rc] : 0.f;
        del_s[row] = grow < T ? d : 0.f;
      }
    }
    __syncthreads();
#pragma unroll 1
    for (int qt = 0; qt < 4; ++qt) {
      const int gq0 = 128 * qcn + 32 * qt;           // first query of the tile
      if (gq0 >= T) break;                           // wave-uniform
      if (gq0 + 31 < 32 * kt) continue;              // every query of the tile is below every key of this wave
      f32x16 sc, dp;  // S[q][key], dP[q][key]: q in registers
      tile2_rows_x_regs(sc, dp, Qs, Os, 32 * qt, ka, va, li, lh);
      f32x16 pd;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float kp[4] = {1.f, 1.f, 1.f, 1.f};
        if (p.drop) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            kp[e] = philox_bits1_rolled(p.rng, (bh * T + min(gq0 + 8 * g + 4 * lh + e, T - 1)) * (uint64_t)T + kc) >= p.thr ? p.inv_keep : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e, lr = 32 * qt + mrow(r, lh), qr = 128 * qcn + lr;
          const float pr = (qr >= key && qr < T) ? __expf(sc[r] - lse_s[lr]) : 0.f;
          pd[r] = pr * kp[e];
          sc[r] = pr * (dp[r] * kp[e] - del_s[lr]);  // dS[q][key]
        }
      }
      acc_xt_regs(dvt, Os, 32 * qt, pd, li, lh);
      acc_xt_regs(dkt, Qs, 32 * qt, sc, li, lh);
    }
  }
  if (key < T) {
    store_t(p.dk + ((long)key * p.B + b) * p.ldd + off, dkt, lh, 1.f);
    store_t(p.dv + ((long)key * p.B + b) * p.ldd + off, dvt, lh, 1.f);
  }
}

}  // namespace blm

using namespace blm;

// Heads per workgroup of the T <= 128 kernels.  Two heads (8 waves, balanced causal tiles per SIMD, see the forward) as long
// as that still leaves one workgroup per CU; below -- the recipes' batch 32 (256 heads), evaluation at batch 20 -- one head per
// workgroup, so that the launch covers twice the CUs (recipe Transformer step 9.23 -> 9.12 ms, evaluation 2.03 -> 2.00 ms;
// at 512 heads both forms tie).  Option "attn_hpw" = 1 | 2 forces one form (blm_set_option; both are parity-tested).
static bool attn_short() { return blm::option(blm::OPT_ATTN_SHORT) != 0; }  // "attn_short" = 0: the 128-row forward also for T <= 32
static int attn_hpw(int heads) {
  const int v = blm::option(blm::OPT_ATTN_HPW);  // "attn_hpw" = 1 | 2 forces one form
  if (v == 1 || v == 2) return v;
  return heads / 2 >= 256 ? 2 : 1;
}

static void fill_m(AttnM& p, int T, int B, int nhead, float pdrop, const blm_rng* rng, int col_offset) {
  p.T = T; p.B = B; p.nhead = nhead;
  p.scale = 0.125f;  // 64^-0.5
  p.drop = pdrop > 0.f;
  if (p.drop) p.rng = *rng;
  const double t = (double)pdrop * 4294967296.0;
  p.thr = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
  p.inv_keep = pdrop < 1.f ? 1.f / (1.f - pdrop) : 0.f;
  p.col_offset = col_offset;
}

template <typename K>
static int set_lds(K kern, size_t lds) {
  BLM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  return BLM_OK;
}

// called by blm_attn_fwd / blm_attn_bwd (attention.hip) when head_dim == 64
int blm_attn_fwd_mfma(const float* q, const float* k, const float* v, int64_t ld, float* out, float* lse, int T, int B,
                      int nhead, float pdrop, const blm_rng* rng, int col_offset, hipStream_t st) {
  AttnM p{};
  fill_m(p, T, B, nhead, pdrop, rng, col_offset);
  p.q = q; p.k = k; p.v = v; p.ld = ld; p.out = out; p.lse = lse;
  const size_t lds = (size_t)2 * AT * LS * sizeof(float);
  static bool once = false;
  if (!once) {
    int rc = set_lds(attn_fwd_mfma_kernel<1>, lds);
    if (rc) return rc;
    rc = set_lds(attn_fwd_mfma_kernel<2>, 2 * lds);
    if (rc) return rc;
    rc = set_lds(attn_fwd_long_kernel, lds);
    if (rc) return rc;
    once = true;
  }
  if (T > AT) {  // flash-style chunk loop, one workgroup per (batch column, head, 128-query block)
    hipLaunchKernelGGL(attn_fwd_long_kernel, dim3(B * nhead, (T + AT - 1) / AT), dim3(256), lds, st, p);
    BLM_HIP(hipGetLastError());
    return BLM_OK;
  }
  if (T <= 32 && attn_short()) {  // one wave per head (attn_fwd_short_kernel)
    static bool once_s = false;
    const size_t lds_s = (size_t)4 * 2 * 32 * LS * sizeof(float);
    if (!once_s) {
      const int rc = set_lds(attn_fwd_short_kernel, lds_s);
      if (rc) return rc;
      once_s = true;
    }
    hipLaunchKernelGGL(attn_fwd_short_kernel, dim3((B * nhead + 3) / 4), dim3(256), lds_s, st, p);
    BLM_HIP(hipGetLastError());
    return BLM_OK;
  }
  if ((B * nhead) % 2 == 0 && attn_hpw(B * nhead) == 2) hipLaunchKernelGGL(attn_fwd_mfma_kernel<2>, dim3(B * nhead / 2), dim3(512), 2 * lds, st, p);
  else hipLaunchKernelGGL(attn_fwd_mfma_kernel<1>, dim3(B * nhead), dim3(256), lds, st, p);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

// Forward on PACKED rows (inference, T <= 32, head_dim 64): q / k / v (R, ld) and out (R, nhead * 64) hold the real tokens of a padded
// (T, B) batch only, rowmap (T * B) maps a padded position to its row or -1.  BLM_ERR_UNSUPPORTED: the caller scatters / gathers
// around blm_attn_fwd instead.
int blm_attn_fwd_rows_mfma(const float* q, const float* k, const float* v, int64_t ld, float* out, const int* rowmap, int T, int B,
                           int nhead, hipStream_t st) {
  if (T > 32 || !attn_short()) return blm_fail(BLM_ERR_UNSUPPORTED, "blm_attn_fwd_rows: packed rows need T <= 32 and the one-wave-per-head form");
  AttnM p{};
  fill_m(p, T, B, nhead, 0.f, nullptr, 0);
  p.q = q; p.k = k; p.v = v; p.ld = ld; p.out = out; p.lse = nullptr; p.rowmap = rowmap;
  static bool once_s = false;
  const size_t lds_s = (size_t)4 * 2 * 32 * LS * sizeof(float);
  if (!once_s) {
    const int rc = set_lds(attn_fwd_short_kernel, lds_s);
    if (rc) return rc;
    once_s = true;
  }
  hipLaunchKernelGGL(attn_fwd_short_kernel, dim3((B * nhead + 3) / 4), dim3(256), lds_s, st, p);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

// floats of workspace the T <= 128 backward can use (0: the path has no use for one)
int64_t blm_attn_bwd_mfma_ws_floats(int T, int B, int nhead) {
  const int64_t n = (int64_t)B * nhead * T * T;
  return (T <= AT && n < (1LL << 30)) ? n : 0;  // the kernels address it with 32-bit byte offsets
}

int blm_attn_bwd_mfma(const float* q, const float* k, const float* v, int64_t ld, const float* out, const float* dout,
                      const float* lse, float* dq, float* dk, float* dv, int64_t ldd, int T, int B, int nhead,
                      float pdrop, const blm_rng* rng, int col_offset, float* ws, hipStream_t st) {
  AttnM p{};
  fill_m(p, T, B, nhead, pdrop, rng, col_offset);
  p.q = q; p.k = k; p.v = v; p.ld = ld; p.o_in = out; p.dout = dout; p.lse = const_cast<float*>(lse);
  p.dq = dq; p.dk = dk; p.dv = dv; p.ldd = ldd;
  p.ds = blm_attn_bwd_mfma_ws_floats(T, B, nhead) > 0 ? ws : nullptr;
  const size_t lds1 = (size_t)2 * AT * LS * sizeof(float), lds2 = lds1 + 2 * AT * sizeof(float);
  static bool once = false;
  if (!once) {
    int rc = set_lds(attn_bwd_dq_mfma_kernel<1>, lds1);
    if (rc) return rc;
    rc = set_lds(attn_bwd_dkv_mfma_kernel<1>, lds2);
    if (rc) return rc;
    rc = set_lds(attn_bwd_dq_mfma_kernel<2>, 2 * lds1);
    if (rc) return rc;
    rc = set_lds(attn_bwd_dkv_mfma_kernel<2>, 2 * lds2);
    if (rc) return rc;
    rc = set_lds(attn_bwd_dq_long_kernel, lds1);
    if (rc) return rc;
    rc = set_lds(attn_bwd_dkv_long_kernel, lds2);
    if (rc) return rc;
    rc = set_lds(attn_bwd_dkv_mfma_kernel<1, true>, lds2);
    if (rc) return rc;
    rc = set_lds(attn_bwd_dkv_mfma_kernel<2, true>, 2 * lds2);
    if (rc) return rc;
    once = true;
  }
  if (T > AT) {
    const dim3 grid(B * nhead, (T + AT - 1) / AT);
    hipLaunchKernelGGL(attn_bwd_dq_long_kernel, grid, dim3(256), lds1, st, p);
    BLM_HIP(hipGetLastError());
    hipLaunchKernelGGL(attn_bwd_dkv_long_kernel, grid, dim3(256), lds2, st, p);
    BLM_HIP(hipGetLastError());
    return BLM_OK;
  }
  const bool two = (B * nhead) % 2 == 0 && attn_hpw(B * nhead) == 2;
  if (p.ds) {  // ONE launch: dK/dV, dS through the workspace, dQ = dS K by the same workgroup
    if (two) hipLaunchKernelGGL((attn_bwd_dkv_mfma_kernel<2, true>), dim3(B * nhead / 2), dim3(512), 2 * lds2, st, p);
    else hipLaunchKernelGGL((attn_bwd_dkv_mfma_kernel<1, true>), dim3(B * nhead), dim3(256), lds2, st, p);
    BLM_HIP(hipGetLastError());
    return BLM_OK;
  }
  if (two) {
    hipLaunchKernelGGL(attn_bwd_dq_mfma_kernel<2>, dim3(B * nhead / 2), dim3(512), 2 * lds1, st, p);
    BLM_HIP(hipGetLastError());
    hipLaunchKernelGGL(attn_bwd_dkv_mfma_kernel<2>, dim3(B * nhead / 2), dim3(512), 2 * lds2, st, p);
  } else {
    hipLaunchKernelGGL(attn_bwd_dq_mfma_kernel<1>, dim3(B * nhead), dim3(256), lds1, st, p);
    BLM_HIP(hipGetLastError());
    hipLaunchKernelGGL(attn_bwd_dkv_mfma_kernel<1>, dim3(B * nhead), dim3(256), lds2, st, p);
  }
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}
