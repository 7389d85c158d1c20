// fp32 MFMA GEMM family for gfx950 (v_mfma_f32_32x32x2_f32), LDS tiled,
// register-staged double buffering, optional variational B operand
// (W = mu + exp(lgstd)*eps formed in the tile loader: Philox eps is generated
// and consumed on chip, W never exists in HBM) and fused epilogues.
//
// Replaces the F.linear / autograd matmul call sites of the reference:
//   model.py:1127-1129 (BayesLinear.forward), :876,:921 (qkv_net / o_net),
//   :1043,:1169 (linear1 / linear2), :1306 (decoder), :812 (LSTM gate GEMMs).
//
// Layout: block tile BM x BN x 32 (BM, BN in {64,128}); 4 waves as 2 x 2, each
// wave owns (BM/2) x (BN/2) = WTM x WTN MFMA tiles of 32 x 32.  LDS holds both
// operands k-major-transposed, tile[k][m|n], so an MFMA operand fetch is one
// conflict-free ds_read_b32 (lanes 0-31 consecutive m, lanes 32-63 the next k
// row).  Operands that are k-contiguous in HBM are transposed on the LDS
// write (row stride R+1: conflict-free b32 scatter); operands that are
// m/n-contiguous are copied with ds_write_b128 (row stride R+4).
#pragma once
#include <cstdlib>

#include "blm_device.h"
#include <type_traits>
#include "blm_host.h"

namespace blm {

using f32x16 = __attribute__((ext_vector_type(16))) float;

struct GemmP {
  int M, N, K;
  const float* A; long lda;
  const float* B; long ldb;
  float* C; long ldc;
  float alpha;
  unsigned flags;
  int epi;
  const float* bias;
  float* aux;
  const float* coef;
  blm_variational vb; int vb_cols;
  float* C2;
  const float* wg_mu;
  blm_variational vc;
  float kl_lambda, kl_inv_n;
  int a_vec, b_vec, fast, vec_epi;
  int gm, gn;
  int eps_quad;                  // Bayesian wgrad: eps drawn once per 4 columns and shared inside the quad (N % 4 == 0, Philox mode)
  int plan_tile, plan_splits;    // gemm_plan.hip: 11 / 12 / 21 / 22 and the number of K slices (>= 1)
  int plan_cus;                  // compute units the plan counts its rounds of workgroup slots with (0: the device's)
  float* colsum_a;  // TN only: += column sums of A (= bias gradient of the layer whose wgrad this is)
  int splits, kper, atomic;  // split-K: block ks covers k in [ks*kper, (ks+1)*kper), partial sums by float atomics
  // BLM_EPI_CE_PART (blm_linear_nll: inference, logits never stored): per (row, column tile) the maximum and the sum of
  // exp(logit - maximum) go to aux[(row * gn + tile) * 2 + {0, 1}], the target's logit to ce_tlogit[row]
  const long long* ce_tgt;
  float* ce_tlogit;
  int tail_from;             // tiles (in launch order) below this index are computed whole by one workgroup; only the rest -- the
                             // tiles beyond the last full round of workgroup slots -- are sliced (0: every tile is sliced)
  // fused activation dropout
  int drop_on; uint32_t drop_thr; float drop_inv_keep; blm_rng drop_rng;
  int drop_B, drop_col_offset, drop_global_cols, drop_quad;
  int split;  // opt-in (blm_set_gemm_mode): 0 fp32 MFMA; 3 / 6: operands split into 2 / 3 bf16 parts at fragment read, 3 / 6 bf16 MFMAs per k16 tile
};

// keep factor of element (m, n) of a (rows, B, N) activation, keyed by global column
__device__ __forceinline__ float gemm_keep(const GemmP& p, int m, int n) {
  const int row = m / p.drop_B, b = m - row * p.drop_B;
  const uint64_t g = ((uint64_t)row * p.drop_global_cols + (uint64_t)(p.drop_col_offset + b)) * (uint64_t)p.N + (uint64_t)n;
  return philox_bits1_rolled(p.drop_rng, g) >= p.drop_thr ? p.drop_inv_keep : 0.f;
}

constexpr int BK = 32;
#if defined(BLM_GEMM_PROF) && !defined(BLM_GEMM_LIFE)
#define BLM_GEMM_LIFE  // the per-trip phase counters (heavy: they serialise the loop) imply the three stamps per workgroup (cheap)
#endif
#ifdef BLM_GEMM_PROF
static __device__ unsigned long long blm_prof[4];  // per translation unit (debug build only)
#endif
#ifdef BLM_GEMM_LIFE
// life of every workgroup of the LAST launch (tools/gemm_drift_probe.py, gemm_phase_probe.py): [4*bid] = wall clock
// (10 ns) at entry, [+1] at the end of the K loop, [+2] HW_ID (CU / SE / slot) | XCC_ID << 32, [+3] wall clock when the
// epilogue's last instruction has been issued
static __device__ long long blm_wg_life[4 * 8192];
#define BLM_PROF_WG_END() do { if (threadIdx.x == 0 && blockIdx.x < 8192) blm_wg_life[4 * blockIdx.x + 3] = wall_clock64(); } while (0)
// tools/gemm_epi_probe.py: 1 = the row-wise epilogue computes but does not store, 2 = every workgroup stores to tile (0, 0)
// (the stores are issued but never leave the L2)
static __device__ int blm_dbg_store;
#else
#define BLM_PROF_WG_END() do { } while (0)
#endif
// Epilogue-only switch kept from an experiment (two MFMA tiles of a wave interleaved by rows 2x+t
// instead of stacked 32t+x; measured neutral on MI355X): the operand paths below are the stacked form.
constexpr bool INTERLEAVE = false;

// ---- global -> registers -------------------------------------------------
// k-contiguous source, tile [R rows][32 k]: 8 lanes cover one 128-B row.
template <int R>
__device__ __forceinline__ void g2r_kmaj(const float* __restrict__ src, long ld, int row0, int rows, int k0, int K,
                                         bool vec, float4 (&r)[R / 32]) {
  const int t = threadIdx.x, kq = t & 7, rr = t >> 3;
  const int k = k0 + 4 * kq;
#pragma unroll
  for (int j = 0; j < R / 32; ++j) {
    const int row = row0 + rr + 32 * j;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((unsigned)row < (unsigned)rows) {
      const float* ptr = src + (long)row * ld + k;
      if (vec && k + 3 < K) {
        v = *reinterpret_cast<const float4*>(ptr);
      } else {
        if (k < K) v.x = ptr[0];
        if (k + 1 < K) v.y = ptr[1];
        if (k + 2 < K) v.z = ptr[2];
        if (k + 3 < K) v.w = ptr[3];
      }
    }
    r[j] = v;
  }
}
// m/n-contiguous source, tile [32 k][C cols].
template <int C>
__device__ __forceinline__ void g2r_nmaj(const float* __restrict__ src, long ld, int k0, int K, int col0, int cols,
                                         bool vec, float4 (&r)[C / 32]) {
  constexpr int TPR = C / 4, RPP = 256 / TPR;
  const int t = threadIdx.x, c4 = t % TPR, kr0 = t / TPR;
  const int col = col0 + 4 * c4;
#pragma unroll
  for (int j = 0; j < C / 32; ++j) {
    const int k = k0 + kr0 + RPP * j;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((unsigned)k < (unsigned)K) {
      const float* ptr = src + (long)k * ld + col;
      if (vec && col + 3 < cols) {
        v = *reinterpret_cast<const float4*>(ptr);
      } else {
        if (col < cols) v.x = ptr[0];
        if (col + 1 < cols) v.y = ptr[1];
        if (col + 2 < cols) v.z = ptr[2];
        if (col + 3 < cols) v.w = ptr[3];
      }
    }
    r[j] = v;
  }
}
// ---- registers -> LDS ------------------------------------------------------
// k-contiguous source: LDS image tile[row][KS] (KS = 32 k + 4 pad floats), written as it was loaded
// (one ds_write_b128 per float4, 8 lanes cover a row: conflict-free)
constexpr int KS = 36;
template <int R>
__device__ __forceinline__ void r2s_kmaj(float* tile, const float4 (&r)[R / 32]) {
  const int t = threadIdx.x, kq = t & 7, rr = t >> 3;
#pragma unroll
  for (int j = 0; j < R / 32; ++j) *reinterpret_cast<float4*>(tile + (rr + 32 * j) * KS + 4 * kq) = r[j];
}
template <int C, int S>
__device__ __forceinline__ void r2s_nmaj(float* tile, const float4 (&r)[C / 32]) {
  constexpr int TPR = C / 4, RPP = 256 / TPR;
  const int t = threadIdx.x, c4 = t % TPR, kr0 = t / TPR;
#pragma unroll
  for (int j = 0; j < C / 32; ++j) *reinterpret_cast<float4*>(tile + (kr0 + RPP * j) * S + 4 * c4) = r[j];
}

// W4 = mu4 + exp(lg4) * eps4 for the four elements (srow, scol..scol+3) of W.
__device__ __forceinline__ float4 sample4(float4 mu, float4 lg, const blm_variational& v, int vcols, int srow,
                                          int scol) {
  const int rel = srow - v.row_lo;
  if ((unsigned)rel >= (unsigned)v.srows || scol >= vcols) return mu;
  const long idx = (long)rel * vcols + scol;
  float4 z;
  if (v.eps) z = *reinterpret_cast<const float4*>(v.eps + idx);
  else z = philox_normal4(v.rng, (uint64_t)idx >> 2);
  mu.x += __expf(lg.x) * z.x;
  mu.y += __expf(lg.y) * z.y;
  mu.z += __expf(lg.z) * z.z;
  mu.w += __expf(lg.w) * z.w;
  return mu;
}


// ---- LDS operand fragments read by inline asm (explicit double buffering, see compute()) ----
__device__ __forceinline__ uint32_t lds_u32(const float* p) { return (uint32_t)(uintptr_t)p; }  // LDS byte offset

using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

// Operand fragments of one GROUP of four MFMA k-steps.  Within a K tile of 32 the k order is
// k(t, s, half) = 8t + 4*half + s  (t = group 0..3, s = step 0..3, half = lane >> 5): a lane's four
// values of a group are then CONTIGUOUS in a k-contiguous LDS row, i.e. one ds_read_b128.
//   FragG<true , W>: image tile[row][KS]   -> W ds_read_b128 per group
//   FragG<false, W>: image tile[k][S]      -> 4 ds_read(2)_b32 per group (rows 8t + 4*half + s)
// All reads are inline asm so that the group t+1 reads can be put in front of the group t MFMAs with
// a counted lgkmcnt (hipcc re-serialises the C++ form).
template <bool KMAJ, int W, int S> struct FragG;
template <int W, int S> struct FragG<true, W, S> {
  static constexpr int NREAD = W;
  f32x4 v[W];
  // base = byte address of (row = lane&31 of MFMA tile 0, k = 4*half)
  __device__ __forceinline__ void read(uint32_t base, int t) {
#pragma unroll
    for (int i = 0; i < W; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[i]) : "v"(base), "n"((32 * i * KS + 8 * t) * 4));
  }
  __device__ __forceinline__ float get(int i, int s) const { return v[i][s]; }
  __device__ __forceinline__ void tie() {
#pragma unroll
    for (int i = 0; i < W; ++i) asm volatile("" : "+v"(v[i]));
  }
};
// The per-step addresses are base + a compile-time constant: they travel in the instruction's 16-bit offset field, not in
// vector registers (ds_read2_b32's two 8-bit offsets cannot hold a row stride, hence two ds_read_b32 per step): no
// vector add per read in a rolled loop, no hoisted address register per read in an unrolled one.
template <int S> struct FragG<false, 2, S> {
  // Row strides that are a multiple of 64 floats (the unpadded LDS-DMA images: S = 64 or 128): ds_read2st64_b32 takes
  // two offsets in units of 256 bytes, i.e. two K ROWS of one column per instruction -- half the LDS read instructions
  // of the two-ds_read_b32 form, offsets still immediates (one extra base register for the second MFMA tile).
  static constexpr bool ST64 = S % 64 == 0;
  static constexpr int NREAD = ST64 ? 4 : 8;
  float v[4][2];    // [k step][tile]   (general form)
  f32x2 w[2][2];    // [tile][k-step pair] (ST64 form)
  // base = byte address of (k row = 4*half, column = lane&31 of MFMA tile 0); tile 1 is 32 floats further
  __device__ __forceinline__ void read(uint32_t base, int t) {
    if constexpr (ST64) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(w[i][h]) : "v"(base + 128u * i),
                       "n"((8 * t + 2 * h) * (S / 64)), "n"((8 * t + 2 * h + 1) * (S / 64)));
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[s][0]) : "v"(base), "n"((8 * t + s) * S * 4));
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[s][1]) : "v"(base), "n"((8 * t + s) * S * 4 + 128));
      }
    }
  }
  __device__ __forceinline__ float get(int i, int s) const {
    if constexpr (ST64) return (s & 1) ? w[i][s >> 1].y : w[i][s >> 1].x;
    else return v[s][i];
  }
  __device__ __forceinline__ void tie() {
    if constexpr (ST64) {
#pragma unroll
      for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(w[i][0]), "+v"(w[i][1]));
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(v[s][0]), "+v"(v[s][1]));
    }
  }
};
template <int S> struct FragG<false, 1, S> {
  static constexpr bool ST64 = S % 64 == 0;  // see FragG<false, 2, S>
  static constexpr int NREAD = ST64 ? 2 : 4;
  float v[4];
  f32x2 w[2];
  __device__ __forceinline__ void read(uint32_t base, int t) {
    if constexpr (ST64) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
        asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(w[h]) : "v"(base), "n"((8 * t + 2 * h) * (S / 64)),
                     "n"((8 * t + 2 * h + 1) * (S / 64)));
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[s]) : "v"(base), "n"((8 * t + s) * S * 4));
    }
  }
  __device__ __forceinline__ float get(int, int s) const {
    if constexpr (ST64) return (s & 1) ? w[s >> 1].y : w[s >> 1].x;
    else return v[s];
  }
  __device__ __forceinline__ void tie() {
    if constexpr (ST64) {
      asm volatile("" : "+v"(w[0]), "+v"(w[1]));
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(v[s]));
    }
  }
};
// ---- LDS-DMA loaders ----------------------------------------------------------------------------
// Timing-only ablations put the distance to the bare-MFMA rate on the register-staged
// global loads.  Here the steady-state tiles go global -> LDS directly (global_load_lds_dwordx4: no
// VGPR destination, no ds_write).  One instruction writes 1 KB = 8 rows x 128 B contiguously, so rows
// cannot be padded; the ds_read_b128 bank conflicts of a 128-B row stride are removed by an XOR
// swizzle of the 16-byte chunk index with (row >> 1) & 7, applied on the SOURCE address of the DMA and
// on the fragment read address (linear destination; cdna_hip_programming.md 5.4 rule 21).  An m/n-
// contiguous operand keeps its tile[k][128] image, unpadded (one instruction = 2 k rows x 512 B); its
// per-step ds_read2_b32 reads are conflict free at any row stride.
#ifndef BLM_GEMM_DMA
#define BLM_GEMM_DMA 1
#endif
template <int OP, int WTM, int WTN, bool SAMP, bool FAST>
constexpr bool use_dma() { return BLM_GEMM_DMA && !SAMP && FAST; }

template <int R>
__device__ __forceinline__ void r2s_kmaj_swz(float* tile, const float4 (&r)[R / 32]) {  // tile[row][32], chunk ^ ((row>>1)&7)
  const int t = threadIdx.x, kq = t & 7, rr = t >> 3;
#pragma unroll
  for (int j = 0; j < R / 32; ++j) {
    const int row = rr + 32 * j;
    *reinterpret_cast<float4*>(tile + row * 32 + 4 * (kq ^ ((row >> 1) & 7))) = r[j];
  }
}

// 16 bytes per lane, global -> LDS at lds_dst (wave uniform) + lane * 16; M0 is written in the same statement
__device__ __forceinline__ void glds16(const float* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// same with a wave-uniform 64-bit base in SGPRs and a 32-bit per-lane byte offset: half the address
// data per instruction and no per-tile VALU address arithmetic (the K advance goes into the scalar base)
__device__ __forceinline__ void glds16s(const float* sbase, uint32_t voff, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

template <int W> struct FragD {  // swizzled tile[row][32]: per-group byte addresses bt[t], MFMA tile i is 32 rows = 4096 B further
  static constexpr int NREAD = W;
  f32x4 v[W];
  __device__ __forceinline__ void read(const uint32_t (&bt)[4], int t) {
#pragma unroll
    for (int i = 0; i < W; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[i]) : "v"(bt[t]), "n"(4096 * i));
  }
  __device__ __forceinline__ float get(int i, int s) const { return v[i][s]; }
  __device__ __forceinline__ void tie() {
#pragma unroll
    for (int i = 0; i < W; ++i) asm volatile("" : "+v"(v[i]));
  }
};

template <int N>  // the counter has 4 bits: more than 15 reads in flight -> wait for "at most 15" (one read early: LDS returns in order)
__device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N > 15 ? 15 : N)); }

// ---- opt-in split-bf16 arithmetic (NOT the default) -------------------------
// x = hi + lo with hi = bf16_rne(x), lo = bf16_rne(x - hi): eight fp32 values of a lane -> the two bf16x8
// operands of v_mfma_f32_32x32x16_bf16.  A.B ~ hi.hi + hi.lo + lo.hi (lo.lo dropped): 4.5e-6 max relative
// error at K = 4096 against 3.6e-7 for the fp32 MFMA, three 32-cycle MFMAs per 32x32x16 against eight 64-cycle
// ones.  v_cvt_pk_bf16_f32 converts two values per instruction: 3 VALU operations per element.
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4v;
__device__ __forceinline__ void split_bf16x8(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
  u32x4v h, l;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x2 v = {x[2 * q], x[2 * q + 1]};
    const uint32_t hu = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
    const f32x2 r = {v.x - __uint_as_float(hu << 16), v.y - __uint_as_float(hu & 0xFFFF0000u)};
    h[q] = hu;
    l[q] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2));
  }
  hi = __builtin_bit_cast(bf16x8, h);
  lo = __builtin_bit_cast(bf16x8, l);
}
// three parts: hi + mid + lo holds all 24 mantissa bits of x (EXACT representation); six of the nine part
// products (everything down to 2^-24 of the product) make an fp32-accurate multiply on the bf16 matrix cores
__device__ __forceinline__ void split_bf16x8_3(const float (&x)[8], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
  u32x4v h, m, l;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x2 v = {x[2 * q], x[2 * q + 1]};
    const bf16x2 hb = __builtin_convertvector(v, bf16x2);
    const f32x2 r = {v.x - (float)hb[0], v.y - (float)hb[1]};
    const bf16x2 mb = __builtin_convertvector(r, bf16x2);
    const f32x2 r2 = {r.x - (float)mb[0], r.y - (float)mb[1]};
    h[q] = __builtin_bit_cast(uint32_t, hb);
    m[q] = __builtin_bit_cast(uint32_t, mb);
    l[q] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r2, bf16x2));
  }
  hi = __builtin_bit_cast(bf16x8, h);
  mid = __builtin_bit_cast(bf16x8, m);
  lo = __builtin_bit_cast(bf16x8, l);
}

// Dropout keep factors for the four rows base_row + {0,1,2,3}*row_stride at this lane's column: the four
// lanes of a quad own four consecutive columns = one Philox block per row, so lane k generates the
// block of row base_row + k and the quad exchanges words by DPP broadcasts (1 Philox call per
// 4 elements instead of 4).  Needs N % 4 == 0 and the quad's first column % 4 == 0.
__device__ __forceinline__ void gemm_keep_quad(const GemmP& p, int base_row, int row_stride, int col, float (&keep)[4]) {
  const int k = threadIdx.x & 3;
  const int m = base_row + k * row_stride;
  const int row = m / p.drop_B, b = m - row * p.drop_B;
  const uint64_t g = ((uint64_t)row * p.drop_global_cols + (uint64_t)(p.drop_col_offset + b)) * (uint64_t)p.N +
                     (uint64_t)(col & ~3);
  const u32x4 u = philox4x32_10_rolled((uint32_t)(g >> 2), (uint32_t)(g >> 34), p.drop_rng.stream, p.drop_rng.step,
                                       (uint32_t)p.drop_rng.seed, (uint32_t)(p.drop_rng.seed >> 32));
#define BLM_QUAD_BCAST(x, j) (uint32_t) __builtin_amdgcn_mov_dpp((int)(x), (j) * 0x55, 0xF, 0xF, true)
#define BLM_KEEP_FROM(j)                                                                                      \
  {                                                                                                           \
    const uint32_t w0 = BLM_QUAD_BCAST(u.x, j), w1 = BLM_QUAD_BCAST(u.y, j), w2 = BLM_QUAD_BCAST(u.z, j),     \
                   w3 = BLM_QUAD_BCAST(u.w, j);                                                               \
    const uint32_t w = k == 0 ? w0 : (k == 1 ? w1 : (k == 2 ? w2 : w3));                                      \
    keep[j] = w >= p.drop_thr ? p.drop_inv_keep : 0.f;                                                        \
  }
  BLM_KEEP_FROM(0) BLM_KEEP_FROM(1) BLM_KEEP_FROM(2) BLM_KEEP_FROM(3)
#undef BLM_KEEP_FROM
#undef BLM_QUAD_BCAST
}

// Keep factors of the two adjacent columns (c0, c0+1), c0 even, of row m: one Philox block.
__device__ __forceinline__ void gemm_keep_pair(const GemmP& p, int m, int c0, float& k0, float& k1) {
  const int row = m / p.drop_B, b = m - row * p.drop_B;
  const uint64_t g = ((uint64_t)row * p.drop_global_cols + (uint64_t)(p.drop_col_offset + b)) * (uint64_t)p.N + (uint64_t)c0;
  const u32x4 u = philox4x32_10_rolled((uint32_t)(g >> 2), (uint32_t)(g >> 34), p.drop_rng.stream, p.drop_rng.step,
                                       (uint32_t)p.drop_rng.seed, (uint32_t)(p.drop_rng.seed >> 32));
  const bool hi = g & 2;
  k0 = (hi ? u.z : u.x) >= p.drop_thr ? p.drop_inv_keep : 0.f;
  k1 = (hi ? u.w : u.y) >= p.drop_thr ? p.drop_inv_keep : 0.f;
}

// Bayesian wgrad: eps of the four rows base_row + {0,1,2,3}*row_stride at this lane's column, same quad
// scheme as gemm_keep_quad: lane k draws the Philox block (4 normals = 4 consecutive columns) of row k,
// the quad exchanges them by DPP broadcasts -- one Philox call + two Box-Muller per 4 elements instead
// of one call per element.  Needs N % 4 == 0 (then element index & 3 == column & 3 == lane & 3).
__device__ __forceinline__ void gemm_eps_quad(const GemmP& p, int base_row, int row_stride, int col, float (&eps)[4]) {
  const int k = threadIdx.x & 3;
  const int rel = min(max(base_row + k * row_stride - p.vc.row_lo, 0), p.vc.srows - 1);
  const uint64_t si = (uint64_t)rel * (uint64_t)p.N + (uint64_t)(min(col, p.N - 1) & ~3);
  const u32x4 u = philox4x32_10_rolled((uint32_t)(si >> 2), (uint32_t)(si >> 34), p.vc.rng.stream, p.vc.rng.step,
                                       (uint32_t)p.vc.rng.seed, (uint32_t)(p.vc.rng.seed >> 32));
  float z0, z1, z2, z3;
  box_muller(u.x, u.y, z0, z1);
  box_muller(u.z, u.w, z2, z3);
#define BLM_QUAD_BCASTF(x, j) __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), (j) * 0x55, 0xF, 0xF, true))
#define BLM_EPS_FROM(j)                                                                                   \
  {                                                                                                       \
    const float w0 = BLM_QUAD_BCASTF(z0, j), w1 = BLM_QUAD_BCASTF(z1, j), w2 = BLM_QUAD_BCASTF(z2, j),     \
                w3 = BLM_QUAD_BCASTF(z3, j);                                                              \
    eps[j] = k == 0 ? w0 : (k == 1 ? w1 : (k == 2 ? w2 : w3));                                            \
  }
  BLM_EPS_FROM(0) BLM_EPS_FROM(1) BLM_EPS_FROM(2) BLM_EPS_FROM(3)
#undef BLM_EPS_FROM
#undef BLM_QUAD_BCASTF
}

// C/D map of the 32x32 MFMA: column = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).  A wave that owns
// two MFMA tiles along a dimension INTERLEAVES them (tile t holds matrix rows/cols 2*x + t), so the
// two operand values a lane needs per k-step are adjacent in LDS (one ds_read_b64) and a lane's two
// output columns are adjacent in C.
template <int EPI>
__device__ __forceinline__ void epi_elem(const GemmP& p, bool accum, int row, int col, float acc, float bias, float keep, bool kl_on = true, bool atomic = false) {
  const long o = (long)row * p.ldc + col;
  float v = p.alpha * acc;
  if constexpr (EPI == BLM_EPI_BIAS) {
    if (kl_on) v += bias;  // K slices: the first one carries the bias
  } else if constexpr (EPI == BLM_EPI_BIAS_GELU) {
    v += bias;
    float cdf, e;
    gelu_parts(v, cdf, e);
    if (p.aux) p.aux[o] = (cdf + v * 0.3989422804014327f * e) * keep;  // d(output)/dz: GELU'(z) * dropout keep
    v = v * cdf * keep;
  } else if constexpr (EPI == BLM_EPI_MUL_DGELU) {
    v *= p.aux[o];
  } else if constexpr (EPI == BLM_EPI_GP_MIX) {
    v += bias;
    if (p.aux) p.aux[o] = v;
    v = gp_mix(v, p.coef, p.N, col) * keep;
  } else if constexpr (EPI == BLM_EPI_MUL_DGP_MIX) {
    v *= keep;
    if (p.C2) p.C2[o] = v;
    v *= dgp_mix(p.aux[o], p.coef, p.N, col);
  } else if constexpr (EPI == BLM_EPI_BAYES_WGRAD) {
    const float dW = v;
    const int rel = row - p.vc.row_lo;
    if ((unsigned)rel < (unsigned)p.vc.srows) {
      const long si = (long)rel * p.N + col;
      const float sig = __expf(p.vc.lgstd[si]);
      float e;
      if (p.vc.eps) e = p.vc.eps[si];
      else if (p.eps_quad) e = keep;  // drawn per quad by the caller (gemm_eps_quad)
      else e = philox_normal1_rolled(p.vc.rng, (uint64_t)si);
      // both gradients are linear in dW, so K slices add up through atomics; the KL terms (no dW in
      // them) are contributed by the first slice only
      const float klw = kl_on ? p.kl_lambda * p.kl_inv_n : 0.f;
      const float g2 = dW * e * sig + klw * (sig * sig - 1.0f);
      if (atomic) atomicAdd(p.C2 + si, g2);
      else p.C2[si] = accum ? p.C2[si] + g2 : g2;
      v = dW + klw * p.wg_mu[o];
    }
  }
  if constexpr (EPI == BLM_EPI_NONE || EPI == BLM_EPI_BIAS || EPI == BLM_EPI_BAYES_WGRAD) {
    if (atomic) { atomicAdd(p.C + o, v); return; }
  }
  p.C[o] = accum ? p.C[o] + v : v;
}

template <int EPI, int WTM, int WTN>
__device__ __forceinline__ void epilogue(const GemmP& p, f32x16 (&acc)[WTM][WTN], int m0, int n0, int wm, int wn,
                                         int li, int lh, bool kl_on = true, bool atomic = false) {
  constexpr bool DROP = (EPI == BLM_EPI_BIAS_GELU || EPI == BLM_EPI_GP_MIX || EPI == BLM_EPI_MUL_DGP_MIX);
  constexpr bool BIAS = (EPI == BLM_EPI_BIAS || EPI == BLM_EPI_BIAS_GELU || EPI == BLM_EPI_GP_MIX);
  const bool accum = p.flags & BLM_GEMM_ACCUMULATE;
  int col[WTN];
  float bias[WTN];
#pragma unroll
  for (int j = 0; j < WTN; ++j) {
    col[j] = n0 + wn * (32 * WTN) + ((INTERLEAVE && WTN == 2) ? 2 * li + j : 32 * j + li);
    bias[j] = 0.f;
    if constexpr (BIAS) bias[j] = col[j] < p.N ? p.bias[col[j]] : 0.f;
  }
  // plain/bias epilogues with two adjacent columns per lane store them as one 8-byte access
  const bool pair_store = INTERLEAVE && WTN == 2 && (EPI == BLM_EPI_NONE || EPI == BLM_EPI_BIAS) && !atomic && (p.ldc % 2 == 0) &&
                          ((reinterpret_cast<uintptr_t>(p.C) & 7) == 0);
#pragma unroll
  for (int i = 0; i < WTM; ++i) {
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const int mrow0 = 8 * rq + 4 * lh;  // first of the 4 MFMA rows this lane holds in registers 4rq..4rq+3
      const int row0 = m0 + wm * (32 * WTM) + ((INTERLEAVE && WTM == 2) ? 2 * mrow0 + i : 32 * i + mrow0);
      constexpr int RS = (INTERLEAVE && WTM == 2) ? 2 : 1;  // matrix-row stride between consecutive MFMA rows
      float keep[4][WTN];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int j = 0; j < WTN; ++j) keep[rr][j] = 1.f;
      if constexpr (EPI == BLM_EPI_BAYES_WGRAD) {
        if (p.eps_quad) {  // wave-uniform; every lane runs it (the quad exchange needs all lanes)
#pragma unroll
          for (int j = 0; j < WTN; ++j) {
            float e4[4];
            gemm_eps_quad(p, row0, RS, col[j], e4);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) keep[rr][j] = e4[rr];
          }
        }
      }
      if constexpr (DROP) {
        if (p.drop_on) {  // wave-uniform; every lane runs it (the quad exchange needs all lanes)
          if (INTERLEAVE && WTN == 2 && p.drop_quad) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) gemm_keep_pair(p, row0 + rr * RS, min(col[0], p.N - 2), keep[rr][0], keep[rr][WTN - 1]);
          } else if (!(INTERLEAVE && WTN == 2) && p.drop_quad) {  // 4 adjacent lanes = 4 adjacent columns
#pragma unroll
            for (int j = 0; j < WTN; ++j) {
              float k4[4];
              gemm_keep_quad(p, row0, RS, col[j], k4);
#pragma unroll
              for (int rr = 0; rr < 4; ++rr) keep[rr][j] = k4[rr];
            }
          } else {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
#pragma unroll
              for (int j = 0; j < WTN; ++j) keep[rr][j] = gemm_keep(p, row0 + rr * RS, min(col[j], p.N - 1));
          }
        }
      }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int row = row0 + rr * RS;
        const int r = 4 * rq + rr;
        if (row >= p.M) continue;
        if constexpr (WTN == 2) {
          if (pair_store && col[1] < p.N) {
            float2* dst = reinterpret_cast<float2*>(p.C + (long)row * p.ldc + col[0]);
            float2 v = make_float2(p.alpha * acc[i][0][r] + bias[0], p.alpha * acc[i][1][r] + bias[1]);
            if (accum) { const float2 old = *dst; v.x += old.x; v.y += old.y; }
            *dst = v;
            continue;
          }
        }
#pragma unroll
        for (int j = 0; j < WTN; ++j)
          if (col[j] < p.N) epi_elem<EPI>(p, accum, row, col[j], acc[i][j][r], bias[j], keep[rr][j], kl_on, atomic);
      }
    }
  }
  BLM_PROF_WG_END();
}

// Row-wise epilogue through LDS: the accumulator tile (columns on lanes, rows in registers) is
// written to the now idle staging buffer 64 rows at a time and read back row-major, so every lane
// handles 4 consecutive columns: 16-byte loads of bias/aux, one Philox block per lane for the
// dropout mask, 16-byte stores of C (4x fewer memory instructions than the register-layout walk).
// (Measured and not kept: the non-temporal hint on these stores -- no change; tools/gemm_epi_probe.py on a -DBLM_GEMM_LIFE
// build shows the stores cost nothing at all: 8192 x 4096 x 512 runs 256 us with them compiled out, 258 us as shipped.)
__device__ __forceinline__ void store4(float* dst, const float4& v) { *reinterpret_cast<float4*>(dst) = v; }

template <int EPI, int WTM, int WTN, int WGN = 2>
__device__ __forceinline__ void epilogue_rows(const GemmP& p, f32x16 (&acc)[WTM][WTN], float* stage, int m0, int n0,
                                              int wm, int wn, int li, int lh) {
  constexpr int BN = 32 * WTN * WGN, SS = BN + 4;
  constexpr int NP = WTM == 2 ? 2 : 1;
  constexpr int TPRW = BN / 4, RPS = (128 * WGN) / TPRW;
  const int t = threadIdx.x, c4 = 4 * (t % TPRW);
  const int col = n0 + c4;
  const bool accum = p.flags & BLM_GEMM_ACCUMULATE;
  float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (EPI == BLM_EPI_BIAS || EPI == BLM_EPI_BIAS_GELU || EPI == BLM_EPI_GP_MIX || EPI == BLM_EPI_CE_PART)
    if (col < p.N && p.bias) bias = *reinterpret_cast<const float4*>(p.bias + col);
#pragma unroll
  for (int pass = 0; pass < NP; ++pass) {
    __syncthreads();
    if (WTM == 1 || wm == pass) {
#pragma unroll
      for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int lrow = (WTM == 2 ? 32 * i : 32 * wm) + (r & 3) + 8 * (r >> 2) + 4 * lh;
            stage[lrow * SS + wn * (32 * WTN) + 32 * j + li] = p.alpha * acc[i][j][r];
          }
    }
    __syncthreads();
    if constexpr (EPI == BLM_EPI_CE_PART) {
      // every lane of a row takes part in the row's reductions (lanes past N carry -inf): TPRW consecutive lanes hold one row
      const bool valid = col < p.N;  // N % 4 == 0: a lane's four columns are all inside or all outside
      for (int lr = t / TPRW; lr < 64; lr += RPS) {
        const int row = m0 + 64 * pass + lr;
        if (row >= p.M) break;  // uniform over the lanes of a row
        float4 v = *reinterpret_cast<const float4*>(stage + lr * SS + c4);
        v.x += bias.x; v.y += bias.y; v.z += bias.z; v.w += bias.w;
        float m = valid ? fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)) : -INFINITY;
#pragma unroll
        for (int o = TPRW / 2; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        float sum = valid ? (__expf(v.x - m) + __expf(v.y - m)) + (__expf(v.z - m) + __expf(v.w - m)) : 0.f;
#pragma unroll
        for (int o = TPRW / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        if (t % TPRW == 0) {
          float* d = p.aux + ((long)row * p.gn + n0 / BN) * 2;
          d[0] = m;
          d[1] = sum;
        }
        const long long tg = p.ce_tgt[row];
        if (valid && tg >= col && tg < col + 4) p.ce_tlogit[row] = tg == col ? v.x : (tg == col + 1 ? v.y : (tg == col + 2 ? v.z : v.w));
      }
    } else if (col < p.N) {
      auto keep4 = [&](int row) {  // dropout keep factors of this lane's 4 consecutive columns: one Philox block
        float4 kp = make_float4(1.f, 1.f, 1.f, 1.f);
        if (p.drop_on) {
          const int rr = row / p.drop_B, b = row - rr * p.drop_B;
          const uint64_t g = ((uint64_t)rr * p.drop_global_cols + (uint64_t)(p.drop_col_offset + b)) * (uint64_t)p.N + (uint64_t)col;
          // unrolled, inlined rounds: 35 % fewer vector instructions than the rolled call (register moves, key adds), and
          // every one of them is paid in matrix time
          const u32x4 u = philox4x32_10((uint32_t)(g >> 2), (uint32_t)(g >> 34), p.drop_rng.stream, p.drop_rng.step,
                                        (uint32_t)p.drop_rng.seed, (uint32_t)(p.drop_rng.seed >> 32));
          kp = make_float4(u.x >= p.drop_thr ? p.drop_inv_keep : 0.f, u.y >= p.drop_thr ? p.drop_inv_keep : 0.f,
                           u.z >= p.drop_thr ? p.drop_inv_keep : 0.f, u.w >= p.drop_thr ? p.drop_inv_keep : 0.f);
        }
        return kp;
      };
      auto row_body = [&](int lr, const float4& a) {
        const int row = m0 + 64 * pass + lr;
        float4 v = *reinterpret_cast<const float4*>(stage + lr * SS + c4);
        const long o = (long)row * p.ldc + col;
        if constexpr (EPI == BLM_EPI_BIAS) {
          v.x += bias.x; v.y += bias.y; v.z += bias.z; v.w += bias.w;
        } else if constexpr (EPI == BLM_EPI_GP_MIX) {  // z = acc + bias kept for backward; out = mixture(z) * keep
          const float4 kp = keep4(row);
          v.x += bias.x; v.y += bias.y; v.z += bias.z; v.w += bias.w;
          if (p.aux) store4(p.aux + o, v);
          v.x = gp_mix(v.x, p.coef, p.N, col) * kp.x;
          v.y = gp_mix(v.y, p.coef, p.N, col + 1) * kp.y;
          v.z = gp_mix(v.z, p.coef, p.N, col + 2) * kp.z;
          v.w = gp_mix(v.w, p.coef, p.N, col + 3) * kp.w;
        } else if constexpr (EPI == BLM_EPI_MUL_DGP_MIX) {  // a = z of the forward; C2 = grad w.r.t. the mixture value
          const float4 kp = keep4(row);
          v.x *= kp.x; v.y *= kp.y; v.z *= kp.z; v.w *= kp.w;
          if (p.C2) store4(p.C2 + o, v);
          v.x *= dgp_mix(a.x, p.coef, p.N, col);
          v.y *= dgp_mix(a.y, p.coef, p.N, col + 1);
          v.z *= dgp_mix(a.z, p.coef, p.N, col + 2);
          v.w *= dgp_mix(a.w, p.coef, p.N, col + 3);
        } else if constexpr (EPI == BLM_EPI_BIAS_GELU) {
          const float4 kp = keep4(row);
          float4 d;
          {  // two values per packed instruction (blm_device.h gelu_parts2)
            const blm_f2 z0 = (blm_f2){v.x, v.y} + (blm_f2){bias.x, bias.y}, z1 = (blm_f2){v.z, v.w} + (blm_f2){bias.z, bias.w};
            const blm_f2 k0 = {kp.x, kp.y}, k1 = {kp.z, kp.w};
            blm_f2 c0, e0, c1, e1;
            gelu_parts2(z0, c0, e0);
            gelu_parts2(z1, c1, e1);
            const blm_f2 d0 = __builtin_elementwise_fma(z0 * 0.3989422804014327f, e0, c0) * k0, y0 = (z0 * c0) * k0;
            const blm_f2 d1 = __builtin_elementwise_fma(z1 * 0.3989422804014327f, e1, c1) * k1, y1 = (z1 * c1) * k1;
            d = make_float4(d0.x, d0.y, d1.x, d1.y);
            v = make_float4(y0.x, y0.y, y1.x, y1.y);
          }
          if (p.aux) store4(p.aux + o, d);
        } else if constexpr (EPI == BLM_EPI_MUL_DGELU) {
          v.x *= a.x; v.y *= a.y; v.z *= a.z; v.w *= a.w;
        }
        float4* dst = reinterpret_cast<float4*>(p.C + o);
        if (accum) { const float4 old = *dst; v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w; }
#ifdef BLM_GEMM_LIFE
        if (blm_dbg_store == 1) { if (v.x == 1.2345e-30f) store4(p.C + o, v); return; }
        if (blm_dbg_store == 2) { store4(p.C + (long)(lr + 64 * pass) * p.ldc + c4, v); return; }
#endif
        store4(p.C + o, v);
      };
      if constexpr (EPI == BLM_EPI_MUL_DGELU || EPI == BLM_EPI_MUL_DGP_MIX) {
        // the second factor of all this pass's rows is requested in one go (a load inside the row loop
        // costs one full memory latency per row: 8 rows x 2 passes per tile)
        constexpr int NIT = 64 / RPS;
        float4 ax[NIT];
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
          const int row = m0 + 64 * pass + t / TPRW + RPS * i;
          ax[i] = row < p.M ? *reinterpret_cast<const float4*>(p.aux + (long)row * p.ldc + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
          const int lr = t / TPRW + RPS * i;
          if (m0 + 64 * pass + lr < p.M) row_body(lr, ax[i]);
        }
      } else {
        const float4 none = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int lr = t / TPRW; lr < 64; lr += RPS) {
          if (m0 + 64 * pass + lr >= p.M) break;
          row_body(lr, none);
        }
      }
    }
  }
  BLM_PROF_WG_END();
}

// FAST: every operand 16-B aligned with a leading dimension and contiguous extent that are
// multiples of 4.  Full K tiles are then fetched by branch-free float4 loads through per-thread
// pointers set up once (rows/cols outside the matrix are clamped, not zeroed: they only feed
// C elements that are never stored); only a K tail tile goes through the guarded loaders.
// WGN = columns of the wave grid: 2 (4 waves as 2 x 2, the tiles 11 / 12 / 21 / 22) or 4 (8 waves as 2 x 4: tile 28 = 128 x 128 run by
// eight waves of 64 x 32 -- the per-wave shape and the two waves per SIMD of the 128 x 64 tile inside ONE barrier domain, so the
// halves cannot drift apart along K and their shared B panel is staged once; LDS-DMA path, whole K tiles only: host-checked).
// Measured and not kept: 128 x 256 on eight waves of 64 x 64 (WTM = WTN = 2, WGN = 4): +2-5 % stand-alone on the 8192 x 4096 x 512
// products, never the in-situ winner of any shape of the ten tuned workloads.
template <int OP, int WTM, int WTN, bool SAMP, bool FAST, int SPLIT = 0, int WGN = 2>  // SPLIT: 0 fp32 MFMA, 3 / 6 opt-in bf16 part products
__global__ __launch_bounds__(128 * WGN, WGN == 2 ? 2 : 1) void gemm_f32_kernel(const GemmP p) {
  constexpr int BM = 64 * WTM, BN = 32 * WTN * WGN, NT = 128 * WGN;
  constexpr bool A_KMAJ = (OP != BLM_GEMM_TN), B_KMAJ = (OP == BLM_GEMM_NT);
  // k-contiguous sources are transposed on the LDS write: stride BM+2 keeps that scatter at most
  // 2-way conflicting (free for ds_write_b32) AND even, so the interleaved two-tile operand read is an
  // aligned ds_read_b64; single-tile waves use BM+1 (conflict-free).  m/n-contiguous: BM+4 (b128 rows).
  // m/n-contiguous operands: image tile[k][BM+4]; k-contiguous ones: tile[row][KS]
  constexpr bool DMA = use_dma<OP, WTM, WTN, SAMP, FAST>();
  constexpr int SA = BM + (DMA ? 0 : 4), SB = BN + (DMA ? 0 : 4);
  constexpr int KSX = DMA ? 32 : KS;  // k-contiguous row stride: unpadded + swizzled under LDS-DMA
  constexpr int TA = A_KMAJ ? BM * KSX : BK * SA, TB = B_KMAJ ? BN * KSX : BK * SB;  // floats per staged tile
  constexpr int NA = BM / 32, NB = BN / 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const As = smem;
  float* const Bs = smem + 2 * TA;

  // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so give each XCD a
  // contiguous run of tile ids.
  // Slices: with tail_from = 0 block b computes slice b / nb of tile b % nb; otherwise the first tail_from tiles are whole
  // (ks = 0, all of K, plain stores) and only the nb - tail_from tiles of the last, partly filled round are sliced.
  const int nb = p.gm * p.gn;
  const bool whole = (int)blockIdx.x < p.tail_from;
  const int ntail = nb - p.tail_from, rtail = (int)blockIdx.x - p.tail_from;
  const int bid = whole ? (int)blockIdx.x : p.tail_from + rtail % ntail, ks = whole ? 0 : rtail / ntail;
  const bool atomic = p.atomic && !whole;
  const int q = nb >> 3, rem = nb & 7, xcd = bid & 7;
  const int id = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
  // Inside the run, groups of 8 row blocks with the row block fastest: the 64 workgroups an XCD holds at a time are an
  // 8 x 8 block of tiles -- 8 A panels + 8 B panels of unique operand rows per K step, every line wanted by 8
  // workgroups at once -- whatever gn is (n fastest made a round 2 x 32 tiles at gn = 32 and 1 x 64 at the decoder's
  // gn = 258, where every row block streamed the whole weight again: 4.3 GB of reads for 84 MB of operands).
  // (gn <= 8 keeps n fastest: the same 8 x 8 set per round, and measured 20 % fewer fetched bytes than the row-block-fastest
  // placement of the same tiles on the CUs -- PMC FETCH_SIZE, roofline shape.)
  constexpr int GROUP = 8;
  int mt, nt;
  if (p.gn <= GROUP) {
    mt = id / p.gn;
    nt = id - mt * p.gn;
  } else {
    const int per = GROUP * p.gn, grp = id / per, first = grp * GROUP, gsz = min(p.gm - first, GROUP), in = id - grp * per;
    mt = first + in % gsz;
    nt = in / gsz;
  }
  const int m0 = mt * BM, n0 = nt * BN;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WGN, wn = wave % WGN, li = lane & 31, lh = lane >> 5;
#ifdef BLM_GEMM_LIFE
  if (t == 0 && blockIdx.x < 8192) {
    blm_wg_life[4 * blockIdx.x] = wall_clock64();
    blm_wg_life[4 * blockIdx.x + 2] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |            // HW_REG_HW_ID
                                      ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);  // HW_REG_XCC_ID
  }
#endif

  float4 ra[NA], rb[NB], rl[SAMP ? NB : 1];
  const float* pa[FAST ? NA : 1];
  const float* pb[FAST ? NB : 1];
  const float* pl[(FAST && SAMP) ? NB : 1];
  bool in_slice[(FAST && SAMP) ? NB : 1];
  long stepA = 0, stepB = 0, stepL = 0;  // element advance per K tile

  if constexpr (FAST) {
    if constexpr (A_KMAJ) {
#pragma unroll
      for (int j = 0; j < NA; ++j) pa[j] = p.A + (long)min(m0 + (t >> 3) + 32 * j, p.M - 1) * p.lda + 4 * (t & 7);
      stepA = BK;
    } else {
      constexpr int TPR = BM / 4, RPP = 256 / TPR;
#pragma unroll
      for (int j = 0; j < NA; ++j) pa[j] = p.A + (long)(t / TPR + RPP * j) * p.lda + min(m0 + 4 * (t % TPR), p.M - 4);
      stepA = (long)BK * p.lda;
    }
    if constexpr (B_KMAJ) {
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int n = min(n0 + (t >> 3) + 32 * j, p.N - 1);
        pb[j] = p.B + (long)n * p.ldb + 4 * (t & 7);
        if constexpr (SAMP) {
          const int rel = n - p.vb.row_lo;
          in_slice[j] = (unsigned)rel < (unsigned)p.vb.srows;
          pl[j] = p.vb.lgstd + (long)min(max(rel, 0), p.vb.srows - 1) * p.vb_cols + 4 * (t & 7);
        }
      }
      stepB = BK;
      stepL = BK;
    } else {
      constexpr int TPR = BN / 4, RPP = 256 / TPR;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int c = min(n0 + 4 * (t % TPR), p.N - 4);
        pb[j] = p.B + (long)(t / TPR + RPP * j) * p.ldb + c;
        if constexpr (SAMP) pl[j] = p.vb.lgstd + c;  // row part added per tile (depends on k)
      }
      stepB = (long)BK * p.ldb;
    }
  }

  float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool do_cs = !A_KMAJ && p.colsum_a != nullptr && n0 == 0;  // one N-tile column of blocks does it

  auto fetch_fast = [&](int kt) {
    if constexpr (FAST) {
#pragma unroll
      for (int j = 0; j < NA; ++j) ra[j] = *reinterpret_cast<const float4*>(pa[j] + kt * stepA);
#pragma unroll
      for (int j = 0; j < NB; ++j) rb[j] = *reinterpret_cast<const float4*>(pb[j] + kt * stepB);
      if constexpr (SAMP) {
        if constexpr (B_KMAJ) {
#pragma unroll
          for (int j = 0; j < NB; ++j) rl[j] = *reinterpret_cast<const float4*>(pl[j] + kt * stepL);
        } else {
          constexpr int TPR = BN / 4, RPP = 256 / TPR;
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            const int rel = kt * BK + t / TPR + RPP * j - p.vb.row_lo;
            in_slice[j] = (unsigned)rel < (unsigned)p.vb.srows;
            rl[j] = *reinterpret_cast<const float4*>(pl[j] + (long)min(max(rel, 0), p.vb.srows - 1) * p.vb_cols);
          }
        }
      }
    }
  };
  auto fetch_slow = [&](int k0) {
    const bool av = p.a_vec, bv = p.b_vec;
    if constexpr (A_KMAJ) g2r_kmaj<BM>(p.A, p.lda, m0, p.M, k0, p.K, av, ra);
    else g2r_nmaj<BM>(p.A, p.lda, k0, p.K, m0, p.M, av, ra);
    if constexpr (B_KMAJ) g2r_kmaj<BN>(p.B, p.ldb, n0, p.N, k0, p.K, bv, rb);
    else g2r_nmaj<BN>(p.B, p.ldb, k0, p.K, n0, p.N, bv, rb);
    if constexpr (SAMP) {
      if constexpr (B_KMAJ) g2r_kmaj<BN>(p.vb.lgstd, p.vb_cols, n0 - p.vb.row_lo, p.vb.srows, k0, p.K, true, rl);
      else g2r_nmaj<BN>(p.vb.lgstd, p.vb_cols, k0 - p.vb.row_lo, p.vb.srows, n0, p.N, true, rl);
    }
  };
  // registers -> LDS (with the variational transform W = mu + exp(lgstd)*eps on the B operand)
  auto stash = [&](int buf, int k0, bool fast) {
    float* At = As + buf * TA;
    float* Bt = Bs + buf * TB;
    if constexpr (SAMP) {
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        int srow, scol;
        if constexpr (B_KMAJ) { srow = n0 + (t >> 3) + 32 * j; scol = k0 + 4 * (t & 7); }
        else { constexpr int TPR = BN / 4, RPP = 256 / TPR; srow = k0 + t / TPR + RPP * j; scol = n0 + 4 * (t % TPR); }
        if constexpr (FAST) {
          if (fast) {  // clamped fetch: element coordinates follow the clamped address
            if constexpr (B_KMAJ) srow = min(srow, p.N - 1); else scol = min(scol, p.N - 4);
            if (in_slice[j]) rb[j] = sample4(rb[j], rl[j], p.vb, p.vb_cols, srow, scol);
            continue;
          }
        }
        rb[j] = sample4(rb[j], rl[j], p.vb, p.vb_cols, srow, scol);
      }
    }
    if constexpr (!A_KMAJ && !DMA) {  // wgrad: A = dY[k][m]; its column sums are the bias gradient
      if (do_cs) {
#pragma unroll
        for (int j = 0; j < NA; ++j) { cs.x += ra[j].x; cs.y += ra[j].y; cs.z += ra[j].z; cs.w += ra[j].w; }
      }
    }
    if constexpr (DMA) {
      if constexpr (A_KMAJ) r2s_kmaj_swz<BM>(At, ra); else r2s_nmaj<BM, SA>(At, ra);
      if constexpr (B_KMAJ) r2s_kmaj_swz<BN>(Bt, rb); else r2s_nmaj<BN, SB>(Bt, rb);
    } else {
      if constexpr (A_KMAJ) r2s_kmaj<BM>(At, ra); else r2s_nmaj<BM, SA>(At, ra);
      if constexpr (B_KMAJ) r2s_kmaj<BN>(Bt, rb); else r2s_nmaj<BN, SB>(Bt, rb);
    }
  };

  f32x16 acc[WTM][WTN];
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j) acc[i][j] = (f32x16)(0.f);

  // One K tile = 4 groups of 4 MFMA k-steps.  The fragments of group t+1 are requested from LDS BEFORE
  // the 4 x WTM x WTN MFMAs of group t are issued (two register sets), so the ds_read latency sits
  // under >= 1024 cycles of matrix work.  Reads and their counted lgkmcnt wait are inline asm
  // (cdna_hip_programming.md 5.7): after the wait every fragment register of the group is tied with an
  // empty "+v" statement and a sched_barrier(0) keeps the MFMAs below it.
  // `mid` runs once in the middle of the tile (after the second fragment group's wait): the deep-
  // prefetch loop writes the NEXT tile to the other LDS buffer there, so that the write latency sits
  // under the remaining 32 MFMA steps and the end-of-tile barrier finds lgkmcnt already at 0
  auto compute = [&](int cur, auto&& mid) {
    using FA = FragG<A_KMAJ, WTM, SA>;
    using FB = FragG<B_KMAJ, WTN, SB>;
    const uint32_t a_base = A_KMAJ ? lds_u32(As + cur * TA + (wm * (32 * WTM) + li) * KS + 4 * lh)
                                   : lds_u32(As + cur * TA + (4 * lh) * SA + wm * (32 * WTM) + li);
    const uint32_t b_base = B_KMAJ ? lds_u32(Bs + cur * TB + (wn * (32 * WTN) + li) * KS + 4 * lh)
                                   : lds_u32(Bs + cur * TB + (4 * lh) * SB + wn * (32 * WTN) + li);
    FA a[2];
    FB b[2];
    a[0].read(a_base, 0);
    b[0].read(b_base, 0);
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
      if (t4 + 1 < 4) {
        a[(t4 + 1) & 1].read(a_base, t4 + 1);
        b[(t4 + 1) & 1].read(b_base, t4 + 1);
        wait_lgkm<FA::NREAD + FB::NREAD>();  // all but the reads just issued have landed
      } else {
        wait_lgkm<0>();
      }
      a[t4 & 1].tie();
      b[t4 & 1].tie();
      if (t4 == 2) mid();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int i = 0; i < WTM; ++i)
#pragma unroll
          for (int j = 0; j < WTN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t4 & 1].get(i, s4), b[t4 & 1].get(j, s4), acc[i][j], 0, 0, 0);
    }
  };

  // LDS-DMA path: fragment addresses of the swizzled tiles and the per-wave DMA sources
  const int swz = (li >> 1) & 7;  // (row >> 1) & 7 of this lane's MFMA row (same for both 32-row tiles)
  float cs1 = 0.f;                // LDS-DMA wgrad: this thread's column of the bias gradient (threads 0..BM-1)
  auto compute_dma = [&](int cur) {
    if constexpr (DMA) {
      if constexpr (!A_KMAJ) {
        if (do_cs && t < BM) {  // column sums of the A tile (= dY tile) straight from LDS
          const float* col = As + cur * TA + t;
#pragma unroll 8
          for (int k = 0; k < BK; ++k) cs1 += col[k * SA];
        }
      }
      using FA = typename std::conditional<A_KMAJ, FragD<WTM>, FragG<false, WTM, SA>>::type;
      using FB = typename std::conditional<B_KMAJ, FragD<WTN>, FragG<false, WTN, SB>>::type;
      uint32_t abt[4], bbt[4];
      const uint32_t a0 = A_KMAJ ? lds_u32(As + cur * TA + (wm * (32 * WTM) + li) * 32) : lds_u32(As + cur * TA + (4 * lh) * SA + wm * (32 * WTM) + li);
      const uint32_t b0 = B_KMAJ ? lds_u32(Bs + cur * TB + (wn * (32 * WTN) + li) * 32) : lds_u32(Bs + cur * TB + (4 * lh) * SB + wn * (32 * WTN) + li);
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) {
        const uint32_t ch = (uint32_t)(((2 * t4 + lh) ^ swz) * 16);
        abt[t4] = a0 + ch;
        bbt[t4] = b0 + ch;
      }
      auto rd_a = [&](FA& f, int t4) { if constexpr (A_KMAJ) f.read(abt, t4); else f.read(a0, t4); };
      auto rd_b = [&](FB& f, int t4) { if constexpr (B_KMAJ) f.read(bbt, t4); else f.read(b0, t4); };
      if constexpr (SPLIT != 0) {
        // one k16 slab = fragment groups 2g and 2g+1: a lane's 8 values (same k set for A and B) -> hi/lo bf16x8.
        // Slab 1 is requested from LDS before the conversion + MFMAs of slab 0.
        FA a[4];
        FB b[4];
        rd_a(a[0], 0); rd_b(b[0], 0); rd_a(a[1], 1); rd_b(b[1], 1);
        rd_a(a[2], 2); rd_b(b[2], 2); rd_a(a[3], 3); rd_b(b[3], 3);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          constexpr int PEND = 2 * (FA::NREAD + FB::NREAD);  // reads of slab 1 may still be in flight (counter is 4 bits)
          if (g == 0) wait_lgkm<(PEND > 15 ? 15 : PEND)>(); else wait_lgkm<0>();
          a[2 * g].tie(); a[2 * g + 1].tie(); b[2 * g].tie(); b[2 * g + 1].tie();
          float xa[WTM][8], xb[WTN][8];
#pragma unroll
          for (int i = 0; i < WTM; ++i)
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) { xa[i][s4] = a[2 * g].get(i, s4); xa[i][4 + s4] = a[2 * g + 1].get(i, s4); }
#pragma unroll
          for (int j = 0; j < WTN; ++j)
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) { xb[j][s4] = b[2 * g].get(j, s4); xb[j][4 + s4] = b[2 * g + 1].get(j, s4); }
          if constexpr (SPLIT == 6) {
            bf16x8 ah[WTM], am[WTM], al[WTM], bh[WTN], bm[WTN], bl[WTN];
#pragma unroll
            for (int i = 0; i < WTM; ++i) split_bf16x8_3(xa[i], ah[i], am[i], al[i]);
#pragma unroll
            for (int j = 0; j < WTN; ++j) split_bf16x8_3(xb[j], bh[j], bm[j], bl[j]);
#pragma unroll
            for (int i = 0; i < WTM; ++i)
#pragma unroll
              for (int j = 0; j < WTN; ++j) {  // smallest terms first
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bm[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bh[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bm[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
              }
          } else {
            bf16x8 ah[WTM], al[WTM], bh[WTN], bl[WTN];
#pragma unroll
            for (int i = 0; i < WTM; ++i) split_bf16x8(xa[i], ah[i], al[i]);
#pragma unroll
            for (int j = 0; j < WTN; ++j) split_bf16x8(xb[j], bh[j], bl[j]);
#pragma unroll
            for (int i = 0; i < WTM; ++i)
#pragma unroll
              for (int j = 0; j < WTN; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
              }
          }
        }
        return;
      }
      FA a[2];
      FB b[2];
      rd_a(a[0], 0);
      rd_b(b[0], 0);
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) {
        if (t4 + 1 < 4) {
          rd_a(a[(t4 + 1) & 1], t4 + 1);
          rd_b(b[(t4 + 1) & 1], t4 + 1);
          wait_lgkm<FA::NREAD + FB::NREAD>();
        } else {
          wait_lgkm<0>();
        }
        a[t4 & 1].tie();
        b[t4 & 1].tie();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
          for (int i = 0; i < WTM; ++i)
#pragma unroll
            for (int j = 0; j < WTN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t4 & 1].get(i, s4), b[t4 & 1].get(j, s4), acc[i][j], 0, 0, 0);
      }
    }
  };
  // wave w moves chunks 4w..4w+3 (1 KB each) of the A and of the B tile: 8 rows x 128 B of a k-contiguous
  // operand (chunk index swizzled), 2 k rows x 512 B of an m/n-contiguous one
  // NQA / NQB chunks of 1 KB per wave and tile: dim / 32 (a tile of `dim` rows x 32 k or 32 k x `dim` columns is dim/8 KB)
  constexpr int NQA = BM / (16 * WGN), NQB = BN / (16 * WGN);  // (dim / 8 chunks) / (2 * WGN waves)
  static_assert(NQA >= 1 && NQB >= 1, "every wave moves at least one chunk of each operand");
  uint32_t dsa[NQA], dsb[NQB];  // per-lane byte offsets from the (K-advanced) scalar base; the host takes this path below 4 GB
  if constexpr (DMA) {
    constexpr int LPA = BM / 4, LPB = BN / 4;  // lanes per k row of an m/n-contiguous tile (rows per instruction: 64 / LP)
#pragma unroll
    for (int q = 0; q < NQA; ++q) {
      const int c = NQA * wave + q;
      if constexpr (A_KMAJ) {
        const int row = 8 * c + (lane >> 3);
        dsa[q] = (uint32_t)(((long)min(m0 + row, p.M - 1) * p.lda + 4 * ((lane & 7) ^ ((row >> 1) & 7))) * 4);
      } else {
        dsa[q] = (uint32_t)(((long)((64 / LPA) * c + lane / LPA) * p.lda + min(m0 + 4 * (lane % LPA), p.M - 4)) * 4);
      }
    }
#pragma unroll
    for (int q = 0; q < NQB; ++q) {
      const int c = NQB * wave + q;
      if constexpr (B_KMAJ) {
        const int row = 8 * c + (lane >> 3);
        dsb[q] = (uint32_t)(((long)min(n0 + row, p.N - 1) * p.ldb + 4 * ((lane & 7) ^ ((row >> 1) & 7))) * 4);
      } else {
        dsb[q] = (uint32_t)(((long)((64 / LPB) * c + lane / LPB) * p.ldb + min(n0 + 4 * (lane % LPB), p.N - 4)) * 4);
      }
    }
  }
  auto dma_issue = [&](int buf, int ktile) {
    if constexpr (DMA) {
      const uint32_t la = __builtin_amdgcn_readfirstlane(lds_u32(As + buf * TA) + (uint32_t)(NQA * wave) * 1024u);
      const uint32_t lb = __builtin_amdgcn_readfirstlane(lds_u32(Bs + buf * TB) + (uint32_t)(NQB * wave) * 1024u);
      // wave uniform by construction; hipcc does not always keep the 64-bit row-stride product in scalar registers once
      // the loop is unrolled (an "s" asm operand must BE one), so the halves go through readfirstlane
      auto uni = [](const float* q) {
        const uint64_t v = reinterpret_cast<uint64_t>(q);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return reinterpret_cast<const float*>(((uint64_t)hi << 32) | lo);
      };
      const float* sa = uni(p.A + (A_KMAJ ? (long)ktile * BK : (long)ktile * BK * p.lda));
      const float* sb = uni(p.B + (B_KMAJ ? (long)ktile * BK : (long)ktile * BK * p.ldb));
#pragma unroll
      for (int q = 0; q < NQA; ++q) glds16s(sa, dsa[q], la + 1024u * q);
#pragma unroll
      for (int q = 0; q < NQB; ++q) glds16s(sb, dsb[q], lb + 1024u * q);
    }
  };

  // K tiles [t0, t1) of this block (split-K: a slice of K); tiles below K/BK are full
  const int kbeg = ks * p.kper, kend = whole ? p.K : min(p.K, kbeg + p.kper);
  const int t0 = kbeg / BK, t1 = (kend + BK - 1) / BK;
  const int tfull = FAST ? min(t1, p.K / BK) : t0;  // tiles [t0, tfull) go through the fast loaders
  if constexpr (DMA) {
    // the first tile takes the DMA path too: no per-thread operand pointers, no register staging in these kernels at all
    // (a 16-tile reduction spends a sixth of its life around the K loop)
#ifdef BLM_GEMM_FIRST_TILE_REG  // traffic A/B only (tools/traffic_probe.sh): the first tile through registers, as before 6cbf842
    if (t0 < tfull) fetch_fast(t0); else fetch_slow(t0 * BK);
    stash(0, t0 * BK, t0 < tfull);
#else
    if (t0 < tfull) {
      dma_issue(0, t0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      fetch_slow(t0 * BK);
      stash(0, t0 * BK, false);
    }
#endif
  } else {
    if (t0 < tfull) fetch_fast(t0); else fetch_slow(t0 * BK);
    stash(0, t0 * BK, t0 < tfull);
  }
  __syncthreads();
  int kt = t0;
#ifdef BLM_GEMM_PROF
  unsigned long long pf_compute = 0, pf_stash = 0, pf_barrier = 0, pf_n = 0;
  unsigned long long ta = 0, tb = 0, tc = 0, td = 0;
#define BLM_PF_NOW(x) { __builtin_amdgcn_sched_barrier(0); x = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
#define BLM_PF_ADD() { pf_compute += tb - ta; pf_stash += tc - tb; pf_barrier += td - tc; pf_n += 1; }
#else
#define BLM_PF_NOW(x)
#define BLM_PF_ADD()
#endif
  if constexpr (DMA) {
    // tile kt+1 travels global -> LDS (other buffer) while tile kt is multiplied; the counted wait
    // retires this wave's DMA, the barrier everybody's, and also fences the reads of the buffer that the
    // next iteration overwrites
    // Two tiles per trip: tile kt sits in buffer (kt - t0) & 1, so inside the trip the buffer index is a compile-time
    // constant and every LDS fragment address is base + immediate -- no vector adds in the loop (on gfx950 each vector
    // instruction is paid in matrix time, tools/mfma_valu_overlap.hip).
    // (Measured and not kept: three LDS stages for the 24 KB tiles, tile kt+2 in flight while kt is multiplied -- the
    // roofline shape does not move, +0.3 %, and the launches that ran three workgroups per CU lose one: -3 ... -8 %.
    // The wait in front of the barrier is not what the loop loses.)
    auto dma_step = [&](int cur, int next_tile) {
      BLM_PF_NOW(ta)
      dma_issue(cur ^ 1, next_tile);
      compute_dma(cur);
      BLM_PF_NOW(tb)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      BLM_PF_NOW(tc)
      __syncthreads();
      BLM_PF_NOW(td)
      BLM_PF_ADD()
    };
#ifdef BLM_GEMM_ONE_TILE_TRIP  // traffic A/B only: the rolled loop for every kernel
    constexpr bool kRolled = true;
#else
    constexpr bool kRolled = SPLIT != 0;
#endif
    if constexpr (kRolled) {  // opt-in split-bf16 kernels: the two-tile body demotes their fragment arrays to scratch
      for (; kt + 1 < tfull; ++kt) dma_step((kt - t0) & 1, kt + 1);
    } else {
      for (; kt + 2 < tfull; kt += 2) {
        dma_step(0, kt + 1);
        dma_step(1, kt + 2);
      }
      if (kt + 1 < tfull) {
        dma_step(0, kt + 1);
        ++kt;
      }
    }
  } else if constexpr (FAST && !SAMP && WTM == 2 && WTN == 2) {  // smaller tiles: the second register set would cost them a workgroup per CU
    // Steady state with the global loads TWO K tiles ahead (two register sets, LDS still double
    // buffered): tile kt+2 is requested before the MFMAs of tile kt, tile kt+1 -- requested one
    // iteration earlier -- is written to LDS after them.  In-kernel stamps on the one-tile-ahead loop
    // showed 0.5-1.3k cycles per K tile parked on the vmcnt in front of the LDS write (short-K
    // launches stream fresh rows all the time), against 8.2k cycles of MFMA issue per tile pair.
    // The refill past the last tile re-reads the last one (clamped index): branch-free body, so
    // every s_waitcnt counts exactly its own set.
    float4 ra1[NA], rb1[NB];
#pragma unroll
    for (int j = 0; j < NA; ++j) ra1[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < NB; ++j) rb1[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fetch2 = [&](float4 (&xa)[NA], float4 (&xb)[NB], int k) {
#pragma unroll
      for (int j = 0; j < NA; ++j) xa[j] = *reinterpret_cast<const float4*>(pa[j] + k * stepA);
#pragma unroll
      for (int j = 0; j < NB; ++j) xb[j] = *reinterpret_cast<const float4*>(pb[j] + k * stepB);
    };
    auto stash2 = [&](float4 (&xa)[NA], float4 (&xb)[NB], int buf) {
      if constexpr (!A_KMAJ) {
        if (do_cs) {
#pragma unroll
          for (int j = 0; j < NA; ++j) { cs.x += xa[j].x; cs.y += xa[j].y; cs.z += xa[j].z; cs.w += xa[j].w; }
        }
      }
      if constexpr (A_KMAJ) r2s_kmaj<BM>(As + buf * TA, xa); else r2s_nmaj<BM, SA>(As + buf * TA, xa);
      if constexpr (B_KMAJ) r2s_kmaj<BN>(Bs + buf * TB, xb); else r2s_nmaj<BN, SB>(Bs + buf * TB, xb);
    };
    if (t0 + 1 < tfull) fetch2(ra1, rb1, t0 + 1);
    for (; kt + 2 < tfull; kt += 2) {  // tile kt is in LDS buffer 0, tile kt+1 in the second register set
      BLM_PF_NOW(ta)
      fetch2(ra, rb, kt + 2);
      asm volatile("" ::: "memory");  // keep the global loads in front of the MFMA phase (hipcc sinks them otherwise)
      compute(0, [&] { stash2(ra1, rb1, 1); });
      BLM_PF_NOW(tb)
      BLM_PF_NOW(tc)
      __syncthreads();
      BLM_PF_NOW(td)
      BLM_PF_ADD()
      BLM_PF_NOW(ta)
      fetch2(ra1, rb1, min(kt + 3, tfull - 1));
      asm volatile("" ::: "memory");
      compute(1, [&] { stash2(ra, rb, 0); });
      BLM_PF_NOW(tb)
      BLM_PF_NOW(tc)
      __syncthreads();
      BLM_PF_NOW(td)
      BLM_PF_ADD()
    }
    if (kt + 1 < tfull) {
      compute(0, [&] { stash2(ra1, rb1, 1); });
      __syncthreads();
      ++kt;
    }
  } else {
    for (; kt + 1 < tfull; ++kt) {  // steady state: next tile is a full one
      fetch_fast(kt + 1);
      asm volatile("" ::: "memory");  // keep the global loads in front of the MFMA phase (hipcc sinks them otherwise)
      compute((kt - t0) & 1, [] {});
      stash((kt + 1 - t0) & 1, (kt + 1) * BK, true);
      __syncthreads();
    }
  }
#ifdef BLM_GEMM_LIFE
  if (t == 0 && blockIdx.x < 8192) blm_wg_life[4 * blockIdx.x + 1] = wall_clock64();
#endif
#ifdef BLM_GEMM_PROF
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&blm_prof[0], pf_compute); atomicAdd(&blm_prof[1], pf_stash); atomicAdd(&blm_prof[2], pf_barrier); atomicAdd(&blm_prof[3], pf_n);
  }
#endif
  for (; kt < t1; ++kt) {  // last full tile and/or the K tail
    const bool more = kt + 1 < t1;
    if (more) fetch_slow((kt + 1) * BK);
    if constexpr (DMA) compute_dma((kt - t0) & 1); else compute((kt - t0) & 1, [] {});
    if (more) stash((kt + 1 - t0) & 1, (kt + 1) * BK, false);
    __syncthreads();
  }

  if constexpr (!A_KMAJ && DMA) {
    if (do_cs && t < BM && m0 + t < p.M) atomicAdd(p.colsum_a + m0 + t, cs1 * p.alpha);
  }
  if constexpr (!A_KMAJ && !DMA) {
    if (do_cs) {  // reduce the per-thread partial column sums over the k-row groups through LDS
      constexpr int TPR = BM / 4, RPP = 256 / TPR;
      float* red = smem;  // all tiles consumed: the staging buffers are free
      *reinterpret_cast<float4*>(red + (t / TPR) * BM + 4 * (t % TPR)) = cs;
      __syncthreads();
      if (t < BM && m0 + t < p.M) {
        float sum = 0.f;
#pragma unroll
        for (int g = 0; g < RPP; ++g) sum += red[g * BM + t];
        atomicAdd(p.colsum_a + m0 + t, sum * p.alpha);
      }
    }
  }

  // ---- epilogue (one straight-line, fully unrolled body per epilogue kind: the accumulator must
  // only ever be indexed by compile-time constants or it is demoted to scratch)
  if (p.vec_epi && !atomic) {  // aligned C/aux, N % 4 == 0, no atomics: 16-byte row-wise epilogue through LDS
    switch (p.epi) {
      case BLM_EPI_NONE: epilogue_rows<BLM_EPI_NONE, WTM, WTN, WGN>(p, acc, smem, m0, n0, wm, wn, li, lh); return;
      case BLM_EPI_BIAS: epilogue_rows<BLM_EPI_BIAS, WTM, WTN, WGN>(p, acc, smem, m0, n0, wm, wn, li, lh); return;
      case BLM_EPI_BIAS_GELU: epilogue_rows<BLM_EPI_BIAS_GELU, WTM, WTN, WGN>(p, acc, smem, m0, n0, wm, wn, li, lh); return;
      case BLM_EPI_MUL_DGELU: epilogue_rows<BLM_EPI_MUL_DGELU, WTM, WTN, WGN>(p, acc, smem, m0, n0, wm, wn, li, lh); return;
      case BLM_EPI_GP_MIX: epilogue_rows<BLM_EPI_GP_MIX, WTM, WTN, WGN>(p, acc, smem, m0, n0, wm, wn, li, lh); return;
      case BLM_EPI_MUL_DGP_MIX: epilogue_rows<BLM_EPI_MUL_DGP_MIX, WTM, WTN, WGN>(p, acc, smem, m0, n0, wm, wn, li, lh); return;
      case BLM_EPI_CE_PART: epilogue_rows<BLM_EPI_CE_PART, WTM, WTN, WGN>(p, acc, smem, m0, n0, wm, wn, li, lh); return;
      default: break;
    }
  }
  switch (p.epi) {
    case BLM_EPI_BIAS: epilogue<BLM_EPI_BIAS, WTM, WTN>(p, acc, m0, n0, wm, wn, li, lh, ks == 0, atomic); break;
    case BLM_EPI_BIAS_GELU: epilogue<BLM_EPI_BIAS_GELU, WTM, WTN>(p, acc, m0, n0, wm, wn, li, lh); break;
    case BLM_EPI_MUL_DGELU: epilogue<BLM_EPI_MUL_DGELU, WTM, WTN>(p, acc, m0, n0, wm, wn, li, lh); break;
    case BLM_EPI_GP_MIX: epilogue<BLM_EPI_GP_MIX, WTM, WTN>(p, acc, m0, n0, wm, wn, li, lh); break;
    case BLM_EPI_MUL_DGP_MIX: epilogue<BLM_EPI_MUL_DGP_MIX, WTM, WTN>(p, acc, m0, n0, wm, wn, li, lh); break;
    case BLM_EPI_BAYES_WGRAD:
      if constexpr (OP == BLM_GEMM_TN) epilogue<BLM_EPI_BAYES_WGRAD, WTM, WTN>(p, acc, m0, n0, wm, wn, li, lh, ks == 0, atomic);
      break;
    default: epilogue<BLM_EPI_NONE, WTM, WTN>(p, acc, m0, n0, wm, wn, li, lh, true, atomic); break;
  }
}

inline int gemm_cu_count() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    n = v;
  }
  return n;
}

template <int OP, int WTM, int WTN, bool SAMP, bool FAST, int WGN = 2>
static int launch_cfg(const GemmP& p, hipStream_t st) {
  constexpr int BM = 64 * WTM, BN = 32 * WTN * WGN;
  constexpr bool A_KMAJ = (OP != BLM_GEMM_TN), B_KMAJ = (OP == BLM_GEMM_NT);
  constexpr bool DMAL = use_dma<OP, WTM, WTN, SAMP, FAST>();
  constexpr int KSX = DMAL ? 32 : KS, PADL = DMAL ? 0 : 4;
  constexpr int TA = A_KMAJ ? BM * KSX : BK * (BM + PADL), TB = B_KMAJ ? BN * KSX : BK * (BN + PADL);
  constexpr size_t lds = (size_t)2 * (TA + TB) * sizeof(float);
  GemmP q = p;
  q.gm = (p.M + BM - 1) / BM;
  q.gn = (p.N + BN - 1) / BN;
  const long nb = (long)q.gm * q.gn;
  // K slices as planned (gemm_plan.hip: legality -- plain, bias or Bayesian-wgrad epilogue, dense or accumulated C -- is the
  // planner's); partial sums meet in C through float atomics, C is zeroed first unless accumulating
  // plan_splits < -1: only the tiles beyond the last full round of workgroup slots (CUs x co-resident workgroups of this
  // tile) are sliced, |plan_splits| ways -- a grid of 1032 tiles on 512 slots runs 1024 tiles whole and 8 x 32 slices
  // instead of 6 x 1032 slices that all pass through atomics.  No full round: every tile is sliced (the uniform form).
  int splits = p.plan_splits > 1 ? p.plan_splits : (p.plan_splits < -1 ? -p.plan_splits : 1);
  q.tail_from = 0;
  if (p.plan_splits < -1) {
    constexpr int OCC = (WTM * WTN == 1) ? 5 : (WTM * WTN * WGN == 4 ? 3 : 2);  // LDS: 32 / 48 / 64 KB of 160 (gemm_plan.hip kTiles)
    const long slots = (long)(p.plan_cus > 0 ? p.plan_cus : gemm_cu_count()) * OCC;
    q.tail_from = (int)(nb / slots * slots);
    if (q.tail_from == nb) { q.tail_from = 0; splits = 1; }  // whole rounds only: nothing to slice
  }
  q.splits = splits;
  q.kper = splits > 1 ? ((p.K + splits - 1) / splits + BK - 1) / BK * BK : (p.K > 0 ? p.K : 1);
  q.atomic = splits > 1;
  const long nblocks = q.tail_from + (nb - q.tail_from) * (long)splits;
  {
    // row-wise epilogue through LDS for aligned, unsliced launches; the register-layout epilogue for atomics and odd alignments
    const bool al = ((reinterpret_cast<uintptr_t>(p.C) | reinterpret_cast<uintptr_t>(p.aux) | reinterpret_cast<uintptr_t>(p.bias) | reinterpret_cast<uintptr_t>(p.C2)) & 15) == 0;
    q.vec_epi = (!q.atomic || q.tail_from > 0) && al && p.N % 4 == 0 && p.ldc % 4 == 0 &&
                (p.epi == BLM_EPI_NONE || p.epi == BLM_EPI_BIAS || p.epi == BLM_EPI_BIAS_GELU || p.epi == BLM_EPI_MUL_DGELU ||
                 p.epi == BLM_EPI_GP_MIX || p.epi == BLM_EPI_MUL_DGP_MIX || p.epi == BLM_EPI_CE_PART);
    if (p.epi == BLM_EPI_CE_PART && !q.vec_epi) return blm_fail(BLM_ERR_UNSUPPORTED, "blm_linear_nll: needs N % 4 == 0 and 16-byte aligned bias / workspace");
  }
  if (q.atomic && !(p.flags & BLM_GEMM_ACCUMULATE))
    BLM_HIP(hipMemsetAsync(p.C, 0, (size_t)p.M * p.N * sizeof(float), st));
  if constexpr (DMAL) {
    if (p.split) {  // opt-in split-bf16 arithmetic: same loaders, tiles and epilogues, different matrix instruction
      auto kern = p.split == 6 ? gemm_f32_kernel<OP, WTM, WTN, SAMP, FAST, 6, WGN> : gemm_f32_kernel<OP, WTM, WTN, SAMP, FAST, 3, WGN>;
      static bool attr_done_s[2] = {false, false};
      if (!attr_done_s[p.split == 6]) {
        BLM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done_s[p.split == 6] = true;
      }
      hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(128 * WGN), lds, st, q);
      BLM_HIP(hipGetLastError());
      return BLM_OK;
    }
  }
  auto kern = gemm_f32_kernel<OP, WTM, WTN, SAMP, FAST, 0, WGN>;
  static bool attr_done = false;  // per instantiation; benign race (idempotent)
  if (!attr_done) {
    BLM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds));
    attr_done = true;
  }
  // HBM-side traffic of a launch with two or more workgroups per CU (tools/traffic_probe.sh, request-size PMC counters,
  // profiles/r03_gemm_traffic_*.txt): the B panels are fetched TWICE per XCD once K exceeds ~1400 -- the arbiter serves the
  // oldest ready wave first, so the workgroup that reached a CU first runs well ahead of its co-resident neighbour, and the
  // panel lines the late half needs have left the XCD's 4 MB L2 (window = 4 MB / 6 KB of panel data per k = ~700 k) by
  // the time it asks.  128x128 tiles (one workgroup per CU, lock step) read exactly the operands.  Measured and not kept:
  // padding the LDS request so that a one-round grid spreads evenly (no change), wave-priority turns between the
  // co-resident workgroups (-25 % of the surplus at K = 4096, more at 8192, step +0.3 %).  The bytes cost no time (1.5 TB/s).
  hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(128 * WGN), lds, st, q);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

template <int OP, bool SAMP>
int launch_op(const GemmP& p, hipStream_t st) {
  // tile and K slices: gemm_plan.hip (override > measured plan table > cost model), already legal for this call
  if (!p.fast) return launch_cfg<OP, 1, 1, SAMP, false>(p, st);  // odd shapes/alignments: guarded loaders only
  if constexpr (!SAMP) {  // tile 28: 128 x 128 on eight waves (LDS-DMA loaders, whole K tiles: the planner offers it only then)
    if (p.plan_tile == 28 && p.K % BK == 0 && !p.split) return launch_cfg<OP, 2, 1, SAMP, true, 4>(p, st);
  }
  switch (p.plan_tile) {
    case 11: return launch_cfg<OP, 1, 1, SAMP, true>(p, st);
    case 12: return launch_cfg<OP, 1, 2, SAMP, true>(p, st);
    case 21: return launch_cfg<OP, 2, 1, SAMP, true>(p, st);
    default: return launch_cfg<OP, 2, 2, SAMP, true>(p, st);
  }
}

}  // namespace blm
