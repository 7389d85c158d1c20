// One LSTM time step in ONE launch: recurrent product h_{t-1} W_hh^T on the f32 matrix cores with the
// cell non-linearity fused behind it (replaces: split-K memset + skinny GEMM M=B + cell kernel; the
// reference gets the same fusion from _VF.lstm, model.py:812).
//
// The recurrent GEMM is skinny (M = B = 64, N = 4H, K = H): the generic 128x128 tiles cannot fill the
// chip without split-K atomics.  Here a workgroup OWNS 8 hidden units = their 32 gate rows (i,f,g,o
// of each unit) for 32 batch rows, so that after the product the four gates of a unit sit in the same
// workgroup and the cell update needs no second pass over HBM:
//   * grid (H/8, ceil(B/32)); 4 waves split K = H in four contiguous quarters, one 32x32 MFMA
//     accumulator tile per wave, reduced across the waves through LDS at the end (no atomics, no
//     memset: the summation order is fixed, results are run-to-run identical).
//   * v_mfma_f32_32x32x2_f32 only needs A and B to agree on WHICH k a (step, lane half) pair means, so
//     lane half 0 walks the first half of the wave's K quarter and half 1 the second half: every lane
//     then consumes a contiguous run of its row, fetched as float4 and read back as ds_read_b128.
//   * operands are staged through wave-private LDS tiles [32 rows][64 k + 4 pad] (coalesced 128-B
//     global segments in, conflict-free b128 reads out) from a 4-deep register ring of global loads
//     (64 KB per wave in flight under the MFMAs); the waves never meet at a barrier inside the K loop.
// Per step and workgroup: 128 KB of h and 128 KB of W_hh (L2/MALL resident across steps), 128 MFMA
// per wave.  H % 32 == 0 and 16-byte aligned operands are required (blm_lstm_step_fwd returns
// BLM_ERR_UNSUPPORTED otherwise and the host uses blm_gemm + blm_lstm_cell_fwd).
#include <cmath>
#include <cstdlib>

#include "blm_device.h"
#include "blm_host.h"

namespace blm {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bu32x4 = __attribute__((ext_vector_type(4))) unsigned int;  // raw_buffer_load_b128 result

constexpr int LSTR = 68;                  // staged tile row stride (floats): 64 k + 4 pad
constexpr int TILE = 32 * LSTR;           // one 32-row operand tile
constexpr int WAVE_LDS = 2 * TILE;        // h tile + W tile
constexpr int RSTR = 40;                  // reduction row stride

struct LstmStepP {
  const float *xw, *whh, *hprev, *cprev;
  float *h, *c, *ga;
  const float* hnoise;  // optional (H): added to h after the cell (VLSTMCell, model.py:2523-2527)
  const float* coef;    // GPLSTMCell gate types 1-4 (model.py:1754-1771): gate `ovr` is the GPNN mixture of its
  float* zsave;         //   pre-activation (coef (4,H), slot order of gp_mix), z kept for the backward pass
  int ovr;              //   -1: plain LSTM; 4: gate type 6 -- the whole hidden projection h W^T + rbias passes through
  const float* rbias;   //   the mixture (coef (4,4H), z (B,4H)) before it is added to xw (model.py:1744-1752)
  int B, H;
  const float* probs;   // NS = 8 (architecture-search cell, model_search_bayes.py:686-710): (4,2) mixing weights on the device
#ifdef BLM_LSTM_PROF
  long long* prof;      // tools/lstm_step_prof.hip only: 8 wall-clock stamps (10 ns units) per wave
  int alias;            // timing-only: 1 = every workgroup reads the W rows of unit block 0, 2 = also the h rows of batch row 0
#endif
};

#ifdef BLM_LSTM_PROF
#define LSTM_STAMP(i) do { if (lane == 0) stamps[i] = wall_clock64(); } while (0)
#else
#define LSTM_STAMP(i) do { } while (0)
#endif

// NS = number of gate streams per hidden unit: 4 (LSTM: i f g o) or 8 (search cell: i f g o | i' f' g' o', every
// gate the probs-weighted mix of the two activations).  A workgroup owns 32 / NS units = 32 rows of the weight.
// RING = chunks of global loads in flight per wave (1, 2 or 4 register sets of 16 KB); REFILL = false when the whole
// K run of a lane fits the ring (nchunk == RING: every load of the step is issued before the first wait, the launch ->
// first-byte latency is paid once and the rest streams in under the MFMAs).
// NW = waves per workgroup = K slices: 4 (one wave per SIMD, two staging buffers each) or 8 (two waves per SIMD with one
// staging buffer each: while one wave of a SIMD stages its next chunk through LDS -- 16 ds_write_b128 + 16 ds_read_b128
// that a single in-order wave cannot overlap with its own dependent MFMA chain -- the other one keeps the matrix pipe busy).
// PIPE (RING = 2, even nchunk): the K loop is software pipelined inside the wave -- while the 32 dependent MFMAs of
// chunk c execute, the wave issues chunk c + 1's LDS writes, the ring refill and chunk c + 1's fragment reads BETWEEN
// them (a dependent MFMA stalls the in-order wave for ~60 cycles at issue: room for one LDS instruction per MFMA), into a
// second fragment register set.  ~310 VGPRs: one workgroup per CU.
// TAIL = false (the lane's K run is a whole number of 32-float chunks, e.g. H % 256 == 0 with 4 waves): no zero-fill
// selects in front of the LDS writes and wave-uniform chunk offsets.  On gfx950 every vector instruction is paid in
// matrix time (tools/mfma_valu_overlap.hip): the 64 v_cndmask + 16 64-bit address adds per chunk of the general form
// were ~ 18 % on top of its 32 MFMAs.
// MB = batch tiles of 32 rows per workgroup (1 or 2).  MB = 2 (the search cell at B > 32, round 5): a workgroup multiplies its 32
// weight rows against 64 batch rows -- two accumulator tiles per wave fed from ONE staged W tile -- so the 8H x H weight is
// streamed once per step instead of once per batch tile, and the grid is H / 4 workgroups = one round of the chip where the
// MB = 1 form needed two (the step is latency-bound: two rounds cost twice one round, tools/search_step_bench.py).
template <int RING, int NS, bool REFILL, int NW, bool PIPE, bool TAIL, int MB = 1>
__device__ __forceinline__ void lstm_step_fwd_body(const LstmStepP& p) {
  constexpr int U = 32 / NS;  // hidden units per workgroup
  constexpr int NT = 1 + MB;  // staged tiles per wave and chunk: MB tiles of h, one of W
  static_assert(MB == 1 || (MB == 2 && NS == 8), "two batch tiles per workgroup: the search cell only (its epilogue has a thread per (batch row, unit) of 64 rows)");
  // ONE staging buffer per wave (LDS runs a wave's instructions in order and the fragments are in registers before the
  // MFMAs start, so the next chunk may overwrite the tile): 70 KB per 4-wave workgroup and <= 256 VGPRs per wave, i.e. TWO
  // workgroups fit a CU -- the step kernels of two independent recurrences (the layers of a stack, run as a wavefront on
  // two streams) overlap each other's launch gaps and load latencies.
  constexpr int NBUF = 1;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const int j0 = blockIdx.x * U, b0 = blockIdx.y * 32 * MB;
  const int H = p.H, B = p.B;
  const int Kw = H / NW, Kh = Kw >> 1, kbase = wave * Kw;
  const int nchunk = (Kh + 31) >> 5;
  float* base = sm + wave * (NBUF * NT * TILE);
#ifdef BLM_LSTM_PROF
  long long stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  LSTM_STAMP(0);

  const int brow = threadIdx.x / U, eu = threadIdx.x % U;
  const int eb = b0 + brow, ej = j0 + eu;
  const bool eok = eb < B && brow < 32 * MB;
  float xg[NS], cprev = 0.f;

  // staging roles: one instruction moves 4 rows x 2 halves x 128 B
  const int srow = lane >> 4, shalf = (lane >> 3) & 1, spart = lane & 7;
  // buffer loads: descriptor (scalar registers, built from kernel arguments) + one 32-bit byte offset per staged row
  // + a scalar offset for the wave's K slice and the chunk: a chunk step costs no vector instruction (hipcc turns
  // base + zext(offset) of a plain pointer into 64-bit vector adds once the offsets are hoisted out of the loop)
  const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.hprev), 0, (int)(4u * (uint32_t)B * (uint32_t)H), 0x00020000);
#ifdef BLM_LSTM_PROF
  // timing-only alias 3: a zero-length descriptor -- every W_hh load returns 0 without touching memory, i.e. the upper bound of
  // what taking the W fetch off the h_t -> h_{t+1} chain (prefetch before h arrives) could buy
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.whh), 0, p.alias == 3 ? 0 : (int)(4u * NS * (uint32_t)H * (uint32_t)H), 0x00020000);
#else
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.whh), 0, (int)(4u * NS * (uint32_t)H * (uint32_t)H), 0x00020000);  // < 4 GB (host check)
#endif
  const int kb4 = __builtin_amdgcn_readfirstlane(kbase * 4);
  uint32_t aoff[MB][8], woff[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int row = 4 * q + srow;
#ifdef BLM_LSTM_PROF
    aoff[0][q] = (uint32_t)(((long)(p.alias >= 2 ? 0 : min(b0 + row, B - 1)) * H + shalf * Kh + 4 * spart) * 4);
    woff[q] = (uint32_t)((((long)(row / U) * H + (p.alias >= 1 ? 0 : j0) + (row % U)) * H + shalf * Kh + 4 * spart) * 4);
#else
#pragma unroll
    for (int m = 0; m < MB; ++m) aoff[m][q] = (uint32_t)(((long)min(b0 + 32 * m + row, B - 1) * H + shalf * Kh + 4 * spart) * 4);
    woff[q] = (uint32_t)((((long)(row / U) * H + j0 + (row % U)) * H + shalf * Kh + 4 * spart) * 4);
#endif
  }
  auto ldg = [](__amdgpu_buffer_rsrc_t r, uint32_t voff, int soff) {
    const bu32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
  };
  // zero-fill of the K tail (TAIL only): component-wise selects, a select of whole float4 lvalues would demote the ring to scratch
  auto fill = [](const float4& x, bool in) {
    if constexpr (TAIL) return make_float4(in ? x.x : 0.f, in ? x.y : 0.f, in ? x.z : 0.f, in ? x.w : 0.f);
    else return x;
  };
  const int soff = srow * LSTR + shalf * 32 + 4 * spart;
  // register ring of RING chunks: with one wave per SIMD the only way to cover the L2/MALL latency is
  // to keep RING x 16 KB per wave of loads in flight while the matrix core works on a chunk.  The loop
  // body is branch free (tail lanes and the refill past the last chunk read a clamped, valid address)
  // so that the s_waitcnt in front of each put() only waits for ITS chunk.
  float4 ra[RING][MB][8], rw[RING][8];
  auto fetch = [&](float4 (&a)[MB][8], float4 (&w)[8], int c) {
    if constexpr (TAIL) {
      const int off = min(32 * min(c, nchunk - 1) + 4 * spart, Kh - 4) - 4 * spart;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
#pragma unroll
        for (int m = 0; m < MB; ++m) a[m][q] = ldg(arsrc, aoff[m][q] + 4u * off, kb4);
        w[q] = ldg(wrsrc, woff[q] + 4u * off, kb4);
      }
    } else {
      const int so = kb4 + 128 * min(c, nchunk - 1);  // wave-uniform
#pragma unroll
      for (int q = 0; q < 8; ++q) {
#pragma unroll
        for (int m = 0; m < MB; ++m) a[m][q] = ldg(arsrc, aoff[m][q], so);
        w[q] = ldg(wrsrc, woff[q], so);
      }
    }
  };
  f32x16 acc[MB];
#pragma unroll
  for (int m = 0; m < MB; ++m) acc[m] = (f32x16)(0.f);
  // one chunk: registers -> LDS tiles `buf` (h tile(s), then the W tile), refill the registers with chunk c + RING, 32 MFMA steps per h tile
  auto chunk = [&](float4 (&a)[MB][8], float4 (&w)[8], int buf, int c) {
    const bool in = 32 * c + 4 * spart < Kh;  // K tail of the last chunk -> zeros
    float* d = base + buf * NT * TILE + soff;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
#pragma unroll
      for (int m = 0; m < MB; ++m) *reinterpret_cast<float4*>(d + m * TILE + 4 * q * LSTR) = fill(a[m][q], in);
      *reinterpret_cast<float4*>(d + MB * TILE + 4 * q * LSTR) = fill(w[q], in);
    }
    if (c == 0) {  // the first chunk's bytes have landed (its LDS writes are issued)
      LSTM_STAMP(1);
#ifdef BLM_LSTM_PROF
      if (lane == 0) stamps[6] = clock64();
#endif
    }
    if constexpr (REFILL) fetch(a, w, c + RING);
    // the tiles are wave private: LDS executes a wave's instructions in order, so the reads below see
    // the writes above without a workgroup barrier; the fence only pins the compiler's order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const float* at = base + buf * NT * TILE + li * LSTR + lh * 32;
    const float* wt = at + MB * TILE;
    float4 av[MB][8], wv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int m = 0; m < MB; ++m) av[m][j] = *reinterpret_cast<const float4*>(at + m * TILE + 4 * j);
      wv[j] = *reinterpret_cast<const float4*>(wt + 4 * j);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {  // MB independent accumulator chains share every W fragment
#pragma unroll
      for (int m = 0; m < MB; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m][j].x, wv[j].x, acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < MB; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m][j].y, wv[j].y, acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < MB; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m][j].z, wv[j].z, acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < MB; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m][j].w, wv[j].w, acc[m], 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
  };
#pragma unroll
  for (int r = 0; r < RING; ++r) fetch(ra[r], rw[r], r);
  // epilogue operands AFTER the ring's first loads: vmcnt retires in order, so anything issued before chunk 0 would be
  // waited for with it, and xw_t / c_{t-1} are first-touch lines (HBM latency) while h and W_hh come from L2.  Issued
  // here their latency hides behind the whole K loop.
#pragma unroll
  for (int g = 0; g < NS; ++g) xg[g] = 0.f;
  if (eok) {
    const long o = (long)eb * NS * H + ej;
#pragma unroll
    for (int g = 0; g < NS; ++g) xg[g] = p.xw[o + (long)g * H];
    // ovr 5 (GPLSTMCell gate type 5, model.py:1759-1760): the cell state enters through the GPNN -- zsave holds
    // c_{t-1} Wg^T (blm_lstm_step_dh, launched before this step); + bias -> z, kept for the backward pass
    cprev = p.ovr == 5 ? p.zsave[(long)eb * H + ej] + p.rbias[ej] : p.cprev[(long)eb * H + ej];
  }
  if constexpr (PIPE && MB == 2) {
    // The pipelined loop for two batch tiles: per chunk 64 MFMAs (two independent accumulator chains sharing every W fragment)
    // with the next chunk's 24 ds_write_b128 (+ the 24 refill loads) and 24 fragment ds_read_b128 issued between them, one LDS
    // instruction per MFMA slot, each slot its own scheduling region (see the one-tile form below).  Whole chunks only.
    static_assert(RING == 2 && NBUF == 1 && REFILL && !TAIL, "the pipelined two-tile loop: ring of two, whole chunks");
    float4 fa[2][MB][8], fw[2][8];
    float* dst = base + soff;
    const float* at = base + li * LSTR + lh * 32;
    auto comp = [](const float4& v, int e) { return e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w)); };
    auto comb = [&](const float4 (&ca)[MB][8], const float4 (&cw)[8], float4 (&a)[MB][8], float4 (&w)[8], int cn, float4 (&na)[MB][8],
                    float4 (&nw)[8]) {
      const int so = kb4 + 128 * min(cn + RING, nchunk - 1);
#pragma unroll
      for (int k = 0; k < 32; ++k) {
#pragma unroll
        for (int m = 0; m < MB; ++m) {
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(ca[m][k >> 2], k & 3), comp(cw[k >> 2], k & 3), acc[m], 0, 0, 0);
          const int slot = k * MB + m;  // 0 .. 63
          if (slot < 24) {         // ds_write_b128 of the next chunk: tile slot % 3 (h tile 0, h tile 1, W), row group slot / 3
            const int t = slot % 3, q = slot / 3;
            if (t < 2) {
              *reinterpret_cast<float4*>(dst + t * TILE + 4 * q * LSTR) = a[t][q];
              a[t][q] = ldg(arsrc, aoff[t][q], so);
            } else {
              *reinterpret_cast<float4*>(dst + 2 * TILE + 4 * q * LSTR) = w[q];
              w[q] = ldg(wrsrc, woff[q], so);
            }
            if (slot == 23) {      // the tiles are complete before they are read back (compiler order; LDS is in order per wave)
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
              __builtin_amdgcn_wave_barrier();
            }
          } else if (slot < 48) {  // ds_read_b128 of the next chunk's fragments
            const int r = slot - 24, t = r % 3, j = r / 3;
            if (t < 2) na[t][j] = *reinterpret_cast<const float4*>(at + t * TILE + 4 * j);
            else nw[j] = *reinterpret_cast<const float4*>(at + 2 * TILE + 4 * j);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __builtin_amdgcn_wave_barrier();
    };
    {  // prologue: chunk 0 staged alone, its registers refilled with chunk RING
      const int so = kb4 + 128 * min(RING, nchunk - 1);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
#pragma unroll
        for (int m = 0; m < MB; ++m) {
          *reinterpret_cast<float4*>(dst + m * TILE + 4 * q * LSTR) = ra[0][m][q];
          ra[0][m][q] = ldg(arsrc, aoff[m][q], so);
        }
        *reinterpret_cast<float4*>(dst + MB * TILE + 4 * q * LSTR) = rw[0][q];
        rw[0][q] = ldg(wrsrc, woff[q], so);
      }
      LSTM_STAMP(1);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int m = 0; m < MB; ++m) fa[0][m][j] = *reinterpret_cast<const float4*>(at + m * TILE + 4 * j);
        fw[0][j] = *reinterpret_cast<const float4*>(at + MB * TILE + 4 * j);
      }
      __builtin_amdgcn_wave_barrier();
    }
#pragma unroll 1
    for (int cc = 0; cc + 2 < nchunk; cc += 2) {
      comb(fa[0], fw[0], ra[1], rw[1], cc + 1, fa[1], fw[1]);
      comb(fa[1], fw[1], ra[0], rw[0], cc + 2, fa[0], fw[0]);
    }
    comb(fa[0], fw[0], ra[1], rw[1], nchunk - 1, fa[1], fw[1]);
#pragma unroll
    for (int k = 0; k < 32; ++k)
#pragma unroll
      for (int m = 0; m < MB; ++m)
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(fa[1][m][k >> 2], k & 3), comp(fw[1][k >> 2], k & 3), acc[m], 0, 0, 0);
  } else if constexpr (PIPE) {
    static_assert(RING == 2 && NBUF == 1 && MB == 1, "the pipelined loop walks chunk pairs through one staging tile");
    float4 fa[2][8], fw[2][8];  // fragment sets: one feeds the MFMAs while the other is filled
    float* dst = base + soff;
    const float* at = base + li * LSTR + lh * 32;
    const float* wt = at + TILE;
    auto put2 = [&](float4 (&a)[8], float4 (&w)[8], int c, int q0) {  // 4 ds_write_b128: rows q0, q0 + 1 of both tiles
      const bool in = 32 * c + 4 * spart < Kh;
#pragma unroll
      for (int q = q0; q < q0 + 2; ++q) {
        *reinterpret_cast<float4*>(dst + 4 * q * LSTR) = fill(a[q], in);
        *reinterpret_cast<float4*>(dst + TILE + 4 * q * LSTR) = fill(w[q], in);
      }
    };
    auto refill2 = [&](float4 (&a)[8], float4 (&w)[8], int c, int q0) {  // 4 global loads of chunk c into the freed registers
      const int offu = 32 * min(c, nchunk - 1);
      const int off = TAIL ? min(offu + 4 * spart, Kh - 4) - 4 * spart : 0;
#pragma unroll
      for (int q = q0; q < q0 + 2; ++q) {
        if constexpr (TAIL) { a[q] = ldg(arsrc, aoff[0][q] + 4u * off, kb4); w[q] = ldg(wrsrc, woff[q] + 4u * off, kb4); }
        else { a[q] = ldg(arsrc, aoff[0][q], kb4 + 4 * offu); w[q] = ldg(wrsrc, woff[q], kb4 + 4 * offu); }
      }
    };
    auto get2 = [&](float4 (&fa_)[8], float4 (&fw_)[8], int j0) {  // 4 ds_read_b128
#pragma unroll
      for (int j = j0; j < j0 + 2; ++j) {
        fa_[j] = *reinterpret_cast<const float4*>(at + 4 * j);
        fw_[j] = *reinterpret_cast<const float4*>(wt + 4 * j);
      }
    };
    auto mfma4 = [&](const float4& a, const float4& w) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, w.x, acc[0], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, w.y, acc[0], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, w.z, acc[0], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, w.w, acc[0], 0, 0, 0);
    };
    // MFMAs of the fragments in (ca, cw) interleaved with the staging of chunk cn (registers a, w) into (na, nw): 32 slots
    // of ONE MFMA + ONE LDS instruction (+ one refill load), each its own scheduling region so that the order survives
    auto comp = [](const float4& v, int e) { return e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w)); };
    auto comb = [&](const float4 (&ca)[8], const float4 (&cw)[8], float4 (&a)[8], float4 (&w)[8], int cn, float4 (&na)[8],
                    float4 (&nw)[8]) {
      const bool in = 32 * cn + 4 * spart < Kh;
      const int offu = 32 * min(cn + RING, nchunk - 1);
      const int off = TAIL ? min(offu + 4 * spart, Kh - 4) - 4 * spart : 0;
      const int so = TAIL ? kb4 : kb4 + 4 * offu;
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(ca[k >> 2], k & 3), comp(cw[k >> 2], k & 3), acc[0], 0, 0, 0);
        if (k < 16) {  // ds_write_b128 of the next chunk: k even -> h tile row q, k odd -> W tile row q
          const int q = k >> 1;
          if ((k & 1) == 0) {
            *reinterpret_cast<float4*>(dst + 4 * q * LSTR) = fill(a[q], in);
            if constexpr (REFILL) a[q] = ldg(arsrc, TAIL ? aoff[0][q] + 4u * off : aoff[0][q], so);
          } else {
            *reinterpret_cast<float4*>(dst + TILE + 4 * q * LSTR) = fill(w[q], in);
            if constexpr (REFILL) w[q] = ldg(wrsrc, TAIL ? woff[q] + 4u * off : woff[q], so);
          }
          if (k == 15) {  // the tile is complete before it is read back (compiler order only; LDS is in order per wave)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
          }
        } else {        // ds_read_b128 of the next chunk's fragments
          const int j = (k - 16) >> 1;
          if ((k & 1) == 0) na[j] = *reinterpret_cast<const float4*>(at + 4 * j);
          else nw[j] = *reinterpret_cast<const float4*>(wt + 4 * j);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_wave_barrier();
    };
    // prologue: chunk 0 staged alone
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      put2(ra[0][0], rw[0], 0, 2 * j);
      if constexpr (REFILL) refill2(ra[0][0], rw[0], RING, 2 * j);
    }
    LSTM_STAMP(1);
#ifdef BLM_LSTM_PROF
    if (lane == 0) stamps[6] = clock64();
#endif
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 4; ++j) get2(fa[0], fw[0], 2 * j);
    __builtin_amdgcn_wave_barrier();
#pragma unroll 1
    for (int cc = 0; cc + 2 < nchunk; cc += 2) {
      comb(fa[0], fw[0], ra[1][0], rw[1], cc + 1, fa[1], fw[1]);
      comb(fa[1], fw[1], ra[0][0], rw[0], cc + 2, fa[0], fw[0]);
    }
    comb(fa[0], fw[0], ra[1][0], rw[1], nchunk - 1, fa[1], fw[1]);
#pragma unroll
    for (int j = 0; j < 8; ++j) mfma4(fa[1][j], fw[1][j]);
  } else if constexpr (REFILL) {
#pragma unroll 1
    for (int cc = 0; cc < nchunk; cc += RING) {  // nchunk % RING == 0 (host picks RING)
#pragma unroll
      for (int r = 0; r < RING; ++r) chunk(ra[r], rw[r], r & (NBUF - 1), cc + r);
    }
  } else {  // nchunk == RING
#pragma unroll
    for (int r = 0; r < RING; ++r) chunk(ra[r], rw[r], r & (NBUF - 1), r);
  }

  // cross-wave reduction of the NW K slices: red[wave][batch row][gate row]
  LSTM_STAMP(2);
#ifdef BLM_LSTM_PROF
  if (lane == 0) stamps[7] = clock64();
#endif
  __syncthreads();  // every wave is done with its staging tiles: the reduction buffer overlays them
  LSTM_STAMP(3);
  float* red = sm;  // NW x 32 MB x RSTR floats = 20 / 40 KB
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
      red[((wave * MB + m) * 32 + row) * RSTR + li] = acc[m][r];
    }
  __syncthreads();
  LSTM_STAMP(4);
#ifdef BLM_LSTM_PROF
  struct ProfTail {
    const LstmStepP& p; long long* st; int wave, lane;
    __device__ ~ProfTail() {
      if (p.prof && lane == 0) {
        st[5] = wall_clock64();
        long long* o = p.prof + ((long)(blockIdx.y * gridDim.x + blockIdx.x) * NW + wave) * 8;
        for (int i = 0; i < 8; ++i) o[i] = st[i];
      }
    }
  } prof_tail{p, stamps, wave, lane};
#endif
  if (eok) {
    float hw[NS];
#pragma unroll
    for (int g = 0; g < NS; ++g) {
      const int n = g * U + eu;
      constexpr int WS = 32 * MB;  // batch rows per wave slab of the reduction buffer (brow < WS)
      hw[g] = (red[(0 * WS + brow) * RSTR + n] + red[(1 * WS + brow) * RSTR + n]) +
              (red[(2 * WS + brow) * RSTR + n] + red[(3 * WS + brow) * RSTR + n]);
      if constexpr (NW == 8)
        hw[g] += (red[(4 * WS + brow) * RSTR + n] + red[(5 * WS + brow) * RSTR + n]) +
                 (red[(6 * WS + brow) * RSTR + n] + red[(7 * WS + brow) * RSTR + n]);
    }
    const long i = (long)eb * H + ej, o = (long)eb * NS * H + ej;
    if constexpr (NS == 8) {  // search cell: eight activations, four probs-weighted mixes (search.hip search_cell_fwd_kernel)
      float a[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float z = xg[k] + hw[k];
        a[k] = (k & 3) == 2 ? tanhf(z) : sigmoidf_(z);
      }
      const float gi = a[0] * p.probs[0] + a[4] * p.probs[1];
      const float gf = a[1] * p.probs[2] + a[5] * p.probs[3];
      const float gg = a[2] * p.probs[4] + a[6] * p.probs[5];
      const float go = a[3] * p.probs[6] + a[7] * p.probs[7];
      const float cn = gf * cprev + gi * gg;
      p.c[i] = cn;
      p.h[i] = go * tanhf(cn);
      if (p.ga) {
#pragma unroll
        for (int k = 0; k < 8; ++k) p.ga[o + (long)k * H] = a[k];
      }
    } else {
      float s[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v = hw[g];
        if (p.ovr == 4) {  // wave-uniform
          const float z = v + p.rbias[g * H + ej];
          if (p.zsave) p.zsave[o + (long)g * H] = z;
          v = gp_mix(z, p.coef, 4 * H, g * H + ej);
        }
        s[g] = xg[g] + v;
      }
      float gi = sigmoidf_(s[0]), gf = sigmoidf_(s[1]), gg = tanhf(s[2]), go = sigmoidf_(s[3]);
      if (p.ovr == 5) {  // wave-uniform: c_in = mixture(z)
        p.zsave[i] = cprev;
        cprev = gp_mix(cprev, p.coef, H, ej);
      }
      if (p.ovr >= 0 && p.ovr < 4) {  // wave-uniform
        const float z = p.ovr == 0 ? s[0] : (p.ovr == 1 ? s[1] : (p.ovr == 2 ? s[2] : s[3]));
        const float a = gp_mix(z, p.coef, H, ej);
        if (p.ovr == 0) gi = a; else if (p.ovr == 1) gf = a; else if (p.ovr == 2) gg = a; else go = a;
        if (p.zsave) p.zsave[i] = z;
      }
      const float cn = gf * cprev + gi * gg;
      p.c[i] = cn;
      p.h[i] = go * tanhf(cn) + (p.hnoise ? p.hnoise[ej] : 0.f);
      if (p.ga) {
        p.ga[o] = gi;
        p.ga[o + H] = gf;
        p.ga[o + 2L * H] = gg;
        p.ga[o + 3L * H] = go;
      }
    }
  }
}


template <int RING, int NS = 4, bool REFILL = true, int NW = 4, bool PIPE = false, bool TAIL = true, int MB = 1>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu((PIPE || MB > 1) ? 1 : 2, (PIPE || MB > 1) ? 1 : 2))) void lstm_step_fwd_kernel(const LstmStepP p) {
  lstm_step_fwd_body<RING, NS, REFILL, NW, PIPE, TAIL, MB>(p);
}

// ------------------------------------------------------------------ forward step, tiny batches (B <= 4)
// The n-best scorer walks its carry chain as ONE long B = 1 sequence (compute_sentence_scores.py, reference :271-274):
// thousands of dependent steps whose matrix is 4H x H but whose batch is a single row.  A 32 x 32 MFMA tile is 97 % idle
// there and the staged-tile kernel above still pays its whole load -> LDS -> MFMA -> LDS-reduce chain (~11 us).  Here a
// WAVE owns one hidden unit: its four gate rows of W_hh (4 x H floats, 16 KB at H = 1024) stream through registers as
// 16-byte loads, all issued before the first use, against the B rows of h; four wave reductions; lane 0 does the
// cell.  No LDS, no barrier, 4 H / 256 floats per lane: the step is launch + one L2 round trip.
template <int BB>
__device__ __forceinline__ void lstm_step_gemv_body(const LstmStepP& p) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + wave;
  const int H = p.H, B = p.B;
  if (j >= H) return;
  float acc[4][BB];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int b = 0; b < BB; ++b) acc[g][b] = 0.f;
  const float* wr = p.whh + (long)j * H;
  const long gs = (long)H * H;  // gate block stride
#pragma unroll 4
  for (int k0 = lane * 4; k0 < H; k0 += 256) {
    float4 w[4], hv[BB];
#pragma unroll
    for (int g = 0; g < 4; ++g) w[g] = *reinterpret_cast<const float4*>(wr + g * gs + k0);
#pragma unroll
    for (int b = 0; b < BB; ++b) hv[b] = *reinterpret_cast<const float4*>(p.hprev + (long)min(b, B - 1) * H + k0);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int b = 0; b < BB; ++b)
        acc[g][b] += w[g].x * hv[b].x + w[g].y * hv[b].y + w[g].z * hv[b].z + w[g].w * hv[b].w;
  }
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int b = 0; b < BB; ++b) acc[g][b] = wave_sum(acc[g][b]);
  if (lane == 0) {
#pragma unroll
    for (int b = 0; b < BB; ++b) {
      if (b < B) {
        const long i = (long)b * H + j, o = (long)b * 4 * H + j;
        const float gi = sigmoidf_(p.xw[o] + acc[0][b]), gf = sigmoidf_(p.xw[o + H] + acc[1][b]);
        const float gg = tanhf(p.xw[o + 2L * H] + acc[2][b]), go = sigmoidf_(p.xw[o + 3L * H] + acc[3][b]);
        const float cn = gf * p.cprev[i] + gi * gg;
        p.c[i] = cn;
        p.h[i] = go * tanhf(cn) + (p.hnoise ? p.hnoise[j] : 0.f);
        if (p.ga) {
          p.ga[o] = gi;
          p.ga[o + H] = gf;
          p.ga[o + 2L * H] = gg;
          p.ga[o + 3L * H] = go;
        }
      }
    }
  }
}
template <int BB>
__global__ __launch_bounds__(256) void lstm_step_fwd_gemv_kernel(const LstmStepP p) { lstm_step_gemv_body<BB>(p); }
// Two independent recurrences in one launch (blockIdx.y picks the step): layer 1 at step t and layer 2 one chunk behind --
// the scorer's B = 1 carry chain is two layers x thousands of ~5 us launches, half of it launch gap
// (blm_lstm_seq_pair_fwd below).
template <int BB>
__global__ __launch_bounds__(256) void lstm_step_fwd_gemv_pair_kernel(const LstmStepP p0, const LstmStepP p1) {
  if (blockIdx.y == 0) lstm_step_gemv_body<BB>(p0); else lstm_step_gemv_body<BB>(p1);
}

// ------------------------------------------------------------------ backward step
// dh_{t-1} = dgates_t . W_hh  (B x 4H times 4H x H), fused with the cell backward of step t-1, so that
// one launch per time step replaces cell kernel + memset + split-K GEMM and dh never visits HBM.
// Both operands are row-contiguous along the contraction index n (dgates_t rows, and rows of the
// TRANSPOSED recurrent weight W_hh^T (H x 4H) that the host makes once per layer and backward pass),
// so the kernel has the forward kernel's shape with v_mfma_f32_16x16x4_f32 tiles: a workgroup owns a
// 16 (batch) x 16 (hidden unit) tile of dh -- 256 workgroups at B = 64, H = 1024 -- wave g contracts
// over gate block g (n in [gH, gH+H)), lane quarter kq over the kq-th quarter of it.  Staged tiles
// are [16 rows][4 quarters x (32 + 4 pad)] with a 152-float row stride: conflict-free ds_read_b128
// for every 16-lane group of the instruction (searched).
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int BSTR = 152;                 // staged row stride (floats)
constexpr int BQ = 36;                    // quarter offset inside a row
constexpr int BTILE = 16 * BSTR;
constexpr int BWAVE_LDS = 2 * BTILE;      // dgates tile + W^T tile
constexpr int BRSTR = 20;                 // reduction row stride

struct LstmBwdP {
  const float *dg, *wt;                   // dgates_t (B,4H), W_hh^T (H,4H)
  const float *dy, *dc_next, *cprev, *c, *ga;
  float *dg_out, *dc_prev, *dh_out;
  const float *zprev, *coef;              // GP gate (see LstmStepP): z of step t-1's overridden gate, coef (4,H)
  float* dact_out;                        //   gradient w.r.t. that gate's mixture value (for the coef gradient)
  float* dz_out;                          // mode 4: dgates_out * mixture'(z) (B,4H) = the next launch's A operand
  int ovr;
  int B, H;
  int G;                                  // contraction length = row length of dg and wt: 4H (LSTM), 8H (search cell)
  long ldo;                               // row stride of dh_out (blm_lstm_step_dh_ld: the product lands in a column window)
  // blm_lstm_step_dh_act: the GPNN2 activation sum (elementwise.hip gpnn2_actsum_*) applied to the product on its way out
  //   act_mode 1: the product is the feature matrix f -> act_feat[.., ld_f] = f,  dh_out = actsum(f) * scale | 1 | 0
  //   act_mode 2: the product is d s               -> dh_out = d s * actsum'(act_feat) * scale on columns < act_M, else 0
  //   act_mode 3: the product is the activation that REPLACES gate f_gate of an LSTM cell (GPNN2 on a gate) -- the cell
  //                forward (elementwise.hip lstm_cell_ovr_fwd_kernel) runs behind it: pre-activations f_xw + f_hw (B,4H),
  //                f_c = c_{t-1}; writes h, c and the activated gates; dh_out still receives the product
  float* act_feat;
  int act_mode, act_M, act_ldf, acts;
  float act_scale;
  // second output of a plain product: add_out[b][k - add_lo] = product[b][k] + add_src[b * add_ld + k] for k in [add_lo, add_lo + add_n)
  const float* add_src;
  long add_ld;
  float* add_out;
  int add_lo, add_n;
  const float *f_xw, *f_hw, *f_c;
  float *f_h, *f_cn, *f_ga;
  int f_gate;
};

__device__ __forceinline__ float gpnn2_actsum_dev(float z, int acts) {
  float v = z;
  if (acts & 1) v += tanhf(z);
  if (acts & 2) v += sigmoidf_(z);
  if (acts & 4) v += fmaxf(z, 0.f);
  if (acts & 8) v += gelu_erf(z);
  return v;
}
__device__ __forceinline__ float gpnn2_dactsum_dev(float z, int acts) {
  float d = 1.f;
  if (acts & 1) { const float th = tanhf(z); d += 1.f - th * th; }
  if (acts & 2) { const float sg = sigmoidf_(z); d += sg * (1.f - sg); }
  if (acts & 4) d += z > 0.f ? 1.f : 0.f;
  if (acts & 8) d += dgelu_erf(z);
  return d;
}


// PIPE: software-pipelined K loop; TAIL = false: whole 32-float chunks only (no zero-fill selects, scalar chunk offsets) -- see the forward kernel
template <int RING, bool REFILL = true, int NW = 4, bool PIPE = false, bool TAIL = true>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(PIPE ? 1 : 2, PIPE ? 1 : 2))) void lstm_step_bwd_kernel(const LstmBwdP p) {
  constexpr int NBUF = 1;  // see the forward kernel; NW = 8 (two waves per SIMD): plain / GP cells only
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4;
  const int k0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
  const int H = p.H, B = p.B;
  const long G = p.G, G4 = 4L * H;       // G4: row length of the cell arrays (ga, dg_out, zprev of mode 4)
  const int Kw = p.G / NW;                // contraction run of one wave
  const int Kq = Kw >> 2;                 // ... and of one lane quarter
  const int nchunk = (Kq + 31) >> 5;
  float* base = sm + wave * (NBUF * 2 * BTILE);

  // epilogue roles (their operand loads are issued behind the first ring fetch)
  const int brow = threadIdx.x >> 4, ecol = threadIdx.x & 15;
  const int eb = b0 + brow, ek = k0 + ecol;
  const bool eok = eb < B && brow < 16;
  const long ei = (long)eb * H + ek, eo = (long)eb * G4 + ek;
  float e_dy = 0.f, e_dcn = 0.f, e_cp = 0.f, e_c = 0.f, e_z = 0.f, e_g[4] = {0.f, 0.f, 0.f, 0.f}, e_z4[4] = {0.f, 0.f, 0.f, 0.f};
  float e_a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, e_p8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // staging roles: one instruction moves 2 rows x 4 quarters x 128 B
  const int srow = lane >> 5, squart = (lane >> 3) & 3, spart = lane & 7;
  // buffer loads (descriptor + 32-bit row offset + scalar slice / chunk offset): see the forward kernel
  const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dg), 0, (int)(4u * (uint32_t)B * (uint32_t)p.G), 0x00020000);
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wt), 0, (int)(4u * (uint32_t)H * (uint32_t)p.G), 0x00020000);
  const int kb4 = __builtin_amdgcn_readfirstlane(wave * Kw * 4);
  uint32_t aoff[8], woff[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int row = 2 * q + srow;
    aoff[q] = (uint32_t)(((long)min(b0 + row, B - 1) * G + squart * Kq + 4 * spart) * 4);
    woff[q] = (uint32_t)(((long)(k0 + row) * G + squart * Kq + 4 * spart) * 4);
  }
  auto ldg = [](__amdgpu_buffer_rsrc_t r, uint32_t voff, int soff) {
    const bu32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
  };
  auto fill = [](const float4& x, bool in) {
    if constexpr (TAIL) return make_float4(in ? x.x : 0.f, in ? x.y : 0.f, in ? x.z : 0.f, in ? x.w : 0.f);
    else return x;
  };
  // chunk c of every staged row: per-lane clamp in the tail form, one scalar offset otherwise
  auto chunk_off = [&](int c, uint32_t& voff_add, int& so) {
    const int offu = 32 * min(c, nchunk - 1);
    if constexpr (TAIL) { voff_add = 4u * (uint32_t)(min(offu + 4 * spart, Kq - 4) - 4 * spart); so = kb4; }
    else { voff_add = 0u; so = kb4 + 4 * offu; }
  };
  const int soff = srow * BSTR + squart * BQ + 4 * spart;
  float4 ra[RING][8], rw[RING][8];
  auto fetch = [&](float4 (&a)[8], float4 (&w)[8], int c) {
    uint32_t va; int so;
    chunk_off(c, va, so);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      a[q] = ldg(arsrc, aoff[q] + va, so);
      w[q] = ldg(wrsrc, woff[q] + va, so);
    }
  };
  f32x4 acc = (f32x4)(0.f);
  auto chunk = [&](float4 (&a)[8], float4 (&w)[8], int buf, int c) {
    const bool in = 32 * c + 4 * spart < Kq;
    float* d = base + buf * 2 * BTILE + soff;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      *reinterpret_cast<float4*>(d + 2 * q * BSTR) = fill(a[q], in);
      *reinterpret_cast<float4*>(d + BTILE + 2 * q * BSTR) = fill(w[q], in);
    }
    if constexpr (REFILL) fetch(a, w, c + RING);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // wave-private tiles, in-order LDS: see the forward kernel
    __builtin_amdgcn_wave_barrier();
    const float* at = base + buf * 2 * BTILE + li * BSTR + lq * BQ;
    const float* wt = at + BTILE;
    float4 av[8], wv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      av[j] = *reinterpret_cast<const float4*>(at + 4 * j);
      wv[j] = *reinterpret_cast<const float4*>(wt + 4 * j);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].x, wv[j].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].y, wv[j].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].z, wv[j].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].w, wv[j].w, acc, 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
  };
#pragma unroll
  for (int r = 0; r < RING; ++r) fetch(ra[r], rw[r], r);
  // epilogue operands behind the ring's first loads (in-order vmcnt: see the forward kernel)
  if (eok && p.dg_out) {
    if (p.dy) e_dy = p.dy[ei];
    if (p.dc_next) e_dcn = p.dc_next[ei];
    e_cp = p.cprev[ei];
    e_c = p.c[ei];
    if ((p.ovr >= 0 && p.ovr < 4) || p.ovr == 5) e_z = p.zprev[ei];
    if (p.ovr == 4) {
#pragma unroll
      for (int g = 0; g < 4; ++g) e_z4[g] = p.zprev[eo + (long)g * H];
    }
    if (p.ovr != 8) {
#pragma unroll
      for (int g = 0; g < 4; ++g) e_g[g] = p.ga[eo + (long)g * H];
    }
  }
  if (eok && p.act_mode == 3) {  // cell forward behind the product: its operands travel behind the first ring fetch as well
    e_cp = p.f_c[ei];
#pragma unroll
    for (int g = 0; g < 4; ++g) e_g[g] = p.f_xw[eo + (long)g * H] + p.f_hw[eo + (long)g * H];
  }
  // ovr == 8: the architecture-search cell (search.hip search_cell_bwd_kernel) fused behind the product: ga / dg_out
  // have 8H-float rows [i f g o | i' f' g' o'], coef = the (4,2) mixing weights, dact_out = per-block partials of their gradient
  if (p.ovr == 8) {
#pragma unroll
    for (int k = 0; k < 8; ++k) e_p8[k] = p.coef[k];
    if (eok) {
#pragma unroll
      for (int k = 0; k < 8; ++k) e_a8[k] = p.ga[(long)eb * 8 * H + ek + (long)k * H];
    }
  }

  if constexpr (PIPE) {
    static_assert(RING == 2 && NBUF == 1 && REFILL, "the pipelined loop walks chunk pairs through one staging tile");
    float4 fa[2][8], fw[2][8];
    float* dst = base + soff;
    const float* at = base + li * BSTR + lq * BQ;
    const float* wt = at + BTILE;
    auto comp = [](const float4& v, int e) { return e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w)); };
    // 32 slots of ONE MFMA + ONE LDS instruction (+ one refill load), each its own scheduling region
    auto comb = [&](const float4 (&ca)[8], const float4 (&cw)[8], float4 (&a)[8], float4 (&w)[8], int cn, float4 (&na)[8],
                    float4 (&nw)[8]) {
      const bool in = 32 * cn + 4 * spart < Kq;
      uint32_t va; int so;
      chunk_off(cn + RING, va, so);
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(ca[k >> 2], k & 3), comp(cw[k >> 2], k & 3), acc, 0, 0, 0);
        if (k < 16) {
          const int q = k >> 1;
          if ((k & 1) == 0) {
            *reinterpret_cast<float4*>(dst + 2 * q * BSTR) = fill(a[q], in);
            a[q] = ldg(arsrc, aoff[q] + va, so);
          } else {
            *reinterpret_cast<float4*>(dst + BTILE + 2 * q * BSTR) = fill(w[q], in);
            w[q] = ldg(wrsrc, woff[q] + va, so);
          }
          if (k == 15) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
          }
        } else {
          const int j = (k - 16) >> 1;
          if ((k & 1) == 0) na[j] = *reinterpret_cast<const float4*>(at + 4 * j);
          else nw[j] = *reinterpret_cast<const float4*>(wt + 4 * j);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_wave_barrier();
    };
    {  // prologue: chunk 0 staged alone
      const bool in = 4 * spart < Kq;
      uint32_t va; int so;
      chunk_off(RING, va, so);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        *reinterpret_cast<float4*>(dst + 2 * q * BSTR) = fill(ra[0][q], in);
        *reinterpret_cast<float4*>(dst + BTILE + 2 * q * BSTR) = fill(rw[0][q], in);
        ra[0][q] = ldg(arsrc, aoff[q] + va, so);
        rw[0][q] = ldg(wrsrc, woff[q] + va, so);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        fa[0][j] = *reinterpret_cast<const float4*>(at + 4 * j);
        fw[0][j] = *reinterpret_cast<const float4*>(wt + 4 * j);
      }
      __builtin_amdgcn_wave_barrier();
    }
#pragma unroll 1
    for (int cc = 0; cc + 2 < nchunk; cc += 2) {
      comb(fa[0], fw[0], ra[1], rw[1], cc + 1, fa[1], fw[1]);
      comb(fa[1], fw[1], ra[0], rw[0], cc + 2, fa[0], fw[0]);
    }
    comb(fa[0], fw[0], ra[1], rw[1], nchunk - 1, fa[1], fw[1]);
#pragma unroll
    for (int k = 0; k < 32; ++k)
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(fa[1][k >> 2], k & 3), comp(fw[1][k >> 2], k & 3), acc, 0, 0, 0);
  } else if constexpr (REFILL) {
#pragma unroll 1
    for (int cc = 0; cc < nchunk; cc += RING) {
#pragma unroll
      for (int r = 0; r < RING; ++r) chunk(ra[r], rw[r], r & (NBUF - 1), cc + r);
    }
  } else {
#pragma unroll
    for (int r = 0; r < RING; ++r) chunk(ra[r], rw[r], r & (NBUF - 1), r);
  }

  __syncthreads();
  float* red = sm;  // NW x 16 x BRSTR floats, overlays the staging tiles
#pragma unroll
  for (int r = 0; r < 4; ++r) red[(wave * 16 + 4 * lq + r) * BRSTR + li] = acc[r];
  __syncthreads();
  if (p.ovr == 8) {  // wave-uniform.  Every thread stays for the block reduction of the mixing-weight partials.
    const float dh8 = (red[(0 * 16 + brow) * BRSTR + ecol] + red[(1 * 16 + brow) * BRSTR + ecol]) +
                      (red[(2 * 16 + brow) * BRSTR + ecol] + red[(3 * 16 + brow) * BRSTR + ecol]);
    float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (eok) {
      const float gi = e_a8[0] * e_p8[0] + e_a8[4] * e_p8[1], gf = e_a8[1] * e_p8[2] + e_a8[5] * e_p8[3];
      const float gg = e_a8[2] * e_p8[4] + e_a8[6] * e_p8[5], go = e_a8[3] * e_p8[6] + e_a8[7] * e_p8[7];
      const float tc = tanhf(e_c);
      const float dhv = dh8 + e_dy;
      const float dc = e_dcn + dhv * go * (1.f - tc * tc);
      const float dgate[4] = {dc * gg, dc * e_cp, dc * gi, dhv * tc};
      p.dc_prev[ei] = dc * gf;
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const float av = e_a8[k + 4 * q];
          const float dact = k == 2 ? 1.f - av * av : av * (1.f - av);
          p.dg_out[(long)eb * 8 * H + ek + (long)(k + 4 * q) * H] = dgate[k] * e_p8[2 * k + q] * dact;
          s8[2 * k + q] = dgate[k] * av;
        }
    }
    __syncthreads();  // everybody has read its dh: the reduction buffer is free again
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float w = wave_sum(s8[k]);
      if (lane == 0) sm[wave * 8 + k] = w;
    }
    __syncthreads();
    if (threadIdx.x < 8)
      p.dact_out[8L * ((long)blockIdx.y * gridDim.x + blockIdx.x) + threadIdx.x] =
          (sm[threadIdx.x] + sm[8 + threadIdx.x]) + (sm[16 + threadIdx.x] + sm[24 + threadIdx.x]);
    return;
  }
  if (!eok) return;
  float dh = (red[(0 * 16 + brow) * BRSTR + ecol] + red[(1 * 16 + brow) * BRSTR + ecol]) +
             (red[(2 * 16 + brow) * BRSTR + ecol] + red[(3 * 16 + brow) * BRSTR + ecol]);
  if constexpr (NW == 8)
    dh += (red[(4 * 16 + brow) * BRSTR + ecol] + red[(5 * 16 + brow) * BRSTR + ecol]) +
          (red[(6 * 16 + brow) * BRSTR + ecol] + red[(7 * 16 + brow) * BRSTR + ecol]);
  if (p.act_mode == 1) {  // wave-uniform
    p.act_feat[(long)eb * p.act_ldf + ek] = dh;
    dh = ek < p.act_M ? gpnn2_actsum_dev(dh, p.acts) * p.act_scale : (ek == p.act_M ? 1.f : 0.f);
  } else if (p.act_mode == 2) {
    dh = ek < p.act_M ? dh * gpnn2_dactsum_dev(p.act_feat[(long)eb * p.act_ldf + ek], p.acts) * p.act_scale : 0.f;
  }
  if (p.dh_out) p.dh_out[(long)eb * p.ldo + ek] = dh;
  if (p.add_out && (unsigned)(ek - p.add_lo) < (unsigned)p.add_n)
    p.add_out[(long)eb * p.add_n + (ek - p.add_lo)] = dh + p.add_src[(long)eb * p.add_ld + ek];
  if (p.act_mode == 3) {  // wave-uniform
    const float gi = p.f_gate == 0 ? dh : sigmoidf_(e_g[0]);
    const float gf = p.f_gate == 1 ? dh : sigmoidf_(e_g[1]);
    const float gg = p.f_gate == 2 ? dh : tanhf(e_g[2]);
    const float go = p.f_gate == 3 ? dh : sigmoidf_(e_g[3]);
    const float cn = gf * e_cp + gi * gg;
    p.f_cn[ei] = cn;
    p.f_h[ei] = go * tanhf(cn);
    p.f_ga[eo] = gi;
    p.f_ga[eo + H] = gf;
    p.f_ga[eo + 2L * H] = gg;
    p.f_ga[eo + 3L * H] = go;
    return;
  }
  if (p.dg_out) {  // cell backward of the step that produced h_{t-1} (elementwise.hip lstm_cell_bwd_kernel)
    if (p.ovr == 5) e_cp = gp_mix(e_z, p.coef, H, ek);  // gate type 5: the cell saw the GPNN mixture of z = c_{t-2} Wg^T + b
    const float gi = e_g[0], gf = e_g[1], gg = e_g[2], go = e_g[3];
    const float tc = tanhf(e_c);
    const float dhv = dh + e_dy;
    const float dc = e_dcn + dhv * go * (1.f - tc * tc);
    const float dgi = dc * gg, dgf = dc * e_cp, dgg = dc * gi, dgo = dhv * tc;  // w.r.t. the activated gates
    float o0 = dgi * gi * (1.f - gi), o1 = dgf * gf * (1.f - gf), o2 = dgg * (1.f - gg * gg), o3 = dgo * go * (1.f - go);
    if (p.ovr >= 0 && p.ovr < 4) {  // wave-uniform: that gate's activation was the GPNN mixture of its pre-activation z
      const float da = p.ovr == 0 ? dgi : (p.ovr == 1 ? dgf : (p.ovr == 2 ? dgg : dgo));
      const float dz = da * dgp_mix(e_z, p.coef, H, ek);
      if (p.ovr == 0) o0 = dz; else if (p.ovr == 1) o1 = dz; else if (p.ovr == 2) o2 = dz; else o3 = dz;
      if (p.dact_out) p.dact_out[ei] = da;
    }
    p.dg_out[eo] = o0;
    p.dg_out[eo + H] = o1;
    p.dg_out[eo + 2L * H] = o2;
    p.dg_out[eo + 3L * H] = o3;
    if (p.ovr == 4) {  // the hidden projection went through the mixture: what the next product needs is d z
      p.dz_out[eo] = o0 * dgp_mix(e_z4[0], p.coef, 4 * H, ek);
      p.dz_out[eo + H] = o1 * dgp_mix(e_z4[1], p.coef, 4 * H, H + ek);
      p.dz_out[eo + 2L * H] = o2 * dgp_mix(e_z4[2], p.coef, 4 * H, 2 * H + ek);
      p.dz_out[eo + 3L * H] = o3 * dgp_mix(e_z4[3], p.coef, 4 * H, 3 * H + ek);
    }
    p.dc_prev[ei] = dc * gf;
    if (p.ovr == 5) {  // d mixture value (for the coefficient gradient) and d z (the A operand of the next dc product)
      p.dact_out[ei] = dc * gf;
      p.dz_out[ei] = dc * gf * dgp_mix(e_z, p.coef, H, ek);
    }
  }
}

// out (cols x rows) = in (rows x cols)^T, 32x32 tiles through LDS
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols) {
  __shared__ float t[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = r0 + ty + 8 * j, c = c0 + tx;
    if (r < rows && c < cols) t[ty + 8 * j][tx] = in[(long)r * cols + c];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = c0 + ty + 8 * j, r = r0 + tx;
    if (r < rows && c < cols) out[(long)c * rows + r] = t[tx][ty + 8 * j];
  }
}

}  // namespace blm

using namespace blm;

// options "lstm_gemv" / "lstm_pipe" / "lstm_tail" (blm_set_option): the tiny-batch step kernel, the software-pipelined K loop
// and the general (K tail) form of the pipelined kernels -- each form is built and parity-tested
static int lstm_gemv() { return blm::option(blm::OPT_LSTM_GEMV); }
static int lstm_pipe() { return blm::option(blm::OPT_LSTM_PIPE); }
static int lstm_mb2() { return blm::option(blm::OPT_LSTM_MB2); }
static int lstm_tail() { return blm::option(blm::OPT_LSTM_TAIL); }

extern "C" int blm_lstm_step_fwd_gp(const float*, const float*, const float*, const float*, float*, float*, float*, const float*, int,
                                    const float*, const float*, float*, int, int, void*);
extern "C" int blm_lstm_step_bwd_gp(const float*, const float*, const float*, const float*, const float*, const float*, const float*,
                                    float*, float*, float*, int, const float*, const float*, float*, float*, int, int, void*);

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int blm_lstm_step_fwd(const float* xw_t, const float* w_hh, const float* h_prev, const float* c_prev, float* h,
                                 float* c, float* gates_act, const float* h_noise, int B, int H, void* stream) {
  return blm_lstm_step_fwd_gp(xw_t, w_hh, h_prev, c_prev, h, c, gates_act, h_noise, -1, nullptr, nullptr, nullptr, B, H, stream);
}

extern "C" int blm_lstm_step_fwd_gp(const float* xw_t, const float* w_hh, const float* h_prev, const float* c_prev, float* h,
                                    float* c, float* gates_act, const float* h_noise, int gate_ovr, const float* coef4,
                                    const float* rbias, float* z_out, int B, int H, void* stream) {
  if (!xw_t || !w_hh || !h_prev || !c_prev || !h || !c || B < 0 || H < 0 || gate_ovr > 5 || (gate_ovr >= 0 && !coef4) ||
      (gate_ovr >= 4 && !rbias) || (gate_ovr == 5 && !z_out))
    return blm_fail(BLM_ERR_INVALID, "blm_lstm_step_fwd: bad arguments");
  if ((long)B * H == 0) return BLM_OK;
  if (H % 32 != 0 || !al16(w_hh) || !al16(h_prev) || 16.0 * H * H >= 4294967296.0 || 4.0 * B * H >= 4294967296.0)  // 32-bit byte offsets
    return blm_fail(BLM_ERR_UNSUPPORTED, "blm_lstm_step_fwd: needs H % 32 == 0, 16-byte aligned h_prev / w_hh and operands under 4 GB");
  LstmStepP p{xw_t, w_hh, h_prev, c_prev, h, c, gates_act, h_noise, coef4, z_out, gate_ovr < 0 ? -1 : gate_ovr, rbias, B, H, nullptr};
  if (gate_ovr < 0 && B <= 4 && H % 4 == 0 && lstm_gemv()) {  // tiny batches (the scorer's carry chain): one wave per hidden unit
    const dim3 g((H + 3) / 4), blk(256);
    hipStream_t s0 = (hipStream_t)stream;
    if (B == 1) hipLaunchKernelGGL(lstm_step_fwd_gemv_kernel<1>, g, blk, 0, s0, p);
    else if (B == 2) hipLaunchKernelGGL(lstm_step_fwd_gemv_kernel<2>, g, blk, 0, s0, p);
    else hipLaunchKernelGGL(lstm_step_fwd_gemv_kernel<4>, g, blk, 0, s0, p);
    BLM_HIP(hipGetLastError());
    return BLM_OK;
  }
  const size_t lds4 = (size_t)4 * WAVE_LDS * sizeof(float);
  static bool once = false;
  if (!once) {
    BLM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_step_fwd_kernel<1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));
    BLM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_step_fwd_kernel<2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));
    BLM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_step_fwd_kernel<2, 4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));
    BLM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_step_fwd_kernel<2, 4, true, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));
    BLM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_step_fwd_kernel<2, 4, true, 4, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));
    once = true;
  }
  const int nchunk = (H / 8 + 31) / 32;  // 32-k chunks per lane half of a wave's K quarter
  const dim3 grid(H / 8, (B + 31) / 32), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (nchunk % 2 == 0 && nchunk >= 4 && lstm_pipe()) {
    if (H % 256 == 0 && !lstm_tail()) hipLaunchKernelGGL((lstm_step_fwd_kernel<2, 4, true, 4, true, false>), grid, block, lds4, st, p);  // whole chunks only
    else hipLaunchKernelGGL((lstm_step_fwd_kernel<2, 4, true, 4, true>), grid, block, lds4, st, p);
  } else if (nchunk == 2) hipLaunchKernelGGL((lstm_step_fwd_kernel<2, 4, false>), grid, block, lds4, st, p);
  else if (nchunk % 2 == 0) hipLaunchKernelGGL((lstm_step_fwd_kernel<2, 4>), grid, block, lds4, st, p);
  else hipLaunchKernelGGL((lstm_step_fwd_kernel<1, 4>), grid, block, lds4, st, p);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int blm_lstm_seq_fwd(const float* xw, const float* w_hh, float* hs, float* cs, float* gates_act,
                                const float* noise_rows, int T, int B, int H, void* stream) {
  if (!blm::extents_ok({T, B, H, 4}) || (T > 0 && (!xw || !w_hh || !hs || !cs)))
    return blm_fail(BLM_ERR_INVALID, "blm_lstm_seq_fwd: bad arguments");
  const size_t bh = (size_t)B * H, bg = 4 * bh;
  for (int t = 0; t < T; ++t) {
    const int rc = blm_lstm_step_fwd(xw + t * bg, w_hh, hs + t * bh, cs + t * bh, hs + (t + 1) * bh, cs + (t + 1) * bh,
                                     gates_act ? gates_act + t * bg : nullptr, noise_rows ? noise_rows + (size_t)t * H : nullptr,
                                     B, H, stream);
    if (rc) return rc;
  }
  return BLM_OK;
}

extern "C" int blm_lstm_seq_pair_fwd(const float* xw_a, const float* w_hh_a, float* hs_a, float* cs_a, float* ga_a, int n_a,
                                     const float* xw_b, const float* w_hh_b, float* hs_b, float* cs_b, float* ga_b, int n_b,
                                     int B, int H, void* stream) {
  if (!blm::extents_ok({n_a, B, H, 4}) || !blm::extents_ok({n_b, B, H, 4}) || B < 1 || H < 1 || (n_a > 0 && (!xw_a || !w_hh_a || !hs_a || !cs_a)) || (n_b > 0 && (!xw_b || !w_hh_b || !hs_b || !cs_b)))
    return blm_fail(BLM_ERR_INVALID, "blm_lstm_seq_pair_fwd: bad arguments");
  const size_t bh = (size_t)B * H, bg = 4 * bh;
  const bool pairable = B <= 4 && H % 4 == 0 && lstm_gemv() && (n_a == 0 || (al16(w_hh_a) && al16(hs_a))) && (n_b == 0 || (al16(w_hh_b) && al16(hs_b))) &&
                        (bh * sizeof(float)) % 16 == 0;
  hipStream_t s0 = (hipStream_t)stream;
  const int n = n_a > n_b ? n_a : n_b;
  for (int t = 0; t < n; ++t) {
    const bool a = t < n_a, b = t < n_b;
    if (a && b && pairable) {
      const LstmStepP pa{xw_a + t * bg, w_hh_a, hs_a + t * bh, cs_a + t * bh, hs_a + (t + 1) * bh, cs_a + (t + 1) * bh,
                         ga_a ? ga_a + t * bg : nullptr, nullptr, nullptr, nullptr, -1, nullptr, B, H, nullptr};
      const LstmStepP pb{xw_b + t * bg, w_hh_b, hs_b + t * bh, cs_b + t * bh, hs_b + (t + 1) * bh, cs_b + (t + 1) * bh,
                         ga_b ? ga_b + t * bg : nullptr, nullptr, nullptr, nullptr, -1, nullptr, B, H, nullptr};
      const dim3 g((H + 3) / 4, 2), blk(256);
      if (B == 1) hipLaunchKernelGGL(lstm_step_fwd_gemv_pair_kernel<1>, g, blk, 0, s0, pa, pb);
      else if (B == 2) hipLaunchKernelGGL(lstm_step_fwd_gemv_pair_kernel<2>, g, blk, 0, s0, pa, pb);
      else hipLaunchKernelGGL(lstm_step_fwd_gemv_pair_kernel<4>, g, blk, 0, s0, pa, pb);
      BLM_HIP(hipGetLastError());
      continue;
    }
    if (a) {
      const int rc = blm_lstm_step_fwd(xw_a + t * bg, w_hh_a, hs_a + t * bh, cs_a + t * bh, hs_a + (t + 1) * bh, cs_a + (t + 1) * bh,
                                       ga_a ? ga_a + t * bg : nullptr, nullptr, B, H, stream);
      if (rc) return rc;
    }
    if (b) {
      const int rc = blm_lstm_step_fwd(xw_b + t * bg, w_hh_b, hs_b + t * bh, cs_b + t * bh, hs_b + (t + 1) * bh, cs_b + (t + 1) * bh,
                                       ga_b ? ga_b + t * bg : nullptr, nullptr, B, H, stream);
      if (rc) return rc;
    }
  }
  return BLM_OK;
}

extern "C" int blm_lstm_search_step_fwd(const float* xw8_t, const float* w8_hh, const float* h_prev, const float* c_prev,
                                        const float* probs, float* h, float* c, float* acts8, int B, int H, void* stream) {
  if (!xw8_t || !w8_hh || !h_prev || !c_prev || !probs || !h || !c || B < 0 || H < 0)
    return blm_fail(BLM_ERR_INVALID, "blm_lstm_search_step_fwd: bad arguments");
  if ((long)B * H == 0) return BLM_OK;
  if (H % 32 != 0 || !al16(w8_hh) || !al16(h_prev) || 32.0 * H * H >= 4294967296.0 || 4.0 * B * H >= 4294967296.0)
    return blm_fail(BLM_ERR_UNSUPPORTED, "blm_lstm_search_step_fwd: needs H % 32 == 0, 16-byte aligned h_prev / w8_hh and operands under 4 GB");
  LstmStepP p{xw8_t, w8_hh, h_prev, c_prev, h, c, acts8, nullptr, nullptr, nullptr, -1, nullptr, B, H, probs};
  const size_t lds = (size_t)4 * WAVE_LDS * sizeof(float);
  static bool once = false;
  if (!once) {
    BLM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_step_fwd_kernel<1, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    BLM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_step_fwd_kernel<2, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    BLM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_step_fwd_kernel<2, 8, true, 4, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    once = true;
  }
  const int nchunk = (H / 8 + 31) / 32;
  const dim3 grid(H / 4, (B + 31) / 32), block(256);
  hipStream_t st = (hipStream_t)stream;
  // B > 32: two batch tiles per workgroup -- the stacked weight streams once, H / 4 workgroups per 64 batch rows (one round of the
  // chip at H = 1024) instead of two rounds of the one-tile form (12.2 us per round, tools/search_step_bench.py)
  if (B > 32 && nchunk % 2 == 0 && H % 256 == 0 && lstm_mb2() && !lstm_tail()) {
    const size_t lds3 = (size_t)4 * 3 * TILE * sizeof(float);
    static bool once3 = false;
    if (!once3) {
      BLM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_step_fwd_kernel<2, 8, true, 4, false, false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
      BLM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_step_fwd_kernel<2, 8, true, 4, true, false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
      once3 = true;
    }
    if (nchunk >= 4 && lstm_pipe())
      hipLaunchKernelGGL((lstm_step_fwd_kernel<2, 8, true, 4, true, false, 2>), dim3(H / 4, (B + 63) / 64), block, lds3, st, p);
    else
      hipLaunchKernelGGL((lstm_step_fwd_kernel<2, 8, true, 4, false, false, 2>), dim3(H / 4, (B + 63) / 64), block, lds3, st, p);
    BLM_HIP(hipGetLastError());
    return BLM_OK;
  }
  if (nchunk % 2 == 0 && nchunk >= 4 && H % 256 == 0 && lstm_pipe() && !lstm_tail())  // pipelined K loop, whole chunks (see blm_lstm_step_fwd)
    hipLaunchKernelGGL((lstm_step_fwd_kernel<2, 8, true, 4, true, false>), grid, block, lds, st, p);
  else if (nchunk % 2 == 0) hipLaunchKernelGGL((lstm_step_fwd_kernel<2, 8>), grid, block, lds, st, p);
  else hipLaunchKernelGGL((lstm_step_fwd_kernel<1, 8>), grid, block, lds, st, p);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int blm_transpose(const float* in, float* out, int rows, int cols, void* stream) {
  if (!in || !out || !blm::extents_ok({rows, cols})) return blm_fail(BLM_ERR_INVALID, "blm_transpose: bad arguments");
  if ((long)rows * cols == 0) return BLM_OK;
  hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, (hipStream_t)stream, in, out, rows, cols);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

static int launch_step_bwd(const LstmBwdP& p, void* stream) {
  if (4.0 * p.H * p.G >= 4294967296.0 || 4.0 * p.B * p.G >= 4294967296.0)  // the kernels address both operands with 32-bit byte offsets
    return blm_fail(BLM_ERR_UNSUPPORTED, "blm_lstm_step_bwd: operands of 4 GB and more are not supported by the fused step");
  const size_t lds4 = (size_t)4 * BWAVE_LDS * sizeof(float);
  static bool once = false;
  if (!once) {
    BLM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_step_bwd_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));
    BLM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_step_bwd_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));
    BLM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>((lstm_step_bwd_kernel<2, true, 4, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));
    BLM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>((lstm_step_bwd_kernel<2, true, 4, true, false>)), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));
    once = true;
  }
  const int nchunk = (p.G / 16 + 31) / 32;  // 32-k chunks per lane quarter of a wave's contraction run
  const dim3 grid(p.H / 16, (p.B + 15) / 16), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (nchunk % 2 == 0 && nchunk >= 4 && lstm_pipe()) {
    if (p.G % 512 == 0 && !lstm_tail()) hipLaunchKernelGGL((lstm_step_bwd_kernel<2, true, 4, true, false>), grid, block, lds4, st, p);  // whole chunks only
    else hipLaunchKernelGGL((lstm_step_bwd_kernel<2, true, 4, true>), grid, block, lds4, st, p);
  }
  else if (nchunk % 2 == 0) hipLaunchKernelGGL(lstm_step_bwd_kernel<2>, grid, block, lds4, st, p);
  else hipLaunchKernelGGL(lstm_step_bwd_kernel<1>, grid, block, lds4, st, p);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int blm_lstm_step_bwd(const float* dgates_t, const float* w_hh_t, const float* dy_prev, const float* dc_next,
                                 const float* c_prev, const float* c, const float* gates_act, float* dgates_out,
                                 float* dc_prev, float* dh_out, int B, int H, void* stream) {
  return blm_lstm_step_bwd_gp(dgates_t, w_hh_t, dy_prev, dc_next, c_prev, c, gates_act, dgates_out, dc_prev, dh_out, -1, nullptr,
                              nullptr, nullptr, nullptr, B, H, stream);
}

extern "C" int blm_lstm_step_bwd_gp(const float* dgates_t, const float* w_hh_t, const float* dy_prev, const float* dc_next,
                                    const float* c_prev, const float* c, const float* gates_act, float* dgates_out,
                                    float* dc_prev, float* dh_out, int gate_ovr, const float* coef4, const float* z_prev,
                                    float* dact_out, float* dz_out, int B, int H, void* stream) {
  if (!dgates_t || !w_hh_t || B < 0 || H < 0 || (!dgates_out && !dh_out) ||
      (dgates_out && (!c_prev || !c || !gates_act || !dc_prev)) || gate_ovr > 5 ||
      (dgates_out && gate_ovr >= 0 && (!coef4 || !z_prev)) || (dgates_out && gate_ovr >= 4 && !dz_out) ||
      (dgates_out && gate_ovr == 5 && !dact_out))
    return blm_fail(BLM_ERR_INVALID, "blm_lstm_step_bwd: bad arguments");
  if ((long)B * H == 0) return BLM_OK;
  if (H % 32 != 0 || !al16(dgates_t) || !al16(w_hh_t))
    return blm_fail(BLM_ERR_UNSUPPORTED, "blm_lstm_step_bwd: needs H % 32 == 0 and 16-byte aligned dgates_t / w_hh_t");
  LstmBwdP p{dgates_t, w_hh_t, dy_prev, dc_next, c_prev, c, gates_act, dgates_out, dc_prev, dh_out, z_prev, coef4, dact_out,
             dz_out, (dgates_out && gate_ovr >= 0) ? gate_ovr : -1, B, H, 4 * H, (long)H, nullptr, 0, 0, 0, 0, 0.f};
  return launch_step_bwd(p, stream);
}

// The backward steps t_hi-1 .. t_lo of one layer from ONE call (the counterpart of blm_lstm_seq_fwd: with two recurrences in flight on
// two streams a step lasts ~5 us of device time per launch, less than a ctypes call costs).  Step T-1 is the plain cell backward
// (dh = dh_T), every earlier one the fused step.  dc_pair (2,B,H): slot k holds dc of the step above on entry; the slots alternate.
extern "C" int blm_lstm_cell_bwd2(const float*, const float*, const float*, const float*, const float*, const float*, float*, float*, int,
                                  int, void*);
extern "C" int blm_lstm_seq_bwd(const float* dh_T, const float* dy, const float* cs, const float* gates_act, const float* w_hh_t,
                                float* dgates, float* dc_pair, int k, float* dh_rows, int T, int t_hi, int t_lo, int B, int H,
                                void* stream) {
  if (!dy || !cs || !gates_act || !w_hh_t || !dgates || !dc_pair || (k != 0 && k != 1) || T < 0 || t_lo < 0 || t_hi > T || t_lo > t_hi ||
      B < 0 || H < 0 || (t_hi == T && t_hi > t_lo && !dh_T))
    return blm_fail(BLM_ERR_INVALID, "blm_lstm_seq_bwd: bad arguments");
  const size_t bh = (size_t)B * H, bg = 4 * bh;
  for (int t = t_hi - 1; t >= t_lo; --t) {
    float* dc_in = dc_pair + (size_t)k * bh;
    float* dc_out = dc_pair + (size_t)(k ^ 1) * bh;
    const int rc = t == T - 1
                       ? blm_lstm_cell_bwd2(dh_T, dy + t * bh, dc_in, cs + t * bh, cs + (t + 1) * bh, gates_act + t * bg, dgates + t * bg,
                                            dc_out, B, H, stream)
                       : blm_lstm_step_bwd(dgates + (t + 1) * bg, w_hh_t, dy + t * bh, dc_in, cs + t * bh, cs + (t + 1) * bh,
                                           gates_act + t * bg, dgates + t * bg, dc_out, dh_rows ? dh_rows + t * bh : nullptr, B, H, stream);
    if (rc) return rc;
    k ^= 1;
  }
  return BLM_OK;
}

extern "C" int64_t blm_lstm_search_step_partials(int B, int H) {
  if (!blm::extents_ok({B, H})) return 0;
  return 8 * (int64_t)(H / 16) * ((B + 15) / 16);
}

extern "C" int blm_lstm_search_step_bwd(const float* dz8_t, const float* w8_t, const float* dy_prev, const float* dc_next,
                                        const float* c_prev, const float* c, const float* acts8, const float* probs,
                                        float* dz8_out, float* dc_prev, float* partial, int B, int H, void* stream) {
  if (!dz8_t || !w8_t || !c_prev || !c || !acts8 || !probs || !dz8_out || !dc_prev || !partial || B < 0 || H < 0)
    return blm_fail(BLM_ERR_INVALID, "blm_lstm_search_step_bwd: bad arguments");
  if ((long)B * H == 0) return BLM_OK;
  if (H % 16 != 0 || !al16(dz8_t) || !al16(w8_t))
    return blm_fail(BLM_ERR_UNSUPPORTED, "blm_lstm_search_step_bwd: needs H % 16 == 0 and 16-byte aligned dz8_t / w8_t");
  LstmBwdP p{dz8_t, w8_t, dy_prev, dc_next, c_prev, c, acts8, dz8_out, dc_prev, nullptr, nullptr, probs, partial, nullptr, 8,
             B, H, 8 * H, (long)H, nullptr, 0, 0, 0, 0, 0.f};
  return launch_step_bwd(p, stream);
}

extern "C" int blm_lstm_step_dh_ld(const float* dz, const float* w_t, float* dh_out, int64_t ldo, int B, int H, int G, void* stream) {
  if (!dz || !w_t || !dh_out || B < 0 || H < 0 || G < 0 || ldo < H) return blm_fail(BLM_ERR_INVALID, "blm_lstm_step_dh: bad arguments");
  if ((long)B * H == 0) return BLM_OK;
  if (H % 16 != 0 || G % 64 != 0 || G == 0 || !al16(dz) || !al16(w_t))
    return blm_fail(BLM_ERR_UNSUPPORTED, "blm_lstm_step_dh: needs H % 16 == 0, G % 64 == 0 and 16-byte aligned dz / w_t");
  LstmBwdP p{dz, w_t, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, dh_out, nullptr, nullptr, nullptr, nullptr, -1,
             B, H, G, (long)ldo, nullptr, 0, 0, 0, 0, 0.f};
  return launch_step_bwd(p, stream);
}

extern "C" int blm_lstm_step_dh_act(const float* dz, const float* w_t, float* out, int64_t ldo, int B, int H, int G, int act_mode,
                                    float* feat, int ld_f, int M, float scale, int acts, void* stream) {
  if (!dz || !w_t || !out || !feat || B < 0 || H < 0 || G < 0 || ldo < H || (act_mode != 1 && act_mode != 2) || M < 0 || M >= H ||
      ld_f < (act_mode == 1 ? H : M) || (acts & ~15))
    return blm_fail(BLM_ERR_INVALID, "blm_lstm_step_dh_act: bad arguments");
  if ((long)B * H == 0) return BLM_OK;
  if (H % 16 != 0 || G % 64 != 0 || G == 0 || !al16(dz) || !al16(w_t))
    return blm_fail(BLM_ERR_UNSUPPORTED, "blm_lstm_step_dh_act: needs H % 16 == 0, G % 64 == 0 and 16-byte aligned dz / w_t");
  LstmBwdP p{dz, w_t, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, out, nullptr, nullptr, nullptr, nullptr, -1,
             B, H, G, (long)ldo, feat, act_mode, M, ld_f, acts, scale};
  return launch_step_bwd(p, stream);
}

extern "C" int blm_lstm_step_dh(const float* dz, const float* w_t, float* dh_out, int B, int H, int G, void* stream) {
  return blm_lstm_step_dh_ld(dz, w_t, dh_out, H, B, H, G, stream);
}

// Internal forms of the skinny product for the GPNN2 time loop below (same argument rules as blm_lstm_step_dh):
// ... with a second output, the column window [lo, lo + n) of the product plus the same window of `add` (row stride add_ld)
static int step_dh_add(const float* dz, const float* w_t, float* out, const float* add, long add_ld, float* add_out, int lo, int n,
                       int B, int H, int G, void* stream) {
  if ((long)B * H == 0) return BLM_OK;
  if (H % 16 != 0 || G % 64 != 0 || G == 0 || !al16(dz) || !al16(w_t))
    return blm_fail(BLM_ERR_UNSUPPORTED, "blm_lstm_gpnn2_seq_fwd: needs H % 16 == 0, G % 64 == 0 and 16-byte aligned operands");
  LstmBwdP p{dz, w_t, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, out, nullptr, nullptr, nullptr, nullptr, -1,
             B, H, G, (long)H, nullptr, 0, 0, 0, 0, 0.f, add, add_ld, add_out, lo, n};
  return launch_step_bwd(p, stream);
}
// ... with the LSTM cell forward behind it: the product (B,H) is the activation that replaces gate `gate`
static int step_dh_cell(const float* dz, const float* w_t, float* out, const float* xw, const float* hw, const float* c_prev, int gate,
                        float* h, float* c, float* ga, int B, int H, int G, void* stream) {
  if ((long)B * H == 0) return BLM_OK;
  if (H % 16 != 0 || G % 64 != 0 || G == 0 || !al16(dz) || !al16(w_t))
    return blm_fail(BLM_ERR_UNSUPPORTED, "blm_lstm_gpnn2_seq_fwd: needs H % 16 == 0, G % 64 == 0 and 16-byte aligned operands");
  LstmBwdP p{dz, w_t, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, out, nullptr, nullptr, nullptr, nullptr, -1,
             B, H, G, (long)H, nullptr, 3, 0, 0, 0, 0.f, nullptr, 0, nullptr, 0, 0, xw, hw, c_prev, h, c, ga, gate};
  return launch_step_bwd(p, stream);
}

// ------------------------------------------------------------------ GPNN2 cells: the time loops (host side)
// One C call per layer and direction: the launches are 3-9 us each and form one dependent chain (the sequences themselves:
// include/bayeslm.h blm_gpnn2_seq, ops._LSTMRecurrentGPNN2).
extern "C" int blm_lstm_cell_fwd(const float*, const float*, const float*, float*, float*, float*, int, int, void*);
extern "C" int blm_lstm_cell_bwd2(const float*, const float*, const float*, const float*, const float*, const float*, float*, float*, int,
                                  int, void*);

static int gpnn2_seq_check(const blm_gpnn2_seq* q, bool bwd, const char* who) {
  if (!q) return blm_fail(BLM_ERR_INVALID, "%s: null descriptor", who);
  if (q->abi_version != BLM_ABI_VERSION) return blm_fail(BLM_ERR_ABI, "%s: abi_version mismatch", who);
  if (q->mode < 0 || q->mode > 2 || q->T < 0 || q->B < 0 || q->H <= 0 || q->M <= 0 || q->MP <= q->M || q->GP < q->MP || q->nF < 1 ||
      (q->nF != 1 && q->nF != q->T) || (q->mode == 0 && (q->gate < 0 || q->gate > 3)))
    return blm_fail(BLM_ERR_INVALID, "%s: bad mode / shape", who);
  if (!bwd && (!q->xw || !q->FT || !q->cwp || !q->hs || !q->cs || !q->ga || !q->feat || !q->sact || !q->gout ||
               (q->mode == 0 && (!q->z4 || !q->pre)) || (q->mode != 2 && !q->w_hh)))
    return blm_fail(BLM_ERR_INVALID, "%s: null buffer", who);
  if (bwd && (!q->Fp || !q->cwt || !q->cs || !q->ga || !q->feat || !q->dy || !q->dh || !q->dcs2 || !q->dgates || !q->df ||
              (q->mode != 2 && (!q->w_hh_t || !q->da)) || (q->mode == 1 && !q->gout)))
    return blm_fail(BLM_ERR_INVALID, "%s: null backward buffer", who);
  return BLM_OK;
}

extern "C" int blm_lstm_gpnn2_seq_fwd(const blm_gpnn2_seq* q, void* stream) {
  if (int rc = gpnn2_seq_check(q, false, "blm_lstm_gpnn2_seq_fwd")) return rc;
  const int T = q->T, B = q->B, H = q->H, M = q->M, MP = q->MP, GP = q->GP;
  const size_t bh = (size_t)B * H, bg = 4 * bh, bm = (size_t)B * MP, bp = (size_t)B * GP, off = (size_t)q->gate * H;
  const float scale = 1.0f / sqrtf((float)M);
  for (int t = 0; t < T; ++t) {
    const float* FT = q->FT + (size_t)(q->nF > 1 ? t : 0) * MP * H;
    int rc = BLM_OK;
    if (q->mode == 0) {  // a gate's pre-activation through the GPNN2
      // three launches per step: h W_hh^T (all gates; the GPNN2 gate's pre-activation = its window + xw leaves as a second output),
      // features + activation sum, coefficient product with the cell update behind it
      rc = step_dh_add(q->hs + t * bh, q->w_hh, q->z4 + t * bg, q->xw + t * bg, 4L * H, q->pre + t * bh, (int)off, H, B, 4 * H, H, stream);
      if (!rc) rc = blm_lstm_step_dh_act(q->pre + t * bh, FT, q->sact + t * bp, GP, B, MP, H, 1, q->feat + t * bm, MP, M, scale, q->acts, stream);
      if (!rc) rc = step_dh_cell(q->sact + t * bp, q->cwp, q->gout + t * bh, q->xw + t * bg, q->z4 + t * bg, q->cs + t * bh, q->gate,
                                 q->hs + (t + 1) * bh, q->cs + (t + 1) * bh, q->ga + t * bg, B, H, GP, stream);
    } else if (q->mode == 1) {  // the cell state enters through the GPNN2: gout[t] = c_in
      rc = blm_lstm_step_dh_act(q->cs + t * bh, FT, q->sact + t * bp, GP, B, MP, H, 1, q->feat + t * bm, MP, M, scale, q->acts, stream);
      if (!rc) rc = blm_lstm_step_dh(q->sact + t * bp, q->cwp, q->gout + t * bh, B, H, GP, stream);
      if (!rc) rc = blm_lstm_step_fwd(q->xw + t * bg, q->w_hh, q->hs + t * bh, q->gout + t * bh, q->hs + (t + 1) * bh, q->cs + (t + 1) * bh,
                                      q->ga + t * bg, nullptr, B, H, stream);
    } else {  // the hidden projection of all four gates is the GPNN2 of h: gout[t] = hw (B,4H), cwp is (4H,GP)
      rc = blm_lstm_step_dh_act(q->hs + t * bh, FT, q->sact + t * bp, GP, B, MP, H, 1, q->feat + t * bm, MP, M, scale, q->acts, stream);
      if (!rc) rc = blm_lstm_step_dh(q->sact + t * bp, q->cwp, q->gout + t * bg, B, 4 * H, GP, stream);
      if (!rc) rc = blm_lstm_cell_fwd(q->xw + t * bg, q->gout + t * bg, q->cs + t * bh, q->hs + (t + 1) * bh, q->cs + (t + 1) * bh,
                                      q->ga + t * bg, B, H, stream);
    }
    if (rc) return rc;
  }
  return BLM_OK;
}

extern "C" int blm_lstm_gpnn2_seq_bwd(const blm_gpnn2_seq* q, void* stream) {
  if (int rc = gpnn2_seq_check(q, true, "blm_lstm_gpnn2_seq_bwd")) return rc;
  const int T = q->T, B = q->B, H = q->H, M = q->M, MP = q->MP, GP = q->GP;
  const size_t bh = (size_t)B * H, bg = 4 * bh, bm = (size_t)B * MP, bp = (size_t)B * GP, off = (size_t)q->gate * H;
  const float scale = 1.0f / sqrtf((float)M);
  int k = 0;  // dcs2 (2,B,H): dc ping-pong, [0] holds dc_T on entry; the final dc_0 is in dcs2[T & 1]
  for (int t = T - 1; t >= 0; --t) {
    const float* Fp = q->Fp + (size_t)(q->nF > 1 ? t : 0) * H * GP;
    float* dc_in = q->dcs2 + k * bh;
    float* dc_out = q->dcs2 + (k ^ 1) * bh;
    int rc = BLM_OK;
    if (q->mode == 0) {
      rc = blm_lstm_cell_ovr_bwd2(q->dh, q->dy + t * bh, dc_in, q->cs + t * bh, q->cs + (t + 1) * bh, q->ga + t * bg, q->gate,
                                  q->dgates + t * bg, q->da + t * bh, dc_out, B, H, stream);              // dh_t = recurrent part + dy_t
      if (!rc) rc = blm_lstm_step_dh_act(q->da + t * bh, q->cwt, q->df + t * bp, GP, B, GP, H, 2, q->feat + t * bm, MP, M, scale, q->acts,
                                         stream);                                                       // d f = (d a . [W | b]) * actsum'(f)
      if (!rc) rc = blm_lstm_step_dh_ld(q->df + t * bp, Fp, q->dgates + t * bg + off, 4 * H, B, H, GP, stream);  // d pre -> the gate's slot
      if (!rc) rc = blm_lstm_step_dh(q->dgates + t * bg, q->w_hh_t, q->dh, B, H, 4 * H, stream);         // dh_{t-1}
    } else if (q->mode == 1) {
      // cell backward of step t with c_in as its incoming cell state: dc_out <- d c_in; then d c_{t-1} = GPNN2 backward
      rc = blm_lstm_cell_bwd2(q->dh, q->dy + t * bh, dc_in, q->gout + t * bh, q->cs + (t + 1) * bh, q->ga + t * bg, q->dgates + t * bg,
                              q->da + t * bh, B, H, stream);
      if (!rc) rc = blm_lstm_step_dh_act(q->da + t * bh, q->cwt, q->df + t * bp, GP, B, GP, H, 2, q->feat + t * bm, MP, M, scale, q->acts,
                                         stream);
      if (!rc) rc = blm_lstm_step_dh(q->df + t * bp, Fp, dc_out, B, H, GP, stream);                      // raw d c_{t-1}
      if (!rc) rc = blm_lstm_step_dh(q->dgates + t * bg, q->w_hh_t, q->dh, B, H, 4 * H, stream);         // dh_{t-1}
    } else {
      rc = blm_lstm_cell_bwd2(q->dh, q->dy + t * bh, dc_in, q->cs + t * bh, q->cs + (t + 1) * bh, q->ga + t * bg, q->dgates + t * bg, dc_out,
                              B, H, stream);
      if (!rc) rc = blm_lstm_step_dh_act(q->dgates + t * bg, q->cwt, q->df + t * bp, GP, B, GP, 4 * H, 2, q->feat + t * bm, MP, M, scale,
                                         q->acts, stream);                                              // d f = (d hw . [W | b]) * actsum'(f)
      if (!rc) rc = blm_lstm_step_dh(q->df + t * bp, Fp, q->dh, B, H, GP, stream);                       // dh_{t-1} = d f . F_t^T
    }
    if (rc) return rc;
    k ^= 1;
  }
  return BLM_OK;
}
