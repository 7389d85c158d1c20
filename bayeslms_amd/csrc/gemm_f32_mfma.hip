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
#include "blm_device.h"
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
  int a_vec, b_vec;
  int gm, gn;
  // fused activation dropout
  int drop_on; uint32_t drop_thr; float drop_inv_keep; blm_rng drop_rng;
  int drop_B, drop_col_offset, drop_global_cols;
};

// keep factor of element (m, n) of a (rows, B, N) activation, keyed by global column
__device__ __forceinline__ float gemm_keep(const GemmP& p, int m, int n) {
  const int row = m / p.drop_B, b = m - row * p.drop_B;
  const uint64_t g = ((uint64_t)row * p.drop_global_cols + (uint64_t)(p.drop_col_offset + b)) * (uint64_t)p.N + (uint64_t)n;
  const u32x4 u = philox_block(p.drop_rng, g >> 2);
  const int c = (int)(g & 3);
  const uint32_t bits = c == 0 ? u.x : (c == 1 ? u.y : (c == 2 ? u.z : u.w));
  return bits >= p.drop_thr ? p.drop_inv_keep : 0.f;
}

constexpr int BK = 32;

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
template <int R, int S>
__device__ __forceinline__ void r2s_kmaj(float* tile, const float4 (&r)[R / 32]) {
  const int t = threadIdx.x, kq = t & 7, rr = t >> 3;
#pragma unroll
  for (int j = 0; j < R / 32; ++j) {
    float* d = tile + (4 * kq) * S + rr + 32 * j;
    d[0] = r[j].x;
    d[S] = r[j].y;
    d[2 * S] = r[j].z;
    d[3 * S] = r[j].w;
  }
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

__device__ __forceinline__ float gp_mix(float z, const float* coef, int N, int n) {
  return tanhf(z) * coef[n] + sigmoidf_(z) * coef[N + n] + fmaxf(z, 0.f) * coef[2 * N + n] +
         gelu_erf(z) * coef[3 * N + n];
}
__device__ __forceinline__ float dgp_mix(float z, const float* coef, int N, int n) {
  const float th = tanhf(z), sg = sigmoidf_(z);
  return (1.f - th * th) * coef[n] + sg * (1.f - sg) * coef[N + n] + (z > 0.f ? coef[2 * N + n] : 0.f) +
         dgelu_erf(z) * coef[3 * N + n];
}

template <int OP, int WTM, int WTN, bool SAMP>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(const GemmP p) {
  constexpr int BM = 64 * WTM, BN = 64 * WTN;
  constexpr bool A_KMAJ = (OP != BLM_GEMM_TN), B_KMAJ = (OP == BLM_GEMM_NT);
  constexpr int SA = A_KMAJ ? BM + 1 : BM + 4;
  constexpr int SB = B_KMAJ ? BN + 1 : BN + 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const As = smem;
  float* const Bs = smem + 2 * BK * SA;

  // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so give each XCD a
  // contiguous run of tile ids; n is fastest so neighbours reuse the same A panel out of that XCD's L2.
  const int nb = p.gm * p.gn, bid = blockIdx.x;
  const int q = nb >> 3, rem = nb & 7, xcd = bid & 7;
  const int id = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
  const int m0 = (id / p.gn) * BM, n0 = (id % p.gn) * BN;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;

  float4 ra[BM / 32], rb[BN / 32], rl[SAMP ? BN / 32 : 1];
  const bool av = p.a_vec, bv = p.b_vec;

  auto fetch = [&](int k0) {
    if constexpr (A_KMAJ) g2r_kmaj<BM>(p.A, p.lda, m0, p.M, k0, p.K, av, ra);
    else g2r_nmaj<BM>(p.A, p.lda, k0, p.K, m0, p.M, av, ra);
    if constexpr (B_KMAJ) g2r_kmaj<BN>(p.B, p.ldb, n0, p.N, k0, p.K, bv, rb);
    else g2r_nmaj<BN>(p.B, p.ldb, k0, p.K, n0, p.N, bv, rb);
    if constexpr (SAMP) {
      if constexpr (B_KMAJ) g2r_kmaj<BN>(p.vb.lgstd, p.vb_cols, n0 - p.vb.row_lo, p.vb.srows, k0, p.K, true, rl);
      else g2r_nmaj<BN>(p.vb.lgstd, p.vb_cols, k0 - p.vb.row_lo, p.vb.srows, n0, p.N, true, rl);
    }
  };
  auto stash = [&](int buf, int k0) {
    float* At = As + buf * BK * SA;
    float* Bt = Bs + buf * BK * SB;
    if constexpr (SAMP) {
      const int t = threadIdx.x;
#pragma unroll
      for (int j = 0; j < BN / 32; ++j) {
        int srow, scol;
        if constexpr (B_KMAJ) { srow = n0 + (t >> 3) + 32 * j; scol = k0 + 4 * (t & 7); }
        else { constexpr int TPR = BN / 4, RPP = 256 / TPR; srow = k0 + t / TPR + RPP * j; scol = n0 + 4 * (t % TPR); }
        rb[j] = sample4(rb[j], rl[j], p.vb, p.vb_cols, srow, scol);
      }
    }
    if constexpr (A_KMAJ) r2s_kmaj<BM, SA>(At, ra); else r2s_nmaj<BM, SA>(At, ra);
    if constexpr (B_KMAJ) r2s_kmaj<BN, SB>(Bt, rb); else r2s_nmaj<BN, SB>(Bt, rb);
  };

  f32x16 acc[WTM][WTN];
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j) acc[i][j] = (f32x16)(0.f);

  const int nk = (p.K + BK - 1) / BK;
  fetch(0);
  stash(0, 0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) fetch((kt + 1) * BK);
    const float* Ab = As + cur * BK * SA + lh * SA + wm * (32 * WTM) + li;
    const float* Bb = Bs + cur * BK * SB + lh * SB + wn * (32 * WTN) + li;
#pragma unroll
    for (int s = 0; s < BK / 2; ++s) {
      float a[WTM], b[WTN];
#pragma unroll
      for (int i = 0; i < WTM; ++i) a[i] = Ab[(2 * s) * SA + 32 * i];
#pragma unroll
      for (int j = 0; j < WTN; ++j) b[j] = Bb[(2 * s) * SB + 32 * j];
#pragma unroll
      for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) stash(cur ^ 1, (kt + 1) * BK);
    __syncthreads();
  }

  // ---- epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const bool accum = p.flags & BLM_GEMM_ACCUMULATE;
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j) {
      const int col = n0 + wn * (32 * WTN) + 32 * j + li;
      if (col >= p.N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * (32 * WTM) + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row >= p.M) continue;
        const long o = (long)row * p.ldc + col;
        float v = p.alpha * acc[i][j][r];
        switch (p.epi) {
          case BLM_EPI_BIAS: v += p.bias[col]; break;
          case BLM_EPI_BIAS_GELU: {
            v += p.bias[col];
            if (p.aux) p.aux[o] = v;
            v = gelu_erf(v);
            if (p.drop_on) v *= gemm_keep(p, row, col);
          } break;
          case BLM_EPI_MUL_DGELU: {
            v *= dgelu_erf(p.aux[o]);
            if (p.drop_on) v *= gemm_keep(p, row, col);
          } break;
          case BLM_EPI_GP_MIX: {
            v += p.bias[col];
            if (p.aux) p.aux[o] = v;
            v = gp_mix(v, p.coef, p.N, col);
            if (p.drop_on) v *= gemm_keep(p, row, col);
          } break;
          case BLM_EPI_MUL_DGP_MIX: {
            v *= dgp_mix(p.aux[o], p.coef, p.N, col);
            if (p.drop_on) v *= gemm_keep(p, row, col);
          } break;
          case BLM_EPI_BAYES_WGRAD: {
            const float dW = v;
            const int rel = row - p.vc.row_lo;
            if ((unsigned)rel < (unsigned)p.vc.srows) {
              const long si = (long)rel * p.N + col;
              const float sig = __expf(p.vc.lgstd[si]);
              float e;
              if (p.vc.eps) e = p.vc.eps[si];
              else {
                const float4 z = philox_normal4(p.vc.rng, (uint64_t)si >> 2);
                const int c = (int)(si & 3);
                e = c == 0 ? z.x : (c == 1 ? z.y : (c == 2 ? z.z : z.w));
              }
              const float g2 = dW * e * sig + p.kl_lambda * (sig * sig - 1.0f) * p.kl_inv_n;
              p.C2[si] = accum ? p.C2[si] + g2 : g2;
              v = dW + p.kl_lambda * p.wg_mu[o] * p.kl_inv_n;
            }
          } break;
          default: break;
        }
        p.C[o] = accum ? p.C[o] + v : v;
      }
    }
}

template <int OP, int WTM, int WTN, bool SAMP>
static int launch_cfg(const GemmP& p, hipStream_t st) {
  constexpr int BM = 64 * WTM, BN = 64 * WTN;
  constexpr bool A_KMAJ = (OP != BLM_GEMM_TN), B_KMAJ = (OP == BLM_GEMM_NT);
  constexpr int SA = A_KMAJ ? BM + 1 : BM + 4, SB = B_KMAJ ? BN + 1 : BN + 4;
  constexpr size_t lds = (size_t)2 * BK * (SA + SB) * sizeof(float);
  GemmP q = p;
  q.gm = (p.M + BM - 1) / BM;
  q.gn = (p.N + BN - 1) / BN;
  auto kern = gemm_f32_kernel<OP, WTM, WTN, SAMP>;
  static bool attr_done = false;  // per instantiation; benign race (idempotent)
  if (!attr_done) {
    BLM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds));
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(q.gm * q.gn), dim3(256), lds, st, q);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

template <int OP, bool SAMP>
static int launch_op(const GemmP& p, hipStream_t st) {
  // Tile choice: 128x128 when it fills the chip (>= 256 blocks) or the problem is large in both
  // dimensions; otherwise shrink the dimension that leaves CUs idle.
  const long b128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
  bool small_m = p.M <= 64, small_n = p.N <= 64;
  if (!small_m && !small_n && b128 < 256) {
    if (p.M <= p.N) small_m = true; else small_n = true;
    const long b2 = (long)((p.M + (small_m ? 63 : 127)) / (small_m ? 64 : 128)) *
                    ((p.N + (small_n ? 63 : 127)) / (small_n ? 64 : 128));
    if (b2 < 256) small_m = small_n = true;
  }
  if (small_m && small_n) return launch_cfg<OP, 1, 1, SAMP>(p, st);
  if (small_m) return launch_cfg<OP, 1, 2, SAMP>(p, st);
  if (small_n) return launch_cfg<OP, 2, 1, SAMP>(p, st);
  return launch_cfg<OP, 2, 2, SAMP>(p, st);
}

}  // namespace blm

using namespace blm;

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int blm_gemm(const blm_gemm_args* a, void* stream) {
  if (!a) return blm_fail(BLM_ERR_INVALID, "blm_gemm: null args");
  if (a->abi_version != BLM_ABI_VERSION) return blm_fail(BLM_ERR_ABI, "blm_gemm: abi_version mismatch");
  if (a->M < 0 || a->N < 0 || a->K < 0) return blm_fail(BLM_ERR_INVALID, "blm_gemm: negative dimension");
  if (a->M == 0 || a->N == 0) return BLM_OK;
  if (!a->A || !a->B || !a->C) return blm_fail(BLM_ERR_INVALID, "blm_gemm: null operand");
  if (a->op < BLM_GEMM_NT || a->op > BLM_GEMM_TN) return blm_fail(BLM_ERR_INVALID, "blm_gemm: bad op");
  const int amin = a->op == BLM_GEMM_TN ? a->M : a->K, bmin = a->op == BLM_GEMM_NT ? a->K : a->N;
  if (a->lda < amin || a->ldb < bmin || a->ldc < a->N) return blm_fail(BLM_ERR_INVALID, "blm_gemm: leading dimension too small");
  GemmP p{};
  p.M = a->M; p.N = a->N; p.K = a->K;
  p.A = a->A; p.lda = a->lda; p.B = a->B; p.ldb = a->ldb; p.C = a->C; p.ldc = a->ldc;
  p.alpha = a->alpha; p.flags = a->flags; p.epi = a->epilogue;
  p.bias = a->bias; p.aux = a->aux; p.coef = a->coef;
  p.vb = a->var_b; p.C2 = a->C2; p.wg_mu = a->wg_mu; p.vc = a->var_c;
  p.kl_lambda = a->kl_lambda; p.kl_inv_n = a->kl_inv_n;
  p.drop_on = a->drop_p > 0.f;
  if (p.drop_on) {
    if (a->drop_B <= 0 || a->M % a->drop_B != 0) return blm_fail(BLM_ERR_INVALID, "blm_gemm: drop_B must divide M");
    const double t = (double)a->drop_p * 4294967296.0;
    p.drop_thr = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
    p.drop_inv_keep = a->drop_p < 1.f ? 1.f / (1.f - a->drop_p) : 0.f;
    p.drop_rng = a->drop_rng;
    p.drop_B = a->drop_B; p.drop_col_offset = a->drop_col_offset;
    p.drop_global_cols = a->drop_global_cols > 0 ? a->drop_global_cols : a->drop_B;
  }
  p.a_vec = aligned16(a->A) && (a->lda % 4 == 0);
  p.b_vec = aligned16(a->B) && (a->ldb % 4 == 0);
  switch (a->epilogue) {
    case BLM_EPI_NONE: break;
    case BLM_EPI_BIAS: case BLM_EPI_BIAS_GELU:
      if (!a->bias) return blm_fail(BLM_ERR_INVALID, "blm_gemm: epilogue needs bias"); break;
    case BLM_EPI_MUL_DGELU:
      if (!a->aux) return blm_fail(BLM_ERR_INVALID, "blm_gemm: epilogue needs aux"); break;
    case BLM_EPI_GP_MIX:
      if (!a->bias || !a->coef) return blm_fail(BLM_ERR_INVALID, "blm_gemm: GP epilogue needs bias and coef"); break;
    case BLM_EPI_MUL_DGP_MIX:
      if (!a->aux || !a->coef) return blm_fail(BLM_ERR_INVALID, "blm_gemm: GP epilogue needs aux and coef"); break;
    case BLM_EPI_BAYES_WGRAD:
      if (a->op != BLM_GEMM_TN) return blm_fail(BLM_ERR_INVALID, "blm_gemm: BAYES_WGRAD needs op TN");
      if (!a->C2 || !a->wg_mu || !a->var_c.lgstd) return blm_fail(BLM_ERR_INVALID, "blm_gemm: BAYES_WGRAD needs C2, wg_mu, var_c.lgstd");
      if (a->var_c.row_lo < 0 || a->var_c.srows < 0 || a->var_c.row_lo + a->var_c.srows > a->M)
        return blm_fail(BLM_ERR_INVALID, "blm_gemm: var_c row window outside W");
      break;
    default: return blm_fail(BLM_ERR_INVALID, "blm_gemm: unknown epilogue");
  }
  const bool samp = a->var_b.lgstd != nullptr;
  if (samp) {
    if (a->op == BLM_GEMM_TN) return blm_fail(BLM_ERR_INVALID, "blm_gemm: var_b not valid for TN");
    // B source matrix is W: NT -> (N x K), NN -> (K x N)
    const int wrows = a->op == BLM_GEMM_NT ? a->N : a->K;
    p.vb_cols = a->op == BLM_GEMM_NT ? a->K : a->N;
    if (p.vb_cols % 4 != 0 || !aligned16(a->var_b.lgstd) || (a->var_b.eps && !aligned16(a->var_b.eps)) || !p.b_vec)
      return blm_fail(BLM_ERR_INVALID, "blm_gemm: fused sampling needs cols % 4 == 0 and 16-byte aligned mu/lgstd/eps");
    if (a->var_b.row_lo < 0 || a->var_b.srows < 0 || a->var_b.row_lo + a->var_b.srows > wrows)
      return blm_fail(BLM_ERR_INVALID, "blm_gemm: var_b row window outside W");
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (a->op) {
    case BLM_GEMM_NT: return samp ? launch_op<BLM_GEMM_NT, true>(p, st) : launch_op<BLM_GEMM_NT, false>(p, st);
    case BLM_GEMM_NN: return samp ? launch_op<BLM_GEMM_NN, true>(p, st) : launch_op<BLM_GEMM_NN, false>(p, st);
    default: return launch_op<BLM_GEMM_TN, false>(p, st);
  }
}
