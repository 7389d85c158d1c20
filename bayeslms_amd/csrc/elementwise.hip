// HBM-bound ops around the GEMMs: embedding(+PE+dropout), dropout, residual+dropout+LayerNorm
// (post-LN tail) forward/backward, cross-entropy forward+backward in one pass over the logits,
// bias-gradient column sums, global-norm clip + SGD momentum, LSTM cell pointwise, axpy.
//
// Dropout masks are Philox bits keyed by the GLOBAL element index of a (rows, global_cols, D)
// tensor, so a data-parallel run that shards columns reproduces the single-process mask.
#include "blm_device.h"
#include "blm_host.h"
#include "blm_dropkey.h"

namespace blm {

constexpr int TPB = 256;

// ------------------------------------------------------------------ embedding
// one wave per (t,b) row, 4 rows per block
__global__ __launch_bounds__(TPB) void embed_fwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ enc,
                                                        const float* __restrict__ pe, float* __restrict__ out, int T,
                                                        int B, int D, long vocab, float scale, DropKey dk) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
  if (row >= (long)T * B) return;
  const int t = (int)(row / B), b = (int)(row % B);
  long id = ids[row];
  if (id < 0 || id >= vocab) id = 0;  // host validates; never read out of bounds
  const float* e = enc + id * D;
  const float* pp = pe ? pe + (long)t * D : nullptr;
  float* o = out + row * D;
  if ((D & 3) == 0) {
    for (int j = lane * 4; j < D; j += 256) {
      float4 v = *reinterpret_cast<const float4*>(e + j);
      float4 q = pp ? *reinterpret_cast<const float4*>(pp + j) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 kp = keep4(dk, t, b, j);
      v.x = (v.x * scale + q.x) * kp.x; v.y = (v.y * scale + q.y) * kp.y;
      v.z = (v.z * scale + q.z) * kp.z; v.w = (v.w * scale + q.w) * kp.w;
      *reinterpret_cast<float4*>(o + j) = v;
    }
  } else {
    for (int j = lane; j < D; j += 64) o[j] = (e[j] * scale + (pp ? pp[j] : 0.f)) * keep1(dk, t, b, j);
  }
}

__global__ __launch_bounds__(TPB) void embed_bwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ dy,
                                                        float* __restrict__ denc, int T, int B, int D, long vocab,
                                                        float scale, DropKey dk) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
  if (row >= (long)T * B) return;
  const int t = (int)(row / B), b = (int)(row % B);
  long id = ids[row];
  if (id < 0 || id >= vocab) return;
  float* g = denc + id * D;
  const float* d = dy + row * D;
  // one dword per lane, 256 contiguous bytes per wave-instruction: the fast float-atomic shape
  for (int j = lane; j < D; j += 64) atomicAdd(g + j, d[j] * scale * keep1(dk, t, b, j));
}

// Deterministic form of the embedding gradient (blm_set_option("deterministic", 1)): ONE wave owns a vocabulary row, walks the
// window's token ids in position order (staged in LDS as 32-bit ids, `chunk` at a time) and adds the matching rows of dy in
// registers -- every output row has one writer and a fixed order of additions; no atomics.  Most waves find no match and only
// scan: n / 64 LDS reads + ballots per row and 512-column slab.
constexpr int EMB_DET_CHUNK = 8192;
__global__ __launch_bounds__(TPB) void embed_bwd_det_kernel(const int64_t* __restrict__ ids, const float* __restrict__ dy,
                                                            float* __restrict__ denc, int T, int B, int D, long vocab,
                                                            float scale, DropKey dk) {
  __shared__ int sid[EMB_DET_CHUNK];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long n = (long)T * B;
  const bool single = n <= EMB_DET_CHUNK;  // the whole window fits: staged once
  const long per_round = (long)gridDim.x * (TPB / 64);
  auto stage = [&](long base) {
    const long m = min((long)EMB_DET_CHUNK, n - base);
    for (long i = threadIdx.x; i < m; i += TPB) {
      const long id = ids[base + i];
      sid[i] = (id >= 0 && id < vocab) ? (int)id : -1;
    }
  };
  if (single) { stage(0); __syncthreads(); }
  for (long v0 = 0; v0 < vocab; v0 += per_round) {       // block-uniform trip count: every wave reaches every barrier
    const long v = v0 + (long)blockIdx.x * (TPB / 64) + wave;
    for (int d0 = 0; d0 < D; d0 += 512) {
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      bool hit = false;
      for (long base = 0; base < n; base += EMB_DET_CHUNK) {
        if (!single) { __syncthreads(); stage(base); __syncthreads(); }
        const int m = (int)min((long)EMB_DET_CHUNK, n - base);
        if (v < vocab) {
          for (int i = 0; i < m; i += 64) {
            const int id = (i + lane < m) ? sid[i + lane] : -1;
            unsigned long long match = __ballot(id == (int)v);
            while (match) {
              const int j = __ffsll((long long)match) - 1;
              match &= match - 1;
              const long row = base + i + j;
              const int t = (int)(row / B), b = (int)(row % B);
              const float* d = dy + row * D;
#pragma unroll
              for (int k = 0; k < 8; ++k) {
                const int c = d0 + k * 64 + lane;
                if (c < D) acc[k] += d[c] * scale * keep1(dk, t, b, c);
              }
              hit = true;
            }
          }
        }
      }
      if (hit) {
        float* g = denc + v * D;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int c = d0 + k * 64 + lane;
          if (c < D) g[c] += acc[k];
        }
      }
    }
  }
}

// ------------------------------------------------------------------ dropout
__global__ __launch_bounds__(TPB) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, long rows,
                                                      DropKey dk) {
  const int B = dk.B, D = dk.D;
  if ((D & 3) == 0) {
    const long d4 = D >> 2, total = rows * B * d4;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
      const long rb = i / d4;
      const int j = (int)(i - rb * d4) << 2;
      const float4 kp = keep4(dk, (int)(rb / B), (int)(rb % B), j);
      float4 v = *reinterpret_cast<const float4*>(x + rb * D + j);
      v.x *= kp.x; v.y *= kp.y; v.z *= kp.z; v.w *= kp.w;
      *reinterpret_cast<float4*>(y + rb * D + j) = v;
    }
  } else {
    const long total = rows * B * D;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
      const long rb = i / D;
      y[i] = x[i] * keep1(dk, (int)(rb / B), (int)(rb % B), (int)(i - rb * D));
    }
  }
}

__global__ __launch_bounds__(TPB) void add_pe_dropout_kernel(const float* __restrict__ x, const float* __restrict__ pe,
                                                             float* __restrict__ y, long rows, DropKey dk) {
  const int B = dk.B, D = dk.D;
  const long total = rows * B * D;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
    const long rb = i / D;
    const int j = (int)(i - rb * D), t = (int)(rb / B);
    y[i] = (x[i] + pe[(long)t * D + j]) * keep1(dk, t, (int)(rb % B), j);
  }
}

// ------------------------------------------------------------------ residual + dropout + LayerNorm
// One wave per row; the row lives in registers (VPT float4 per lane) when D == 256*VPT -- or, RAG, when D is any multiple of 4 up to
// 256*VPT (the lanes past the row carry zeros: 384, 640, train.py's default 200 ...); else the generic kernel re-reads s from s_out.
template <int VPT, bool RAG = false>
__global__ __launch_bounds__(TPB) void add_drop_ln_fwd_reg(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ out,
                                                           float* __restrict__ s_out, float* __restrict__ mean_o,
                                                           float* __restrict__ rstd_o, long M, float eps, DropKey dk) {
  const int lane = threadIdx.x & 63, D = dk.D;
  const long row = (long)blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
  if (row >= M) return;
  const int t = (int)(row / dk.B), b = (int)(row % dk.B);
  float4 s[VPT];
  float sum = 0.f;
#pragma unroll
  for (int v = 0; v < VPT; ++v) {
    const int j = (v * 64 + lane) * 4;
    if (RAG && j >= D) { s[v] = make_float4(0.f, 0.f, 0.f, 0.f); continue; }
    const float4 a = *reinterpret_cast<const float4*>(x + row * D + j);
    const float4 c = *reinterpret_cast<const float4*>(y + row * D + j);
    const float4 kp = keep4(dk, t, b, j);
    s[v] = make_float4(a.x + c.x * kp.x, a.y + c.y * kp.y, a.z + c.z * kp.z, a.w + c.w * kp.w);
    sum += s[v].x + s[v].y + s[v].z + s[v].w;
  }
  const float mean = wave_sum(sum) / D;
  float var = 0.f;
#pragma unroll
  for (int v = 0; v < VPT; ++v) {
    if (RAG && (v * 64 + lane) * 4 >= D) continue;
    const float dx = s[v].x - mean, dy = s[v].y - mean, dz = s[v].z - mean, dw = s[v].w - mean;
    var += dx * dx + dy * dy + dz * dz + dw * dw;
  }
  const float rstd = rsqrtf(wave_sum(var) / D + eps);
  if (lane == 0) {
    if (mean_o) mean_o[row] = mean;
    if (rstd_o) rstd_o[row] = rstd;
  }
#pragma unroll
  for (int v = 0; v < VPT; ++v) {
    const int j = (v * 64 + lane) * 4;
    if (RAG && j >= D) continue;
    const float4 g = *reinterpret_cast<const float4*>(gamma + j);
    const float4 be = *reinterpret_cast<const float4*>(beta + j);
    if (s_out) *reinterpret_cast<float4*>(s_out + row * D + j) = s[v];
    float4 o;
    o.x = (s[v].x - mean) * rstd * g.x + be.x; o.y = (s[v].y - mean) * rstd * g.y + be.y;
    o.z = (s[v].z - mean) * rstd * g.z + be.z; o.w = (s[v].w - mean) * rstd * g.w + be.w;
    *reinterpret_cast<float4*>(out + row * D + j) = o;
  }
}

__global__ __launch_bounds__(TPB) void add_drop_ln_fwd_generic(const float* __restrict__ x, const float* __restrict__ y,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* __restrict__ out,
                                                               float* __restrict__ s_buf, float* __restrict__ mean_o,
                                                               float* __restrict__ rstd_o, long M, float eps,
                                                               DropKey dk) {
  const int lane = threadIdx.x & 63, D = dk.D;
  const long row = (long)blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
  if (row >= M) return;
  const int t = (int)(row / dk.B), b = (int)(row % dk.B);
  float* s = s_buf + row * D;  // s_out if given, else `out` doubles as scratch
  float sum = 0.f;
  for (int j = lane; j < D; j += 64) {
    const float v = x[row * D + j] + y[row * D + j] * keep1(dk, t, b, j);
    s[j] = v;
    sum += v;
  }
  const float mean = wave_sum(sum) / D;
  float var = 0.f;
  for (int j = lane; j < D; j += 64) {
    const float d = s[j] - mean;
    var += d * d;
  }
  const float rstd = rsqrtf(wave_sum(var) / D + eps);
  if (lane == 0) {
    if (mean_o) mean_o[row] = mean;
    if (rstd_o) rstd_o[row] = rstd;
  }
  for (int j = lane; j < D; j += 64) out[row * D + j] = (s[j] - mean) * rstd * gamma[j] + beta[j];
}

constexpr int LN_BWD_BLOCKS = 256;

// Each wave walks rows row = blockIdx*4 + wave + k*gridDim*4; per-lane partial dgamma/dbeta for its
// columns stay in registers, then waves combine through LDS and the block writes one partial row.
template <int VPT, bool RAG = false>
__global__ __launch_bounds__(TPB) void add_drop_ln_bwd_reg(const float* __restrict__ dout, const float* __restrict__ s,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ mean_i,
                                                           const float* __restrict__ rstd_i, float* __restrict__ dx,
                                                           float* __restrict__ dy, float* __restrict__ ws, long M,
                                                           DropKey dk) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // [4 waves][2][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, D = dk.D;
  float4 pg[VPT], pb[VPT], g[VPT];
#pragma unroll
  for (int v = 0; v < VPT; ++v) {
    pg[v] = pb[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    g[v] = (RAG && (v * 64 + lane) * 4 >= D) ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(gamma + (v * 64 + lane) * 4);
  }
  for (long row = (long)blockIdx.x * 4 + wave; row < M; row += (long)gridDim.x * 4) {
    const int t = (int)(row / dk.B), b = (int)(row % dk.B);
    const float mean = mean_i[row], rstd = rstd_i[row];
    float4 xh[VPT], dxh[VPT];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
      const int j = (v * 64 + lane) * 4;
      if (RAG && j >= D) { xh[v] = dxh[v] = make_float4(0.f, 0.f, 0.f, 0.f); continue; }
      const float4 d = *reinterpret_cast<const float4*>(dout + row * D + j);
      const float4 sv = *reinterpret_cast<const float4*>(s + row * D + j);
      xh[v] = make_float4((sv.x - mean) * rstd, (sv.y - mean) * rstd, (sv.z - mean) * rstd, (sv.w - mean) * rstd);
      dxh[v] = make_float4(d.x * g[v].x, d.y * g[v].y, d.z * g[v].z, d.w * g[v].w);
      pg[v].x += d.x * xh[v].x; pg[v].y += d.y * xh[v].y; pg[v].z += d.z * xh[v].z; pg[v].w += d.w * xh[v].w;
      pb[v].x += d.x; pb[v].y += d.y; pb[v].z += d.z; pb[v].w += d.w;
      s1 += dxh[v].x + dxh[v].y + dxh[v].z + dxh[v].w;
      s2 += dxh[v].x * xh[v].x + dxh[v].y * xh[v].y + dxh[v].z * xh[v].z + dxh[v].w * xh[v].w;
    }
    const float m1 = wave_sum(s1) / D, m2 = wave_sum(s2) / D;
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
      const int j = (v * 64 + lane) * 4;
      if (RAG && j >= D) continue;
      float4 ds;
      ds.x = rstd * (dxh[v].x - m1 - xh[v].x * m2); ds.y = rstd * (dxh[v].y - m1 - xh[v].y * m2);
      ds.z = rstd * (dxh[v].z - m1 - xh[v].z * m2); ds.w = rstd * (dxh[v].w - m1 - xh[v].w * m2);
      *reinterpret_cast<float4*>(dx + row * D + j) = ds;
      if (dy) {
        const float4 kp = keep4(dk, t, b, j);
        ds.x *= kp.x; ds.y *= kp.y; ds.z *= kp.z; ds.w *= kp.w;
        *reinterpret_cast<float4*>(dy + row * D + j) = ds;
      }
    }
  }
#pragma unroll
  for (int v = 0; v < VPT; ++v) {
    const int j = (v * 64 + lane) * 4;
    if (RAG && j >= D) continue;
    *reinterpret_cast<float4*>(sm + (wave * 2 + 0) * D + j) = pg[v];
    *reinterpret_cast<float4*>(sm + (wave * 2 + 1) * D + j) = pb[v];
  }
  __syncthreads();
  for (int j = threadIdx.x; j < 2 * D; j += TPB) {
    const int which = j / D, c = j - which * D;
    ws[((long)blockIdx.x * 2 + which) * D + c] =
        sm[(0 * 2 + which) * D + c] + sm[(1 * 2 + which) * D + c] + sm[(2 * 2 + which) * D + c] + sm[(3 * 2 + which) * D + c];
  }
}

__global__ __launch_bounds__(TPB) void add_drop_ln_bwd_generic(const float* __restrict__ dout, const float* __restrict__ s,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ mean_i,
                                                               const float* __restrict__ rstd_i, float* __restrict__ dx,
                                                               float* __restrict__ dy, float* __restrict__ ws, long M,
                                                               DropKey dk) {
  // one wave per block-row walk; partial dgamma/dbeta accumulated straight into this WAVE's pair of ws rows (zeroed by the
  // launcher): a lane owns its columns there, so no atomics and a fixed order of additions (the finish kernel adds the rows)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, D = dk.D;
  float* wg = ws + (((long)blockIdx.x * 4 + wave) * 2 + 0) * D;
  float* wb = ws + (((long)blockIdx.x * 4 + wave) * 2 + 1) * D;
  for (long row = (long)blockIdx.x * 4 + wave; row < M; row += (long)gridDim.x * 4) {
    const int t = (int)(row / dk.B), b = (int)(row % dk.B);
    const float mean = mean_i[row], rstd = rstd_i[row];
    float s1 = 0.f, s2 = 0.f;
    for (int j = lane; j < D; j += 64) {
      const float d = dout[row * D + j], xh = (s[row * D + j] - mean) * rstd, dxh = d * gamma[j];
      s1 += dxh;
      s2 += dxh * xh;
      wg[j] += d * xh;
      wb[j] += d;
    }
    const float m1 = wave_sum(s1) / D, m2 = wave_sum(s2) / D;
    for (int j = lane; j < D; j += 64) {
      const float d = dout[row * D + j], xh = (s[row * D + j] - mean) * rstd;
      const float ds = rstd * (d * gamma[j] - m1 - xh * m2);
      dx[row * D + j] = ds;
      if (dy) dy[row * D + j] = ds * keep1(dk, t, b, j);
    }
  }
}

// block = 64 columns x 4 partial-row groups; partial rows are summed 4-way in parallel, then via LDS
__global__ __launch_bounds__(TPB) void ln_bwd_finish(const float* __restrict__ ws, int nblk, int D,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float sm[2][4][64];
  const int c = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + c;
  float a = 0.f, b = 0.f;
  if (j < D) {
    int k = grp;
    for (; k + 28 < nblk; k += 32) {  // 8 partial rows (16 independent loads) in flight per thread
      float ta[8], tb[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        ta[u] = ws[((long)(k + 4 * u) * 2 + 0) * D + j];
        tb[u] = ws[((long)(k + 4 * u) * 2 + 1) * D + j];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { a += ta[u]; b += tb[u]; }
    }
    for (; k < nblk; k += 4) {
      a += ws[((long)k * 2 + 0) * D + j];
      b += ws[((long)k * 2 + 1) * D + j];
    }
  }
  sm[0][grp][c] = a;
  sm[1][grp][c] = b;
  __syncthreads();
  if (grp == 0 && j < D) {
    dgamma[j] += sm[0][0][c] + sm[0][1][c] + sm[0][2][c] + sm[0][3][c];
    dbeta[j] += sm[1][0][c] + sm[1][1][c] + sm[1][2][c] + sm[1][3][c];
  }
}

// ------------------------------------------------------------------ cross entropy
// One block per row: online (max, sumexp) pass, then the gradient pass (row is L2-resident: V*4 B).
__global__ __launch_bounds__(TPB) void ce_kernel(const float* __restrict__ logits, long ld, const int64_t* __restrict__ tgt,
                                                 float* __restrict__ nll, float* __restrict__ lse_out,
                                                 float* __restrict__ dlogits, float gscale, int V) {
  __shared__ float red[TPB / 64];
  const long row = blockIdx.x;
  const float* x = logits + row * ld;
  float m = -INFINITY, l = 0.f;
  const bool vec = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(logits) & 15) == 0);
  const int V4 = vec ? (V & ~3) : 0;
  for (int j = threadIdx.x * 4; j < V4; j += TPB * 4) {
    const float4 v = *reinterpret_cast<const float4*>(x + j);
    const float mx = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
    if (mx > m) { l *= __expf(m - mx); m = mx; }
    l += __expf(v.x - m) + __expf(v.y - m) + __expf(v.z - m) + __expf(v.w - m);
  }
  for (int j = V4 + threadIdx.x; j < V; j += TPB) {
    const float v = x[j];
    if (v > m) { l *= __expf(m - v); m = v; }
    l += __expf(v - m);
  }
  const long t = tgt[row];
  const bool valid = t >= 0 && t < V;
  const float xt = valid ? x[t] : 0.f;  // read before any in-place gradient write (block_max has barriers)
  const float M_ = block_max<TPB / 64>(m, red);
  const float L_ = block_sum<TPB / 64>(m == -INFINITY ? 0.f : l * __expf(m - M_), red);
  const float lse = M_ + __logf(L_);
  if (threadIdx.x == 0) {
    nll[row] = valid ? lse - xt : 0.f;
    if (lse_out) lse_out[row] = lse;
  }
  if (dlogits) {
    if (!valid) gscale = 0.f;  // a row without a target in [0, V) (torch's ignore_index = -100, padding ids) has no loss and no gradient
    float* d = dlogits + row * ld;
    for (int j = threadIdx.x * 4; j < V4; j += TPB * 4) {
      float4 v = *reinterpret_cast<const float4*>(x + j);
      v.x = __expf(v.x - lse) * gscale; v.y = __expf(v.y - lse) * gscale;
      v.z = __expf(v.z - lse) * gscale; v.w = __expf(v.w - lse) * gscale;
      if (valid && t >= j && t < j + 4) {
        if (t == j) v.x -= gscale; else if (t == j + 1) v.y -= gscale; else if (t == j + 2) v.z -= gscale; else v.w -= gscale;
      }
      *reinterpret_cast<float4*>(d + j) = v;
    }
    for (int j = V4 + threadIdx.x; j < V; j += TPB) d[j] = (__expf(x[j] - lse) - ((valid && j == t) ? 1.f : 0.f)) * gscale;
  }
}

// Same result with the whole row held in registers: 1024 threads x NV float4 cover V <= 4096 NV logits, so
// the row is read from HBM exactly once (the two-pass kernel above re-reads it for the gradient, and at
// V = 33000 the 2048 rows in flight no longer fit L2/MALL: 3 passes over 1.08 GB instead of 2), and all
// NV loads of a thread are in flight together.
template <int NV>
__global__ __launch_bounds__(1024) void ce_row_kernel(const float* __restrict__ logits, long ld, const int64_t* __restrict__ tgt,
                                                      float* __restrict__ nll, float* __restrict__ lse_out,
                                                      float* __restrict__ dlogits, float gscale, int V) {
  __shared__ float red[16];
  const long row = blockIdx.x;
  const float* x = logits + row * ld;
  const int t0 = threadIdx.x * 4;
  float4 v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int j = t0 + 4096 * i;
    if (j + 3 < V) v[i] = *reinterpret_cast<const float4*>(x + j);
    else {
      v[i].x = j < V ? x[j] : -INFINITY;
      v[i].y = j + 1 < V ? x[j + 1] : -INFINITY;
      v[i].z = j + 2 < V ? x[j + 2] : -INFINITY;
      v[i].w = -INFINITY;
    }
  }
  const long t = tgt[row];
  const bool valid = t >= 0 && t < V;
  const float xt = valid ? x[t] : 0.f;
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < NV; ++i) m = fmaxf(m, fmaxf(fmaxf(v[i].x, v[i].y), fmaxf(v[i].z, v[i].w)));
  const float M_ = block_max<16>(m, red);
  float l = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) l += __expf(v[i].x - M_) + __expf(v[i].y - M_) + __expf(v[i].z - M_) + __expf(v[i].w - M_);
  const float L_ = block_sum<16>(l, red);
  const float lse = M_ + __logf(L_);
  if (threadIdx.x == 0) {
    nll[row] = valid ? lse - xt : 0.f;
    if (lse_out) lse_out[row] = lse;
  }
  if (dlogits) {
    if (!valid) gscale = 0.f;  // no target in [0, V): no loss, no gradient (see ce_kernel)
    float* d = dlogits + row * ld;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int j = t0 + 4096 * i;
      float4 g = make_float4(__expf(v[i].x - lse) * gscale, __expf(v[i].y - lse) * gscale, __expf(v[i].z - lse) * gscale,
                             __expf(v[i].w - lse) * gscale);
      if (valid && t >= j && t < j + 4) {
        if (t == j) g.x -= gscale; else if (t == j + 1) g.y -= gscale; else if (t == j + 2) g.z -= gscale; else g.w -= gscale;
      }
      if (j + 3 < V) *reinterpret_cast<float4*>(d + j) = g;
      else {
        if (j < V) d[j] = g.x;
        if (j + 1 < V) d[j + 1] = g.y;
        if (j + 2 < V) d[j + 2] = g.z;
      }
    }
  }
}

// Two-model scoring (compute_sentence_scores_bayes_jianwei.py:157-168): NLL of the INTERPOLATED LOGITS
// z = alpha * a + (1 - alpha) * b, one pass over both logit rows, z never stored.
__global__ __launch_bounds__(TPB) void ce_interp_kernel(const float* __restrict__ la, const float* __restrict__ lb, long ld,
                                                        float alpha, const int64_t* __restrict__ tgt,
                                                        float* __restrict__ nll, int V) {
  __shared__ float red[TPB / 64];
  const long row = blockIdx.x;
  const float* a = la + row * ld;
  const float* b = lb + row * ld;
  const float beta = 1.f - alpha;
  float m = -INFINITY, l = 0.f;
  const bool vec = ((ld & 3) == 0) && (((reinterpret_cast<uintptr_t>(la) | reinterpret_cast<uintptr_t>(lb)) & 15) == 0);
  const int V4 = vec ? (V & ~3) : 0;
  for (int j = threadIdx.x * 4; j < V4; j += TPB * 4) {
    const float4 u = *reinterpret_cast<const float4*>(a + j), w = *reinterpret_cast<const float4*>(b + j);
    const float4 v = make_float4(alpha * u.x + beta * w.x, alpha * u.y + beta * w.y, alpha * u.z + beta * w.z, alpha * u.w + beta * w.w);
    const float mx = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
    if (mx > m) { l *= __expf(m - mx); m = mx; }
    l += __expf(v.x - m) + __expf(v.y - m) + __expf(v.z - m) + __expf(v.w - m);
  }
  for (int j = V4 + threadIdx.x; j < V; j += TPB) {
    const float v = alpha * a[j] + beta * b[j];
    if (v > m) { l *= __expf(m - v); m = v; }
    l += __expf(v - m);
  }
  const long t = tgt[row];
  const bool valid = t >= 0 && t < V;
  const float xt = valid ? alpha * a[t] + beta * b[t] : 0.f;
  const float M_ = block_max<TPB / 64>(m, red);
  const float L_ = block_sum<TPB / 64>(m == -INFINITY ? 0.f : l * __expf(m - M_), red);
  if (threadIdx.x == 0) nll[row] = valid ? M_ + __logf(L_) - xt : 0.f;
}

__global__ __launch_bounds__(TPB) void ce_bwd_kernel(const float* __restrict__ logits, long ld,
                                                     const int64_t* __restrict__ tgt, const float* __restrict__ lse_in,
                                                     const float* __restrict__ g_dev, float scale,
                                                     float* __restrict__ dlogits, int V) {
  const long row = blockIdx.x;
  const float* x = logits + row * ld;
  float* d = dlogits + row * ld;
  const long t = tgt[row];
  const bool valid = t >= 0 && t < V;
  const float lse = lse_in[row], gs = valid ? g_dev[0] * scale : 0.f;  // no target in [0, V): no gradient
  for (int j = threadIdx.x; j < V; j += TPB) d[j] = (__expf(x[j] - lse) - ((valid && j == t) ? 1.f : 0.f)) * gs;
}

// deterministic single-block sum: out += sum(x[0..n))
__global__ __launch_bounds__(1024) void sum_kernel(const float* __restrict__ x, long n, float* out) {
  __shared__ float red[16];
  float a = 0.f;
  for (long i = threadIdx.x; i < n; i += 1024) a += x[i];
  const float t = block_sum<16>(a, red);
  if (threadIdx.x == 0) out[0] += t;
}

// ------------------------------------------------------------------ column sums (bias gradients)
// block = 32 column-quads x 8 row lanes; grid.y row chunks; partials combined with float atomics.
__global__ __launch_bounds__(TPB) void colsum_kernel(const float* __restrict__ x, long ld, float* __restrict__ out, int M,
                                                     int N, float* __restrict__ out2 = nullptr) {
  __shared__ float4 sm[8][32];
  const int cq = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int col = (blockIdx.x * 32 + cq) * 4;
  const int rows_per = (M + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col < N) {
    const bool vec = (col + 3 < N) && ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    for (int r = r0 + rl; r < r1; r += 8) {
      const float* p = x + (long)r * ld + col;
      if (vec) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
      } else {
        a.x += p[0];
        if (col + 1 < N) a.y += p[1];
        if (col + 2 < N) a.z += p[2];
        if (col + 3 < N) a.w += p[3];
      }
    }
  }
  sm[rl][cq] = a;
  __syncthreads();
  if (rl == 0 && col < N) {
    for (int k = 1; k < 8; ++k) { a.x += sm[k][cq].x; a.y += sm[k][cq].y; a.z += sm[k][cq].z; a.w += sm[k][cq].w; }
    atomicAdd(out + col, a.x);
    if (col + 1 < N) atomicAdd(out + col + 1, a.y);
    if (col + 2 < N) atomicAdd(out + col + 2, a.z);
    if (col + 3 < N) atomicAdd(out + col + 3, a.w);
    if (out2) {  // the same sums into a second vector (an LSTM layer's b_ih and b_hh receive the same gradient)
      atomicAdd(out2 + col, a.x);
      if (col + 1 < N) atomicAdd(out2 + col + 1, a.y);
      if (col + 2 < N) atomicAdd(out2 + col + 2, a.z);
      if (col + 3 < N) atomicAdd(out2 + col + 3, a.w);
    }
  }
}

// GPNN coefficient gradient: 4 column sums of g * act_i(z); block = 64 columns x 4 row lanes
__global__ __launch_bounds__(TPB) void gp_coef_grad_kernel(const float* __restrict__ g, const float* __restrict__ z,
                                                           float* __restrict__ dcoef, int M, int N) {
  __shared__ float sm[4][4][64];
  const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + c;
  const int rows_per = (M + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (col < N) {
    for (int r = r0 + rl; r < r1; r += 4) {
      const float gv = g[(long)r * N + col], zv = z[(long)r * N + col];
      a0 += gv * tanhf(zv);
      a1 += gv * sigmoidf_(zv);
      a2 += gv * fmaxf(zv, 0.f);
      a3 += gv * gelu_erf(zv);
    }
  }
  sm[0][rl][c] = a0; sm[1][rl][c] = a1; sm[2][rl][c] = a2; sm[3][rl][c] = a3;
  __syncthreads();
  if (rl == 0 && col < N) {
#pragma unroll
    for (int i = 0; i < 4; ++i) atomicAdd(dcoef + (long)i * N + col, sm[i][0][c] + sm[i][1][c] + sm[i][2][c] + sm[i][3][c]);
  }
}

// ------------------------------------------------------------------ clip + SGD
// Deterministic (replicas must stay bit-identical after the all-reduce): each block writes its
// partial sum, one block adds the partials in a fixed order.
__global__ __launch_bounds__(TPB) void sqnorm_multi_kernel(const float* const* grads, const int64_t* sizes, float* ws) {
  __shared__ float red[TPB / 64];
  const float* g = grads[blockIdx.y];
  const long n = sizes[blockIdx.y];
  float a = 0.f;
  const long n4 = ((reinterpret_cast<uintptr_t>(g) & 15) == 0) ? (n >> 2) : 0;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n4; i += (long)gridDim.x * TPB) {
    const float4 v = reinterpret_cast<const float4*>(g)[i];
    a += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long)gridDim.x * TPB) a += g[i] * g[i];
  const float t = block_sum<TPB / 64>(a, red);
  if (threadIdx.x == 0) ws[(long)blockIdx.y * gridDim.x + blockIdx.x] = t;
}

// WD: torch.optim.SGD(weight_decay) semantics, g' = c*g + wd*p after the clip (train_search_bayes.py:391-392)
template <bool WD>
__global__ __launch_bounds__(TPB) void clip_sgd_multi_kernel(float* const* params, const float* const* grads,
                                                             float* const* bufs, const int64_t* sizes, const float* sq,
                                                             float clip, float lr, float mom, int first, float gs, float wd) {
  float* p = params[blockIdx.y];
  const float* g = grads[blockIdx.y];
  float* m = bufs[blockIdx.y];
  const long n = sizes[blockIdx.y];
  // gradients are gs * g (gs = 1/world for DP averaging): norm of the scaled gradient
  const float norm = sqrtf(sq[0]) * gs;
  const float c = fminf(1.0f, clip / (norm + 1e-6f)) * gs;
  const bool al = (((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m)) & 15) == 0);
  const long n4 = al ? (n >> 2) : 0;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n4; i += (long)gridDim.x * TPB) {
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 mv = first ? make_float4(0.f, 0.f, 0.f, 0.f) : reinterpret_cast<float4*>(m)[i];
    float4 pv = reinterpret_cast<float4*>(p)[i];
    if (WD) {
      mv.x = mom * mv.x + (c * gv.x + wd * pv.x); mv.y = mom * mv.y + (c * gv.y + wd * pv.y);
      mv.z = mom * mv.z + (c * gv.z + wd * pv.z); mv.w = mom * mv.w + (c * gv.w + wd * pv.w);
    } else {
      mv.x = mom * mv.x + c * gv.x; mv.y = mom * mv.y + c * gv.y; mv.z = mom * mv.z + c * gv.z; mv.w = mom * mv.w + c * gv.w;
    }
    pv.x -= lr * mv.x; pv.y -= lr * mv.y; pv.z -= lr * mv.z; pv.w -= lr * mv.w;
    reinterpret_cast<float4*>(m)[i] = mv;
    reinterpret_cast<float4*>(p)[i] = pv;
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long)gridDim.x * TPB) {
    const float mv = (first ? 0.f : mom * m[i]) + (WD ? c * g[i] + wd * p[i] : c * g[i]);
    m[i] = mv;
    p[i] -= lr * mv;
  }
}

// ------------------------------------------------------------------ LSTM cell
__global__ __launch_bounds__(TPB) void lstm_cell_fwd_kernel(const float* __restrict__ xw, const float* __restrict__ hw,
                                                            const float* __restrict__ c_prev, float* __restrict__ h,
                                                            float* __restrict__ c, float* __restrict__ ga, int B, int H) {
  const long total = (long)B * H;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
    const long b = i / H, j = i - b * H, o = b * 4 * H + j;
    const float gi = sigmoidf_(xw[o] + hw[o]);
    const float gf = sigmoidf_(xw[o + H] + hw[o + H]);
    const float gg = tanhf(xw[o + 2 * H] + hw[o + 2 * H]);
    const float go = sigmoidf_(xw[o + 3 * H] + hw[o + 3 * H]);
    const float cn = gf * c_prev[i] + gi * gg;
    c[i] = cn;
    h[i] = go * tanhf(cn);
    if (ga) { ga[o] = gi; ga[o + H] = gf; ga[o + 2 * H] = gg; ga[o + 3 * H] = go; }
  }
}

__global__ __launch_bounds__(TPB) void lstm_cell_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ dh2,
                                                            const float* __restrict__ dc_next,
                                                            const float* __restrict__ c_prev, const float* __restrict__ c,
                                                            const float* __restrict__ ga, float* __restrict__ dgates,
                                                            float* __restrict__ dc_prev, int B, int H) {
  const long total = (long)B * H;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
    const long b = i / H, j = i - b * H, o = b * 4 * H + j;
    const float gi = ga[o], gf = ga[o + H], gg = ga[o + 2 * H], go = ga[o + 3 * H];
    const float tc = tanhf(c[i]);
    const float dhv = dh[i] + (dh2 ? dh2[i] : 0.f);
    const float dc = (dc_next ? dc_next[i] : 0.f) + dhv * go * (1.f - tc * tc);
    dgates[o] = dc * gg * gi * (1.f - gi);
    dgates[o + H] = dc * c_prev[i] * gf * (1.f - gf);
    dgates[o + 2 * H] = dc * gi * (1.f - gg * gg);
    dgates[o + 3 * H] = dhv * tc * go * (1.f - go);
    dc_prev[i] = dc * gf;
  }
}

__global__ __launch_bounds__(TPB) void lstm_cell_ovr_fwd_kernel(const float* __restrict__ xw, const float* __restrict__ hw,
                                                                const float* __restrict__ c_prev,
                                                                const float* __restrict__ ovr, int gidx,
                                                                float* __restrict__ h, float* __restrict__ c,
                                                                float* __restrict__ ga, int B, int H) {
  const long total = (long)B * H;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
    const long b = i / H, j = i - b * H, o = b * 4 * H + j;
    const float ov = ovr[i];
    const float gi = gidx == 0 ? ov : sigmoidf_(xw[o] + hw[o]);
    const float gf = gidx == 1 ? ov : sigmoidf_(xw[o + H] + hw[o + H]);
    const float gg = gidx == 2 ? ov : tanhf(xw[o + 2 * H] + hw[o + 2 * H]);
    const float go = gidx == 3 ? ov : sigmoidf_(xw[o + 3 * H] + hw[o + 3 * H]);
    const float cn = gf * c_prev[i] + gi * gg;
    c[i] = cn;
    h[i] = go * tanhf(cn);
    if (ga) { ga[o] = gi; ga[o + H] = gf; ga[o + 2 * H] = gg; ga[o + 3 * H] = go; }
  }
}

__global__ __launch_bounds__(TPB) void lstm_cell_ovr_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ dh2, const float* __restrict__ dc_next,
                                                                const float* __restrict__ c_prev, const float* __restrict__ c,
                                                                const float* __restrict__ ga, int gidx,
                                                                float* __restrict__ dgates, float* __restrict__ d_ovr,
                                                                float* __restrict__ dc_prev, int B, int H) {
  const long total = (long)B * H;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
    const long b = i / H, j = i - b * H, o = b * 4 * H + j;
    const float gi = ga[o], gf = ga[o + H], gg = ga[o + 2 * H], go = ga[o + 3 * H];
    const float tc = tanhf(c[i]);
    const float dhv = dh[i] + (dh2 ? dh2[i] : 0.f);
    const float dc = (dc_next ? dc_next[i] : 0.f) + dhv * go * (1.f - tc * tc);
    const float dgi = dc * gg, dgf = dc * c_prev[i], dgg = dc * gi, dgo = dhv * tc;  // w.r.t. activated gates
    dgates[o] = gidx == 0 ? 0.f : dgi * gi * (1.f - gi);
    dgates[o + H] = gidx == 1 ? 0.f : dgf * gf * (1.f - gf);
    dgates[o + 2 * H] = gidx == 2 ? 0.f : dgg * (1.f - gg * gg);
    dgates[o + 3 * H] = gidx == 3 ? 0.f : dgo * go * (1.f - go);
    d_ovr[i] = gidx == 0 ? dgi : (gidx == 1 ? dgf : (gidx == 2 ? dgg : dgo));
    dc_prev[i] = dc * gf;
  }
}

__global__ __launch_bounds__(TPB) void gp_mix_fwd_kernel(const float* __restrict__ z, const float* __restrict__ coef,
                                                         float* __restrict__ out, long total, int N) {
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
    const int n = (int)(i % N);
    const float v = z[i];
    out[i] = tanhf(v) * coef[n] + sigmoidf_(v) * coef[N + n] + fmaxf(v, 0.f) * coef[2 * N + n] + gelu_erf(v) * coef[3 * N + n];
  }
}
__global__ __launch_bounds__(TPB) void gp_mix_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ z,
                                                         const float* __restrict__ coef, float* __restrict__ dz, long total,
                                                         int N) {
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
    const int n = (int)(i % N);
    const float v = z[i], th = tanhf(v), sg = sigmoidf_(v);
    dz[i] = dout[i] * ((1.f - th * th) * coef[n] + sg * (1.f - sg) * coef[N + n] + (v > 0.f ? coef[2 * N + n] : 0.f) +
                       dgelu_erf(v) * coef[3 * N + n]);
  }
}
__global__ __launch_bounds__(TPB) void add_rowvec_kernel(float* __restrict__ x, const float* __restrict__ v, long total, int H) {
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) x[i] += v[i % H];
}

__global__ __launch_bounds__(TPB) void axpy_kernel(const float* __restrict__ x, float* __restrict__ y, long n, float a) {
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long)gridDim.x * TPB) y[i] += a * x[i];
}

// GPNN2 random-feature head (model.py:2068-2076): s = (f + sum_{a in acts} a(f)) * scale for the M real feature columns,
// s[:, M] = 1 (the column that carries coef.bias when the coefficient matrix is padded with it), zeros beyond; `acts` is a
// bit set in the mixture's slot order (1 tanh, 2 sigmoid, 4 relu, 8 gelu).  ld_f / ld_s: row strides of f and s.
__global__ __launch_bounds__(TPB) void gpnn2_actsum_fwd_kernel(const float* __restrict__ f, float* __restrict__ s, long rows, int M,
                                                               int ld_f, int ld_s, float scale, int acts) {
  const long total = rows * ld_s;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
    const long r = i / ld_s;
    const int m = (int)(i - r * ld_s);
    float v = m == M ? 1.f : 0.f;
    if (m < M) {
      const float z = f[r * ld_f + m];
      v = z;
      if (acts & 1) v += tanhf(z);
      if (acts & 2) v += sigmoidf_(z);
      if (acts & 4) v += fmaxf(z, 0.f);
      if (acts & 8) v += gelu_erf(z);
      v *= scale;
    }
    s[i] = v;
  }
}
// df = ds * (1 + sum a'(f)) * scale on the real columns, 0 on the padding (df has row stride ld_s, like ds)
__global__ __launch_bounds__(TPB) void gpnn2_actsum_bwd_kernel(const float* __restrict__ ds, const float* __restrict__ f,
                                                               float* __restrict__ df, long rows, int M, int ld_f, int ld_s,
                                                               float scale, int acts) {
  const long total = rows * ld_s;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
    const long r = i / ld_s;
    const int m = (int)(i - r * ld_s);
    float v = 0.f;
    if (m < M) {
      const float z = f[r * ld_f + m];
      float d = 1.f;
      if (acts & 1) { const float th = tanhf(z); d += 1.f - th * th; }
      if (acts & 2) { const float sg = sigmoidf_(z); d += sg * (1.f - sg); }
      if (acts & 4) d += z > 0.f ? 1.f : 0.f;
      if (acts & 8) d += dgelu_erf(z);
      v = ds[i] * d * scale;
    }
    df[i] = v;
  }
}
// out[r, c] = a[r, c] + b[r, c] on column windows of wider matrices (row strides lda / ldb / ldo)
__global__ __launch_bounds__(TPB) void add_cols_kernel(const float* __restrict__ a, long lda, const float* __restrict__ b, long ldb,
                                                       float* __restrict__ out, long ldo, long rows, int cols) {
  const long total = rows * cols;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
    const long r = i / cols;
    const int c = (int)(i - r * cols);
    out[r * ldo + c] = a[r * lda + c] + b[r * ldb + c];
  }
}

static int grid_for(long items) {
  long g = (items + TPB - 1) / TPB;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace blm

using namespace blm;
#define ST static_cast<hipStream_t>(stream)

// dst[v,:] += src[slot[v],:] for the vocabulary rows v that took part in this step (slot[v] >= 0): the late,
// compact half of the tied embedding gradient joins the flat gradient buffer after its all-reduce.
__global__ __launch_bounds__(TPB) void rows_gather_add_kernel(float* __restrict__ dst, const int64_t* __restrict__ slot,
                                                              const float* __restrict__ src, long V, int D, long n_src) {
  const int lane = threadIdx.x & 63;
  const long v = (long)blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
  if (v >= V) return;
  const long s = slot[v];
  if (s < 0 || s >= n_src) return;
  float* d = dst + v * D;
  const float* r = src + s * D;
  if ((D & 3) == 0 && ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) == 0) {
    for (int j = lane * 4; j < D; j += 256) {
      float4 a = *reinterpret_cast<const float4*>(d + j);
      const float4 b = *reinterpret_cast<const float4*>(r + j);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
      *reinterpret_cast<float4*>(d + j) = a;
    }
  } else {
    for (int j = lane; j < D; j += 64) d[j] += r[j];
  }
}

extern "C" int blm_embed_fwd(const int64_t* ids, const float* enc, const float* pe, float* out, int T, int B, int D,
                             int64_t vocab, float scale, float p, const blm_rng* rng, int col_offset, int global_cols,
                             void* stream) {
  if (!ids || !enc || !out || T < 0 || B < 0 || D <= 0 || vocab <= 0) return blm_fail(BLM_ERR_INVALID, "blm_embed_fwd: bad arguments");
  if (p > 0.f && !rng) return blm_fail(BLM_ERR_INVALID, "blm_embed_fwd: dropout needs rng");
  if ((long)T * B == 0) return BLM_OK;
  const long rows = (long)T * B;
  hipLaunchKernelGGL(embed_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(TPB), 0, ST, ids, enc, pe, out, T, B, D,
                     (long)vocab, scale, make_key(p, rng, B, D, col_offset, global_cols));
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int blm_embed_bwd(const int64_t* ids, const float* dy, float* denc, int T, int B, int D, int64_t vocab,
                             float scale, float p, const blm_rng* rng, int col_offset, int global_cols, void* stream) {
  if (!ids || !dy || !denc || T < 0 || B < 0 || D <= 0 || vocab <= 0) return blm_fail(BLM_ERR_INVALID, "blm_embed_bwd: bad arguments");
  if (p > 0.f && !rng) return blm_fail(BLM_ERR_INVALID, "blm_embed_bwd: dropout needs rng");
  if ((long)T * B == 0) return BLM_OK;
  const long rows = (long)T * B;
  if (blm::option(blm::OPT_DETERMINISTIC) && vocab <= 0x7fffffffL) {  // one wave per vocabulary row, additions in position order (ids staged as 32-bit)
    long g = (vocab + 3) / 4;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(embed_bwd_det_kernel, dim3((unsigned)g), dim3(TPB), 0, ST, ids, dy, denc, T, B, D, (long)vocab, scale,
                       make_key(p, rng, B, D, col_offset, global_cols));
    BLM_HIP(hipGetLastError());
    return BLM_OK;
  }
  hipLaunchKernelGGL(embed_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(TPB), 0, ST, ids, dy, denc, T, B, D,
                     (long)vocab, scale, make_key(p, rng, B, D, col_offset, global_cols));
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int blm_dropout(const float* x, float* y, int rows, int B, int D, float p, const blm_rng* rng, int col_offset,
                           int global_cols, void* stream) {
  if (!x || !y || !blm::extents_ok({rows, B, D})) return blm_fail(BLM_ERR_INVALID, "blm_dropout: bad arguments");
  if (p > 0.f && !rng) return blm_fail(BLM_ERR_INVALID, "blm_dropout: dropout needs rng");
  const long n = (long)rows * B * D;
  if (n == 0) return BLM_OK;
  hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(n / 4 + 1)), dim3(TPB), 0, ST, x, y, (long)rows,
                     make_key(p, rng, B, D, col_offset, global_cols));
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int blm_dropout_rows(const float* x, float* y, int rows, int row0, int B, int D, float p, const blm_rng* rng,
                                int col_offset, int global_cols, void* stream) {
  if (!x || !y || row0 < 0 || !blm::extents_ok({rows, B, D}) || !blm::extents_ok({row0, B, D})) return blm_fail(BLM_ERR_INVALID, "blm_dropout_rows: bad arguments");
  if (p > 0.f && !rng) return blm_fail(BLM_ERR_INVALID, "blm_dropout_rows: dropout needs rng");
  const long n = (long)rows * B * D;
  if (n == 0) return BLM_OK;
  DropKey dk = make_key(p, rng, B, D, col_offset, global_cols);
  dk.row0 = row0;
  hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(n / 4 + 1)), dim3(TPB), 0, ST, x, y, (long)rows, dk);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int blm_add_pe_dropout(const float* x, const float* pe, float* out, int T, int B, int D, float p,
                                  const blm_rng* rng, int col_offset, int global_cols, void* stream) {
  if (!x || !pe || !out || !blm::extents_ok({T, B, D})) return blm_fail(BLM_ERR_INVALID, "blm_add_pe_dropout: bad arguments");
  if (p > 0.f && !rng) return blm_fail(BLM_ERR_INVALID, "blm_add_pe_dropout: dropout needs rng");
  const long n = (long)T * B * D;
  if (n == 0) return BLM_OK;
  hipLaunchKernelGGL(add_pe_dropout_kernel, dim3(grid_for(n)), dim3(TPB), 0, ST, x, pe, out, (long)T,
                     make_key(p, rng, B, D, col_offset, global_cols));
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int blm_add_dropout_ln_fwd(const float* x, const float* y, const float* gamma, const float* beta, float* out,
                                      float* s_out, float* mean, float* rstd, int rows, int B, int D, float eps_ln,
                                      float p, const blm_rng* rng, int col_offset, int global_cols, void* stream) {
  if (!x || !y || !gamma || !beta || !out || rows < 0 || B < 0 || D <= 0) return blm_fail(BLM_ERR_INVALID, "blm_add_dropout_ln_fwd: bad arguments");
  if (p > 0.f && !rng) return blm_fail(BLM_ERR_INVALID, "blm_add_dropout_ln_fwd: dropout needs rng");
  const long M = (long)rows * B;
  if (M == 0) return BLM_OK;
  const DropKey dk = make_key(p, rng, B, D, col_offset, global_cols);
  const dim3 grid((unsigned)((M + 3) / 4));
  const bool al = (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(out) |
                     reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta) | reinterpret_cast<uintptr_t>(s_out)) & 15) == 0);
#define LN_FWD(V) hipLaunchKernelGGL(add_drop_ln_fwd_reg<V>, grid, dim3(TPB), 0, ST, x, y, gamma, beta, out, s_out, mean, rstd, M, eps_ln, dk)
  if (al && D == 256) LN_FWD(1);
  else if (al && D == 512) LN_FWD(2);
  else if (al && D == 768) LN_FWD(3);   // the widths between the powers of two (768, 1280, 1536) used to take the generic kernel:
  else if (al && D == 1024) LN_FWD(4);  // a d_model 768 model spent a sixth of its step there (tools/shape_cliff_probe.py)
  else if (al && D == 1280) LN_FWD(5);
  else if (al && D == 1536) LN_FWD(6);
  else if (al && D == 2048) LN_FWD(8);
#define LN_FWD_RAG(V) hipLaunchKernelGGL((add_drop_ln_fwd_reg<V, true>), grid, dim3(TPB), 0, ST, x, y, gamma, beta, out, s_out, mean, rstd, M, eps_ln, dk)
  else if (al && D % 4 == 0 && D < 256) LN_FWD_RAG(1);    // any other multiple of 4 up to 1024: the same kernels, the tail lanes idle
  else if (al && D % 4 == 0 && D < 512) LN_FWD_RAG(2);
  else if (al && D % 4 == 0 && D < 768) LN_FWD_RAG(3);
  else if (al && D % 4 == 0 && D < 1024) LN_FWD_RAG(4);
#undef LN_FWD_RAG
  else
    hipLaunchKernelGGL(add_drop_ln_fwd_generic, grid, dim3(TPB), 0, ST, x, y, gamma, beta, out, s_out ? s_out : out, mean,
                       rstd, M, eps_ln, dk);
#undef LN_FWD
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int64_t blm_ln_bwd_ws_floats(int M, int D) {
  (void)M;
  return (int64_t)2 * LN_BWD_BLOCKS * D;
}

extern "C" int blm_add_dropout_ln_bwd(const float* dout, const float* s, const float* gamma, const float* mean,
                                      const float* rstd, float* dx, float* dy, float* dgamma, float* dbeta, float* ws,
                                      int rows, int B, int D, float p, const blm_rng* rng, int col_offset,
                                      int global_cols, void* stream) {
  if (!dout || !s || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || !ws || rows < 0 || B < 0 || D <= 0)
    return blm_fail(BLM_ERR_INVALID, "blm_add_dropout_ln_bwd: bad arguments");
  if (p > 0.f && !rng) return blm_fail(BLM_ERR_INVALID, "blm_add_dropout_ln_bwd: dropout needs rng");
  const long M = (long)rows * B;
  if (M == 0) return BLM_OK;
  const DropKey dk = make_key(p, rng, B, D, col_offset, global_cols);
  int nblk = (int)((M + 3) / 4);
  if (nblk > LN_BWD_BLOCKS) nblk = LN_BWD_BLOCKS;
  const bool al = (((reinterpret_cast<uintptr_t>(dout) | reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(dx) |
                     reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0);
  const size_t lds = (size_t)8 * D * sizeof(float);
#define LN_BWD(V) hipLaunchKernelGGL(add_drop_ln_bwd_reg<V>, dim3(nblk), dim3(TPB), lds, ST, dout, s, gamma, mean, rstd, dx, dy, ws, M, dk)
  if (al && D == 256) LN_BWD(1);
  else if (al && D == 512) LN_BWD(2);
  else if (al && D == 768) LN_BWD(3);
  else if (al && D == 1024) LN_BWD(4);
  else if (al && D == 1280) LN_BWD(5);
  else if (al && D == 1536) LN_BWD(6);
  else if (al && D == 2048) LN_BWD(8);
#define LN_BWD_RAG(V) hipLaunchKernelGGL((add_drop_ln_bwd_reg<V, true>), dim3(nblk), dim3(TPB), lds, ST, dout, s, gamma, mean, rstd, dx, dy, ws, M, dk)
  else if (al && D % 4 == 0 && D < 256) LN_BWD_RAG(1);
  else if (al && D % 4 == 0 && D < 512) LN_BWD_RAG(2);
  else if (al && D % 4 == 0 && D < 768) LN_BWD_RAG(3);
  else if (al && D % 4 == 0 && D < 1024) LN_BWD_RAG(4);
#undef LN_BWD_RAG
  else {  // four partial rows per workgroup (one per wave): a quarter of the blocks fills the same workspace
    nblk = (nblk + 3) / 4;
    BLM_HIP(hipMemsetAsync(ws, 0, (size_t)2 * 4 * nblk * D * sizeof(float), ST));
    hipLaunchKernelGGL(add_drop_ln_bwd_generic, dim3(nblk), dim3(TPB), 0, ST, dout, s, gamma, mean, rstd, dx, dy, ws, M, dk);
    nblk *= 4;
  }
#undef LN_BWD
  BLM_HIP(hipGetLastError());
  hipLaunchKernelGGL(ln_bwd_finish, dim3((D + 63) / 64), dim3(TPB), 0, ST, ws, nblk, D, dgamma, dbeta);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int blm_ce_bwd(const float* logits, int64_t ld, const int64_t* tgt, const float* lse, const float* g_dev,
                          float scale, float* dlogits, int M, int V, void* stream) {
  if (!logits || !tgt || !lse || !g_dev || !dlogits || M < 0 || V <= 0 || ld < V) return blm_fail(BLM_ERR_INVALID, "blm_ce_bwd: bad arguments");
  if (M == 0) return BLM_OK;
  hipLaunchKernelGGL(ce_bwd_kernel, dim3(M), dim3(TPB), 0, ST, logits, (long)ld, tgt, lse, g_dev, scale, dlogits, V);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int blm_ce_fwd_bwd(const float* logits, int64_t ld, const int64_t* tgt, float* nll, float* lse, float* loss_sum,
                              float* dlogits, float grad_scale, int M, int V, void* stream) {
  if (!logits || !tgt || !nll || M < 0 || V <= 0 || ld < V) return blm_fail(BLM_ERR_INVALID, "blm_ce_fwd_bwd: bad arguments");
  if (M == 0) return BLM_OK;
  const bool vec = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(logits) & 15) == 0) && (!dlogits || (reinterpret_cast<uintptr_t>(dlogits) & 15) == 0);
  if (vec && V <= 4096 * 3)
    hipLaunchKernelGGL(ce_row_kernel<3>, dim3(M), dim3(1024), 0, ST, logits, (long)ld, tgt, nll, lse, dlogits, grad_scale, V);
  else if (vec && V <= 4096 * 9)
    hipLaunchKernelGGL(ce_row_kernel<9>, dim3(M), dim3(1024), 0, ST, logits, (long)ld, tgt, nll, lse, dlogits, grad_scale, V);
  else if (vec && V <= 4096 * 16)
    hipLaunchKernelGGL(ce_row_kernel<16>, dim3(M), dim3(1024), 0, ST, logits, (long)ld, tgt, nll, lse, dlogits, grad_scale, V);
  else
    hipLaunchKernelGGL(ce_kernel, dim3(M), dim3(TPB), 0, ST, logits, (long)ld, tgt, nll, lse, dlogits, grad_scale, V);
  BLM_HIP(hipGetLastError());
  if (loss_sum) {
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1024), 0, ST, nll, (long)M, loss_sum);
    BLM_HIP(hipGetLastError());
  }
  return BLM_OK;
}

extern "C" int blm_ce_interp_fwd(const float* logits_a, const float* logits_b, int64_t ld, float alpha, const int64_t* tgt,
                                 float* nll, int M, int V, void* stream) {
  if (!logits_a || !logits_b || !tgt || !nll || M < 0 || V <= 0 || ld < V) return blm_fail(BLM_ERR_INVALID, "blm_ce_interp_fwd: bad arguments");
  if (M == 0) return BLM_OK;
  hipLaunchKernelGGL(ce_interp_kernel, dim3(M), dim3(TPB), 0, ST, logits_a, logits_b, (long)ld, alpha, tgt, nll, V);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int blm_gp_coef_grad(const float* g, const float* z, float* dcoef, int M, int N, void* stream) {
  if (!g || !z || !dcoef || !blm::extents_ok({M, N})) return blm_fail(BLM_ERR_INVALID, "blm_gp_coef_grad: bad arguments");
  if (M == 0 || N == 0) return BLM_OK;
  int gy = (M + 127) / 128;
  if (gy > 64) gy = 64;
  if (blm::option(blm::OPT_DETERMINISTIC)) gy = 1;  // one row chunk: each sum has one writer, rows added in a fixed order
  hipLaunchKernelGGL(gp_coef_grad_kernel, dim3((N + 63) / 64, gy), dim3(TPB), 0, ST, g, z, dcoef, M, N);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int blm_colsum2(const float* x, int64_t ld, float* out, float* out2, int M, int N, int accumulate, void* stream) {
  if (!x || !out || !blm::extents_ok({M, N}) || ld < N || !blm::extents_ok({M, ld}) || out == out2) return blm_fail(BLM_ERR_INVALID, "blm_colsum: bad arguments");
  if (N == 0) return BLM_OK;
  if (!accumulate) {
    BLM_HIP(hipMemsetAsync(out, 0, (size_t)N * sizeof(float), ST));
    if (out2) BLM_HIP(hipMemsetAsync(out2, 0, (size_t)N * sizeof(float), ST));
  }
  if (M == 0) return BLM_OK;
  int gy = (M + 255) / 256;
  if (gy > 64) gy = 64;
  if (blm::option(blm::OPT_DETERMINISTIC)) gy = 1;  // one row chunk: each sum has one writer, rows added in a fixed order
  hipLaunchKernelGGL(colsum_kernel, dim3((N + 127) / 128, gy), dim3(TPB), 0, ST, x, (long)ld, out, M, N, out2);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int blm_colsum(const float* x, int64_t ld, float* out, int M, int N, int accumulate, void* stream) {
  return blm_colsum2(x, ld, out, nullptr, M, N, accumulate, stream);
}

// Up to 8 small vectors initialised by ONE launch: dst_i = (src_i or 0) + (src2_i or 0).  The set-up of a recurrent layer is a
// handful of (B,H)- and (4H)-sized copies, zero fills and bias sums -- 5 us each as launches of their own.
struct InitMulti {
  float* dst[8];
  const float* src[8];
  const float* src2[8];
  long n[8];
};
__global__ __launch_bounds__(TPB) void init_multi_kernel(const InitMulti p) {
  const int i = blockIdx.y;
  // no __restrict__: dst may BE src or src2 (in-place accumulation, ops._bias_pair_grads: grad = grad + db); every element is read
  // by the lane that writes it, before it writes it
  float* d = p.dst[i];
  const float* a = p.src[i];
  const float* b = p.src2[i];
  const long n = p.n[i];
  const bool vec = (n & 3) == 0 && ((reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0;
  const long stride = (long)gridDim.x * TPB;
  if (vec) {
    for (long j = (long)blockIdx.x * TPB + threadIdx.x; j < (n >> 2); j += stride) {
      float4 v = a ? reinterpret_cast<const float4*>(a)[j] : make_float4(0.f, 0.f, 0.f, 0.f);
      if (b) {
        const float4 w = reinterpret_cast<const float4*>(b)[j];
        v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
      }
      reinterpret_cast<float4*>(d)[j] = v;
    }
  } else {
    for (long j = (long)blockIdx.x * TPB + threadIdx.x; j < n; j += stride) d[j] = (a ? a[j] : 0.f) + (b ? b[j] : 0.f);
  }
}

extern "C" int blm_init_multi(int count, float* const* dst, const float* const* src, const float* const* src2, const int64_t* n,
                              void* stream) {
  if (count < 0 || count > 8 || (count > 0 && (!dst || !n))) return blm_fail(BLM_ERR_INVALID, "blm_init_multi: bad arguments (at most 8 vectors)");
  InitMulti p{};
  long most = 0;
  int m = 0;
  for (int i = 0; i < count; ++i) {
    if (n[i] < 0 || (n[i] > 0 && !dst[i])) return blm_fail(BLM_ERR_INVALID, "blm_init_multi: bad arguments");
    if (n[i] == 0) continue;
    p.dst[m] = dst[i];
    p.src[m] = src ? src[i] : nullptr;
    p.src2[m] = src2 ? src2[i] : nullptr;
    p.n[m] = (long)n[i];
    most = n[i] > most ? (long)n[i] : most;
    ++m;
  }
  if (m == 0) return BLM_OK;
  long gx = (most / 4 + TPB - 1) / TPB;
  gx = gx < 1 ? 1 : (gx > 1024 ? 1024 : gx);
  hipLaunchKernelGGL(init_multi_kernel, dim3((unsigned)gx, m), dim3(TPB), 0, ST, p);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int64_t blm_sqnorm_ws_floats(int n) { return (int64_t)(n == 1 ? 1024 : 64) * (n > 0 ? n : 1); }

extern "C" int blm_sqnorm_multi(const float* const* grads, const int64_t* sizes, int n, float* sq, float* ws,
                                void* stream) {
  if (!grads || !sizes || !sq || !ws || n < 0) return blm_fail(BLM_ERR_INVALID, "blm_sqnorm_multi: bad arguments");
  if (n == 0) return BLM_OK;
  const int gx = n == 1 ? 1024 : 64;
  hipLaunchKernelGGL(sqnorm_multi_kernel, dim3(gx, n), dim3(TPB), 0, ST, grads, sizes, ws);
  BLM_HIP(hipGetLastError());
  hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1024), 0, ST, ws, (long)gx * n, sq);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int blm_clip_sgd_multi(float* const* params, const float* const* grads, float* const* bufs,
                                  const int64_t* sizes, int n, const float* sq, float clip, float lr, float momentum,
                                  int first, float grad_scale, void* stream) {
  if (!params || !grads || !bufs || !sizes || !sq || n < 0) return blm_fail(BLM_ERR_INVALID, "blm_clip_sgd_multi: bad arguments");
  if (n == 0) return BLM_OK;
  hipLaunchKernelGGL(clip_sgd_multi_kernel<false>, dim3(n == 1 ? 2048 : 64, n), dim3(TPB), 0, ST, params, grads, bufs, sizes,
                     sq, clip, lr, momentum, first, grad_scale, 0.f);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int blm_clip_sgd_multi_wd(float* const* params, const float* const* grads, float* const* bufs,
                                     const int64_t* sizes, int n, const float* sq, float clip, float lr, float momentum,
                                     int first, float grad_scale, float weight_decay, void* stream) {
  if (!params || !grads || !bufs || !sizes || !sq || n < 0) return blm_fail(BLM_ERR_INVALID, "blm_clip_sgd_multi_wd: bad arguments");
  if (n == 0) return BLM_OK;
  hipLaunchKernelGGL(clip_sgd_multi_kernel<true>, dim3(n == 1 ? 2048 : 64, n), dim3(TPB), 0, ST, params, grads, bufs, sizes,
                     sq, clip, lr, momentum, first, grad_scale, weight_decay);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int blm_lstm_cell_fwd(const float* xw, const float* hw, const float* c_prev, float* h, float* c,
                                 float* gates_act, int B, int H, void* stream) {
  if (!xw || !hw || !c_prev || !h || !c || B < 0 || H < 0) return blm_fail(BLM_ERR_INVALID, "blm_lstm_cell_fwd: bad arguments");
  if ((long)B * H == 0) return BLM_OK;
  hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3(grid_for((long)B * H)), dim3(TPB), 0, ST, xw, hw, c_prev, h, c, gates_act, B, H);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int blm_lstm_cell_bwd(const float* dh, const float* dc_next, const float* c_prev, const float* c,
                                 const float* gates_act, float* dgates, float* dc_prev, int B, int H, void* stream) {
  return blm_lstm_cell_bwd2(dh, nullptr, dc_next, c_prev, c, gates_act, dgates, dc_prev, B, H, stream);
}

extern "C" int blm_lstm_cell_bwd2(const float* dh, const float* dh2, const float* dc_next, const float* c_prev,
                                  const float* c, const float* gates_act, float* dgates, float* dc_prev, int B, int H,
                                  void* stream) {
  if (!dh || !c_prev || !c || !gates_act || !dgates || !dc_prev || B < 0 || H < 0)
    return blm_fail(BLM_ERR_INVALID, "blm_lstm_cell_bwd: bad arguments");
  if ((long)B * H == 0) return BLM_OK;
  hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(grid_for((long)B * H)), dim3(TPB), 0, ST, dh, dh2, dc_next, c_prev, c, gates_act,
                     dgates, dc_prev, B, H);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int blm_lstm_cell_ovr_fwd(const float* xw, const float* hw, const float* c_prev, const float* gate_ovr, int gate_idx,
                                     float* h, float* c, float* gates_act, int B, int H, void* stream) {
  if (!xw || !hw || !c_prev || !gate_ovr || !h || !c || B < 0 || H < 0 || gate_idx < 0 || gate_idx > 3)
    return blm_fail(BLM_ERR_INVALID, "blm_lstm_cell_ovr_fwd: bad arguments");
  if ((long)B * H == 0) return BLM_OK;
  hipLaunchKernelGGL(lstm_cell_ovr_fwd_kernel, dim3(grid_for((long)B * H)), dim3(TPB), 0, ST, xw, hw, c_prev, gate_ovr, gate_idx,
                     h, c, gates_act, B, H);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int blm_lstm_cell_ovr_bwd(const float* dh, const float* dc_next, const float* c_prev, const float* c,
                                     const float* gates_act, int gate_idx, float* dgates, float* d_ovr, float* dc_prev,
                                     int B, int H, void* stream) {
  if (!dh || !c_prev || !c || !gates_act || !dgates || !d_ovr || !dc_prev || B < 0 || H < 0 || gate_idx < 0 || gate_idx > 3)
    return blm_fail(BLM_ERR_INVALID, "blm_lstm_cell_ovr_bwd: bad arguments");
  if ((long)B * H == 0) return BLM_OK;
  hipLaunchKernelGGL(lstm_cell_ovr_bwd_kernel, dim3(grid_for((long)B * H)), dim3(TPB), 0, ST, dh, (const float*)nullptr, dc_next, c_prev,
                     c, gates_act, gate_idx, dgates, d_ovr, dc_prev, B, H);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int blm_lstm_cell_ovr_bwd2(const float* dh, const float* dh2, const float* dc_next, const float* c_prev, const float* c,
                                      const float* gates_act, int gate_idx, float* dgates, float* d_ovr, float* dc_prev, int B,
                                      int H, void* stream) {
  if (!dh || !c_prev || !c || !gates_act || !dgates || !d_ovr || !dc_prev || B < 0 || H < 0 || gate_idx < 0 || gate_idx > 3)
    return blm_fail(BLM_ERR_INVALID, "blm_lstm_cell_ovr_bwd2: bad arguments");
  if ((long)B * H == 0) return BLM_OK;
  hipLaunchKernelGGL(lstm_cell_ovr_bwd_kernel, dim3(grid_for((long)B * H)), dim3(TPB), 0, ST, dh, dh2, dc_next, c_prev, c, gates_act,
                     gate_idx, dgates, d_ovr, dc_prev, B, H);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int blm_gp_mix_fwd(const float* z, const float* coef, float* out, int M, int N, void* stream) {
  if (!z || !coef || !out || M < 0 || N < 0) return blm_fail(BLM_ERR_INVALID, "blm_gp_mix_fwd: bad arguments");
  if ((long)M * N == 0) return BLM_OK;
  hipLaunchKernelGGL(gp_mix_fwd_kernel, dim3(grid_for((long)M * N)), dim3(TPB), 0, ST, z, coef, out, (long)M * N, N);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int blm_gp_mix_bwd(const float* dout, const float* z, const float* coef, float* dz, int M, int N, void* stream) {
  if (!dout || !z || !coef || !dz || M < 0 || N < 0) return blm_fail(BLM_ERR_INVALID, "blm_gp_mix_bwd: bad arguments");
  if ((long)M * N == 0) return BLM_OK;
  hipLaunchKernelGGL(gp_mix_bwd_kernel, dim3(grid_for((long)M * N)), dim3(TPB), 0, ST, dout, z, coef, dz, (long)M * N, N);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int blm_gpnn2_actsum_fwd(const float* f, float* s, int64_t rows, int M, int ld_f, int ld_s, float scale, int acts,
                                    void* stream) {
  if (!f || !s || rows < 0 || M < 0 || ld_f < M || ld_s < M || (acts & ~15))
    return blm_fail(BLM_ERR_INVALID, "blm_gpnn2_actsum_fwd: bad arguments");
  if (rows * (long)ld_s == 0) return BLM_OK;
  hipLaunchKernelGGL(gpnn2_actsum_fwd_kernel, dim3(grid_for(rows * (long)ld_s)), dim3(TPB), 0, ST, f, s, (long)rows, M, ld_f, ld_s,
                     scale, acts);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int blm_gpnn2_actsum_bwd(const float* ds, const float* f, float* df, int64_t rows, int M, int ld_f, int ld_s, float scale,
                                    int acts, void* stream) {
  if (!ds || !f || !df || rows < 0 || M < 0 || ld_f < M || ld_s < M || (acts & ~15))
    return blm_fail(BLM_ERR_INVALID, "blm_gpnn2_actsum_bwd: bad arguments");
  if (rows * (long)ld_s == 0) return BLM_OK;
  hipLaunchKernelGGL(gpnn2_actsum_bwd_kernel, dim3(grid_for(rows * (long)ld_s)), dim3(TPB), 0, ST, ds, f, df, (long)rows, M, ld_f,
                     ld_s, scale, acts);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int blm_add_cols(const float* a, int64_t lda, const float* b, int64_t ldb, float* out, int64_t ldo, int64_t rows, int cols,
                            void* stream) {
  if (!a || !b || !out || !blm::extents_ok({rows, cols}) || lda < cols || ldb < cols || ldo < cols || !blm::extents_ok({rows, lda}) ||
      !blm::extents_ok({rows, ldb}) || !blm::extents_ok({rows, ldo}))
    return blm_fail(BLM_ERR_INVALID, "blm_add_cols: bad arguments");
  if (rows * (long)cols == 0) return BLM_OK;
  hipLaunchKernelGGL(add_cols_kernel, dim3(grid_for(rows * (long)cols)), dim3(TPB), 0, ST, a, (long)lda, b, (long)ldb, out, (long)ldo,
                     (long)rows, cols);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int blm_add_rowvec(float* x, const float* v, int B, int H, void* stream) {
  if (!x || !v || B < 0 || H < 0) return blm_fail(BLM_ERR_INVALID, "blm_add_rowvec: bad arguments");
  if ((long)B * H == 0) return BLM_OK;
  hipLaunchKernelGGL(add_rowvec_kernel, dim3(grid_for((long)B * H)), dim3(TPB), 0, ST, x, v, (long)B * H, H);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int blm_axpy(const float* x, float* y, int64_t n, float a, void* stream) {
  if (!x || !y || n < 0) return blm_fail(BLM_ERR_INVALID, "blm_axpy: bad arguments");
  if (n == 0) return BLM_OK;
  hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n)), dim3(TPB), 0, ST, x, y, (long)n, a);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}

extern "C" int blm_rows_gather_add(float* dst, const int64_t* slot, const float* src, int64_t V, int D, int64_t n_src,
                                   void* stream) {
  if (!dst || !slot || !src || V < 0 || D <= 0 || n_src < 0) return blm_fail(BLM_ERR_INVALID, "blm_rows_gather_add: bad arguments");
  if (V == 0 || n_src == 0) return BLM_OK;
  hipLaunchKernelGGL(rows_gather_add_kernel, dim3((unsigned)((V + 3) / 4)), dim3(TPB), 0, ST, dst, slot, src, (long)V, D,
                     (long)n_src);
  BLM_HIP(hipGetLastError());
  return BLM_OK;
}
