// Probe (measurement only, not linked into the library): an fp32-accurate GEMM on the bf16 matrix
// cores by operand splitting.  C[M,N] = A[M,K] . B[N,K]^T with A = A0 + A1 + A2, B = B0 + B1 + B2, every part
// a bf16 matrix (round-to-nearest-even residual split), and the six largest of the nine part products
//   A0B0, A0B1, A1B0, A1B1, A0B2, A2B0
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  With exact fp32 accumulation (tools/split_bf16_accuracy.py,
// CPU emulation) the six-product form is as accurate as an fp32 GEMM (1.5e-7 vs 3.6e-7 max-rel at K = 4096); the
// matrix core's own accumulation of the 16 products of an instruction is not exact fp32, and on the hardware the
// six-product form lands at 2.0e-6 and the three-product form (two parts) at 4.5e-6.  Matrix-core time per
// 32x32x16 tile: 6 x 32 (3 x 32) cycles against 8 x 64 for v_mfma_f32_32x32x2_f32: the question this probe answers
// is how much of that survives the operand traffic (3 x 2 B, or 2 x 2 B = the fp32 bytes, per element).
//
// Measured on MI355X (this naive single-stage kernel; the production fp32 kernel runs this shape at 127 TFLOP/s):
//   M 8192 N 512 K 4096:  x6 238 us = 144 TF-equivalent (2.0e-6), x3 131 us = 262 TF-eq (4.5e-6), x1 66 us = 523 TF (2.2e-3)
//   M 8192 N 4096 K 512:  x6 271 us = 127 TF-eq (6.1e-7),         x3 162 us = 213 TF-eq (4.3e-6), x1 85 us = 404 TF
// The split pass over X (fp32 -> 3 bf16 matrices) is 64 us; a production kernel would split inside its tile loader.
//
// Shape = the roofline kernel's (M 8192, N 512, K 4096).  128x128x64 tiles, 4 waves (64x64 each), one LDS
// stage of six part tiles (rows padded to 144 B: conflict-free ds_read_b128), next stage prefetched in
// registers.  Prints: split pass time, GEMM time, fp32-equivalent TFLOP/s, max relative error vs fp64 on
// sampled entries.     hipcc --offload-arch=gfx950 -O3 -o split_bf16_gemm_probe split_bf16_gemm_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;  // plain vector: HIP's uint4 struct in a 2-D array defeats SROA (scratch)

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int RS = 144;                     // LDS row stride in bytes (128 + 16 pad)
constexpr int PART = 128 * RS;              // one part tile (128 rows)
constexpr int LDS_BYTES = 6 * PART;         // A0 A1 A2 B0 B1 B2

__device__ __forceinline__ uint16_t bf16_rne(float x) {
  uint32_t u = __float_as_uint(x);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf16_f(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

// x -> three bf16 parts, each matrix stored contiguously: out + p * n
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, uint16_t* __restrict__ out, long n) {
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
    const float4 v = *reinterpret_cast<const float4*>(x + i);
    const float f[4] = {v.x, v.y, v.z, v.w};
    uint16_t p0[4], p1[4], p2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      p0[e] = bf16_rne(f[e]);
      const float r1 = f[e] - bf16_f(p0[e]);
      p1[e] = bf16_rne(r1);
      p2[e] = bf16_rne(r1 - bf16_f(p1[e]));
    }
    *reinterpret_cast<uint2*>(out + i) = make_uint2(p0[0] | ((uint32_t)p0[1] << 16), p0[2] | ((uint32_t)p0[3] << 16));
    *reinterpret_cast<uint2*>(out + n + i) = make_uint2(p1[0] | ((uint32_t)p1[1] << 16), p1[2] | ((uint32_t)p1[3] << 16));
    *reinterpret_cast<uint2*>(out + 2 * n + i) = make_uint2(p2[0] | ((uint32_t)p2[1] << 16), p2[2] | ((uint32_t)p2[3] << 16));
  }
}

// NPROD = 6 (fp32-accurate), 3 (two parts, 4.5e-6) or 1 (plain bf16): same data movement of the parts it uses
template <int NPROD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_split_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B,
                                                         float* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  constexpr int NP = NPROD == 6 ? 3 : (NPROD == 3 ? 2 : 1);  // parts per operand
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware order: consecutive workgroup ids land on different XCDs; give each XCD a contiguous band of M tiles
  const int nbn = N / BN, nbm = M / BM, nb = nbn * nbm;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = nb >> 3;
  const int bid = (nb & 7) == 0 ? xcd * per + slot : blockIdx.x;
  const int m0 = (bid / nbn) * BM, n0 = (bid % nbn) * BN;
  const long an = (long)M * K, bn = (long)N * K;

  // staging: a part tile is 128 rows x 128 B = 1024 16-B chunks = 4 per thread; thread -> (row = c >> 3, chunk = c & 7)
  u32x4 ra[NP][4], rb[NP][4];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = t + 256 * u, row = c >> 3, ch = c & 7;
        ra[p][u] = *reinterpret_cast<const u32x4*>(A + p * an + (long)(m0 + row) * K + k0 + ch * 8);
        rb[p][u] = *reinterpret_cast<const u32x4*>(B + p * bn + (long)(n0 + row) * K + k0 + ch * 8);
      }
  };
  auto stash = [&]() {
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = t + 256 * u, row = c >> 3, ch = c & 7;
        *reinterpret_cast<u32x4*>(sm + p * PART + row * RS + ch * 16) = ra[p][u];
        *reinterpret_cast<u32x4*>(sm + (3 + p) * PART + row * RS + ch * 16) = rb[p][u];
      }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16)(0.f);

  fetch(0);
  for (int k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();
    stash();
    __syncthreads();
    fetch(k0 + BK < K ? k0 + BK : k0);  // branch-free (a conditional refill demotes the register ring to scratch)
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      bf16x8 a[NP][2], b[NP][2];
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a[p][i] = *reinterpret_cast<const bf16x8*>(sm + p * PART + (wm * 64 + i * 32 + li) * RS + s * 32 + lh * 16);
          b[p][i] = *reinterpret_cast<const bf16x8*>(sm + (3 + p) * PART + (wn * 64 + i * 32 + li) * RS + s * 32 + lh * 16);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          // smallest terms first
          if constexpr (NPROD == 6) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], b[0][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[2][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[1][j], acc[i][j], 0, 0, 0);
          }
          if constexpr (NPROD >= 3) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1][j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], acc[i][j], 0, 0, 0);
        }
    }
  }
  // C^T-free store: acc[i][j][r] is row (r&3) + 8*(r>>2) + 4*lh, column li of the 32x32 tile
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int col = n0 + wn * 64 + j * 32 + li;
        C[(long)row * N + col] = acc[i][j][r];
      }
}

template <int NPROD>
static float time_gemm(const uint16_t* A, const uint16_t* B, float* C, int M, int N, int K, int reps) {
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_kernel<NPROD>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  const dim3 grid((M / BM) * (N / BN));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm_split_kernel<NPROD>, grid, dim3(256), LDS_BYTES, 0, A, B, C, M, N, K);
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm_split_kernel<NPROD>, grid, dim3(256), LDS_BYTES, 0, A, B, C, M, N, K);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

static double check(const std::vector<float>& hx, const std::vector<float>& hw, const float* dC, int M, int N, int K) {
  std::vector<float> hc((size_t)M * N);
  CK(hipMemcpy(hc.data(), dC, hc.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0, scale = 0;
  srand(3);
  for (int s = 0; s < 4096; ++s) {
    const int m = rand() % M, n = rand() % N;
    double ref = 0;
    for (int k = 0; k < K; ++k) ref += (double)hx[(size_t)m * K + k] * (double)hw[(size_t)n * K + k];
    worst = fmax(worst, fabs(ref - (double)hc[(size_t)m * N + n]));
    scale = fmax(scale, fabs(ref));
  }
  return worst / scale;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 8192, N = argc > 2 ? atoi(argv[2]) : 512, K = argc > 3 ? atoi(argv[3]) : 4096;
  if (M % BM || N % BN || K % BK) { printf("shape must tile by %dx%dx%d\n", BM, BN, BK); return 1; }
  std::vector<float> hx((size_t)M * K), hw((size_t)N * K);
  srand(1);
  for (auto& v : hx) v = (float)rand() / (float)RAND_MAX * 2.f - 1.f;
  for (auto& v : hw) v = ((float)rand() / (float)RAND_MAX * 2.f - 1.f) / sqrtf((float)K);
  float *dx, *dw, *dc;
  uint16_t *ax, *bw;
  CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dw, hw.size() * 4)); CK(hipMalloc(&dc, (size_t)M * N * 4));
  CK(hipMalloc(&ax, hx.size() * 6)); CK(hipMalloc(&bw, hw.size() * 6));
  CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(split3_kernel, dim3(2048), dim3(256), 0, 0, dx, ax, (long)hx.size());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(split3_kernel, dim3(2048), dim3(256), 0, 0, dx, ax, (long)hx.size());
  CK(hipEventRecord(e1));
  hipLaunchKernelGGL(split3_kernel, dim3(2048), dim3(256), 0, 0, dw, bw, (long)hw.size());
  CK(hipEventSynchronize(e1));
  float ms_split;
  CK(hipEventElapsedTime(&ms_split, e0, e1));
  CK(hipDeviceSynchronize());
  const double gf = 2.0 * M * N * K * 1e-9;
  printf("shape M=%d N=%d K=%d (%.2f GFLOP); split of X into 3 bf16 parts: %.1f us\n", M, N, K, gf, ms_split * 1e3);
  const float t6 = time_gemm<6>(ax, bw, dc, M, N, K, 20);
  printf("bf16 x6 (3 parts, fp32-accurate): %.1f us = %.1f fp32-equivalent TFLOP/s, max-rel error vs fp64 %.2e\n", t6 * 1e3,
         gf / t6, check(hx, hw, dc, M, N, K));
  const float t3 = time_gemm<3>(ax, bw, dc, M, N, K, 20);
  printf("bf16 x3 (2 parts):                %.1f us = %.1f fp32-equivalent TFLOP/s, max-rel error vs fp64 %.2e\n", t3 * 1e3,
         gf / t3, check(hx, hw, dc, M, N, K));
  const float t1 = time_gemm<1>(ax, bw, dc, M, N, K, 20);
  printf("bf16 x1 (plain bf16 operands):    %.1f us = %.1f TFLOP/s, max-rel error vs fp64 %.2e\n", t1 * 1e3, gf / t1,
         check(hx, hw, dc, M, N, K));
  return 0;
}
