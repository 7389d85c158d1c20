// Diagnostic: sustained v_mfma_f32_32x32x2_f32 rate with no memory traffic (what the chip holds
// under load), for sizing the fp32 GEMM's real ceiling: cycles per MFMA from s_memtime inside the
// kernel, wall TFLOP/s and the implied clock, at 1 and 2 waves per SIMD, bare and with the GEMM's
// filler mix (one ds_read_b128 + one v_add per 4 MFMAs).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
template <int FILL>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters, float seed) {
  __shared__ float4 lds[1024];
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  float x = seed + threadIdx.x * 1e-3f, y = seed * 0.5f + threadIdx.x * 2e-3f;
  lds[threadIdx.x] = make_float4(x, y, x, y);
  __syncthreads();
  int addr = threadIdx.x;
  float4 f = make_float4(0, 0, 0, 0);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (FILL) { f = lds[addr & 1023]; addr += 64; }
    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
    if (FILL) { x += f.x * 1e-9f; }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int FILL>
void run(float* out, unsigned long long* cyc, int blocks) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<FILL><<<blocks, 256>>>(out, cyc, 1000, 0.37f);
  hipEventRecord(e0);
  k<FILL><<<blocks, 256>>>(out, cyc, iters, 0.37f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2048];
  hipMemcpy(h, cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < blocks; ++i) avg += (double)h[i]; avg /= blocks;
  const double per_wave_mfma = iters * 4.0;
  const int waves_per_simd = (blocks + 255) / 256;
  double flops = (double)blocks * 4 * iters * 4 * 4096.0;
  printf("fill=%d blocks=%d (%d wave/SIMD)  %.3f ms  %.1f TFLOP/s  | %.1f cycles per own MFMA, %.1f per SIMD MFMA, implied clock %.2f GHz\n",
         FILL, blocks, waves_per_simd, ms, flops / ms / 1e9, avg / per_wave_mfma, avg / per_wave_mfma / waves_per_simd, avg / (ms * 1e6));
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 4096 * 256 * 4);
  hipMalloc(&cyc, 4096 * 8);
  for (int blocks : {256, 512}) { run<0>(out, cyc, blocks); run<1>(out, cyc, blocks); }
  return 0;
}
