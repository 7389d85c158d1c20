// Diagnostic: sustained v_mfma_f32_32x32x2_f32 rate with no memory traffic (what the chip holds
// under load), for sizing the fp32 GEMM's real ceiling.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  float x = seed + threadIdx.x * 1e-3f, y = seed * 0.5f + threadIdx.x * 2e-3f;
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* out;
  hipMalloc(&out, 4096 * 256 * 4);
  for (int blocks : {256, 512, 1024}) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<blocks, 256>>>(out, 1000, 0.37f);
    hipEventRecord(e0);
    k<<<blocks, 256>>>(out, iters, 0.37f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * iters * 4 * 4096.0;
    printf("blocks=%d  %.3f ms  %.1f TFLOP/s\n", blocks, ms, flops / ms / 1e9);
  }
  return 0;
}
