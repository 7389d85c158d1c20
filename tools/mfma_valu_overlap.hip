// Diagnostic: do vector-ALU instructions of ONE wave overlap with matrix instructions of ANOTHER wave on the same
// SIMD?  (Inside one in-order wave they overlap 10-15 %, tools/split_mfma_overlap.hip.)  A 512-thread workgroup =
// 8 waves, wave i on SIMD i % 4: with role = wave / 4 every SIMD holds one matrix wave and one vector wave; with
// role = wave % 2 the roles sit on different SIMDs (the no-contention reference).  Vector work = Philox4x32-10
// (v_mad_u64_u32 + xor), the dropout-mask generator of the attention / GEMM epilogues, or v_exp_f32 + fma chains.
// Prints the time of matrix only, vector only, both.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
  c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
}

// mode bit 0: matrix waves work, bit 1: vector waves work.  SHARE: roles share SIMDs.  VK: 0 Philox, 1 exp chain.
template <bool SHARE, int VK, int CHAINS = 1>
__global__ __launch_bounds__(512) void k(float* out, int mode, int mi, int vi, float seed) {
  const int wave = threadIdx.x >> 6;
  const int role = SHARE ? (wave >> 2) : (wave & 1);
  float res = 0.f;
  if (role == 0) {
    if (mode & 1) {
      f32x16 a = {0}, b = {0}, c = {0}, d = {0};
      float x = seed + threadIdx.x * 1e-3f, y = seed * 0.5f;
      for (int i = 0; i < mi; ++i) {
        if (CHAINS == 1) {  // one dependent chain, as the S / O accumulators of the attention kernels
          a = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a, 0, 0, 0);
        } else {            // four independent accumulators, as a GEMM wave tile
          a = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a, 0, 0, 0);
          b = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, b, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, c, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, d, 0, 0, 0);
        }
      }
      for (int r = 0; r < 16; ++r) res += a[r] + b[r] + c[r] + d[r];
    }
  } else if (mode & 2) {
    if (VK == 0) {
      uint32_t acc = 0;
      for (int i = 0; i < vi; ++i) {
        uint32_t c0 = threadIdx.x + i, c1 = blockIdx.x, c2 = 7u, c3 = acc, k0 = 1111u, k1 = 5u;
#pragma unroll
        for (int r = 0; r < 10; ++r) { philox_round(c0, c1, c2, c3, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
        acc ^= c0 ^ c1 ^ c2 ^ c3;
      }
      res = (float)acc;
    } else {
      float v = seed + threadIdx.x * 1e-4f;
      for (int i = 0; i < vi; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v = __builtin_amdgcn_exp2f(v * 0.25f) + v * 0.125f;
      }
      res = v;
    }
  }
  out[blockIdx.x * 512 + threadIdx.x] = res;
}

template <bool SHARE, int VK, int CHAINS = 1>
void run(float* out, int blocks, int mi, int vi) {
  float t[4] = {0, 0, 0, 0};
  for (int mode = 1; mode <= 3; ++mode) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<SHARE, VK, CHAINS><<<blocks, 512>>>(out, mode, mi / 10, vi / 10, 0.37f);
    hipEventRecord(e0);
    k<SHARE, VK, CHAINS><<<blocks, 512>>>(out, mode, mi, vi, 0.37f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&t[mode], e0, e1);
  }
  printf("%-22s %-8s %d MFMA chain(s) blocks=%d: matrix only %.3f ms, vector only %.3f ms, both %.3f ms  (sum %.3f, max %.3f) -> %.0f %% of the shorter one hidden\n",
         SHARE ? "roles share a SIMD" : "roles on separate SIMDs", VK == 0 ? "Philox" : "exp/fma", CHAINS, blocks, t[1], t[2], t[3], t[1] + t[2],
         t[1] > t[2] ? t[1] : t[2], 100.0 * (t[1] + t[2] - t[3]) / (t[1] < t[2] ? t[1] : t[2]));
}

int main() {
  float* out;
  hipMalloc(&out, 1024 * 512 * 4);
  const int mi = 5000;  // x4 MFMAs of 64 cycles = 1.28 M cycles
  run<true, 0>(out, 256, mi, 2600);
  run<false, 0>(out, 256, mi, 2600);
  run<true, 1>(out, 256, mi, 2000);
  run<false, 1>(out, 256, mi, 2000);
  run<true, 0>(out, 256, mi, 1300);
  run<true, 0>(out, 256, mi, 5200);
  run<true, 0, 4>(out, 256, mi, 2600);
  run<true, 1, 4>(out, 256, mi, 2000);
  run<false, 0, 4>(out, 256, mi, 2600);
  return 0;
}
