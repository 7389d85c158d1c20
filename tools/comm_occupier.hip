// Stand-in for RCCL's channel workgroups on a ONE-GPU box (tools/comm_occupancy_rehearsal.py; VERDICT r3 "next" #2).
//
// What a gradient bucket's all-reduce costs the backward GEMMs that run beside it is CU time, not bytes: RCCL launches
// one persistent workgroup per channel for the duration of the collective.  The footprint is read off the gfx950 code object
// of the librccl.so in this image (rcclGenericKernel<1..4, *>, llvm-readelf --notes): 256 threads = one wave per SIMD,
// 261-280 vector + 17-32 accumulation registers per lane, 19,744 bytes of LDS, 352-360 bytes of scratch.  With ~300 of a
// SIMD's 512 registers per lane gone, a CU that hosts a channel cannot also host the eight-wave 128 x 128 GEMM workgroup
// (2 waves x 142 registers per SIMD), and holds one 4-wave GEMM workgroup where it would hold two or three.
//
// occupier_kernel reproduces that footprint (register claims through asm clobbers, a static LDS array) and behaves like a
// ring step loop: each workgroup streams its slice of a bucket (dst += src: two reads + one write per element, what a
// local reduce moves through HBM) in `chunks` pieces and paces itself so that the whole bucket takes `ticks` of the 100 MHz
// wall clock -- the time the bucket would spend on the xGMI links at a given bus bandwidth.  Every wait is bounded: a
// workgroup leaves after `ticks` + a fixed slack whatever the clock says, so the grid always drains.
#include <hip/hip_runtime.h>

#include <cstdint>

namespace {

constexpr int kThreads = 256;
constexpr int kLdsBytes = 19744;

__global__ __launch_bounds__(kThreads) void occupier_kernel(float* __restrict__ dst, const float* __restrict__ src, long n_per_wg,
                                                            int chunks, long ticks, int claim_regs) {
  __shared__ float lds[kLdsBytes / 4];
  // register footprint of rcclGenericKernel: 280 vector registers (v255 + a23 clobbered => >= 256 + 24)
  if (claim_regs) {
    asm volatile("v_mov_b32 v255, 0" ::: "v255");
    asm volatile("v_accvgpr_write_b32 a23, 0" ::: "a23");
  }
  const long t0 = (long)wall_clock64();
  const long base = (long)blockIdx.x * n_per_wg;
  const long per_chunk = (n_per_wg + chunks - 1) / chunks;
  float keep = 0.f;
  for (int c = 0; c < chunks; ++c) {
    const long lo = base + (long)c * per_chunk;
    long hi = lo + per_chunk;
    if (hi > base + n_per_wg) hi = base + n_per_wg;
    for (long i = lo + threadIdx.x * 4; i + 3 < hi; i += kThreads * 4) {
      const float4 a = *reinterpret_cast<const float4*>(src + i);
      float4 b = *reinterpret_cast<float4*>(dst + i);
      b.x += a.x; b.y += a.y; b.z += a.z; b.w += a.w;
      *reinterpret_cast<float4*>(dst + i) = b;
      keep += b.x;
    }
    lds[(threadIdx.x + c) % (kLdsBytes / 4)] = keep;
    // pace: this chunk's share of the bucket's time on the links.  Bounded spin (<= 2^20 sleeps of ~0.6 us).
    const long due = t0 + ticks * (long)(c + 1) / chunks;
    for (int spin = 0; spin < (1 << 20) && (long)wall_clock64() < due; ++spin) __builtin_amdgcn_s_sleep(32);
  }
  if (keep == 123456.789f) dst[base] = lds[threadIdx.x];  // keeps the LDS array and the sum alive
}

}  // namespace

// bucket of `n` floats (dst += src) on `wgs` channel workgroups, paced to take `us` microseconds in total.
extern "C" int occupier_launch(float* dst, const float* src, long n, int wgs, double us, int chunks, int claim_regs, void* stream) {
  if (!dst || !src || n <= 0 || wgs <= 0 || wgs > 1024 || chunks <= 0 || us < 0) return 1;
  const long n_per_wg = (n / wgs) / 4 * 4;
  if (n_per_wg < 4) return 1;
  const long ticks = (long)(us * 100.0);  // wall_clock64: 100 MHz
  hipLaunchKernelGGL(occupier_kernel, dim3(wgs), dim3(kThreads), 0, static_cast<hipStream_t>(stream), dst, src, n_per_wg, chunks, ticks,
                     claim_regs);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
