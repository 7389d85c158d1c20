// Diagnostic: can the bf16 split (VALU) of one fragment set overlap the bf16 MFMAs of another on a CDNA4 SIMD?
// One iteration = what a wave of the split GEMM does per k16 slab of a 64x64 wave tile: split 4 fragments of 8 fp32
// values into 3 bf16 parts each (VALU) and issue 24 v_mfma_f32_32x32x16_bf16 (768 matrix-pipe cycles).
// Variants: MFMA only, VALU only, both back to back (what the GEMM does), both software-pipelined (split of
// iteration i+1 written between the MFMAs of iteration i).  Cycles per iteration from s_memtime, at 1 and 2 waves/SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o split_mfma_overlap split_mfma_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ void split3(const float (&x)[8], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
  u32x4 h, m, l;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x2 v = {x[2 * q], x[2 * q + 1]};
    const uint32_t hu = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
    const f32x2 r = {v.x - __uint_as_float(hu << 16), v.y - __uint_as_float(hu & 0xFFFF0000u)};
    const uint32_t mu = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2));
    const f32x2 r2 = {r.x - __uint_as_float(mu << 16), r.y - __uint_as_float(mu & 0xFFFF0000u)};
    h[q] = hu; m[q] = mu;
    l[q] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r2, bf16x2));
  }
  hi = __builtin_bit_cast(bf16x8, h); mid = __builtin_bit_cast(bf16x8, m); lo = __builtin_bit_cast(bf16x8, l);
}

// one value pair -> its three packed bf16 part words
__device__ __forceinline__ void split_pair(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
  const f32x2 v = {x0, x1};
  h = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
  const f32x2 r = {v.x - __uint_as_float(h << 16), v.y - __uint_as_float(h & 0xFFFF0000u)};
  m = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2));
  const f32x2 r2 = {r.x - __uint_as_float(m << 16), r.y - __uint_as_float(m & 0xFFFF0000u)};
  l = __builtin_bit_cast(uint32_t, __builtin_convertvector(r2, bf16x2));
}

// MODE 4: the 24 MFMAs of the current part set with the split of the next set's 16 value pairs written BETWEEN them
// (3 MFMAs : 2 pairs), every unit fenced by sched_barrier so the order is the source order
template <int UNUSED>
__global__ __launch_bounds__(256) void k4(float* out, unsigned long long* cyc, int iters, float seed) {
  f32x16 acc[4] = {(f32x16)(0.f), (f32x16)(0.f), (f32x16)(0.f), (f32x16)(0.f)};
  float x[4][8];
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int e = 0; e < 8; ++e) x[f][e] = seed * (1.f + f) + threadIdx.x * 1e-3f * (e + 1);
  uint32_t P[2][4][3][4];  // [set][fragment][part][word] packed words
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int q = 0; q < 4; ++q) split_pair(x[f][2 * q], x[f][2 * q + 1], P[s][f][0][q], P[s][f][1][q], P[s][f][2][q]);
  auto body = [&](const int cur) {
    const int nxt = cur ^ 1;
#pragma unroll
    for (int u = 0; u < 8; ++u) {  // 8 units of 3 MFMAs + 2 pairs
#pragma unroll
      for (int w = 0; w < 3; ++w) {
        const int n = 3 * u + w, q = n >> 2, t = n & 3, i = t >> 1, j = t & 1;
        const int ia = q == 0 ? 2 : (q == 1 ? 0 : (q < 4 ? 1 : 0)), ib = q == 0 ? 0 : (q == 1 ? 2 : (q == 2 ? 1 : (q == 3 ? 0 : (q == 4 ? 1 : 0))));
        const u32x4 av = {P[cur][i][ia][0], P[cur][i][ia][1], P[cur][i][ia][2], P[cur][i][ia][3]};
        const u32x4 bv = {P[cur][2 + j][ib][0], P[cur][2 + j][ib][1], P[cur][2 + j][ib][2], P[cur][2 + j][ib][3]};
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc[t], 0, 0, 0);
        if (w < 2) {
          const int pr = 2 * u + w, f = pr >> 2, qq = pr & 3;
          x[f][2 * qq] += 1e-7f; x[f][2 * qq + 1] += 1e-7f;
          split_pair(x[f][2 * qq], x[f][2 * qq + 1], P[nxt][f][0][qq], P[nxt][f][1][qq], P[nxt][f][2][qq]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it += 2) { body(0); body(1); }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// MODE 5: same work as k4, one scheduling region per slab; the compiler orders it by sched_group_barrier (1 MFMA : NV VALU)
template <int NV>
__global__ __launch_bounds__(256) void k5(float* out, unsigned long long* cyc, int iters, float seed) {
  f32x16 acc[4] = {(f32x16)(0.f), (f32x16)(0.f), (f32x16)(0.f), (f32x16)(0.f)};
  float x[4][8];
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int e = 0; e < 8; ++e) x[f][e] = seed * (1.f + f) + threadIdx.x * 1e-3f * (e + 1);
  uint32_t P[2][4][3][4];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int q = 0; q < 4; ++q) split_pair(x[f][2 * q], x[f][2 * q + 1], P[s][f][0][q], P[s][f][1][q], P[s][f][2][q]);
  auto body = [&](const int cur) {
    const int nxt = cur ^ 1;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int n = 0; n < 24; ++n) {
      const int q = n >> 2, t = n & 3, i = t >> 1, j = t & 1;
      const int ia = q == 0 ? 2 : (q == 1 ? 0 : (q < 4 ? 1 : 0)), ib = q == 0 ? 0 : (q == 1 ? 2 : (q == 2 ? 1 : (q == 3 ? 0 : (q == 4 ? 1 : 0))));
      const u32x4 av = {P[cur][i][ia][0], P[cur][i][ia][1], P[cur][i][ia][2], P[cur][i][ia][3]};
      const u32x4 bv = {P[cur][2 + j][ib][0], P[cur][2 + j][ib][1], P[cur][2 + j][ib][2], P[cur][2 + j][ib][3]};
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        x[f][2 * qq] += 1e-7f; x[f][2 * qq + 1] += 1e-7f;
        split_pair(x[f][2 * qq], x[f][2 * qq + 1], P[nxt][f][0][qq], P[nxt][f][1][qq], P[nxt][f][2][qq]);
      }
#pragma unroll
    for (int n = 0; n < 24; ++n) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it += 2) { body(0); body(1); }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>  // 0 MFMA only, 1 VALU only, 2 back to back, 3 pipelined
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters, float seed) {
  f32x16 acc[4] = {(f32x16)(0.f), (f32x16)(0.f), (f32x16)(0.f), (f32x16)(0.f)};
  float x[4][8];
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int e = 0; e < 8; ++e) x[f][e] = seed * (1.f + f) + threadIdx.x * 1e-3f * (e + 1);
  bf16x8 p[4][3], pn[4][3];
#pragma unroll
  for (int f = 0; f < 4; ++f) { split3(x[f], p[f][0], p[f][1], p[f][2]); split3(x[f], pn[f][0], pn[f][1], pn[f][2]); }
  float sink = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 1 || MODE == 2) {
#pragma unroll
      for (int f = 0; f < 4; ++f) {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[f][e] += 1e-7f;  // new values every iteration (one add per element, as a load would cost an issue slot)
        split3(x[f], p[f][0], p[f][1], p[f][2]);
      }
    }
    if (MODE == 3) {
#pragma unroll
      for (int f = 0; f < 4; ++f) {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[f][e] += 1e-7f;
        split3(x[f], pn[f][0], pn[f][1], pn[f][2]);
      }
    }
    if (MODE != 1) {
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const int ia = q == 0 ? 2 : (q == 1 ? 0 : (q < 4 ? 1 : 0)), ib = q == 0 ? 0 : (q == 1 ? 2 : (q == 2 ? 1 : (q == 3 ? 0 : (q == 4 ? 1 : 0))));
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[2 * i + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p[i][ia], p[2 + j][ib], acc[2 * i + j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int f = 0; f < 4; ++f)  // consume every word of every part: otherwise most of the split is dead code
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const u32x4 w = __builtin_bit_cast(u32x4, p[f][c]);
          asm volatile("" ::"v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]));
        }
    }
    if (MODE == 3) {
#pragma unroll
      for (int q = 0; q < 24; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);
      }
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int c = 0; c < 3; ++c) { bf16x8 t = p[f][c]; p[f][c] = pn[f][c]; pn[f][c] = t; }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = sink;
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, float* out, unsigned long long* cyc, int blocks) {
  const int iters = 4000;
  k<MODE><<<blocks, 256>>>(out, cyc, 100, 0.37f);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(out, cyc, iters, 0.37f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  // s_memtime ticks at 100 MHz on this part: report wall ns per iteration instead
  printf("%-28s %d block(s)/CU: %.0f ns / iteration (= %.0f cycles at 2.3 GHz)\n", name, blocks / 256, ms * 1e6 / iters, ms * 1e6 / iters * 2.3);
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 2 * 256 * 256 * 4); hipMalloc(&cyc, 2 * 256 * 8);
  for (int blocks : {256, 512}) {
    run<0>("24 MFMA only", out, cyc, blocks);
    run<1>("split of 4 fragments only", out, cyc, blocks);
    run<2>("split then MFMA", out, cyc, blocks);
    run<3>("pipelined (sched groups)", out, cyc, blocks);
    {
      const int iters = 4000;
      k4<0><<<blocks, 256>>>(out, cyc, 100, 0.37f);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      k4<0><<<blocks, 256>>>(out, cyc, iters, 0.37f);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("%-28s %d block(s)/CU: %.0f ns / iteration\n", "hand-interleaved 3:2", blocks / 256, ms * 1e6 / iters);
    }
#define RUN5(NV) { const int iters = 4000; k5<NV><<<blocks, 256>>>(out, cyc, 100, 0.37f); hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); \
      hipEventRecord(e0); k5<NV><<<blocks, 256>>>(out, cyc, iters, 0.37f); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); \
      printf("sched groups 1 MFMA : %d VALU    %d block(s)/CU: %.0f ns / iteration\n", NV, blocks / 256, ms * 1e6 / iters); }
    RUN5(4) RUN5(6) RUN5(8) RUN5(9) RUN5(12)
  }
  return 0;
}
