// Dropout keep factors shared by the HBM-bound kernels (elementwise.hip, search.hip).
// Masks are Philox bits keyed by the GLOBAL element index of a (rows, global_cols, D) tensor, so a
// data-parallel run that shards columns reproduces the single-process mask.
#pragma once
#include "blm_device.h"
#include "blm_host.h"

namespace blm {

struct DropKey {
  blm_rng rng;
  uint32_t thr;    // drop iff bits < thr
  float inv_keep;  // 1/(1-p)
  int B, D, col_offset, global_cols;
  bool on;
  int row0;        // global index of the call's first row (a (rows, B, D) block of a longer tensor: chunked LSTM stacks)
};

__host__ static DropKey make_key(float p, const blm_rng* rng, int B, int D, int col_offset, int global_cols) {
  DropKey k{};
  k.on = p > 0.f && rng != nullptr;
  if (k.on) k.rng = *rng;
  const double t = (double)p * 4294967296.0;
  k.thr = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
  k.inv_keep = p < 1.f ? 1.f / (1.f - p) : 0.f;
  k.B = B; k.D = D; k.col_offset = col_offset; k.global_cols = global_cols > 0 ? global_cols : B;
  return k;
}

// Scale factors (0 or 1/(1-p)) for the 4 consecutive features j..j+3 (j % 4 == 0, D % 4 == 0) of local (row, b).
__device__ __forceinline__ float4 keep4(const DropKey& k, int row, int b, int j) {
  if (!k.on) return make_float4(1.f, 1.f, 1.f, 1.f);
  const uint64_t g = ((uint64_t)(row + k.row0) * k.global_cols + (uint64_t)(k.col_offset + b)) * (uint64_t)k.D + (uint64_t)j;
  const u32x4 u = philox_block(k.rng, g >> 2);
  return make_float4(u.x >= k.thr ? k.inv_keep : 0.f, u.y >= k.thr ? k.inv_keep : 0.f,
                     u.z >= k.thr ? k.inv_keep : 0.f, u.w >= k.thr ? k.inv_keep : 0.f);
}
__device__ __forceinline__ float keep1(const DropKey& k, int row, int b, int j) {
  if (!k.on) return 1.f;
  const uint64_t g = ((uint64_t)(row + k.row0) * k.global_cols + (uint64_t)(k.col_offset + b)) * (uint64_t)k.D + (uint64_t)j;
  const u32x4 u = philox_block(k.rng, g >> 2);
  const int c = (int)(g & 3);
  const uint32_t bits = c == 0 ? u.x : (c == 1 ? u.y : (c == 2 ? u.z : u.w));
  return bits >= k.thr ? k.inv_keep : 0.f;
}

}  // namespace blm
