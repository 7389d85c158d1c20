// Microbenchmark for the decision "persistent LSTM recurrence vs one launch per time step" (VERDICT r1 #3,
// SURVEY K6): what does ONE time step's cross-CU exchange cost inside a persistent launch on MI355X?
//
// Geometry of the would-be persistent kernel at cfg2 (B = 64, H = 1024): 256 co-resident workgroups (one per CU:
// 128 KB of LDS hold a workgroup's W_hh slice), each owns 8 hidden units x 32 batch rows.  Per step a workgroup
//   (1) publishes its h slice   : 32 x 8 floats = 1 KB      (write-through sc1 stores, drained)
//   (2) joins a grid-wide barrier over all 8 XCDs
//   (3) reads the h rows it multiplies with: 32 x 1024 floats = 128 KB written by 128 other workgroups (sc1 loads)
// Variants timed (us per step = (t(S2) - t(S1)) / (S2 - S1), so launch overhead cancels):
//   flat   : one monotonic counter, lane-0 agent atomic add, relaxed sc1 poll + s_sleep
//   xcd    : per-XCD counter -> XCD leader adds to a top counter -> per-XCD generation word
//   +xchg  : barrier plus (1) and (3)
//   launch : the same exchange as ONE LAUNCH PER STEP (what lstm_step.hip pays today): kernel boundary + gather
//   flags  : (round 4) NO barrier: every producer raises a word of its own after its slice has drained (sc1), a consumer wave
//            polls all 256 words with ONE dwordx4 sc1 load per lane, and the slices are then read THROUGH the L2 (plain loads)
//            -- legal without an invalidate because every step writes a fresh region that no XCD has read before (as the
//            real kernel's y[t] rows would be); 32 workgroups of an XCD then share one fetch of each line
// Every spin is bounded (a timeout word is set and the kernel drains), grid = 256 <= one workgroup per CU.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/grid_barrier_bench tools/grid_barrier_bench.hip && tools/grid_barrier_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                          \
  do {                                                                                    \
    hipError_t e_ = (x);                                                                  \
    if (e_ != hipSuccess) {                                                               \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));   \
      exit(1);                                                                            \
    }                                                                                     \
  } while (0)

constexpr int NWG = 256, TPB = 256;
constexpr int SLICE_FLOATS = 256;             // 1 KB published per workgroup and step
constexpr int GATHER_FLOATS = 32 * 1024;      // 128 KB read per workgroup and step
constexpr unsigned SPIN_LIMIT = 1u << 22;     // ~ seconds; then the timeout word is set and every loop exits
typedef unsigned v4u __attribute__((ext_vector_type(4)));

struct Sync {            // every polled word on a 128-byte line of its own
  unsigned flat[32];
  unsigned top[32];
  unsigned xcd_cnt[8][32];
  unsigned xcd_gen[8][32];
  unsigned census[8][32];
  unsigned timeout[32];
  unsigned ready[NWG];   // mode 4/5: ready[p] = number of steps producer p has published
};

__device__ __forceinline__ unsigned ld_relaxed(unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 7u;
}

// lane 0 of the workgroup; returns false on timeout
__device__ bool wait_ge(unsigned* p, unsigned want, unsigned* tmo) {
  for (unsigned spins = 0; spins < SPIN_LIMIT; ++spins) {
    if ((int)(ld_relaxed(p) - want) >= 0) return true;
    if ((spins & 1023u) == 1023u && ld_relaxed(tmo)) return false;
    __builtin_amdgcn_s_sleep(1);
  }
  __hip_atomic_store(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return false;
}

// flat barrier, epoch e = 1, 2, ...; call by all threads
__device__ bool barrier_flat(Sync* s, unsigned e) {
  __shared__ int ok;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its sc1 stores
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(&s->flat[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ok = wait_ge(&s->flat[0], e * NWG, &s->timeout[0]);
  }
  __syncthreads();
  return ok;
}

// XCD-hierarchical barrier; n_x = workgroups on this XCD, n_xcd = XCDs that hold workgroups
__device__ bool barrier_xcd(Sync* s, unsigned e, unsigned x, unsigned n_x, unsigned n_xcd) {
  __shared__ int ok2;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(&s->xcd_cnt[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool good = true;
    if (old == e * n_x - 1) {  // last arriver of this XCD = its leader for this epoch
      __hip_atomic_fetch_add(&s->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      good = wait_ge(&s->top[0], e * n_xcd, &s->timeout[0]);
      if (good) __hip_atomic_store(&s->xcd_gen[x][0], e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      good = wait_ge(&s->xcd_gen[x][0], e, &s->timeout[0]);
    }
    ok2 = good;
  }
  __syncthreads();
  return ok2;
}

// (1) publish 1 KB: one dwordx4 sc1 store per lane of wave 0 (8 whole 128-B lines by one instruction)
__device__ __forceinline__ void publish(float* hx, int wg, unsigned step, float seed) {
  if (threadIdx.x < 64) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(hx, 0, NWG * SLICE_FLOATS * 4, 0x00020000);
    v4u v;
    const float f = seed + (float)step;
    v.x = v.y = v.z = v.w = __float_as_uint(f);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (wg * SLICE_FLOATS + threadIdx.x * 4) * 4, 0, 16);  // aux 16 = sc1
  }
}

// (3) gather 128 KB = the slices of the 128 workgroups of this workgroup's batch half: 8192 x 16 B, 32 per thread
template <int AUX>
__device__ __forceinline__ float gather_aux(const float* hx, int wg) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(hx), 0, NWG * SLICE_FLOATS * 4, 0x00020000);
  const int half = wg & 1;
  float acc = 0.f;
#pragma unroll 4
  for (int i = 0; i < 32; i += 8) {
    v4u v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int chunk = (i + j) * TPB + threadIdx.x;
      const int src_wg = (chunk >> 6) * 2 + half;
      v[j] = __builtin_amdgcn_raw_buffer_load_b128(r, (src_wg * SLICE_FLOATS + (chunk & 63) * 4) * 4, 0, AUX);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += __uint_as_float(v[j].x) + __uint_as_float(v[j].w);
  }
  return acc;
}

// mode 4/5: wave 0 waits until every producer has published step `want` (one 1 KB sc1 load per poll), bounded
__device__ bool wait_all_ready(Sync* s, unsigned want) {
  __shared__ int ok3;
  if (threadIdx.x < 64) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(s->ready, 0, NWG * 4, 0x00020000);
    bool good = false;
    for (unsigned spins = 0; spins < SPIN_LIMIT; ++spins) {
      const v4u f = __builtin_amdgcn_raw_buffer_load_b128(r, threadIdx.x * 16, 0, 16);
      const bool mine = (int)(f.x - want) >= 0 && (int)(f.y - want) >= 0 && (int)(f.z - want) >= 0 && (int)(f.w - want) >= 0;
      if (__all(mine)) { good = true; break; }
      if ((spins & 1023u) == 1023u && ld_relaxed(&s->timeout[0])) break;
    }
    if (!good && threadIdx.x == 0) __hip_atomic_store(&s->timeout[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 0) ok3 = good;
  }
  __syncthreads();
  return ok3;
}

__device__ __forceinline__ float gather(const float* hx, int wg) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(hx), 0, NWG * SLICE_FLOATS * 4, 0x00020000);
  const int half = wg & 1;
  float acc = 0.f;
#pragma unroll 4
  for (int i = 0; i < 32; i += 8) {
    v4u v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int chunk = (i + j) * TPB + threadIdx.x;       // 0 .. 8191: 16-B chunk of this half's 128 KB
      const int src_wg = (chunk >> 6) * 2 + half;          // 64 chunks per 1 KB slice
      v[j] = __builtin_amdgcn_raw_buffer_load_b128(r, (src_wg * SLICE_FLOATS + (chunk & 63) * 4) * 4, 0, 16);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += __uint_as_float(v[j].x) + __uint_as_float(v[j].w);
  }
  return acc;
}

// mode: 0 flat, 1 xcd, 2 flat + exchange, 3 xcd + exchange
__global__ __launch_bounds__(TPB) void persistent_kernel(Sync* s, float* hx0, float* hx1, float* out, int steps, int mode, float seed) {
  extern __shared__ __attribute__((aligned(16))) char lds[];  // 128 KB requested: one workgroup per CU, as the real kernel
  __shared__ unsigned sh[2];
  const int wg = blockIdx.x;
  const unsigned x = xcc_id();
  // census: how many workgroups sit on each XCD (placement is not promised)
  if (threadIdx.x == 0) __hip_atomic_fetch_add(&s->census[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (!barrier_flat(s, 1)) return;
  if (threadIdx.x == 0) {
    unsigned nx = 0, nxcd = 0;
    for (int i = 0; i < 8; ++i) {
      const unsigned c = ld_relaxed(&s->census[i][0]);
      if ((unsigned)i == x) nx = c;
      nxcd += c != 0;
    }
    sh[0] = nx;
    sh[1] = nxcd;
  }
  __syncthreads();
  const unsigned n_x = sh[0], n_xcd = sh[1];
  float acc = 0.f;
  reinterpret_cast<float*>(lds)[threadIdx.x] = 0.f;
  if (mode >= 4) {  // hx0 = steps fresh regions of NWG slices each
    for (int t = 0; t < steps; ++t) {
      float* hw = hx0 + (size_t)t * NWG * SLICE_FLOATS;
      if (mode == 5) publish(hw, wg, (unsigned)t, seed);
      if (threadIdx.x < 64) {  // the publishing wave: slice drained, then its word
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) __hip_atomic_store(&s->ready[wg], (unsigned)t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (!wait_all_ready(s, (unsigned)t + 1)) break;
      if (mode == 5) acc += gather_aux<0>(hw, wg);
    }
    out[wg * TPB + threadIdx.x] = acc;
    return;
  }
  for (int t = 0; t < steps; ++t) {
    float* hw = (t & 1) ? hx1 : hx0;  // double-buffered exchange area: step t writes one, reads it back after the barrier
    if (mode >= 2) publish(hw, wg, (unsigned)t, seed);
    const bool ok = (mode & 1) ? barrier_xcd(s, (unsigned)t + 1, x, n_x, n_xcd) : barrier_flat(s, (unsigned)t + 2);
    if (!ok) break;
    if (mode >= 2) acc += gather(hw, wg);
  }
  out[wg * TPB + threadIdx.x] = acc;
}

// the same exchange, one launch per step: publish (from the previous launch's "result") + gather
__global__ __launch_bounds__(TPB) void step_kernel(float* hw, const float* hr, float* out, int t) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  reinterpret_cast<float*>(lds)[threadIdx.x] = 0.f;
  const float acc = gather(hr, blockIdx.x);
  publish(hw, blockIdx.x, (unsigned)t, 1.0f);
  out[blockIdx.x * TPB + threadIdx.x] = acc;
}

static float run_persistent(Sync* s, float* hx0, float* hx1, float* out, int steps, int mode, hipStream_t st, int lds) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CHECK(hipMemsetAsync(s, 0, sizeof(Sync), st));
    CHECK(hipEventRecord(a, st));
    hipLaunchKernelGGL(persistent_kernel, dim3(NWG), dim3(TPB), lds, st, s, hx0, hx1, out, steps, mode, 1.0f + rep);  // another value per repeat: a line kept from the last one would show
    CHECK(hipEventRecord(b, st));
    CHECK(hipStreamSynchronize(st));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  Sync h;
  CHECK(hipMemcpy(&h, s, sizeof(Sync), hipMemcpyDeviceToHost));
  if (h.timeout[0]) {
    fprintf(stderr, "TIMEOUT in mode %d (grid not co-resident?)\n", mode);
    exit(2);
  }
  return best;
}

int main() {
  hipStream_t st;
  CHECK(hipStreamCreate(&st));
  Sync* s;
  float *hx0, *hx1, *out;
  CHECK(hipMalloc(&s, sizeof(Sync)));
  CHECK(hipMalloc(&hx0, NWG * SLICE_FLOATS * 4));
  CHECK(hipMalloc(&hx1, NWG * SLICE_FLOATS * 4));
  CHECK(hipMalloc(&out, NWG * TPB * 4));
  CHECK(hipMemset(hx0, 0, NWG * SLICE_FLOATS * 4));
  CHECK(hipMemset(hx1, 0, NWG * SLICE_FLOATS * 4));
  const int lds = 128 * 1024;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(persistent_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(step_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  int nblk = 0;
  CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, persistent_kernel, TPB, lds));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  printf("device %s, %d CUs, occupancy query %d workgroup(s)/CU at %d KB LDS\n", prop.gcnArchName, prop.multiProcessorCount, nblk, lds >> 10);
  if (prop.multiProcessorCount * nblk < NWG) {
    fprintf(stderr, "grid of %d would not be co-resident\n", NWG);
    return 3;
  }
  const char* names[6] = {"barrier flat", "barrier xcd", "barrier flat + publish 1 KB + gather 128 KB",
                          "barrier xcd  + publish 1 KB + gather 128 KB", "ready words only (256 words, one 1 KB poll)",
                          "ready words + publish 1 KB + gather 128 KB through the L2 (fresh region per step)"};
  const int S1 = 64, S2 = 1088;
  float* fresh;
  CHECK(hipMalloc(&fresh, (size_t)(S2 + 8) * NWG * SLICE_FLOATS * 4));
  printf("{\"tool\": \"grid_barrier_bench\", \"workgroups\": %d, \"results_us_per_step\": {", NWG);
  for (int mode = 0; mode < 6; ++mode) {
    if (mode >= 4) hx0 = fresh;
    run_persistent(s, hx0, hx1, out, 8, mode, st, lds);  // warm
    const float t1 = run_persistent(s, hx0, hx1, out, S1, mode, st, lds);
    if (mode >= 2 && mode != 4) {  // every word of every step's hand-off must be the value published in THAT step (no stale line)
      std::vector<float> h(NWG * TPB);
      CHECK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
      const float want = 64.f * (S1 * 5.f + S1 * (S1 - 1) / 2);  // last repeat: seed 5
      for (size_t i = 0; i < h.size(); ++i)
        if (h[i] != want) {
          fprintf(stderr, "STALE hand-off in mode %d: out[%zu] = %.1f, want %.1f\n", mode, i, h[i], want);
          return 4;
        }
    }
    const float t2 = run_persistent(s, hx0, hx1, out, S2, mode, st, lds);
    printf("\"%s\": %.3f, ", names[mode], 1e3f * (t2 - t1) / (S2 - S1));
  }
  // one launch per step
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  float best = 1e30f;
  const int S = 1024;
  for (int rep = 0; rep < 5; ++rep) {
    CHECK(hipEventRecord(a, st));
    for (int t = 0; t < S; ++t)
      hipLaunchKernelGGL(step_kernel, dim3(NWG), dim3(TPB), lds, st, (t & 1) ? hx1 : hx0, (t & 1) ? hx0 : hx1, out, t);
    CHECK(hipEventRecord(b, st));
    CHECK(hipStreamSynchronize(st));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  printf("\"one launch per step: gather 128 KB + publish 1 KB\": %.3f}}\n", 1e3f * best / S);
  return 0;
}
