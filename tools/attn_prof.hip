// Phase stamps of the causal attention forward kernel (cfg3 shape: T 128, B 64, 8 heads x 64) built from the production
// source with -DBLM_ATTN_PROF.  Per wave: entry -> loads landed + LDS written -> barrier -> S tiles + row max ->
// exp + row sum -> dropout + P V -> stores.  Medians per query tile.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DBLM_ATTN_PROF -I include -I bayeslms_amd/csrc -o tools/attn_prof \
//         tools/attn_prof.hip bayeslms_amd/csrc/capi.hip
#include "../bayeslms_amd/csrc/attention_mfma.hip"

#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int HPW>
static void run(float pdrop) {
  const int T = 128, B = 64, nh = 8, d = nh * 64;
  float *qkv, *out, *lse;
  CK(hipMalloc(&qkv, (size_t)T * B * 3 * d * 4));
  CK(hipMalloc(&out, (size_t)T * B * d * 4));
  CK(hipMalloc(&lse, (size_t)B * nh * T * 4));
  CK(hipMemset(qkv, 0, (size_t)T * B * 3 * d * 4));
  const int nwg = B * nh / HPW, nwave = 4 * HPW;
  long long* prof;
  CK(hipMalloc(&prof, (size_t)nwg * nwave * 8 * 8));
  AttnM p{};
  blm_rng rng{1234, 0x20000000u, 1};
  fill_m(p, T, B, nh, pdrop, &rng, 0);
  p.q = qkv; p.k = qkv + d; p.v = qkv + 2 * d; p.ld = 3 * d; p.out = out; p.lse = lse; p.prof = prof;
  const size_t lds = (size_t)HPW * 2 * AT * LS * sizeof(float);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_mfma_kernel<HPW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(attn_fwd_mfma_kernel<HPW>, dim3(nwg), dim3(256 * HPW), lds, 0, p);
  CK(hipDeviceSynchronize());
  std::vector<long long> h((size_t)nwg * nwave * 8);
  CK(hipMemcpy(h.data(), prof, h.size() * 8, hipMemcpyDeviceToHost));
  long long first = 1LL << 62, last = 0;
  std::vector<double> ph[4][6];
  for (size_t i = 0; i < h.size(); i += 8) {
    const long long* s = &h[i];
    first = std::min(first, s[0]);
    const int qt = (int)s[7];
    long long end = s[6] ? s[6] : s[2];
    last = std::max(last, end);
    if (s[6]) for (int k = 0; k < 6; ++k) ph[qt][k].push_back((s[k + 1] - s[k]) * 0.01);
  }
  auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
  printf("HPW %d p %.1f: kernel span %.2f us\n", HPW, pdrop, (last - first) * 0.01);
  for (int qt = 3; qt >= 0; --qt)
    printf("  query tile %d (%d key tiles): loads+LDS write %.2f, barrier %.2f, S tiles+max %.2f, exp+sum %.2f, dropout+PV %.2f, store %.2f us\n",
           qt, qt + 1, med(ph[qt][0]), med(ph[qt][1]), med(ph[qt][2]), med(ph[qt][3]), med(ph[qt][4]), med(ph[qt][5]));
}

int main() {
  run<1>(0.f);
  run<2>(0.f);
  run<2>(0.2f);
  return 0;
}
