// Where does one fused LSTM forward step (cfg2: B = 64, H = 1024) spend its time?  Builds the production kernel
// (bayeslms_amd/csrc/lstm_step.hip) with -DBLM_LSTM_PROF: every wave records wall-clock stamps (s_memrealtime, 10 ns)
// at kernel entry / first chunk landed / K loop done / after the reduction barrier / after the second barrier / end,
// over a DEPENDENT chain of launches.  Prints per-phase medians and the gap between launches.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DBLM_LSTM_PROF -I include -I bayeslms_amd/csrc -o tools/lstm_step_prof \
//         tools/lstm_step_prof.hip bayeslms_amd/csrc/capi.hip
#include "../bayeslms_amd/csrc/lstm_step.hip"

#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int RING, bool REFILL, int NW, bool PIPE = false, bool TAIL = true>
static void run(const char* name, int B, int H, int alias = 0) {
  const int steps = 64;
  float *xw, *w, *h[2], *c[2], *ga;
  CK(hipMalloc(&xw, (size_t)B * 4 * H * 4));
  CK(hipMalloc(&w, (size_t)4 * H * H * 4));
  CK(hipMalloc(&ga, (size_t)B * 4 * H * 4));
  for (int i = 0; i < 2; ++i) { CK(hipMalloc(&h[i], (size_t)B * H * 4)); CK(hipMalloc(&c[i], (size_t)B * H * 4)); CK(hipMemset(h[i], 0, (size_t)B * H * 4)); CK(hipMemset(c[i], 0, (size_t)B * H * 4)); }
  CK(hipMemset(xw, 0, (size_t)B * 4 * H * 4));
  CK(hipMemset(w, 0, (size_t)4 * H * H * 4));
  const int nwg = (H / 8) * ((B + 31) / 32);
  long long* prof;
  const size_t per = (size_t)nwg * NW * 8;
  CK(hipMalloc(&prof, per * steps * 8));
  CK(hipMemset(prof, 0, per * steps * 8));
  const size_t lds = (size_t)NW * WAVE_LDS * sizeof(float);
  auto kern = lstm_step_fwd_kernel<RING, 4, REFILL, NW, PIPE, TAIL>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  for (int rep = 0; rep < 2; ++rep)
    for (int t = 0; t < steps; ++t) {
      LstmStepP p{xw, w, h[t & 1], c[t & 1], h[1 - (t & 1)], c[1 - (t & 1)], ga, nullptr, nullptr, nullptr, -1, nullptr, B, H, nullptr, prof + per * t, alias};
      hipLaunchKernelGGL(kern, dim3(H / 8, (B + 31) / 32), dim3(64 * NW), lds, st, p);
    }
  CK(hipStreamSynchronize(st));
  std::vector<long long> hp(per * steps);
  CK(hipMemcpy(hp.data(), prof, hp.size() * 8, hipMemcpyDeviceToHost));
  auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  // per step: first entry, last end over all waves; per wave: phase durations
  std::vector<double> span, gap, ph[5], entry_spread, mhz;
  double prev_end = 0;
  for (int t = 8; t < steps; ++t) {
    long long first = 1LL << 62, last = 0, last_entry = 0;
    for (size_t i = 0; i < per; i += 8) {
      const long long* s = &hp[per * t + i];
      first = std::min(first, s[0]);
      last_entry = std::max(last_entry, s[0]);
      last = std::max(last, s[5]);
      for (int k = 0; k < 5; ++k) ph[k].push_back((s[k + 1] - s[k]) * 0.01);
      if (s[2] > s[1]) mhz.push_back((double)(s[7] - s[6]) / ((s[2] - s[1]) * 0.01));
    }
    span.push_back((last - first) * 0.01);
    entry_spread.push_back((last_entry - first) * 0.01);
    if (prev_end > 0) gap.push_back(first * 0.01 - prev_end);
    prev_end = last * 0.01;
  }
  printf("%s: span first-entry..last-end %.2f us, launch gap (prev last-end .. first entry) %.2f us, entry spread %.2f us | per wave medians: "
         "entry->first chunk landed %.2f, ->K loop done %.2f, ->barrier1 %.2f, ->barrier2 %.2f, ->end %.2f us; shader clock in the K loop %.0f MHz\n",
         name, med(span), med(gap), med(entry_spread), med(ph[0]), med(ph[1]), med(ph[2]), med(ph[3]), med(ph[4]), med(mhz));
}

int main() {
  run<2, true, 4>("4 waves ring 2       ", 64, 1024);
  run<2, true, 4, true>("4 waves ring 2 PIPED ", 64, 1024);
  run<2, false, 8>("8 waves ring 2 all   ", 64, 1024);
  run<2, true, 4>("4 waves ring 2, W aliased (L2-hot)      ", 64, 1024, 1);
  run<2, true, 4>("4 waves ring 2, W and h aliased (L2-hot)", 64, 1024, 2);
  // the production form (pipelined, no-tail), and the upper bound of hiding the W fetch: W loads return 0 without memory access
  run<2, true, 4, true, false>("PIPED no-tail (production)              ", 64, 1024);
  run<2, true, 4, true, false>("PIPED no-tail, W loads cost nothing     ", 64, 1024, 3);
  run<2, true, 4, true, false>("PIPED no-tail, B = 32                   ", 32, 1024);
  run<2, true, 4, true, false>("PIPED no-tail, B = 32, W loads free     ", 32, 1024, 3);
  return 0;
}
