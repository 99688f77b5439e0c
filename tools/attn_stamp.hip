// Diagnostic: per-workgroup phase stamps of the attention kernel at the image tower's shape (n = 256, T = 197, H = 12).
// Build: make -C tools attn_stamp (compiles attention.hip with -DATTN_STAMP).  Prints the distribution of the phase
// durations, the number of workgroups resident per CU and the chip-wide timeline.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#include "../tap-clip_amd/csrc/kernels.h"
using namespace tapclip;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 256, T = argc > 2 ? atoi(argv[2]) : 197, H = argc > 3 ? atoi(argv[3]) : 12, D = H * 64;
  const size_t M = (size_t)n * T;
  std::vector<uint16_t> h(M * 3 * D);
  uint64_t s = 12345;
  for (auto& v : h) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = (uint16_t)(0x3C00 + ((s >> 40) & 0x3FF)) ^ (uint16_t)((s >> 33) & 0x8000); }
  bf16_t *qkv, *out;
  CK(hipMalloc(&qkv, h.size() * 2)); CK(hipMemcpy(qkv, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&out, M * D * 2));
  const int nwg = n * H;
  unsigned long long* st;
  CK(hipMalloc(&st, (size_t)nwg * 8 * 8));
  AttnArgs a;
  a.qkv_hi = qkv; a.qkv_lo = nullptr; a.out_hi = out; a.out_lo = nullptr; a.probs = nullptr;
  a.n_seq = n; a.T = T; a.H = H; a.D = D; a.causal = 0;
  hipStream_t stream; CK(hipStreamCreate(&stream));
  for (int i = 0; i < 20; ++i) CK(launch_attention(a, false, stream));
  CK(hipStreamSynchronize(stream));
  a.stamps = st;
  CK(hipMemset(st, 0, (size_t)nwg * 64));
  CK(launch_attention(a, false, stream));
  CK(hipStreamSynchronize(stream));
  std::vector<unsigned long long> v((size_t)nwg * 8);
  CK(hipMemcpy(v.data(), st, v.size() * 8, hipMemcpyDeviceToHost));
  unsigned long long t0 = ~0ull, t1 = 0;
  for (int b = 0; b < nwg; ++b) { t0 = std::min(t0, v[b * 8]); t1 = std::max(t1, v[b * 8 + 6]); }
  printf("kernel span %.1f us, %d workgroups\n", (t1 - t0) * 0.01, nwg);
  auto pct = [](std::vector<double> x, double p) { std::sort(x.begin(), x.end()); return x[(size_t)(p * (x.size() - 1))]; };
  const char* names[] = {"start->own loads landed", "..->staged (barrier)", "..->wave0 tile0 done", "..->wave0 tile1 done", "..->wave0 stores complete", "..->all waves done", "whole workgroup"};
  for (int k = 0; k < 7; ++k) {
    std::vector<double> d;
    for (int b = 0; b < nwg; ++b) {
      const unsigned long long* w = &v[b * 8];
      double x = k == 6 ? (w[6] - w[0]) : (w[k + 1] - w[k]);
      d.push_back(x * 0.01);
    }
    printf("%-28s median %6.2f us  p10 %6.2f  p90 %6.2f\n", names[k], pct(d, 0.5), pct(d, 0.1), pct(d, 0.9));
  }
  // residency: how many workgroups overlap in time on one (xcc, se, cu)
  std::map<unsigned long long, std::vector<std::pair<unsigned long long, int>>> ev;
  for (int b = 0; b < nwg; ++b) {
    const unsigned long long id = v[b * 8 + 7];
    const unsigned hw = (unsigned)id, xcc = (unsigned)(id >> 32);
    const unsigned long long cu = ((unsigned long long)xcc << 16) | (((hw >> 13) & 7) << 8) | ((hw >> 8) & 15);  // SE_ID[15:13], CU_ID[11:8]
    ev[cu].push_back({v[b * 8], +1});
    ev[cu].push_back({v[b * 8 + 6], -1});
  }
  std::map<int, int> hist;
  for (auto& kv : ev) {
    std::sort(kv.second.begin(), kv.second.end());
    int cur = 0, mx = 0;
    for (auto& e : kv.second) { cur += e.second; mx = std::max(mx, cur); }
    hist[mx]++;
  }
  printf("distinct CUs seen: %zu; max co-resident workgroups per CU:", ev.size());
  for (auto& kv : hist) printf("  %d WGs on %d CUs", kv.first, kv.second);
  printf("\n");
  // chip-wide timeline: workgroups in each phase, sampled every 2 us
  for (unsigned long long t = t0; t < t1; t += 200) {
    int ph[4] = {0, 0, 0, 0};
    for (int b = 0; b < nwg; ++b) {
      const unsigned long long* w = &v[b * 8];
      if (t < w[0] || t >= w[6]) continue;
      if (t < w[2]) ph[0]++; else if (t < w[5]) ph[1]++; else ph[2]++;
    }
    printf("t=%5.1f us: loading %4d  computing/storing %4d  draining %4d\n", (t - t0) * 0.01, ph[0], ph[1], ph[2]);
  }
  return 0;
}
