// Correctness (sampled outputs vs a host double-precision evaluation of the same quantised operands) and
// timing of the MX-fp8 GEMM at the image tower's shapes.  Measurement aid; not part of the product path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../tap-clip_amd/csrc/kernels.h"
using namespace tapclip;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static float e4m3_decode(uint8_t c) {
  const int s = c >> 7, e = (c >> 3) & 15, m = c & 7;
  float v = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -v : v;
}
static float bf2f_host(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 10;
  const int64_t M = argc > 2 ? atoll(argv[2]) : 50432;
  const int64_t m_pad = (M + 7) / 8 * 8;
  const int KMAX = 3072, NMAX = 3072;
  std::vector<uint8_t> hA((size_t)M * KMAX), hW((size_t)NMAX * KMAX), hAs((size_t)(KMAX / 64) * m_pad * 2), hWs((size_t)(KMAX / 64) * NMAX * 2);
  uint64_t s = 99;
  auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33); };
  auto rnd_e4m3 = [&]() { uint8_t c = (uint8_t)rnd(); if ((c & 0x7f) == 0x7f) c ^= 1; return c; };  // no NaN codes
  for (auto& v : hA) v = rnd_e4m3();
  for (auto& v : hW) v = rnd_e4m3();
  for (auto& v : hAs) v = (uint8_t)(127 - 8 + rnd() % 5);
  for (auto& v : hWs) v = (uint8_t)(127 - 10 + rnd() % 5);
  std::vector<float> hbias(NMAX);
  for (auto& v : hbias) v = (float)((int)(rnd() % 200) - 100) * 0.01f;
  uint8_t *A, *W, *As, *Ws;
  float* bias;
  bf16_t* out;
  CK(hipMalloc(&A, hA.size())); CK(hipMalloc(&W, hW.size())); CK(hipMalloc(&As, hAs.size())); CK(hipMalloc(&Ws, hWs.size()));
  CK(hipMalloc(&bias, NMAX * 4)); CK(hipMalloc(&out, (size_t)M * NMAX * 2));
  CK(hipMemcpy(A, hA.data(), hA.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(W, hW.data(), hW.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(bias, hbias.data(), NMAX * 4, hipMemcpyHostToDevice));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  struct Shape { const char* name; int N, K; };
  const Shape shapes[] = {{"qkv      N2304 K768 ", 2304, 768}, {"out_proj N768  K768 ", 768, 768}, {"fc       N3072 K768 ", 3072, 768},
                          {"proj     N768  K3072", 768, 3072}, {"qkv-like N2304 K3072", 2304, 3072},
                          {"qkv-like N2304 K1536", 2304, 1536}, {"qkv-like N2304 K2304", 2304, 2304}, {"qkv-like N2304 K384 ", 2304, 384}};
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (const Shape& sh : shapes) {
    // scale arrays are laid out for THIS K (k-step major)
    const int ksn = sh.K / 64;
    CK(hipMemcpy(As, hAs.data(), (size_t)ksn * m_pad * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(Ws, hWs.data(), (size_t)ksn * sh.N * 2, hipMemcpyHostToDevice));
    Mx8GemmArgs g;
    g.A = A; g.A_scale = As; g.lda = sh.K; g.m_pad = m_pad;
    g.W = W; g.W_scale = Ws; g.bias = bias;
    g.M = M; g.N = sh.N; g.K = sh.K; g.out_bf16 = out; g.ldo = sh.N;
    CK(hipMemsetAsync(out, 0xff, (size_t)M * sh.N * 2, st));
    CK(launch_gemm_mx8(g, EPI_BIAS_BF16, st));
    CK(hipStreamSynchronize(st));
    std::vector<uint16_t> ho((size_t)M * sh.N);
    CK(hipMemcpy(ho.data(), out, ho.size() * 2, hipMemcpyDeviceToHost));
    // sampled check (A row stride = K of this shape; W row stride = K)
    double worst = 0;
    int bad = 0;
    for (int t = 0; t < 4000; ++t) {
      const int64_t m = t < 64 ? (t < 32 ? t : M - 1 - (t - 32)) : rnd() % M;
      const int n = t < 64 ? (t * 37) % sh.N : rnd() % sh.N;
      double acc = hbias[n];
      for (int k = 0; k < sh.K; ++k) {
        const int b = k / 32;
        const double sa = ldexp(1.0, hAs[((size_t)(b >> 1) * m_pad + m) * 2 + (b & 1)] - 127);
        const double sw = ldexp(1.0, hWs[((size_t)(b >> 1) * sh.N + n) * 2 + (b & 1)] - 127);
        acc += (double)e4m3_decode(hA[m * sh.K + k]) * sa * (double)e4m3_decode(hW[(size_t)n * sh.K + k]) * sw;
      }
      const double got = bf2f_host(ho[m * sh.N + n]);
      const double err = fabs(got - acc) / (fabs(acc) + 1.0);
      if (err > worst) worst = err;
      if (err > 8e-3) ++bad;  // bf16 output rounding is 4e-3 relative
    }
    for (int w = 0; w < 20; ++w) CK(launch_gemm_mx8(g, EPI_BIAS_BF16, st));
    std::vector<float> t;
    for (int round = 0; round < iters; ++round) {
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < 4; ++i) CK(launch_gemm_mx8(g, EPI_BIAS_BF16, st));
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      t.push_back(ms / 4);
    }
    std::sort(t.begin(), t.end());
    const double med = 1e3 * t[t.size() / 2];
    printf("mx8 %s M%lld: sampled worst rel err %.2e, %d bad of 4000 | median %8.1f us %7.1f TFLOP/s  min %8.1f us\n", sh.name, (long long)M, worst,
           bad, med, 2.0 * M * sh.N * sh.K / (med * 1e-6) / 1e12, 1e3 * t[0]);
    fflush(stdout);
  }
  return 0;
}
