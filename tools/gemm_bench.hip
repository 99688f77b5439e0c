// Micro-benchmark of the MFMA GEMM kernels at the image tower's shapes (development tool; not part
// of the library).  Build: make -C tools.  Usage: tools/gemm_bench [iters]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <vector>

#include "../tap-clip_amd/csrc/kernels.h"

using namespace tapclip;

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

static uint16_t f2bf_host(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7FFF + ((u >> 16) & 1);
  return (uint16_t)(u >> 16);
}

static bf16_t* rand_bf16(size_t n, uint64_t seed, float scale) {
  std::vector<uint16_t> h(n);
  uint64_t s = seed * 0x9E3779B97F4A7C15ull + 12345;
  for (size_t i = 0; i < n; ++i) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    float u = (float)((s >> 40) & 0xFFFFFF) / 8388608.0f - 1.0f;  // [-1, 1)
    h[i] = f2bf_host(u * scale);
  }
  bf16_t* d;
  CK(hipMalloc(&d, n * 2));
  CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
  return d;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20;
  const int64_t M = argc > 2 ? atoll(argv[2]) : 50432;
  struct Shape { const char* name; int N, K, epi; };
  const bool text = argc > 3 && std::string(argv[3]) == "text";  // the text tower's shapes (run with M = 6045)
  const Shape text_shapes[] = {
      {"t_qkv    N1536 K512  bias->bf16 ", 1536, 512, EPI_BIAS_BF16},
      {"t_out    N512  K512  bias->bf16 ", 512, 512, EPI_BIAS_BF16},
      {"t_fc     N2048 K512  gelu->bf16 ", 2048, 512, EPI_BIAS_GELU_BF16},
      {"t_proj   N512  K2048 bias->bf16 ", 512, 2048, EPI_BIAS_BF16},
      {"t_qkv-like N1536 K1024          ", 1536, 1024, EPI_BIAS_BF16},  // slope/intercept probes
      {"t_qkv-like N1536 K2048          ", 1536, 2048, EPI_BIAS_BF16},
      {"t_qkv-like N1536 K128           ", 1536, 128, EPI_BIAS_BF16},
      {"t_out-like N512  K128           ", 512, 128, EPI_BIAS_BF16},
      {"t_out-like N512  K1024          ", 512, 1024, EPI_BIAS_BF16},
  };
  const Shape image_shapes[] = {
      {"qkv      N2304 K768  bias->bf16 ", 2304, 768, EPI_BIAS_BF16},
      {"out_proj N768  K768  bias->bf16 ", 768, 768, EPI_BIAS_BF16},
      {"fc_gelu  N3072 K768  gelu->bf16 ", 3072, 768, EPI_BIAS_GELU_BF16},
      {"proj     N768  K3072 bias->bf16 ", 768, 3072, EPI_BIAS_BF16},
      {"fc_ident N3072 K768  noact->bf16", 3072, 768, EPI_BIAS_GELU_BF16 + 700},
      {"fc_quick N3072 K768  quick->bf16", 3072, 768, EPI_BIAS_GELU_BF16 + 100},
      {"fc_tr    N3072 K768  bias->bf16 ", 3072, 768, EPI_BIAS_BF16},
      {"qkv-like N2304 K3072 bias->bf16 ", 2304, 3072, EPI_BIAS_BF16},   // slope/intercept probe
      {"qkv-like N2304 K1536 bias->bf16 ", 2304, 1536, EPI_BIAS_BF16},
  };
  const Shape* all_shapes = text ? text_shapes : image_shapes;
  // argv[4] / $GEMM_BENCH_ONLY: only the shapes whose label contains this substring (profiling one kernel family)
  const char* only = argc > 4 ? argv[4] : getenv("GEMM_BENCH_ONLY");
  std::vector<Shape> picked;
  for (int k = 0; k < (text ? (int)(sizeof(text_shapes) / sizeof(text_shapes[0])) : (int)(sizeof(image_shapes) / sizeof(image_shapes[0]))); ++k)
    if (!only || strstr(all_shapes[k].name, only)) picked.push_back(all_shapes[k]);
  if (picked.empty()) { printf("no shape matches %s\n", only); return 1; }
  const Shape* shapes = picked.data();
  const int lda_pad = getenv("GEMM_BENCH_LDA_PAD") ? atoi(getenv("GEMM_BENCH_LDA_PAD")) : 0;  // elements added to A's row stride
  bf16_t* A = rand_bf16((size_t)M * (3072 + lda_pad), 1, 1.0f);
  bf16_t* W = rand_bf16((size_t)3072 * 3072, 2, 0.03f);
  float* bias;
  CK(hipMalloc(&bias, 3072 * 4));
  CK(hipMemset(bias, 0, 3072 * 4));
  bf16_t* obf;
  CK(hipMalloc(&obf, (size_t)M * 3072 * 2));
  float* of32;
  CK(hipMalloc(&of32, (size_t)M * 768 * 4));
  CK(hipMemset(of32, 0, (size_t)M * 768 * 4));
  float* split_ws;
  CK(hipMalloc(&split_ws, gemm256_split_ws_bytes()));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  // interleaved rounds in ONE process (guide rule 24): every round times each shape once; report median / min
  const int n_shapes = (int)picked.size();
  std::vector<std::vector<float>> times(n_shapes);
  auto run = [&](const Shape& sh, int reps) {
    GemmArgs g;
    g.A_hi = A; g.A_lo = nullptr; g.lda = sh.K + lda_pad;
    g.W_hi = W; g.W_lo = nullptr;
    g.bias = bias;
    g.M = M; g.N = sh.N; g.K = sh.K;
    g.out_hi = obf; g.out_lo = nullptr; g.out_f32 = of32; g.ldo = sh.N;
    g.add_table = nullptr; g.rows_per_group = 0; g.act = sh.epi / 100;
    g.split_ws = split_ws;
    const int epi = sh.epi % 100;
    for (int i = 0; i < reps; ++i) CK(launch_gemm(g, epi, false, s));
  };
  for (int w = 0; w < 30; ++w) run(shapes[0], 10);  // ~100 ms of warm-up: clocks settle
  CK(hipStreamSynchronize(s));
  for (int round = 0; round < iters; ++round) {
    for (int k = 0; k < n_shapes; ++k) {
      run(shapes[k], 1);  // untimed: same-shape cache state
      CK(hipEventRecord(e0, s));
      run(shapes[k], 4);
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      times[k].push_back(ms / 4);
    }
  }
  // ---- attention core at the image tower's shape (buffers are sized for M = 50 432: not in the text mode)
  const bool with_attn = !only || strstr("attention", only);
  if (!text && M >= 64 * 577 && with_attn) {
    AttnArgs a;
    a.qkv_hi = A; a.qkv_lo = nullptr; a.out_hi = obf; a.out_lo = nullptr; a.probs = nullptr;
    a.n_seq = (int)(M / 197); a.T = 197; a.H = 12; a.D = 768; a.causal = 0;
    std::vector<float> t;
    for (int round = 0; round < iters; ++round) {
      CK(launch_attention(a, false, s));
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < 4; ++i) CK(launch_attention(a, false, s));
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      t.push_back(ms / 4);
    }
    std::sort(t.begin(), t.end());
    printf("attention n%d T197 H12: median %8.1f us  min %8.1f us\n", a.n_seq, 1e3 * t[t.size() / 2], 1e3 * t[0]);
  }
  if (!text && M >= 64 * 577 && with_attn) {  // ViT-L/14@336 attention: 577 tokens, 16 heads (flash-style kernel)
    AttnArgs a;
    a.qkv_hi = A; a.qkv_lo = nullptr; a.out_hi = obf; a.out_lo = nullptr; a.probs = nullptr;
    a.n_seq = 64; a.T = 577; a.H = 16; a.D = 1024; a.causal = 0;
    std::vector<float> t;
    for (int round = 0; round < iters; ++round) {
      CK(launch_attention(a, false, s));
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < 4; ++i) CK(launch_attention(a, false, s));
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      t.push_back(ms / 4);
    }
    std::sort(t.begin(), t.end());
    const double fl = 4.0 * a.n_seq * a.H * 577.0 * 577.0 * 64.0;
    printf("attention n%d T577 H16: median %8.1f us  %6.1f TFLOP/s\n", a.n_seq, 1e3 * t[t.size() / 2], fl / (t[t.size() / 2] * 1e-3) / 1e12);
  }
  for (int k = 0; k < n_shapes; ++k) {
    std::sort(times[k].begin(), times[k].end());
    const double med = 1e3 * times[k][times[k].size() / 2], mn = 1e3 * times[k][0];
    const double fl = 2.0 * M * shapes[k].N * shapes[k].K;
    printf("%s M%lld: median %8.1f us %7.1f TFLOP/s   min %8.1f us %7.1f TFLOP/s\n", shapes[k].name, (long long)M, med,
           fl / (med * 1e-6) / 1e12, mn, fl / (mn * 1e-6) / 1e12);
  }
  return 0;
}
