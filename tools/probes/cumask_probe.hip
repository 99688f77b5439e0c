// CU-masked streams on MI355X: which XCDs a mask selects, and what the tower's kernel families cost on a PART of the chip,
// alone and side by side (GEMM on one partition, the HBM-bound LayerNorm / attention kernels on the other).
// Measurement aid (tools/README.md); not part of the product path.   usage: cumask_probe [iters]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../tap-clip_amd/csrc/kernels.h"
using namespace tapclip;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void census_kernel(unsigned* out) {
  unsigned xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  if (threadIdx.x == 0) out[blockIdx.x] = ((xcc & 15) << 16) | (hw & 0xFFFF);
  // stay resident a little so that the blocks spread over every CU the mask allows
  for (int i = 0; i < 2000; ++i) asm volatile("s_nop 15");
}

static hipStream_t masked_stream(const std::vector<uint32_t>& m) {
  hipStream_t s;
  CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)m.size(), m.data()));
  return s;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 10;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  printf("device CUs %d\n", ncu);
  unsigned* d_out;
  CK(hipMalloc(&d_out, 4096 * 4));
  std::vector<unsigned> h(4096);
  auto census = [&](const char* tag, const std::vector<uint32_t>& mask) {
    hipStream_t s = masked_stream(mask);
    hipLaunchKernelGGL(census_kernel, dim3(2048), dim3(64), 0, s, d_out);
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h.data(), d_out, 2048 * 4, hipMemcpyDeviceToHost));
    int per_xcc[16] = {0};
    std::vector<unsigned> cus;
    for (int i = 0; i < 2048; ++i) {
      per_xcc[(h[i] >> 16) & 15]++;
      cus.push_back(((h[i] >> 16) << 16) | ((h[i] >> 8) & 0xFF));  // xcc | se/sh/cu bits
    }
    std::sort(cus.begin(), cus.end());
    const int distinct = (int)(std::unique(cus.begin(), cus.end()) - cus.begin());
    printf("%-28s blocks per XCC:", tag);
    for (int x = 0; x < 8; ++x) printf(" %4d", per_xcc[x]);
    printf("   distinct CUs seen %d\n", distinct);
    CK(hipStreamDestroy(s));
  };
  const int words = (ncu + 31) / 32;
  std::vector<uint32_t> all(words, 0xFFFFFFFFu), first64(words, 0), every8th(words, 0), lo192(words, 0), hi64(words, 0);
  first64[0] = first64[1] = 0xFFFFFFFFu;
  for (int i = 0; i < ncu; ++i) {
    if (i % 8 >= 6) hi64[i / 32] |= 1u << (i % 32);   // guess A: CU index interleaves the XCDs (cu % 8 = XCD)
    else lo192[i / 32] |= 1u << (i % 32);
    if (i % 8 == 0) every8th[i / 32] |= 1u << (i % 32);
  }
  census("all", all);
  census("first 64 bits", first64);
  census("bits with i % 8 == 0", every8th);
  census("bits with i % 8 < 6", lo192);
  census("bits with i % 8 >= 6", hi64);

  // ---- kernel families on partitions.  Partition masks are chosen from the census above by argv[2]:
  //   "interleave" (bit i -> XCD i % 8) or "block" (bit i -> XCD i / 32)
  const bool interleave = !(argc > 2 && !strcmp(argv[2], "block"));
  std::vector<uint32_t> mG(words, 0), mL(words, 0);
  for (int i = 0; i < ncu; ++i) {
    const int xcd = interleave ? i % 8 : i / 32;
    (xcd < 6 ? mG : mL)[i / 32] |= 1u << (i % 32);
  }
  hipStream_t sG = masked_stream(mG), sL = masked_stream(mL), sAll;
  CK(hipStreamCreate(&sAll));
  const int64_t M = 50432;
  const int D = 768;
  bf16_t *A, *W, *O, *qkv, *ao, *dlt, *xn;
  float *bias, *x, *split_ws;
  CK(hipMalloc(&A, (size_t)M * 3072 * 2)); CK(hipMemset(A, 0x11, (size_t)M * 3072 * 2));
  CK(hipMalloc(&W, (size_t)3072 * 3072 * 2)); CK(hipMemset(W, 0x22, (size_t)3072 * 3072 * 2));
  CK(hipMalloc(&O, (size_t)M * 3072 * 2));
  CK(hipMalloc(&qkv, (size_t)M * 3 * D * 2)); CK(hipMemset(qkv, 0x33, (size_t)M * 3 * D * 2));
  CK(hipMalloc(&ao, (size_t)M * D * 2));
  CK(hipMalloc(&dlt, (size_t)M * D * 2)); CK(hipMemset(dlt, 0, (size_t)M * D * 2));
  CK(hipMalloc(&xn, (size_t)M * D * 2));
  CK(hipMalloc(&x, (size_t)M * D * 4)); CK(hipMemset(x, 0, (size_t)M * D * 4));
  CK(hipMalloc(&bias, 3072 * 4)); CK(hipMemset(bias, 0, 3072 * 4));
  CK(hipMalloc(&split_ws, gemm256_split_ws_bytes()));
  float* gam; CK(hipMalloc(&gam, D * 4)); CK(hipMemset(gam, 0, D * 4));

  auto gemm = [&](hipStream_t s, int n_cu, int N, int K, int epi) {
    GemmArgs g;
    g.A_hi = A; g.A_lo = nullptr; g.lda = K; g.W_hi = W; g.W_lo = nullptr; g.bias = bias;
    g.M = M; g.N = N; g.K = K; g.out_hi = O; g.out_lo = nullptr; g.out_f32 = nullptr; g.ldo = N;
    g.add_table = nullptr; g.rows_per_group = 0; g.act = 0; g.split_ws = split_ws; g.n_cu = n_cu;
    CK(launch_gemm(g, epi, false, s));
  };
  auto block_gemms = [&](hipStream_t s, int n_cu) {  // the four GEMMs of one ViT-B/16 block
    gemm(s, n_cu, 2304, 768, EPI_BIAS_BF16);
    gemm(s, n_cu, 768, 768, EPI_BIAS_BF16);
    gemm(s, n_cu, 3072, 768, EPI_BIAS_GELU_BF16);
    gemm(s, n_cu, 768, 3072, EPI_BIAS_BF16);
  };
  auto block_mem = [&](hipStream_t s) {  // the HBM-bound kernels of one block: LN1-like, attention, LN2-like
    CK(launch_add_layernorm_ex(3, x, dlt, nullptr, dlt, nullptr, gam, gam, M, D, xn, nullptr, s, true));
    AttnArgs a;
    a.qkv_hi = qkv; a.qkv_lo = nullptr; a.out_hi = ao; a.out_lo = nullptr; a.probs = nullptr;
    a.n_seq = 256; a.T = 197; a.H = 12; a.D = D; a.causal = 0;
    CK(launch_attention(a, false, s));
    CK(launch_add_layernorm_ex(2, x, dlt, nullptr, nullptr, nullptr, gam, gam, M, D, xn, nullptr, s, true));
  };
  hipEvent_t e0, e1, f0, f1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
  auto time1 = [&](const char* tag, hipStream_t s, auto fn) {
    for (int i = 0; i < 3; ++i) fn();
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) fn();
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-46s %8.1f us per block\n", tag, 1e3 * ms / iters);
    return ms / iters;
  };
  time1("GEMMs of a block, whole chip (256 CUs)", sAll, [&] { block_gemms(sAll, 0); });
  time1("mem-bound kernels of a block, whole chip", sAll, [&] { block_mem(sAll); });
  time1("GEMMs of a block, 6 XCDs (192 CUs), alone", sG, [&] { block_gemms(sG, 192); });
  time1("mem-bound kernels, 2 XCDs (64 CUs), alone", sL, [&] { block_mem(sL); });
  // side by side: both partitions busy for the same number of blocks
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, sG)); CK(hipEventRecord(f0, sL));
    for (int i = 0; i < iters; ++i) { block_gemms(sG, 192); block_mem(sL); }
    CK(hipEventRecord(e1, sG)); CK(hipEventRecord(f1, sL));
    CK(hipEventSynchronize(e1)); CK(hipEventSynchronize(f1));
    float mg, ml; CK(hipEventElapsedTime(&mg, e0, e1)); CK(hipEventElapsedTime(&ml, f0, f1));
    printf("side by side: GEMMs on 6 XCDs %8.1f us per block | mem-bound on 2 XCDs %8.1f us per block\n", 1e3 * mg / iters, 1e3 * ml / iters);
  }
  return 0;
}
