// Diagnostic build of the MX-fp8 GEMM with s_memtime stamps at the phase boundaries (gemm_mx8.hip compiled with
// -DMX8_STAMP): prints where one workgroup's two wave groups spend a k-step.  Read the SHARES, not the length.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../tap-clip_amd/csrc/kernels.h"
using namespace tapclip;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
int main(int argc, char** argv) {
  const int64_t M = 50432;
  const int N = argc > 1 ? atoi(argv[1]) : 2304, K = argc > 2 ? atoi(argv[2]) : 768;
  const int64_t m_pad = M;
  uint8_t *A, *W, *As, *Ws; float* bias; bf16_t* out; unsigned long long* st;
  CK(hipMalloc(&A, (size_t)M * K)); CK(hipMalloc(&W, (size_t)N * K)); CK(hipMalloc(&As, (size_t)(K / 64) * m_pad * 2)); CK(hipMalloc(&Ws, (size_t)(K / 64) * N * 2));
  CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&out, (size_t)M * N * 2)); CK(hipMalloc(&st, 2 * 48 * 8 * 8));
  CK(hipMemset(A, 0x38, (size_t)M * K)); CK(hipMemset(W, 0x30, (size_t)N * K)); CK(hipMemset(As, 120, (size_t)(K / 64) * m_pad * 2));
  CK(hipMemset(Ws, 120, (size_t)(K / 64) * N * 2)); CK(hipMemset(bias, 0, N * 4)); CK(hipMemset(st, 0, 2 * 48 * 8 * 8));
  Mx8GemmArgs g;
  g.A = A; g.A_scale = As; g.lda = K; g.m_pad = m_pad; g.W = W; g.W_scale = Ws; g.bias = bias; g.M = M; g.N = N; g.K = K;
  g.out_bf16 = out; g.ldo = N; g.stamps = st;
  for (int i = 0; i < 3; ++i) CK(launch_gemm_mx8(g, EPI_BIAS_BF16, 0));
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(2 * 48 * 8);
  CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
  const char* seg[] = {"READ: epilogue/bias + issue ds_reads", "wait (B: dma) + lgkmcnt(0)", "barrier 1", "COMPUTE: 8 MFMA + DMA issue", "wait (A: dma)", "barrier 2"};
  const int KS = K / 64;
  for (int grp = 0; grp < 2; ++grp) {
    printf("group %c (k-steps of tiles 2.. of one workgroup; cycles of the 100 MHz s_memtime clock x 24 ~ shader clocks)\n", grp ? 'B' : 'A');
    double sum[6] = {0, 0, 0, 0, 0, 0}, sum0[6] = {0, 0, 0, 0, 0, 0};
    int n = 0, n0 = 0;
    for (int k = KS; k < 47; ++k) {  // skip the first tile
      const unsigned long long* t = &h[(grp * 48 + k) * 8];
      const bool first = k % KS == 0;
      for (int s2 = 0; s2 < 6; ++s2) (first ? sum0 : sum)[s2] += (double)(uint32_t)((uint32_t)t[s2 + 1] - (uint32_t)t[s2]);
      (first ? n0 : n)++;
    }
    double tot = 0, tot0 = 0;
    for (int s2 = 0; s2 < 6; ++s2) { tot += sum[s2] / n; tot0 += sum0[s2] / n0; }
    for (int s2 = 0; s2 < 6; ++s2) printf("  %-40s steady %7.1f (%4.1f %%)   first step of a tile %7.1f\n", seg[s2], sum[s2] / n, 100 * sum[s2] / n / tot, sum0[s2] / n0);
    printf("  total per k-step: steady %.1f, first %.1f  (stamp ticks)\n", tot, tot0);
  }
  // raw ticks of one step, to calibrate the clock
  printf("raw: %llu %llu %llu\n", h[8 * 20], h[8 * 20 + 3], h[8 * 21]);
  return 0;
}
