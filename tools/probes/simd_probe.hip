// Which SIMD does wave w of a 512-thread (8-wave) workgroup run on?  HW_REG_HW_ID bits [5:4] = SIMD_ID.
// The staggered-halves GEMMs want one wave of each phase group on every SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void probe(unsigned* out) {
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
}
int main() {
  const int nb = 512;
  unsigned* d;
  hipMalloc(&d, nb * 8 * 4);
  hipLaunchKernelGGL(probe, dim3(nb), dim3(512), 96 * 1024, 0, d);  // big LDS: one workgroup per CU at a time
  unsigned h[nb * 8];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int pattern_count[256] = {0};
  for (int b = 0; b < nb; ++b) {
    int key = 0;
    for (int w = 0; w < 8; ++w) key |= (((h[b * 8 + w] >> 4) & 3)) << (2 * w);
    if (b < 6) {
      printf("block %d: simd of waves 0..7 =", b);
      for (int w = 0; w < 8; ++w) printf(" %u", (h[b * 8 + w] >> 4) & 3);
      printf("   (cu %u se %u)\n", (h[b * 8] >> 8) & 15, (h[b * 8] >> 13) & 7);
    }
    // classify: do waves w and w+4 share a SIMD?
    int same4 = 0, same1 = 0;
    for (int w = 0; w < 4; ++w) same4 += ((key >> (2 * w)) & 3) == ((key >> (2 * (w + 4))) & 3);
    for (int w = 0; w < 8; w += 2) same1 += ((key >> (2 * w)) & 3) == ((key >> (2 * (w + 1))) & 3);
    pattern_count[same4 * 8 + same1]++;
  }
  for (int i = 0; i < 64; ++i)
    if (pattern_count[i]) printf("blocks with %d of 4 (w,w+4) pairs and %d of 4 (2k,2k+1) pairs on one SIMD: %d\n", i / 8, i % 8, pattern_count[i]);
  return 0;
}
