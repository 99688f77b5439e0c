#include <hip/hip_runtime.h>
__device__ __forceinline__ float xor16_max(float x) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_max(float x) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__global__ void k(const float* a, float* o) {
  float x = a[threadIdx.x];
  float m = xor32_max(xor16_max(x));
  float ref = fmaxf(x, __shfl_xor(x, 16, 64));
  ref = fmaxf(ref, __shfl_xor(ref, 32, 64));
  o[threadIdx.x] = m;
  o[64 + threadIdx.x] = ref;
}
int main() {
  float h[64], *d, *o, r[128];
  for (int i = 0; i < 64; ++i) h[i] = (float)((i * 37) % 64);
  hipMalloc(&d, 256); hipMalloc(&o, 512);
  hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
  hipMemcpy(r, o, 512, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; ++i) if (r[i] != r[64 + i]) ++bad;
  printf("permlane swap reduction: %d of 64 lanes differ from the shuffle reduction\n", bad);
  return 0;
}
