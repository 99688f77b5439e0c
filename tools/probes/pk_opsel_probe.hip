// Does a packed-fp32 VALU op (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) with a non-default op_sel return wrong lanes
// when another queue keeps the matrix cores / the memory system busy on the same CUs?
//
// Background (DESIGN.md "24-bit residual planes"): the x24 LayerNorm of layernorm.hip returned, beside a busy second
// stream, rows whose last 16 lanes had  y = beta  instead of  gamma * xhat + beta  in ONE component: the low half of a
//   v_pk_fma_f32 vD, vA, vB, vC op_sel:[0,1,0] op_sel_hi:[1,0,1]
// had dropped its product.  This probe runs such ops in a loop on register operands and on freshly loaded operands,
// compares every result with scalar v_fma_f32 / v_mul_f32 / v_add_f32 in the kernel, and counts mismatches per
// 16-lane quad -- alone, beside an MFMA hog and beside a memory hog on a second stream.
//
//   hipcc -O3 --offload-arch=gfx950 -o pk_opsel_probe pk_opsel_probe.hip && ./pk_opsel_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x)                                                                              \
  do {                                                                                        \
    hipError_t e_ = (x);                                                                      \
    if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); }       \
  } while (0)

constexpr int NPAT = 14;
static const char* PAT_NAME[NPAT] = {
    "pk_fma default                               ", "pk_fma op_sel:[0,1,0] op_sel_hi:[1,0,1] (swap) ",
    "pk_mul op_sel_hi:[0,1] (broadcast src0.lo)    ", "pk_add op_sel:[0,1] op_sel_hi:[1,0] (swap)    ",
    "pk_fma op_sel:[1,0,0] (src0.hi for both)      ", "pk_fma swap, operands fresh from global loads ",
    "pk_add op_sel_hi:[1,0] neg (broadcast src1.lo)", "pk_fma op_sel_hi:[1,1,0] (broadcast src2.lo)  ",
    "pk_mul op_sel:[0,1] op_sel_hi:[1,0] (swap)    ", "pk_mov_b32 op_sel:[1,0] (swap)                ",
    "pk_mul op_sel:[1,1] op_sel_hi:[0,0] (full swap)", "pk_fma op_sel:[0,0,1] op_sel_hi:[1,1,0] (swap2)",
    "pk_add v, X, X op_sel:[0,1] op_sel_hi:[1,0]   ", "pk_mul v, X, X op_sel:[0,1] op_sel_hi:[1,0]   "};

__device__ __forceinline__ float sfma(float a, float b, float c) {
  float r;
  asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float smul(float a, float b) {
  float r;
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float sadd(float a, float b) {
  float r;
  asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// counts[pattern][quad]: mismatching results; every lane runs `iters` rounds of every pattern
__global__ __launch_bounds__(256) void victim(int iters, const float* __restrict__ src, unsigned long long* counts) {
  const int lane = threadIdx.x & 63, quad = lane >> 4;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned bad[NPAT] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  f32x2 a = {1.0f + 0.001f * (gid & 1023), -0.5f + 0.003f * (gid & 511)};
  f32x2 b = {0.75f + 0.002f * (gid & 255), 1.25f - 0.001f * (gid & 127)};
  f32x2 c = {0.01f * (gid & 63), -0.02f * (gid & 31)};
  for (int it = 0; it < iters; ++it) {
    f32x2 r;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    bad[0] += (r.x != sfma(a.x, b.x, c.x)) | (r.y != sfma(a.y, b.y, c.y));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    bad[1] += (r.x != sfma(a.x, b.y, c.x)) | (r.y != sfma(a.y, b.x, c.y));
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    bad[2] += (r.x != smul(a.x, b.x)) | (r.y != smul(a.x, b.y));
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
    bad[3] += (r.x != sadd(a.x, b.y)) | (r.y != sadd(a.y, b.x));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    bad[4] += (r.x != sfma(a.y, b.x, c.x)) | (r.y != sfma(a.y, b.y, c.y));
    {  // the LayerNorm's situation: gamma / beta arrive from memory right before the op
      const f32x4 g = *reinterpret_cast<const f32x4*>(src + ((gid * 4 + it * 1024) & 0xFFFFC));
      const f32x2 ga = {g[0], g[1]}, be = {g[2], g[3]};
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(ga), "v"(b), "v"(be));
      bad[5] += (r.x != sfma(ga.x, b.y, be.x)) | (r.y != sfma(ga.y, b.x, be.y));
    }
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    bad[6] += (r.x != sadd(a.x, -b.x)) | (r.y != sadd(a.y, -b.x));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    bad[7] += (r.x != sfma(a.x, b.x, c.x)) | (r.y != sfma(a.y, b.y, c.x));
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
    bad[8] += (r.x != smul(a.x, b.y)) | (r.y != smul(a.y, b.x));
    asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(a), "v"(b));
    bad[9] += (r.x != a.y) | (r.y != b.x);
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0]" : "=v"(r) : "v"(a), "v"(b));
    bad[10] += (r.x != smul(a.y, b.y)) | (r.y != smul(a.x, b.x));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    bad[11] += (r.x != sfma(a.x, b.x, c.y)) | (r.y != sfma(a.y, b.y, c.x));
    // the form hipcc uses for a row sum: BOTH sources are the same register pair
    asm volatile("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a));
    bad[12] += (r.x != sadd(a.x, a.y)) | (r.y != sadd(a.y, a.x));
    asm volatile("v_pk_mul_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(b));
    bad[13] += (r.x != smul(b.x, b.y)) | (r.y != smul(b.y, b.x));
    a.x += 0.0009765625f; b.y -= 0.00048828125f; c.x += 0.001953125f;  // keep the operands moving
  }
#pragma unroll
  for (int p = 0; p < NPAT; ++p)
    if (bad[p]) atomicAdd(&counts[p * 4 + quad], (unsigned long long)bad[p]);
}

// matrix-core hog: dependent MFMA chains, no memory traffic
__global__ __launch_bounds__(256) void mfma_hog(int iters, float* sink) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc1, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc2, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc3, 0, 0, 0);
  }
  if (acc0[0] + acc1[1] + acc2[2] + acc3[3] == 12345.678f) sink[0] = 1.f;
}

// LDS hog: ds_read_b128 / ds_write_b128 traffic, no MFMA
__global__ __launch_bounds__(256) void lds_hog(int iters, float* sink) {
  __shared__ f32x4 buf[2048];
  for (int i = threadIdx.x; i < 2048; i += 256) buf[i] = f32x4{(float)i, 1.f, 2.f, 3.f};
  __syncthreads();
  f32x4 s = {0, 0, 0, 0};
  int idx = threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    s += buf[idx & 2047];
    s += buf[(idx + 256) & 2047];
    s += buf[(idx + 512) & 2047];
    s += buf[(idx + 768) & 2047];
    idx += 17;
  }
  if (s[0] + s[1] + s[2] + s[3] == 12345.678f) sink[0] = 1.f;
}

// GEMM-like hog: LDS fragment reads feeding MFMAs, a barrier per step
__global__ __launch_bounds__(256) void gemm_hog(int iters, float* sink) {
  __shared__ bf16x8 buf[2048];
  for (int i = threadIdx.x; i < 2048; i += 256)
    for (int e = 0; e < 8; ++e) buf[i][e] = (__bf16)(0.001f * (i + e));
  __syncthreads();
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  int idx = threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    const bf16x8 a0 = buf[idx & 2047], a1 = buf[(idx + 256) & 2047], b0 = buf[(idx + 512) & 2047], b1 = buf[(idx + 768) & 2047];
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b1, acc[1], 0, 0, 0);
    acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b0, acc[2], 0, 0, 0);
    acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc[3], 0, 0, 0);
    idx += 33;
    if ((it & 15) == 15) __syncthreads();
  }
  if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 12345.678f) sink[0] = 1.f;
}

// variants of the GEMM-like hog, to see which ingredient matters
template <bool FEED, bool BARRIER>
__global__ __launch_bounds__(256) void gemm_hog_v(int iters, float* sink) {
  __shared__ bf16x8 buf[2048];
  for (int i = threadIdx.x; i < 2048; i += 256)
    for (int e = 0; e < 8; ++e) buf[i][e] = (__bf16)(0.001f * (i + e));
  __syncthreads();
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  bf16x8 ra, rb, junk = buf[threadIdx.x];
  for (int i = 0; i < 8; ++i) { ra[i] = (__bf16)(0.001f * (threadIdx.x + i)); rb[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
  int idx = threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    const bf16x8 a0 = buf[idx & 2047], a1 = buf[(idx + 256) & 2047], b0 = buf[(idx + 512) & 2047], b1 = buf[(idx + 768) & 2047];
    if (FEED) {  // the LDS reads feed the MFMAs
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b0, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc[3], 0, 0, 0);
    } else {     // same LDS traffic and same MFMAs, but the MFMAs run on registers
      for (int e = 0; e < 8; ++e) junk[e] = (__bf16)((float)junk[e] + (float)a0[e] + (float)a1[e] + (float)b0[e] + (float)b1[e]);
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ra, rb, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ra, rb, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ra, rb, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ra, rb, acc[3], 0, 0, 0);
    }
    idx += 33;
    if (BARRIER && (it & 15) == 15) __syncthreads();
  }
  if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + (float)junk[0] == 12345.678f) sink[0] = 1.f;
}

// memory hog: streams a large buffer
__global__ __launch_bounds__(256) void mem_hog(const f32x4* __restrict__ p, size_t n, int passes, float* sink) {
  f32x4 s = {0, 0, 0, 0};
  for (int k = 0; k < passes; ++k)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
  if (s[0] + s[1] + s[2] + s[3] == 12345.678f) sink[0] = 1.f;
}

int main(int argc, char** argv) {
  float *src, *sink;
  unsigned long long* counts;
  f32x4* big;
  const size_t big_n = (size_t)1 << 26;  // 1 GiB
  CHECK(hipMalloc(&src, 4 << 20));
  CHECK(hipMalloc(&sink, 64));
  CHECK(hipMalloc(&counts, NPAT * 4 * sizeof(unsigned long long)));
  CHECK(hipMalloc(&big, big_n * sizeof(f32x4)));
  CHECK(hipMemset(big, 0, big_n * sizeof(f32x4)));
  {
    std::vector<float> h(1 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.5f + 1e-6f * (float)(i % 100003);
    CHECK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  }
  hipStream_t s1, s2;
  CHECK(hipStreamCreate(&s1));
  CHECK(hipStreamCreate(&s2));
  const char* modes[8] = {"alone", "beside an MFMA hog", "beside a memory hog", "beside an LDS hog", "beside a GEMM-like hog (LDS + MFMA + barriers)", "beside LDS-fed MFMAs WITHOUT barriers", "beside LDS reads + MFMAs on REGISTERS, with barriers", "beside LDS reads + MFMAs on registers, no barriers"};
  for (int mode = 0; mode < 8; ++mode) {
    CHECK(hipMemset(counts, 0, NPAT * 4 * sizeof(unsigned long long)));
    CHECK(hipDeviceSynchronize());
    const int rounds = argc > 1 ? atoi(argv[1]) : 40;
    for (int r = 0; r < rounds; ++r) {
      if (mode == 1) hipLaunchKernelGGL(mfma_hog, dim3(1024), dim3(256), 0, s2, 40000, sink);
      if (mode == 3) hipLaunchKernelGGL(lds_hog, dim3(1024), dim3(256), 0, s2, 40000, sink);
      if (mode == 4) hipLaunchKernelGGL(gemm_hog, dim3(1024), dim3(256), 0, s2, 40000, sink);
      if (mode == 5) hipLaunchKernelGGL((gemm_hog_v<true, false>), dim3(1024), dim3(256), 0, s2, 40000, sink);
      if (mode == 6) hipLaunchKernelGGL((gemm_hog_v<false, true>), dim3(1024), dim3(256), 0, s2, 40000, sink);
      if (mode == 7) hipLaunchKernelGGL((gemm_hog_v<false, false>), dim3(1024), dim3(256), 0, s2, 40000, sink);
      if (mode == 2) hipLaunchKernelGGL(mem_hog, dim3(2048), dim3(256), 0, s2, big, big_n, 2, sink);
      // many small victim launches, like the tower's 100-block LayerNorm
      for (int k = 0; k < 50; ++k) hipLaunchKernelGGL(victim, dim3(100), dim3(256), 0, s1, 64, src, counts);
      CHECK(hipStreamSynchronize(s1));
      CHECK(hipStreamSynchronize(s2));
    }
    std::vector<unsigned long long> h(NPAT * 4);
    CHECK(hipMemcpy(h.data(), counts, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    const double per = (double)rounds * 50 * 100 * 256 * 64;
    printf("== %s (%.3g results per pattern)\n", modes[mode], per);
    for (int p = 0; p < NPAT; ++p)
      if (mode == 0 || h[p * 4] + h[p * 4 + 1] + h[p * 4 + 2] + h[p * 4 + 3] > 0 || p == 0)
        printf("  %s mismatches by 16-lane quad: %llu %llu %llu %llu\n", PAT_NAME[p], h[p * 4], h[p * 4 + 1], h[p * 4 + 2], h[p * 4 + 3]);
  }
  return 0;
}
