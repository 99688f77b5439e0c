// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands on gfx950: operand lane map, scale semantics, rate.
// Measurement aid (hipcc --offload-arch=gfx950 -O3 -o mx_fp8_probe mx_fp8_probe.hip); not part of the product.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static float e4m3_decode(uint8_t c) {
  const int s = c >> 7, e = (c >> 3) & 15, m = c & 7;
  float v;
  if (e == 15 && m == 7) v = NAN;
  else if (e == 0) v = ldexpf((float)m, -9);
  else v = ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -v : v;
}
static uint8_t e4m3_encode(float x) {
  int best = 0;
  float bd = 1e30f;
  for (int c = 0; c < 256; ++c) {
    const float v = e4m3_decode((uint8_t)c);
    if (std::isnan(v)) continue;
    const float d = fabsf(v - x);
    if (d < bd) { bd = d; best = c; }
  }
  return (uint8_t)best;
}

__global__ void one(const uint8_t* A, const uint8_t* B, const uint8_t* sa, const uint8_t* sb, float* C) {
  // hypothesis: lane l holds A[row l & 31][k = 32 (l >> 5) + 0..31], B[k][col l & 31] likewise (B given as Bt[col][k])
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  v8i a = *reinterpret_cast<const v8i*>(A + r * 64 + 32 * h);
  v8i b = *reinterpret_cast<const v8i*>(B + r * 64 + 32 * h);
  const int scale_a = sa[r * 2 + h], scale_b = sb[r * 2 + h];  // byte 0 of the VGPR
  v16f acc = {0};
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, scale_a, 0, scale_b);
  for (int i = 0; i < 16; ++i) {
    const int row = (i & 3) + 8 * (i >> 2) + 4 * h;  // C/D map of the 32x32 shapes
    C[row * 32 + r] = acc[i];
  }
}

__global__ void rate(float* out, int iters) {
  v8i a, b;
  for (int i = 0; i < 8; ++i) { a[i] = 0x38383838 + threadIdx.x; b[i] = 0x30303030 + i; }
  v16f acc[4];
  for (int j = 0; j < 4; ++j) acc[j] = v16f{0};
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[j], 0, 0, 0, 127, 0, 127);
  float s = 0;
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
typedef short bf8 __attribute__((ext_vector_type(8)));
__global__ void rate_bf16(float* out, int iters) {
  bf8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = 0x3f80 + threadIdx.x; b[i] = 0x3f00 + i; }
  v16f acc[4];
  for (int j = 0; j < 4; ++j) acc[j] = v16f{0};
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
  float s = 0;
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  std::vector<uint8_t> A(32 * 64), B(32 * 64), sa(64), sb(64);
  std::vector<float> Af(32 * 64), Bf(32 * 64);
  uint64_t s = 12345;
  auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (int)((s >> 33) % 7) - 3; };
  for (int i = 0; i < 32 * 64; ++i) {
    Af[i] = (float)rnd();
    Bf[i] = (float)rnd() * 0.5f + (float)(i % 64 > 31) ;  // asymmetric, exact in e4m3
    A[i] = e4m3_encode(Af[i]); B[i] = e4m3_encode(Bf[i]);
    Af[i] = e4m3_decode(A[i]); Bf[i] = e4m3_decode(B[i]);
  }
  for (int i = 0; i < 64; ++i) { sa[i] = (uint8_t)(127 + (i % 5) - 2); sb[i] = (uint8_t)(127 + (i % 3) - 1); }
  uint8_t *dA, *dB, *dsa, *dsb; float* dC;
  CK(hipMalloc(&dA, A.size())); CK(hipMalloc(&dB, B.size())); CK(hipMalloc(&dsa, 64)); CK(hipMalloc(&dsb, 64)); CK(hipMalloc(&dC, 32 * 32 * 4));
  CK(hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(dsa, sa.data(), 64, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, sb.data(), 64, hipMemcpyHostToDevice));
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 0) { for (int i = 0; i < 64; ++i) sa[i] = sb[i] = 127; }
    else { for (int i = 0; i < 64; ++i) { sa[i] = (uint8_t)(127 + (i % 5) - 2); sb[i] = (uint8_t)(127 + (i % 3) - 1); } }
    CK(hipMemcpy(dsa, sa.data(), 64, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, sb.data(), 64, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(one, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dC);
    std::vector<float> C(32 * 32);
    CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
    // the kernel gives lane (r, h) the 32 bytes M[r][32 h + j]; hypothesis hyp says byte j of lane half h is k = kmap(h, j)
    for (int hyp = 0; hyp < 3; ++hyp) {
      auto kmap = [&](int h, int j) {
        if (hyp == 0) return 32 * h + j;
        if (hyp == 1) return j < 16 ? 16 * h + j : 32 + 16 * h + (j - 16);
        return 8 * (2 * (j / 8) + h) + j % 8;
      };
      for (int shyp = 0; shyp < 2; ++shyp) {  // scale of lane (r, h) applies to: 0 = the lane's own 32 bytes; 1 = k block h (k in [32h, 32h+32))
        int bad = 0; double maxd = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
          double e = 0;
          for (int h = 0; h < 2; ++h) for (int b = 0; b < 32; ++b) {
            const int k = kmap(h, b);
            const int blk = shyp == 0 ? h : k / 32;
            e += (double)Af[i * 64 + 32 * h + b] * Bf[j * 64 + 32 * h + b] * ldexp(1.0, sa[i * 2 + blk] - 127) * ldexp(1.0, sb[j * 2 + blk] - 127);
            (void)k;
          }
          const double d = fabs(e - C[i * 32 + j]);
          if (d > 1e-3) ++bad;
          if (d > maxd) maxd = d;
        }
        printf("pass %d (scales %s) kmap %d scalemap %d: %d of 1024 wrong, max diff %g\n", pass, pass ? "varied" : "unit", hyp, shyp, bad, maxd);
      }
    }
    printf("C[0][0..3] = %g %g %g %g   C[1][0] = %g\n", C[0], C[1], C[2], C[3], C[32]);
  }
  float* dout; CK(hipMalloc(&dout, 1024 * 256 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int which = 0; which < 2; ++which)
    for (int wpc : {256, 512}) {
      const int iters = 20000;
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0, 0));
        if (which == 0) hipLaunchKernelGGL(rate, dim3(256), dim3(wpc), 0, 0, dout, iters);
        else hipLaunchKernelGGL(rate_bf16, dim3(256), dim3(wpc), 0, 0, dout, iters);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      }
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double fl = 256.0 * (wpc / 64) * iters * 4 * 2.0 * 32 * 32 * (which == 0 ? 64 : 16);
      printf("%s bare loop, %d waves/CU: %.1f TFLOP/s (%.2f ms)\n", which == 0 ? "mx-fp8 32x32x64" : "bf16 32x32x16", wpc / 64, fl / ms / 1e9, ms);
    }
  return 0;
}
