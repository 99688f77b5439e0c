// Probe: how exactly does v_mfma_f32_16x16x32_{f16,bf16} add a LARGE accumulator input C to small products?
// (development tool; question raised by the base-2 score path of attention_long.hip, where C = -running maximum)
// One wave: D = A (16 x 32) . B (32 x 16) + C, C = c0 everywhere, products of magnitude ~ |c0| / 32 so that D is near 0.
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma_c_probe mfma_c_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(8))) __bf16 b8;
typedef __attribute__((ext_vector_type(4))) float f4;

// lane l: A row (l & 15), k = 8 (l >> 4) .. + 7; B column (l & 15), same k; D rows 4 (l >> 4) .. + 3, column l & 15
__global__ void probe(const float* A, const float* B, float c0, float* D16, float* Dbf) {
  const int l = threadIdx.x, r = l & 15, g = l >> 4;
  h8 a, b; b8 ab, bb;
  for (int j = 0; j < 8; ++j) {
    a[j] = (_Float16)A[r * 32 + 8 * g + j];
    b[j] = (_Float16)B[(8 * g + j) * 16 + r];
    ab[j] = (__bf16)A[r * 32 + 8 * g + j];
    bb[j] = (__bf16)B[(8 * g + j) * 16 + r];
  }
  f4 c = {c0, c0, c0, c0};
  f4 d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  f4 e = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) {
    D16[(4 * g + i) * 16 + r] = d[i];
    Dbf[(4 * g + i) * 16 + r] = e[i];
  }
}

int main() {
  std::vector<float> A(16 * 32), B(32 * 16);
  float *dA, *dB, *d16, *dbf;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&d16, 256 * 4); hipMalloc(&dbf, 256 * 4);
  for (float c0 : {0.f, -4.f, -16.f, -40.f, -100.f, -1000.f}) {
    // values representable in BOTH bf16 and half (5 significant bits), products positive, sum of a row ~ |c0|
    srand(7);
    const float mag = std::sqrt(std::fabs(c0) / 32.f + 0.05f);
    for (auto& x : A) x = std::ldexp((float)(16 + rand() % 16), -5) * mag;
    for (auto& x : B) x = std::ldexp((float)(16 + rand() % 16), -5) * mag;
    // round to 8 significant bits so that both types hold them exactly
    auto r8 = [](float x) { uint32_t u; memcpy(&u, &x, 4); u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000u; memcpy(&x, &u, 4); return x; };
    for (auto& x : A) x = r8(x);
    for (auto& x : B) x = r8(x);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, c0, d16, dbf);
    std::vector<float> h16(256), hbf(256);
    hipMemcpy(h16.data(), d16, 1024, hipMemcpyDeviceToHost);
    hipMemcpy(hbf.data(), dbf, 1024, hipMemcpyDeviceToHost);
    double e16 = 0, ebf = 0, ref_abs = 0, dot_abs = 0;
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        double s = 0;
        for (int k = 0; k < 32; ++k) s += (double)A[i * 32 + k] * (double)B[k * 16 + j];
        dot_abs = std::fmax(dot_abs, std::fabs(s));
        s += c0;
        e16 = std::fmax(e16, std::fabs(h16[i * 16 + j] - s));
        ebf = std::fmax(ebf, std::fabs(hbf[i * 16 + j] - s));
        ref_abs = std::fmax(ref_abs, std::fabs(s));
      }
    printf("c0 %8.1f  max|dot| %8.3f  max|D| %8.3f   max abs err  f16 %.3e   bf16 %.3e   (fp32 ulp of c0: %.3e)\n", c0, dot_abs, ref_abs, e16, ebf,
           c0 == 0 ? 0.0 : std::ldexp(1.0, (int)std::floor(std::log2(std::fabs(c0))) - 23));
  }
  return 0;
}
