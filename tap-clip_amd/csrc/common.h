// Shared device/host helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
// TAPCLIP_TU_NO_PK_F32 (defined by a .hip file BEFORE it includes this header; the file ends with
// TAPCLIP_TU_NO_PK_F32_END): no packed-fp32 VALU ops (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) in any function of
// that translation unit -- the helpers of this header and of the HIP headers included, so that they still inline (a
// function is only inlined into callers with at least its target features).
// Measured on MI355X (tools/probes/pk_opsel_table.hip, profiles/r02_pk_opsel_table.txt): a packed-fp32 op whose LOW
// result takes src0's low register and src1's HIGH register -- op_sel:[0,1] / [0,1,x], any op_sel_hi -- returns wrong
// values in lanes 48..63 while a wave of ANOTHER kernel runs LDS-fed MFMAs on the same CU (1-12 % of those results
// beside a GEMM-like kernel on a second stream; never alone; the other 72 of the 96 encodings never; nor the form with
// src0 == src1, the pair swap of a row sum).  hipcc emits that encoding on its own from plain C++ -- in this library
// gamma * (x, y) with the pair held as (y, x), in the LayerNorm of the 24-bit residual planes -- so that LayerNorm
// returned, beside the other tower's GEMMs, rows with one component at beta.  Files whose kernels can run beside
// another stream's GEMM and in which the compiler forms such ops are compiled this way; tests/test_abi.py
// disassembles the built libraries and fails on any packed-fp32 op with op_sel = [0,1,...] (same-register form
// included: the check is on the encoding).
#if defined(TAPCLIP_TU_NO_PK_F32) && defined(__HIP_DEVICE_COMPILE__)
#pragma clang attribute push(__attribute__((target("no-packed-fp32-ops"))), apply_to = function)
#define TAPCLIP_TU_NO_PK_F32_END _Pragma("clang attribute pop")
#else
#define TAPCLIP_TU_NO_PK_F32_END
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>

#if defined(TAPCLIP_TU_NO_PK_F32) && defined(__HIP_DEVICE_COMPILE__)
// threadIdx / blockIdx / blockDim / gridDim are device-library accessors compiled WITH packed-fp32 ops: they do not
// inline into a no-packed-fp32 function (a callee is inlined only into callers that have all of its target features) and
// would become real calls.  Inside such a translation unit they are the hardware builtins instead.
namespace tapclip {
struct dim3u { unsigned x, y, z; };
__device__ __forceinline__ dim3u tid3() { return {__builtin_amdgcn_workitem_id_x(), __builtin_amdgcn_workitem_id_y(), __builtin_amdgcn_workitem_id_z()}; }
__device__ __forceinline__ dim3u bid3() { return {__builtin_amdgcn_workgroup_id_x(), __builtin_amdgcn_workgroup_id_y(), __builtin_amdgcn_workgroup_id_z()}; }
__device__ __forceinline__ dim3u bdim3() { return {__builtin_amdgcn_workgroup_size_x(), __builtin_amdgcn_workgroup_size_y(), __builtin_amdgcn_workgroup_size_z()}; }
__device__ __forceinline__ dim3u gdim3() {
  return {__builtin_amdgcn_grid_size_x() / __builtin_amdgcn_workgroup_size_x(), __builtin_amdgcn_grid_size_y() / __builtin_amdgcn_workgroup_size_y(),
          __builtin_amdgcn_grid_size_z() / __builtin_amdgcn_workgroup_size_z()};
}
}  // namespace tapclip
#define threadIdx (::tapclip::tid3())
#define blockIdx (::tapclip::bid3())
#define blockDim (::tapclip::bdim3())
#define gridDim (::tapclip::gdim3())
#endif

namespace tapclip {

// The 16-bit operand type.  Every kernel handles it as raw 16-bit patterns (`bf16_t`) through the helpers below,
// so the whole library can be compiled a second time with -DTAPCLIP_FP16 (-> libtapclip_fp16.so): IEEE half
// operands on v_mfma_f32_16x16x32_f16 at the bf16 MFMA rate, 11 significand bits instead of 8 -- the fast path
// at ~8x smaller operand rounding error (precision "fp16": inference of the image tower; its exponent range is
// 6e-8 .. 65504, so the gradients of the prompt-tuning backward stay on the bf16 library).
typedef uint16_t bf16_t;  // raw 16-bit operand bits (bf16, or fp16 under TAPCLIP_FP16)
#ifdef TAPCLIP_FP16
typedef _Float16 half_t;
#define TAPCLIP_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#else
typedef __bf16 half_t;
#define TAPCLIP_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#endif
typedef __attribute__((ext_vector_type(8))) half_t bf16x8_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// round-to-nearest-even fp32 -> 16-bit operand (v_cvt_pk_bf16_f32 / v_cvt_f16_f32, NaN-safe)
__device__ __forceinline__ bf16_t f2bf(float x) {
  half_t b = (half_t)x;
  return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ float bf2f(bf16_t b) {
#ifdef TAPCLIP_FP16
  return (float)__builtin_bit_cast(half_t, b);
#else
  return __builtin_bit_cast(float, ((uint32_t)b) << 16);
#endif
}
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) half_t bf16x2_t;
// two fp32 -> packed pair (bf16: one v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
// ---- MXFP8 (OCP MX: e4m3 elements, one e8m0 scale per 32 consecutive k): quantisation helpers.
// The shared exponent follows the OCP MX v1.0 recipe: 2^(floor(log2(amax)) - 8) (8 = e4m3's largest exponent),
// elements saturate at +-448.  Scale arrays are k-step major (kernels.h Mx8GemmArgs).
__device__ __forceinline__ uint32_t mx8_scale_byte(float amax) {  // amax >= 0
  const int e = (int)(__float_as_uint(amax) >> 23) - 8;
  return (uint32_t)(e < 0 ? 0 : e);  // zero / denormal blocks: 2^-127
}
__device__ __forceinline__ float mx8_inv_scale(uint32_t byte) { return __uint_as_float((254u - byte) << 23); }  // 2^(127 - byte)
// a register with no defined content and no instruction behind it: both halves of a packed word are written by the two
// conversions, so zeroing it first (what passing 0 as the old value costs: one v_mov per word) is wasted issue
__device__ __forceinline__ int mx8_undef_word() {
  int v;
  asm("" : "=v"(v));
  return v;
}
__device__ __forceinline__ uint32_t mx8_pack4(float a, float b, float c, float d, float inv) {
  a = __builtin_amdgcn_fmed3f(a * inv, -448.f, 448.f);
  b = __builtin_amdgcn_fmed3f(b * inv, -448.f, 448.f);
  c = __builtin_amdgcn_fmed3f(c * inv, -448.f, 448.f);
  d = __builtin_amdgcn_fmed3f(d * inv, -448.f, 448.f);
  int v = mx8_undef_word();
  v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, v, false);  // bytes 0, 1 (round to nearest even, OCP e4m3 on gfx950)
  v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);   // bytes 2, 3
  return (uint32_t)v;
}
// max(m, |a|, |b|) in one op (fmaxf / fabsf on packed results no longer fuse into v_max3_f32 by themselves).  ONLY for values a
// VALU instruction produced: hipcc does not pad an asm statement for the MFMA -> VALU read hazard (gfx950 has no interlock; see
// attention_long.hip flash2_tile_max and profiles/r05_flash2_asm_hazard.txt), so raw MFMA results take amax3_visible.
__device__ __forceinline__ float amax3_raw(float m, float a, float b) {
  float d;
  asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(d) : "v"(m), "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ float amax3_visible(float m, float a, float b) { return fmaxf(m, fmaxf(__builtin_fabsf(a), __builtin_fabsf(b))); }
// the same on two pairs (the scale multiply as v_pk_mul_f32)
typedef float f32x2_mx_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t mx8_pack4(f32x2_mx_t ab, f32x2_mx_t cd, float inv) {
  ab = ab * f32x2_mx_t{inv, inv};
  cd = cd * f32x2_mx_t{inv, inv};
  int v = mx8_undef_word();
  v = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(ab[0], -448.f, 448.f), __builtin_amdgcn_fmed3f(ab[1], -448.f, 448.f), v, false);
  v = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(cd[0], -448.f, 448.f), __builtin_amdgcn_fmed3f(cd[1], -448.f, 448.f), v, true);
  return (uint32_t)v;
}
__device__ __forceinline__ size_t mx8_scale_index(int64_t row, int block, int64_t rows_pad) {
  return ((size_t)(block >> 1) * rows_pad + row) * 2 + (block & 1);
}

// split x into hi + lo bf16 (lo = bf16(x - float(hi))): the bf16x3 operand form
__device__ __forceinline__ void split_bf(float x, bf16_t& hi, bf16_t& lo) {
  hi = f2bf(x);
  lo = f2bf(x - bf2f(hi));
}

// Reductions over the four lanes l, l ^ 16, l ^ 32, l ^ 48 (one per 16-lane row) on the VALU: v_permlane16_swap /
// v_permlane32_swap of (x, x) leave row pairs (halves) of x in the two results, so one max / add finishes an xor
// step -- no LDS round trip (__shfl_xor lowers to ds_bpermute_b32: ~100 cycles, four of them in sequence per softmax row).
__device__ __forceinline__ float xor16_max(float x) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_max(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor16_sum(float x) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_sum(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float rows_max(float x) { return xor32_max(xor16_max(x)); }
__device__ __forceinline__ float rows_sum(float x) { return xor32_sum(xor16_sum(x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// exact-erf GELU (nn.GELU default).  erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7): one v_rcp, one
// v_exp and six FMAs instead of the ~30-instruction libm erff -- the c_fc epilogue applies it to
// 3072 values per token, so its VALU cost is visible next to a K = 768 MFMA loop.
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __expf(-ax * ax);
  const float y = fmaf(-p * t, e, 1.0f);
  return copysignf(y, x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752f)); }
// GELU for the bf16 fast path: x * sigmoid(x * (a + b x^2 + c x^4)) fitted to the exact-erf GELU,
// max |error| 2.5e-5 over all x (vs 2^-9 relative bf16 rounding of the stored value): 9 VALU ops with
// 2 transcendentals instead of 15.  Coefficients are pre-multiplied by -log2(e); x is clamped to +-10 inside
// the polynomial (it turns over beyond |x| ~ 11; sigmoid is saturated there anyway).
__device__ __forceinline__ float gelu_erf_fast(float x) {
  const float xc = __builtin_amdgcn_fmed3f(x, -10.0f, 10.0f);
  const float x2 = xc * xc;
  float p = fmaf(1.0142631e-3f, x2, -1.0677573e-1f);   // -log2e * (c x^2 + b)
  p = fmaf(p, x2, -2.3011213f);                        // -log2e * a
  const float e = __builtin_amdgcn_exp2f(p * xc);
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}
// Transcendental-free GELU (round 4):  gelu(x) = max(x, 0) + f(min(|x|, 4.5)),  f(t) = gelu(t) - t = -t Phi(-t)  -- a smooth bump
// (-0.17 at t = 0.75) with a Gaussian tail, -1.5e-5 at the clamp -- by a degree-10 polynomial in u = t / 2.25 - 1 (Chebyshev-range
// variable: coefficients below 0.44, Horner in fp32 loses nothing).  Max |error| against the exact-erf form 1.44e-5 over all x
// (the sigmoid fit above: 2.5e-5).  Cost per PAIR of values: 2 v_min (|x| is a source modifier), 1 + 10 v_pk_fma_f32, 2 v_max,
// 1 v_pk_add_f32 = 8 issue slots per value against 12 for the fit (whose v_exp_f32 / v_rcp_f32 are quarter rate).
#ifndef TAPCLIP_GELU_FORM
#define TAPCLIP_GELU_FORM 0  // 0: the sigmoid fit, packed (default); 1: this polynomial; 2: the sigmoid fit, scalar (A/B only)
#endif
typedef float f32x2_pk_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_pk_t gelu_poly2(f32x2_pk_t x) {
  f32x2_pk_t t, r;
  // (plain fminf / fmaxf: they canonicalise their operand first, one more VALU op per value -- the round-4 measurement of this
  // form used raw v_min / v_max in asm statements, which hipcc does not pad for the MFMA -> VALU read hazard: not kept)
  t[0] = fminf(__builtin_fabsf(x[0]), 4.5f);
  t[1] = fminf(__builtin_fabsf(x[1]), 4.5f);
  r[0] = fmaxf(x[0], 0.f);
  r[1] = fmaxf(x[1], 0.f);
  const f32x2_pk_t u = __builtin_elementwise_fma(t, f32x2_pk_t{0.44444445f, 0.44444445f}, f32x2_pk_t{-1.0f, -1.0f});
  constexpr float C[11] = {-2.749713324e-02f, 1.330395043e-01f, -2.465924025e-01f, 1.472157985e-01f, 2.029682845e-01f, -4.347813427e-01f,
                           2.049071938e-01f,  1.763149798e-01f, -1.763803661e-01f, -2.178246900e-02f, 4.258675873e-02f};
  f32x2_pk_t p = {C[10], C[10]};
#pragma unroll
  for (int k = 9; k >= 0; --k) p = __builtin_elementwise_fma(p, u, f32x2_pk_t{C[k], C[k]});
  return r + p;
}
__device__ __forceinline__ float gelu_poly(float x) { return gelu_poly2(f32x2_pk_t{x, x})[0]; }  // (same op sequence: same bits)
// gelu_erf_fast on a PAIR of values with the plain ops packed (v_pk_mul / v_pk_fma / v_pk_add_f32: the same roundings, the same
// bits as the scalar form): 2 v_med3 + 6 packed ops + 2 v_exp + 2 v_rcp = 6 issue slots per value instead of 8.5.
__device__ __forceinline__ f32x2_pk_t gelu_erf_fast2(f32x2_pk_t x) {
  const f32x2_pk_t xc = {__builtin_amdgcn_fmed3f(x[0], -10.0f, 10.0f), __builtin_amdgcn_fmed3f(x[1], -10.0f, 10.0f)};
  const f32x2_pk_t x2 = xc * xc;
  f32x2_pk_t p = __builtin_elementwise_fma(f32x2_pk_t{1.0142631e-3f, 1.0142631e-3f}, x2, f32x2_pk_t{-1.0677573e-1f, -1.0677573e-1f});
  p = __builtin_elementwise_fma(p, x2, f32x2_pk_t{-2.3011213f, -2.3011213f});
  const f32x2_pk_t a = p * xc;
  const f32x2_pk_t d = f32x2_pk_t{__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])} + f32x2_pk_t{1.0f, 1.0f};
  return x * f32x2_pk_t{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}
// QuickGELU, packed the same way (bits of gelu_quick_fast below)
__device__ __forceinline__ f32x2_pk_t gelu_quick_fast2(f32x2_pk_t x) {
  constexpr float K = -1.702f * 1.4426950408889634f;
  const f32x2_pk_t a = f32x2_pk_t{K, K} * x;
  const f32x2_pk_t d = f32x2_pk_t{1.0f, 1.0f} + f32x2_pk_t{__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
  return x * f32x2_pk_t{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}
// GELU of the 16-bit fast paths (bf16 and the IEEE-half build).  (The erf form in the half build changed nothing measurable --
// cfg-1 logits 1.3e-3 -> 1.7e-3 rel-max, same 9.8e-4 rel-L2 -- and cost 40 us per c_fc.)
__device__ __forceinline__ float gelu_fast16(float x) { return TAPCLIP_GELU_FORM == 1 ? gelu_poly(x) : gelu_erf_fast(x); }
template <typename V4>
__device__ __forceinline__ void gelu_quick_fast_x4(V4& v) {
  const f32x2_pk_t a = gelu_quick_fast2(f32x2_pk_t{v[0], v[1]}), b = gelu_quick_fast2(f32x2_pk_t{v[2], v[3]});
  v[0] = a[0]; v[1] = a[1]; v[2] = b[0]; v[3] = b[1];
}
// four values of an accumulator at once (the tiled GEMMs' epilogues)
template <typename V4>
__device__ __forceinline__ void gelu_fast16_x4(V4& v) {
  if (TAPCLIP_GELU_FORM == 1) {
    const f32x2_pk_t a = gelu_poly2(f32x2_pk_t{v[0], v[1]}), b = gelu_poly2(f32x2_pk_t{v[2], v[3]});
    v[0] = a[0]; v[1] = a[1]; v[2] = b[0]; v[3] = b[1];
  } else if (TAPCLIP_GELU_FORM == 0) {
    const f32x2_pk_t a = gelu_erf_fast2(f32x2_pk_t{v[0], v[1]}), b = gelu_erf_fast2(f32x2_pk_t{v[2], v[3]});
    v[0] = a[0]; v[1] = a[1]; v[2] = b[0]; v[3] = b[1];
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = gelu_erf_fast(v[e]);
  }
}
// derivatives (backward of c_fc's activation); the bf16 forward uses the fitted GELU, whose derivative
// differs from the exact one by < 2e-4
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.0f + erf_as(x * 0.70710678118654752f));
  return fmaf(x * 0.3989422804014327f, __expf(-0.5f * x * x), cdf);
}
// derivative of the FITTED form above, g(x) = x s(t), t = a x + b x^3 + c x^5, s = sigmoid: g' = s + x s (1 - s) t'(x).
// The backward of the 16-bit towers uses it: it differentiates the function their forward really applied (the exact-erf
// derivative is < 2e-4 away), with 2 transcendentals and 10 plain ops instead of 3 and ~20.
__device__ __forceinline__ float gelu_fit_grad(float x) {
  constexpr float L2E = 1.4426950408889634f;
  constexpr float A = 2.3011213f / L2E, B3 = 3.0f * 1.0677573e-1f / L2E, C5 = -5.0f * 1.0142631e-3f / L2E;
  const float xc = __builtin_amdgcn_fmed3f(x, -10.0f, 10.0f);
  const float x2 = xc * xc;
  float p = fmaf(1.0142631e-3f, x2, -1.0677573e-1f);
  p = fmaf(p, x2, -2.3011213f);
  const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(p * xc));
  const float tp = fmaf(fmaf(C5, x2, B3), x2, A);
  return fmaf(xc * fmaf(-sg, sg, sg), tp, sg);
}
__device__ __forceinline__ float gelu_quick_grad(float x) {
  const float s = 1.0f / (1.0f + __expf(-1.702f * x));
  return s * (1.0f + 1.702f * x * (1.0f - s));
}
__device__ __forceinline__ float gelu_quick(float x) { return x / (1.0f + __expf(-1.702f * x)); }
// bf16 fast path: v_exp_f32 + v_rcp_f32 (1 ulp each) instead of the IEEE division sequence
__device__ __forceinline__ float gelu_quick_fast(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * x));
}

}  // namespace tapclip
