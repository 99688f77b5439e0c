// LayerNorm (eps 1e-5, affine) as a wavefront reduction: one 64-lane wave per row, fp32 statistics.
// Replaces the ln_pre / ln_1 / ln_2 / ln_post nn.LayerNorm calls inside open_clip's blocks
// (SURVEY.md section 2.1 K2).  HBM-bound: reads the fp32 residual row once, writes the bf16 GEMM
// operand (hi, and lo for bf16x3) or fp32 (ln_pre in place, unit API).
// (no packed-fp32 ops in this file: a LayerNorm runs beside the other tower's GEMMs -- common.h)
#define TAPCLIP_TU_NO_PK_F32
#include "common.h"
#include "kernels.h"

namespace tapclip {
namespace {

// MODE 0: bf16 hi   1: bf16 hi + lo   2: fp32   3: MXFP8 (e4m3 bytes + one e8m0 scale per 32 columns; vector kernel only)
// ADD: the row first receives pending residual branches (bf16 hi [+ lo], outputs of the preceding out_proj /
// c_proj GEMMs) before it is normalised:
//   1  x += d1, written back            2  x + d1 normalised, x NOT written back (LN2: saves 4 of 12 B/element)
//   3  x += d1 + d2, written back       (the next block's LN1 then folds both branches: 14 B/element; a block's two
//                                        LayerNorms move 22 B/element instead of 24)
// XF: the format of the residual stream (row stride d when not fp32):
//   0  fp32 x
//   1  16-bit x16 (bf16 / IEEE half) -- the fp8 precision only, where a 2^-9 rounding of x per block is far below the
//      MXFP8 operand error and the LayerNorms are 20 % of the step
//   2  "x24": the fp32 value rounded (nearest-even) to 16 significand bits and kept as two planes, x16 = its upper 16
//      bits and x8 = the next 8: 3 bytes per element at 2^-17 relative rounding (2e-5 over the 24 roundings of ViT-B:
//      a tenth of the IEEE-half operand error of the "fp16" mode, 1 % of bf16's).  The LayerNorm kernels are HBM-bound
//      (6 TB/s) and their bytes drop 14 -> 12 (LN1) and 8 -> 7 (LN2): -1.3 % on the step.  Default of the image
//      tower's 16-bit modes (TAPCLIP_X24=0: fp32 stream).
// PRE (XF = 2, ADD = 0): the row is first read as fp32 from x, normalised with (gamma_pre, beta_pre) -- ln_pre of the
//   image tower -- and THAT is the residual row, written in the XF format; then the LayerNorm proper (block 0's ln_1)
//   runs on it: one pass over the patch embeddings instead of an in-place fp32 ln_pre followed by a second kernel.
__device__ __forceinline__ uint32_t x24_bits(float v) {  // fp32 bits rounded to nearest-even at bit 8
  const uint32_t u = __float_as_uint(v);
  const uint32_t r = u + 0x7Fu + ((u >> 8) & 1u);
  // exponent all ones: no rounding add (it would carry an all-ones-mantissa NaN over into -0.0 / +0.0), and a NaN whose
  // payload sits in the 8 dropped bits gets its quiet bit set, so that a NaN in the residual stays a NaN in the planes
  const bool special = (u & 0x7F800000u) == 0x7F800000u;
  return special ? (u | ((u & 0x007FFFFFu) ? 0x00400000u : 0u)) : r;
}
typedef uint32_t ln_u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void x24_store(bf16_t* hi, uint8_t* lo, const float4& v) {
  const uint32_t a = x24_bits(v.x), b = x24_bits(v.y), c = x24_bits(v.z), e = x24_bits(v.w);
#ifndef TAPCLIP_X24_PLAIN  // see x24_load
  __builtin_nontemporal_store((ln_u32x2_t{(a >> 16) | (b & 0xFFFF0000u), (c >> 16) | (e & 0xFFFF0000u)}), reinterpret_cast<ln_u32x2_t*>(hi));
  __builtin_nontemporal_store(((a >> 8) & 0xFFu) | (b & 0xFF00u) | ((c << 8) & 0xFF0000u) | ((e << 16) & 0xFF000000u), reinterpret_cast<uint32_t*>(lo));
#else
  *reinterpret_cast<uint2*>(hi) = make_uint2((a >> 16) | (b & 0xFFFF0000u), (c >> 16) | (e & 0xFFFF0000u));
  *reinterpret_cast<uint32_t*>(lo) = ((a >> 8) & 0xFFu) | (b & 0xFF00u) | ((c << 8) & 0xFF0000u) | ((e << 16) & 0xFF000000u);
#endif
}
__device__ __forceinline__ float4 x24_load(const bf16_t* hi, const uint8_t* lo) {
  // The planes are streamed non-temporally like the fp32 stream (NTX below): 11.53 -> 11.19 ms per step with the planes
  // (stores alone: nothing; loads alone: -1.9 %).
#ifndef TAPCLIP_X24_PLAIN
  const ln_u32x2_t hh = __builtin_nontemporal_load(reinterpret_cast<const ln_u32x2_t*>(hi));
  const uint2 h = make_uint2(hh[0], hh[1]);
  const uint32_t l = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(lo));
#else
  const uint2 h = *reinterpret_cast<const uint2*>(hi);
  const uint32_t l = *reinterpret_cast<const uint32_t*>(lo);
#endif
  return make_float4(__uint_as_float((h.x << 16) | ((l << 8) & 0xFF00u)), __uint_as_float((h.x & 0xFFFF0000u) | (l & 0xFF00u)),
                     __uint_as_float((h.y << 16) | ((l >> 8) & 0xFF00u)), __uint_as_float((h.y & 0xFFFF0000u) | ((l >> 16) & 0xFF00u)));
}

// NTX (XF = 0): the fp32 residual row is loaded and stored NON-TEMPORALLY -- the image tower's 16-bit modes.  A row of the
//   residual stream is touched once per LayerNorm and not again for ~250 us and ~1 GB of other traffic, so keeping it
//   out of the L2 / Infinity Cache leaves them to the tensors the NEXT kernel reads (this kernel's 16-bit output, the
//   GEMM outputs): 11.98 -> 11.59 ms per step (bench.py, same box, interleaved), QKV 177 -> 169 us, c_fc 276 -> 263 us,
//   attention 82 -> 77 us, the LayerNorm kernels themselves unchanged.  (The text tower's stream is 12 MB and lives in
//   the caches anyway.)
template <int MODE, int NV, int ADD, int XF = 0, bool PRE = false, bool NTX = false>  // NV float4 per lane: d = 256 * NV
__global__ __launch_bounds__(256) void ln_vec_kernel(float* __restrict__ x, int64_t ldx, bf16_t* __restrict__ x16, uint8_t* __restrict__ x8,
                                                     const bf16_t* __restrict__ d_hi, const bf16_t* __restrict__ d_lo,
                                                     const bf16_t* __restrict__ e_hi, const bf16_t* __restrict__ e_lo,
                                                     const float* __restrict__ gamma_pre, const float* __restrict__ beta_pre,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, int64_t rows, int d,
                                                     bf16_t* out_hi, bf16_t* out_lo, float* out_f32,
                                                     uint8_t* out_q, uint8_t* out_sc, int64_t rows_pad) {
  static_assert(!PRE || (XF == 2 && ADD == 0), "PRE: fp32 source -> x24 residual");
  constexpr bool XH = XF == 1;
  const int tid = (int)__builtin_amdgcn_workitem_id_x();  // (threadIdx / blockIdx are device-library calls that do not
  const int lane = tid & 63;                              //  inline into a function without packed-fp32 ops)
  const int64_t row = (int64_t)__builtin_amdgcn_workgroup_id_x() * 4 + (tid >> 6);
  if (row >= rows) return;
  float* xr = (XF != 0 && !PRE) ? nullptr : x + row * ldx;
  float4 v[NV];
  float s = 0.f;
  if (PRE) {  // ln_pre in registers; its output is the residual row
    float s0 = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      {  // (the fp32 patch embeddings are dead after this read: streamed past the caches too, -0.5 % on the step)
        typedef float lf32x4_t __attribute__((ext_vector_type(4)));
        const lf32x4_t t4 = __builtin_nontemporal_load(reinterpret_cast<const lf32x4_t*>(xr + 4 * lane + 256 * j));
        v[j] = make_float4(t4[0], t4[1], t4[2], t4[3]);
      }
      s0 += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    }
    const float mean0 = wave_sum(s0) / (float)d;
    float ss0 = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      v[j].x -= mean0; v[j].y -= mean0; v[j].z -= mean0; v[j].w -= mean0;
      ss0 += (v[j].x * v[j].x + v[j].y * v[j].y) + (v[j].z * v[j].z + v[j].w * v[j].w);
    }
    const float rstd0 = rsqrtf(wave_sum(ss0) / (float)d + 1e-5f);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = 4 * lane + 256 * j;
      const float4 gm = *reinterpret_cast<const float4*>(gamma_pre + c);
      const float4 bt = *reinterpret_cast<const float4*>(beta_pre + c);
      v[j] = make_float4(v[j].x * rstd0 * gm.x + bt.x, v[j].y * rstd0 * gm.y + bt.y, v[j].z * rstd0 * gm.z + bt.z, v[j].w * rstd0 * gm.w + bt.w);
      x24_store(x16 + row * d + c, x8 + row * d + c, v[j]);
      // the blocks read the residual back from its 24-bit planes: normalise exactly what they will see
      v[j] = make_float4(__uint_as_float(x24_bits(v[j].x) & 0xFFFFFF00u), __uint_as_float(x24_bits(v[j].y) & 0xFFFFFF00u),
                         __uint_as_float(x24_bits(v[j].z) & 0xFFFFFF00u), __uint_as_float(x24_bits(v[j].w) & 0xFFFFFF00u));
    }
  }
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = 4 * lane + 256 * j;
    if (PRE) {
    } else if (XH) {
      // (streamed non-temporally like the 24-bit planes: +0.8 % on the fp8 step)
      const ln_u32x2_t hh = __builtin_nontemporal_load(reinterpret_cast<const ln_u32x2_t*>(x16 + row * d + c));
      const uint2 h = make_uint2(hh[0], hh[1]);
      v[j] = make_float4(bf2f((bf16_t)(h.x & 0xFFFF)), bf2f((bf16_t)(h.x >> 16)), bf2f((bf16_t)(h.y & 0xFFFF)), bf2f((bf16_t)(h.y >> 16)));
    } else if (XF == 2) {
      v[j] = x24_load(x16 + row * d + c, x8 + row * d + c);
    } else if (NTX) {
      typedef float lf32x4_t __attribute__((ext_vector_type(4)));
      const lf32x4_t t4 = __builtin_nontemporal_load(reinterpret_cast<const lf32x4_t*>(xr + c));
      v[j] = make_float4(t4[0], t4[1], t4[2], t4[3]);
    } else {
      v[j] = *reinterpret_cast<const float4*>(xr + c);
    }
    if (ADD) {
      auto add4 = [&](const bf16_t* p) {
        const uint2 h = *reinterpret_cast<const uint2*>(p + row * d + c);
        v[j].x += bf2f((bf16_t)(h.x & 0xFFFF)); v[j].y += bf2f((bf16_t)(h.x >> 16));
        v[j].z += bf2f((bf16_t)(h.y & 0xFFFF)); v[j].w += bf2f((bf16_t)(h.y >> 16));
      };
      // (low branch planes exist in the split-bf16 mode only, MODE 1: without a runtime test in the other modes the
      //  vectors' loads are not separated by branches and all go out before the first use -- LayerNorm 63.2 -> 59.4 us
      //  on the 24-bit planes)
      add4(d_hi);
      if (MODE == 1 && d_lo != nullptr) add4(d_lo);
      if (ADD == 3) {
        add4(e_hi);
        if (MODE == 1 && e_lo != nullptr) add4(e_lo);
      }
      if (ADD != 2) {
        if (XH) __builtin_nontemporal_store((ln_u32x2_t{pack_bf2(v[j].x, v[j].y), pack_bf2(v[j].z, v[j].w)}), reinterpret_cast<ln_u32x2_t*>(x16 + row * d + c));
        else if (XF == 2) x24_store(x16 + row * d + c, x8 + row * d + c, v[j]);
        else if (NTX) {
          typedef float lf32x4_t __attribute__((ext_vector_type(4)));
          __builtin_nontemporal_store((lf32x4_t{v[j].x, v[j].y, v[j].z, v[j].w}), reinterpret_cast<lf32x4_t*>(xr + c));
        } else {
          // (MODE 0 / 1 have no fp32 LayerNorm output: out_f32 then names where the UPDATED residual row goes instead of
          //  back into x -- the training forward keeps every block's residual for the backward without copying it)
          float* xw = (MODE != 2 && out_f32 != nullptr) ? out_f32 + row * ldx : xr;
          *reinterpret_cast<float4*>(xw + c) = v[j];
        }
      }
    }
    s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
  }
  const float mean = wave_sum(s) / (float)d;
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    v[j].x -= mean; v[j].y -= mean; v[j].z -= mean; v[j].w -= mean;
    ss += (v[j].x * v[j].x + v[j].y * v[j].y) + (v[j].z * v[j].z + v[j].w * v[j].w);
  }
  const float rstd = rsqrtf(wave_sum(ss) / (float)d + 1e-5f);
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = 4 * lane + 256 * j;
    const float4 gm = *reinterpret_cast<const float4*>(gamma + c);
    const float4 bt = *reinterpret_cast<const float4*>(beta + c);
    float y[4] = {v[j].x * rstd * gm.x + bt.x, v[j].y * rstd * gm.y + bt.y, v[j].z * rstd * gm.z + bt.z,
                  v[j].w * rstd * gm.w + bt.w};
    if (MODE == 3) {
      // a 32-block of the row = the 4 values of 8 consecutive lanes: block maximum by three xor-shuffles
      float am = fmaxf(fmaxf(fabsf(y[0]), fabsf(y[1])), fmaxf(fabsf(y[2]), fabsf(y[3])));
      am = fmaxf(am, __shfl_xor(am, 1, 64));
      am = fmaxf(am, __shfl_xor(am, 2, 64));
      am = fmaxf(am, __shfl_xor(am, 4, 64));
      const uint32_t byte = mx8_scale_byte(am);
      *reinterpret_cast<uint32_t*>(out_q + row * d + c) = mx8_pack4(y[0], y[1], y[2], y[3], mx8_inv_scale(byte));
      if ((lane & 7) == 0) out_sc[mx8_scale_index(row, c >> 5, rows_pad)] = (uint8_t)byte;
    } else if (MODE == 2) {
      *reinterpret_cast<float4*>(out_f32 + row * d + c) = make_float4(y[0], y[1], y[2], y[3]);
    } else {
      bf16_t h[4], l[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (MODE == 1) split_bf(y[e], h[e], l[e]);
        else h[e] = f2bf(y[e]);
      }
      uint2 ph;
      ph.x = (uint32_t)h[0] | ((uint32_t)h[1] << 16);
      ph.y = (uint32_t)h[2] | ((uint32_t)h[3] << 16);
      *reinterpret_cast<uint2*>(out_hi + row * d + c) = ph;
      if (MODE == 1) {
        uint2 pl;
        pl.x = (uint32_t)l[0] | ((uint32_t)l[1] << 16);
        pl.y = (uint32_t)l[2] | ((uint32_t)l[3] << 16);
        *reinterpret_cast<uint2*>(out_lo + row * d + c) = pl;
      }
    }
  }
}

// generic widths (d % 64 == 0, small test models): scalar, three passes over an L1-resident row
template <int MODE, int ADD>
__global__ __launch_bounds__(256) void ln_generic_kernel(float* __restrict__ x, int64_t ldx,
                                                         const bf16_t* __restrict__ d_hi, const bf16_t* __restrict__ d_lo,
                                                         const bf16_t* __restrict__ e_hi, const bf16_t* __restrict__ e_lo,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, int64_t rows, int d,
                                                         bf16_t* out_hi, bf16_t* out_lo, float* out_f32) {
  const int tid = (int)__builtin_amdgcn_workitem_id_x();  // (threadIdx / blockIdx are device-library calls that do not
  const int lane = tid & 63;                              //  inline into a function without packed-fp32 ops)
  const int64_t row = (int64_t)__builtin_amdgcn_workgroup_id_x() * 4 + (tid >> 6);
  if (row >= rows) return;
  float* xr = x + row * ldx;
  auto val = [&](int c) {  // the row element with its pending branches (same order of additions as ln_vec_kernel)
    float t = xr[c];
    if (ADD) {
      t += bf2f(d_hi[row * d + c]);
      if (d_lo != nullptr) t += bf2f(d_lo[row * d + c]);
      if (ADD == 3) {
        t += bf2f(e_hi[row * d + c]);
        if (e_lo != nullptr) t += bf2f(e_lo[row * d + c]);
      }
    }
    return t;
  };
  float* xw = (MODE != 2 && out_f32 != nullptr) ? out_f32 + row * ldx : xr;  // (see ln_vec_kernel: the updated row's destination)
  if (ADD == 1 || ADD == 3) {  // each lane updates (and later re-reads) only its own elements
    for (int c = lane; c < d; c += 64) xw[c] = val(c);
  }
  auto cur = [&](int c) { return ADD == 2 ? val(c) : (ADD == 0 ? xr[c] : xw[c]); };
  float s = 0.f;
  for (int c = lane; c < d; c += 64) s += cur(c);
  const float mean = wave_sum(s) / (float)d;
  float ss = 0.f;
  for (int c = lane; c < d; c += 64) {
    const float t = cur(c) - mean;
    ss += t * t;
  }
  const float rstd = rsqrtf(wave_sum(ss) / (float)d + 1e-5f);
  // every lane has finished READING the row (the reductions above are wave-wide) before any write,
  // so out_f32 may alias x
  for (int c = lane; c < d; c += 64) {
    const float y = (cur(c) - mean) * rstd * gamma[c] + beta[c];
    if (MODE == 2) {
      out_f32[row * d + c] = y;
    } else if (MODE == 1) {
      bf16_t h, l;
      split_bf(y, h, l);
      out_hi[row * d + c] = h;
      out_lo[row * d + c] = l;
    } else {
      out_hi[row * d + c] = f2bf(y);
    }
  }
}

template <int MODE, int ADD>
hipError_t launch_mode(float* x, int64_t ldx, const bf16_t* dh, const bf16_t* dl, const bf16_t* eh, const bf16_t* el, const float* gamma,
                       const float* beta, int64_t rows, int32_t d, bf16_t* hi, bf16_t* lo, float* f32, hipStream_t s, uint8_t* q = nullptr,
                       uint8_t* qs = nullptr, int64_t rows_pad = 0, bf16_t* x16 = nullptr, bool ntx = false) {
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  if (x16 != nullptr) {  // 16-bit residual stream: MXFP8 output only
    if constexpr (MODE == 3) {
      if (d % 256 != 0 || d / 256 > 4) return hipErrorInvalidValue;
      switch (d / 256) {
        case 1: hipLaunchKernelGGL((ln_vec_kernel<3, 1, ADD, 1>), grid, block, 0, s, nullptr, 0, x16, nullptr, dh, dl, eh, el, nullptr, nullptr, gamma, beta, rows, d, hi, lo, f32, q, qs, rows_pad); break;
        case 2: hipLaunchKernelGGL((ln_vec_kernel<3, 2, ADD, 1>), grid, block, 0, s, nullptr, 0, x16, nullptr, dh, dl, eh, el, nullptr, nullptr, gamma, beta, rows, d, hi, lo, f32, q, qs, rows_pad); break;
        case 3: hipLaunchKernelGGL((ln_vec_kernel<3, 3, ADD, 1>), grid, block, 0, s, nullptr, 0, x16, nullptr, dh, dl, eh, el, nullptr, nullptr, gamma, beta, rows, d, hi, lo, f32, q, qs, rows_pad); break;
        default: hipLaunchKernelGGL((ln_vec_kernel<3, 4, ADD, 1>), grid, block, 0, s, nullptr, 0, x16, nullptr, dh, dl, eh, el, nullptr, nullptr, gamma, beta, rows, d, hi, lo, f32, q, qs, rows_pad); break;
      }
      return hipGetLastError();
    } else {
      return hipErrorInvalidValue;
    }
  }
  if (ntx && MODE == 0 && d % 256 == 0 && d / 256 <= 4 && ldx % 4 == 0) {
    if constexpr (MODE == 0) {
      switch (d / 256) {
        case 1: hipLaunchKernelGGL((ln_vec_kernel<0, 1, ADD, 0, false, true>), grid, block, 0, s, x, ldx, nullptr, nullptr, dh, dl, eh, el, nullptr, nullptr, gamma, beta, rows, d, hi, lo, f32, q, qs, rows_pad); break;
        case 2: hipLaunchKernelGGL((ln_vec_kernel<0, 2, ADD, 0, false, true>), grid, block, 0, s, x, ldx, nullptr, nullptr, dh, dl, eh, el, nullptr, nullptr, gamma, beta, rows, d, hi, lo, f32, q, qs, rows_pad); break;
        case 3: hipLaunchKernelGGL((ln_vec_kernel<0, 3, ADD, 0, false, true>), grid, block, 0, s, x, ldx, nullptr, nullptr, dh, dl, eh, el, nullptr, nullptr, gamma, beta, rows, d, hi, lo, f32, q, qs, rows_pad); break;
        default: hipLaunchKernelGGL((ln_vec_kernel<0, 4, ADD, 0, false, true>), grid, block, 0, s, x, ldx, nullptr, nullptr, dh, dl, eh, el, nullptr, nullptr, gamma, beta, rows, d, hi, lo, f32, q, qs, rows_pad); break;
      }
    }
    return hipGetLastError();
  }
  if (d % 256 == 0 && d / 256 <= 4 && ldx % 4 == 0) {
    switch (d / 256) {
      case 1: hipLaunchKernelGGL((ln_vec_kernel<MODE, 1, ADD>), grid, block, 0, s, x, ldx, nullptr, nullptr, dh, dl, eh, el, nullptr, nullptr, gamma, beta, rows, d, hi, lo, f32, q, qs, rows_pad); break;
      case 2: hipLaunchKernelGGL((ln_vec_kernel<MODE, 2, ADD>), grid, block, 0, s, x, ldx, nullptr, nullptr, dh, dl, eh, el, nullptr, nullptr, gamma, beta, rows, d, hi, lo, f32, q, qs, rows_pad); break;
      case 3: hipLaunchKernelGGL((ln_vec_kernel<MODE, 3, ADD>), grid, block, 0, s, x, ldx, nullptr, nullptr, dh, dl, eh, el, nullptr, nullptr, gamma, beta, rows, d, hi, lo, f32, q, qs, rows_pad); break;
      default: hipLaunchKernelGGL((ln_vec_kernel<MODE, 4, ADD>), grid, block, 0, s, x, ldx, nullptr, nullptr, dh, dl, eh, el, nullptr, nullptr, gamma, beta, rows, d, hi, lo, f32, q, qs, rows_pad); break;
    }
  } else {
    if constexpr (MODE == 3) return hipErrorInvalidValue;  // MXFP8 output: widths 256 .. 1024 only
    else hipLaunchKernelGGL((ln_generic_kernel<MODE, ADD>), grid, block, 0, s, x, ldx, dh, dl, eh, el, gamma, beta, rows, d, hi, lo, f32);
  }
  return hipGetLastError();
}

template <int MODE>
hipError_t launch_add_mode(int add, float* x, const bf16_t* dh, const bf16_t* dl, const bf16_t* eh, const bf16_t* el, const float* gamma,
                           const float* beta, int64_t rows, int32_t d, bf16_t* hi, bf16_t* lo, hipStream_t s, uint8_t* q = nullptr,
                           uint8_t* qs = nullptr, int64_t rows_pad = 0, bf16_t* x16 = nullptr, bool ntx = false, float* x_wb = nullptr) {
  if (x_wb != nullptr && (MODE > 1 || ntx || x16 != nullptr || (add != 1 && add != 3))) return hipErrorInvalidValue;
  switch (add) {
    case 0: return launch_mode<MODE, 0>(x, d, nullptr, nullptr, nullptr, nullptr, gamma, beta, rows, d, hi, lo, nullptr, s, q, qs, rows_pad, x16, ntx);
    case 1: return launch_mode<MODE, 1>(x, d, dh, dl, nullptr, nullptr, gamma, beta, rows, d, hi, lo, x_wb, s, q, qs, rows_pad, x16, ntx);
    case 2: return launch_mode<MODE, 2>(x, d, dh, dl, nullptr, nullptr, gamma, beta, rows, d, hi, lo, nullptr, s, q, qs, rows_pad, x16, ntx);
    case 3: return launch_mode<MODE, 3>(x, d, dh, dl, eh, el, gamma, beta, rows, d, hi, lo, x_wb, s, q, qs, rows_pad, x16, ntx);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace

hipError_t launch_layernorm(const float* x, int64_t ldx, const float* gamma, const float* beta, int64_t rows,
                            int32_t d, bf16_t* out_hi, bf16_t* out_lo, float* out_f32, hipStream_t s, bool stream_x) {
  if (rows <= 0 || d <= 0 || d % 64 != 0) return hipErrorInvalidValue;
  float* xm = const_cast<float*>(x);  // not written without ADD
  if (out_f32 != nullptr) return launch_mode<2, 0>(xm, ldx, nullptr, nullptr, nullptr, nullptr, gamma, beta, rows, d, nullptr, nullptr, out_f32, s);
  if (out_lo != nullptr) return launch_mode<1, 0>(xm, ldx, nullptr, nullptr, nullptr, nullptr, gamma, beta, rows, d, out_hi, out_lo, nullptr, s);
  return launch_mode<0, 0>(xm, ldx, nullptr, nullptr, nullptr, nullptr, gamma, beta, rows, d, out_hi, nullptr, nullptr, s, nullptr, nullptr, 0, nullptr, stream_x);
}

hipError_t launch_add_layernorm(float* x, const bf16_t* delta_hi, const bf16_t* delta_lo, const float* gamma,
                                const float* beta, int64_t rows, int32_t d, bf16_t* out_hi, bf16_t* out_lo,
                                hipStream_t s, float* x_wb) {
  if (rows <= 0 || d <= 0 || d % 64 != 0 || delta_hi == nullptr) return hipErrorInvalidValue;
  if (x_wb == nullptr) return launch_add_layernorm_ex(1, x, delta_hi, delta_lo, nullptr, nullptr, gamma, beta, rows, d, out_hi, out_lo, s, false);
  if (out_lo != nullptr) return launch_add_mode<1>(1, x, delta_hi, delta_lo, nullptr, nullptr, gamma, beta, rows, d, out_hi, out_lo, s, nullptr, nullptr, 0, nullptr, false, x_wb);
  return launch_add_mode<0>(1, x, delta_hi, nullptr, nullptr, nullptr, gamma, beta, rows, d, out_hi, nullptr, s, nullptr, nullptr, 0, nullptr, false, x_wb);
}

// add: 1 = x += d1 (written back); 2 = normalise x + d1 without writing x back; 3 = x += d1 + d2 (written back)
hipError_t launch_add_layernorm_ex(int add, float* x, const bf16_t* d1_hi, const bf16_t* d1_lo, const bf16_t* d2_hi,
                                   const bf16_t* d2_lo, const float* gamma, const float* beta, int64_t rows, int32_t d,
                                   bf16_t* out_hi, bf16_t* out_lo, hipStream_t s, bool stream_x) {
  if (rows <= 0 || d <= 0 || d % 64 != 0 || add < 1 || add > 3 || d1_hi == nullptr || (add == 3 && d2_hi == nullptr)) return hipErrorInvalidValue;
  if (out_lo != nullptr) return launch_add_mode<1>(add, x, d1_hi, d1_lo, d2_hi, d2_lo, gamma, beta, rows, d, out_hi, out_lo, s);
  return launch_add_mode<0>(add, x, d1_hi, nullptr, d2_hi, nullptr, gamma, beta, rows, d, out_hi, nullptr, s, nullptr, nullptr, 0, nullptr, stream_x);
}

// ---- 24-bit residual stream (XF = 2; the image tower's bf16 / IEEE-half modes).  bf16 output only.
//   pre != 0: src_f32 [rows, d] (row stride ld_src) -> ln_pre -> residual planes (xhi, xlo) -> LayerNorm(gamma, beta) -> out
//   add = 1, 2, 3 as launch_add_layernorm_ex, on the planes
namespace {
template <int ADD, bool PRE>
hipError_t launch_x24_t(const float* src, int64_t ld_src, bf16_t* xhi, uint8_t* xlo, const bf16_t* d1, const bf16_t* d2, const float* gp,
                        const float* bp, const float* gamma, const float* beta, int64_t rows, int32_t d, bf16_t* out, hipStream_t s) {
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  float* x = const_cast<float*>(src);
  switch (d / 256) {
    case 1: hipLaunchKernelGGL((ln_vec_kernel<0, 1, ADD, 2, PRE>), grid, block, 0, s, x, ld_src, xhi, xlo, d1, nullptr, d2, nullptr, gp, bp, gamma, beta, rows, d, out, nullptr, nullptr, nullptr, nullptr, 0); break;
    case 2: hipLaunchKernelGGL((ln_vec_kernel<0, 2, ADD, 2, PRE>), grid, block, 0, s, x, ld_src, xhi, xlo, d1, nullptr, d2, nullptr, gp, bp, gamma, beta, rows, d, out, nullptr, nullptr, nullptr, nullptr, 0); break;
    case 3: hipLaunchKernelGGL((ln_vec_kernel<0, 3, ADD, 2, PRE>), grid, block, 0, s, x, ld_src, xhi, xlo, d1, nullptr, d2, nullptr, gp, bp, gamma, beta, rows, d, out, nullptr, nullptr, nullptr, nullptr, 0); break;
    default: hipLaunchKernelGGL((ln_vec_kernel<0, 4, ADD, 2, PRE>), grid, block, 0, s, x, ld_src, xhi, xlo, d1, nullptr, d2, nullptr, gp, bp, gamma, beta, rows, d, out, nullptr, nullptr, nullptr, nullptr, 0); break;
  }
  return hipGetLastError();
}
}  // namespace

bool layernorm_x24_supports(int32_t d) { return d % 256 == 0 && d >= 256 && d <= 1024; }

hipError_t launch_layernorm_x24(int add, int pre, const float* src_f32, int64_t ld_src, bf16_t* xhi, uint8_t* xlo, const bf16_t* d1,
                                const bf16_t* d2, const float* gamma_pre, const float* beta_pre, const float* gamma, const float* beta,
                                int64_t rows, int32_t d, bf16_t* out, hipStream_t s) {
  if (rows <= 0 || !layernorm_x24_supports(d) || !xhi || !xlo || !out || !gamma || !beta) return hipErrorInvalidValue;
  if (pre) {
    if (add != 0 || !src_f32 || !gamma_pre || !beta_pre || ld_src % 4 != 0) return hipErrorInvalidValue;
    return launch_x24_t<0, true>(src_f32, ld_src, xhi, xlo, nullptr, nullptr, gamma_pre, beta_pre, gamma, beta, rows, d, out, s);
  }
  if ((add >= 1 && !d1) || (add == 3 && !d2)) return hipErrorInvalidValue;
  switch (add) {
    case 0: return launch_x24_t<0, false>(nullptr, 0, xhi, xlo, nullptr, nullptr, nullptr, nullptr, gamma, beta, rows, d, out, s);
    case 1: return launch_x24_t<1, false>(nullptr, 0, xhi, xlo, d1, nullptr, nullptr, nullptr, gamma, beta, rows, d, out, s);
    case 2: return launch_x24_t<2, false>(nullptr, 0, xhi, xlo, d1, nullptr, nullptr, nullptr, gamma, beta, rows, d, out, s);
    case 3: return launch_x24_t<3, false>(nullptr, 0, xhi, xlo, d1, d2, nullptr, nullptr, gamma, beta, rows, d, out, s);
    default: return hipErrorInvalidValue;
  }
}

// CLS rows of the 24-bit residual (+ the pending c_proj branch) -> fp32 [B, D] for the pool / ln_post / proj kernel
namespace {
__global__ void gather_cls24_kernel(const bf16_t* __restrict__ xhi, const uint8_t* __restrict__ xlo, const bf16_t* __restrict__ delta,
                                    int tokens, int D, int64_t total, float* __restrict__ out) {
  const int64_t i = (int64_t)__builtin_amdgcn_workgroup_id_x() * 256 + __builtin_amdgcn_workitem_id_x();
  if (i >= total) return;
  const int64_t b = i / D;
  const int64_t src = b * tokens * (int64_t)D + (i - b * D);
  const float v = __uint_as_float(((uint32_t)xhi[src] << 16) | ((uint32_t)xlo[src] << 8));
  out[i] = v + (delta ? bf2f(delta[src]) : 0.f);
}
}  // namespace

hipError_t launch_gather_cls24(const bf16_t* xhi, const uint8_t* xlo, const bf16_t* delta, int32_t B, int32_t tokens, int32_t D, float* out,
                               hipStream_t s) {
  const int64_t total = (int64_t)B * D;
  hipLaunchKernelGGL(gather_cls24_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, xhi, xlo, delta, tokens, D, total, out);
  return hipGetLastError();
}

// MXFP8 output (the A operand of the fp8 path's QKV / c_fc GEMMs): out_q [rows, d] e4m3, out_sc [d/64][rows_pad][2].
// add: 0 = plain LayerNorm; 1, 2, 3 as launch_add_layernorm_ex.
hipError_t launch_layernorm_mx8(int add, float* x, bf16_t* x16, const bf16_t* d1_hi, const bf16_t* d2_hi, const float* gamma,
                                const float* beta, int64_t rows, int32_t d, uint8_t* out_q, uint8_t* out_sc, int64_t rows_pad, hipStream_t s) {
  if (rows <= 0 || d <= 0 || d % 256 != 0 || d > 1024 || rows_pad < rows || !out_q || !out_sc || (!x && !x16)) return hipErrorInvalidValue;
  if ((add >= 1 && d1_hi == nullptr) || (add == 3 && d2_hi == nullptr)) return hipErrorInvalidValue;
  return launch_add_mode<3>(add, x, d1_hi, nullptr, d2_hi, nullptr, gamma, beta, rows, d, nullptr, nullptr, s, out_q, out_sc, rows_pad, x16);
}

}  // namespace tapclip

TAPCLIP_TU_NO_PK_F32_END
