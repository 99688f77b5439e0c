// LayerNorm (eps 1e-5, affine) as a wavefront reduction: one 64-lane wave per row, fp32 statistics.
// Replaces the ln_pre / ln_1 / ln_2 / ln_post nn.LayerNorm calls inside open_clip's blocks
// (SURVEY.md section 2.1 K2).  HBM-bound: reads the fp32 residual row once, writes the bf16 GEMM
// operand (hi, and lo for bf16x3) or fp32 (ln_pre in place, unit API).
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
// XH: the residual stream itself is 16-bit (x16, row stride d) instead of fp32 x -- the fp8 precision only, where
// a 2^-9 rounding of x per block is far below the MXFP8 operand error and the LayerNorms are 20 % of the step.
template <int MODE, int NV, int ADD, bool XH = false>  // NV float4 per lane: d = 256 * NV
__global__ __launch_bounds__(256) void ln_vec_kernel(float* __restrict__ x, int64_t ldx, bf16_t* __restrict__ x16,
                                                     const bf16_t* __restrict__ d_hi, const bf16_t* __restrict__ d_lo,
                                                     const bf16_t* __restrict__ e_hi, const bf16_t* __restrict__ e_lo,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, int64_t rows, int d,
                                                     bf16_t* out_hi, bf16_t* out_lo, float* out_f32,
                                                     uint8_t* out_q, uint8_t* out_sc, int64_t rows_pad) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float* xr = XH ? nullptr : x + row * ldx;
  float4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = 4 * lane + 256 * j;
    if (XH) {
      const uint2 h = *reinterpret_cast<const uint2*>(x16 + row * d + c);
      v[j] = make_float4(bf2f((bf16_t)(h.x & 0xFFFF)), bf2f((bf16_t)(h.x >> 16)), bf2f((bf16_t)(h.y & 0xFFFF)), bf2f((bf16_t)(h.y >> 16)));
    } else {
      v[j] = *reinterpret_cast<const float4*>(xr + c);
    }
    if (ADD) {
      auto add4 = [&](const bf16_t* p) {
        const uint2 h = *reinterpret_cast<const uint2*>(p + row * d + c);
        v[j].x += bf2f((bf16_t)(h.x & 0xFFFF)); v[j].y += bf2f((bf16_t)(h.x >> 16));
        v[j].z += bf2f((bf16_t)(h.y & 0xFFFF)); v[j].w += bf2f((bf16_t)(h.y >> 16));
      };
      add4(d_hi);
      if (d_lo != nullptr) add4(d_lo);
      if (ADD == 3) {
        add4(e_hi);
        if (e_lo != nullptr) add4(e_lo);
      }
      if (ADD != 2) {
        if (XH) *reinterpret_cast<uint2*>(x16 + row * d + c) = make_uint2(pack_bf2(v[j].x, v[j].y), pack_bf2(v[j].z, v[j].w));
        else *reinterpret_cast<float4*>(xr + c) = v[j];
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
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
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
  if (ADD == 1 || ADD == 3) {  // each lane updates (and later re-reads) only its own elements
    for (int c = lane; c < d; c += 64) xr[c] = val(c);
  }
  auto cur = [&](int c) { return ADD == 2 ? val(c) : xr[c]; };
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
                       uint8_t* qs = nullptr, int64_t rows_pad = 0, bf16_t* x16 = nullptr) {
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  if (x16 != nullptr) {  // 16-bit residual stream: MXFP8 output only
    if constexpr (MODE == 3) {
      if (d % 256 != 0 || d / 256 > 4) return hipErrorInvalidValue;
      switch (d / 256) {
        case 1: hipLaunchKernelGGL((ln_vec_kernel<3, 1, ADD, true>), grid, block, 0, s, nullptr, 0, x16, dh, dl, eh, el, gamma, beta, rows, d, hi, lo, f32, q, qs, rows_pad); break;
        case 2: hipLaunchKernelGGL((ln_vec_kernel<3, 2, ADD, true>), grid, block, 0, s, nullptr, 0, x16, dh, dl, eh, el, gamma, beta, rows, d, hi, lo, f32, q, qs, rows_pad); break;
        case 3: hipLaunchKernelGGL((ln_vec_kernel<3, 3, ADD, true>), grid, block, 0, s, nullptr, 0, x16, dh, dl, eh, el, gamma, beta, rows, d, hi, lo, f32, q, qs, rows_pad); break;
        default: hipLaunchKernelGGL((ln_vec_kernel<3, 4, ADD, true>), grid, block, 0, s, nullptr, 0, x16, dh, dl, eh, el, gamma, beta, rows, d, hi, lo, f32, q, qs, rows_pad); break;
      }
      return hipGetLastError();
    } else {
      return hipErrorInvalidValue;
    }
  }
  if (d % 256 == 0 && d / 256 <= 4 && ldx % 4 == 0) {
    switch (d / 256) {
      case 1: hipLaunchKernelGGL((ln_vec_kernel<MODE, 1, ADD>), grid, block, 0, s, x, ldx, nullptr, dh, dl, eh, el, gamma, beta, rows, d, hi, lo, f32, q, qs, rows_pad); break;
      case 2: hipLaunchKernelGGL((ln_vec_kernel<MODE, 2, ADD>), grid, block, 0, s, x, ldx, nullptr, dh, dl, eh, el, gamma, beta, rows, d, hi, lo, f32, q, qs, rows_pad); break;
      case 3: hipLaunchKernelGGL((ln_vec_kernel<MODE, 3, ADD>), grid, block, 0, s, x, ldx, nullptr, dh, dl, eh, el, gamma, beta, rows, d, hi, lo, f32, q, qs, rows_pad); break;
      default: hipLaunchKernelGGL((ln_vec_kernel<MODE, 4, ADD>), grid, block, 0, s, x, ldx, nullptr, dh, dl, eh, el, gamma, beta, rows, d, hi, lo, f32, q, qs, rows_pad); break;
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
                           uint8_t* qs = nullptr, int64_t rows_pad = 0, bf16_t* x16 = nullptr) {
  switch (add) {
    case 0: return launch_mode<MODE, 0>(x, d, nullptr, nullptr, nullptr, nullptr, gamma, beta, rows, d, hi, lo, nullptr, s, q, qs, rows_pad, x16);
    case 1: return launch_mode<MODE, 1>(x, d, dh, dl, nullptr, nullptr, gamma, beta, rows, d, hi, lo, nullptr, s, q, qs, rows_pad, x16);
    case 2: return launch_mode<MODE, 2>(x, d, dh, dl, nullptr, nullptr, gamma, beta, rows, d, hi, lo, nullptr, s, q, qs, rows_pad, x16);
    case 3: return launch_mode<MODE, 3>(x, d, dh, dl, eh, el, gamma, beta, rows, d, hi, lo, nullptr, s, q, qs, rows_pad, x16);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace

hipError_t launch_layernorm(const float* x, int64_t ldx, const float* gamma, const float* beta, int64_t rows,
                            int32_t d, bf16_t* out_hi, bf16_t* out_lo, float* out_f32, hipStream_t s) {
  if (rows <= 0 || d <= 0 || d % 64 != 0) return hipErrorInvalidValue;
  float* xm = const_cast<float*>(x);  // not written without ADD
  if (out_f32 != nullptr) return launch_mode<2, 0>(xm, ldx, nullptr, nullptr, nullptr, nullptr, gamma, beta, rows, d, nullptr, nullptr, out_f32, s);
  if (out_lo != nullptr) return launch_mode<1, 0>(xm, ldx, nullptr, nullptr, nullptr, nullptr, gamma, beta, rows, d, out_hi, out_lo, nullptr, s);
  return launch_mode<0, 0>(xm, ldx, nullptr, nullptr, nullptr, nullptr, gamma, beta, rows, d, out_hi, nullptr, nullptr, s);
}

hipError_t launch_add_layernorm(float* x, const bf16_t* delta_hi, const bf16_t* delta_lo, const float* gamma,
                                const float* beta, int64_t rows, int32_t d, bf16_t* out_hi, bf16_t* out_lo,
                                hipStream_t s) {
  return launch_add_layernorm_ex(1, x, delta_hi, delta_lo, nullptr, nullptr, gamma, beta, rows, d, out_hi, out_lo, s);
}

// add: 1 = x += d1 (written back); 2 = normalise x + d1 without writing x back; 3 = x += d1 + d2 (written back)
hipError_t launch_add_layernorm_ex(int add, float* x, const bf16_t* d1_hi, const bf16_t* d1_lo, const bf16_t* d2_hi,
                                   const bf16_t* d2_lo, const float* gamma, const float* beta, int64_t rows, int32_t d,
                                   bf16_t* out_hi, bf16_t* out_lo, hipStream_t s) {
  if (rows <= 0 || d <= 0 || d % 64 != 0 || add < 1 || add > 3 || d1_hi == nullptr || (add == 3 && d2_hi == nullptr)) return hipErrorInvalidValue;
  if (out_lo != nullptr) return launch_add_mode<1>(add, x, d1_hi, d1_lo, d2_hi, d2_lo, gamma, beta, rows, d, out_hi, out_lo, s);
  return launch_add_mode<0>(add, x, d1_hi, nullptr, d2_hi, nullptr, gamma, beta, rows, d, out_hi, nullptr, s);
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
