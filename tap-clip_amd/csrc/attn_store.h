// Output stores shared by the attention kernels (attention.hip, attention_long.hip): one query row's head slice per lane group.
#pragma once
#include "common.h"
#include "kernels.h"

namespace tapclip {
namespace {

// Stores of one query row's head slice: lane (r, g) holds O[q][16 g + 4 dt + e] / sum for dt = 0..3, i.e. 16
// consecutive columns.  bf16: 32 bytes per lane.
template <bool SPLIT>
__device__ __forceinline__ void store_o_bf16(const AttnArgs& a, const f32x4_t (&oc)[4], float inv, int64_t row, int head, int g) {
  const int64_t off = row * a.D + head * 64 + 16 * g;
  uint32_t wh[8], wl[8];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    if (SPLIT) {
      bf16_t h[4], l[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) split_bf(oc[dt][e] * inv, h[e], l[e]);
      wh[2 * dt] = (uint32_t)h[0] | ((uint32_t)h[1] << 16);
      wh[2 * dt + 1] = (uint32_t)h[2] | ((uint32_t)h[3] << 16);
      wl[2 * dt] = (uint32_t)l[0] | ((uint32_t)l[1] << 16);
      wl[2 * dt + 1] = (uint32_t)l[2] | ((uint32_t)l[3] << 16);
    } else {
      wh[2 * dt] = pack_bf2(oc[dt][0] * inv, oc[dt][1] * inv);
      wh[2 * dt + 1] = pack_bf2(oc[dt][2] * inv, oc[dt][3] * inv);
    }
  }
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  u32x4_t* dh = reinterpret_cast<u32x4_t*>(a.out_hi + off);
  // (non-temporal stores make this kernel faster ALONE -- 73.3 -> 67.6 us in tools/gemm_bench -- and slower in the
  // tower, 79 -> 85 us with out_proj + 1.5 us behind it: there the output buffer is cache-resident from the previous
  // block and its consumer reads it from the cache.  Plain stores.)
  dh[0] = u32x4_t{wh[0], wh[1], wh[2], wh[3]};
  dh[1] = u32x4_t{wh[4], wh[5], wh[6], wh[7]};
  if (SPLIT) {
    u32x4_t* dl = reinterpret_cast<u32x4_t*>(a.out_lo + off);
    dl[0] = u32x4_t{wl[0], wl[1], wl[2], wl[3]};
    dl[1] = u32x4_t{wl[4], wl[5], wl[6], wl[7]};
  }
}
// MXFP8: the head's 64 columns are two 32-blocks, block b held by the lanes g = 2 b, 2 b + 1 of the row.  Called by
// all lanes (the shuffle needs them); `valid` masks the stores.
__device__ __forceinline__ void store_o_mx8(const AttnArgs& a, const f32x4_t (&oc)[4], float inv, int64_t row, int head, int g,
                                            bool valid) {
  float am = 0.f;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int e = 0; e < 4; ++e) am = fmaxf(am, fabsf(oc[dt][e] * inv));
  am = xor16_max(am);
  const uint32_t byte = mx8_scale_byte(am);
  if (!valid) return;
  const float is = mx8_inv_scale(byte);
  uint4 pk;
  pk.x = mx8_pack4(oc[0][0] * inv, oc[0][1] * inv, oc[0][2] * inv, oc[0][3] * inv, is);
  pk.y = mx8_pack4(oc[1][0] * inv, oc[1][1] * inv, oc[1][2] * inv, oc[1][3] * inv, is);
  pk.z = mx8_pack4(oc[2][0] * inv, oc[2][1] * inv, oc[2][2] * inv, oc[2][3] * inv, is);
  pk.w = mx8_pack4(oc[3][0] * inv, oc[3][1] * inv, oc[3][2] * inv, oc[3][3] * inv, is);
  *reinterpret_cast<uint4*>(a.out_q + row * a.D + head * 64 + 16 * g) = pk;
  if ((g & 1) == 0) a.out_q_scale[((size_t)head * a.out_m_pad + row) * 2 + (g >> 1)] = (uint8_t)byte;
}

}  // namespace
}  // namespace tapclip
