// Backward kernels of the prompt-tuning step (reference train.py:99-105: loss.backward() reaches only
// `prompt_learner.context_bank.*` and `logit_scale`; every CLIP weight is frozen, clip_wrapper.py:19-20).
// The gradient therefore flows  logits -> text features -> token -1 of the text transformer ->
// activations only (dX, never dW) -> the scaled context tokens.  The image tower needs no backward.
//
// GEMM dgrads re-use gemm*.hip with transposed packed weights; this file holds what is not a GEMM:
//   attn_bwd_kernel      softmax-attention backward per (sequence, head): fp32 operands in LDS, the five
//                        products on the exact f32-input MFMA (the text tower is 65 x 93 tokens)
//   ln_bwd_kernel        LayerNorm backward, fused with the residual-gradient accumulation
//   pool_project_bwd     d(token pick -> @ text_projection -> L2 norm)   (model_wrapper.py:73-75)
//   logits_bwd           d(scale * img . txt^T) w.r.t. txt and log-scale  (model_wrapper.py:79)
//   pack_transpose       W[N,K] fp32 -> W^T[K,N] bf16 hi (+ lo)
#include <type_traits>

#ifndef TAPCLIP_AB_KEEP_PK  // (tools/Makefile ab_pk: the A/B build that measured what this costs)
#define TAPCLIP_TU_NO_PK_F32  // common.h: no packed-fp32 VALU ops in this translation unit -- the MI355X op_sel erratum
#endif
#include "common.h"
#include "kernels.h"

namespace tapclip {
namespace {

__device__ __forceinline__ float ld_bf(const bf16_t* hi, const bf16_t* lo, int64_t i) {
  float v = bf2f(hi[i]);
  if (lo != nullptr) v += bf2f(lo[i]);
  return v;
}
__device__ __forceinline__ void st_bf(bf16_t* hi, bf16_t* lo, int64_t i, float v) {
  if (lo != nullptr) {
    bf16_t h, l;
    split_bf(v, h, l);
    hi[i] = h;
    lo[i] = l;
  } else {
    hi[i] = f2bf(v);
  }
}

// ---- attention backward.  One 512-thread workgroup per (sequence, head); T <= 96 (the text tower's
// prompt_len + 77).  The arithmetic is fp32: the five products run on the exact f32-input MFMA
// (v_mfma_f32_16x16x4_f32: A[row = l & 15][k = l >> 4], B[k = l >> 4][col = l & 15], one float per lane per
// operand), so bf16 and bf16x3 towers share one kernel and the result is fp32-accurate.
//   S = q k^T (q carries the folded 1/sqrt(64)),  P = softmax(S),  dV = P^T dO,
//   dP = dO v^T,  delta_i = sum_d dO_id O_id,  dS = P (dP - delta),  dq = dS k,  dk = dS^T q.
// LDS: TWO operand buffers [Tp][LDO] (Tp = T rounded up to 16, pad rows zero), staged three times -- (q, k) for S,
// (dO, v) for dV and dS, (q, k) again for dq and dk: the re-reads come from L2 -- plus P / dS [Tp][Tp+1] and delta[Tp].
// H16 (no low planes, i.e. the bf16 tower): the operands stay 16-bit in LDS and are widened when read, 63 KB for
// T = 93, so TWO workgroups share a CU; with fp32 operands (bf16x3: hi + lo summed) it is 88 KB and one.  (Round 2:
// all four operands resident as fp32 were 137 KB -- 520 workgroups of the 65 x 8 text tower at one per CU are
// 2.03 rounds that cost three: 123 us per launch.)
constexpr int BWD_LD = 65;    // fp32 operand row (conflict-free row and column walks)
constexpr int BWD_LD16 = 66;  // 16-bit operand row: 33 dwords

__device__ __forceinline__ f32x4_t mfma4(float a, float b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---- 16-bit operand fragments of v_mfma_f32_16x16x32 (lane (lr, lq) holds A[row lr][k = 8 lq + j] / B[k = 8 lq + j][col lr],
// j = 0..7) for the H16 variant of the kernel below.
typedef uint32_t bwd_u32x4_t __attribute__((ext_vector_type(4)));
// eight consecutive elements of one LDS row (4-byte aligned: the 16-bit operand rows are 33 dwords)
__device__ __forceinline__ bf16x8_t frag_row16(const bf16_t* p) {
  const uint32_t* w = reinterpret_cast<const uint32_t*>(p);
  const bwd_u32x4_t v = {w[0], w[1], w[2], w[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}
// eight elements of one LDS COLUMN (rows k0 .. k0 + 7 at stride ld); rows >= kmax read as zero
__device__ __forceinline__ bf16x8_t frag_col16(const bf16_t* base, int k0, int ld, int col, int kmax) {
  s16x8_t v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (k0 + j < kmax) ? (short)base[(k0 + j) * ld + col] : (short)0;
  return __builtin_bit_cast(bf16x8_t, v);
}
// eight fp32 values as a (hi, lo) pair of 16-bit fragments: hi + lo carries 16 significand bits of each
__device__ __forceinline__ void frag_split16(const float (&v)[8], bf16x8_t& hi, bf16x8_t& lo) {
  s16x8_t h, l;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    bf16_t hh, ll;
    split_bf(v[j], hh, ll);
    h[j] = (short)hh;
    l[j] = (short)ll;
  }
  hi = __builtin_bit_cast(bf16x8_t, h);
  lo = __builtin_bit_cast(bf16x8_t, l);
}

template <int NT, bool H16>  // NT = Tp / 16 key / query tiles: compile-time trip counts, so the LDS reads of a product pipeline
__global__ __launch_bounds__(512) void attn_bwd_kernel(AttnBwdArgs a) {
  extern __shared__ float sh[];
  const int T = a.T, D = a.D;
  constexpr int Tp = NT * 16, nt = NT, LP = Tp + 1;
  constexpr int NW = 8;  // waves
  constexpr int LDO = H16 ? BWD_LD16 : BWD_LD;
  typedef typename std::conditional<H16, bf16_t, float>::type op_t;
  float* P = sh;  // [Tp][Tp+1]
  float* delta = P + Tp * LP;
  op_t* bufA = reinterpret_cast<op_t*>(delta + Tp);
  op_t* bufB = bufA + Tp * LDO;
  auto ld = [](const op_t* b, int idx) -> float {
    if constexpr (H16) return bf2f(b[idx]);
    else return b[idx];
  };
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int seq = blockIdx.x / a.H, head = blockIdx.x - seq * a.H;
  const int64_t row0 = (int64_t)seq * T;
  const int64_t ldq = 3 * (int64_t)D;
  const int qcol = head * 64, kcol = D + head * 64, vcol = 2 * D + head * 64;

  // Staging of one [T, 64] slice (hi [+ lo] planes, row stride `stride`, first column `col`) into an operand buffer, 8
  // elements (16 B per plane) per thread and step, in two halves: the global loads into registers (issued early, so
  // that they fly under the previous phase's arithmetic) and the LDS writes.
  constexpr int NSTEP = (Tp * 8 + 511) / 512;
  struct Staged { uint4 h[NSTEP], l[NSTEP]; };
  auto stage_load = [&](Staged& r, const bf16_t* hi, const bf16_t* lo, int64_t stride, int col) {
#pragma unroll
    for (int it = 0; it < NSTEP; ++it) {
      const int e = tid + 512 * it, i = e >> 3, d0 = (e & 7) * 8;
      r.h[it] = make_uint4(0, 0, 0, 0);
      r.l[it] = make_uint4(0, 0, 0, 0);
      if (e < Tp * 8 && i < T) {
        const int64_t g = (row0 + i) * stride + col + d0;
        r.h[it] = *reinterpret_cast<const uint4*>(hi + g);
        if (!H16 && lo != nullptr) r.l[it] = *reinterpret_cast<const uint4*>(lo + g);
      }
    }
  };
  auto stage_store = [&](op_t* dst, const Staged& r) {
#pragma unroll
    for (int it = 0; it < NSTEP; ++it) {
      const int e = tid + 512 * it, i = e >> 3, d0 = (e & 7) * 8;
      if (e >= Tp * 8) continue;
      const uint32_t hw[4] = {r.h[it].x, r.h[it].y, r.h[it].z, r.h[it].w}, lw[4] = {r.l[it].x, r.l[it].y, r.l[it].z, r.l[it].w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if constexpr (H16) {
          *reinterpret_cast<uint32_t*>(dst + i * LDO + d0 + 2 * u) = hw[u];  // (LDO and d0 are even: 4-byte aligned)
        } else {
          // (bf2f(0) + bf2f(0) = 0: absent low planes and pad rows need no special case)
          dst[i * LDO + d0 + 2 * u] = bf2f((bf16_t)(hw[u] & 0xFFFF)) + bf2f((bf16_t)(lw[u] & 0xFFFF));
          dst[i * LDO + d0 + 2 * u + 1] = bf2f((bf16_t)(hw[u] >> 16)) + bf2f((bf16_t)(lw[u] >> 16));
        }
      }
    }
  };

  Staged ra, rb;
  stage_load(ra, a.qkv_hi, a.qkv_lo, ldq, qcol);  // q
  stage_load(rb, a.qkv_hi, a.qkv_lo, ldq, kcol);  // k
  {  // delta_i = <dO_i, O_i>: one wave per row, both read straight from global; all of a wave's rows in flight at once
    constexpr int RPW = Tp / NW;
    float sd[RPW];
#pragma unroll
    for (int u = 0; u < RPW; ++u) {
      const int i = wave + NW * u;
      sd[u] = 0.f;
      if (i < T) {
        const int64_t g = (row0 + i) * D + head * 64 + lane;
        sd[u] = ld_bf(a.dout_hi, a.dout_lo, g) * ld_bf(a.out_hi, a.out_lo, g);
      }
    }
    stage_store(bufA, ra);
    stage_store(bufB, rb);
#pragma unroll
    for (int u = 0; u < RPW; ++u) {
      const float t = wave_sum(sd[u]);
      if (lane == 0) delta[wave + NW * u] = t;
    }
  }
  __syncthreads();

  // S[i][j] = q_i . k_j   (tile (ti, tj): A = q rows, B = k rows, K = 64)
  for (int tile = wave; tile < nt * nt; tile += NW) {
    const int ti = tile / nt, tj = tile - ti * nt;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    if constexpr (H16) {
      // q and k are exact 16-bit values: two K = 32 MFMAs give the same products as the sixteen f32 ones (fp32 accumulation)
      const bf16_t* qa = bufA + (ti * 16 + lr) * LDO + 8 * lq;
      const bf16_t* kb = bufB + (tj * 16 + lr) * LDO + 8 * lq;
      const bf16x8_t a0 = frag_row16(qa), a1 = frag_row16(qa + 32), b0 = frag_row16(kb), b1 = frag_row16(kb + 32);
      acc = TAPCLIP_MFMA_16x16x32(a0, b0, acc);
      acc = TAPCLIP_MFMA_16x16x32(a1, b1, acc);
    } else {
      const op_t* qa = bufA + (ti * 16 + lr) * LDO + lq;
      const op_t* kb = bufB + (tj * 16 + lr) * LDO + lq;
      float av[16], bv[16];  // all operands of the tile first, then the MFMA chain: one LDS wait instead of sixteen
#pragma unroll
      for (int u = 0; u < 16; ++u) { av[u] = ld(qa, 4 * u); bv[u] = ld(kb, 4 * u); }
#pragma unroll
      for (int u = 0; u < 16; ++u) acc = mfma4(av[u], bv[u], acc);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = ti * 16 + 4 * lq + e, j = tj * 16 + lr;
      float sv = acc[e];
      if (j >= T || (a.causal && j > i)) sv = -INFINITY;
      if (j == T - 1) sv += a.last_key_bias;  // tied padding (tied.hip): the forward counted the last key m times, ln m here
      P[i * LP + j] = sv;
    }
  }
  stage_load(ra, a.dout_hi, a.dout_lo, D, head * 64);  // dO  (in flight under the barrier and the softmax)
  stage_load(rb, a.qkv_hi, a.qkv_lo, ldq, vcol);       // v
  __syncthreads();  // (q, k are free from here on)
  // row softmax (pad query rows see all-finite scores of zero vectors; their dO is zero, so they never count): four
  // rows per wave at a time, 16 lanes and NT elements per lane each, reductions by four xor-shuffles inside the 16-lane
  // groups (one row per wave with 64-lane reductions was 19 us of the kernel)
  {
    const int g = lane >> 4, l16 = lane & 15;
    constexpr int RPW = Tp / NW;
#pragma unroll
    for (int b = 0; b < (RPW + 3) / 4; ++b) {
      const int u = 4 * b + g;
      const bool on = u < RPW;
      const int i = on ? wave + NW * u : 0;
      float pv[NT], mx = -INFINITY;
#pragma unroll
      for (int m = 0; m < NT; ++m) {
        pv[m] = on ? P[i * LP + l16 + 16 * m] : 0.f;
        mx = fmaxf(mx, pv[m]);
      }
#pragma unroll
      for (int o = 8; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 16));
      float sum = 0.f;
#pragma unroll
      for (int m = 0; m < NT; ++m) {
        pv[m] = __expf(pv[m] - mx);
        sum += pv[m];
      }
#pragma unroll
      for (int o = 8; o >= 1; o >>= 1) sum += __shfl_xor(sum, o, 16);
      const float inv = 1.0f / sum;
      if (on) {
#pragma unroll
        for (int m = 0; m < NT; ++m) P[i * LP + l16 + 16 * m] = pv[m] * inv;
      }
    }
  }
  stage_store(bufA, ra);
  stage_store(bufB, rb);
  __syncthreads();
  // dV[j][d] = sum_i P[i][j] dO[i][d]   (tile (tj, td): A[row j][k i] = P[i][j], B[k i][col d] = dO[i][d], K = Tp)
  for (int tile = wave; tile < nt * 4; tile += NW) {
    const int tj = tile >> 2, td = tile & 3;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    if constexpr (H16) {
      // A = P^T as a (hi, lo) pair of 16-bit fragments (P is fp32: 2^-17 per element after the split), B = dO (exact)
#pragma unroll
      for (int k0 = 0; k0 < Tp; k0 += 32) {
        float pv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) pv[j] = (k0 + 8 * lq + j < Tp) ? P[(k0 + 8 * lq + j) * LP + tj * 16 + lr] : 0.f;
        bf16x8_t ah, al;
        frag_split16(pv, ah, al);
        const bf16x8_t b = frag_col16(bufA, k0 + 8 * lq, LDO, td * 16 + lr, Tp);
        acc = TAPCLIP_MFMA_16x16x32(ah, b, acc);
        acc = TAPCLIP_MFMA_16x16x32(al, b, acc);
      }
    } else {
      float av[Tp / 4], bv[Tp / 4];
#pragma unroll
      for (int u = 0; u < Tp / 4; ++u) { av[u] = P[(4 * u + lq) * LP + tj * 16 + lr]; bv[u] = ld(bufA, (4 * u + lq) * LDO + td * 16 + lr); }
#pragma unroll
      for (int u = 0; u < Tp / 4; ++u) acc = mfma4(av[u], bv[u], acc);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int j = tj * 16 + 4 * lq + e;
      if (j < T) st_bf(a.dqkv_hi, a.dqkv_lo, (row0 + j) * ldq + vcol + td * 16 + lr, acc[e]);
    }
  }
  stage_load(ra, a.qkv_hi, a.qkv_lo, ldq, qcol);  // q  (in flight under dS)
  stage_load(rb, a.qkv_hi, a.qkv_lo, ldq, kcol);  // k
  __syncthreads();
  // dS[i][j] = P[i][j] (dO_i . v_j - delta_i), in place
  for (int tile = wave; tile < nt * nt; tile += NW) {
    const int ti = tile / nt, tj = tile - ti * nt;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    if constexpr (H16) {
      const bf16_t* oa = bufA + (ti * 16 + lr) * LDO + 8 * lq;
      const bf16_t* vb = bufB + (tj * 16 + lr) * LDO + 8 * lq;
      const bf16x8_t a0 = frag_row16(oa), a1 = frag_row16(oa + 32), b0 = frag_row16(vb), b1 = frag_row16(vb + 32);
      acc = TAPCLIP_MFMA_16x16x32(a0, b0, acc);
      acc = TAPCLIP_MFMA_16x16x32(a1, b1, acc);
    } else {
      const op_t* oa = bufA + (ti * 16 + lr) * LDO + lq;
      const op_t* vb = bufB + (tj * 16 + lr) * LDO + lq;
      float av[16], bv[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) { av[u] = ld(oa, 4 * u); bv[u] = ld(vb, 4 * u); }
#pragma unroll
      for (int u = 0; u < 16; ++u) acc = mfma4(av[u], bv[u], acc);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = ti * 16 + 4 * lq + e, j = tj * 16 + lr;
      P[i * LP + j] *= (acc[e] - delta[i]);
    }
  }
  __syncthreads();  // (dO, v are free)
  stage_store(bufA, ra);
  stage_store(bufB, rb);
  __syncthreads();
  // dq[i][d] = sum_j dS[i][j] k[j][d];   dk[j][d] = sum_i dS[i][j] q[i][d]
  for (int tile = wave; tile < nt * 4 * 2; tile += NW) {
    const bool is_dk = tile >= nt * 4;
    const int tl = is_dk ? tile - nt * 4 : tile;
    const int tr = tl >> 2, td = tl & 3;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    if constexpr (H16) {
      // A = dS (dq) or dS^T (dk) as (hi, lo) 16-bit fragments, B = k or q (exact)
#pragma unroll
      for (int k0 = 0; k0 < Tp; k0 += 32) {
        float pv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int kk = k0 + 8 * lq + j;
          pv[j] = kk < Tp ? (is_dk ? P[kk * LP + tr * 16 + lr] : P[(tr * 16 + lr) * LP + kk]) : 0.f;
        }
        bf16x8_t ah, al;
        frag_split16(pv, ah, al);
        const bf16x8_t b = frag_col16(is_dk ? bufA : bufB, k0 + 8 * lq, LDO, td * 16 + lr, Tp);
        acc = TAPCLIP_MFMA_16x16x32(ah, b, acc);
        acc = TAPCLIP_MFMA_16x16x32(al, b, acc);
      }
    } else {
      float av[Tp / 4], bv[Tp / 4];
      if (!is_dk) {
#pragma unroll
        for (int u = 0; u < Tp / 4; ++u) { av[u] = P[(tr * 16 + lr) * LP + 4 * u + lq]; bv[u] = ld(bufB, (4 * u + lq) * LDO + td * 16 + lr); }
      } else {
#pragma unroll
        for (int u = 0; u < Tp / 4; ++u) { av[u] = P[(4 * u + lq) * LP + tr * 16 + lr]; bv[u] = ld(bufA, (4 * u + lq) * LDO + td * 16 + lr); }
      }
#pragma unroll
      for (int u = 0; u < Tp / 4; ++u) acc = mfma4(av[u], bv[u], acc);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int rr = tr * 16 + 4 * lq + e;
      if (rr < T) st_bf(a.dqkv_hi, a.dqkv_lo, (row0 + rr) * ldq + (is_dk ? kcol : qcol) + td * 16 + lr, acc[e]);
    }
  }
}

// ---- LayerNorm backward fused with the residual-gradient add: dres[row] += dLN/dx (dy), one wave per row.
// y = (x - mean) rstd gamma + beta;  g = dy gamma;  dx = rstd (g - mean(g) - xhat mean(g xhat))
// pk_hi (pk_lo): optional 16-bit plane(s) of the UPDATED dres, [rows, d] -- the next GEMM's A operand (its own pack pass
// over dres was 7.9 us x 2 per block at 6 045 x 512)
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ dy, int64_t rows, int d, float* dres,
                                                     bf16_t* __restrict__ pk_hi, bf16_t* __restrict__ pk_lo) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * d;
  const float* gr = dy + row * d;
  // the row lives in registers (d <= 1024: at most 16 elements per lane; the launcher checks): one read of x, dy and
  // gamma instead of four passes over them (13.4 -> 12.4 us at 6 045 x 512: 48 MB of traffic either way)
  constexpr int MAXE = 16;
  float xv[MAXE], gv[MAXE];
  float s = 0.f;
#pragma unroll
  for (int u = 0; u < MAXE; ++u) {
    const int c = lane + 64 * u;
    xv[u] = c < d ? xr[c] : 0.f;
    gv[u] = c < d ? gr[c] * gamma[c] : 0.f;
    s += xv[u];
  }
  const float mean = wave_sum(s) / (float)d;
  float ss = 0.f;
#pragma unroll
  for (int u = 0; u < MAXE; ++u) {
    const float t = (lane + 64 * u < d) ? xv[u] - mean : 0.f;
    xv[u] = t;
    ss += t * t;
  }
  const float rstd = rsqrtf(wave_sum(ss) / (float)d + 1e-5f);
  float sg = 0.f, sgx = 0.f;
#pragma unroll
  for (int u = 0; u < MAXE; ++u) {
    xv[u] *= rstd;  // xhat
    sg += gv[u];
    sgx += gv[u] * xv[u];
  }
  sg = wave_sum(sg) / (float)d;
  sgx = wave_sum(sgx) / (float)d;
#pragma unroll
  for (int u = 0; u < MAXE; ++u) {
    const int c = lane + 64 * u;
    if (c < d) {
      const float v = dres[row * d + c] + rstd * (gv[u] - sg - xv[u] * sgx);
      dres[row * d + c] = v;
      if (pk_lo != nullptr) {
        bf16_t h, l;
        split_bf(v, h, l);
        pk_hi[row * d + c] = h;
        pk_lo[row * d + c] = l;
      } else if (pk_hi != nullptr) {
        pk_hi[row * d + c] = f2bf(v);
      }
    }
  }
}

// ---- d(pool -> project -> normalise): hidden row p (token tok of sequence n), y = p W, t = y/|y|.
// dy = (dt - t <t, dt>) / |y|;  dp = dy W^T, written into d_hidden[n, tok, :] (other rows are zeroed by the caller).
__global__ __launch_bounds__(256) void pool_project_bwd_kernel(const float* __restrict__ hidden, int tokens, int K,
                                                               int tok_fixed, const float* __restrict__ proj, int E,
                                                               int normalize, const float* __restrict__ dt,
                                                               float* d_hidden) {
  extern __shared__ float sh[];  // row[K] | y[E] | dy[E] | red[8]
  float* row = sh;
  float* y = sh + K;
  float* dy = y + E;
  float* red = dy + E;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t n = blockIdx.x;
  int tok = tok_fixed;
  if (tok < 0) tok += tokens;
  const int64_t roff = (n * tokens + tok) * (int64_t)K;
  for (int c = tid; c < K; c += 256) row[c] = hidden[roff + c];
  __syncthreads();
  // (eight independent partial sums: one dependent chain of K loads from L2 was 25 us of this kernel's 63)
  for (int e = tid; e < E; e += 256) {
    float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int kk = 0;
    for (; kk + 8 <= K; kk += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) s8[u] = fmaf(row[kk + u], proj[(int64_t)(kk + u) * E + e], s8[u]);
    }
    for (; kk < K; ++kk) s8[0] = fmaf(row[kk], proj[(int64_t)kk * E + e], s8[0]);
    y[e] = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
  }
  __syncthreads();
  if (normalize) {
    float ss = 0.f, sd = 0.f;
    for (int e = tid; e < E; e += 256) {
      ss += y[e] * y[e];
      sd += y[e] * dt[n * E + e];
    }
    ss = wave_sum(ss);
    sd = wave_sum(sd);
    if (lane == 0) {
      red[wave] = ss;
      red[4 + wave] = sd;
    }
    __syncthreads();
    const float nrm2 = red[0] + red[1] + red[2] + red[3];
    const float ydt = red[4] + red[5] + red[6] + red[7];
    const float inv = 1.0f / sqrtf(nrm2);
    for (int e = tid; e < E; e += 256) dy[e] = (dt[n * E + e] - y[e] * ydt / nrm2) * inv;
  } else {
    for (int e = tid; e < E; e += 256) dy[e] = dt[n * E + e];
  }
  __syncthreads();
  // dp[c] = <dy, proj[c, :]>: a thread per row, 16-byte loads of its row, eight independent partial sums.  (Measured: 52.6 us
  // for the kernel against 63.2 with one dependent chain; 16 lanes per row + a 4-step xor reduction 60.3; a wave per row
  // 155.  With one 256-thread block per sequence the kernel is bound by its CU's L2 path: every block reads the whole
  // projection matrix twice.)
  for (int c = tid; c < K; c += 256) {
    const float* pr = proj + (int64_t)c * E;
    float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int e = 0;
    if ((E & 7) == 0 && (reinterpret_cast<uintptr_t>(pr) & 15) == 0) {
      for (; e + 8 <= E; e += 8) {
        const float4 p0 = *reinterpret_cast<const float4*>(pr + e), p1 = *reinterpret_cast<const float4*>(pr + e + 4);
        s8[0] = fmaf(dy[e], p0.x, s8[0]); s8[1] = fmaf(dy[e + 1], p0.y, s8[1]);
        s8[2] = fmaf(dy[e + 2], p0.z, s8[2]); s8[3] = fmaf(dy[e + 3], p0.w, s8[3]);
        s8[4] = fmaf(dy[e + 4], p1.x, s8[4]); s8[5] = fmaf(dy[e + 5], p1.y, s8[5]);
        s8[6] = fmaf(dy[e + 6], p1.z, s8[6]); s8[7] = fmaf(dy[e + 7], p1.w, s8[7]);
      }
    }
    for (; e < E; ++e) s8[0] = fmaf(dy[e], pr[e], s8[0]);
    d_hidden[roff + c] = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
  }
}

// ---- logits = scale * img txt^T:  d_txt[c,e] = scale * sum_b dl[b,c] img[b,e];  d(log scale) = sum dl * logits
__global__ __launch_bounds__(256) void logits_bwd_txt_kernel(const float* __restrict__ dl, const float* __restrict__ img,
                                                             float scale, int B, int C, int E, float* d_txt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)C * E) return;
  const int c = (int)(i / E), e = (int)(i - (int64_t)c * E);
  // (eight independent partial sums instead of one dependent chain of B loads: 55 -> ~12 us at B = 256)
  float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int b = 0;
  for (; b + 8 <= B; b += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) s8[u] = fmaf(dl[(int64_t)(b + u) * C + c], img[(int64_t)(b + u) * E + e], s8[u]);
  }
  for (; b < B; ++b) s8[0] = fmaf(dl[(int64_t)b * C + c], img[(int64_t)b * E + e], s8[0]);
  d_txt[i] = scale * (((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7])));
}
__global__ __launch_bounds__(256) void dot_reduce_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         int64_t n, float* out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 256) s = fmaf(a[i], b[i], s);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (red[0] + red[1]) + (red[2] + red[3]);
}

template <bool SPLIT>
__global__ void pack_transpose_kernel(const float* __restrict__ src, int64_t N, int K, int64_t scale_rows, float scale,
                                      bf16_t* hi, bf16_t* lo) {
  // dst[k][n] = src[n][k] (rows n < scale_rows scaled first)
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * K) return;
  const int64_t kk = i / N, n = i - kk * N;
  float v = src[n * K + kk];
  if (n < scale_rows) v *= scale;
  if (SPLIT) {
    bf16_t h, l;
    split_bf(v, h, l);
    hi[i] = h;
    lo[i] = l;
  } else {
    hi[i] = f2bf(v);
  }
}

__global__ void fill_zero_kernel(float* p, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0.f;
}

inline unsigned nblk(int64_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

namespace {
size_t attn_bwd_lds(int T, bool h16) {
  const int Tp = (T + 15) & ~15;
  return (size_t)(Tp * (Tp + 1) + Tp) * sizeof(float) + (h16 ? (size_t)2 * Tp * BWD_LD16 * 2 : (size_t)2 * Tp * BWD_LD * sizeof(float));
}
}  // namespace
size_t attn_bwd_lds_bytes(int T) { return attn_bwd_lds(T, false); }  // the larger (fp32-operand) variant

template <int NT, bool H16>
hipError_t launch_attn_bwd_t(const AttnBwdArgs& a, hipStream_t s) {
  static bool attr = false;
  const size_t lds = attn_bwd_lds(NT * 16, H16);
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_kernel<NT, H16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return e;
    attr = true;
  }
  hipLaunchKernelGGL((attn_bwd_kernel<NT, H16>), dim3((unsigned)(a.n_seq * a.H)), dim3(512), lds, s, a);
  return hipGetLastError();
}

hipError_t launch_attention_bwd(const AttnBwdArgs& a, hipStream_t s) {
  if (a.T <= 0 || a.T > 96 || a.D != a.H * 64 || a.n_seq <= 0 || a.D % 8 != 0) return hipErrorInvalidValue;
  if (attn_bwd_lds_bytes(a.T) > 160 * 1024) return hipErrorInvalidValue;
  // 16-bit operands in LDS when no tensor has a low plane (the bf16 tower)
  const bool h16 = a.qkv_lo == nullptr && a.dout_lo == nullptr;
#define TAPCLIP_ABWD_CASE(N) \
  case N: return h16 ? launch_attn_bwd_t<N, true>(a, s) : launch_attn_bwd_t<N, false>(a, s);
  switch ((a.T + 15) / 16) {
    TAPCLIP_ABWD_CASE(1)
    TAPCLIP_ABWD_CASE(2)
    TAPCLIP_ABWD_CASE(3)
    TAPCLIP_ABWD_CASE(4)
    TAPCLIP_ABWD_CASE(5)
    default: return h16 ? launch_attn_bwd_t<6, true>(a, s) : launch_attn_bwd_t<6, false>(a, s);
  }
#undef TAPCLIP_ABWD_CASE
}

hipError_t launch_ln_bwd(const float* x, const float* gamma, const float* dy, int64_t rows, int32_t d, float* dres,
                         bf16_t* pk_hi, bf16_t* pk_lo, hipStream_t s) {
  if (rows <= 0 || d <= 0 || d > 1024 || (pk_lo != nullptr && pk_hi == nullptr)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(ln_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, gamma, dy, rows, d, dres, pk_hi, pk_lo);
  return hipGetLastError();
}

hipError_t launch_pool_project_bwd(const float* hidden, int64_t n, int32_t tokens, int32_t K, int32_t tok,
                                   const float* proj, int32_t E, int32_t normalize, const float* dt, float* d_hidden,
                                   hipStream_t s) {
  hipLaunchKernelGGL(fill_zero_kernel, dim3(nblk(n * tokens * K)), dim3(256), 0, s, d_hidden, n * tokens * K);
  hipLaunchKernelGGL(pool_project_bwd_kernel, dim3((unsigned)n), dim3(256), (K + 2 * E + 8) * sizeof(float), s, hidden,
                     tokens, K, tok, proj, E, normalize, dt, d_hidden);
  return hipGetLastError();
}

hipError_t launch_logits_bwd(const float* dl, const float* logits, const float* img, float scale, int32_t B, int32_t C,
                             int32_t E, float* d_txt, float* d_logscale, hipStream_t s) {
  hipLaunchKernelGGL(logits_bwd_txt_kernel, dim3(nblk((int64_t)C * E)), dim3(256), 0, s, dl, img, scale, B, C, E, d_txt);
  if (d_logscale) hipLaunchKernelGGL(dot_reduce_kernel, dim3(1), dim3(256), 0, s, dl, logits, (int64_t)B * C, d_logscale);
  return hipGetLastError();
}

hipError_t launch_pack_transpose(const float* src, int64_t N, int32_t K, int64_t scale_rows, float scale, bf16_t* hi,
                                 bf16_t* lo, hipStream_t s) {
  if (lo) hipLaunchKernelGGL((pack_transpose_kernel<true>), dim3(nblk(N * K)), dim3(256), 0, s, src, N, K, scale_rows, scale, hi, lo);
  else hipLaunchKernelGGL((pack_transpose_kernel<false>), dim3(nblk(N * K)), dim3(256), 0, s, src, N, K, scale_rows, scale, hi, lo);
  return hipGetLastError();
}

}  // namespace tapclip
TAPCLIP_TU_NO_PK_F32_END
