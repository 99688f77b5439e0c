// Small-M MFMA GEMM for gfx950, built for LATENCY:  C[M,N] = A[M,K] . W[N,K]^T (+ bias, fused epilogues) at M of a few
// hundred to a few thousand rows -- the text tower (reference call sites
// models/model_wrapper.py:58,72) once the tied padding rows are merged (tied.hip): 65 classes x 24..26 distinct rows =
// 1 560..1 690 rows against N = 512..2 048, K = 512..2 048 -- and the backward's dX GEMMs at the same sizes.
//
// At these sizes a GEMM is not MFMA-bound and not HBM-bound: it is one tile per CU and what it costs is the launch, the
// first operand bytes' way to the LDS, a chain of K / 32 dependent k-steps and the epilogue.  The persistent 256 x 256
// kernel (gemm256.hip) spends 9.6 us of fixed cost + 0.59 us per k-step there (19 us at K = 512 with 42 of 256 CUs busy),
// the register-staged 128 x 128 kernel (gemm.hip) ~40 us.  Here:
//  * one output tile per 256-thread workgroup (4 waves as 2 (m) x 2 (n)), the tile shape chosen per launch from
//    {128 x 128, 128 x 64, 64 x 64} so that the launch is ONE round of <= 256 workgroups with the fewest operand bytes
//    per CU ((BM + BN) K 2 B: what a CU must pull through its ~60-100 GB/s L2 port is the loop's floor);
//  * operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4, no staging registers) through a deep ring of
//    32-deep k-steps -- 8 stages of 8-16 KiB: K = 256 of both operands in flight from the first cycle on -- with
//    hand-counted s_waitcnt vmcnt(N) and one s_barrier per k-step;
//  * split-bf16 (SPLIT: the parity mode's text tower) stages A_hi, A_lo, W_hi, W_lo of a k-step ONCE and issues the three
//    MFMA products on the resident fragments (hi.hi + lo.hi + hi.lo) -- not three passes over K as the tiled kernels do
//    (VERDICT r03 item 1): 2x the bytes and 3x the MFMAs of one product per k-step, K / 32 dependent steps instead of
//    3 K / 32.
// LDS image, swizzle and fragment reads are gemm256.hip's (64-B rows, 16-B chunk p of row R holds source k-chunk
// p ^ (3 * ((R >> 3) & 1)), applied on the DMA's source address and again on the read side); the MFMA is issued swapped
// (D = Wfrag . Afrag^T) so that a lane holds 4 consecutive n of one m.
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace tapclip {
namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

constexpr int BKS = 32;

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int EPI, bool SPLIT, int BM, int BN, int NS>
__global__ __launch_bounds__(256, 1) void gemm_lat_kernel(GemmArgs g) {
  constexpr int PL = SPLIT ? 2 : 1;                 // operand planes per stage
  constexpr int A_BYTES = BM * BKS * 2, W_BYTES = BN * BKS * 2;
  constexpr int STAGE = PL * (A_BYTES + W_BYTES);   // [A_hi | W_hi | A_lo | W_lo]
  constexpr int AP = BM / 64, WP = BN / 64;         // 1-KiB pieces (16 rows x 64 B) per wave per plane
  constexpr int NDMA = PL * (AP + WP);              // LDS-DMA instructions per wave per stage
  constexpr int MI = BM / 32, NJ = BN / 32;         // 16 x 16 MFMA tiles per wave: MI (m) x NJ (n)
  static_assert((NS - 1) * NDMA <= 63, "vmcnt is a 6-bit field");
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;

  const int tiles_n = g.N / BN;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;  // n fastest: neighbours share their A rows
  const int64_t m0 = (int64_t)tm * BM;
  const int n0 = tn * BN;
  const int KS = g.K / BKS;

  // LDS-DMA source offsets (bytes): piece j = rows 16 j .. 16 j + 15 of a plane; wave w issues pieces w, w + 4, ...;
  // lane l covers row 16 j + (l >> 2), LDS chunk (l & 3) <- source chunk (l & 3) ^ (3 * ((l >> 5) & 1))
  const int src_chunk = (lane & 3) ^ (3 * ((lane >> 5) & 1));
  uint32_t a_off[AP], w_off[WP];
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    int64_t m = m0 + 16 * (wave + 4 * i) + (lane >> 2);
    if (m >= g.M) m = g.M - 1;  // clamp: rows past M are computed, never stored
    a_off[i] = (uint32_t)((m * g.lda + src_chunk * 8) * 2);
  }
#pragma unroll
  for (int i = 0; i < WP; ++i) w_off[i] = (uint32_t)(((int64_t)(n0 + 16 * (wave + 4 * i) + (lane >> 2)) * g.K + src_chunk * 8) * 2);

  auto stage_dma = [&](int st, int ks) {
    uint8_t* base = smem + st * STAGE;
#pragma unroll
    for (int p = 0; p < PL; ++p) {
      const uint8_t* Ap = reinterpret_cast<const uint8_t*>(p ? g.A_lo : g.A_hi) + ks * (BKS * 2);
      const uint8_t* Wp = reinterpret_cast<const uint8_t*>(p ? g.W_lo : g.W_hi) + ks * (BKS * 2);
      uint8_t* pb = base + p * (A_BYTES + W_BYTES);
#pragma unroll
      for (int i = 0; i < AP; ++i)
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(Ap + a_off[i]), (lds_void_t*)(pb + (wave + 4 * i) * 1024), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < WP; ++i)
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(Wp + w_off[i]), (lds_void_t*)(pb + A_BYTES + (wave + 4 * i) * 1024), 16, 0, 0);
    }
  };

  // accumulators start at the bias and take the k-steps in ascending order, the three split products of a k-step in the
  // order hi.hi, A_lo.W_hi, A_hi.W_lo per 32-deep step -- exactly what gemm256.hip does, so a row's bits do not depend on
  // which of THESE TWO kernels its launch's row count selects (the batch-invariance tests compare across that boundary).
  // (gemm.hip, the fallback of TAPCLIP_GEMM_LAT=0 / TAPCLIP_GEMM_TILE=128 / operands past the 32-bit offset limit, stages
  // 64-deep slices and runs each product over k[0:64] before the next: another fp32 summation order, equal to rounding only.)
  // (the bias is loaded BEFORE the ring's first DMA is issued: vmcnt retires in order, and a load behind the prologue's
  // stages would make its first use wait for all of them)
  f32x4_t acc[NJ][MI];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    f32x4_t bv = {0.f, 0.f, 0.f, 0.f};
    if (EPI != EPI_PATCH_F32 && g.bias != nullptr) bv = *reinterpret_cast<const f32x4_t*>(g.bias + n0 + wn * (BN / 2) + j * 16 + 4 * q);
#pragma unroll
    for (int i = 0; i < MI; ++i) acc[j][i] = bv;
  }

  // prologue: the first NS - 1 k-steps in flight at once
#pragma unroll
  for (int i = 0; i < NS - 1; ++i)
    if (i < KS) stage_dma(i, i);

  const int frag_off = r * 64 + ((q ^ (3 * ((r >> 3) & 1))) << 4);
  const int a_base = (wm * (BM / 2)) * 64 + frag_off;
  const int w_base = A_BYTES + (wn * (BN / 2)) * 64 + frag_off;

  for (int ks = 0; ks < KS; ++ks) {
    // stage ks has landed once at most the stages issued after it are outstanding: in[ks] = min(KS, ks + NS - 1) - 1 - ks of them
    {
      const int later = (KS - 1 - ks) < (NS - 2) ? (KS - 1 - ks) : (NS - 2);
      switch (later) {  // (vmcnt takes an immediate)
        case 0: wait_vm<0>(); break;
        case 1: wait_vm<1 * NDMA>(); break;
        case 2: wait_vm<2 * NDMA>(); break;
        case 3: wait_vm<(NS > 4 ? 3 : 2) * NDMA>(); break;
        case 4: wait_vm<(NS > 5 ? 4 : 2) * NDMA>(); break;
        case 5: wait_vm<(NS > 6 ? 5 : 2) * NDMA>(); break;
        default: wait_vm<(NS - 2) * NDMA>(); break;
      }
    }
    __builtin_amdgcn_s_barrier();  // every wave's pieces of stage ks are in LDS; every wave has read stage ks - 1 into registers
    asm volatile("" ::: "memory");
    if (ks + NS - 1 < KS) stage_dma((ks + NS - 1) % NS, ks + NS - 1);  // into the slot of step ks - 1
    const uint8_t* base = smem + (ks % NS) * STAGE;
    bf16x8_t wf[PL][NJ], af[PL][MI];
#pragma unroll
    for (int p = 0; p < PL; ++p) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) wf[p][j] = *reinterpret_cast<const bf16x8_t*>(base + p * (A_BYTES + W_BYTES) + w_base + j * 1024);
#pragma unroll
      for (int i = 0; i < MI; ++i) af[p][i] = *reinterpret_cast<const bf16x8_t*>(base + p * (A_BYTES + W_BYTES) + a_base + i * 1024);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        acc[j][i] = TAPCLIP_MFMA_16x16x32(wf[0][j], af[0][i], acc[j][i]);
        if (SPLIT) {
          acc[j][i] = TAPCLIP_MFMA_16x16x32(wf[0][j], af[1][i], acc[j][i]);  // W_hi . A_lo
          acc[j][i] = TAPCLIP_MFMA_16x16x32(wf[1][j], af[0][i], acc[j][i]);  // W_lo . A_hi
        }
      }
  }

  // ---- epilogue: lane holds D[n = 4 q + e][m = r] of each 16 x 16 tile (e = 0..3)
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int64_t m = m0 + wm * (BM / 2) + i * 16 + r;
    if (m >= g.M) continue;
    int64_t orow = m;
    const float* addrow = nullptr;
    if (EPI == EPI_PATCH_F32) {
      const int64_t b = m / g.rows_per_group;
      const int p = (int)(m - b * g.rows_per_group);
      orow = b * (g.rows_per_group + 1) + 1 + p;
      addrow = g.add_table + (int64_t)(1 + p) * g.N;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int n = n0 + wn * (BN / 2) + j * 16 + 4 * q;
      f32x4_t v = acc[j][i];
      if (EPI == EPI_PATCH_F32) {
        const float4 pv = *reinterpret_cast<const float4*>(addrow + n);
        v[0] += pv.x; v[1] += pv.y; v[2] += pv.z; v[3] += pv.w;
      }
      if (EPI == EPI_BIAS_GELU_BF16) {
        if (g.act == 0) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = SPLIT ? gelu_erf(v[e]) : gelu_fast16(v[e]);  // see common.h
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = SPLIT ? gelu_quick(v[e]) : gelu_quick_fast(v[e]);
        }
      }
      if (EPI == EPI_GELU_BWD_BF16) {
        const int64_t o = orow * g.ldo + n;
        const uint2 uh = *reinterpret_cast<const uint2*>(g.aux_hi + o);  // the 4 upstream gradients of this lane
        float up[4] = {bf2f((bf16_t)(uh.x & 0xFFFF)), bf2f((bf16_t)(uh.x >> 16)), bf2f((bf16_t)(uh.y & 0xFFFF)), bf2f((bf16_t)(uh.y >> 16))};
        if (SPLIT) {
          const uint2 ul = *reinterpret_cast<const uint2*>(g.aux_lo + o);
          up[0] += bf2f((bf16_t)(ul.x & 0xFFFF)); up[1] += bf2f((bf16_t)(ul.x >> 16));
          up[2] += bf2f((bf16_t)(ul.y & 0xFFFF)); up[3] += bf2f((bf16_t)(ul.y >> 16));
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (g.act == 0 ? (SPLIT ? gelu_erf_grad(v[e]) : gelu_fit_grad(v[e])) : gelu_quick_grad(v[e])) * up[e];
      }
      if (EPI == EPI_BIAS_BF16 || EPI == EPI_BIAS_GELU_BF16 || EPI == EPI_GELU_BWD_BF16) {
        bf16_t h[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (SPLIT) split_bf(v[e], h[e], l[e]);
          else h[e] = f2bf(v[e]);
        }
        uint2 ph;
        ph.x = (uint32_t)h[0] | ((uint32_t)h[1] << 16);
        ph.y = (uint32_t)h[2] | ((uint32_t)h[3] << 16);
        *reinterpret_cast<uint2*>(g.out_hi + orow * g.ldo + n) = ph;
        if (SPLIT) {
          uint2 pl;
          pl.x = (uint32_t)l[0] | ((uint32_t)l[1] << 16);
          pl.y = (uint32_t)l[2] | ((uint32_t)l[3] << 16);
          *reinterpret_cast<uint2*>(g.out_lo + orow * g.ldo + n) = pl;
        }
      } else {
        float4* dst = reinterpret_cast<float4*>(g.out_f32 + orow * g.ldo + n);
        if (EPI == EPI_BIAS_RESID_F32) {
          const float4 rv = *dst;
          v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
        }
        *dst = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  }
}

template <int EPI, bool SPLIT, int BM, int BN>
hipError_t launch_t(const GemmArgs& a, hipStream_t s) {
  constexpr int NS = SPLIT ? ((BM + BN) >= 256 ? 4 : 6) : 8;  // <= 144 KiB of ring (split 128 x 64: 6 stages of 24 KiB; every other shape <= 128 KiB)
  constexpr int smem_bytes = NS * (SPLIT ? 2 : 1) * (BM + BN) * BKS * 2;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_lat_kernel<EPI, SPLIT, BM, BN, NS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int64_t nwg = ((a.M + BM - 1) / BM) * (a.N / BN);
  hipLaunchKernelGGL((gemm_lat_kernel<EPI, SPLIT, BM, BN, NS>), dim3((unsigned)nwg), dim3(256), smem_bytes, s, a);
  return hipGetLastError();
}

template <int EPI, bool SPLIT>
hipError_t launch_e(const GemmArgs& a, hipStream_t s) {
  // the tile with the least (rounds of 256 workgroups) x (time of one tile); near-ties go to the larger tile.  Round 5: the
  // time of a tile is a measured line in K -- 5.8 + 0.0120 K us (128 x 128), 3.1 + 0.0096 K (128 x 64), 1.6 + 0.0044 K (64 x 64)
  // at one 16-bit product (tools/gemm_sweep.sh at 1 576 rows, profiles/r05_gemm_small_batch_sweep.log) -- where rounds x
  // (operand bytes + a constant) chose 128 x 128 for the two rounds of c_fc at batch 8 (30 us; 64 x 64: five rounds, 26 us).
  // The three split-bf16 products scale every term alike: the same choice.
  static const int forced = [] {
    const char* e = getenv("TAPCLIP_GEMM_LAT_TILE");  // experiments: 0 = 128x128, 1 = 128x64, 2 = 64x64
    return e ? atoi(e) : -1;
  }();
  const int bm[3] = {128, 128, 64}, bn[3] = {128, 64, 64};
  int best = -1;
  double best_cost = 0;
  for (int c = 0; c < 3; ++c) {
    if (a.N % bn[c] != 0) continue;
    const int64_t tiles = ((a.M + bm[c] - 1) / bm[c]) * (a.N / bn[c]);
    const double fix[3] = {5.8, 3.1, 1.6}, per_k[3] = {0.0120, 0.0096, 0.0044};
    const double cost = (double)((tiles + 255) / 256) * (fix[c] + per_k[c] * a.K);
    if (best < 0 || cost < 0.95 * best_cost) best = c, best_cost = cost;
  }
  if (forced >= 0 && forced < 3 && a.N % bn[forced] == 0) best = forced;
  switch (best) {
    case 0: return launch_t<EPI, SPLIT, 128, 128>(a, s);
    case 1: return launch_t<EPI, SPLIT, 128, 64>(a, s);
    case 2: return launch_t<EPI, SPLIT, 64, 64>(a, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace

bool gemm_lat_supports(const GemmArgs& a) {
  return a.M > 0 && a.N % 64 == 0 && a.K % BKS == 0 && a.K >= BKS && (a.lda % 8) == 0 && (a.ldo % 4) == 0 &&
         (a.M * a.lda * 2 < (int64_t)0xFFFF0000) && ((int64_t)a.N * a.K * 2 < (int64_t)0xFFFF0000);
}

hipError_t launch_gemm_lat(const GemmArgs& a, int epilogue, bool split, hipStream_t s) {
  if (!gemm_lat_supports(a)) return hipErrorInvalidValue;
#define TAPCLIP_LAT_CASE(E) \
  case E:                   \
    return split ? launch_e<E, true>(a, s) : launch_e<E, false>(a, s);
  switch (epilogue) {
    TAPCLIP_LAT_CASE(EPI_BIAS_BF16)
    TAPCLIP_LAT_CASE(EPI_BIAS_GELU_BF16)
    TAPCLIP_LAT_CASE(EPI_BIAS_RESID_F32)
    TAPCLIP_LAT_CASE(EPI_PATCH_F32)
    TAPCLIP_LAT_CASE(EPI_BIAS_F32)
    TAPCLIP_LAT_CASE(EPI_GELU_BWD_BF16)
    default:
      return hipErrorInvalidValue;
  }
#undef TAPCLIP_LAT_CASE
}

}  // namespace tapclip
