// bf16 MFMA GEMM for gfx950:  C[M,N] = A[M,K] . W[N,K]^T  with fused epilogues.
//
// Replaces the nn.Linear / conv1 calls that open_clip's towers execute for the reference
// (SURVEY.md section 2.1 K1,K3,K5,K6,K7; reference call sites models/clip_wrapper.py:47,
// models/model_wrapper.py:58,72).
//
// Tile: 128 (m) x 128 (n) x 64 (k) per 256-thread workgroup = 4 waves as 2(m) x 2(n); each wave owns
// 64 x 64 = 4 x 4 tiles of v_mfma_f32_16x16x32_bf16.  Both operands are K-contiguous, so the MFMA is
// issued "swapped" (D = Wfrag . Afrag^T): a lane then holds 4 CONSECUTIVE n for one m, i.e. 8-byte
// (bf16) / 16-byte (fp32) vector stores and vector bias loads in the epilogue.
// LDS image: [128 rows][64 bf16] = 128-B rows, 16-B chunk index XOR (row & 7) so the ds_read_b128
// fragment reads of 16 different rows at one k-chunk spread over 8 slots (guide T2).
// bf16x3 (SPLIT): every 64-deep k-slice is visited three times in turn -- (A_hi,W_hi), (A_lo,W_hi), (A_hi,W_lo) -- into the
// same accumulators.
#include <cstdlib>

// (NOT built with TAPCLIP_TU_NO_PK_F32: measured in round 4, tools/ab_pk.sh -- the split-bf16 epilogues of this kernel carry
// the text tower of the default mode, and without packed-fp32 ops its text_features take 5.45 instead of 4.92 ms, the
// training step 18.9 instead of 18.6 ms (1.6 %); the file stays on the disassembly guard of tests/test_abi.py, like
// gemm256.hip and attention.hip)
#include "common.h"
#include "kernels.h"

namespace tapclip {

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile

// Per-thread staging slots: 4 x 16 B of the A tile and 4 x 16 B of the W tile (chunk c = tid + 256 i:
// row = c >> 3, 16-byte k-chunk = c & 7).  Kept as plain register arrays indexed by unrolled constants
// (a struct passed by reference was demoted to scratch memory by hipcc and serialised every load).
#define TAPCLIP_STAGE_LOAD(KT_IDX)                                                              \
  {                                                                                             \
    int seg_ = 0, kk_ = (KT_IDX);                                                               \
    if (SPLIT) { /* the three products of one k-slice in turn, as gemm256.hip / gemm_lat.hip */ \
      seg_ = kk_ % 3;                                                                           \
      kk_ /= 3;                                                                                 \
    }                                                                                           \
    const bf16_t* Ap_ = (SPLIT && seg_ == 1) ? g.A_lo : g.A_hi;                                 \
    const bf16_t* Wp_ = (SPLIT && seg_ == 2) ? g.W_lo : g.W_hi;                                 \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                          \
      ra[i_] = *reinterpret_cast<const uint4*>(Ap_ + a_off[i_] + kk_ * BK);                     \
      rw[i_] = *reinterpret_cast<const uint4*>(Wp_ + w_off[i_] + kk_ * BK);                     \
    }                                                                                           \
  }

#define TAPCLIP_STAGE_WRITE(BUF)                                                                \
  {                                                                                             \
    uint8_t* bA_ = smem + (BUF) * 2 * TILE_BYTES;                                               \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                          \
      *reinterpret_cast<uint4*>(bA_ + lds_off[i_]) = ra[i_];                                    \
      *reinterpret_cast<uint4*>(bA_ + TILE_BYTES + lds_off[i_]) = rw[i_];                       \
    }                                                                                           \
  }

template <int EPI, bool SPLIT>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];  // [2 buffers][A | W] = 64 KiB

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware, bijective block remap (guide T1): blocks that share an XCD (same blockIdx % 8) walk a
  // contiguous run of tiles, n fastest, so the A row-panel is re-used out of that XCD's L2.
  const int tiles_n = g.N / BN;
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, qd = nwg >> 3, rm = nwg & 7;
    bid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
  }
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int64_t m0 = (int64_t)tm * BM;
  const int n0 = tn * BN;

  const int KT1 = g.K / BK;
  const int KT = SPLIT ? 3 * KT1 : KT1;

  // accumulators start at the bias: same summation order as gemm256.hip, so a row's result does not
  // depend on which kernel (i.e. which batch size) computed it
  f32x4_t acc[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f32x4_t bv = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (EPI != EPI_PATCH_F32 && g.bias != nullptr) bv = *reinterpret_cast<const f32x4_t*>(g.bias + n0 + wn * 64 + j * 16 + 4 * q);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = bv;
  }

  // per-thread staging addresses (element offsets into A / W, byte offsets into an LDS tile)
  int64_t a_off[4], w_off[4];
  int lds_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i;
    const int row = c >> 3, kc = c & 7;
    int64_t m = m0 + row;
    if (m >= g.M) m = g.M - 1;  // clamp: rows past M are computed but never stored
    a_off[i] = m * g.lda + kc * 8;
    w_off[i] = (int64_t)(n0 + row) * g.K + kc * 8;
    lds_off[i] = row * 128 + ((kc ^ (row & 7)) << 4);
  }
  uint4 ra[4], rw[4];
  TAPCLIP_STAGE_LOAD(0)
  TAPCLIP_STAGE_WRITE(0)
  __syncthreads();

  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    const uint8_t* bufA = smem + cur * 2 * TILE_BYTES;
    const uint8_t* bufW = bufA + TILE_BYTES;
    // prefetch the next K tile into registers (the last iteration re-reads its own tile: harmless,
    // keeps the loop body branch-free around the loads)
    const int ktn = kt + 1 < KT ? kt + 1 : kt;
    TAPCLIP_STAGE_LOAD(ktn)

#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int kc = 4 * s + q;
      bf16x8_t af[4], wf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wm * 64 + i * 16 + r;
        af[i] = *reinterpret_cast<const bf16x8_t*>(bufA + row * 128 + ((kc ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = wn * 64 + j * 16 + r;
        wf[j] = *reinterpret_cast<const bf16x8_t*>(bufW + row * 128 + ((kc ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[j][i] = TAPCLIP_MFMA_16x16x32(wf[j], af[i], acc[j][i]);
    }

    TAPCLIP_STAGE_WRITE(cur ^ 1)  // the other buffer: its last readers passed the previous barrier
    __syncthreads();
  }

  // ---- epilogue: lane holds D[n = 4q + e][m = r] of each 16x16 tile (e = 0..3)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t m = m0 + wm * 64 + i * 16 + r;
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
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + 4 * q;
      f32x4_t v = acc[j][i];
      if (EPI == EPI_PATCH_F32) {
        const float4 pv = *reinterpret_cast<const float4*>(addrow + n);
        v[0] += pv.x; v[1] += pv.y; v[2] += pv.z; v[3] += pv.w;
      }
      if (EPI == EPI_BIAS_GELU_BF16) {
        if (g.act == 0) {
          // bf16 path: the fitted form (2.5e-5 abs, common.h), same as gemm256.hip so that a row's result
          // does not depend on the kernel / batch size; bf16x3 keeps the 5e-7 erf
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = SPLIT ? gelu_erf(v[e]) : gelu_fast16(v[e]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = SPLIT ? gelu_quick(v[e]) : gelu_quick_fast(v[e]);
        }
      }
      if (EPI == EPI_GELU_BWD_BF16) {
        const int64_t o = orow * g.ldo + n;
        const uint2 uh = *reinterpret_cast<const uint2*>(g.aux_hi + o);  // the 4 upstream gradients of this lane: one 8-byte load
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

template <int EPI, bool SPLIT>
hipError_t launch_t(const GemmArgs& a, hipStream_t s) {
  static bool attr_set = false;
  const int smem_bytes = 4 * TILE_BYTES;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<EPI, SPLIT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int64_t tiles_m = (a.M + BM - 1) / BM;
  const int64_t nwg = tiles_m * (a.N / BN);
  hipLaunchKernelGGL((gemm_kernel<EPI, SPLIT>), dim3((unsigned)nwg), dim3(256), smem_bytes, s, a);
  return hipGetLastError();
}

}  // namespace

hipError_t launch_gemm256(const GemmArgs& a, int epilogue, bool split, hipStream_t s);  // gemm256.hip
bool gemm256_supports(const GemmArgs& a);
hipError_t launch_gemm_lat(const GemmArgs& a, int epilogue, bool split, hipStream_t s);  // gemm_lat.hip
bool gemm_lat_supports(const GemmArgs& a);

hipError_t launch_gemm(const GemmArgs& a, int epilogue, bool split, hipStream_t s) {
  if (a.M <= 0 || a.N % BN != 0 || a.K % BK != 0 || a.K <= 0) return hipErrorInvalidValue;
  if ((a.lda % 8) != 0 || (a.ldo % 4) != 0) return hipErrorInvalidValue;
  // Large M (the image tower: M = 50 432; the 65-class text pass on every row: M = 6 045): the persistent 256-row kernel
  // of gemm256.hip.  The register-staged kernel in this file is latency-bound at K = 512 (57 us per text
  // GEMM against ~15 us) and keeps only the small problems: few classes, unit tests, tiny models.
  // TAPCLIP_GEMM_TILE=128|256 pins the choice (tests exercise both kernels on the same problem)
  static const int forced = [] {
    const char* e = getenv("TAPCLIP_GEMM_TILE");
    return e ? atoi(e) : 0;
  }();
  // Round 4: with the tied padding rows merged the 65-class text pass is M = 65 x 24..26 = 1 560..1 690 rows -- one partial
  // round of tiles for either tiled kernel, i.e. a latency problem: below 2 048 rows the one-tile-per-CU LDS-DMA kernel of
  // gemm_lat.hip runs (deep ring, tile shape per launch, the three split-bf16 products on ONE staging of the operands).
  // Measured at M = 1 560 before it existed (tools/train_phases.py, text_features alone): bf16 2.30 ms on the persistent
  // kernel against 4.53 ms on the register-staged one of this file; split-bf16 5.79 against 4.92 ms.
  // TAPCLIP_GEMM_LAT=0 takes it out (A/B: then 1 024 rows for one product, 2 048 for three decide between the other two);
  // TAPCLIP_GEMM256_MIN_M moves the persistent kernel's threshold (experiments).
  static const bool use_lat = [] {
    const char* e = getenv("TAPCLIP_GEMM_LAT");
    return e == nullptr || atoi(e) != 0;
  }();
  static const int64_t min_m_env = [] {
    const char* e = getenv("TAPCLIP_GEMM256_MIN_M");
    return e ? (int64_t)atoll(e) : (int64_t)0;
  }();
  const int64_t min_m = min_m_env > 0 ? min_m_env : ((split || use_lat) ? 2048 : 1024);
  if (gemm256_supports(a) && (forced == 256 || (forced != 128 && a.M >= min_m))) return launch_gemm256(a, epilogue, split, s);
  if (use_lat && forced != 128 && gemm_lat_supports(a)) return launch_gemm_lat(a, epilogue, split, s);
#define TAPCLIP_GEMM_CASE(E)                                      \
  case E:                                                         \
    return split ? launch_t<E, true>(a, s) : launch_t<E, false>(a, s);
  switch (epilogue) {
    TAPCLIP_GEMM_CASE(EPI_BIAS_BF16)
    TAPCLIP_GEMM_CASE(EPI_BIAS_GELU_BF16)
    TAPCLIP_GEMM_CASE(EPI_BIAS_RESID_F32)
    TAPCLIP_GEMM_CASE(EPI_PATCH_F32)
    TAPCLIP_GEMM_CASE(EPI_BIAS_F32)
    TAPCLIP_GEMM_CASE(EPI_GELU_BWD_BF16)
    default:
      return hipErrorInvalidValue;
  }
#undef TAPCLIP_GEMM_CASE
}

}  // namespace tapclip
