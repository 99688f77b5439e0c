// Large-M bf16 MFMA GEMM for gfx950:  C[M,N] = A[M,K] . W[N,K]^T, 256 x BN x 64 tiles (BN = 256 | 128),
// 512-thread workgroups (8 waves as 2(m) x 4(n)), operands staged global -> LDS by LDS-DMA
// (global_load_lds_dwordx4: no staging VGPRs, no ds_write), double-buffered.
//
// Why 256-wide tiles: a CU's MFMA peak is ~4070 FLOP/clk against ~56 B/clk from L2, so a tile needs
// >= 73 FLOP per staged byte; 128 x 128 x 64 gives 64, 256 x 128 gives 85, 256 x 256 gives 128.
// Tile order: XCD-aware (blocks with equal blockIdx % 8 share an L2 and walk a contiguous run of tiles)
// and grouped (8 m-tiles x all n-tiles per group, m fastest) so the ~64 tiles an XCD has in flight
// share 8 A panels and 8 W panels out of its 4 MiB L2.
//
// LDS image of a tile: [rows][64 bf16] = 128-B rows; 16-B chunk p of row r holds source k-chunk
// p ^ (r & 7).  LDS-DMA writes lane-linearly (lane l -> row l >> 3, chunk l & 7 of an 8-row piece), so
// the swizzle is applied to each lane's SOURCE address and again on the ds_read_b128 side (guide
// rule 21).  The MFMA is issued swapped (D = Wfrag . Afrag^T): a lane holds 4 consecutive n of one m.
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace tapclip {
namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

constexpr int BM = 256, BK = 64;
constexpr int GROUP_M = 8;

template <int EPI, bool SPLIT, int BN>
__global__ __launch_bounds__(512) void gemm256_kernel(GemmArgs g) {
  constexpr int A_BYTES = BM * BK * 2;
  constexpr int W_BYTES = BN * BK * 2;
  constexpr int STAGE = A_BYTES + W_BYTES;
  constexpr int NJ = BN / 64;  // 16-wide n sub-tiles per wave (wave covers BN / 4 columns)
  constexpr int WI = BN / 64;  // W pieces (8 rows x 128 B) per wave per stage
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];  // [2 stages][A | W]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int wm = wave >> 2, wn = wave & 3;

  // ---- block -> tile: XCD-contiguous, then grouped ordering
  const int tiles_m = (int)((g.M + BM - 1) / BM);
  const int tiles_n = g.N / BN;
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, qd = nwg >> 3, rm = nwg & 7;
    bid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
  }
  const int per_group = GROUP_M * tiles_n;
  const int grp = bid / per_group;
  const int first_m = grp * GROUP_M;
  const int gsize = tiles_m - first_m < GROUP_M ? tiles_m - first_m : GROUP_M;
  const int in_grp = bid - grp * per_group;
  const int tm = first_m + in_grp % gsize;
  const int tn = in_grp / gsize;
  const int64_t m0 = (int64_t)tm * BM;
  const int n0 = tn * BN;

  const int KT1 = g.K / BK;
  const int KT = SPLIT ? 3 * KT1 : KT1;

  // ---- LDS-DMA source offsets (elements).  Piece j of a tile = rows 8j .. 8j+7; wave w issues pieces
  // w, w+8, ...; lane l covers row 8j + (l >> 3), LDS chunk (l & 7) <- source chunk (l & 7) ^ (l >> 3).
  const int src_chunk = (lane & 7) ^ (lane >> 3);
  int64_t a_off[4], w_off[WI];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int64_t m = m0 + 8 * (wave + 8 * i) + (lane >> 3);
    if (m >= g.M) m = g.M - 1;  // clamp: rows past M are computed but never stored
    a_off[i] = m * g.lda + src_chunk * 8;
  }
#pragma unroll
  for (int i = 0; i < WI; ++i) w_off[i] = (int64_t)(n0 + 8 * (wave + 8 * i) + (lane >> 3)) * g.K + src_chunk * 8;

  auto stage = [&](int buf, int kt) {
    int seg = 0, kk = kt;
    if (SPLIT) {
      seg = kt / KT1;
      kk = kt - seg * KT1;
    }
    const bf16_t* Ap = (SPLIT && seg == 1) ? g.A_lo : g.A_hi;
    const bf16_t* Wp = (SPLIT && seg == 2) ? g.W_lo : g.W_hi;
    uint8_t* base = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(Ap + a_off[i] + kk * BK),
                                       (lds_void_t*)(base + (wave + 8 * i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < WI; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(Wp + w_off[i] + kk * BK),
                                       (lds_void_t*)(base + A_BYTES + (wave + 8 * i) * 1024), 16, 0, 0);
  };

  f32x4_t acc[NJ][8];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  stage(0, 0);
  __syncthreads();  // emits s_waitcnt vmcnt(0) for the outstanding LDS-DMA, then the barrier

  // fragment read offsets: row & 7 == r & 7 for every sub-tile (their bases are multiples of 16)
  const int swz = r & 7;
  const int a_row = (wm * 128 + r) * 128;
  const int w_row = A_BYTES + (wn * (BN / 4) + r) * 128;

  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < KT) stage(cur ^ 1, kt + 1);  // the other stage: its readers passed the previous barrier
    const uint8_t* base = smem + cur * STAGE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int coff = ((4 * s + q) ^ swz) << 4;
      bf16x8_t wf[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) wf[j] = *reinterpret_cast<const bf16x8_t*>(base + w_row + j * 2048 + coff);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(base + a_row + i * 2048 + coff);
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af, acc[j][i], 0, 0, 0);
      }
    }
    __syncthreads();  // vmcnt(0) (next stage landed) + barrier (everyone done reading this stage)
  }

  // ---- epilogue: lane holds D[n = 4q + e][m = r] of each 16 x 16 tile (e = 0..3)
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int64_t m = m0 + wm * 128 + i * 16 + r;
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
      const int n = n0 + wn * (BN / 4) + j * 16 + 4 * q;
      f32x4_t v = acc[j][i];
      if (EPI != EPI_PATCH_F32 && g.bias != nullptr) {
        const float4 bv = *reinterpret_cast<const float4*>(g.bias + n);
        v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
      }
      if (EPI == EPI_PATCH_F32) {
        const float4 pv = *reinterpret_cast<const float4*>(addrow + n);
        v[0] += pv.x; v[1] += pv.y; v[2] += pv.z; v[3] += pv.w;
      }
      if (EPI == EPI_BIAS_GELU_BF16) {
        if (g.act == 0) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_quick(v[e]);
        }
      }
      if (EPI == EPI_BIAS_BF16 || EPI == EPI_BIAS_GELU_BF16) {
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

template <int EPI, bool SPLIT, int BN>
hipError_t launch_t(const GemmArgs& a, hipStream_t s) {
  static bool attr_set = false;
  const int smem_bytes = 2 * (BM * BK * 2 + BN * BK * 2);
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<EPI, SPLIT, BN>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int64_t nwg = ((a.M + BM - 1) / BM) * (a.N / BN);
  hipLaunchKernelGGL((gemm256_kernel<EPI, SPLIT, BN>), dim3((unsigned)nwg), dim3(512), smem_bytes, s, a);
  return hipGetLastError();
}

template <int EPI>
hipError_t launch_e(const GemmArgs& a, bool split, hipStream_t s) {
  // 256-wide n tiles when that still gives >= ~6 rounds of 256 workgroups, else 128-wide
  const int64_t tiles_m = (a.M + BM - 1) / BM;
  static const int forced_bn = [] {
    const char* e = getenv("TAPCLIP_GEMM_BN");  // tests: pin the n-tile width
    return e ? atoi(e) : 0;
  }();
  const bool wide = (a.N % 256 == 0) && (forced_bn == 256 || (forced_bn != 128 && tiles_m * (a.N / 256) >= 6 * 256));
  if (wide) return split ? launch_t<EPI, true, 256>(a, s) : launch_t<EPI, false, 256>(a, s);
  return split ? launch_t<EPI, true, 128>(a, s) : launch_t<EPI, false, 128>(a, s);
}

}  // namespace

hipError_t launch_gemm256(const GemmArgs& a, int epilogue, bool split, hipStream_t s) {
  switch (epilogue) {
    case EPI_BIAS_BF16: return launch_e<EPI_BIAS_BF16>(a, split, s);
    case EPI_BIAS_GELU_BF16: return launch_e<EPI_BIAS_GELU_BF16>(a, split, s);
    case EPI_BIAS_RESID_F32: return launch_e<EPI_BIAS_RESID_F32>(a, split, s);
    case EPI_PATCH_F32: return launch_e<EPI_PATCH_F32>(a, split, s);
    case EPI_BIAS_F32: return launch_e<EPI_BIAS_F32>(a, split, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace tapclip
