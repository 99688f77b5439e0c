// MX-fp8 (OCP e4m3 elements, one e8m0 scale per 32 consecutive k) MFMA GEMM for gfx950:
//   C[M,N] = (A . 2^sa)[M,K] . (W . 2^sw)[N,K]^T + bias, fused epilogues,
// on v_mfma_scale_f32_32x32x64_f8f6f4 (2x the bf16 MFMA rate, half the staged bytes per FLOP).
//
// Serves the "fp8" precision of the image tower (BASELINE.json configs[4]):
// the nn.Linear calls of open_clip's ResidualAttentionBlock (reference call site
// models/clip_wrapper.py:47) with activations and weights quantised to MXFP8.
//
// Same skeleton as gemm256.hip (read that file's header first): persistent 256 x 256 tiles, 8 waves as
// 2(m) x 4(n), a ring of 4 LDS-DMA stages, staggered half-groups, hand-counted vmcnt.  What differs:
//  * a stage is 64 k deep: [256 rows][64 B] of A, the same of W, and 1 KiB of scales (2 bytes per row:
//    the two 32-blocks of the step).  Scales live in global memory k-step major, [K/64][rows][2], so a
//    tile's 512 B of a step are contiguous and ride the ring as one more DMA piece per wave (8 lanes).
//  * operand lane map of the 32x32x64 form (measured, tools/probes/mx_fp8_probe.hip): lane (r = l & 31,
//    h = l >> 5) supplies row r; its register bytes 0..15 belong to the step's first 32-block and bytes
//    16..31 to the second, and the scale VGPR of lane (r, h) is the scale of row r's block h.  Within a
//    block the k order only has to agree between the two operands: both read 16 B at byte 16 h and 16 B at
//    byte 32 + 16 h of the row.
//  * LDS image: 64-B rows, 16-B chunk c of row R at slot c ^ ((R >> 2) & 3): the four 16-lane groups of a
//    ds_read_b128 (rows {0-3,12-15,20-27} / {4-11,16-19,28-31} of a 32-row fragment, one chunk) then
//    touch 16 distinct slots of the 256-B bank window -- for the natural row order of the A fragments and
//    for both row permutations of the W fragments below.
//  * swapped product (D = Wfrag . Afrag^T) and a free choice of which W row feeds which MFMA row: lane
//    (m = l & 31, h) ends up with 16 CONSECUTIVE n of each 32 x 32 tile -- columns 32 jt + 16 h + reg of the
//    wave's 64 for the bf16 / fp32 outputs, columns 32 h + 16 jt + reg for the MXFP8 output, where a lane
//    then owns one whole 32-block of its row and the block maximum needs no cross-lane step.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace tapclip {
namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef int i32x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

// LDS accesses of the epilogue transpose go through inline asm (see gemm256.hip: a compiler-visible LDS
// access that may alias an in-flight LDS-DMA gets s_waitcnt vmcnt(0) in front of it)
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint8_t*)p;
}
__device__ __forceinline__ void lds_write_b128(uint32_t addr, u32x4_t v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_read_b128_nowait(uint32_t a0, u32x4_t& v0) {
  asm volatile("ds_read_b128 %0, %1" : "=v"(v0) : "v"(a0) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_wait_x1(u32x4_t& v0) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v0) : "n"(N) : "memory");
}

// Counted vmcnt wait chosen at run time INSIDE one asm statement (mode: 0..3 -> vmcnt(C0..C3), anything else ->
// no wait).  A C++-level branch around the waits splits the k-step into basic blocks, and LLVM then sinks the
// (side-effect-free) MFMAs of the COMPUTE phase below the wait: the wait no longer runs under them.
template <int C0, int C1, int C2, int C3>
__device__ __forceinline__ void wait_vmcnt_mode(int mode) {
  asm volatile(
      "s_cmp_lg_u32 %0, 0\n\t"
      "s_cbranch_scc1 .Lwm1_%=\n\t"
      "s_waitcnt vmcnt(%1)\n\t"
      "s_branch .Lwme_%=\n"
      ".Lwm1_%=:\n\t"
      "s_cmp_lg_u32 %0, 1\n\t"
      "s_cbranch_scc1 .Lwm2_%=\n\t"
      "s_waitcnt vmcnt(%2)\n\t"
      "s_branch .Lwme_%=\n"
      ".Lwm2_%=:\n\t"
      "s_cmp_lg_u32 %0, 2\n\t"
      "s_cbranch_scc1 .Lwm3_%=\n\t"
      "s_waitcnt vmcnt(%3)\n\t"
      "s_branch .Lwme_%=\n"
      ".Lwm3_%=:\n\t"
      "s_cmp_lg_u32 %0, 3\n\t"
      "s_cbranch_scc1 .Lwme_%=\n\t"
      "s_waitcnt vmcnt(%4)\n"
      ".Lwme_%=:"
      :
      : "s"(mode), "n"(C0), "n"(C1), "n"(C2), "n"(C3)
      : "memory", "scc");
}
// LDS-DMA of 16 B per lane by lanes 0..7 only, without a compiler-visible branch (EXEC is all ones around it)
__device__ __forceinline__ void dma_lanes_0_7(uint32_t voff, const uint8_t* sbase, uint32_t lds_base) {
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_mov_b64 exec, 0xff\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %1\n\t"
      "s_mov_b64 exec, -1"
      :
      : "v"(voff), "s"(sbase), "s"(lds_base)
      : "memory", "m0");
}

constexpr int BM = 256, BN = 256, BK = 64, NS = 4;
constexpr int A_BYTES = BM * BK, W_BYTES = BN * BK, SC_BYTES = 1024;
constexpr int STAGE = A_BYTES + W_BYTES + SC_BYTES;  // 33 KiB
constexpr int MAX_N_BIAS = 4096;

template <int EPI>
__global__ __launch_bounds__(512) void gemm_mx8_kernel(Mx8GemmArgs g) {
  constexpr int NDMA = 5;                 // 2 A pieces + 2 W pieces + 1 scale piece per wave per stage
  // stores per wave of a clean epilogue: bf16 16 (8 tiles x 2 passes), MXFP8 8 data + 4 scale
  constexpr int NST = EPI == EPI_BIAS_GELU_MX8 ? 12 : 16;
  constexpr bool CLEAN_EPI = EPI == EPI_BIAS_BF16 || EPI == EPI_BIAS_GELU_MX8;
  constexpr int WAIT_STEADY = (NS - 2) * NDMA;
  constexpr int WAIT_RELAXED = WAIT_STEADY + NST;
  static_assert(WAIT_RELAXED <= 63, "vmcnt is a 6-bit field");
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];  // [NS stages][A | W | scales], 8 x 1 KiB epilogue scratch, bias[N]
  float* bias_lds = reinterpret_cast<float*>(smem + NS * STAGE + 8 * 1024);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave >> 2, wn = wave & 3;
  uint8_t* scratch = smem + NS * STAGE + wave * 1024;  // wave-private epilogue transpose buffer

  const int tiles_m = (int)((g.M + BM - 1) / BM);
  const int tiles_n = g.N / BN;
  const int total = tiles_m * tiles_n;
  const int xcd = blockIdx.x & 7, bpx = gridDim.x >> 3;
  const int cq = total >> 3, crm = total & 7;
  const int chunk_lo = xcd * cq + (xcd < crm ? xcd : crm);
  const int chunk_hi = chunk_lo + cq + (xcd < crm ? 1 : 0);
  int lid = chunk_lo + (blockIdx.x >> 3);
  if (lid >= chunk_hi) return;

  for (int i = tid; i < g.N; i += 512) bias_lds[i] = g.bias ? g.bias[i] : 0.f;

  const int KS = g.K / BK;

  auto tile_origin = [&](int id, int64_t& m0, int& n0) {
    const int GM = g.group_m;
    const int per_group = GM * tiles_n;
    const int grp = id / per_group;
    const int first_m = grp * GM;
    const int gsize = tiles_m - first_m < GM ? tiles_m - first_m : GM;
    const int in_grp = id - grp * per_group;
    m0 = (int64_t)(first_m + in_grp % gsize) * BM;
    n0 = (in_grp / gsize) * BN;
  };
  // LDS-DMA sources.  Operand piece j of a stage = rows 16 j .. 16 j + 15 (1 KiB); wave w issues pieces w
  // and w + 8; lane l covers row 16 j + (l >> 2), LDS chunk l & 3 <- source chunk (l & 3) ^ ((row >> 2) & 3).
  // Scale piece: the stage's 1 KiB = 64 chunks of 16 B (8 rows x 2 B): 0..31 A rows, 32..63 W rows; wave w
  // moves chunks 8 w .. 8 w + 7 with its lanes 0..7.
  auto tile_offsets = [&](int64_t m0, int n0, uint32_t (&a_off)[2], uint32_t (&w_off)[2], uint32_t& s_off, const uint8_t*& s_base,
                          int64_t& s_step) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = 16 * (wave + 8 * i) + (lane >> 2);
      const int sc = (lane & 3) ^ ((row >> 2) & 3);
      int64_t m = m0 + row;
      if (m >= g.M) m = g.M - 1;  // rows past M are computed but never stored
      a_off[i] = (uint32_t)(m * g.lda + sc * 16);
      w_off[i] = (uint32_t)((int64_t)(n0 + row) * g.K + sc * 16);
    }
    const int c = 8 * (wave & 3) + (lane & 7);  // 16-B chunk = 8 rows
    if (wave < 4) {
      int64_t m = m0 + 8 * c;
      if (m + 8 > g.m_pad) m = g.m_pad - 8;
      s_base = g.A_scale;
      s_off = (uint32_t)(m * 2);
      s_step = g.m_pad * 2;
    } else {
      s_base = g.W_scale;
      s_off = (uint32_t)((n0 + 8 * c) * 2);
      s_step = (int64_t)(g.w_scale_rows > 0 ? g.w_scale_rows : g.N) * 2;
    }
  };
  auto stage_dma = [&](int st, int ks, const uint32_t (&a_off)[2], const uint32_t (&w_off)[2], uint32_t s_off, const uint8_t* s_base,
                       int64_t s_step) {
    const uint8_t* Ap = g.A + ks * BK;
    const uint8_t* Wp = g.W + ks * BK;
    uint8_t* base = smem + st * STAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(Ap + a_off[i]), (lds_void_t*)(base + (wave + 8 * i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(Wp + w_off[i]), (lds_void_t*)(base + A_BYTES + (wave + 8 * i) * 1024), 16, 0, 0);
    dma_lanes_0_7(s_off, s_base + ks * s_step, lds_addr(base + A_BYTES + W_BYTES + wave * 128));
  };

  // fragment read offsets.  A fragment i: rows wm*128 + 32 i + r.  W fragment jt: MFMA row index r is fed by
  // tile row nrow(r) (+ JT_ROWS * jt), so that D row (reg & 3) + 8 (reg >> 2) + 4 h is column 16 h + reg of
  // tile jt (bf16 / fp32 outputs: JT_ROWS = 32) or column 32 h + 16 jt + reg (MXFP8 output: JT_ROWS = 16).
  constexpr bool QOUT = EPI == EPI_BIAS_GELU_MX8;
  constexpr int JT_ROWS = QOUT ? 16 : 32;
  const int arow = wm * 128 + r;
  const int nrow = (QOUT ? 32 : 16) * ((r >> 2) & 1) + 4 * (r >> 3) + (r & 3);
  const int wrow = wn * 64 + nrow;
  // bits 2, 3 of the row: 32 i, JT_ROWS jt and the wave bases are multiples of 16 and do not touch them
  const int a_c0 = (h ^ ((arow >> 2) & 3)) << 4, a_c1 = ((2 + h) ^ ((arow >> 2) & 3)) << 4;
  const int w_c0 = (h ^ ((wrow >> 2) & 3)) << 4, w_c1 = ((2 + h) ^ ((wrow >> 2) & 3)) << 4;
  const int a_base = arow * 64;
  const int w_base = A_BYTES + wrow * 64;
  const int as_base = A_BYTES + W_BYTES + arow * 2 + h;
  const int ws_base = A_BYTES + W_BYTES + 512 + wrow * 2 + h;

  int f_lid = lid, f_ks = 0;
  uint32_t a_off[2], w_off[2], s_off;
  const uint8_t* s_base;
  int64_t s_step;
  {
    int64_t fm0;
    int fn0;
    tile_origin(f_lid, fm0, fn0);
    tile_offsets(fm0, fn0, a_off, w_off, s_off, s_base, s_step);
  }
  // The cursor never stops issuing: past its last tile it re-reads that tile (into stages nobody consumes), so
  // every wait sees the same number of DMA events in flight and the k-step needs no "ran out" variant.
  auto issue_dma = [&](int st) { stage_dma(st, f_ks, a_off, w_off, s_off, s_base, s_step); };
  auto advance = [&]() {
    if (++f_ks == KS) {
      f_ks = 0;
      if (f_lid + bpx < chunk_hi) {
        f_lid += bpx;
        int64_t fm0;
        int fn0;
        tile_origin(f_lid, fm0, fn0);
        tile_offsets(fm0, fn0, a_off, w_off, s_off, s_base, s_step);
      }
    }
  };
  auto fetch_next = [&](int st) {
    issue_dma(st);
    advance();
  };

#pragma unroll
  for (int i = 0; i < NS - 1; ++i) fetch_next(i);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT_STEADY) : "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  const bool grp_b = wave >= 4;
  if (grp_b) {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

#ifdef MX8_STAMP
  // stamps are buffered in spare LDS (a global store per stamp would add vmcnt events to the counted waits)
  int stamp_n = 0;
  const bool stamper = blockIdx.x == 40 && lane == 0 && (wave == 0 || wave == 4);
  uint32_t* stamp_lds = reinterpret_cast<uint32_t*>(smem + NS * STAGE + 8 * 1024 + MAX_N_BIAS * 4);
#define STAMP(slot)                                                                                  \
  do {                                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                               \
    unsigned long long t_;                                                                           \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                       \
    if (stamper && stamp_n < 48) stamp_lds[((wave >> 2) * 48 + stamp_n) * 8 + (slot)] = (uint32_t)t_; \
    __builtin_amdgcn_sched_barrier(0);                                                               \
  } while (0)
#else
#define STAMP(slot)
#endif
  int st = 0;
  int relaxed = 0;
  int64_t m0 = 0;
  int n0 = 0;
  f32x16_t acc[2][4];

  // The DMA of step h + NS - 1 is issued in the COMPUTE phase of step h, between the wave's own 64-cycle
  // MFMAs (issue slots are free there; in the READ phase the pieces queue behind the ds_reads).  Group A
  // waits at the end of that COMPUTE phase, with steps h + 2 and h + 3 allowed in flight; group B waits in
  // its READ phase, BEFORE it issues step h + 3, so only step h + 2 may be in flight there.  For the first
  // NS - 2 waits after a clean epilogue its NST stores may stay in flight as well ("relaxed").
  auto wait_dma = [&](bool here) {  // here: this group waits at this site (wave-uniform)
    const int mode = here ? (grp_b ? 2 : 0) + (relaxed > 0 ? 1 : 0) : 4;
    wait_vmcnt_mode<WAIT_STEADY, WAIT_RELAXED, WAIT_STEADY - NDMA, WAIT_RELAXED - NDMA>(__builtin_amdgcn_readfirstlane(mode));
  };
  bool adv_pending = false;

  auto epilogue = [&](int64_t m0, int n0) {
    // lane (m = r, h) holds columns 16 h + 0..15 of each 32 x 32 tile.  Each tile leaves in two passes of 16
    // rows through a wave-private [16 rows][64 B] LDS block (16-B chunk c of row rr at slot c ^ ((rr >> 1) & 3):
    // conflict-free both ways): the half of the lanes that own those rows write 32 B each, then every lane
    // reads 16 B back in row order and stores it, so a store instruction covers 16 rows x 64 B.  A wave's
    // LDS instructions execute in order: no barrier; the read of pass k is waited for (counted) after the
    // writes and the read of pass k + 1 have been issued.
    if (EPI == EPI_BIAS_F32) {  // unit API: plain stores, lane holds columns 32 jt + 16 h + 0..15 of row m
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t m = m0 + wm * 128 + 32 * i + r;
        if (m >= g.M) continue;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
          float* dst = g.out_f32 + m * g.ldo + n0 + wn * 64 + 32 * jt + 16 * h;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            *reinterpret_cast<f32x4_t*>(dst + 4 * e) = f32x4_t{acc[jt][i][4 * e], acc[jt][i][4 * e + 1], acc[jt][i][4 * e + 2], acc[jt][i][4 * e + 3]};
        }
      }
      return;
    }
    const uint32_t sbase = lds_addr(scratch);
    const int row16 = r & 15;
    if (EPI == EPI_BIAS_GELU_MX8) {
      // lane (m = r, h) holds the 32-block h of its row: activation, block maximum, e8m0 scale and e4m3
      // elements all in-lane.  32 rows x 64 B leave in two passes of 16 rows through the wave's LDS block
      // (as below); the scale bytes of a 32-row block are 64 contiguous bytes (k-step major scale layout).
      auto body = [&](auto act_tag) {
        constexpr int ACT = decltype(act_tag)::value;
        const uint32_t wq0 = sbase + row16 * 64 + (((2 * h) ^ ((row16 >> 1) & 3)) << 4);
        const uint32_t wq1 = sbase + row16 * 64 + (((2 * h + 1) ^ ((row16 >> 1) & 3)) << 4);
        const int rr = lane >> 2, rc = lane & 3;
        const uint32_t rd = sbase + rr * 64 + ((rc ^ ((rr >> 1) & 3)) << 4);
        const bool full = m0 + BM <= g.M;
        const int64_t mrow = m0 + wm * 128 + rr;
        uint8_t* optr = g.out_q + mrow * g.ldo + n0 + wn * 64 + 16 * rc;
        const int64_t step16 = 16 * g.ldo;
        uint8_t* sptr = g.out_q_scale + ((size_t)((n0 >> 6) + wn) * g.out_m_pad + (m0 + wm * 128 + r)) * 2 + h;
        u32x4_t val[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          // (pairs: the activation's plain ops and the scale multiply issue as packed fp32 ops -- same roundings, same bits --
          // 12.5 -> 9 VALU issue slots per value; the epilogue is VALU-bound)
          f32x2_pk_t y[16];
          float amax = 0.f;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const f32x2_pk_t x = {acc[e >> 3][i][(2 * e) & 15], acc[e >> 3][i][(2 * e + 1) & 15]};
            y[e] = ACT == 0 ? gelu_erf_fast2(x) : ACT == 1 ? gelu_quick_fast2(x) : x;
            amax = ACT == 2 ? amax3_visible(amax, y[e][0], y[e][1]) : amax3_raw(amax, y[e][0], y[e][1]);  // (ACT 2: y is the MFMA's own register)
          }
          const uint32_t byte = mx8_scale_byte(amax);
          const float inv = mx8_inv_scale(byte);
          u32x4_t p0, p1;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            p0[e] = mx8_pack4(y[2 * e], y[2 * e + 1], inv);
            p1[e] = mx8_pack4(y[8 + 2 * e], y[8 + 2 * e + 1], inv);
          }
          if (full || m0 + wm * 128 + 32 * i + r < g.M) sptr[64 * i] = (uint8_t)byte;
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const int k = 2 * i + p;
            if ((r >> 4) == p) {
              lds_write_b128(wq0, p0);
              lds_write_b128(wq1, p1);
            }
            lds_read_b128_nowait(rd, val[k & 1]);
            if (k > 0) {
              lds_wait_x1<3>(val[(k - 1) & 1]);
              if (full || mrow + 16 * (k - 1) < g.M) *reinterpret_cast<u32x4_t*>(optr + (k - 1) * step16) = val[(k - 1) & 1];
            }
          }
        }
        lds_wait_x1<0>(val[1]);
        if (full || mrow + 112 < g.M) *reinterpret_cast<u32x4_t*>(optr + 7 * step16) = val[1];
      };
      if (g.act == 0) body(std::integral_constant<int, 0>{});
      else if (g.act == 1) body(std::integral_constant<int, 1>{});
      else body(std::integral_constant<int, 2>{});
      return;
    }
    const uint32_t wr0 = sbase + row16 * 64 + (((2 * h) ^ ((row16 >> 1) & 3)) << 4);
    const uint32_t wr1 = sbase + row16 * 64 + (((2 * h + 1) ^ ((row16 >> 1) & 3)) << 4);
    const int rr = lane >> 2, rc = lane & 3;
    const uint32_t rd = sbase + rr * 64 + ((rc ^ ((rr >> 1) & 3)) << 4);
    const bool full = m0 + BM <= g.M;
    const int64_t gM = g.M;
    const int64_t mrow = m0 + wm * 128 + rr;
    bf16_t* optr = g.out_bf16 + mrow * g.ldo + n0 + wn * 64 + 8 * rc;  // this lane's read-back position, block (0, 0, 0)
    const int64_t step16 = 16 * g.ldo;
    u32x4_t val[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
        const f32x16_t v = acc[jt][i];
        u32x4_t p0, p1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          p0[e] = pack_bf2(v[2 * e], v[2 * e + 1]);
          p1[e] = pack_bf2(v[8 + 2 * e], v[8 + 2 * e + 1]);
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const int k = i * 4 + jt * 2 + p;
          if ((r >> 4) == p) {
            lds_write_b128(wr0, p0);
            lds_write_b128(wr1, p1);
          }
          lds_read_b128_nowait(rd, val[k & 1]);
          if (k > 0) {
            const int kp = k - 1, ip = kp >> 2, jp = (kp >> 1) & 1, pp = kp & 1;
            lds_wait_x1<3>(val[kp & 1]);
            if (full || mrow + 32 * ip + 16 * pp < gM) *reinterpret_cast<u32x4_t*>(optr + (2 * ip + pp) * step16 + 32 * jp) = val[kp & 1];
          }
        }
      }
    }
    lds_wait_x1<0>(val[1]);
    if (full || mrow + 112 < gM) *reinterpret_cast<u32x4_t*>(optr + 7 * step16 + 32) = val[1];
  };

  for (;;) {
    tile_origin(lid, m0, n0);
    const int next_lid = lid + bpx;
    const bool has_next = next_lid < chunk_hi;
    // accumulators start at the bias
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
      const float* bp = bias_lds + n0 + wn * 64 + (QOUT ? 32 * h + 16 * jt : 32 * jt + 16 * h);
      f32x16_t bv;
#pragma unroll
      for (int e = 0; e < 16; ++e) bv[e] = bp[e];
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[jt][i] = bv;
    }
    for (int ks = 0; ks < KS; ++ks) {
      // ================= READ phase
      STAMP(0);
      if (adv_pending) advance();  // (branchy cursor arithmetic: kept away from the MFMAs)
      const uint8_t* base = smem + st * STAGE;
      u32x4_t wf[2][2], af[4][2];
      int wsc[2], asc[4];
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
        wf[jt][0] = *reinterpret_cast<const u32x4_t*>(base + w_base + jt * (JT_ROWS * 64) + w_c0);
        wf[jt][1] = *reinterpret_cast<const u32x4_t*>(base + w_base + jt * (JT_ROWS * 64) + w_c1);
        wsc[jt] = base[ws_base + jt * (JT_ROWS * 2)];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[i][0] = *reinterpret_cast<const u32x4_t*>(base + a_base + i * 2048 + a_c0);
        af[i][1] = *reinterpret_cast<const u32x4_t*>(base + a_base + i * 2048 + a_c1);
        asc[i] = base[as_base + i * 64];
      }
      STAMP(1);
      wait_dma(grp_b);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      STAMP(2);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      STAMP(3);
      // ================= COMPUTE phase
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const i32x8_t a8 = {(int)af[i][0][0], (int)af[i][0][1], (int)af[i][0][2], (int)af[i][0][3],
                            (int)af[i][1][0], (int)af[i][1][1], (int)af[i][1][2], (int)af[i][1][3]};
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
          const i32x8_t w8 = {(int)wf[jt][0][0], (int)wf[jt][0][1], (int)wf[jt][0][2], (int)wf[jt][0][3],
                              (int)wf[jt][1][0], (int)wf[jt][1][1], (int)wf[jt][1][2], (int)wf[jt][1][3]};
          acc[jt][i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w8, a8, acc[jt][i], 0, 0, 0, wsc[jt], 0, asc[i]);
        }
        if (i == 0) {
          __builtin_amdgcn_sched_barrier(0);
          issue_dma((st + NS - 1) % NS);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      adv_pending = true;
      __builtin_amdgcn_sched_barrier(0);  // all 8 MFMAs are issued before the wait below, so they run under it
      STAMP(4);
      wait_dma(!grp_b);
      relaxed -= relaxed > 0 ? 1 : 0;
      STAMP(5);
      st = (st + 1) % NS;
      if (ks != KS - 1) {  // (the barrier behind a tile's last COMPUTE phase is placed around the epilogue, below)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      STAMP(6);
#ifdef MX8_STAMP
      ++stamp_n;
#endif
    }
    // ================= tile boundary (as in gemm256.hip): ONE epilogue call site; group A passes the phase barrier
    // first (its epilogue opens its READ phase of the next tile), group B runs the epilogue straight behind its last
    // COMPUTE phase, before that barrier -- the two waves of a SIMD do their epilogue VALU work (GELU, re-quantisation,
    // packing) side by side instead of one after the other with the partner parked at the barrier.  Group B's
    // vector-memory ops keep their order (DMA of step h + NS - 1 in its COMPUTE phase, then the NST stores), so the
    // wait counts are the ones of the previous placement.
    if (!grp_b) {
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    epilogue(m0, n0);
    relaxed = (CLEAN_EPI && m0 + BM <= g.M) ? NS - 2 : 0;
    if (grp_b) {
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    if (!has_next) break;
    lid = next_lid;
  }
  if (!grp_b) __builtin_amdgcn_s_barrier();  // group A is one phase ahead and owes the barrier group B started with
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the cursor's surplus DMAs must land before the LDS is released
#ifdef MX8_STAMP
  if (stamper)
    for (int i = 0; i < 48 * 8; ++i) g.stamps[(wave >> 2) * 48 * 8 + i] = stamp_lds[(wave >> 2) * 48 * 8 + i];
#endif
}

template <int EPI>
hipError_t launch_mx8_t(const Mx8GemmArgs& a, hipStream_t s) {
#ifdef MX8_STAMP
  const int smem_bytes = NS * STAGE + 8 * 1024 + MAX_N_BIAS * 4 + 4096;
#else
  const int smem_bytes = NS * STAGE + 8 * 1024 + a.N * 4;
#endif
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_mx8_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       NS * STAGE + 8 * 1024 + MAX_N_BIAS * 4 + 4096);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
    n_cu = prop.multiProcessorCount / 8 * 8;
    if (n_cu < 8) n_cu = 8;
  }
  const int64_t tiles = ((a.M + BM - 1) / BM) * (a.N / BN);
  const int64_t nwg = tiles < n_cu ? (tiles + 7) / 8 * 8 : n_cu;
  hipLaunchKernelGGL((gemm_mx8_kernel<EPI>), dim3((unsigned)nwg), dim3(512), smem_bytes, s, a);
  return hipGetLastError();
}

}  // namespace

bool gemm_mx8_supports(const Mx8GemmArgs& a) {
  return a.M > 0 && a.N % BN == 0 && a.K % BK == 0 && a.K >= BK * NS && a.N <= MAX_N_BIAS && a.lda % 16 == 0 && a.m_pad % 8 == 0 &&
         a.m_pad >= 8 && (uint64_t)a.M * (uint64_t)a.lda < (1ull << 32) && (uint64_t)a.N * (uint64_t)a.K < (1ull << 32) &&
         (uint64_t)a.m_pad * 2 < (1ull << 32);
}

hipError_t launch_gemm_mx8(const Mx8GemmArgs& a, int epilogue, hipStream_t s) {
  if (!gemm_mx8_supports(a)) return hipErrorInvalidValue;
  switch (epilogue) {
    case EPI_BIAS_BF16:
      if (a.out_bf16 == nullptr || a.ldo % 8 != 0) return hipErrorInvalidValue;
      return launch_mx8_t<EPI_BIAS_BF16>(a, s);
    case EPI_BIAS_F32:
      if (a.out_f32 == nullptr || a.ldo % 4 != 0) return hipErrorInvalidValue;
      return launch_mx8_t<EPI_BIAS_F32>(a, s);
    case EPI_BIAS_GELU_MX8:
      if (a.out_q == nullptr || a.out_q_scale == nullptr || a.ldo % 16 != 0 || a.out_m_pad < a.M) return hipErrorInvalidValue;
      return launch_mx8_t<EPI_BIAS_GELU_MX8>(a, s);
    default:
      return hipErrorInvalidValue;
  }
}

}  // namespace tapclip
