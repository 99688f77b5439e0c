// Large-M bf16 MFMA GEMM for gfx950:  C[M,N] = A[M,K] . W[N,K]^T (+ bias, fused epilogues), persistent,
// 256 x BN output tiles (BN = 256 | 128), 512-thread workgroups (8 waves as 2(m) x 4(n)), operands
// streamed global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging VGPRs, no ds_write) through a
// ring of NS stages of K depth 32.
//
// Replaces the nn.Linear / conv1 calls of open_clip's image tower (SURVEY.md section 2.1 K1,K3,K5,K6,K7;
// reference call site models/clip_wrapper.py:47) at M = batch x 197 rows.
//
// Design notes (all measured on MI355X, profiles/ r01):
//  * tile size: a CU's MFMA peak is ~4070 FLOP/clk; its vector-memory path delivered ~19 B/clk of LDS-DMA in this
//    kernel (32 one-KiB pieces per k-step at ~55 cycles each; all-L2-hit diagnostic: 15 % faster),
//    so FLOP per staged byte decides: 128 x 128 gives 64, 256 x 128 gives 85, 256 x 256 gives 128 (the largest
//    tile 8 waves x 256 registers hold).
//  * persistent + ring: one workgroup per CU walks its XCD's run of tiles; the K steps of successive
//    output tiles form ONE linear sequence through the ring, NS - 1 steps are always in flight, and the
//    waits are hand-counted s_waitcnt vmcnt(N) with raw s_barrier (vmcnt counts loads, stores and LDS-DMA
//    together, in issue order; __syncthreads() would add vmcnt(0)).  A 2-stage version of this kernel
//    with __syncthreads() spent ~40 % of a K = 768 launch outside the MFMA loop (prologue latency, store
//    drain) and stalled on every L2 miss.
//  * epilogue stores stay in flight across the next tile's first NS - 2 steps: their count (NST) is added
//    to the wait count there ("relaxed").  That is only valid when the epilogue issues exactly NST
//    vector-memory ops and no loads, so the bias lives in LDS (loaded once per workgroup) and seeds the
//    accumulators; epilogues with loads (residual, patch) and ragged / split tiles take vmcnt(0).
//  * tile order: XCD-aware (blocks with equal blockIdx % 8 share an L2 and own a contiguous chunk of the
//    tile order) and grouped (8 m-tiles x all n-tiles, m fastest).
//  * LDS image of a stage: [rows][32 bf16] = 64-B rows; 16-B chunk p of row R holds source k-chunk
//    p ^ (3 * ((R >> 3) & 1)): with that XOR the four 16-lane groups of a ds_read_b128 fragment read
//    (rows 0..15 x one k-chunk, middle rows one chunk over) touch 16 distinct 16-B slots.  LDS-DMA writes
//    lane-linearly (lane l -> row l >> 2, chunk l & 3 of a 16-row piece), so the swizzle is applied to each
//    lane's SOURCE address and again on the read side (guide rule 21).
//  * the MFMA is issued swapped (D = Wfrag . Afrag^T): a lane holds 4 consecutive n of one m (8-byte bf16 /
//    16-byte fp32 stores).
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace tapclip {
namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

// LDS accesses of the epilogue transpose go through inline asm: hipcc treats an in-flight LDS-DMA as a
// pending LDS write and puts s_waitcnt vmcnt(0) in front of any compiler-visible LDS access that may alias
// it, which would drain the prefetch ring at every epilogue.  The scratch region never aliases the ring.
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint8_t*)p;
}
__device__ __forceinline__ void lds_write_b64(uint32_t addr, unsigned long long v) {
  asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
// one or two 16-byte reads and the wait for them in ONE statement (guide section 5.7, form (i))
__device__ __forceinline__ void lds_read_b128_x2(uint32_t a0, uint32_t a1, u32x4_t& v0, u32x4_t& v1) {
  asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v0), "=&v"(v1) : "v"(a0), "v"(a1) : "memory");
}
// un-waited 16-byte read (guide section 5.7 form (ii)): the destination is only valid after one of the
// lds_wait_* statements below, which name it "+v" so that no consumer can be scheduled above the wait
__device__ __forceinline__ void lds_read_b128_nowait(uint32_t a0, u32x4_t& v0) {
  asm volatile("ds_read_b128 %0, %1" : "=v"(v0) : "v"(a0) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_wait_x2(u32x4_t& v0, u32x4_t& v1) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(v0), "+v"(v1) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_wait_x1(u32x4_t& v0) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v0) : "n"(N) : "memory");
}
__device__ __forceinline__ void lds_read_b128_x1(uint32_t a0, u32x4_t& v0) {
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v0) : "v"(a0) : "memory");
}

// epilogue store of 16 bytes per lane: non-temporal.  (Measured in the tower, same box, interleaved: plain 11.72 ms,
// nt 11.45 ms per step; sc1 / sc0 sc1 write-through 11.9-12.0 ms, nt sc1 11.59 ms.)
__device__ __forceinline__ void st16_policy(u32x4_t* p, const u32x4_t& v) { __builtin_nontemporal_store(v, p); }

constexpr int BM = 256, BKS = 32;
constexpr int MAX_N_BIAS = 4096;  // bias vector kept in LDS

template <int EPI, bool SPLIT, int BN, int NS>
__global__ __launch_bounds__(512) void gemm256_kernel(GemmArgs g) {
  constexpr int A_BYTES = BM * BKS * 2;   // 16 KiB
  constexpr int W_BYTES = BN * BKS * 2;
  constexpr int STAGE = A_BYTES + W_BYTES;
  constexpr int NJ = BN / 64;             // 16-wide n sub-tiles per wave (wave covers BN / 4 columns)
  constexpr int AP = 2;                   // A pieces (16 rows x 64 B = 1 KiB) per wave per stage
  constexpr int WP = BN / 128;            // W pieces per wave per stage
  constexpr int NDMA = AP + WP;
  // bf16 outputs of the non-split path leave through a per-wave 2-KiB LDS transpose, 16 rows at a time,
  // as 16-byte-per-lane stores of whole 128-B (BN = 256) / 64-B (BN = 128) row segments: the natural
  // fragment store (8 B per lane, 16 rows x 32 B per instruction) was store-ISSUE bound -- 232 MB of
  // QKV output cost ~100 us of a 236 us launch (gemm_bench NOSTORE experiment, profiles/ r01).
  // (the GELU epilogue keeps the direct 8-byte fragment stores: its VALU work between them hides the store
  // issue cost, and the extra LDS round trip only adds to an already VALU-bound epilogue: 284 vs 314 us)
#ifndef TAPCLIP_GELU_TR
#define TAPCLIP_GELU_TR 0
#endif
#ifndef TAPCLIP_GELU_FULL_LINES
#define TAPCLIP_GELU_FULL_LINES 0  // (A/B: tools/Makefile gemm_bench_alt ALT_DEFS=-DTAPCLIP_GELU_FULL_LINES=1)
#endif
#ifndef TAPCLIP_EPI_EARLY_B
#define TAPCLIP_EPI_EARLY_B 1  // group B's epilogue in the same phase as group A's (see the tile boundary below)
#endif
  constexpr bool TR_EPI = !SPLIT && (EPI == EPI_BIAS_BF16 || (TAPCLIP_GELU_TR && EPI == EPI_BIAS_GELU_BF16));
  constexpr int ROW_CHUNKS = BN / 32;     // 16-B chunks per 16-row scratch row (wave covers BN/4 columns)
  // The GELU epilogue stores straight from the accumulators, but with the W rows of each PAIR of 16-wide
  // sub-tiles interleaved in groups of 4 (sub-tile 2J takes columns 32J + 8q' + 0..3, sub-tile 2J+1 columns
  // 32J + 8q' + 4..7): a lane then holds 8 consecutive n of its row for the pair and stores 16 bytes, an
  // instruction covers 16 rows x 64 B instead of 16 rows x 32 B, and there are half as many of them.
  // Which W row feeds which MFMA output row is free: it is only the fragment's LDS read address.
  constexpr bool PERM = !SPLIT && EPI == EPI_BIAS_GELU_BF16 && !TR_EPI;
  constexpr int NST = TR_EPI ? 8 * ROW_CHUNKS / 4 : PERM ? NJ * 4 : NJ * 8;  // stores per wave of a clean epilogue
  constexpr int WAIT_STEADY = (NS - 2) * NDMA;
  constexpr int WAIT_RELAXED = WAIT_STEADY + NST;
  static_assert(WAIT_RELAXED <= 63, "vmcnt is a 6-bit field");
  constexpr bool CLEAN_EPI = !SPLIT && (EPI == EPI_BIAS_BF16 || EPI == EPI_BIAS_GELU_BF16 || EPI == EPI_BIAS_F32);
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];  // [NS stages][A | W] then bias[N]
  float* bias_lds = reinterpret_cast<float*>(smem + NS * STAGE + 8 * 2048);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int wm = wave >> 2, wn = wave & 3;
  uint8_t* scratch = smem + NS * STAGE + wave * 2048;  // wave-private epilogue transpose buffer

  // ---- this workgroup's run of tiles: XCD x (= blockIdx % 8, the blocks that share an L2) owns a
  // contiguous chunk of the grouped tile order; its gridDim/8 workgroups take ids j, j + bpx, ...
  const int tiles_m = (int)((g.M + BM - 1) / BM);
  const int tiles_n = g.N / BN;
  // (tiles >= g.split_from, if any, are not walked this way: they are K-split over the workgroups, below)
  const int total = g.split_parts > 0 ? g.split_from : tiles_m * tiles_n;
  const int xcd = blockIdx.x & 7, bpx = gridDim.x >> 3;
  const int cq = total >> 3, crm = total & 7;
  const int chunk_lo = xcd * cq + (xcd < crm ? xcd : crm);
  const int chunk_hi = chunk_lo + cq + (xcd < crm ? 1 : 0);
  // tail split: workgroup b < (tiles - split_from) * split_parts also computes part b % split_parts of
  // tile split_from + b / split_parts, after its whole tiles
  const int n_split_items = g.split_parts > 0 ? (tiles_m * tiles_n - g.split_from) * g.split_parts : 0;
  const bool has_split = (int)blockIdx.x < n_split_items;
  const int split_lid = has_split ? g.split_from + (int)blockIdx.x / g.split_parts : -1;
  const int split_part = has_split ? (int)blockIdx.x % g.split_parts : 0;
  int lid = chunk_lo + (blockIdx.x >> 3);
  bool cur_partial = false;
  if (lid >= chunk_hi) {
    if (!has_split) return;
    lid = split_lid;
    cur_partial = true;
  }

  // bias -> LDS (zeros when absent); visible after the first barrier below
  for (int i = tid; i < g.N; i += 512) bias_lds[i] = g.bias ? g.bias[i] : 0.f;

  const int KS1 = g.K / BKS;
  const int KS = SPLIT ? 3 * KS1 : KS1;
  const int KSP = g.split_parts > 0 ? KS / g.split_parts : KS;  // K steps of a split part

  // tile id -> (m0, n0): grouped ordering, group_m (8) m-tiles x all n-tiles per group, m fastest
  // (gemm_bench: 6..12 are within 2 % of each other, 2 and 32 lose 5-10 %)
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
  // LDS-DMA source offsets (32-bit BYTE offsets: the launcher checks that A and W are < 4 GiB).
  // Piece j of a stage = rows 16j .. 16j+15; wave w issues pieces w, w + 8; lane l covers row
  // 16j + (l >> 2), LDS chunk (l & 3) <- source chunk (l & 3) ^ (3 * ((l >> 5) & 1)).
  const int src_chunk = (lane & 3) ^ (3 * ((lane >> 5) & 1));
  auto tile_offsets = [&](int64_t m0, int n0, uint32_t (&a_off)[AP], uint32_t (&w_off)[WP]) {
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      int64_t m = m0 + 16 * (wave + 8 * i) + (lane >> 2);
      if (m >= g.M) m = g.M - 1;  // clamp: rows past M are computed but never stored
      a_off[i] = (uint32_t)((m * g.lda + src_chunk * 8) * 2);
    }
#pragma unroll
    for (int i = 0; i < WP; ++i)
      w_off[i] = (uint32_t)(((int64_t)(n0 + 16 * (wave + 8 * i) + (lane >> 2)) * g.K + src_chunk * 8) * 2);
  };
  auto stage_dma = [&](int st, int ks, const uint32_t (&a_off)[AP], const uint32_t (&w_off)[WP]) {
    // split-bf16: the sequence of 3 K / 32 steps takes the three products of one 32-deep slice in turn -- hi.hi, A_lo.W_hi,
    // A_hi.W_lo, then the next slice -- the order in which gemm_lat.hip, which stages a slice once, issues them (round 4; the
    // products used to run as three passes over K).  A row's bits must not depend on which kernel its launch selects.
    int seg = 0, kk = ks;
    if (SPLIT) {
      kk = ks / 3;
      seg = ks - kk * 3;
    }
    const uint8_t* Ap = reinterpret_cast<const uint8_t*>((SPLIT && seg == 1) ? g.A_lo : g.A_hi) + kk * (BKS * 2);
    const uint8_t* Wp = reinterpret_cast<const uint8_t*>((SPLIT && seg == 2) ? g.W_lo : g.W_hi) + kk * (BKS * 2);
    uint8_t* base = smem + st * STAGE;
#pragma unroll
    for (int i = 0; i < AP; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(Ap + a_off[i]), (lds_void_t*)(base + (wave + 8 * i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < WP; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(Wp + w_off[i]), (lds_void_t*)(base + A_BYTES + (wave + 8 * i) * 1024), 16, 0, 0);
  };

  // fragment read offsets within a stage: row R = sub-tile base (multiple of 16) + r
  const int frag_off = r * 64 + ((q ^ (3 * ((r >> 3) & 1))) << 4);
  const int a_base = (wm * 128) * 64 + frag_off;
  // PERM: lane r of sub-tile j reads W row 32 (j >> 1) + 8 (r >> 2) + 4 (j & 1) + (r & 3); bit 3 of that row
  // is bit 2 of r, and the four 16-lane groups of the read still touch 16 distinct 16-B slots
  const int w_frag_off = PERM ? (8 * (r >> 2) + (r & 3)) * 64 + ((q ^ (3 * ((r >> 2) & 1))) << 4) : frag_off;
  const int w_base = A_BYTES + (wn * (BN / 4)) * 64 + w_frag_off;
  auto w_sub_off = [](int j) { return PERM ? (j >> 1) * 2048 + (j & 1) * 256 : j * 1024; };  // bytes
  // column (relative to n0 + wn * BN/4) of element e = 0 of this lane's accumulator acc[j][.]
  auto col_of = [&](int j) { return PERM ? 32 * (j >> 1) + 8 * q + 4 * (j & 1) : j * 16 + 4 * q; };

  // ---- fetch cursor: the (tile, k-step) whose DMA is issued next; runs NS - 1 steps ahead of compute
  int f_lid = lid, f_ks = cur_partial ? split_part * KSP : 0;
  int f_ks_end = cur_partial ? f_ks + KSP : KS;
  bool f_valid = true, f_partial = cur_partial;
  uint32_t a_off[AP], w_off[WP];
  {
    int64_t fm0;
    int fn0;
    tile_origin(f_lid, fm0, fn0);
    tile_offsets(fm0, fn0, a_off, w_off);
  }
  auto fetch_next = [&](int st) {  // issue the cursor's DMA into stage st, then advance the cursor
    stage_dma(st, f_ks, a_off, w_off);
    if (++f_ks == f_ks_end) {
      f_ks = 0;
      f_ks_end = KS;
      f_lid += bpx;
      f_valid = !f_partial && f_lid < chunk_hi;
      if (!f_valid && !f_partial && has_split) {  // whole tiles done: the K-split part comes last
        f_valid = f_partial = true;
        f_lid = split_lid;
        f_ks = split_part * KSP;
        f_ks_end = f_ks + KSP;
      }
      if (f_valid) {
        int64_t fm0;
        int fn0;
        tile_origin(f_lid, fm0, fn0);
        tile_offsets(fm0, fn0, a_off, w_off);
      }
    }
  };

  // prologue: steps 0 .. NS-2 in flight, then step 0 landed
#pragma unroll
  for (int i = 0; i < NS - 1; ++i)
    if (f_valid) fetch_next(i);
  if (f_valid) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT_STEADY) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // ---- staggered halves.  Waves 0-3 (group A, the upper 128 rows) and waves 4-7 (group B, the lower
  // 128 rows) run the same two-phase step -- READ: epilogue of the finished tile if any, issue the DMA
  // of step h + NS - 1, load this step's 12 fragments into registers; COMPUTE: 32 (16) MFMAs out of
  // registers -- but B runs one phase (one barrier) behind A.  A CU places wave w and wave w + 4 on
  // the same SIMD, so on every SIMD one wave's MFMA phase runs beside the other wave's LDS reads, DMA
  // issue and epilogue VALU/stores instead of all 8 waves stalling on LDS together after each barrier.
  // Each wave waits for its OWN DMA pieces of step h + 1 before the barrier that precedes group A's read
  // of it: group A at the end of its COMPUTE phase, group B at the end of its READ phase.
  const bool grp_b = wave >= 4;  // wave-uniform (readfirstlane above)
  if (grp_b) {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

  int st = 0;        // ring stage of the step being computed
  int relaxed = 0;   // waits for which the previous epilogue's NST stores may still be in flight
  int64_t m0 = 0;
  int n0 = 0;
  f32x4_t acc[NJ][8];

  auto wait_dma = [&](bool issued) {
    // next step's stage must have landed: all but the youngest (NS-2) stages of DMA -- plus, for the
    // first NS-2 waits after a clean epilogue, its NST stores -- may stay in flight.  Once the cursor has
    // run out (the workgroup's last steps) fewer ops are in flight than the counts assume: vmcnt(0).
    if (!issued) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (relaxed > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT_RELAXED) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT_STEADY) : "memory");
    if (relaxed > 0) --relaxed;
  };

  auto epilogue = [&](int64_t m0, int n0) {
    // The lane id is re-derived behind an opaque asm so that the epilogue's per-lane address arithmetic is computed
    // here, once per tile, instead of being hoisted out of the tile loop and held in VGPRs across the k-loop
    // (with 128 accumulator + 48 fragment registers live there the hoisted values were spilled, and a scratch
    // reload is a vector-memory load: the compiler drains vmcnt -- the whole DMA ring -- in front of it).
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int lane = lane_e, r = lane_e & 15, q = lane_e >> 4;
    if (TR_EPI) {
      // lane (r, q) holds, per 16-row block i and sub-tile j, 4 consecutive n of row 16 i + r.  Write the
      // block as [16 rows][BN/4 bf16] (16-B chunk c of row rr at slot c ^ (rr & 7): conflict-free both
      // ways), read it back 16 B per lane in row order, store whole row segments.  Wave-private, and a
      // wave's LDS instructions execute in order: no barrier, no wait between the write and the read.
      const int64_t mrow0 = m0 + wm * 128;
      const int ncol0 = n0 + wn * (BN / 4);
      const bool full = m0 + BM <= g.M;  // workgroup-uniform
      const uint32_t sbase = lds_addr(scratch);
      // read-back positions of this lane: 16-B chunk idx = lane (+ 64) of the [16][ROW_CHUNKS] block
      uint32_t rd_addr[ROW_CHUNKS / 4 > 0 ? ROW_CHUNKS / 4 : 1];
      int rd_row[ROW_CHUNKS / 4 > 0 ? ROW_CHUNKS / 4 : 1], rd_col[ROW_CHUNKS / 4 > 0 ? ROW_CHUNKS / 4 : 1];
#pragma unroll
      for (int t = 0; t < ROW_CHUNKS / 4; ++t) {
        const int idx = lane + 64 * t;
        const int rr = idx / ROW_CHUNKS, c = idx % ROW_CHUNKS;
        rd_row[t] = rr;
        rd_col[t] = c * 8;
        rd_addr[t] = sbase + rr * (ROW_CHUNKS * 16) + ((c ^ (rr & (ROW_CHUNKS - 1))) << 4);
      }
      // software pipeline over the 8 row blocks: a wave's LDS instructions execute in order, so block
      // i + 1 may be written right behind the (un-waited) reads of block i; the reads of block i are
      // waited for with a COUNTED lgkmcnt (NJ writes + R reads of block i + 1 stay in flight) just before
      // its global stores, and block i + 1's GELU/pack VALU work runs under block i's LDS latency.
      // (the stores are non-temporal: QKV 165 -> 160 us, c_fc without GELU 228 -> 218 us in tools/gemm_bench; in the
      // tower, beside the non-temporal residual stream of layernorm.hip, -2.4 % on the step (QKV 181 -> 171, c_fc 277
      // -> 268, c_proj 244 -> 237 us).  The same hint on the attention kernel's stores or on the LayerNorm kernels'
      // 16-bit outputs LOSES in the tower -- re-used buffers that their consumers read from the cache -- as do
      // non-temporal loads of q|k|v, of the branch tensors and of the images.)
      constexpr int R = ROW_CHUNKS / 4;  // 16-B reads (= global stores) per lane per block
      auto tr_body = [&](auto act_tag) {
      constexpr int ACT = decltype(act_tag)::value;
      u32x4_t val[2][2];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          f32x4_t v = acc[j][i];
          if (EPI == EPI_BIAS_GELU_BF16) {
            if (ACT == 0) gelu_fast16_x4(v);
            if (ACT == 1) gelu_quick_fast_x4(v);
          }
          const unsigned long long pk = (unsigned long long)pack_bf2(v[0], v[1]) | ((unsigned long long)pack_bf2(v[2], v[3]) << 32);
          const int chunk = 2 * j + (q >> 1);
          lds_write_b64(sbase + r * (ROW_CHUNKS * 16) + ((chunk ^ (r & (ROW_CHUNKS - 1))) << 4) + (q & 1) * 8, pk);
        }
#pragma unroll
        for (int t = 0; t < R; ++t) lds_read_b128_nowait(rd_addr[t], val[i & 1][t]);
        if (i > 0) {  // block i - 1: everything but this block's NJ writes + R reads has completed
          if (R == 2) lds_wait_x2<NJ + R>(val[(i - 1) & 1][0], val[(i - 1) & 1][1]);
          else lds_wait_x1<NJ + R>(val[(i - 1) & 1][0]);
#pragma unroll
          for (int t = 0; t < R; ++t) {
            const int64_t m = mrow0 + (i - 1) * 16 + rd_row[t];
            if (full || m < g.M) st16_policy(reinterpret_cast<u32x4_t*>(g.out_hi + m * g.ldo + ncol0 + rd_col[t]), val[(i - 1) & 1][t]);
          }
        }
      }
      if (R == 2) lds_wait_x2<0>(val[1][0], val[1][1]);
      else lds_wait_x1<0>(val[1][0]);
#pragma unroll
      for (int t = 0; t < R; ++t) {
        const int64_t m = mrow0 + 7 * 16 + rd_row[t];
        if (full || m < g.M) st16_policy(reinterpret_cast<u32x4_t*>(g.out_hi + m * g.ldo + ncol0 + rd_col[t]), val[1][t]);
      }
      };
      if (EPI != EPI_BIAS_GELU_BF16 || g.act == 0) tr_body(std::integral_constant<int, 0>{});
      else if (g.act == 1) tr_body(std::integral_constant<int, 1>{});
      else tr_body(std::integral_constant<int, 2>{});
      return;
    }
    if (PERM && TAPCLIP_GELU_FULL_LINES && BN == 256) {
      // Full-line variant of the store below (round 4, VERDICT r03 "what's weak" 7: the direct fragment stores cover 16 rows
      // x 64 B per instruction -- half lines -- and c_fc wrote 435 MB for its 310 MB output).  A wave's 64 columns are ONE
      // 128-B line per row and lane (r, q) holds its chunks q (sub-tile pair 0) and 4 + q (pair 1).  Rows r and r ^ 8 trade
      // one chunk each through a DPP rotate by 8 inside the 16-lane row: lanes r < 8 end up with chunk q of rows r and r + 8,
      // lanes r >= 8 with chunk 4 + q of the same two rows, so each of the two store instructions of a 16-row block writes
      // 8 rows x 128 B -- whole lines.  Same number of stores, 12 extra VALU ops per block.
      auto body = [&](auto act_tag) {
        constexpr int ACT = decltype(act_tag)::value;
        const bool upper = r >= 8;  // this lane keeps the chunks 4 + q
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          u32x4_t pk[2];
#pragma unroll
          for (int J = 0; J < 2; ++J) {
            f32x4_t v0 = acc[2 * J][i], v1 = acc[2 * J + 1][i];
            if (ACT == 0) {
              gelu_fast16_x4(v0);
              gelu_fast16_x4(v1);
            }
            if (ACT == 1) {
              gelu_quick_fast_x4(v0);
              gelu_quick_fast_x4(v1);
            }
            pk[J][0] = pack_bf2(v0[0], v0[1]);
            pk[J][1] = pack_bf2(v0[2], v0[3]);
            pk[J][2] = pack_bf2(v1[0], v1[1]);
            pk[J][3] = pack_bf2(v1[2], v1[3]);
          }
          // send the chunk this lane does not keep (pair 1 from the lower rows, pair 0 from the upper ones) to lane r ^ 8
          u32x4_t got;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            got[e] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(upper ? pk[0][e] : pk[1][e]), 0x128 /* row_ror:8 */, 0xF, 0xF, false);
          const u32x4_t keep = upper ? pk[1] : pk[0];
          // lower lanes: (row r: own, row r + 8: received); upper lanes: (row r - 8: received, row r: own)
          const int64_t mA = m0 + wm * 128 + i * 16 + (r & 7), mB = mA + 8;
          bf16_t* col = g.out_hi + n0 + wn * (BN / 4) + 8 * q + (upper ? 32 : 0);
          if (mA < g.M) st16_policy(reinterpret_cast<u32x4_t*>(col + mA * g.ldo), upper ? got : keep);
          if (mB < g.M) st16_policy(reinterpret_cast<u32x4_t*>(col + mB * g.ldo), upper ? keep : got);
        }
      };
      if (g.act == 0) body(std::integral_constant<int, 0>{});
      else if (g.act == 1) body(std::integral_constant<int, 1>{});
      else body(std::integral_constant<int, 2>{});
      return;
    }
    if (PERM) {
      // activation -> bf16, 8 consecutive n per lane and sub-tile pair; the activation kind is decided once
      // per epilogue (wave-uniform), not per element
      auto body = [&](auto act_tag) {
        constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int64_t m = m0 + wm * 128 + i * 16 + r;
          if (m >= g.M) continue;
          bf16_t* orow = g.out_hi + m * g.ldo + n0 + wn * (BN / 4) + 8 * q;
#pragma unroll
          for (int J = 0; J < NJ / 2; ++J) {
            f32x4_t v0 = acc[2 * J][i], v1 = acc[2 * J + 1][i];
            if (ACT == 0) {
              gelu_fast16_x4(v0);
              gelu_fast16_x4(v1);
            }
            if (ACT == 1) {
              gelu_quick_fast_x4(v0);
              gelu_quick_fast_x4(v1);
            }
            u32x4_t pk;
            pk[0] = pack_bf2(v0[0], v0[1]);
            pk[1] = pack_bf2(v0[2], v0[3]);
            pk[2] = pack_bf2(v1[0], v1[1]);
            pk[3] = pack_bf2(v1[2], v1[3]);
            st16_policy(reinterpret_cast<u32x4_t*>(orow + 32 * J), pk);
          }
        }
      };
      if (g.act == 0) body(std::integral_constant<int, 0>{});
      else if (g.act == 1) body(std::integral_constant<int, 1>{});
      else body(std::integral_constant<int, 2>{});  // 7 (tools/gemm_bench): no activation
      return;
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
  };

  for (;;) {
    tile_origin(lid, m0, n0);
    int next_lid = lid + bpx;
    bool next_partial = false;
    bool has_next = !cur_partial && next_lid < chunk_hi;
    if (!has_next && !cur_partial && has_split) {
      has_next = next_partial = true;
      next_lid = split_lid;
    }
    const int ks_begin = cur_partial ? split_part * KSP : 0;
    const int ks_end = cur_partial ? ks_begin + KSP : KS;

    // accumulators start at the bias (lane holds n = n0 + wn * BN/4 + 16 j + 4 q + e for every m)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(bias_lds + n0 + wn * (BN / 4) + col_of(j));
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[j][i] = (EPI == EPI_PATCH_F32 || cur_partial) ? f32x4_t{0.f, 0.f, 0.f, 0.f} : bv;
    }

    for (int ks = ks_begin; ks < ks_end; ++ks) {
      // ================= READ phase
      // stage (st + NS - 1) % NS: its last readers (group B, one phase ago) passed the previous barrier
      const bool issued = f_valid;
      if (issued) fetch_next((st + NS - 1) % NS);
      const uint8_t* base = smem + st * STAGE;
      bf16x8_t wf[NJ], af[8];
#pragma unroll
      for (int j = 0; j < NJ; ++j) wf[j] = *reinterpret_cast<const bf16x8_t*>(base + w_base + w_sub_off(j));
#pragma unroll
      for (int i = 0; i < 8; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(base + a_base + i * 1024);
      if (grp_b) wait_dma(issued);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fragments in registers: the stage may be reused
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      // ================= COMPUTE phase
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j][i] = TAPCLIP_MFMA_16x16x32(wf[j], af[i], acc[j][i]);
      if (!grp_b) wait_dma(issued);
      st = (st + 1) % NS;
      if (ks != ks_end - 1) {  // (the barrier behind a tile's last COMPUTE phase is placed around the epilogue, below)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    }
    // ================= tile boundary: the finished tile's epilogue, ONE call site for both groups.
    // Group A passes the phase barrier first -- its epilogue opens what is its READ phase of the next tile's step 0.
    // Group B runs the epilogue straight behind its last COMPUTE phase, BEFORE that barrier: in the same phase as
    // group A's, so that the two waves of a SIMD do their epilogue VALU work (GELU, packing) and stores side by
    // side -- one wave alone issues a VALU op every 4 cycles, two together one every 2 -- instead of one after the
    // other with the partner parked at the barrier (two epilogue lengths per tile with the matrix pipe idle).
    // vmcnt bookkeeping is the same for both groups: the NST stores sit between the DMA of steps h + NS - 2 and
    // h + NS - 1 in issue order, so the next NS - 2 waits may leave them in flight.
    if (!TAPCLIP_EPI_EARLY_B || !grp_b) {
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    if (cur_partial) {
      // K-split part (always a workgroup's last item): raw fp32 accumulators (no bias, no activation) ->
      // split_ws[slot][256][BN]
      float* dst = g.split_ws + (size_t)((split_lid - g.split_from) * g.split_parts + split_part) * (BM * BN);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          *reinterpret_cast<f32x4_t*>(dst + (wm * 128 + i * 16 + r) * BN + wn * (BN / 4) + col_of(j)) = acc[j][i];
    } else {
      epilogue(m0, n0);
      // exactly NST stores were issued and the cursor is still issuing: the next NS-2 waits may skip them
      relaxed = (CLEAN_EPI && f_valid && (m0 + BM <= g.M)) ? NS - 2 : 0;
    }
    if (TAPCLIP_EPI_EARLY_B && grp_b) {
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    if (!has_next) break;
    lid = next_lid;
    cur_partial = next_partial;
  }
  // group A is one phase ahead and owes the barrier group B started with
  if (!grp_b) __builtin_amdgcn_s_barrier();
}

// ---- fix-up of the K-split tail tiles: out = epilogue(bias + sum of the parts)
template <int EPI, int BN, bool SPLIT = false>
__global__ __launch_bounds__(256) void splitk_fixup_kernel(GemmArgs g, int n_tiles) {
  const int tiles_n = g.N / BN;
  const int tiles_m = (int)((g.M + BM - 1) / BM);
  constexpr int CHUNKS = BM * BN / 4 / 256;
  const int t = blockIdx.x / CHUNKS;                     // tail tile index
  const int e4 = (blockIdx.x % CHUNKS) * 256 + threadIdx.x;  // float4 index inside the tile
  const int rr = e4 / (BN / 4), c4 = e4 % (BN / 4);
  // tile id -> origin (same grouped order as the GEMM kernel)
  const int id = g.split_from + t;
  const int GM = g.group_m;
  const int per_group = GM * tiles_n, grp = id / per_group, first_m = grp * GM;
  const int gsize = tiles_m - first_m < GM ? tiles_m - first_m : GM;
  const int in_grp = id - grp * per_group;
  const int64_t m = (int64_t)(first_m + in_grp % gsize) * BM + rr;
  const int n = (in_grp / gsize) * BN + c4 * 4;
  if (t >= n_tiles || m >= g.M) return;
  f32x4_t v = g.bias ? *reinterpret_cast<const f32x4_t*>(g.bias + n) : f32x4_t{0.f, 0.f, 0.f, 0.f};
  for (int p = 0; p < g.split_parts; ++p)
    v += *reinterpret_cast<const f32x4_t*>(g.split_ws + (size_t)(t * g.split_parts + p) * (BM * BN) + rr * BN + c4 * 4);
  if (EPI == EPI_BIAS_GELU_BF16) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = g.act == 0 ? (SPLIT ? gelu_erf(v[e]) : gelu_fast16(v[e])) : (SPLIT ? gelu_quick(v[e]) : gelu_quick_fast(v[e]));  // (as the GEMM's own epilogue, per precision)
  }
  if (EPI == EPI_BIAS_F32) {  // (the dX GEMMs of the prompt-tuning backward: fp32 gradients)
    *reinterpret_cast<f32x4_t*>(g.out_f32 + m * g.ldo + n) = v;
    return;
  }
  if (SPLIT) {  // split-bf16 outputs: hi and lo planes
    bf16_t h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split_bf(v[e], h[e], l[e]);
    *reinterpret_cast<uint2*>(g.out_hi + m * g.ldo + n) = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
    *reinterpret_cast<uint2*>(g.out_lo + m * g.ldo + n) = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
    return;
  }
  uint2 ph;
  ph.x = pack_bf2(v[0], v[1]);
  ph.y = pack_bf2(v[2], v[3]);
  *reinterpret_cast<uint2*>(g.out_hi + m * g.ldo + n) = ph;
}

constexpr int SPLIT_WS_TILES = 256;  // at most one partial tile per workgroup

// CUs a launch is sized for (one persistent workgroup each; a multiple of 8 so every XCD gets the same number)
static int device_cus() {
  static int n_cu_dev = 0;
  if (n_cu_dev == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    n_cu_dev = prop.multiProcessorCount / 8 * 8;
    if (n_cu_dev < 8) n_cu_dev = 8;
  }
  return n_cu_dev;
}

// K-split of the last, partial round (launch_t below): into how many parts its `rem` tiles are cut, 0 = whole tiles
static int tail_split_parts(int epi, bool split, int bn, int64_t tiles, int n_cu, int32_t K, bool have_ws) {
  static const bool no_tail_split = getenv("TAPCLIP_NO_TAIL_SPLIT") != nullptr;
  if (!(epi == EPI_BIAS_BF16 || epi == EPI_BIAS_GELU_BF16 || epi == EPI_BIAS_F32) || !have_ws || no_tail_split) return 0;
  const int64_t full = tiles / n_cu, rem = tiles - full * n_cu;
  const int ks = (split ? 3 : 1) * (K / BKS);
  if (!(rem > 0 && rem * 2 <= n_cu && (full == 0 || (bn == 256 && !split)))) return 0;
  int parts = (int)(n_cu / rem);
  // a part shorter than ~16 K steps is all pipeline fill and drain: it costs more than the idle CUs
  static const int min_ks = [] {
    const char* e = getenv("TAPCLIP_TAIL_MIN_KS");
    return e ? atoi(e) : 16;
  }();
  while (parts > 1 && (ks % parts != 0 || ks / parts < (min_ks > 4 ? min_ks : 4))) --parts;
  return (parts >= 2 && rem * parts <= SPLIT_WS_TILES) ? parts : 0;
}

template <int EPI, bool SPLIT, int BN>
hipError_t launch_t(const GemmArgs& a, hipStream_t s) {
  constexpr int NS = 4;
  static bool attr_set = false;
  const int smem_bytes = NS * (BM * BKS * 2 + BN * BKS * 2) + 8 * 2048 + MAX_N_BIAS * 4;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<EPI, SPLIT, BN, NS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int64_t tiles = ((a.M + BM - 1) / BM) * (a.N / BN);
  // persistent: one workgroup per CU
  const int n_cu_dev = device_cus();
  if (n_cu_dev == 0) return hipErrorUnknown;
  const int n_cu = (a.n_cu >= 8 && a.n_cu < n_cu_dev) ? a.n_cu / 8 * 8 : n_cu_dev;  // (a CU-masked stream: one workgroup per CU of the mask)
  int64_t nwg = tiles < n_cu ? (tiles + 7) / 8 * 8 : n_cu;
  GemmArgs b = a;
  static const int gm_env = [] {
    const char* e = getenv("TAPCLIP_GROUP_M");
    return e ? atoi(e) : 0;
  }();
  if (gm_env > 0) b.group_m = gm_env;
  // Tail split (bf16 outputs, 256-wide tiles): 591 tiles of an N = 768 GEMM are 2.31 rounds of 256 workgroups.
  // The whole rounds run as whole tiles; each of the remaining R tiles is computed by S = floor(256 / R)
  // workgroups over 1/S of K into fp32 partial tiles, summed by splitk_fixup_kernel.
  b.split_parts = 0;
  // The same machinery covers a GEMM with fewer tiles than half the CUs (the text tower's c_proj: 96 tiles of
  // K = 2048 on 256 CUs, 40 us): every tile is K-split (split_from = 0) and the grid grows to tiles x parts.
  // (EPI_BIAS_F32: the K = 1536 / 2048, N = 512 dX GEMMs of the text tower's backward -- one partial round of 48 - 96 tiles
  // and 48 - 64 dependent k-steps, 37.5 us; K-split two ways + the fix-up: see DESIGN.md section 6)
  // (split-bf16, round 3: only the second case -- EVERY tile of the launch split alike, so a row's bits depend on the
  // launch's M and never on which rows share it; its three products are one K sequence of 3 K / 32 steps, cut anywhere.
  // The text tower of the fp16 default mode at 65 classes: c_proj 192 dependent steps on 96 of 256 CUs.)
  {
    const int parts = tail_split_parts(EPI, SPLIT, BN, tiles, n_cu, a.K, a.split_ws != nullptr);
    if (parts >= 2) {
      const int64_t full = tiles / n_cu, rem = tiles - full * n_cu;
      b.split_from = (int)(full * n_cu);
      b.split_parts = parts;
      if (full == 0) nwg = (rem * parts + 7) / 8 * 8;
    }
  }
  hipLaunchKernelGGL((gemm256_kernel<EPI, SPLIT, BN, NS>), dim3((unsigned)nwg), dim3(512), smem_bytes, s, b);
  if (b.split_parts > 0) {
    const int n_tail = (int)(tiles - b.split_from);
    if constexpr (EPI == EPI_BIAS_BF16 || EPI == EPI_BIAS_GELU_BF16 || EPI == EPI_BIAS_F32)
      hipLaunchKernelGGL((splitk_fixup_kernel<EPI, BN, SPLIT>), dim3((unsigned)(n_tail * (BM * BN / 4 / 256))), dim3(256), 0, s, b, n_tail);
  }
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
  // 256-wide tiles have 1.5x the FLOP per staged byte (the main loop is L2->LDS bound) but half as many tiles: take them
  // unless their last, partial round wastes more than that gains.  Round 5: the estimate knows the K-split of that round
  // (a split tail costs ~0.55 of a round, not a whole one) and prices a 128-wide round at 0.7 of a 256-wide one (measured
  // 0.65 at K = 768, 0.76 at K = 3072; it assumed 0.625).  Measured at the row counts of small batches (tools/gemm_sweep.sh,
  // profiles/r05_gemm_small_batch_sweep.log): c_proj at 25 216 rows (batch 128) 175 -> 116 us, out_proj 55 -> 45, c_fc /
  // c_proj at 6 304 rows (batch 32) 54 -> 50 / 57 -> 48; every choice at 50 432 rows (batch 256) is the one made before.
  bool wide = false;
  if (a.N % 256 == 0) {
    const int n_cu = (a.n_cu >= 8 && a.n_cu < device_cus()) ? a.n_cu / 8 * 8 : device_cus();
    auto rounds = [&](int bn, double per_round) {
      const int64_t tiles = tiles_m * (a.N / bn);
      const int64_t full = tiles / n_cu, rem = tiles - full * n_cu;
      if (rem == 0) return (double)full * per_round;
      const bool cut = tail_split_parts(EPI, split, bn, tiles, n_cu, a.K, a.split_ws != nullptr) >= 2;
      return ((double)full + (cut ? 0.55 : 1.0)) * per_round;
    };
    const double t256 = n_cu > 0 ? rounds(256, 1.0) : 0.0, t128 = n_cu > 0 ? rounds(128, 0.7) : 1.0;
    wide = forced_bn == 256 || (forced_bn != 128 && t256 <= t128);
  }
  // (the GELU-backward epilogue used to be pinned to 128-wide tiles: at 256 it spilled 84 B/lane -- its exp + erf polynomial
  // beside 128 accumulator registers.  Since the epilogue moved out of the k-loop (round 3) the 256-wide instantiation
  // allocates 249 VGPRs without scratch, and the text tower's N = 2048 recompute GEMM runs one round of 192 tiles.)
  if (wide) return split ? launch_t<EPI, true, 256>(a, s) : launch_t<EPI, false, 256>(a, s);
  return split ? launch_t<EPI, true, 128>(a, s) : launch_t<EPI, false, 128>(a, s);
}

}  // namespace

size_t gemm256_split_ws_bytes() { return (size_t)SPLIT_WS_TILES * BM * 256 * sizeof(float); }

bool gemm256_supports(const GemmArgs& a) {
  return a.N <= MAX_N_BIAS && a.K % BKS == 0 && a.K >= 4 * BKS && (a.M * a.lda * 2 < (int64_t)0xFFFF0000) &&
         ((int64_t)a.N * a.K * 2 < (int64_t)0xFFFF0000);
}

hipError_t launch_gemm256(const GemmArgs& a, int epilogue, bool split, hipStream_t s) {
  switch (epilogue) {
    case EPI_BIAS_BF16: return launch_e<EPI_BIAS_BF16>(a, split, s);
    case EPI_BIAS_GELU_BF16: return launch_e<EPI_BIAS_GELU_BF16>(a, split, s);
    case EPI_BIAS_RESID_F32: return launch_e<EPI_BIAS_RESID_F32>(a, split, s);
    case EPI_PATCH_F32: return launch_e<EPI_PATCH_F32>(a, split, s);
    case EPI_BIAS_F32: return launch_e<EPI_BIAS_F32>(a, split, s);
    case EPI_GELU_BWD_BF16: return launch_e<EPI_GELU_BWD_BF16>(a, split, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace tapclip
TAPCLIP_TU_NO_PK_F32_END
