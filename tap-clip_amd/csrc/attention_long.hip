// Attention core for LONG sequences (T > 256 keys, e.g. ViT-L/14@336: 577 tokens), 16-bit operands, no mask: the second
// generation of the flash-style kernel of attention.hip (which keeps the split-bf16 and causal cases).  Same contract
// (kernels.h AttnArgs; the scaled-dot-product step inside nn.MultiheadAttention of open_clip's
// ResidualAttentionBlock, reference call site models/clip_wrapper.py:46-47).
//
// Its own translation unit for two build switches (csrc/Makefile):
//  * -fno-honor-nans: the score maxima are plain fmaxf, which hipcc then folds into v_max3_f32 without canonicalising the MFMA
//    outputs first (flash2_tile_max below, and why no asm statement reads an MFMA result any more).
//  * no packed-fp32 VALU ops (common.h TAPCLIP_TU_NO_PK_F32), like every kernel file that runs VALU arithmetic beside LDS-fed MFMAs
//    (the round-2 erratum: op_sel = [0,1,..] encodings, which hipcc may form from any packed multiply).  For a while in round 5
//    this file was thought to show a SECOND erratum -- NaN / far-off rows in 15-25 of 1.18 M query rows per launch with
//    v_pk_mul_f32 .. op_sel_hi:[1,0] rescaling the O accumulators.  It did not: those rows came from the asm maximum reading MFMA
//    results too early (profiles/r05_flash2_asm_hazard.txt); the build that keeps packed ops is clean now and no faster.
#include <cstdlib>

#ifndef TAPCLIP_TU_NO_NANS
#error "attention_long.hip is built with -fno-honor-nans -DTAPCLIP_TU_NO_NANS (csrc/Makefile): its maxima are plain fmaxf"
#endif
#ifndef TAPCLIP_AB_KEEP_PK  // (tools/: the A/B build that keeps them)
#define TAPCLIP_TU_NO_PK_F32
#endif
#include "common.h"
#include "kernels.h"
#include "attn_store.h"

namespace tapclip {
namespace {

// ---- What the ablation of the first flash kernel showed (profiles/r04_flash_attention_ablation.txt: 97 of its 164 us were neither
// VALU nor MFMA) and what its traffic looked like (profiles/r05_pmc_attention_long.json: 1.06 GB fetched per launch for a 454 MB
// q|k|v) decide the shape.  Measured at ViT-L/14@336, batch 128: 332-346 -> 258 us per launch (tools/attn_bench;
// profiles/r05_attn_bench_*.log, r05_flash2_ablation.log; the steps in between: docs/HISTORY.md Part A (a)).
//  * K/V blocks (64 keys) go global -> LDS by LDS-DMA into a ring of NS stages, ONE barrier per block, the DMA of block
//    j + NS - 1 in flight under the products of block j -- no staging registers, no second barrier, no LDS store issue.
//    The DMA writes lane-linear 1-KiB pieces (8 key rows), so both images are swizzled on the SOURCE side at 16-byte
//    granularity: K position p of row k holds chunk p ^ (k & 7) (conflict-free ds_read_b128 of the A fragments), V position p
//    holds chunk p ^ ((k >> 1) & 3).  The transposed V reads stay conflict-free at that granularity because address lane pp of
//    product dt takes the d-quad 4 pp + (dt ^ (pp & 1)): the two halves of every 16-byte chunk are read by different lanes of
//    one instruction (the 8-byte swizzle of the kernels above cannot be produced by a 16-byte DMA).  A lane still ends up with
//    16 consecutive d of its query -- lanes with odd g hold them with products 0 <-> 1, 2 <-> 3 exchanged (undone at the store).
//  * ALL the query tiles of a wave (QT) use each K / V fragment read: with one tile per read (16 KB of LDS reads per 16 MFMAs)
//    the LDS array ran at the MFMA pipe's own rate.
//  * the row sums of P come from the matrix pipe (an all-ones A fragment: one MFMA per 32 keys and tile) instead of 16 VALU adds
//    per lane: the loop is VALU-issue bound (16 exp + ~50 plain ops per tile and block beside 16 MFMAs), the matrix pipe is not.
//    They are the sums of the ROUNDED probabilities, i.e. of exactly what multiplies V.
//  * grid: the query chunks of one (sequence, head) sit 8 workgroup ids apart -- same XCD under round-robin placement,
//    dispatched together -- so the later reads of a head's K/V are L2 hits (dim3(pairs, chunks) put them n_seq * H ids apart:
//    every chunk re-read its 148 KB from HBM / Infinity Cache).
//  * the last key block is run with as many 16-key tiles as it holds (577 = 9 x 64 + 1: one tile instead of four).
//  * LAZY > 0: the running maximum of a query moves only when a block's maximum exceeds it by more than LAZY (in log2 units;
//    un-normalised probabilities then reach 2^LAZY instead of 1 -- the same relative precision in a floating-point P, and
//    65504 is far away), so the O accumulators are rescaled in the first block and after that almost never.
//  * two things hipcc must be kept from: (1) it puts `s_waitcnt vmcnt(0)` in front of the ds_read_tr BUILTIN while an LDS-DMA is
//    outstanding (no alias information: the read could be of what the DMA writes), so the transposed V reads are inline asm with
//    hand-counted lgkmcnt; (2) global loads still pending at the loop's entry (the Q fragments) turn their first use inside the
//    loop into a vmcnt(0) in EVERY iteration, draining the DMAs issued behind them -- the fragments are "consumed" by an empty
//    asm before the first DMA.  And none of its geometries may spill (tests/test_build_resources.py): scratch traffic would sit in
//    the hand-counted vmcnt queue.
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

#ifndef TAPCLIP_FLASH2_LAZY
#define TAPCLIP_FLASH2_LAZY 8
#endif
#ifndef TAPCLIP_FLASH2_ABL
#define TAPCLIP_FLASH2_ABL 0  // timing-only ablations (tools/Makefile attn_bench_alt): 1 no softmax VALU, 2 no MFMAs, 4 no LDS fragment reads, 8 no DMA / barriers after the first block
#endif
#ifndef TAPCLIP_FLASH2_VAHEAD
#define TAPCLIP_FLASH2_VAHEAD 1  // V fragments in flight ahead of their products (1 .. 3 of the 4 or 8 per step)
#endif
#ifndef TAPCLIP_FLASH2_KAHEAD
#define TAPCLIP_FLASH2_KAHEAD 1  // K fragment reads in flight ahead of their products, in key tiles (1 .. 3)
#endif
__device__ __forceinline__ f32x4_t f2mm(const bf16x8_t& x, const bf16x8_t& y, const f32x4_t& c) {
  if constexpr ((TAPCLIP_FLASH2_ABL & 2) != 0) {
    asm volatile("" ::"v"(x), "v"(y));
    return c;
  } else {
    return TAPCLIP_MFMA_16x16x32(x, y, c);
  }
}

struct Flash2Lane {
  int k_off0;  // K fragment reads: row r, chunk (g | 4 s) ^ (r & 7): s = 1 is this offset ^ 64
  int v_off0;  // V transposed reads of product dt: row 4 g + qq, chunk ((2 pp + (dt >> 1)) ^ kappa), half (dt ^ pp) & 1: this offset ^ (16 (dt >> 1) | 8 (dt & 1))
  int g;
};

// maximum of a query's NKT x 4 scores in this lane and of the same query's other three lanes.  Plain fmaxf: this file is compiled
// with -fno-honor-nans (Makefile; the operands are finite, masked scores are -inf), so hipcc neither canonicalises the MFMA
// outputs first (one v_max x, x, x per score) nor keeps from folding two scores per v_max3_f32.  Until round 5 this was an asm
// statement of raw v_max3_f32 -- and hipcc does NOT look inside an asm statement for the MFMA -> VALU read hazard: gfx950 has no
// interlock there (an 8-pass MFMA's result may be read 11 wait states after its issue; the compiler pads its OWN instructions
// only).  With one query tile per wave the statement followed the last product directly and read the last key tile's scores
// four instructions later -- stale registers: a block maximum that missed its largest scores, P beyond the half range, NaN rows
// (found with the base-2 build; profiles/r05_flash2_asm_hazard.txt).  No asm statement of this library reads an MFMA result now.
template <int NKT>
__device__ __forceinline__ float flash2_tile_max(const f32x4_t (&sc)[NKT]) {
  float d = fmaxf(sc[0][0], sc[0][1]);
  d = fmaxf(fmaxf(d, sc[0][2]), sc[0][3]);
#pragma unroll
  for (int kt = 1; kt < NKT; ++kt) {
    d = fmaxf(fmaxf(d, sc[kt][0]), sc[kt][1]);
    d = fmaxf(fmaxf(d, sc[kt][2]), sc[kt][3]);
  }
  auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(d), __float_as_uint(d), false, false);
  d = fmaxf(__uint_as_float(r16[0]), __uint_as_float(r16[1]));
  auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(d), __float_as_uint(d), false, false);
  return fmaxf(__uint_as_float(r32[0]), __uint_as_float(r32[1]));
}

// one key block (NKT 16-key tiles) for the NQ query tiles of a wave
// QL2: q arrives in log2 units (AttnArgs::q_log2: log2(e) folded into Wq, bq at pack time) and the score accumulators start at
// -m, so exp2 takes the MFMA output as it is
template <int QT, int NQ, int NKT, bool MASK, bool QL2>
__device__ __forceinline__ void flash2_step(const uint8_t* Kb, const uint8_t* Vb, const Flash2Lane& ln, const bf16x8_t (&qh)[QT][2],
                                            f32x4_t (&oc)[QT][4], float (&m)[QT], float (&l)[QT], f32x4_t (&nm)[QT], int key_base, int T, bool first) {
  constexpr float LOG2E = 1.44269504088896340736f;
  constexpr int LAZY = TAPCLIP_FLASH2_LAZY;
  f32x4_t sc[NQ][NKT];
  {
    // K fragments KA key tiles ahead of their products (the scheduler, left alone, hoists all eight reads: 32 registers)
    constexpr int KA = TAPCLIP_FLASH2_KAHEAD < NKT ? TAPCLIP_FLASH2_KAHEAD : NKT;
    constexpr int ABL = TAPCLIP_FLASH2_ABL;
    bf16x8_t kf[KA + 1][2];
    auto kread = [&](int kt, int slot) {
      if constexpr ((ABL & 4) != 0) {
        kf[slot][0] = qh[0][0];
        kf[slot][1] = qh[0][1];
      } else {
        kf[slot][0] = *reinterpret_cast<const bf16x8_t*>(Kb + ln.k_off0 + kt * 2048);
        kf[slot][1] = *reinterpret_cast<const bf16x8_t*>(Kb + (ln.k_off0 ^ 64) + kt * 2048);
      }
    };
#pragma unroll
    for (int kt = 0; kt < KA; ++kt) kread(kt, kt);
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (kt + KA < NKT) kread(kt + KA, (kt + KA) % (KA + 1));
#pragma unroll
      for (int t = 0; t < NQ; ++t) {
        sc[t][kt] = f2mm(kf[kt % (KA + 1)][0], qh[t][0], QL2 ? nm[t] : (f32x4_t{0.f, 0.f, 0.f, 0.f}));
        sc[t][kt] = f2mm(kf[kt % (KA + 1)][1], qh[t][1], sc[t][kt]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const bf16_t one = f2bf(1.0f);
  const s16x8_t ov = {(short)one, (short)one, (short)one, (short)one, (short)one, (short)one, (short)one, (short)one};
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ov);
  bf16x8_t ph[NQ][(NKT + 1) / 2];
#pragma unroll
  for (int t = 0; t < NQ; ++t) {
    if (MASK) {
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (key_base + kt * 16 + 4 * ln.g + e >= T) sc[t][kt][e] = -INFINITY;
    }
    float bm;
    if constexpr ((TAPCLIP_FLASH2_ABL & 1) != 0) bm = 0.f;
    else bm = flash2_tile_max<NKT>(sc[t]);  // finite: every block holds at least one key of the sequence
    if constexpr (QL2) {
      // scores arrive as (q . k) log2e - m log2e (m[] holds m log2e, nm[] its negative as the accumulators' start; both 0 before
      // the first block): bm is the block's maximum RELATIVE to the running one.  It moves -- and everything held at the old
      // maximum is rescaled, once -- in the first block and whenever it would be exceeded by more than LAZY.
      if (first || LAZY == 0 || __builtin_amdgcn_ballot_w64(bm > (float)LAZY) != 0) {  // wave-uniform
        const float up = first ? bm : fmaxf(bm, 0.f);
        m[t] += up;
        if (!first) {  // (first block: l and O are zeros, and exp2(-bm) may overflow)
          const float alpha = __builtin_amdgcn_exp2f(-up);
          l[t] *= alpha;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int e = 0; e < 4; ++e) oc[t][dt][e] *= alpha;
        }
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
          for (int e = 0; e < 4; ++e) sc[t][kt][e] -= up;
        nm[t] = f32x4_t{-m[t], -m[t], -m[t], -m[t]};
      }
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if constexpr ((TAPCLIP_FLASH2_ABL & 1) == 0) sc[t][kt][e] = __builtin_amdgcn_exp2f(sc[t][kt][e]);
    } else {
    // (m starts at -inf: the first block always moves it; alpha = exp2(-inf) = 0 meets zeros)
    if (LAZY == 0 || __builtin_amdgcn_ballot_w64(bm * LOG2E > m[t] * LOG2E + (float)LAZY) != 0) {  // wave-uniform
      const float m_new = fmaxf(m[t], bm);
      const float alpha = __builtin_amdgcn_exp2f((m[t] - m_new) * LOG2E);
      m[t] = m_new;
      l[t] *= alpha;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int e = 0; e < 4; ++e) oc[t][dt][e] *= alpha;
    }
    const float nmx = -m[t] * LOG2E;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if constexpr ((TAPCLIP_FLASH2_ABL & 1) == 0) sc[t][kt][e] = __builtin_amdgcn_exp2f(fmaf(sc[t][kt][e], LOG2E, nmx));
    }
#pragma unroll
    for (int s2 = 0; s2 < (NKT + 1) / 2; ++s2) {
      bf16_t h[8];
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) h[jj] = (2 * s2 + (jj >> 2) < NKT) ? f2bf(sc[t][(2 * s2 + (jj >> 2)) < NKT ? 2 * s2 + (jj >> 2) : 0][jj & 3]) : (bf16_t)0;
      const s16x8_t hv = {(short)h[0], (short)h[1], (short)h[2], (short)h[3], (short)h[4], (short)h[5], (short)h[6], (short)h[7]};
      ph[t][s2] = __builtin_bit_cast(bf16x8_t, hv);
    }
    // row sum of the block on the matrix pipe: every row of A is ones, so every element of the result is the query's sum
    f32x4_t bs = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s2 = 0; s2 < (NKT + 1) / 2; ++s2) bs = f2mm(ones, ph[t][s2], bs);
    l[t] += bs[0];
  }
  {
    // V fragments one product ahead of their use, read by INLINE ASM: hipcc puts an s_waitcnt vmcnt(0) in front of the
    // ds_read_tr builtin whenever an LDS-DMA is outstanding (no alias information on the builtin: it could be reading what the
    // DMA writes) -- every step then waited for the block it had just requested, two blocks ahead.  The waits for the
    // fragments are therefore hand-counted too (lgkmcnt retires LDS reads in order; nothing else of this wave is in the
    // queue here: the K reads above were consumed by their products).
    // (NKT = 1: the upper half of the k range multiplies zeros into a re-read of the same four keys)
    constexpr int NP = 4 * ((NKT + 1) / 2);
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    const uint32_t vb = (uint32_t)(uintptr_t)(lds_void_t*)Vb;
    uint32_t va[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) va[dt] = vb + (uint32_t)(ln.v_off0 ^ (16 * (dt >> 1) + 8 * (dt & 1)));
    constexpr int VA = TAPCLIP_FLASH2_VAHEAD < NP ? TAPCLIP_FLASH2_VAHEAD : NP - 1;
    u32x2_t lo[VA + 1], hi[VA + 1];
    auto vissue = [&](int i) {
      const int s2 = i >> 2, dt = i & 3, b = i % (VA + 1);
      if constexpr ((TAPCLIP_FLASH2_ABL & 4) == 0) {
        if (s2 == 0) {
          asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo[b]) : "v"(va[dt]));
          if (NKT > 1) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(hi[b]) : "v"(va[dt]));
          else asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi[b]) : "v"(va[dt]));
        } else {
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:4096" : "=v"(lo[b]) : "v"(va[dt]));
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:6144" : "=v"(hi[b]) : "v"(va[dt]));
        }
      } else {
        lo[b] = u32x2_t{va[dt], va[dt]};
        hi[b] = lo[b];
      }
    };
#pragma unroll
    for (int i = 0; i < VA; ++i) vissue(i);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      if (i + VA < NP) vissue(i + VA);
      const int b = i % (VA + 1);
      if constexpr ((TAPCLIP_FLASH2_ABL & 4) == 0) {
        // fragment i has landed once at most the reads issued after it are outstanding: two per fragment, min(VA, NP - 1 - i) fragments
        constexpr int dummy = 0;
        (void)dummy;
        const int later = (NP - 1 - i) < VA ? (NP - 1 - i) : VA;
        if (later >= 3) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(lo[b]), "+v"(hi[b]));
        else if (later == 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(lo[b]), "+v"(hi[b]));
        else if (later == 1) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(lo[b]), "+v"(hi[b]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[b]), "+v"(hi[b]));
      }
      const u32x4_t v4 = {lo[b][0], lo[b][1], hi[b][0], hi[b][1]};
      const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, v4);
#pragma unroll
      for (int t = 0; t < NQ; ++t) oc[t][i & 3] = f2mm(vf, ph[t][i >> 2], oc[t][i & 3]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

struct Flash2Ctx {
  const uint8_t* qkv8;
  int64_t row0_bytes;  // byte offset of the sequence's first q|k|v row
  int64_t ld_bytes;
  int k_src, v_src;    // byte offset inside a row of this lane's K / V source chunk
  int drow;            // row of the lane inside its 8-row piece
  int wave, T;
};

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction takes an immediate)
template <int MAXN>
__device__ __forceinline__ void wait_vm_rt(int n) {
  if constexpr (MAXN == 0) {
    wait_vm<0>();
  } else {
    if (n >= MAXN) wait_vm<MAXN>();
    else wait_vm_rt<MAXN - 1>(n);
  }
}

template <int WAVES, int QT, int NS, int KB, int NQ, bool QL2>
__device__ __forceinline__ void flash2_run(const AttnArgs& a, uint8_t* smem, const Flash2Ctx& cx, const Flash2Lane& ln, const bf16x8_t (&qh)[QT][2],
                                           int tile0, int64_t row0, int head, int r) {
  constexpr int KEYS = KB * 16;                       // keys per block (KB 16-key tiles: 4 or 2)
  constexpr int VOFF = KEYS * 128;                    // [KEYS][128 B] K image | the same for V
  constexpr int STAGE = 2 * VOFF;
  constexpr int NPC = STAGE / 1024;                   // a block is NPC 1-KiB pieces (half K, half V): wave w moves pieces w, w + WAVES, ...
  constexpr int NPW = (NPC + WAVES - 1) / WAVES;
  const int T = cx.T;
  const int my_pieces = (cx.wave + WAVES * (NPW - 1) < NPC) ? NPW : NPW - 1;  // wave-uniform (12 waves: four waves move two, eight one)
  // source of piece i of a block: a wave-uniform block base (scalar registers) + a 32-bit lane offset that never changes -- the
  // address arithmetic of a block is scalar (per-lane 64-bit row * stride products were 80 VALU cycles per piece: a fifth of
  // the step).  Only the request for the LAST, partial block clamps its rows (a copy of the sequence's last row stands in for
  // the rows past it: masked keys; finite V against P = 0) and pays a multiply.
  uint32_t dma_off[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int piece = cx.wave + WAVES * i;
    dma_off[i] = (uint32_t)(((piece < NPC / 2 ? piece : piece - NPC / 2) * 8 + cx.drow) * (int)cx.ld_bytes + (piece < NPC / 2 ? cx.k_src : cx.v_src));
  }
  auto dma_block = [&](int kb, int slot) {
    const uint8_t* base = cx.qkv8 + cx.row0_bytes + (int64_t)kb * (KEYS * cx.ld_bytes);  // wave-uniform
    const bool partial = (kb + 1) * KEYS > T;                                            // wave-uniform
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int piece = cx.wave + WAVES * i;
      if (piece < NPC) {  // wave-uniform
        uint32_t off = dma_off[i];
        if (partial) {
          int key = (piece < NPC / 2 ? piece : piece - NPC / 2) * 8 + cx.drow;
          key = kb * KEYS + key < T ? key : T - 1 - kb * KEYS;
          off = (uint32_t)(key * (int)cx.ld_bytes + (piece < NPC / 2 ? cx.k_src : cx.v_src));
        }
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(base + off), (lds_void_t*)(smem + slot * STAGE + piece * 1024), 16, 0, 0);
      }
    }
  };
  float m[QT], l[QT];
  f32x4_t oc[QT][4], nm[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    m[t] = QL2 ? 0.f : -INFINITY;
    nm[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    l[t] = 0.f;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oc[t][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }

  const int n_kb = (T + KEYS - 1) / KEYS, n_full = T / KEYS;
#pragma unroll
  for (int i = 0; i < NS - 1; ++i)
    if (i < n_kb) dma_block(i, i);
  int slot = 0;
  // block kb has landed once at most the blocks issued after it (my_pieces DMA instructions each) are outstanding
  auto block_ready = [&](int kb) {
    if (!(TAPCLIP_FLASH2_ABL & 8) || kb == 0) {
      const int later = (n_kb - 1 - kb) < (NS - 2) ? (n_kb - 1 - kb) : (NS - 2);
      wait_vm_rt<(NS - 2) * NPW>((TAPCLIP_FLASH2_ABL & 8) ? 0 : later * my_pieces);
      __builtin_amdgcn_s_barrier();  // every wave's pieces of block kb are in LDS; every wave is done with block kb - 1
      asm volatile("" ::: "memory");
    }
  };
  // the full blocks: ONE step variant inside the loop (with the tail variants beside it the accumulators changed registers
  // from variant to variant and were copied back at the loop's end)
  for (int kb = 0; kb < n_full; ++kb) {
    block_ready(kb);
    if (kb + NS - 1 < n_kb && !(TAPCLIP_FLASH2_ABL & 8)) dma_block(kb + NS - 1, slot == 0 ? NS - 1 : slot - 1);  // into the slot of block kb - 1
    if constexpr (NQ > 0) flash2_step<QT, NQ, KB, false, QL2>(smem + slot * STAGE, smem + slot * STAGE + VOFF, ln, qh, oc, m, l, nm, kb * KEYS, T, kb == 0);
    slot = slot + 1 == NS ? 0 : slot + 1;
  }
  if (n_full < n_kb) {  // the last, partial block: 1 .. 63 keys (nothing left to request)
    block_ready(n_full);
    if constexpr (NQ > 0) {
      const uint8_t* Kb = smem + slot * STAGE;
      const int rem = T - n_full * KEYS;
      if (rem <= 16) flash2_step<QT, NQ, 1, true, QL2>(Kb, Kb + VOFF, ln, qh, oc, m, l, nm, n_full * KEYS, T, n_full == 0);
      else if (rem <= 32 || KB == 2) flash2_step<QT, NQ, 2, true, QL2>(Kb, Kb + VOFF, ln, qh, oc, m, l, nm, n_full * KEYS, T, n_full == 0);
      else flash2_step<QT, NQ, (KB == 4 ? 4 : 2), true, QL2>(Kb, Kb + VOFF, ln, qh, oc, m, l, nm, n_full * KEYS, T, n_full == 0);
    }
  }
#pragma unroll
  for (int t = 0; t < NQ; ++t) {
    const int qi = (tile0 + WAVES * t) * 16 + r;
    const float inv = 1.0f / l[t];
    // lanes with odd g hold their 16 consecutive d with products 0 <-> 1 and 2 <-> 3 exchanged
    f32x4_t oo[4];
    const bool odd = (ln.g & 1) != 0;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int e = 0; e < 4; ++e) oo[dt][e] = odd ? oc[t][dt ^ 1][e] : oc[t][dt][e];
    if (a.out_q != nullptr) store_o_mx8(a, oo, inv, row0 + qi, head, ln.g, qi < T);
    else if (qi < T) store_o_bf16<false>(a, oo, inv, row0 + qi, head, ln.g);
  }
}

// WAVES waves per workgroup, QT query tiles per wave (a chunk = WAVES * QT * 16 queries), WPS = waves per SIMD the build is
// held to (register budget 512 / WPS)
template <int WAVES, int QT, int NS, int KB, int WPS, bool QL2>
__global__ __launch_bounds__(WAVES * 64, WPS) void attn_flash2_kernel(AttnArgs a, int chunks) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int T = a.T, D = a.D;
  // workgroup id -> ((sequence, head), query chunk): the chunks of a pair are 8 ids apart
  const int per = 8 * chunks;
  const int grp = blockIdx.x / per, rem = blockIdx.x - grp * per;
  const int pair = grp * 8 + (rem & 7), chunk = rem >> 3;
  if (pair >= a.n_seq * a.H) return;  // (whole workgroup: the grid is rounded up to 8 pairs)
  const int seq = pair / a.H, head = pair - seq * a.H;
  const int64_t row0 = (int64_t)seq * T;
  const int64_t ld = 3 * (int64_t)D;
  const int qcol = head * 64, kcol = D + head * 64, vcol = 2 * D + head * 64;
  const int n_qt = (T + 15) >> 4;
  const int tile0 = chunk * (WAVES * QT) + wave;  // round-robin deal: tiles tile0, tile0 + WAVES, ...
  int nq = 0;                                     // wave-uniform
#pragma unroll
  for (int t = 0; t < QT; ++t) nq += (tile0 + WAVES * t < n_qt) ? 1 : 0;

  bf16x8_t qh[QT][2];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    int qc = (tile0 + WAVES * t) * 16 + r;
    if (qc >= T) qc = T - 1;
#pragma unroll
    for (int s = 0; s < 2; ++s) qh[t][s] = *reinterpret_cast<const bf16x8_t*>(a.qkv_hi + (row0 + qc) * ld + qcol + 32 * s + 8 * g);
  }

  // the Q fragments are consumed HERE as far as the compiler's wait-count pass can tell: left pending, their first use inside
  // the key loop gets an s_waitcnt vmcnt(0) that also drains the block DMAs issued behind them, in every iteration
#pragma unroll
  for (int t = 0; t < QT; ++t) asm volatile("" ::"v"(qh[t][0]), "v"(qh[t][1]));

  Flash2Ctx cx;
  cx.qkv8 = reinterpret_cast<const uint8_t*>(a.qkv_hi);
  cx.row0_bytes = row0 * ld * 2;
  cx.ld_bytes = ld * 2;
  cx.drow = lane >> 3;
  cx.k_src = kcol * 2 + (((lane & 7) ^ cx.drow) << 4);
  cx.v_src = vcol * 2 + (((lane & 7) ^ ((cx.drow >> 1) & 3)) << 4);
  cx.wave = wave;
  cx.T = T;

  Flash2Lane ln;
  ln.g = g;
  ln.k_off0 = r * 128 + ((g ^ (r & 7)) << 4);
  const int qq = r >> 2, pp = r & 3;
  const int kappa = 2 * (g & 1) + (qq >> 1);  // ((4 g + qq) >> 1) & 3
  ln.v_off0 = (4 * g + qq) * 128 + (((2 * pp) ^ kappa) << 4) + ((pp & 1) << 3);

  if (nq == QT) flash2_run<WAVES, QT, NS, KB, QT, QL2>(a, smem, cx, ln, qh, tile0, row0, head, r);
#ifdef TAPCLIP_FLASH2_NQ_FULL  // (debug: a wave with any valid tile computes all of its tiles)
  else if (nq > 0) flash2_run<WAVES, QT, NS, KB, QT, QL2>(a, smem, cx, ln, qh, tile0, row0, head, r);
#endif
  else if (QT > 1 && nq == 1) flash2_run<WAVES, QT, NS, KB, 1, QL2>(a, smem, cx, ln, qh, tile0, row0, head, r);
  else if (QT > 2 && nq == 2) flash2_run<WAVES, QT, NS, KB, (QT > 2 ? 2 : 0), QL2>(a, smem, cx, ln, qh, tile0, row0, head, r);
  else if (QT > 3 && nq == 3) flash2_run<WAVES, QT, NS, KB, (QT > 3 ? 3 : 0), QL2>(a, smem, cx, ln, qh, tile0, row0, head, r);
  else flash2_run<WAVES, QT, NS, KB, 0, QL2>(a, smem, cx, ln, qh, tile0, row0, head, r);
}

template <int WAVES, int QT, int NS, int KB, int WPS, bool QL2 = false>
hipError_t launch_flash2_cfg(const AttnArgs& a, hipStream_t s) {
  constexpr int smem_bytes = NS * KB * 4096;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_flash2_kernel<WAVES, QT, NS, KB, WPS, QL2>), hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int n_qt = (a.T + 15) / 16;
  const int chunks = (n_qt + WAVES * QT - 1) / (WAVES * QT);
  const int pairs8 = (a.n_seq * a.H + 7) / 8;
  hipLaunchKernelGGL((attn_flash2_kernel<WAVES, QT, NS, KB, WPS, QL2>), dim3((unsigned)(pairs8 * 8 * chunks)), dim3(WAVES * 64), smem_bytes, s, a, chunks);
  return hipGetLastError();
}

}  // namespace

static int g_flash2_cfg = -1;  // -1: not read yet; 0: default; 1: the first flash kernel; WAVES * 10 + QT pins a geometry
int flash2_cfg() {
  if (g_flash2_cfg < 0) {  // TAPCLIP_FLASH2_CFG (A/B, tools/attn_bench)
    const char* e = getenv("TAPCLIP_FLASH2_CFG");
    g_flash2_cfg = e ? atoi(e) : 0;
  }
  return g_flash2_cfg;
}
void flash2_set_cfg(int cfg) { g_flash2_cfg = cfg; }  // tools/attn_bench: switch geometries inside one process

hipError_t launch_flash2(const AttnArgs& a, hipStream_t s) {
  const int cfg = flash2_cfg();
  if (a.q_log2) {  // (the geometries kept for A/B are built for plain q only)
    return launch_flash2_cfg<4, 2, 3, 4, 3, true>(a, s);  // 250 us
  }
  switch (cfg) {
    // (measured at ViT-L/14@336, batch 128, same box, interleaved, every element of every output compared with the first
    //  kernel's: profiles/r05_attn_bench_*.log)
    case 122: return launch_flash2_cfg<12, 2, 6, 4, 3>(a, s);  // one 12-wave workgroup per CU, 2 chunks of 384 queries: 365-390 us
    case 822: return launch_flash2_cfg<8, 2, 6, 2, 4>(a, s);   // 8 waves, 32-key blocks (16 fewer score registers: 128 VGPRs), 3 chunks: 315 us
    // (6 waves x 2 tiles, 3 - 5 stages: 490 - 550 us -- 2 + 2 + 1 + 1 waves per SIMD twice do not fit three per SIMD, so ONE workgroup
    //  per CU; profiles/r05_attn_bench_six_waves.log.  5 waves x 2 tiles, 4 chunks of 160 queries, 3 - 5 stages: the same time as
    //  this geometry to 0.1 % at 257 / 577 / 1 025 tokens)
    // (4 waves, 32-key blocks, four workgroups per CU -- <4, 2, 4, 2, 4> -- : 270 us with 5 spilled dwords; not kept)
    default: return launch_flash2_cfg<4, 2, 3, 4, 3>(a, s);    // 4 waves x 2 tiles (128 queries), 64-key blocks, three workgroups per CU: 258 us
  }
}


}  // namespace tapclip
TAPCLIP_TU_NO_PK_F32_END
