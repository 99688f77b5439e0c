// Multi-head self-attention core for gfx950 (head dim 64, T <= 256 keys): softmax(Q K^T) V per
// (sequence, head), with optional write-back of the softmax probabilities.
//
// Replaces the scaled-dot-product step inside nn.MultiheadAttention that open_clip's
// ResidualAttentionBlock runs, and produces the per-head attention map
// that the reference's forward hook is documented to capture (reference models/clip_wrapper.py:29-40;
// K4').  1/sqrt(64) is folded into Wq/bq at weight-pack time.
//
// One 512-thread workgroup (8 waves; two of them fit a CU's LDS, i.e. 4 waves per SIMD) per (sequence, head); the head's whole K and V (<= 256 x 64 bf16) sit in
// LDS; each wave owns 16-query tiles.  All products are issued "transposed" so softmax is lane-local:
//   S^T[key, q] = K . Q^T   (A = K rows from LDS by ds_read_b128, B = Q rows straight from global)
//     -> lane (r = lane & 15, g = lane >> 4) holds, for query r, keys 16*kt + 4*g + e (e = 0..3) of
//        every key tile kt: a query's row is spread over only 4 lanes (xor 16, xor 32 reductions).
//   O^T[d, q] = V^T . P^T   (A = V^T via ds_read_b64_tr_b16 transposed LDS reads, B = P^T = the S^T
//        registers themselves, packed to bf16: element jj of k-step s2 is key 16*(2*s2 + (jj>>2)) + 4*g
//        + (jj&3); the V^T fragment is read with the same key order)
//     Which d feeds which MFMA row is only the transposed read's address: product dt takes the column quads
//     {16 pp + 4 dt .. + 3, pp = 0..3}, so lane (q, g) ends up with d = 16 g + 4 dt + e -- 16 CONSECUTIVE d
//     of its query over the four products: 32-byte bf16 / 16-byte MXFP8 stores, 128 / 64 B per row.
// bf16x3 (SPLIT): K, V, Q and P are hi/lo pairs and every product is three MFMAs.
#include <cstdlib>

#include "common.h"
#include "kernels.h"
#ifndef TAPCLIP_TU_NO_NANS
#error "attention.hip is built with -fno-honor-nans -DTAPCLIP_TU_NO_NANS (csrc/Makefile): its maxima are plain fmaxf"
#endif
#include "attn_store.h"

namespace tapclip {
namespace {

typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;

__device__ __forceinline__ bf16x8_t tr_pair(const uint8_t* base0, const uint8_t* base1) {
  // two transposed 4-key x 16-d blocks -> one 8-element (k = 8 keys) MFMA fragment
  s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(base0));
  s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(base1));
  s16x8_t v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

// ---- softmax arithmetic of the three kernels below.  The score loop is VALU-issue bound (a flash step of 16 scores per lane was
// 73 plain VALU ops + 17 v_exp_f32 beside 16 MFMAs), and a third of those ops were avoidable:
//  * fmaxf() has IEEE maxNum semantics, so hipcc canonicalises every MFMA output in front of it (v_max_f32 x, x, x: one extra op
//    per score) -- unless it may assume there are no NaNs: this file is compiled with -fno-honor-nans (Makefile; the operands are
//    finite, masked scores are -inf), and the nested fmaxf below become v_max3_f32, two scores per instruction.  (Rounds 3-5 had an
//    asm statement of raw v_max3_f32 here.  hipcc does not pad an asm statement for the MFMA -> VALU read hazard -- gfx950 has no
//    interlock there -- and in attention_long.hip such a statement read scores four instructions after their MFMA:
//    profiles/r05_flash2_asm_hazard.txt.  No asm statement of this library reads an MFMA result now.)
//  * exp argument and row sum two scores at a time: v_pk_fma_f32 / v_pk_add_f32 issue in the time of the scalar op.  Plain
//    operand order only (the [0,1] op_sel encodings are the ones tests/test_abi.py rejects).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float fold_max(float m, const f32x4_t& v) { return fmaxf(fmaxf(fmaxf(fmaxf(m, v[0]), v[1]), v[2]), v[3]); }
// sum2[0] + sum2[1] as ONE scalar add (left to itself hipcc forms v_pk_add_f32 vX, vX, vX op_sel:[0,1], the pair swap)
__device__ __forceinline__ float pair_sum(f32x2_t v) {
  float a = v[0];
  asm("" : "+v"(a));
  return a + v[1];
}
// v <- exp2(v * log2e + nmx) element-wise, sum2 += (v[0] + v[2], v[1] + v[3])
#ifndef TAPCLIP_LEAN_PACKED
#define TAPCLIP_LEAN_PACKED 0
#endif
template <bool PACKED = true>  // (false: the 80-VGPR instantiation -- aligned register pairs for every score cost it 3 more spilled dwords)
__device__ __forceinline__ void exp_sum4(f32x4_t& v, float nmx, f32x2_t& sum2) {
  if constexpr (!PACKED) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = __builtin_amdgcn_exp2f(fmaf(v[e], 1.44269504088896340736f, nmx));
      sum2[0] += v[e];
    }
    return;
  }
  const f32x2_t l2 = {1.44269504088896340736f, 1.44269504088896340736f}, n2 = {nmx, nmx};
  const f32x2_t a0 = __builtin_elementwise_fma(f32x2_t{v[0], v[1]}, l2, n2);
  const f32x2_t a1 = __builtin_elementwise_fma(f32x2_t{v[2], v[3]}, l2, n2);
  const f32x2_t p0 = {__builtin_amdgcn_exp2f(a0[0]), __builtin_amdgcn_exp2f(a0[1])};
  const f32x2_t p1 = {__builtin_amdgcn_exp2f(a1[0]), __builtin_amdgcn_exp2f(a1[1])};
  sum2 += p0;
  sum2 += p1;
  v = f32x4_t{p0[0], p0[1], p1[0], p1[1]};
}

template <int NKT, bool SPLIT, bool TIED = false>  // TIED: the last key counts exp(last_key_bias) times (tied.hip) -- its own
__global__ __launch_bounds__(512, (NKT % 2) ? 6 : 1) void attn_kernel(AttnArgs a) {  // instantiation: the 80-VGPR build of the image tower must not carry it
  constexpr int KEYS = NKT * 16;
  constexpr int TILE = KEYS * 128;  // bytes of one [KEYS][64] bf16 image
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* Kh = smem;                  // swizzled rows (b128 reads)
  uint8_t* Vh = smem + TILE;           // rows with swizzled 32-B blocks (transposed reads)
  uint8_t* Kl = smem + 2 * TILE;       // SPLIT only
  uint8_t* Vl = smem + 3 * TILE;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int T = a.T, D = a.D;
  const int seq = blockIdx.x / a.H, head = blockIdx.x - seq * a.H;
  const int64_t row0 = (int64_t)seq * T;
  const int64_t ld = 3 * (int64_t)D;
  const int qcol = head * 64, kcol = D + head * 64, vcol = 2 * D + head * 64;

#ifdef ATTN_STAMP
  // diagnostic build: 10-ns stamps of this workgroup's phases + where it ran (never in the library build)
  auto stamp = [&](int slot) {
    if (a.stamps != nullptr && tid == 0) a.stamps[(size_t)blockIdx.x * 8 + slot] = __builtin_amdgcn_s_memrealtime();
  };
  if (a.stamps != nullptr && tid == 0) {
    unsigned hw_id, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    a.stamps[(size_t)blockIdx.x * 8 + 7] = ((unsigned long long)xcc << 32) | hw_id;
  }
  stamp(0);
#endif
  // ---- Q fragments of every q-tile this wave owns, issued BEFORE the K/V staging so their latency
  // overlaps it.  B[k = d = 32*s + 8*g + j][col = q]; queries past T are clamped (never stored).
  constexpr int QT_MAX = (NKT + 7) / 8;  // 16-query tiles per wave (8 waves)
  const int n_qt = (T + 15) >> 4;
  constexpr bool LEAN = (NKT % 2) != 0;  // 80-VGPR build (three workgroups per CU): Q fragments loaded per tile
  bf16x8_t qh[LEAN ? 1 : QT_MAX][2], ql[SPLIT ? QT_MAX : 1][2];
  auto load_q = [&](int t, int slot) {
    int qc = (wave + 8 * t) * 16 + r;
    if (qc >= T) qc = T - 1;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int64_t off = (row0 + qc) * ld + qcol + 32 * s + 8 * g;
      qh[slot][s] = *reinterpret_cast<const bf16x8_t*>(a.qkv_hi + off);
      if (SPLIT) ql[slot][s] = *reinterpret_cast<const bf16x8_t*>(a.qkv_lo + off);
    }
  };
  if (!LEAN) {
#pragma unroll
    for (int t = 0; t < QT_MAX; ++t) load_q(t, t);
  }

  // ---- stage K (swizzled) and V (plain) of this head into LDS; rows >= T are zero.  The trip count
  // is a compile-time constant so all the global loads are in flight together.
  constexpr int N_IT = (KEYS * 8 + 511) / 512;
  {
    uint4 kv[N_IT], vv[N_IT], kvl[SPLIT ? N_IT : 1], vvl[SPLIT ? N_IT : 1];
#pragma unroll
    for (int it = 0; it < N_IT; ++it) {
      const int c = tid + 512 * it;
      const int key = c >> 3, kc = c & 7;
      kv[it] = make_uint4(0, 0, 0, 0);
      vv[it] = kv[it];
      if (SPLIT) {
        kvl[it] = kv[it];
        vvl[it] = kv[it];
      }
      if (key < T) {
        const int64_t base = (row0 + key) * ld + kc * 8;
        kv[it] = *reinterpret_cast<const uint4*>(a.qkv_hi + base + kcol);
        vv[it] = *reinterpret_cast<const uint4*>(a.qkv_hi + base + vcol);
        if (SPLIT) {
          kvl[it] = *reinterpret_cast<const uint4*>(a.qkv_lo + base + kcol);
          vvl[it] = *reinterpret_cast<const uint4*>(a.qkv_lo + base + vcol);
        }
      }
    }
#pragma unroll
    for (int it = 0; it < N_IT; ++it) {
      const int c = tid + 512 * it;
      const int key = c >> 3, kc = c & 7;
      if (key >= KEYS) continue;  // KEYS * 8 need not be a multiple of 512
      const int ko = key * 128 + ((kc ^ (key & 7)) << 4);
      // V image: 8-byte quad u of a 32-B block sits at quad u ^ ((key >> 1) & 3): the transposed reads below
      // (one quad index, all four blocks, 8 keys per 32-lane group) are then conflict-free
      const int vs = (key >> 1) & 3;
      const int vo = key * 128 + ((kc >> 1) << 5) + (((kc & 1) ^ (vs >> 1)) << 4);
      *reinterpret_cast<uint4*>(Kh + ko) = kv[it];
      *reinterpret_cast<uint4*>(Vh + vo) = (vs & 1) ? make_uint4(vv[it].z, vv[it].w, vv[it].x, vv[it].y) : vv[it];
      if (SPLIT) {
        *reinterpret_cast<uint4*>(Kl + ko) = kvl[it];
        *reinterpret_cast<uint4*>(Vl + vo) = (vs & 1) ? make_uint4(vvl[it].z, vvl[it].w, vvl[it].x, vvl[it].y) : vvl[it];
      }
    }
  }
#ifdef ATTN_STAMP
  stamp(1);  // thread 0's own loads have landed and its LDS writes are issued
#endif
  __syncthreads();
#ifdef ATTN_STAMP
  stamp(2);  // K/V of the head staged
#endif

#pragma unroll
  for (int t = 0; t < QT_MAX; ++t) {
    const int qt = wave + 8 * t;
    if (qt >= n_qt) break;       // wave-uniform
    const int qi = qt * 16 + r;  // this lane's query (column of S^T)
    if (LEAN) load_q(t, 0);
    const int qs = LEAN ? 0 : t;

    // ---- S^T = K . Q^T
    f32x4_t sc[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      sc[kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      const int key = kt * 16 + r;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int off = key * 128 + (((4 * s + g) ^ (key & 7)) << 4);
        const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(Kh + off);
        sc[kt] = TAPCLIP_MFMA_16x16x32(kf, qh[qs][s], sc[kt]);
        if (SPLIT) {
          const bf16x8_t kfl = *reinterpret_cast<const bf16x8_t*>(Kl + off);
          sc[kt] = TAPCLIP_MFMA_16x16x32(kfl, qh[qs][s], sc[kt]);
          sc[kt] = TAPCLIP_MFMA_16x16x32(kf, ql[t][s], sc[kt]);
        }
      }
    }

    // ---- mask + softmax over keys (lane-local + 2 shuffles).  Only key tiles that reach past T (or,
    // causal, past the diagonal) need the mask.  P stays un-normalised (largest element 1); 1/sum is
    // applied to the 16 O values, and to P only where it is written back.
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if ((kt * 16 + 15 >= T) || a.causal) {  // wave-uniform
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int key = kt * 16 + 4 * g + e;
          if (key >= T || (a.causal && key > qi)) sc[kt][e] = -INFINITY;
        }
      }
      // tied padding (tied.hip): the last key stands for m identical rows -- exp(s + ln m) = m exp(s)
      if (TIED && kt == ((T - 1) >> 4)) {  // kernel-uniform
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (4 * g + e == ((T - 1) & 15)) sc[kt][e] += a.last_key_bias;
      }
      mx = fold_max(mx, sc[kt]);
    }
    mx = rows_max(mx);
    const float LOG2E = 1.44269504088896340736f;
    const float nmx = -mx * LOG2E;
    f32x2_t sum2 = {0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) exp_sum4<!LEAN || TAPCLIP_LEAN_PACKED>(sc[kt], nmx, sum2);  // exp(s - max); 0 when masked
    float sum = pair_sum(sum2);
    sum = rows_sum(sum);
    const float inv = 1.0f / sum;

    // ---- optional probability write-back: probs[seq, head, q, key]
    if (a.probs != nullptr && qi < T) {
      float* prow = a.probs + (((int64_t)seq * a.H + head) * T + qi) * T;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int key = kt * 16 + 4 * g + e;
          if (key < T) prow[key] = sc[kt][e] * inv;
        }
    }

    // ---- O^T = V^T . P^T
    f32x4_t oc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oc[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // transposed-read lane address: in-group index i = 4*qq + pp supplies row qq, columns 4*pp..4*pp+3
    const int qq = r >> 2, pp = r & 3;
#pragma unroll
    for (int s2 = 0; s2 < (NKT + 1) / 2; ++s2) {
      // (odd NKT: the last step has one key tile; its upper half multiplies zeros into a re-read of the same keys)
      const bool half_step = (NKT % 2) && s2 == NKT / 2;
      bf16x8_t ph, pl;
      {
        bf16_t h[8], l[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          const float p = (half_step && jj >= 4) ? 0.f : sc[(2 * s2 + (jj >> 2)) < NKT ? 2 * s2 + (jj >> 2) : NKT - 1][jj & 3];
          if (SPLIT) split_bf(p, h[jj], l[jj]);
          else h[jj] = f2bf(p);
        }
        s16x8_t hv = {(short)h[0], (short)h[1], (short)h[2], (short)h[3], (short)h[4], (short)h[5], (short)h[6], (short)h[7]};
        ph = __builtin_bit_cast(bf16x8_t, hv);
        if (SPLIT) {
          s16x8_t lv = {(short)l[0], (short)l[1], (short)l[2], (short)l[3], (short)l[4], (short)l[5], (short)l[6], (short)l[7]};
          pl = __builtin_bit_cast(bf16x8_t, lv);
        }
      }
      const int key0 = 16 * (2 * s2) + 4 * g + qq;
      const int key1 = half_step ? key0 : 16 * (2 * s2 + 1) + 4 * g + qq;
      const int sw = (key0 >> 1) & 3;  // == (key1 >> 1) & 3: the keys differ by 16
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const int coff = (pp << 5) + ((dt ^ sw) << 3);  // block pp, swizzled quad dt: columns 16 pp + 4 dt .. + 3
        const bf16x8_t vf = tr_pair(Vh + key0 * 128 + coff, Vh + key1 * 128 + coff);
        oc[dt] = TAPCLIP_MFMA_16x16x32(vf, ph, oc[dt]);
        if (SPLIT) {
          const bf16x8_t vfl = tr_pair(Vl + key0 * 128 + coff, Vl + key1 * 128 + coff);
          oc[dt] = TAPCLIP_MFMA_16x16x32(vfl, ph, oc[dt]);
          oc[dt] = TAPCLIP_MFMA_16x16x32(vf, pl, oc[dt]);
        }
      }
    }

    // ---- store: lane holds O[q = qi][d = 16*g + 4*dt + e]
    if (!SPLIT && a.out_q != nullptr) store_o_mx8(a, oc, inv, row0 + qi, head, g, qi < T);  // (kernel-uniform branch)
    else if (qi < T) store_o_bf16<SPLIT>(a, oc, inv, row0 + qi, head, g);
#ifdef ATTN_STAMP
    stamp(3 + t);  // wave 0: its q-tile t done (stores issued)
#endif
  }
#ifdef ATTN_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  stamp(5);  // wave 0's stores complete
  __syncthreads();
  stamp(6);  // all waves done
#endif
}

// ---- long sequences (T > 256, e.g. ViT-L/14@336: 577 tokens): the same transposed products, flash style.  Since round 5 this
// kernel serves the split-bf16 and the causal case only; 16-bit operands without a mask run attention_long.hip (launch_attention).
// Grid (sequence*head, query chunk); the workgroup walks the keys in blocks of KB*16, staging one K/V block
// at a time in LDS; each wave owns QT 16-query tiles and keeps their running max m, partial row sum l
// (lane-local: the rescale factor is row-uniform, so the 4 lanes of a query are reduced once at the end)
// and O^T accumulators, rescaled by exp(m_old - m_new) whenever the block maximum grows.
template <int KB, int QT, bool SPLIT>
__global__ __launch_bounds__(512) void attn_flash_kernel(AttnArgs a) {
  constexpr int KEYS = KB * 16;
  constexpr int TILE = KEYS * 128;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* Kh = smem;
  uint8_t* Vh = smem + TILE;
  uint8_t* Kl = smem + 2 * TILE;
  uint8_t* Vl = smem + 3 * TILE;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int T = a.T, D = a.D;
  const int seq = blockIdx.x / a.H, head = blockIdx.x - seq * a.H;
  const int64_t row0 = (int64_t)seq * T;
  const int64_t ld = 3 * (int64_t)D;
  const int qcol = head * 64, kcol = D + head * 64, vcol = 2 * D + head * 64;
  const int n_qt = (T + 15) >> 4;
  // query tile t of this wave: the chunk's 8 QT tiles are dealt to the waves ROUND-ROBIN (tile = chunk base + 8 t + wave), so a
  // ragged last chunk (T = 577: 37 tiles = 16 + 16 + 5) gives five waves ONE tile each and the workgroup lasts one tile time;
  // dealt in runs of QT (waves 0, 1 two tiles, wave 2 one, the rest none) it lasted two with 2.5 of 8 waves busy
  const int qt_base = blockIdx.y * 8 * QT + wave;
  auto tile_of = [&](int t) { return qt_base + 8 * t; };
  const float LOG2E = 1.44269504088896340736f;
#ifndef TAPCLIP_FLASH_ABL
#define TAPCLIP_FLASH_ABL 0  // timing-only ablations (tools/Makefile attn_bench_alt): 1 no softmax VALU, 2 no MFMAs, 4 no LDS fragment reads, 8 no re-staging / barriers
#endif
  constexpr int ABL = TAPCLIP_FLASH_ABL;
  auto mm = [](const bf16x8_t& x, const bf16x8_t& y, const f32x4_t& c) {
    if constexpr ((ABL & 2) != 0) {
      asm volatile("" ::"v"(x), "v"(y));
      return c;
    } else {
      return TAPCLIP_MFMA_16x16x32(x, y, c);
    }
  };

  bf16x8_t qh[QT][2], ql[SPLIT ? QT : 1][2];
  float m[QT], l[QT];
  f32x4_t oc[QT][4];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    int qc = tile_of(t) * 16 + r;
    if (qc >= T) qc = T - 1;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int64_t off = (row0 + qc) * ld + qcol + 32 * s + 8 * g;
      qh[t][s] = *reinterpret_cast<const bf16x8_t*>(a.qkv_hi + off);
      if (SPLIT) ql[t][s] = *reinterpret_cast<const bf16x8_t*>(a.qkv_lo + off);
    }
    m[t] = -INFINITY;
    l[t] = 0.f;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oc[t][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }

  const int n_kb = (T + KEYS - 1) / KEYS;
  const int qq = r >> 2, pp = r & 3;
  // K/V rows of a key block: global -> registers -> LDS.  Without the hi/lo split the registers of block kb + 1 are
  // loaded BEFORE block kb is computed (the workgroup is alone on its CU at this register count, so nothing else
  // would hide that latency) and written to LDS after it.
  constexpr int N_IT = (KEYS * 8 + 511) / 512;  // (KEYS * 8 need not be a multiple of 512: guarded below)
  constexpr bool PREFETCH = !SPLIT;
  uint4 kvr[N_IT], vvr[N_IT], kvlr[SPLIT ? N_IT : 1], vvlr[SPLIT ? N_IT : 1];
  auto fetch_block = [&](int kb) {
#pragma unroll
    for (int it = 0; it < N_IT; ++it) {
      const int c = tid + 512 * it;
      const int kk = c >> 3, kc = c & 7;
      const int key = kb * KEYS + kk;
      kvr[it] = make_uint4(0, 0, 0, 0);
      vvr[it] = kvr[it];
      if (SPLIT) {
        kvlr[it] = kvr[it];
        vvlr[it] = kvr[it];
      }
      if (key < T && kk < KEYS) {
        const int64_t base = (row0 + key) * ld + kc * 8;
        kvr[it] = *reinterpret_cast<const uint4*>(a.qkv_hi + base + kcol);
        vvr[it] = *reinterpret_cast<const uint4*>(a.qkv_hi + base + vcol);
        if (SPLIT) {
          kvlr[it] = *reinterpret_cast<const uint4*>(a.qkv_lo + base + kcol);
          vvlr[it] = *reinterpret_cast<const uint4*>(a.qkv_lo + base + vcol);
        }
      }
    }
  };
  if (PREFETCH) fetch_block(0);
  for (int kb = 0; kb < n_kb; ++kb) {
    const int key_base = kb * KEYS;
    if (!(ABL & 8) || kb == 0) __syncthreads();  // every wave is done reading the previous block
    if (!PREFETCH) fetch_block(kb);
#pragma unroll
    for (int it = 0; it < N_IT; ++it) {
      const int c = tid + 512 * it;
      const int kk = c >> 3, kc = c & 7;
      if (kk >= KEYS || ((ABL & 8) && kb > 0)) continue;
      const int ko = kk * 128 + ((kc ^ (kk & 7)) << 4);
      const int vs = (kk >> 1) & 3;  // quad swizzle of the V image, as in attn_kernel
      const int vo = kk * 128 + ((kc >> 1) << 5) + (((kc & 1) ^ (vs >> 1)) << 4);
      *reinterpret_cast<uint4*>(Kh + ko) = kvr[it];
      *reinterpret_cast<uint4*>(Vh + vo) = (vs & 1) ? make_uint4(vvr[it].z, vvr[it].w, vvr[it].x, vvr[it].y) : vvr[it];
      if (SPLIT) {
        *reinterpret_cast<uint4*>(Kl + ko) = kvlr[it];
        *reinterpret_cast<uint4*>(Vl + vo) = (vs & 1) ? make_uint4(vvlr[it].z, vvlr[it].w, vvlr[it].x, vvlr[it].y) : vvlr[it];
      }
    }
    if (!(ABL & 8) || kb == 0) __syncthreads();
    if (PREFETCH && kb + 1 < n_kb && !(ABL & 8)) fetch_block(kb + 1);  // in flight during the products below

#pragma unroll
    for (int t = 0; t < QT; ++t) {
      if (tile_of(t) >= n_qt) break;  // wave-uniform
      const int qi = tile_of(t) * 16 + r;
      f32x4_t sc[KB];
#pragma unroll
      for (int kt = 0; kt < KB; ++kt) {
        sc[kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const int kk = kt * 16 + r;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int off = kk * 128 + (((4 * s + g) ^ (kk & 7)) << 4);
          const bf16x8_t kf = (ABL & 4) ? qh[t][s] : *reinterpret_cast<const bf16x8_t*>(Kh + off);
          sc[kt] = mm(kf, qh[t][s], sc[kt]);
          if (SPLIT) {
            const bf16x8_t kfl = *reinterpret_cast<const bf16x8_t*>(Kl + off);
            sc[kt] = mm(kfl, qh[t][s], sc[kt]);
            sc[kt] = mm(kf, ql[t][s], sc[kt]);
          }
        }
      }
      if constexpr ((ABL & 1) == 0) {
      float bm = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < KB; ++kt) {
        if ((key_base + kt * 16 + 15 >= T) || a.causal) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int key = key_base + kt * 16 + 4 * g + e;
            if (key >= T || (a.causal && key > qi)) sc[kt][e] = -INFINITY;
          }
        }
        bm = fold_max(bm, sc[kt]);
      }
      bm = rows_max(bm);
      const float m_new = fmaxf(m[t], bm);
      // m_new == -inf only while every key so far is masked (then p = 0 and nothing is accumulated)
      const float alpha = (m_new == -INFINITY) ? 1.0f : __builtin_amdgcn_exp2f((m[t] - m_new) * LOG2E);
      const float nmx = (m_new == -INFINITY) ? 0.0f : -m_new * LOG2E;
      f32x2_t bs2 = {0.f, 0.f};
#pragma unroll
      for (int kt = 0; kt < KB; ++kt) {
        exp_sum4(sc[kt], nmx, bs2);
      }
      l[t] = fmaf(l[t], alpha, pair_sum(bs2));
      m[t] = m_new;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int e = 0; e < 4; ++e) oc[t][dt][e] *= alpha;
      }

#pragma unroll
      for (int s2 = 0; s2 < KB / 2; ++s2) {
        bf16x8_t ph, pl;
        {
          bf16_t h[8], lo8[8];
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) {
            const float p = sc[2 * s2 + (jj >> 2)][jj & 3];
            if (SPLIT) split_bf(p, h[jj], lo8[jj]);
            else h[jj] = f2bf(p);
          }
          s16x8_t hv = {(short)h[0], (short)h[1], (short)h[2], (short)h[3], (short)h[4], (short)h[5], (short)h[6], (short)h[7]};
          ph = __builtin_bit_cast(bf16x8_t, hv);
          if (SPLIT) {
            s16x8_t lv = {(short)lo8[0], (short)lo8[1], (short)lo8[2], (short)lo8[3], (short)lo8[4], (short)lo8[5], (short)lo8[6], (short)lo8[7]};
            pl = __builtin_bit_cast(bf16x8_t, lv);
          }
        }
        const int key0 = 16 * (2 * s2) + 4 * g + qq;
        const int key1 = 16 * (2 * s2 + 1) + 4 * g + qq;
        const int sw = (key0 >> 1) & 3;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const int coff = (pp << 5) + ((dt ^ sw) << 3);
          const bf16x8_t vf = (ABL & 4) ? ph : tr_pair(Vh + key0 * 128 + coff, Vh + key1 * 128 + coff);
          oc[t][dt] = mm(vf, ph, oc[t][dt]);
          if (SPLIT) {
            const bf16x8_t vfl = tr_pair(Vl + key0 * 128 + coff, Vl + key1 * 128 + coff);
            oc[t][dt] = mm(vfl, ph, oc[t][dt]);
            oc[t][dt] = mm(vf, pl, oc[t][dt]);
          }
        }
      }
    }
  }

#pragma unroll
  for (int t = 0; t < QT; ++t) {
    if (tile_of(t) >= n_qt) break;
    const int qi = tile_of(t) * 16 + r;
    float sum = l[t];
    sum = rows_sum(sum);
    const float inv = 1.0f / sum;
    if (!SPLIT && a.out_q != nullptr) store_o_mx8(a, oc[t], inv, row0 + qi, head, g, qi < T);
    else if (qi < T) store_o_bf16<SPLIT>(a, oc[t], inv, row0 + qi, head, g);
  }
}

template <bool SPLIT, int QT, int KB>
hipError_t launch_flash_q(const AttnArgs& a, hipStream_t s) {
  static bool attr_set = false;
  const int smem_bytes = KB * 16 * 128 * (SPLIT ? 4 : 2);
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_flash_kernel<KB, QT, SPLIT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int n_qt = (a.T + 15) / 16;
  const int chunks = (n_qt + 8 * QT - 1) / (8 * QT);
  hipLaunchKernelGGL((attn_flash_kernel<KB, QT, SPLIT>), dim3((unsigned)(a.n_seq * a.H), (unsigned)chunks), dim3(512),
                     smem_bytes, s, a);
  return hipGetLastError();
}

template <bool SPLIT>
hipError_t launch_flash(const AttnArgs& a, hipStream_t s) {
  // Query tiles per wave (QT) and key tiles per block (KB) decide the register count, and that decides whether one
  // or two workgroups share a CU.  bf16 / fp16: QT = 2, KB = 4 -> 114 VGPRs, two workgroups per CU (4 waves per
  // SIMD): 161 us at n = 64, T = 577, against 209 us for QT = 3, KB = 8 (224 VGPRs, one workgroup per CU, fewer
  // K/V re-stagings).  bf16x3 keeps QT = 2 / 3 at KB = 8 (its hi + lo fragments do not fit the smaller budget).
  // TAPCLIP_FLASH_QT / TAPCLIP_FLASH_KB pin the choice (tools/gemm_bench).
  static const int forced_qt = [] {
    const char* e = getenv("TAPCLIP_FLASH_QT");
    return e ? atoi(e) : 0;
  }();
  static const int forced_kb = [] {
    const char* e = getenv("TAPCLIP_FLASH_KB");
    return e ? atoi(e) : 0;
  }();
  const int n_qt = (a.T + 15) / 16;
  if (SPLIT) {
    const int qt = forced_qt ? forced_qt : (n_qt <= 16 ? 2 : 3);
    if (qt == 2) return launch_flash_q<SPLIT, 2, 8>(a, s);
    return launch_flash_q<SPLIT, 3, 8>(a, s);
  }
  const int qt = forced_qt ? forced_qt : 2;
  const int kb = forced_kb ? forced_kb : 4;
  if (qt == 3) return kb == 4 ? launch_flash_q<SPLIT, 3, 4>(a, s) : kb == 6 ? launch_flash_q<SPLIT, 3, 6>(a, s) : launch_flash_q<SPLIT, 3, 8>(a, s);
  if (qt == 1) return kb == 4 ? launch_flash_q<SPLIT, 1, 4>(a, s) : kb == 6 ? launch_flash_q<SPLIT, 1, 6>(a, s) : launch_flash_q<SPLIT, 1, 8>(a, s);
  return kb == 4 ? launch_flash_q<SPLIT, 2, 4>(a, s) : kb == 6 ? launch_flash_q<SPLIT, 2, 6>(a, s) : launch_flash_q<SPLIT, 2, 8>(a, s);
}

template <int NKT, bool SPLIT, bool TIED>
hipError_t launch_tt(const AttnArgs& a, hipStream_t s) {
  static bool attr_set = false;
  const int smem_bytes = NKT * 16 * 128 * (SPLIT ? 4 : 2);
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_kernel<NKT, SPLIT, TIED>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL((attn_kernel<NKT, SPLIT, TIED>), dim3((unsigned)(a.n_seq * a.H)), dim3(512), smem_bytes, s, a);
  return hipGetLastError();
}
template <int NKT, bool SPLIT>
hipError_t launch_t(const AttnArgs& a, hipStream_t s) {
  if constexpr (NKT % 2 == 0) {
    if (a.last_key_bias != 0.f) return launch_tt<NKT, SPLIT, true>(a, s);
  }
  return launch_tt<NKT, SPLIT, false>(a, s);
}

template <bool SPLIT>
hipError_t dispatch(const AttnArgs& a, hipStream_t s) {
  // 197 tokens (ViT-B/16): 13 key tiles make the K and V images 2 x 26 KiB, so three workgroups fit a CU's LDS, and
  // that instantiation is compiled for 6 waves per SIMD (80 VGPRs)
  static const bool no13 = getenv("TAPCLIP_ATTN_NO13") != nullptr;
  if (!SPLIT && !no13 && a.last_key_bias == 0.f && a.T > 192 && a.T <= 208) return launch_t<13, SPLIT>(a, s);
  const int nkt = ((a.T + 31) / 32) * 2;  // even number of 16-key tiles
  switch (nkt) {
    case 2: return launch_t<2, SPLIT>(a, s);
    case 4: return launch_t<4, SPLIT>(a, s);
    case 6: return launch_t<6, SPLIT>(a, s);
    case 8: return launch_t<8, SPLIT>(a, s);
    case 10: return launch_t<10, SPLIT>(a, s);
    case 12: return launch_t<12, SPLIT>(a, s);
    case 14: return launch_t<14, SPLIT>(a, s);
    case 16: return launch_t<16, SPLIT>(a, s);
    default: return hipErrorInvalidValue;
  }
}

__global__ void head_mean_kernel(const float* __restrict__ probs, int H, int64_t tt, int64_t total, float* out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int64_t seq = i / tt, e = i - seq * tt;
  const float* p = probs + seq * H * tt + e;
  float s = 0.f;
  for (int h = 0; h < H; ++h) s += p[h * tt];  // same summation order as torch .mean(dim=1) for small H
  out[i] = s / (float)H;
}

// ---- attention for ONE query row per sequence (the pooled token of a tower's LAST block).
// The image tower's output is the CLS row only (open_clip pools token 0 before ln_post / proj; reference call site
// models/clip_wrapper.py:46-47), so in the last block the other 196 rows' queries, attention outputs, out_proj and
// MLP are dead work: K and V are still needed for every token, Q and everything behind the attention core only for the
// pooled row (tower.hip run_last_block_pooled).  One wave per (sequence, head): lane l = 8 g + c works on dims
// 8 c .. 8 c + 7 of token t0 + g, so a token's 128-B K / V row is ONE coalesced 8-lane access; the 8-lane dot
// products are finished by three xor-shuffles, scores and probabilities live in LDS (T floats), softmax and P.V in
// fp32 (no rounding of P: this path is more accurate than the full kernel, not bit-identical to it).
// HBM-bound: it reads K and V of the layer once (2/3 of the q|k|v bytes the full kernel reads).
struct AttnPoolArgs {
  const bf16_t* q_hi;   // [n_seq, D] pooled queries (1/sqrt(64) folded into Wq)
  const bf16_t* q_lo;   // bf16x3 only
  const bf16_t* kv_hi;  // [n_seq * T, 3 D]: k at columns D + 64 h, v at 2 D + 64 h (the q columns are not read)
  const bf16_t* kv_lo;
  bf16_t* out_hi;       // [n_seq, D]
  bf16_t* out_lo;
  int32_t n_seq, T, H, D;
  int32_t q_log2;       // q carries log2(e) too (AttnArgs::q_log2): the scores are base-2 exponents
};

constexpr int POOL_MAX_T = 1024;

template <bool SPLIT>
__global__ __launch_bounds__(256) void attn_pool_kernel(AttnPoolArgs a) {
  __shared__ float sc[4][POOL_MAX_T];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int item = blockIdx.x * 4 + wave;  // (sequence, head)
  if (item >= a.n_seq * a.H) return;
  const int seq = item / a.H, h = item - seq * a.H;
  const int g = lane >> 3, c = lane & 7;
  const int64_t ld = 3 * (int64_t)a.D;
  auto load8 = [&](const bf16_t* hi, const bf16_t* lo, int64_t off, float (&v)[8]) {
    const uint4 u = *reinterpret_cast<const uint4*>(hi + off);
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[2 * j] = bf2f((bf16_t)(w[j] & 0xFFFF));
      v[2 * j + 1] = bf2f((bf16_t)(w[j] >> 16));
    }
    if (SPLIT) {
      const uint4 ul = *reinterpret_cast<const uint4*>(lo + off);
      const uint32_t wl[4] = {ul.x, ul.y, ul.z, ul.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[2 * j] += bf2f((bf16_t)(wl[j] & 0xFFFF));
        v[2 * j + 1] += bf2f((bf16_t)(wl[j] >> 16));
      }
    }
  };
  float q[8];
  load8(a.q_hi, a.q_lo, (int64_t)seq * a.D + 64 * h + 8 * c, q);
  const int64_t row0 = (int64_t)seq * a.T;
  const int64_t kcol = a.D + 64 * h + 8 * c, vcol = 2 * (int64_t)a.D + 64 * h + 8 * c;
  // ---- scores s_t = q . k_t, 8 tokens per iteration
  float mx = -3.0e38f;
#pragma unroll 4
  for (int t0 = 0; t0 < a.T; t0 += 8) {
    const int t = t0 + g;
    float k[8];
    float dot = 0.f;
    if (t < a.T) {
      load8(a.kv_hi, a.kv_lo, (row0 + t) * ld + kcol, k);
#pragma unroll
      for (int j = 0; j < 8; ++j) dot = fmaf(q[j], k[j], dot);
    }
    dot += __shfl_xor(dot, 1, 64);
    dot += __shfl_xor(dot, 2, 64);
    dot += __shfl_xor(dot, 4, 64);
    if (t < a.T) {
      if (c == 0) sc[wave][t] = dot;
      mx = fmaxf(mx, dot);
    }
  }
  mx = wave_max(mx);
  __builtin_amdgcn_wave_barrier();
  // ---- p_t = exp(s_t - max), sum (a wave's LDS accesses execute in order; the waves are independent)
  float sum = 0.f;
  for (int t = lane; t < a.T; t += 64) {
    const float p = a.q_log2 ? __builtin_amdgcn_exp2f(sc[wave][t] - mx) : __expf(sc[wave][t] - mx);
    sc[wave][t] = p;
    sum += p;
  }
  sum = wave_sum(sum);
  __builtin_amdgcn_wave_barrier();
  // ---- o = sum_t p_t v_t
  float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int t0 = 0; t0 < a.T; t0 += 8) {
    const int t = t0 + g;
    if (t < a.T) {
      float v[8];
      load8(a.kv_hi, a.kv_lo, (row0 + t) * ld + vcol, v);
      const float p = sc[wave][t];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = fmaf(p, v[j], o[j]);
    }
  }
  const float inv = 1.0f / sum;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    o[j] += __shfl_xor(o[j], 8, 64);
    o[j] += __shfl_xor(o[j], 16, 64);
    o[j] += __shfl_xor(o[j], 32, 64);
    o[j] *= inv;
  }
  if (g == 0) {
    const int64_t off = (int64_t)seq * a.D + 64 * h + 8 * c;
    bf16_t hi[8], lo[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (SPLIT) split_bf(o[j], hi[j], lo[j]);
      else hi[j] = f2bf(o[j]);
    }
    *reinterpret_cast<uint4*>(a.out_hi + off) = make_uint4((uint32_t)hi[0] | ((uint32_t)hi[1] << 16), (uint32_t)hi[2] | ((uint32_t)hi[3] << 16),
                                                           (uint32_t)hi[4] | ((uint32_t)hi[5] << 16), (uint32_t)hi[6] | ((uint32_t)hi[7] << 16));
    if (SPLIT)
      *reinterpret_cast<uint4*>(a.out_lo + off) = make_uint4((uint32_t)lo[0] | ((uint32_t)lo[1] << 16), (uint32_t)lo[2] | ((uint32_t)lo[3] << 16),
                                                             (uint32_t)lo[4] | ((uint32_t)lo[5] << 16), (uint32_t)lo[6] | ((uint32_t)lo[7] << 16));
  }
}

}  // namespace

hipError_t launch_attention_pooled(const bf16_t* q_hi, const bf16_t* q_lo, const bf16_t* qkv_hi, const bf16_t* qkv_lo, bf16_t* out_hi,
                                   bf16_t* out_lo, int32_t n_seq, int32_t T, int32_t H, int32_t D, bool split, hipStream_t s, bool q_log2) {
  if (T <= 0 || T > POOL_MAX_T || D != H * 64 || n_seq <= 0 || !q_hi || !qkv_hi || !out_hi) return hipErrorInvalidValue;
  if (split && (!q_lo || !qkv_lo || !out_lo)) return hipErrorInvalidValue;
  AttnPoolArgs a{q_hi, q_lo, qkv_hi, qkv_lo, out_hi, out_lo, n_seq, T, H, D, q_log2 ? 1 : 0};
  const unsigned grid = (unsigned)((n_seq * H + 3) / 4);
  if (split) hipLaunchKernelGGL(attn_pool_kernel<true>, dim3(grid), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(attn_pool_kernel<false>, dim3(grid), dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_attention(const AttnArgs& a, bool split, hipStream_t s) {
  if (a.T <= 0 || a.D != a.H * 64 || a.n_seq <= 0) return hipErrorInvalidValue;
  static const bool force_flash = getenv("TAPCLIP_ATTN_FORCE_FLASH") != nullptr;  // tools/gemm_bench: compare the two kernels
  if (a.last_key_bias != 0.f && (a.T > 256 || a.causal)) return hipErrorInvalidValue;  // (a tied key has no position: no mask; the flash kernel does not know it)
  if (a.T > 256 || (force_flash && a.probs == nullptr && a.last_key_bias == 0.f)) {  // whole-head-in-LDS kernel holds at most 256 keys: flash-style kernel (no probability write-back)
    if (a.probs != nullptr) return hipErrorInvalidValue;
    // 16-bit operands without a mask: the LDS-DMA kernel (round 5); split-bf16 and causal stay on the first flash kernel
    if (!split && !a.causal && flash2_cfg() != 1) return launch_flash2(a, s);  // attention_long.hip
    if (a.q_log2) return hipErrorInvalidValue;  // (only that kernel reads base-2 scores)
    return split ? launch_flash<true>(a, s) : launch_flash<false>(a, s);
  }
  if (a.q_log2) return hipErrorInvalidValue;
  return split ? dispatch<true>(a, s) : dispatch<false>(a, s);
}

hipError_t launch_head_mean(const float* probs, int32_t n, int32_t H, int32_t T, float* out, hipStream_t s) {
  const int64_t tt = (int64_t)T * T, total = (int64_t)n * tt;
  hipLaunchKernelGGL(head_mean_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, probs, H, tt, total, out);
  return hipGetLastError();
}

}  // namespace tapclip
TAPCLIP_TU_NO_PK_F32_END
