// Host side of the C ABI (include/tapclip.h): tower handles, weight packing, the per-layer launch
// sequence of the vision and text towers, and the small standalone ops.  Kernels: gemm.hip,
// layernorm.hip, attention.hip, elementwise.hip.
//
// Launch sequence of one pre-LN residual block (open_clip ResidualAttentionBlock, reached through
// reference models/clip_wrapper.py:47 and models/model_wrapper.py:58,72):
//   [x += previous c_proj branch] LN1 (fp32 x -> bf16)   layernorm.hip
//   QKV GEMM + bias -> bf16 q|k|v                       gemm*.hip EPI_BIAS_BF16 (1/sqrt(hd) folded into Wq, bq)
//   attention core (+ probs)                            attention.hip
//   out_proj GEMM + bias -> bf16 branch d               gemm*.hip EPI_BIAS_BF16
//   x += d, LN2                                         layernorm.hip (fused add + norm)
//   c_fc GEMM + bias + GELU -> bf16                     gemm*.hip EPI_BIAS_GELU_BF16
//   c_proj GEMM + bias -> bf16 branch d                 gemm*.hip EPI_BIAS_BF16 (added by the next LN1 / the tail)
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <set>
#include <string>
#include <vector>

#include "../../include/tapclip.h"
#include "kernels.h"

using namespace tapclip;

namespace {

thread_local char g_err[1024] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                       \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess)                                                                   \
      return fail(TAPCLIP_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

struct Packed {  // a [rows, ld] bf16 matrix (hi, and lo for bf16x3)
  bf16_t* hi = nullptr;
  bf16_t* lo = nullptr;
};

struct PackedMx8 {  // a [rows, K] MXFP8 matrix: e4m3 bytes + e8m0 scales [K/64][rows][2]
  uint8_t* q = nullptr;
  uint8_t* s = nullptr;
};

struct LayerW {
  float *ln1_g = nullptr, *ln1_b = nullptr, *ln2_g = nullptr, *ln2_b = nullptr;
  Packed wqkv, wo, wfc, wpr;
  PackedMx8 qqkv, qo, qfc, qpr;  // fp8 precision: the block GEMM weights as MXFP8 (instead of the bf16 copies)
  Packed wqkv_t, wo_t, wfc_t, wpr_t;  // text towers: transposed copies [K, N] for the dX GEMMs of the backward
  float *bqkv = nullptr, *bo = nullptr, *bfc = nullptr, *bpr = nullptr;
};

struct ProfRec {
  int slot;
  hipEvent_t start, stop;
};

size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

}  // namespace

struct tapclip_tower {
  tapclip_tower_cfg cfg;
  bool split = false;
  bool fp8 = false;       // TAPCLIP_PREC_FP8: block GEMMs on MXFP8 (image tower only)
  bool x24 = false;       // image tower, bf16 / IEEE-half modes: the residual stream of the blocks in 24-bit planes (layernorm.hip XF = 2)
  bool prune_last = true; // image tower: the last block computes K / V for every token but everything else for the CLS row only
  bool q_log2 = false;    // image tower of more than 256 tokens, 16-bit operands: log2(e) folded into Wq, bq beside 1/sqrt(64) (AttnArgs::q_log2)
  bool ksplit = true;     // K-split the tiles of partial GEMM rounds over idle CUs (TAPCLIP_FLAG_KSPLIT)
  int tokens_vision = 0;  // G*G + 1
  int Kp = 0;             // padded 3*p*p
  std::vector<LayerW> layers;
  std::vector<void*> allocs;
  std::set<std::string> required, loaded;
  // vision
  Packed conv;
  float *cls = nullptr, *pos = nullptr, *lnpre_g = nullptr, *lnpre_b = nullptr, *lnpost_g = nullptr,
        *lnpost_b = nullptr, *proj = nullptr;
  // text
  float *tok_emb = nullptr, *lnfin_g = nullptr, *lnfin_b = nullptr, *text_proj = nullptr;
  int* bad_token = nullptr;   // device flag of tapclip_embed_tokens: set when a token id is outside the table
  int* tied_flag = nullptr;   // device flag of the tied-padding entry points (tied.hip): raised when the claimed run of identical rows is not one
  float* split_ws = nullptr;  // scratch for the K-split tail tiles of gemm256.hip (64 MiB, handle-owned)
  // profiling
  bool prof_on = false;
  std::vector<ProfRec> prof;
  size_t prof_used = 0;
  double prof_ms[TAPCLIP_PROFILE_SLOTS] = {0};
  int64_t prof_n[TAPCLIP_PROFILE_SLOTS] = {0};
};

namespace {

struct Workspace {
  float* x = nullptr;
  bf16_t *xn_hi = nullptr, *xn_lo = nullptr;
  bf16_t *qkv_hi = nullptr, *qkv_lo = nullptr;
  bf16_t *ao_hi = nullptr, *ao_lo = nullptr;
  bf16_t *h_hi = nullptr, *h_lo = nullptr;
  bf16_t *d_hi = nullptr, *d_lo = nullptr;  // pending residual branch (c_proj output; out_proj's in the backward's recompute)
  bf16_t *a_hi = nullptr, *a_lo = nullptr;  // forward: out_proj's branch (folded into x by the NEXT block's LN1, with d)
  float* probs = nullptr;
  // fp8 precision: MXFP8 activations live in the front of the bf16 buffers they replace (e4m3 [M, K] then the
  // scales [K/64][m_pad][2]: 1.03 bytes per element against 2)
  uint8_t *xn_q = nullptr, *xn_s = nullptr, *ao_q = nullptr, *ao_s = nullptr, *h_q = nullptr, *h_s = nullptr;
  bf16_t* x16 = nullptr;  // fp8 precision: the residual stream of the blocks is 16-bit
  bf16_t* x24_hi = nullptr;  // x24: upper 16 bits of the residual stream
  uint8_t* x24_lo = nullptr;  //      next 8 bits
  int64_t m_pad = 0;
  size_t bytes = 0;
};

Workspace carve(const tapclip_tower* t, int64_t n_seq, int tokens, void* base) {
  Workspace w;
  const int64_t M = n_seq * tokens;
  const int64_t D = t->cfg.width, F = t->cfg.mlp_dim;
  // the im2col patch matrix aliases the MLP hidden buffer (never live together)
  int64_t hid_elems = M * F;
  if (t->cfg.kind == TAPCLIP_TOWER_VISION) {
    const int64_t pe = n_seq * (tokens - 1) * (int64_t)t->Kp;
    if (pe > hid_elems) hid_elems = pe;
  }
  size_t off = 0;
  auto take = [&](size_t bytes) {
    void* p = base ? static_cast<char*>(base) + off : nullptr;
    off += align_up(bytes);
    return p;
  };
  w.x = static_cast<float*>(take(M * D * 4));
  w.xn_hi = static_cast<bf16_t*>(take(M * D * 2));
  w.qkv_hi = static_cast<bf16_t*>(take(M * 3 * D * 2));
  w.ao_hi = static_cast<bf16_t*>(take(M * D * 2));
  w.h_hi = static_cast<bf16_t*>(take(hid_elems * 2));
  w.d_hi = static_cast<bf16_t*>(take(M * D * 2));
  w.a_hi = static_cast<bf16_t*>(take(M * D * 2));
  if (t->split) {
    w.d_lo = static_cast<bf16_t*>(take(M * D * 2));
    w.a_lo = static_cast<bf16_t*>(take(M * D * 2));
    w.xn_lo = static_cast<bf16_t*>(take(M * D * 2));
    w.qkv_lo = static_cast<bf16_t*>(take(M * 3 * D * 2));
    w.ao_lo = static_cast<bf16_t*>(take(M * D * 2));
    w.h_lo = static_cast<bf16_t*>(take(hid_elems * 2));
  }
  if (t->cfg.kind == TAPCLIP_TOWER_TEXT)
    w.probs = static_cast<float*>(take((size_t)n_seq * t->cfg.heads * tokens * tokens * 4));
  if (t->fp8) {
    w.m_pad = (M + 7) / 8 * 8;
    auto view = [&](bf16_t* buf, int64_t K, uint8_t*& q, uint8_t*& sc) {
      q = reinterpret_cast<uint8_t*>(buf);
      sc = buf ? q + align_up((size_t)M * K) : nullptr;
    };
    view(w.xn_hi, D, w.xn_q, w.xn_s);
    view(w.ao_hi, D, w.ao_q, w.ao_s);
    view(w.h_hi, F, w.h_q, w.h_s);
    w.x16 = static_cast<bf16_t*>(take(M * D * 2));
  }
  if (t->x24) {
    w.x24_hi = static_cast<bf16_t*>(take(M * D * 2));
    w.x24_lo = static_cast<uint8_t*>(take(M * D));
  }
  w.bytes = off;
  return w;
}

// roctx ranges per kernel family (tracing), off by default: TAPCLIP_ROCTX=1 resolves
// roctxRangePushA / roctxRangePop from libroctx64.so at first use, and every ProfScope -- one per kernel family launch
// sequence -- becomes a named range that `rocprofv3 --marker-trace --kernel-trace` shows around its kernels.
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    const char* e = getenv("TAPCLIP_ROCTX");
    if (!e || !*e || *e == '0') return;
    void* h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
    push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
    pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
    if (!push || !pop) push = nullptr, pop = nullptr;
  }
};
const Roctx& roctx() {
  static const Roctx r;
  return r;
}
const char* const kSlotNames[TAPCLIP_PROFILE_SLOTS] = {"tapclip:patch_embed", "tapclip:layernorm", "tapclip:gemm_qkv", "tapclip:attention",
                                                       "tapclip:gemm_out_proj", "tapclip:gemm_fc_gelu", "tapclip:gemm_proj", "tapclip:pool_proj",
                                                       "tapclip:pooled_tail"};

struct ProfScope {
  tapclip_tower* t;
  hipStream_t s;
  ProfRec* rec = nullptr;
  bool range = false;
  ProfScope(tapclip_tower* tw, int slot, hipStream_t st) : t(tw), s(st) {
    if (roctx().push && slot >= 0 && slot < TAPCLIP_PROFILE_SLOTS) range = roctx().push(kSlotNames[slot]) >= 0;
    if (!t->prof_on) return;
    if (t->prof_used == t->prof.size()) {
      ProfRec r;
      r.slot = slot;
      if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) return;
      t->prof.push_back(r);
    }
    rec = &t->prof[t->prof_used++];
    rec->slot = slot;
    (void)hipEventRecord(rec->start, s);
  }
  void cancel() {  // nothing was launched inside this scope after all: give the record back
    if (rec) {
      --t->prof_used;
      rec = nullptr;
    }
  }
  ~ProfScope() {
    if (rec) (void)hipEventRecord(rec->stop, s);
    if (range) (void)roctx().pop();
  }
};

int dev_alloc(tapclip_tower* t, size_t bytes, void** out) {
  void* p = nullptr;
  if (hipMalloc(&p, bytes ? bytes : 4) != hipSuccess) return fail(TAPCLIP_ENOMEM, "hipMalloc(%zu) failed", bytes);
  t->allocs.push_back(p);
  *out = p;
  return TAPCLIP_OK;
}

// copy an fp32 tensor into tower-owned memory (optionally scaling the first scale_n elements)
int own_f32(tapclip_tower* t, const float* src, int64_t n, int64_t scale_n, float scale, float** dst, hipStream_t s) {
  void* p;
  int rc = dev_alloc(t, n * 4, &p);
  if (rc) return rc;
  HIP_TRY(launch_scale_copy(src, n, scale_n, scale, static_cast<float*>(p), s));
  *dst = static_cast<float*>(p);
  return TAPCLIP_OK;
}

int own_packed(tapclip_tower* t, const float* src, int64_t rows, int cols, int dst_ld, int64_t scale_rows, float scale,
               Packed* out, hipStream_t s) {
  void *h, *l = nullptr;
  int rc = dev_alloc(t, rows * dst_ld * 2, &h);
  if (rc) return rc;
  if (t->split) {
    rc = dev_alloc(t, rows * dst_ld * 2, &l);
    if (rc) return rc;
  }
  HIP_TRY(launch_pack(src, rows, cols, cols, dst_ld, scale_rows, scale, static_cast<bf16_t*>(h),
                      static_cast<bf16_t*>(l), s));
  out->hi = static_cast<bf16_t*>(h);
  out->lo = static_cast<bf16_t*>(l);
  return TAPCLIP_OK;
}

int own_packed_mx8(tapclip_tower* t, const float* src, int64_t rows, int cols, int64_t scale_rows, float scale, PackedMx8* out,
                   hipStream_t s) {
  void *q, *sc;
  int rc = dev_alloc(t, rows * cols, &q);
  if (rc) return rc;
  if ((rc = dev_alloc(t, (size_t)(cols / 64) * rows * 2, &sc))) return rc;
  HIP_TRY(launch_quantize_mx8(src, rows, cols, cols, scale_rows, scale, static_cast<uint8_t*>(q), cols, static_cast<uint8_t*>(sc), rows, s));
  out->q = static_cast<uint8_t*>(q);
  out->s = static_cast<uint8_t*>(sc);
  return TAPCLIP_OK;
}

bool shape_is(const int64_t* shape, int ndim, std::initializer_list<int64_t> want) {
  if (ndim != (int)want.size()) return false;
  int i = 0;
  for (int64_t w : want)
    if (shape[i++] != w) return false;
  return true;
}

std::string shape_str(const int64_t* shape, int ndim) {
  std::string s = "[";
  for (int i = 0; i < ndim; ++i) s += (i ? "," : "") + std::to_string(shape[i]);
  return s + "]";
}

// m-tiles per group of the persistent GEMM's XCD-aware tile order, per kernel family (slot 2 QKV, 4 out_proj, 5 c_fc,
// 6 c_proj).  Measured in the tower at ViT-B/16, batch 256 (bench.py, same box): QKV 159 -> 154-157 us at 6; out_proj
// 69 us at 8 and 76 us at 7, 9 or 10; c_fc 251 -> 245 us at 2-4; c_proj 227 -> 218-221 us at 1-4: -1 % on the step
// against 8 everywhere (ViT-L/14@336, batch 128: QKV 416 -> 397, c_proj 556 -> 538 us, c_fc 630 us at 3, 6 and 8 but 720 us
// at 2: hence 3).  TAPCLIP_GM="q,o,f,p" overrides (experiments).
int group_m_for(int slot) {
  static int gm[4] = {6, 8, 3, 2};
  static const bool init = [] {
    const char* e = getenv("TAPCLIP_GM");
    if (e) sscanf(e, "%d,%d,%d,%d", &gm[0], &gm[1], &gm[2], &gm[3]);
    for (int& v : gm) v = v < 1 ? 1 : v;  // the tile order divides by it
    return true;
  }();
  (void)init;
  switch (slot) {
    case 2: return gm[0];
    case 4: return gm[1];
    case 5: return gm[2];
    case 6: return gm[3];
    default: return 8;
  }
}

int group_m_mx8_for(int slot) {  // the same for the MXFP8 GEMM (TAPCLIP_GM8="q,o,f,p")
  static int gm[4] = {8, 8, 8, 8};
  static const bool init = [] {
    const char* e = getenv("TAPCLIP_GM8");
    if (e) sscanf(e, "%d,%d,%d,%d", &gm[0], &gm[1], &gm[2], &gm[3]);
    for (int& v : gm) v = v < 1 ? 1 : v;
    return true;
  }();
  (void)init;
  switch (slot) {
    case 2: return gm[0];
    case 4: return gm[1];
    case 5: return gm[2];
    case 6: return gm[3];
    default: return 8;
  }
}

// Bisection aids (TAPCLIP_DEBUG_SYNC=<mask>: stream sync after LN (1) / GEMM (2) / attention (4) launches;
// TAPCLIP_DEBUG_STOP=<k>: run_blocks returns early, leaving an UNFINISHED residual) exist only in builds made with
// -DTAPCLIP_DEBUG_KNOBS (tools/dbg_ws2.py): an environment variable must not be able to make the shipped library
// return TAPCLIP_OK with a half-computed result.
#ifdef TAPCLIP_DEBUG_KNOBS
int dbg_sync_mask() {
  static const int m = [] { const char* e = getenv("TAPCLIP_DEBUG_SYNC"); return e ? atoi(e) : 0; }();
  return m;
}
int dbg_stop_at() {
  static const int m = [] { const char* e = getenv("TAPCLIP_DEBUG_STOP"); return e ? atoi(e) : 0; }();
  return m;
}
#define DBG_SYNC(bit, s) do { if (dbg_sync_mask() & (bit)) (void)hipStreamSynchronize(s); } while (0)
#else
constexpr int dbg_stop_at() { return 0; }
#define DBG_SYNC(bit, s) do { } while (0)
#endif

int gemm(tapclip_tower* t, int slot, int epi, const bf16_t* a_hi, const bf16_t* a_lo, int64_t lda, const Packed& w,
         const float* bias, int64_t M, int N, int K, bf16_t* o_hi, bf16_t* o_lo, float* o_f32, int64_t ldo,
         hipStream_t s, const float* add_table = nullptr, int rows_per_group = 0, const bf16_t* aux_hi = nullptr,
         const bf16_t* aux_lo = nullptr) {
  GemmArgs g;
  g.aux_hi = aux_hi; g.aux_lo = aux_lo;
  if (t->split_ws == nullptr && M >= 2048) {  // first large GEMM: allocate the tail-split scratch once
    void* p = nullptr;
    if (dev_alloc(t, gemm256_split_ws_bytes(), &p) == TAPCLIP_OK) t->split_ws = static_cast<float*>(p);
  }
  g.split_ws = t->ksplit ? t->split_ws : nullptr;  // (TAPCLIP_FLAG_KSPLIT: 0 = no K-split of partial rounds -- least CU-time)
  g.A_hi = a_hi; g.A_lo = a_lo; g.lda = lda;
  g.W_hi = w.hi; g.W_lo = w.lo;
  g.bias = bias;
  g.M = M; g.N = N; g.K = K;
  g.out_hi = o_hi; g.out_lo = o_lo; g.out_f32 = o_f32; g.ldo = ldo;
  g.add_table = add_table; g.rows_per_group = rows_per_group;
  g.act = t->cfg.act;
  g.group_m = group_m_for(slot);
  ProfScope ps(t, slot, s);
  HIP_TRY(launch_gemm(g, epi, t->split, s));
  DBG_SYNC(2, s);
  return TAPCLIP_OK;
}

int gemm_mx8(tapclip_tower* t, int slot, int epi, const uint8_t* a_q, const uint8_t* a_s, int64_t m_pad, const PackedMx8& w,
             const float* bias, int64_t M, int N, int K, bf16_t* o_bf16, uint8_t* o_q, uint8_t* o_s, hipStream_t s, int64_t ldo = 0,
             int w_scale_rows = 0) {
  Mx8GemmArgs g;
  g.A = a_q; g.A_scale = a_s; g.lda = K; g.m_pad = m_pad;
  g.W = w.q; g.W_scale = w.s; g.w_scale_rows = w_scale_rows; g.bias = bias;
  g.M = M; g.N = N; g.K = K;
  g.out_bf16 = o_bf16; g.out_q = o_q; g.out_q_scale = o_s; g.out_m_pad = m_pad; g.ldo = ldo > 0 ? ldo : N;
  g.act = t->cfg.act;
  g.group_m = group_m_mx8_for(slot);
  ProfScope ps(t, slot, s);
  HIP_TRY(launch_gemm_mx8(g, epi, s));
  return TAPCLIP_OK;
}

// The fp8 precision of the image tower (BASELINE.json configs[4]): the same block, with the four GEMMs on the
// block-scaled MXFP8 MFMA.  Their A operands are quantised where they are produced -- LayerNorm (layernorm.hip
// MODE 3), the attention core's output (attention.hip store_o_mx8), the GELU epilogue of c_fc (gemm_mx8.hip) --
// q|k|v and the two residual branches stay bf16, the residual stream fp32.
int run_last_block_pooled(tapclip_tower* t, float* x, int64_t n_seq, int tokens, const Workspace& w, float* pooled_out, bf16_t** pooled_d_hi,
                          bf16_t** pooled_d_lo, hipStream_t s);

int run_blocks_fp8(tapclip_tower* t, int64_t n_seq, int tokens, const Workspace& w, hipStream_t s, float* pooled_out = nullptr,
                   bf16_t** pooled_d_hi = nullptr, bf16_t** pooled_d_lo = nullptr) {
  const int64_t M = n_seq * tokens;
  const int D = t->cfg.width, F = t->cfg.mlp_dim, H = t->cfg.heads;
  int rc;
  for (int li = 0; li < t->cfg.layers; ++li) {
    const LayerW& L = t->layers[li];
    {
      ProfScope ps(t, 1, s);
      // block l > 0: x still lacks BOTH branches of block l - 1 (its LN2 did not write x back)
      HIP_TRY(launch_layernorm_mx8(li == 0 ? 0 : 3, nullptr, w.x16, w.a_hi, w.d_hi, L.ln1_g, L.ln1_b, M, D, w.xn_q, w.xn_s, w.m_pad, s));
    }
    if (li == t->cfg.layers - 1 && pooled_out != nullptr)  // CLS-only last block (see run_blocks)
      return run_last_block_pooled(t, nullptr, n_seq, tokens, w, pooled_out, pooled_d_hi, pooled_d_lo, s);
    if ((rc = gemm_mx8(t, 2, EPI_BIAS_BF16, w.xn_q, w.xn_s, w.m_pad, L.qqkv, L.bqkv, M, 3 * D, D, w.qkv_hi, nullptr, nullptr, s))) return rc;
    {
      AttnArgs a;
      a.qkv_hi = w.qkv_hi; a.qkv_lo = nullptr;
      a.out_hi = nullptr; a.out_lo = nullptr;
      a.out_q = w.ao_q; a.out_q_scale = w.ao_s; a.out_m_pad = w.m_pad;
      a.probs = nullptr;
      a.n_seq = (int)n_seq; a.T = tokens; a.H = H; a.D = D; a.causal = 0;
      a.q_log2 = t->q_log2 ? 1 : 0;
      ProfScope ps(t, 3, s);
      HIP_TRY(launch_attention(a, false, s));
    }
    if ((rc = gemm_mx8(t, 4, EPI_BIAS_BF16, w.ao_q, w.ao_s, w.m_pad, L.qo, L.bo, M, D, D, w.a_hi, nullptr, nullptr, s))) return rc;
    {
      ProfScope ps(t, 1, s);
      // the last block folds out_proj's branch into x here, so that only c_proj's is pending on return
      HIP_TRY(launch_layernorm_mx8(li == t->cfg.layers - 1 ? 1 : 2, nullptr, w.x16, w.a_hi, nullptr, L.ln2_g, L.ln2_b, M, D, w.xn_q, w.xn_s, w.m_pad, s));
    }
    if ((rc = gemm_mx8(t, 5, EPI_BIAS_GELU_MX8, w.xn_q, w.xn_s, w.m_pad, L.qfc, L.bfc, M, F, D, nullptr, w.h_q, w.h_s, s))) return rc;
    if ((rc = gemm_mx8(t, 6, EPI_BIAS_BF16, w.h_q, w.h_s, w.m_pad, L.qpr, L.bpr, M, D, F, w.d_hi, nullptr, nullptr, s))) return rc;
  }
  return TAPCLIP_OK;
}

// L residual blocks over x [n_seq*tokens, D] (fp32).  The residual adds are DEFERRED: out_proj and c_proj
// write their branch as bf16 (hi [+ lo]) into w.d and the next LayerNorm kernel applies x += d before
// normalising (the fp32 read-modify-write in a GEMM epilogue cost more than the GEMM's MFMA work at
// N = 768).  On return the last c_proj branch is still pending in w.d: the caller folds it in.
// capture_only: the caller wants the last block's attention capture (probs_last / attn_out_last), not the hidden states
// -- pass 1 of FullModel.forward, reference models/model_wrapper.py:57-62 discards the transformer's output -- so the last
// block stops after its attention core (literal capture: after its fp32 out_proj).
// pooled_out != nullptr (image tower): only the CLS row of the last block's output is wanted -- open_clip pools token 0
// before ln_post / proj, reference call site models/clip_wrapper.py:46-47 -- so the last block runs K / V for every
// token and Q, the attention core, out_proj, LN2 and the MLP for the CLS rows only (run_last_block_pooled); the rest of
// that block is dead work in the reference too, its results are discarded there.  On return pooled_out [n_seq, D] fp32
// holds the CLS rows of the tower's output WITHOUT the last c_proj branch, which is pending in pooled_d (hi [+ lo]).
int run_last_block_pooled(tapclip_tower* t, float* x, int64_t n_seq, int tokens, const Workspace& w, float* pooled_out, bf16_t** pooled_d_hi,
                          bf16_t** pooled_d_lo, hipStream_t s);

int run_blocks(tapclip_tower* t, float* x, int64_t n_seq, int tokens, int causal, const Workspace& w,
               float* probs_last, float* attn_out_last, hipStream_t s, bool capture_only = false, float* pooled_out = nullptr,
               bf16_t** pooled_d_hi = nullptr, bf16_t** pooled_d_lo = nullptr, float last_key_bias = 0.f) {
  const bool x24 = t->x24 && t->cfg.kind == TAPCLIP_TOWER_VISION;  // residual stream in w.x24_hi / w.x24_lo instead of x
  // image tower, 16-bit modes: the fp32 residual rows are streamed past the caches (layernorm.hip NTX)
  static const bool no_ntx = getenv("TAPCLIP_NO_STREAM_X") != nullptr;
  const bool stream_x = !x24 && !t->split && t->cfg.kind == TAPCLIP_TOWER_VISION && !no_ntx;
  const int64_t M = n_seq * tokens;
  const int D = t->cfg.width, F = t->cfg.mlp_dim, H = t->cfg.heads;
  for (int li = 0; li < t->cfg.layers; ++li) {
    const LayerW& L = t->layers[li];
    const bool last = li == t->cfg.layers - 1;
    {
      ProfScope ps(t, 1, s);
      // block l > 0: x still lacks BOTH branches of block l - 1 (its LN2 did not write x back)
      if (x24) {
        // (block 0: ln_pre + this LayerNorm were one kernel, launched by the caller)
        if (li == 0) ps.cancel();
        else HIP_TRY(launch_layernorm_x24(3, 0, nullptr, 0, w.x24_hi, w.x24_lo, w.a_hi, w.d_hi, nullptr, nullptr, L.ln1_g, L.ln1_b, M, D, w.xn_hi, s));
      } else if (li == 0) HIP_TRY(launch_layernorm(x, D, L.ln1_g, L.ln1_b, M, D, w.xn_hi, w.xn_lo, nullptr, s, stream_x));
      else HIP_TRY(launch_add_layernorm_ex(3, x, w.a_hi, w.a_lo, w.d_hi, w.d_lo, L.ln1_g, L.ln1_b, M, D, w.xn_hi, w.xn_lo, s, stream_x));
      DBG_SYNC(1, s);
    }
    if (last && pooled_out != nullptr) return run_last_block_pooled(t, x, n_seq, tokens, w, pooled_out, pooled_d_hi, pooled_d_lo, s);
    const int dbg_stop = dbg_stop_at();
    if (dbg_stop == 1 && li == 1) return TAPCLIP_OK;
    if (dbg_stop == 5 && li == 0) return TAPCLIP_OK;  // (3, 4, 5: block 0 after its first LayerNorm / LN2 / c_fc -- tools/dbg_ws2.py)
    int rc = gemm(t, 2, EPI_BIAS_BF16, w.xn_hi, w.xn_lo, D, L.wqkv, L.bqkv, M, 3 * D, D, w.qkv_hi, w.qkv_lo, nullptr,
                  3 * D, s);
    if (rc) return rc;
    if (dbg_stop == 2 && li == 1) return TAPCLIP_OK;
    {
      AttnArgs a;
      a.qkv_hi = w.qkv_hi; a.qkv_lo = w.qkv_lo;
      a.out_hi = w.ao_hi; a.out_lo = w.ao_lo;
      a.probs = last ? probs_last : nullptr;
      a.n_seq = (int)n_seq; a.T = tokens; a.H = H; a.D = D; a.causal = causal;
      a.last_key_bias = last_key_bias;
      a.q_log2 = t->q_log2 ? 1 : 0;
      ProfScope ps(t, 3, s);
      HIP_TRY(launch_attention(a, t->split, s));
      DBG_SYNC(4, s);
    }
    if (last && attn_out_last != nullptr) {
      // what the reference's hook literally captures: the attention module's output (pre residual), fp32
      rc = gemm(t, 4, EPI_BIAS_F32, w.ao_hi, w.ao_lo, D, L.wo, L.bo, M, D, D, nullptr, nullptr, attn_out_last, D, s);
      if (rc) return rc;
    }
    if (last && capture_only) return TAPCLIP_OK;
    rc = gemm(t, 4, EPI_BIAS_BF16, w.ao_hi, w.ao_lo, D, L.wo, L.bo, M, D, D, w.a_hi, w.a_lo, nullptr, D, s);
    if (rc) return rc;
    {
      ProfScope ps(t, 1, s);
      // LN2 normalises x + branch without writing x back (8 instead of 12 B/element); the last block does write,
      // so that only c_proj's branch is pending on return
      if (x24) HIP_TRY(launch_layernorm_x24(last ? 1 : 2, 0, nullptr, 0, w.x24_hi, w.x24_lo, w.a_hi, nullptr, nullptr, nullptr, L.ln2_g, L.ln2_b, M, D, w.xn_hi, s));
      else HIP_TRY(launch_add_layernorm_ex(last ? 1 : 2, x, w.a_hi, w.a_lo, nullptr, nullptr, L.ln2_g, L.ln2_b, M, D, w.xn_hi, w.xn_lo, s, stream_x));
      DBG_SYNC(1, s);
    }
    if (dbg_stop == 3 && li == 0) return TAPCLIP_OK;
    rc = gemm(t, 5, EPI_BIAS_GELU_BF16, w.xn_hi, w.xn_lo, D, L.wfc, L.bfc, M, F, D, w.h_hi, w.h_lo, nullptr, F, s);
    if (rc) return rc;
    if (dbg_stop == 4 && li == 0) return TAPCLIP_OK;
    rc = gemm(t, 6, EPI_BIAS_BF16, w.h_hi, w.h_lo, F, L.wpr, L.bpr, M, D, F, w.d_hi, w.d_lo, nullptr, D, s);
    if (rc) return rc;
  }
  return TAPCLIP_OK;
}


int run_last_block_pooled(tapclip_tower* t, float* x, int64_t n_seq, int tokens, const Workspace& w, float* pooled_out, bf16_t** pooled_d_hi,
                          bf16_t** pooled_d_lo, hipStream_t s) {
  // (LN1 of this block has run: x -- or its 24-bit planes -- holds every earlier branch, w.xn its normalised rows)
  const LayerW& L = t->layers[t->cfg.layers - 1];
  const int64_t M = n_seq * tokens;
  const int D = t->cfg.width, F = t->cfg.mlp_dim, H = t->cfg.heads;
  const bool x24 = t->x24;
  int rc;
  // buffers of the CLS-row tensors: fronts of workspace regions that are dead by the time they are written
  bf16_t *q_hi = w.h_hi, *q_lo = w.h_lo;        // [n, D]   (h: free until the pooled c_fc)
  bf16_t *ao_hi = w.ao_hi, *ao_lo = w.ao_lo;    // [n, D]
  bf16_t *a_hi = w.a_hi, *a_lo = w.a_lo;        // [n, D]   (the previous block's branch was folded by LN1)
  bf16_t *xn_hi = w.xn_hi, *xn_lo = w.xn_lo;    // [n, D]   (after the K / V and Q GEMMs have read every row)
  bf16_t *h_hi = w.h_hi, *h_lo = w.h_lo;        // [n, F]   (after the attention core has read q)
  bf16_t *d_hi = w.qkv_hi, *d_lo = w.qkv_lo;    // [n, D]   (after the attention core has read k, v)
  // K and V of every token: rows D .. 3D - 1 of in_proj into columns D .. 3D - 1 of the q|k|v buffer
  if (t->fp8) {
    // (MXFP8 operands: the weight's row range is addressed inside its full-height scale plane)
    PackedMx8 wkv{L.qqkv.q + (size_t)D * D, L.qqkv.s + (size_t)D * 2};
    if ((rc = gemm_mx8(t, 2, EPI_BIAS_BF16, w.xn_q, w.xn_s, w.m_pad, wkv, L.bqkv + D, M, 2 * D, D, w.qkv_hi + D, nullptr, nullptr, s, 3 * D, 3 * D)))
      return rc;
  } else {
    Packed wkv{L.wqkv.hi + (size_t)D * D, L.wqkv.lo ? L.wqkv.lo + (size_t)D * D : nullptr};
    if ((rc = gemm(t, 2, EPI_BIAS_BF16, w.xn_hi, w.xn_lo, D, wkv, L.bqkv + D, M, 2 * D, D, w.qkv_hi + D, w.qkv_lo ? w.qkv_lo + D : nullptr,
                   nullptr, 3 * D, s))) return rc;
  }
  ProfScope ps(t, 8, s);  // everything on the n_seq CLS rows: its own slot ("pooled_tail"); slot 7 stays the pool / ln_post / proj kernel
  // fp8 precision: the block's LayerNorm wrote MXFP8 rows only, and the residual stream is 16-bit (w.x16): the CLS rows are
  // gathered to fp32 first and normalised again in 16 bits for the (16-bit) skinny GEMMs of the pooled tail
  const bf16_t* qa_hi = w.xn_hi;
  const bf16_t* qa_lo = w.xn_lo;
  int64_t qa_ld = (int64_t)tokens * D;
  if (t->fp8) {
    HIP_TRY(launch_gather_cls16(w.x16, nullptr, (int)n_seq, tokens, D, pooled_out, s));
    HIP_TRY(launch_layernorm(pooled_out, D, L.ln1_g, L.ln1_b, n_seq, D, w.xn_hi, nullptr, nullptr, s));  // (xn_q is dead: K | V are done)
    qa_lo = nullptr;
    qa_ld = D;
  }
  // Q of the CLS rows: A = row b * tokens of xn (row stride tokens * D)
  auto small_gemm = [&](int epi, const bf16_t* ah, const bf16_t* al, int64_t lda, const Packed& wt, const float* bias, int N, int K, bf16_t* oh,
                        bf16_t* ol) {
    GemmArgs a;
    a.A_hi = ah; a.A_lo = al; a.lda = lda;
    a.W_hi = wt.hi; a.W_lo = wt.lo;
    a.bias = bias;
    a.M = n_seq; a.N = N; a.K = K;
    a.out_hi = oh; a.out_lo = ol; a.out_f32 = nullptr; a.ldo = N;
    a.add_table = nullptr; a.rows_per_group = 0; a.act = t->cfg.act;
    a.split_ws = nullptr;  // (the tiled fallback below runs WITHOUT the K-split of partial rounds: its split is chosen from the tile count, i.e. from the batch)
    // a few hundred rows: the split-K skinny kernel (gemm_skinny.hip), for EVERY batch size -- its K slices depend on (N, K)
    // only, so a pooled row sums in the same order whatever its batch (bitwise batch invariance).  Its fp32 slabs live in
    // the handle's K-split scratch (allocated here when no large GEMM has done it yet: small batches).
    if (gemm_skinny_supports(a, epi) && gemm_skinny_ws_bytes(n_seq, N, K) <= gemm256_split_ws_bytes()) {
      if (t->split_ws == nullptr) {
        void* p = nullptr;
        if (dev_alloc(t, gemm256_split_ws_bytes(), &p) != TAPCLIP_OK) return hipErrorOutOfMemory;
        t->split_ws = static_cast<float*>(p);
      }
      return launch_gemm_skinny(a, epi, t->split, t->split_ws, gemm256_split_ws_bytes(), s);
    }
    // shapes the skinny kernel does not take (more than 1024 rows, N or K off its tiling): the tiled kernels, whole tiles only
    // -- still a fixed summation order per row, but which kernel runs now depends on the batch, so the bitwise batch
    // invariance of the pooled rows is a property of the skinny path only
    return launch_gemm(a, epi, t->split, s);
  };
  HIP_TRY(small_gemm(EPI_BIAS_BF16, qa_hi, qa_lo, qa_ld, L.wqkv, L.bqkv, D, D, q_hi, q_lo));
  HIP_TRY(launch_attention_pooled(q_hi, q_lo, w.qkv_hi, w.qkv_lo, ao_hi, ao_lo, (int)n_seq, tokens, H, D, t->split, s, t->q_log2));
  HIP_TRY(small_gemm(EPI_BIAS_BF16, ao_hi, ao_lo, D, L.wo, L.bo, D, D, a_hi, a_lo));
  // the CLS rows of the residual stream -> fp32 [n, D], + out_proj's branch, LN2
  if (t->fp8) {
    // (gathered above, before the CLS rows' LN1)
  } else if (x24) HIP_TRY(launch_gather_cls24(w.x24_hi, w.x24_lo, nullptr, (int)n_seq, tokens, D, pooled_out, s));
  else HIP_TRY(hipMemcpy2DAsync(pooled_out, (size_t)D * 4, x, (size_t)tokens * D * 4, (size_t)D * 4, (size_t)n_seq, hipMemcpyDeviceToDevice, s));
  HIP_TRY(launch_add_layernorm(pooled_out, a_hi, a_lo, L.ln2_g, L.ln2_b, n_seq, D, xn_hi, xn_lo, s));
  HIP_TRY(small_gemm(EPI_BIAS_GELU_BF16, xn_hi, xn_lo, D, L.wfc, L.bfc, F, D, h_hi, h_lo));
  HIP_TRY(small_gemm(EPI_BIAS_BF16, h_hi, h_lo, F, L.wpr, L.bpr, D, F, d_hi, d_lo));
  *pooled_d_hi = d_hi;
  *pooled_d_lo = d_lo;
  return TAPCLIP_OK;
}

// ---- saved activations of the recomputed forward (backward only)
struct Saved {
  std::vector<float*> x0, x1;
  std::vector<bf16_t*> qkv_hi, qkv_lo, ao_hi, ao_lo;
  size_t bytes = 0;
};

Saved carve_saved(const tapclip_tower* t, int64_t M, void* base) {
  Saved sv;
  const int64_t D = t->cfg.width;
  const int L = t->cfg.layers;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    void* p = base ? static_cast<char*>(base) + off : nullptr;
    off += align_up(bytes);
    return p;
  };
  for (int l = 0; l < L; ++l) {
    sv.x0.push_back(static_cast<float*>(take(M * D * 4)));
    sv.x1.push_back(static_cast<float*>(take(M * D * 4)));
    sv.qkv_hi.push_back(static_cast<bf16_t*>(take(M * 3 * D * 2)));
    sv.ao_hi.push_back(static_cast<bf16_t*>(take(M * D * 2)));
    sv.qkv_lo.push_back(t->split ? static_cast<bf16_t*>(take(M * 3 * D * 2)) : nullptr);
    sv.ao_lo.push_back(t->split ? static_cast<bf16_t*>(take(M * D * 2)) : nullptr);
  }
  sv.bytes = off;
  return sv;
}

// Forward of the blocks that keeps what the backward needs (sv): the residual stream before each LayerNorm, q|k|v and
// the attention output of every block.  The residual stream is not copied into sv: it MOVES through it -- every
// add + LayerNorm reads the previous saved row block and writes the updated rows into the next one (x0[0] <- x_in,
// x1[l] = x0[l] + attention branch, x0[l+1] = x1[l] + MLP branch); 25 device copies of [M, D] fp32 per call are gone.
// On return *x_last = x1[L-1]: x without the last c_proj branch, which is pending in w.d.
int run_forward_saving(tapclip_tower* t, const float* x_in, int64_t n_seq, int tokens, int causal, const Workspace& w,
                       const Saved& sv, hipStream_t s, const float** x_last, float last_key_bias = 0.f) {
  const int64_t M = n_seq * tokens;
  const int D = t->cfg.width, F = t->cfg.mlp_dim, H = t->cfg.heads, L = t->cfg.layers;
  // (x_in == nullptr: the caller has written the input rows into sv.x0[0] itself -- the tied-padding entry point compacts into it)
  if (x_in != nullptr) HIP_TRY(hipMemcpyAsync(sv.x0[0], x_in, (size_t)M * D * 4, hipMemcpyDeviceToDevice, s));
  int rc;
  for (int li = 0; li < L; ++li) {
    const LayerW& Lw = t->layers[li];
    if (li == 0) HIP_TRY(launch_layernorm(sv.x0[0], D, Lw.ln1_g, Lw.ln1_b, M, D, w.xn_hi, w.xn_lo, nullptr, s));
    else HIP_TRY(launch_add_layernorm(sv.x1[li - 1], w.d_hi, w.d_lo, Lw.ln1_g, Lw.ln1_b, M, D, w.xn_hi, w.xn_lo, s, sv.x0[li]));
    if ((rc = gemm(t, 2, EPI_BIAS_BF16, w.xn_hi, w.xn_lo, D, Lw.wqkv, Lw.bqkv, M, 3 * D, D, sv.qkv_hi[li], sv.qkv_lo[li], nullptr, 3 * D, s))) return rc;
    AttnArgs a;
    a.qkv_hi = sv.qkv_hi[li]; a.qkv_lo = sv.qkv_lo[li];
    a.out_hi = sv.ao_hi[li]; a.out_lo = sv.ao_lo[li];
    a.probs = nullptr;
    a.n_seq = (int)n_seq; a.T = tokens; a.H = H; a.D = D; a.causal = causal;
    a.last_key_bias = last_key_bias;
    HIP_TRY(launch_attention(a, t->split, s));
    if ((rc = gemm(t, 4, EPI_BIAS_BF16, sv.ao_hi[li], sv.ao_lo[li], D, Lw.wo, Lw.bo, M, D, D, w.d_hi, w.d_lo, nullptr, D, s))) return rc;
    HIP_TRY(launch_add_layernorm(sv.x0[li], w.d_hi, w.d_lo, Lw.ln2_g, Lw.ln2_b, M, D, w.xn_hi, w.xn_lo, s, sv.x1[li]));
    if ((rc = gemm(t, 5, EPI_BIAS_GELU_BF16, w.xn_hi, w.xn_lo, D, Lw.wfc, Lw.bfc, M, F, D, w.h_hi, w.h_lo, nullptr, F, s))) return rc;
    if ((rc = gemm(t, 6, EPI_BIAS_BF16, w.h_hi, w.h_lo, F, Lw.wpr, Lw.bpr, M, D, F, w.d_hi, w.d_lo, nullptr, D, s))) return rc;
  }
  *x_last = sv.x1[L - 1];
  return TAPCLIP_OK;
}

// Backward sweep over the saved activations.  dx (fp32 [M, D]) enters holding dL/d(hidden) and leaves holding
// dL/d(x_in).  Scratch re-uses the forward workspace: g = w.d (branch gradient as a GEMM operand),
// w.h = dL/dh then dL/dz, w.ao = dL/d(attention out), w.qkv = dL/d(qkv), w.x = fp32 dL/d(LN output)
int run_backward_sweep(tapclip_tower* t, float* dx, int64_t n_seq, int tokens, int causal, const Workspace& w,
                       const Saved& sv, hipStream_t s, float last_key_bias = 0.f) {
  const int64_t M = n_seq * tokens;
  const int D = t->cfg.width, F = t->cfg.mlp_dim, H = t->cfg.heads, L = t->cfg.layers;
  int rc;
  float* dn = w.x;
  // (the 16-bit planes of dx that each branch's first GEMM reads are written by the LayerNorm backward that produced dx;
  // only the gradient that enters the sweep is packed by its own pass)
  HIP_TRY(launch_pack(dx, M, D, D, D, 0, 1.f, w.d_hi, w.d_lo, s));
  for (int li = L - 1; li >= 0; --li) {
    const LayerW& Lw = t->layers[li];
    // MLP branch: m = gelu(LN2(x1) Wfc^T + bfc) Wpr^T + bpr
    if ((rc = gemm(t, 6, EPI_BIAS_BF16, w.d_hi, w.d_lo, D, Lw.wpr_t, nullptr, M, F, D, w.h_hi, w.h_lo, nullptr, F, s))) return rc;
    HIP_TRY(launch_layernorm(sv.x1[li], D, Lw.ln2_g, Lw.ln2_b, M, D, w.xn_hi, w.xn_lo, nullptr, s));
    if ((rc = gemm(t, 5, EPI_GELU_BWD_BF16, w.xn_hi, w.xn_lo, D, Lw.wfc, Lw.bfc, M, F, D, w.h_hi, w.h_lo, nullptr, F, s, nullptr, 0, w.h_hi, w.h_lo))) return rc;
    if ((rc = gemm(t, 5, EPI_BIAS_F32, w.h_hi, w.h_lo, F, Lw.wfc_t, nullptr, M, D, F, nullptr, nullptr, dn, D, s))) return rc;
    HIP_TRY(launch_ln_bwd(sv.x1[li], Lw.ln2_g, dn, M, D, dx, w.d_hi, w.d_lo, s));
    // attention branch: a = attention(LN1(x0) Wqkv^T + b) Wo^T + bo
    if ((rc = gemm(t, 4, EPI_BIAS_BF16, w.d_hi, w.d_lo, D, Lw.wo_t, nullptr, M, D, D, w.ao_hi, w.ao_lo, nullptr, D, s))) return rc;
    AttnBwdArgs b;
    b.qkv_hi = sv.qkv_hi[li]; b.qkv_lo = sv.qkv_lo[li];
    b.out_hi = sv.ao_hi[li]; b.out_lo = sv.ao_lo[li];
    b.dout_hi = w.ao_hi; b.dout_lo = w.ao_lo;
    b.dqkv_hi = w.qkv_hi; b.dqkv_lo = w.qkv_lo;
    b.n_seq = (int)n_seq; b.T = tokens; b.H = H; b.D = D; b.causal = causal;
    b.last_key_bias = last_key_bias;
    HIP_TRY(launch_attention_bwd(b, s));
    if ((rc = gemm(t, 2, EPI_BIAS_F32, w.qkv_hi, w.qkv_lo, 3 * D, Lw.wqkv_t, nullptr, M, D, 3 * D, nullptr, nullptr, dn, D, s))) return rc;
    HIP_TRY(launch_ln_bwd(sv.x0[li], Lw.ln1_g, dn, M, D, dx, li > 0 ? w.d_hi : nullptr, li > 0 ? w.d_lo : nullptr, s));
  }
  return TAPCLIP_OK;
}

// Forward again (activation recomputation keeps tapclip_text_backward stateless), then the sweep.
int run_backward(tapclip_tower* t, const float* x_in, float* dx, int64_t n_seq, int tokens, int causal,
                 const Workspace& w, const Saved& sv, hipStream_t s) {
  const float* x_last = nullptr;
  int rc = run_forward_saving(t, x_in, n_seq, tokens, causal, w, sv, s, &x_last);
  if (rc) return rc;
  return run_backward_sweep(t, dx, n_seq, tokens, causal, w, sv, s);
}

int check_ready(const tapclip_tower* t) {
  if (t->loaded.size() == t->required.size()) return TAPCLIP_OK;
  std::string missing;
  int n = 0;
  for (const auto& k : t->required)
    if (!t->loaded.count(k)) {
      if (n++ < 6) missing += (missing.empty() ? "" : ", ") + k;
    }
  return fail(TAPCLIP_ESTATE, "tower is missing %d weight tensor(s): %s%s", n, missing.c_str(), n > 6 ? ", ..." : "");
}

}  // namespace

extern "C" {

const char* tapclip_last_error(void) { return g_err; }
int tapclip_abi_version(void) { return TAPCLIP_ABI_VERSION; }

int tapclip_tower_create(const tapclip_tower_cfg* cfg, tapclip_tower_t** out) {
  if (!cfg || !out) return fail(TAPCLIP_EINVAL, "null argument");
  if (cfg->kind != TAPCLIP_TOWER_VISION && cfg->kind != TAPCLIP_TOWER_TEXT) return fail(TAPCLIP_EINVAL, "bad tower kind %d", cfg->kind);
  if (cfg->width <= 0 || cfg->width % 128 != 0) return fail(TAPCLIP_EINVAL, "width %d must be a positive multiple of 128", cfg->width);
  if (cfg->heads <= 0 || cfg->width != cfg->heads * 64) return fail(TAPCLIP_EINVAL, "head dim must be 64 (width %d, heads %d)", cfg->width, cfg->heads);
  if (cfg->mlp_dim <= 0 || cfg->mlp_dim % 128 != 0) return fail(TAPCLIP_EINVAL, "mlp_dim %d must be a positive multiple of 128", cfg->mlp_dim);
  if (cfg->layers <= 0 || cfg->embed_dim <= 0 || cfg->embed_dim > 1024) return fail(TAPCLIP_EINVAL, "bad layers/embed_dim");
  if (cfg->precision != TAPCLIP_PREC_BF16 && cfg->precision != TAPCLIP_PREC_BF16X3 && cfg->precision != TAPCLIP_PREC_FP8) return fail(TAPCLIP_EINVAL, "bad precision %d", cfg->precision);
  if (cfg->precision == TAPCLIP_PREC_FP8) {
    if (cfg->kind != TAPCLIP_TOWER_VISION) return fail(TAPCLIP_EINVAL, "TAPCLIP_PREC_FP8 is an image-tower precision (the text tower is differentiated: use bf16 / bf16x3)");
    if (cfg->width % 256 != 0 || cfg->width > 1024 || cfg->mlp_dim % 256 != 0 || cfg->mlp_dim > 4096) return fail(TAPCLIP_EINVAL, "TAPCLIP_PREC_FP8 needs width %% 256 == 0 (<= 1024) and mlp_dim %% 256 == 0 (<= 4096), got %d / %d", cfg->width, cfg->mlp_dim);
  }
  if (cfg->act != TAPCLIP_ACT_GELU_ERF && cfg->act != TAPCLIP_ACT_QUICK_GELU) return fail(TAPCLIP_EINVAL, "bad activation %d", cfg->act);
  tapclip_tower* t = new tapclip_tower();
  t->cfg = *cfg;
  t->split = cfg->precision == TAPCLIP_PREC_BF16X3;
  t->fp8 = cfg->precision == TAPCLIP_PREC_FP8;
  t->layers.resize(cfg->layers);
  if (cfg->kind == TAPCLIP_TOWER_VISION) {
    if (cfg->patch <= 0 || cfg->image_size <= 0 || cfg->image_size % cfg->patch != 0) {
      delete t;
      return fail(TAPCLIP_EINVAL, "image_size %d must be a positive multiple of patch %d", cfg->image_size, cfg->patch);
    }
    const int G = cfg->image_size / cfg->patch;
    t->tokens_vision = G * G + 1;
    // 24-bit residual planes (layernorm.hip XF = 2) in the 16-bit modes: -1.3 % on the step.  TAPCLIP_X24=0 keeps the
    // fp32 stream.  (They were opt-in for a while: a tower on them returned wrong rows beside a busy second stream.
    // Not a race -- their LayerNorms were the kernels in which hipcc formed v_pk_fma_f32 with op_sel:[0,1,0], which
    // MI355X gets wrong in lanes 48..63 beside another kernel's MFMAs: common.h TAPCLIP_TU_NO_PK_F32, DESIGN.md.)
    static const bool want_x24 = [] { const char* e = getenv("TAPCLIP_X24"); return e == nullptr || atoi(e) != 0; }();
    t->x24 = !t->split && !t->fp8 && layernorm_x24_supports(cfg->width) && want_x24;
    static const bool no_prune = [] { const char* e = getenv("TAPCLIP_PRUNE_LAST"); return e && atoi(e) == 0; }();
    t->prune_last = !no_prune;
    // the towers whose attention runs in attention_long.hip hand it base-2 scores: log2(e) goes into the q rows of in_proj at
    // pack time (one rounding of the product, as for any weight), the kernel's exp2 then takes the MFMA output as it is
    // (-3 % of the kernel).  TAPCLIP_Q_LOG2=0 at creation keeps natural-log scores (A/B); so does the A/B switch to the first
    // flash kernel, which does not know the flag.
    static const bool no_q_log2 = [] { const char* e = getenv("TAPCLIP_Q_LOG2"); return e && atoi(e) == 0; }();
    t->q_log2 = t->tokens_vision > 256 && !t->split && !no_q_log2 && flash2_cfg() != 1;
    t->Kp = (3 * cfg->patch * cfg->patch + 63) / 64 * 64;
    for (const char* k : {"conv1.weight", "class_embedding", "positional_embedding", "ln_pre.weight", "ln_pre.bias",
                          "ln_post.weight", "ln_post.bias", "proj"})
      t->required.insert(k);
  } else {
    if (cfg->ctx_len <= 0 || cfg->vocab <= 0) {
      delete t;
      return fail(TAPCLIP_EINVAL, "text tower needs ctx_len and vocab");
    }
    for (const char* k : {"token_embedding.weight", "positional_embedding", "ln_final.weight", "ln_final.bias", "text_projection"})
      t->required.insert(k);
  }
  for (int i = 0; i < cfg->layers; ++i)
    for (const char* k : {"ln_1.weight", "ln_1.bias", "attn.in_proj_weight", "attn.in_proj_bias", "attn.out_proj.weight",
                          "attn.out_proj.bias", "ln_2.weight", "ln_2.bias", "mlp.c_fc.weight", "mlp.c_fc.bias",
                          "mlp.c_proj.weight", "mlp.c_proj.bias"})
      t->required.insert("transformer.resblocks." + std::to_string(i) + "." + k);
  *out = t;
  return TAPCLIP_OK;
}

void tapclip_tower_destroy(tapclip_tower_t* t) {
  if (!t) return;
  for (void* p : t->allocs) (void)hipFree(p);
  for (auto& r : t->prof) {
    (void)hipEventDestroy(r.start);
    (void)hipEventDestroy(r.stop);
  }
  delete t;
}

int tapclip_tower_load_weight(tapclip_tower_t* t, const char* key_c, const float* src, const int64_t* shape, int32_t ndim,
                              tapclip_stream_t stream) {
  if (!t || !key_c || !src || !shape) return fail(TAPCLIP_EINVAL, "null argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const std::string key(key_c);
  if (!t->required.count(key)) return fail(TAPCLIP_EINVAL, "unexpected key '%s' for this tower", key_c);
  if (t->loaded.count(key)) return fail(TAPCLIP_ESTATE, "key '%s' loaded twice", key_c);
  const int64_t D = t->cfg.width, F = t->cfg.mlp_dim, E = t->cfg.embed_dim;
  auto bad = [&](const char* want) {
    return fail(TAPCLIP_EINVAL, "size mismatch for %s: got %s, expected %s", key_c, shape_str(shape, ndim).c_str(), want);
  };
  int rc = TAPCLIP_OK;
  const std::string pre = "transformer.resblocks.";
  if (key.compare(0, pre.size(), pre) == 0) {
    const size_t dot = key.find('.', pre.size());
    const int li = atoi(key.substr(pre.size(), dot - pre.size()).c_str());
    const std::string sub = key.substr(dot + 1);
    LayerW& L = t->layers[li];
    auto vec = [&](int64_t n, float** dst, int64_t scale_n = 0, float scale = 1.f) -> int {
      if (!shape_is(shape, ndim, {n})) return bad(("[" + std::to_string(n) + "]").c_str());
      return own_f32(t, src, n, scale_n, scale, dst, s);
    };
    auto mat = [&](int64_t rows, int64_t cols, Packed* dst, Packed* dst_t, int64_t scale_rows = 0, float scale = 1.f) -> int {
      if (!shape_is(shape, ndim, {rows, cols})) return bad(("[" + std::to_string(rows) + "," + std::to_string(cols) + "]").c_str());
      int r = own_packed(t, src, rows, (int)cols, (int)cols, scale_rows, scale, dst, s);
      if (r || t->cfg.kind != TAPCLIP_TOWER_TEXT) return r;
      // W^T [cols, rows] for dX = dY . W (only the text tower is ever differentiated)
      void *h, *l = nullptr;
      if ((r = dev_alloc(t, rows * cols * 2, &h))) return r;
      if (t->split && (r = dev_alloc(t, rows * cols * 2, &l))) return r;
      HIP_TRY(launch_pack_transpose(src, rows, (int)cols, scale_rows, scale, static_cast<bf16_t*>(h), static_cast<bf16_t*>(l), s));
      dst_t->hi = static_cast<bf16_t*>(h);
      dst_t->lo = static_cast<bf16_t*>(l);
      return TAPCLIP_OK;
    };
    // folded softmax scale (exact in bf16), times log2(e) where the attention kernel reads base-2 scores
    const float qscale = (1.0f / sqrtf(64.0f)) * (t->q_log2 ? 1.44269504088896340736f : 1.0f);
    if (sub == "ln_1.weight") rc = vec(D, &L.ln1_g);
    else if (sub == "ln_1.bias") rc = vec(D, &L.ln1_b);
    else if (sub == "ln_2.weight") rc = vec(D, &L.ln2_g);
    else if (sub == "ln_2.bias") rc = vec(D, &L.ln2_b);
    else if (t->fp8 && (sub == "attn.in_proj_weight" || sub == "attn.out_proj.weight" || sub == "mlp.c_fc.weight" || sub == "mlp.c_proj.weight")) {
      // fp8 precision: the block GEMM weights are kept as MXFP8 only (1/sqrt(64) folded into the q rows first)
      const bool qkv = sub == "attn.in_proj_weight", fc = sub == "mlp.c_fc.weight", pr = sub == "mlp.c_proj.weight";
      const int64_t rows = qkv ? 3 * D : fc ? F : D, cols = pr ? F : D;
      if (!shape_is(shape, ndim, {rows, cols})) return bad(("[" + std::to_string(rows) + "," + std::to_string(cols) + "]").c_str());
      rc = own_packed_mx8(t, src, rows, (int)cols, qkv ? D : 0, qkv ? qscale : 1.f, qkv ? &L.qqkv : fc ? &L.qfc : pr ? &L.qpr : &L.qo, s);
      // the LAST block also keeps 16-bit copies: its CLS-only tail (run_last_block_pooled) runs the M = batch GEMMs of
      // the pooled rows on the 16-bit skinny kernel (14 - 25 MB per tower; those rows then see no MXFP8 rounding at all)
      if (!rc && li == t->cfg.layers - 1 && t->prune_last)  // (TAPCLIP_PRUNE_LAST=0 at creation: no pooled tail, no copies)
        rc = own_packed(t, src, rows, (int)cols, (int)cols, qkv ? D : 0, qkv ? qscale : 1.f, qkv ? &L.wqkv : fc ? &L.wfc : pr ? &L.wpr : &L.wo, s);
    }
    else if (sub == "attn.in_proj_weight") rc = mat(3 * D, D, &L.wqkv, &L.wqkv_t, D, qscale);
    else if (sub == "attn.in_proj_bias") rc = vec(3 * D, &L.bqkv, D, qscale);
    else if (sub == "attn.out_proj.weight") rc = mat(D, D, &L.wo, &L.wo_t);
    else if (sub == "attn.out_proj.bias") rc = vec(D, &L.bo);
    else if (sub == "mlp.c_fc.weight") rc = mat(F, D, &L.wfc, &L.wfc_t);
    else if (sub == "mlp.c_fc.bias") rc = vec(F, &L.bfc);
    else if (sub == "mlp.c_proj.weight") rc = mat(D, F, &L.wpr, &L.wpr_t);
    else if (sub == "mlp.c_proj.bias") rc = vec(D, &L.bpr);
    else return fail(TAPCLIP_EINVAL, "unexpected key '%s'", key_c);
  } else if (t->cfg.kind == TAPCLIP_TOWER_VISION) {
    const int64_t p = t->cfg.patch, N = t->tokens_vision;
    if (key == "conv1.weight") {
      if (!shape_is(shape, ndim, {D, 3, p, p})) return bad("[width,3,patch,patch]");
      rc = own_packed(t, src, D, (int)(3 * p * p), t->Kp, 0, 1.f, &t->conv, s);
    } else if (key == "class_embedding") {
      if (!shape_is(shape, ndim, {D})) return bad("[width]");
      rc = own_f32(t, src, D, 0, 1.f, &t->cls, s);
    } else if (key == "positional_embedding") {
      if (!shape_is(shape, ndim, {N, D})) return bad("[tokens,width]");
      rc = own_f32(t, src, N * D, 0, 1.f, &t->pos, s);
    } else if (key == "proj") {
      if (!shape_is(shape, ndim, {D, E})) return bad("[width,embed_dim]");
      rc = own_f32(t, src, D * E, 0, 1.f, &t->proj, s);
    } else {
      if (!shape_is(shape, ndim, {D})) return bad("[width]");
      float** dst = key == "ln_pre.weight" ? &t->lnpre_g : key == "ln_pre.bias" ? &t->lnpre_b : key == "ln_post.weight" ? &t->lnpost_g : &t->lnpost_b;
      rc = own_f32(t, src, D, 0, 1.f, dst, s);
    }
  } else {
    if (key == "token_embedding.weight") {
      if (!shape_is(shape, ndim, {(int64_t)t->cfg.vocab, D})) return bad("[vocab,width]");
      rc = own_f32(t, src, (int64_t)t->cfg.vocab * D, 0, 1.f, &t->tok_emb, s);
    } else if (key == "positional_embedding") {
      if (!shape_is(shape, ndim, {(int64_t)t->cfg.ctx_len, D})) return bad("[ctx_len,width]");
      rc = own_f32(t, src, (int64_t)t->cfg.ctx_len * D, 0, 1.f, &t->pos, s);
    } else if (key == "text_projection") {
      if (!shape_is(shape, ndim, {D, E})) return bad("[width,embed_dim]");
      rc = own_f32(t, src, D * E, 0, 1.f, &t->text_proj, s);
    } else {
      if (!shape_is(shape, ndim, {D})) return bad("[width]");
      rc = own_f32(t, src, D, 0, 1.f, key == "ln_final.weight" ? &t->lnfin_g : &t->lnfin_b, s);
    }
  }
  if (rc) return rc;
  t->loaded.insert(key);
  return TAPCLIP_OK;
}

int tapclip_tower_ready(const tapclip_tower_t* t) {
  if (!t) return fail(TAPCLIP_EINVAL, "null tower");
  return check_ready(t);
}

size_t tapclip_tower_workspace_bytes(const tapclip_tower_t* t, int64_t n_seq, int32_t tokens) {
  if (!t || n_seq <= 0 || tokens <= 0) return 0;
  return carve(t, n_seq, tokens, nullptr).bytes;
}

int tapclip_encode_image(tapclip_tower_t* t, const float* images, int32_t B, float* out, int32_t normalize,
                         void* workspace, size_t workspace_bytes, tapclip_stream_t stream) {
  if (!t || !images || !out || !workspace) return fail(TAPCLIP_EINVAL, "null argument");
  if (t->cfg.kind != TAPCLIP_TOWER_VISION) return fail(TAPCLIP_EINVAL, "encode_image needs a vision tower");
  if (B <= 0) return fail(TAPCLIP_EINVAL, "batch must be positive");
  int rc = check_ready(t);
  if (rc) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int N = t->tokens_vision, D = t->cfg.width, G2 = N - 1;
  const Workspace w = carve(t, B, N, workspace);
  if (w.bytes > workspace_bytes) return fail(TAPCLIP_EWORKSPACE, "workspace %zu B < required %zu B", workspace_bytes, w.bytes);
  {
    ProfScope ps(t, 0, s);
    // patch gather into the (not yet live) MLP-hidden buffer, then conv1-as-GEMM with the
    // positional-embedding add and the [b, 1+p] row placement fused into the epilogue
    HIP_TRY(launch_im2col(images, B, t->cfg.image_size, t->cfg.patch, t->Kp, w.h_hi, w.h_lo, s));
    GemmArgs g;
    g.A_hi = w.h_hi; g.A_lo = w.h_lo; g.lda = t->Kp;
    g.W_hi = t->conv.hi; g.W_lo = t->conv.lo;
    g.bias = nullptr;
    g.M = (int64_t)B * G2; g.N = D; g.K = t->Kp;
    g.out_hi = nullptr; g.out_lo = nullptr; g.out_f32 = w.x; g.ldo = D;
    g.add_table = t->pos; g.rows_per_group = G2; g.act = 0;
    HIP_TRY(launch_gemm(g, EPI_PATCH_F32, t->split, s));
    HIP_TRY(launch_class_token(t->cls, t->pos, B, N, D, w.x, s));
  }
  {
    ProfScope ps(t, 1, s);
    // (fp8: ln_pre writes the blocks' 16-bit residual stream; else it normalises the fp32 stream in place)
    if (t->fp8) HIP_TRY(launch_layernorm(w.x, D, t->lnpre_g, t->lnpre_b, (int64_t)B * N, D, w.x16, nullptr, nullptr, s));
    // x24: ln_pre, the write of the 24-bit residual planes and block 0's ln_1 in one pass over the patch embeddings
    else if (t->x24) HIP_TRY(launch_layernorm_x24(0, 1, w.x, D, w.x24_hi, w.x24_lo, nullptr, nullptr, t->lnpre_g, t->lnpre_b, t->layers[0].ln1_g, t->layers[0].ln1_b, (int64_t)B * N, D, w.xn_hi, s));
    else HIP_TRY(launch_layernorm(w.x, D, t->lnpre_g, t->lnpre_b, (int64_t)B * N, D, nullptr, nullptr, w.x, s));
  }
  // CLS-only last block (TAPCLIP_PRUNE_LAST=0 or tapclip_tower_set_flag(TAPCLIP_FLAG_PRUNE_LAST_BLOCK, 0) computes every
  // row of every block)
  const bool pooled = t->prune_last && t->cfg.layers >= 1;
  if (pooled) {
    // CLS rows of the output: fp32 [B, D].  With the 24-bit planes or the fp8 path's 16-bit stream the fp32 buffer w.x is
    // free; with an fp32 residual stream (w.x live) they go to the front of the pending-branch buffer w.d, which LN1 of
    // the last block has consumed.
    float* cls = (t->x24 || t->fp8) ? w.x : reinterpret_cast<float*>(w.d_hi);
    bf16_t *pd_hi = nullptr, *pd_lo = nullptr;
    rc = t->fp8 ? run_blocks_fp8(t, B, N, w, s, cls, &pd_hi, &pd_lo) : run_blocks(t, w.x, B, N, 0, w, nullptr, nullptr, s, false, cls, &pd_hi, &pd_lo);
    if (rc) return rc;
    ProfScope ps(t, 7, s);
    HIP_TRY(launch_pool_project(cls, pd_hi, pd_lo, B, 1, D, nullptr, 0, t->lnpost_g, t->lnpost_b, t->proj, t->cfg.embed_dim, normalize, out, s));
    return TAPCLIP_OK;
  }
  rc = t->fp8 ? run_blocks_fp8(t, B, N, w, s) : run_blocks(t, w.x, B, N, 0, w, nullptr, nullptr, s);
  if (rc) return rc;
  {
    ProfScope ps(t, 7, s);
    // the last c_proj branch is still pending: the pool kernel adds it to the CLS rows it gathers
    if (t->fp8 || t->x24) {
      // [B, D] fp32 at the front of the (now free) fp32 buffer
      if (t->fp8) HIP_TRY(launch_gather_cls16(w.x16, w.d_hi, B, N, D, w.x, s));
      else HIP_TRY(launch_gather_cls24(w.x24_hi, w.x24_lo, w.d_hi, B, N, D, w.x, s));
      HIP_TRY(launch_pool_project(w.x, nullptr, nullptr, B, 1, D, nullptr, 0, t->lnpost_g, t->lnpost_b, t->proj, t->cfg.embed_dim, normalize, out, s));
    } else {
      HIP_TRY(launch_pool_project(w.x, w.d_hi, w.d_lo, B, N, D, nullptr, 0, t->lnpost_g, t->lnpost_b, t->proj, t->cfg.embed_dim, normalize, out, s));
    }
  }
  return TAPCLIP_OK;
}

int tapclip_text_forward(tapclip_tower_t* t, const float* x_in, int32_t n_seq, int32_t tokens, int32_t causal,
                         float* out_hidden, float* attn_heads, float* attn_mean, float* attn_out, void* workspace,
                         size_t workspace_bytes, tapclip_stream_t stream) {
  if (!t || !x_in || !workspace) return fail(TAPCLIP_EINVAL, "null argument");
  if (t->cfg.kind != TAPCLIP_TOWER_TEXT) return fail(TAPCLIP_EINVAL, "text_forward needs a text tower");
  if (n_seq <= 0 || tokens <= 0) return fail(TAPCLIP_EINVAL, "bad n_seq/tokens (%d, %d)", n_seq, tokens);
  if (tokens > 256 && (attn_heads || attn_mean)) return fail(TAPCLIP_EINVAL, "attention write-back needs tokens <= 256 (got %d)", tokens);
  int rc = check_ready(t);
  if (rc) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int D = t->cfg.width;
  const Workspace w = carve(t, n_seq, tokens, workspace);
  if (w.bytes > workspace_bytes) return fail(TAPCLIP_EWORKSPACE, "workspace %zu B < required %zu B", workspace_bytes, w.bytes);
  float* x = out_hidden ? out_hidden : w.x;  // the residual stream lives in the caller's output
  if (x != x_in) HIP_TRY(hipMemcpyAsync(x, x_in, (size_t)n_seq * tokens * D * 4, hipMemcpyDeviceToDevice, s));
  float* probs = attn_heads ? attn_heads : (attn_mean ? w.probs : nullptr);
  rc = run_blocks(t, x, n_seq, tokens, causal, w, probs, attn_out, s, /*capture_only=*/out_hidden == nullptr);
  if (rc) return rc;
  if (out_hidden) HIP_TRY(launch_add_delta(x, w.d_hi, w.d_lo, (int64_t)n_seq * tokens * D, s));  // last pending branch
  if (attn_mean) HIP_TRY(launch_head_mean(probs, n_seq, t->cfg.heads, tokens, attn_mean, s));
  return TAPCLIP_OK;
}

size_t tapclip_text_backward_workspace_bytes(const tapclip_tower_t* t, int64_t n_seq, int32_t tokens) {
  if (!t || n_seq <= 0 || tokens <= 0 || t->cfg.kind != TAPCLIP_TOWER_TEXT) return 0;
  return align_up(carve(t, n_seq, tokens, nullptr).bytes) + carve_saved(t, n_seq * tokens, nullptr).bytes;
}

int tapclip_text_backward(tapclip_tower_t* t, const float* x_in, const float* grad_hidden, int32_t n_seq, int32_t tokens,
                          int32_t causal, float* grad_x, void* workspace, size_t workspace_bytes, tapclip_stream_t stream) {
  if (!t || !x_in || !grad_hidden || !grad_x || !workspace) return fail(TAPCLIP_EINVAL, "null argument");
  if (t->cfg.kind != TAPCLIP_TOWER_TEXT) return fail(TAPCLIP_EINVAL, "text_backward needs a text tower");
  if (n_seq <= 0 || tokens <= 0) return fail(TAPCLIP_EINVAL, "bad n_seq/tokens");
  if (attn_bwd_lds_bytes(tokens) > 160 * 1024) return fail(TAPCLIP_EINVAL, "text_backward supports at most 96 tokens per sequence (got %d)", tokens);
  int rc = check_ready(t);
  if (rc) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const Workspace w = carve(t, n_seq, tokens, workspace);
  const size_t off = align_up(w.bytes);
  const Saved sv = carve_saved(t, (int64_t)n_seq * tokens, static_cast<char*>(workspace) + off);
  if (off + sv.bytes > workspace_bytes) return fail(TAPCLIP_EWORKSPACE, "workspace %zu B < required %zu B", workspace_bytes, off + sv.bytes);
  if (grad_x != grad_hidden)
    HIP_TRY(hipMemcpyAsync(grad_x, grad_hidden, (size_t)n_seq * tokens * t->cfg.width * 4, hipMemcpyDeviceToDevice, s));
  return run_backward(t, x_in, grad_x, n_seq, tokens, causal, w, sv, s);
}

size_t tapclip_text_saved_bytes(const tapclip_tower_t* t, int64_t n_seq, int32_t tokens) {
  if (!t || n_seq <= 0 || tokens <= 0 || t->cfg.kind != TAPCLIP_TOWER_TEXT) return 0;
  return carve_saved(t, n_seq * tokens, nullptr).bytes;
}

int tapclip_text_forward_saved(tapclip_tower_t* t, const float* x_in, int32_t n_seq, int32_t tokens, int32_t causal,
                               float* out_hidden, void* saved, size_t saved_bytes, void* workspace, size_t workspace_bytes,
                               tapclip_stream_t stream) {
  if (!t || !x_in || !out_hidden || !saved || !workspace) return fail(TAPCLIP_EINVAL, "null argument");
  if (t->cfg.kind != TAPCLIP_TOWER_TEXT) return fail(TAPCLIP_EINVAL, "text_forward_saved needs a text tower");
  if (n_seq <= 0 || tokens <= 0) return fail(TAPCLIP_EINVAL, "bad n_seq/tokens");
  int rc = check_ready(t);
  if (rc) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t M = (int64_t)n_seq * tokens;
  const Workspace w = carve(t, n_seq, tokens, workspace);
  if (w.bytes > workspace_bytes) return fail(TAPCLIP_EWORKSPACE, "workspace %zu B < required %zu B", workspace_bytes, w.bytes);
  const Saved sv = carve_saved(t, M, saved);
  if (sv.bytes > saved_bytes) return fail(TAPCLIP_EWORKSPACE, "saved buffer %zu B < required %zu B", saved_bytes, sv.bytes);
  const float* x_last = nullptr;
  if ((rc = run_forward_saving(t, x_in, n_seq, tokens, causal, w, sv, s, &x_last))) return rc;
  HIP_TRY(hipMemcpyAsync(out_hidden, x_last, (size_t)M * t->cfg.width * 4, hipMemcpyDeviceToDevice, s));
  HIP_TRY(launch_add_delta(out_hidden, w.d_hi, w.d_lo, M * t->cfg.width, s));  // last pending branch
  return TAPCLIP_OK;
}

int tapclip_text_backward_saved(tapclip_tower_t* t, const void* saved, size_t saved_bytes, const float* grad_hidden, int32_t n_seq,
                                int32_t tokens, int32_t causal, float* grad_x, void* workspace, size_t workspace_bytes,
                                tapclip_stream_t stream) {
  if (!t || !saved || !grad_hidden || !grad_x || !workspace) return fail(TAPCLIP_EINVAL, "null argument");
  if (t->cfg.kind != TAPCLIP_TOWER_TEXT) return fail(TAPCLIP_EINVAL, "text_backward_saved needs a text tower");
  if (n_seq <= 0 || tokens <= 0) return fail(TAPCLIP_EINVAL, "bad n_seq/tokens");
  if (attn_bwd_lds_bytes(tokens) > 160 * 1024) return fail(TAPCLIP_EINVAL, "text_backward supports at most 96 tokens per sequence (got %d)", tokens);
  int rc = check_ready(t);
  if (rc) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t M = (int64_t)n_seq * tokens;
  const Workspace w = carve(t, n_seq, tokens, workspace);
  if (w.bytes > workspace_bytes) return fail(TAPCLIP_EWORKSPACE, "workspace %zu B < required %zu B", workspace_bytes, w.bytes);
  const Saved sv = carve_saved(t, M, const_cast<void*>(saved));
  if (sv.bytes > saved_bytes) return fail(TAPCLIP_EWORKSPACE, "saved buffer %zu B < required %zu B", saved_bytes, sv.bytes);
  if (grad_x != grad_hidden) HIP_TRY(hipMemcpyAsync(grad_x, grad_hidden, (size_t)M * t->cfg.width * 4, hipMemcpyDeviceToDevice, s));
  return run_backward_sweep(t, grad_x, n_seq, tokens, causal, w, sv, s);
}

// ---- tied padding rows (tied.hip): the same three text entry points on the DISTINCT rows of every sequence.
namespace {
struct Tied {
  int Tc = 0;         // rows per sequence the tower runs on: tokens - tail_run + 1
  float bias = 0.f;   // ln(tail_run): the last compact key counts tail_run times in every softmax
  Workspace w;
  float* extra = nullptr;  // fp32 [n * Tc, D] behind the carved workspace
  size_t bytes = 0;
};
Tied carve_tied(const tapclip_tower* t, int64_t n_seq, int tokens, int tail_run, void* base) {
  Tied td;
  td.Tc = tokens - tail_run + 1;
  td.bias = logf((float)tail_run);
  td.w = carve(t, n_seq, td.Tc, base);
  const size_t off = align_up(td.w.bytes);
  td.extra = base ? reinterpret_cast<float*>(static_cast<char*>(base) + off) : nullptr;
  td.bytes = off + align_up((size_t)n_seq * td.Tc * t->cfg.width * 4);
  return td;
}
int tied_begin(tapclip_tower* t, int n_seq, int tokens, int tail_run, hipStream_t s) {
  if (t->cfg.kind != TAPCLIP_TOWER_TEXT) return fail(TAPCLIP_EINVAL, "the tied-padding entry points need a text tower");
  if (n_seq <= 0 || tokens <= 0) return fail(TAPCLIP_EINVAL, "bad n_seq/tokens (%d, %d)", n_seq, tokens);
  if (tail_run < 1 || tail_run > tokens) return fail(TAPCLIP_EINVAL, "tail_run %d outside [1, tokens = %d]", tail_run, tokens);
  int rc = check_ready(t);
  if (rc) return rc;
  if (!t->tied_flag) {
    void* p = nullptr;
    if ((rc = dev_alloc(t, sizeof(int), &p))) return rc;
    t->tied_flag = static_cast<int*>(p);
    HIP_TRY(hipMemsetAsync(t->tied_flag, 0, sizeof(int), s));  // (sticky afterwards: see tapclip_text_tied_violations)
  }
  return TAPCLIP_OK;
}
}  // namespace

int tapclip_text_tail_run(const float* x, int32_t n_seq, int32_t tokens, int32_t width, int32_t* run_out, tapclip_stream_t stream) {
  if (!x || !run_out) return fail(TAPCLIP_EINVAL, "null argument");
  if (n_seq <= 0 || tokens <= 0 || width <= 0) return fail(TAPCLIP_EINVAL, "bad n_seq/tokens/width (%d, %d, %d)", n_seq, tokens, width);
  hipStream_t s = static_cast<hipStream_t>(stream);
  int* dev = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dev), sizeof(int)));
  int run = tokens;
  hipError_t e = hipMemcpyAsync(dev, &run, sizeof(int), hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = launch_tail_run(x, n_seq, tokens, width, dev, s);
  if (e == hipSuccess) e = hipMemcpyAsync(&run, dev, sizeof(int), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);  // off the hot path: once per token bank
  (void)hipFree(dev);
  if (e != hipSuccess) return fail(TAPCLIP_EHIP, "tapclip_text_tail_run failed: %s", hipGetErrorString(e));
  *run_out = run;
  return TAPCLIP_OK;
}

size_t tapclip_text_tied_workspace_bytes(const tapclip_tower_t* t, int64_t n_seq, int32_t tokens, int32_t tail_run) {
  if (!t || n_seq <= 0 || tokens <= 0 || tail_run < 1 || tail_run > tokens || t->cfg.kind != TAPCLIP_TOWER_TEXT) return 0;
  return carve_tied(t, n_seq, tokens, tail_run, nullptr).bytes;
}

int tapclip_text_tied_violations(tapclip_tower_t* t, int32_t* out, tapclip_stream_t stream) {
  if (!t || !out) return fail(TAPCLIP_EINVAL, "null argument");
  *out = 0;
  if (!t->tied_flag) return TAPCLIP_OK;  // no tied call yet
  hipStream_t s = static_cast<hipStream_t>(stream);
  int v = 0;
  HIP_TRY(hipMemcpyAsync(&v, t->tied_flag, sizeof(int), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemsetAsync(t->tied_flag, 0, sizeof(int), s));  // read and clear
  HIP_TRY(hipStreamSynchronize(s));
  *out = v;
  return TAPCLIP_OK;
}

int tapclip_text_forward_tied(tapclip_tower_t* t, const float* x_in, int32_t n_seq, int32_t tokens, int32_t tail_run, float* out_hidden,
                              float* attn_heads, float* attn_mean, float* attn_out, void* workspace, size_t workspace_bytes,
                              tapclip_stream_t stream) {
  if (!t || !x_in || !workspace) return fail(TAPCLIP_EINVAL, "null argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc = tied_begin(t, n_seq, tokens, tail_run, s);
  if (rc) return rc;
  const int D = t->cfg.width, H = t->cfg.heads;
  const Tied td = carve_tied(t, n_seq, tokens, tail_run, workspace);
  if (td.Tc > 256 && (attn_heads || attn_mean)) return fail(TAPCLIP_EINVAL, "attention write-back needs at most 256 distinct tokens (got %d)", td.Tc);
  if (td.bytes > workspace_bytes) return fail(TAPCLIP_EWORKSPACE, "workspace %zu B < required %zu B", workspace_bytes, td.bytes);
  const Workspace& w = td.w;
  HIP_TRY(launch_tied_compact(x_in, n_seq, tokens, td.Tc, D, w.x, t->tied_flag, s));
  float* probs = (attn_heads || attn_mean) ? w.probs : nullptr;
  rc = run_blocks(t, w.x, n_seq, td.Tc, 0, w, probs, attn_out ? td.extra : nullptr, s, /*capture_only=*/out_hidden == nullptr, nullptr, nullptr,
                  nullptr, td.bias);
  if (rc) return rc;
  if (out_hidden) {
    HIP_TRY(launch_add_delta(w.x, w.d_hi, w.d_lo, (int64_t)n_seq * td.Tc * D, s));  // last pending branch
    HIP_TRY(launch_tied_expand_rows(w.x, n_seq, tokens, td.Tc, D, 0, out_hidden, t->tied_flag, s));
  }
  if (attn_heads) HIP_TRY(launch_tied_expand_map(probs, n_seq * H, 1, tokens, td.Tc, attn_heads, t->tied_flag, s));
  if (attn_mean) HIP_TRY(launch_tied_expand_map(probs, n_seq, H, tokens, td.Tc, attn_mean, t->tied_flag, s));
  if (attn_out) HIP_TRY(launch_tied_expand_rows(td.extra, n_seq, tokens, td.Tc, D, 0, attn_out, t->tied_flag, s));
  return TAPCLIP_OK;
}

int tapclip_text_forward_saved_tied(tapclip_tower_t* t, const float* x_in, int32_t n_seq, int32_t tokens, int32_t tail_run,
                                    float* out_hidden, void* saved, size_t saved_bytes, void* workspace, size_t workspace_bytes,
                                    tapclip_stream_t stream) {
  if (!t || !x_in || !out_hidden || !saved || !workspace) return fail(TAPCLIP_EINVAL, "null argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc = tied_begin(t, n_seq, tokens, tail_run, s);
  if (rc) return rc;
  const int D = t->cfg.width;
  const Tied td = carve_tied(t, n_seq, tokens, tail_run, workspace);
  if (td.bytes > workspace_bytes) return fail(TAPCLIP_EWORKSPACE, "workspace %zu B < required %zu B", workspace_bytes, td.bytes);
  const int64_t Mc = (int64_t)n_seq * td.Tc;
  const Saved sv = carve_saved(t, Mc, saved);
  if (sv.bytes > saved_bytes) return fail(TAPCLIP_EWORKSPACE, "saved buffer %zu B < required %zu B", saved_bytes, sv.bytes);
  HIP_TRY(launch_tied_compact(x_in, n_seq, tokens, td.Tc, D, sv.x0[0], t->tied_flag, s));
  const float* x_last = nullptr;
  if ((rc = run_forward_saving(t, nullptr, n_seq, td.Tc, 0, td.w, sv, s, &x_last, td.bias))) return rc;
  HIP_TRY(hipMemcpyAsync(td.extra, x_last, (size_t)Mc * D * 4, hipMemcpyDeviceToDevice, s));
  HIP_TRY(launch_add_delta(td.extra, td.w.d_hi, td.w.d_lo, Mc * D, s));  // last pending branch
  HIP_TRY(launch_tied_expand_rows(td.extra, n_seq, tokens, td.Tc, D, 0, out_hidden, t->tied_flag, s));
  return TAPCLIP_OK;
}

int tapclip_text_backward_saved_tied(tapclip_tower_t* t, const void* saved, size_t saved_bytes, const float* grad_hidden, int32_t n_seq,
                                     int32_t tokens, int32_t tail_run, float* grad_x, void* workspace, size_t workspace_bytes,
                                     tapclip_stream_t stream) {
  if (!t || !saved || !grad_hidden || !grad_x || !workspace) return fail(TAPCLIP_EINVAL, "null argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc = tied_begin(t, n_seq, tokens, tail_run, s);
  if (rc) return rc;
  const int D = t->cfg.width;
  const Tied td = carve_tied(t, n_seq, tokens, tail_run, workspace);
  if (attn_bwd_lds_bytes(td.Tc) > 160 * 1024) return fail(TAPCLIP_EINVAL, "text backward supports at most 96 distinct tokens per sequence (got %d)", td.Tc);
  if (td.bytes > workspace_bytes) return fail(TAPCLIP_EWORKSPACE, "workspace %zu B < required %zu B", workspace_bytes, td.bytes);
  const int64_t Mc = (int64_t)n_seq * td.Tc;
  const Saved sv = carve_saved(t, Mc, const_cast<void*>(saved));
  if (sv.bytes > saved_bytes) return fail(TAPCLIP_EWORKSPACE, "saved buffer %zu B < required %zu B", saved_bytes, sv.bytes);
  // the tied rows are ONE variable: its output gradient is the sum over the rows it stands for, and its input gradient
  // (the sum of the per-row gradients) is returned in the group's first row, zeros in the others
  HIP_TRY(launch_tied_sum_tail(grad_hidden, n_seq, tokens, td.Tc, D, td.extra, s));
  if ((rc = run_backward_sweep(t, td.extra, n_seq, td.Tc, 0, td.w, sv, s, td.bias))) return rc;
  HIP_TRY(launch_tied_expand_rows(td.extra, n_seq, tokens, td.Tc, D, 1, grad_x, t->tied_flag, s));
  return TAPCLIP_OK;
}

int tapclip_text_pool_project_backward(tapclip_tower_t* t, const float* hidden, int32_t n_seq, int32_t tokens, int32_t normalize,
                                       const float* grad_out, float* grad_hidden, tapclip_stream_t stream) {
  if (!t || !hidden || !grad_out || !grad_hidden) return fail(TAPCLIP_EINVAL, "null argument");
  if (t->cfg.kind != TAPCLIP_TOWER_TEXT) return fail(TAPCLIP_EINVAL, "needs a text tower");
  int rc = check_ready(t);
  if (rc) return rc;
  HIP_TRY(launch_pool_project_bwd(hidden, n_seq, tokens, t->cfg.width, -1, t->text_proj, t->cfg.embed_dim, normalize, grad_out,
                                  grad_hidden, static_cast<hipStream_t>(stream)));
  return TAPCLIP_OK;
}

int tapclip_logits_backward(const float* grad_logits, const float* logits, const float* img, float scale, int32_t B, int32_t C,
                            int32_t E, float* grad_txt, float* grad_log_scale, tapclip_stream_t stream) {
  if (!grad_logits || !logits || !img || !grad_txt || B <= 0 || C <= 0 || E <= 0) return fail(TAPCLIP_EINVAL, "bad logits_backward arguments");
  HIP_TRY(launch_logits_bwd(grad_logits, logits, img, scale, B, C, E, grad_txt, grad_log_scale, static_cast<hipStream_t>(stream)));
  return TAPCLIP_OK;
}

int tapclip_text_pool_project(tapclip_tower_t* t, const float* hidden, int32_t n_seq, int32_t tokens, const int64_t* index,
                              int32_t apply_ln_final, int32_t normalize, float* out, tapclip_stream_t stream) {
  if (!t || !hidden || !out) return fail(TAPCLIP_EINVAL, "null argument");
  if (t->cfg.kind != TAPCLIP_TOWER_TEXT) return fail(TAPCLIP_EINVAL, "needs a text tower");
  int rc = check_ready(t);
  if (rc) return rc;
  HIP_TRY(launch_pool_project(hidden, nullptr, nullptr, n_seq, tokens, t->cfg.width, index, -1, apply_ln_final ? t->lnfin_g : nullptr,
                              apply_ln_final ? t->lnfin_b : nullptr, t->text_proj, t->cfg.embed_dim, normalize, out,
                              static_cast<hipStream_t>(stream)));
  return TAPCLIP_OK;
}

int tapclip_embed_tokens(tapclip_tower_t* t, const int64_t* tokens, int32_t n_seq, int32_t len, int32_t add_pos, float* out,
                         tapclip_stream_t stream) {
  if (!t || !tokens || !out) return fail(TAPCLIP_EINVAL, "null argument");
  if (t->cfg.kind != TAPCLIP_TOWER_TEXT) return fail(TAPCLIP_EINVAL, "needs a text tower");
  if (!t->tok_emb || !t->pos) return fail(TAPCLIP_ESTATE, "token_embedding / positional_embedding not loaded");
  if (add_pos && len > t->cfg.ctx_len) return fail(TAPCLIP_EINVAL, "len %d > ctx_len %d", len, t->cfg.ctx_len);
  if (n_seq <= 0 || len <= 0) return fail(TAPCLIP_EINVAL, "bad n_seq/len (%d, %d)", n_seq, len);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!t->bad_token) {
    void* p = nullptr;
    int rc = dev_alloc(t, sizeof(int), &p);
    if (rc) return rc;
    t->bad_token = static_cast<int*>(p);
  }
  HIP_TRY(hipMemsetAsync(t->bad_token, 0, sizeof(int), s));
  HIP_TRY(launch_embed_tokens(t->tok_emb, t->cfg.vocab, t->pos, tokens, n_seq, len, t->cfg.width, add_pos, out, t->bad_token, s));
  // this entry point is off the hot path (prompt construction, encode_text): it waits for the lookup so that an
  // out-of-range id is reported like torch's embedding would (an error), not clamped silently
  int bad = 0;
  HIP_TRY(hipMemcpyAsync(&bad, t->bad_token, sizeof(int), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (bad) return fail(TAPCLIP_EINVAL, "token id outside [0, %d) in tapclip_embed_tokens", t->cfg.vocab);
  return TAPCLIP_OK;
}

int tapclip_attribution(const float* attn_map, int32_t n, int32_t T, int32_t T2, int32_t P, int32_t normalize, float* out,
                        tapclip_stream_t stream) {
  if (!attn_map || !out || n <= 0 || T <= 0 || T2 < T || P <= 0) return fail(TAPCLIP_EINVAL, "bad attribution arguments");
  HIP_TRY(launch_attribution(attn_map, n, T, T2, P, normalize, out, static_cast<hipStream_t>(stream)));
  return TAPCLIP_OK;
}

int tapclip_build_prompts(const float* ctx, const float* tok, const float* attribution, int32_t attr_cols, int32_t n,
                          int32_t P, int32_t L, int32_t D, float* out, tapclip_stream_t stream) {
  if (!ctx || !tok || !out || n <= 0 || P <= 0 || L <= 0 || D <= 0) return fail(TAPCLIP_EINVAL, "bad build_prompts arguments");
  if (attribution && attr_cols != P && attr_cols != 1) return fail(TAPCLIP_EINVAL, "attribution has %d columns, expected %d or 1", attr_cols, P);
  HIP_TRY(launch_build_prompts(ctx, tok, attribution, attr_cols, n, P, L, D, out, static_cast<hipStream_t>(stream)));
  return TAPCLIP_OK;
}

int tapclip_build_prompts_backward(const float* d_out, const float* attribution, int32_t attr_cols, int32_t n, int32_t P, int32_t L,
                                   int32_t D, float* d_ctx, tapclip_stream_t stream) {
  if (!d_out || !d_ctx || n <= 0 || P <= 0 || L < 0 || D <= 0) return fail(TAPCLIP_EINVAL, "bad build_prompts_backward arguments");
  if (attribution && attr_cols != P && attr_cols != 1) return fail(TAPCLIP_EINVAL, "attribution must be [n,%d] or [n,1], got %d columns", P, attr_cols);
  HIP_TRY(launch_build_prompts_backward(d_out, attribution, attr_cols, n, P, L, D, d_ctx, static_cast<hipStream_t>(stream)));
  return TAPCLIP_OK;
}

int tapclip_build_prompts_mlp(int32_t method, const float* ctx, const float* tok, const float* attribution, int32_t attr_cols,
                              const float* w1, const float* b1, const float* w2, const float* b2, int32_t n, int32_t P, int32_t L, int32_t D,
                              float* out, tapclip_stream_t stream) {
  if (!ctx || !tok || !attribution || !w1 || !b1 || !w2 || !b2 || !out || n <= 0 || P <= 0 || L <= 0 || D <= 0)
    return fail(TAPCLIP_EINVAL, "bad build_prompts_mlp arguments");
  if (method != TAPCLIP_ADJUST_GATE && method != TAPCLIP_ADJUST_RESIDUAL) return fail(TAPCLIP_EINVAL, "unknown adjustor method %d", method);
  if (attr_cols != P && attr_cols != 1) return fail(TAPCLIP_EINVAL, "attribution has %d columns, expected %d or 1", attr_cols, P);
  HIP_TRY(launch_build_prompts_mlp(method, ctx, tok, attribution, attr_cols, w1, b1, w2, b2, n, P, L, D, out, static_cast<hipStream_t>(stream)));
  return TAPCLIP_OK;
}

int tapclip_logits(const float* img, const float* txt, float scale, int32_t B, int32_t C, int32_t E, float* out,
                   tapclip_stream_t stream) {
  if (!img || !txt || !out || B <= 0 || C <= 0 || E <= 0) return fail(TAPCLIP_EINVAL, "bad logits arguments");
  HIP_TRY(launch_logits(img, txt, scale, B, C, E, out, static_cast<hipStream_t>(stream)));
  return TAPCLIP_OK;
}

int tapclip_preprocess_u8(const uint8_t* pixels, const int64_t* desc, int32_t B, int32_t size, const float* mean_std,
                          void* workspace, float* out, tapclip_stream_t stream) {
  if (!pixels || !desc || !mean_std || !workspace || !out) return fail(TAPCLIP_EINVAL, "null argument");
  if (B <= 0 || B > 65535 || size <= 0 || size > 4096) return fail(TAPCLIP_EINVAL, "preprocess needs 0 < B <= 65535 and 0 < size <= 4096 (got %d, %d)", B, size);
  for (int i = 3; i < 6; ++i)
    if (!(mean_std[i] > 0.f)) return fail(TAPCLIP_EINVAL, "preprocess: std[%d] must be positive", i - 3);
  HIP_TRY(launch_preprocess_u8(pixels, desc, B, size, mean_std, static_cast<uint8_t*>(workspace), out, static_cast<hipStream_t>(stream)));
  return TAPCLIP_OK;
}

int tapclip_layernorm_f32(const float* x, const float* gamma, const float* beta, int64_t rows, int32_t d, float* y,
                          tapclip_stream_t stream) {
  if (!x || !gamma || !beta || !y) return fail(TAPCLIP_EINVAL, "null argument");
  if (rows <= 0 || d <= 0 || d % 64 != 0) return fail(TAPCLIP_EINVAL, "layernorm needs rows > 0 and d %% 64 == 0 (got %lld, %d)", (long long)rows, d);
  HIP_TRY(launch_layernorm(x, d, gamma, beta, rows, d, nullptr, nullptr, y, static_cast<hipStream_t>(stream)));
  return TAPCLIP_OK;
}

size_t tapclip_gemm_scratch_bytes(int64_t M, int32_t N, int32_t K) {
  return 2 * (align_up((size_t)M * K * 2) + align_up((size_t)N * K * 2));
}

int tapclip_gemm_f32(const float* A, const float* W, const float* bias, int64_t M, int32_t N, int32_t K, int32_t precision,
                     float* C, void* scratch, size_t scratch_bytes, tapclip_stream_t stream) {
  if (!A || !W || !C || !scratch) return fail(TAPCLIP_EINVAL, "null argument");
  if (M <= 0 || N <= 0 || N % 128 != 0 || K <= 0 || K % 64 != 0) return fail(TAPCLIP_EINVAL, "gemm needs N %% 128 == 0 and K %% 64 == 0 (M %lld N %d K %d)", (long long)M, N, K);
  if (scratch_bytes < tapclip_gemm_scratch_bytes(M, N, K)) return fail(TAPCLIP_EWORKSPACE, "gemm scratch too small");
  const bool split = precision == TAPCLIP_PREC_BF16X3;
  hipStream_t s = static_cast<hipStream_t>(stream);
  char* p = static_cast<char*>(scratch);
  if (precision == TAPCLIP_PREC_FP8) {  // quantise both operands to MXFP8, then the block-scaled MFMA GEMM
    if (N % 256 != 0 || K < 256) return fail(TAPCLIP_EINVAL, "fp8 gemm needs N %% 256 == 0 and K >= 256 (N %d K %d)", N, K);
    const int64_t m_pad = (M + 7) / 8 * 8;
    uint8_t* aq = reinterpret_cast<uint8_t*>(p); p += align_up((size_t)M * K);
    uint8_t* as = reinterpret_cast<uint8_t*>(p); p += align_up((size_t)(K / 64) * m_pad * 2);
    uint8_t* wq = reinterpret_cast<uint8_t*>(p); p += align_up((size_t)N * K);
    uint8_t* ws = reinterpret_cast<uint8_t*>(p);
    HIP_TRY(launch_quantize_mx8(A, M, K, K, 0, 1.f, aq, K, as, m_pad, s));
    HIP_TRY(launch_quantize_mx8(W, N, K, K, 0, 1.f, wq, K, ws, N, s));
    return tapclip_mx8_gemm(aq, as, M, m_pad, wq, ws, bias, N, K, 0, 0, C, nullptr, nullptr, stream);
  }
  bf16_t* a_hi = reinterpret_cast<bf16_t*>(p); p += align_up((size_t)M * K * 2);
  bf16_t* a_lo = reinterpret_cast<bf16_t*>(p); p += align_up((size_t)M * K * 2);
  bf16_t* w_hi = reinterpret_cast<bf16_t*>(p); p += align_up((size_t)N * K * 2);
  bf16_t* w_lo = reinterpret_cast<bf16_t*>(p);
  HIP_TRY(launch_pack(A, M, K, K, K, 0, 1.f, a_hi, split ? a_lo : nullptr, s));
  HIP_TRY(launch_pack(W, N, K, K, K, 0, 1.f, w_hi, split ? w_lo : nullptr, s));
  GemmArgs g;
  g.A_hi = a_hi; g.A_lo = a_lo; g.lda = K;
  g.W_hi = w_hi; g.W_lo = w_lo;
  g.bias = bias;
  g.M = M; g.N = N; g.K = K;
  g.out_hi = nullptr; g.out_lo = nullptr; g.out_f32 = C; g.ldo = N;
  g.add_table = nullptr; g.rows_per_group = 0; g.act = 0;
  HIP_TRY(launch_gemm(g, EPI_BIAS_F32, split, s));
  return TAPCLIP_OK;
}

int tapclip_mx8_quantize(const float* x, int64_t rows, int32_t K, uint8_t* q, uint8_t* scales, int64_t rows_pad,
                         tapclip_stream_t stream) {
  if (!x || !q || !scales) return fail(TAPCLIP_EINVAL, "null argument");
  if (rows <= 0 || K <= 0 || K % 64 != 0 || rows_pad < rows) return fail(TAPCLIP_EINVAL, "mx8_quantize needs K %% 64 == 0 and rows_pad >= rows (rows %lld K %d rows_pad %lld)", (long long)rows, K, (long long)rows_pad);
  HIP_TRY(launch_quantize_mx8(x, rows, K, K, 0, 1.f, q, K, scales, rows_pad, static_cast<hipStream_t>(stream)));
  return TAPCLIP_OK;
}

int tapclip_mx8_gemm(const uint8_t* a_q, const uint8_t* a_scale, int64_t M, int64_t m_pad, const uint8_t* w_q,
                     const uint8_t* w_scale, const float* bias, int32_t N, int32_t K, int32_t epilogue, int32_t act,
                     float* out_f32, uint8_t* out_q, uint8_t* out_q_scale, tapclip_stream_t stream) {
  if (!a_q || !a_scale || !w_q || !w_scale) return fail(TAPCLIP_EINVAL, "null argument");
  Mx8GemmArgs g;
  g.A = a_q; g.A_scale = a_scale; g.lda = K; g.m_pad = m_pad;
  g.W = w_q; g.W_scale = w_scale; g.bias = bias;
  g.M = M; g.N = N; g.K = K; g.act = act;
  int epi;
  if (epilogue == 0) {
    if (!out_f32) return fail(TAPCLIP_EINVAL, "epilogue 0 needs out_f32");
    g.out_f32 = out_f32; g.ldo = N; epi = EPI_BIAS_F32;
  } else if (epilogue == 1) {
    if (!out_q || !out_q_scale) return fail(TAPCLIP_EINVAL, "epilogue 1 needs out_q and out_q_scale");
    g.out_q = out_q; g.out_q_scale = out_q_scale; g.out_m_pad = m_pad; g.ldo = N; epi = EPI_BIAS_GELU_MX8;
  } else {
    return fail(TAPCLIP_EINVAL, "bad mx8 epilogue %d", epilogue);
  }
  if (!gemm_mx8_supports(g)) return fail(TAPCLIP_EINVAL, "mx8 gemm needs N %% 256 == 0, K %% 64 == 0, K >= 256, m_pad %% 8 == 0 (M %lld N %d K %d m_pad %lld)", (long long)M, N, K, (long long)m_pad);
  HIP_TRY(launch_gemm_mx8(g, epi, static_cast<hipStream_t>(stream)));
  return TAPCLIP_OK;
}


// ---- the one exchange step of the data-parallel path for hosts WITHOUT torch.distributed (SURVEY.md section 8b lists
// tapclip_allgather among the exports; the Python side keeps torch.distributed, tap-clip_amd/dist.py): a thin layer over
// RCCL.  librccl is opened at the first call (dlopen by its soname: inside a PyTorch process that is the copy
// torch already loaded), so the library itself carries no link-time dependency on it.
namespace {
struct Id128 {  // ncclUniqueId: 128 opaque bytes, passed BY VALUE to ncclCommInitRank
  char b[128];
};
struct RcclApi {
  void* handle = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, Id128, int) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*CommGetAsyncError)(void*, int*) = nullptr;  // optional (RCCL has had both since 2.4): without them tapclip_comm_check
  int (*CommAbort)(void*) = nullptr;                // reports what the enqueue status told, nothing more
};
RcclApi& rccl() {
  static RcclApi api = [] {
    RcclApi a;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      a.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (a.handle) break;
    }
    if (a.handle) {
      a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(a.handle, "ncclGetUniqueId"));
      a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(a.handle, "ncclCommInitRank"));
      a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(a.handle, "ncclAllGather"));
      a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(a.handle, "ncclCommDestroy"));
      a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(a.handle, "ncclGetErrorString"));
      a.CommGetAsyncError = reinterpret_cast<decltype(a.CommGetAsyncError)>(dlsym(a.handle, "ncclCommGetAsyncError"));
      a.CommAbort = reinterpret_cast<decltype(a.CommAbort)>(dlsym(a.handle, "ncclCommAbort"));
      if (!a.GetUniqueId || !a.CommInitRank || !a.AllGather || !a.CommDestroy) a.handle = nullptr;
    }
    return a;
  }();
  return api;
}
int rccl_fail(const char* what, int rc) {
  const RcclApi& a = rccl();
  return fail(TAPCLIP_EHIP, "%s failed: %s", what, a.GetErrorString ? a.GetErrorString(rc) : "RCCL error");
}
}  // namespace

struct tapclip_comm {
  void* nccl = nullptr;
  int rank = 0, world = 1;
};

int tapclip_comm_unique_id(void* id_out) {
  if (!id_out) return fail(TAPCLIP_EINVAL, "null argument");
  RcclApi& a = rccl();
  if (!a.handle) return fail(TAPCLIP_ESTATE, "librccl not found (dlopen of librccl.so.1 failed)");
  const int rc = a.GetUniqueId(id_out);
  return rc ? rccl_fail("ncclGetUniqueId", rc) : TAPCLIP_OK;
}

int tapclip_comm_create(const void* id, int32_t rank, int32_t world, tapclip_comm_t** out) {
  if (!id || !out) return fail(TAPCLIP_EINVAL, "null argument");
  if (world < 1 || rank < 0 || rank >= world) return fail(TAPCLIP_EINVAL, "bad rank / world (%d, %d)", rank, world);
  RcclApi& a = rccl();
  if (!a.handle) return fail(TAPCLIP_ESTATE, "librccl not found (dlopen of librccl.so.1 failed)");
  Id128 uid;
  memcpy(uid.b, id, sizeof(uid.b));
  tapclip_comm* c = new (std::nothrow) tapclip_comm;
  if (!c) return fail(TAPCLIP_ENOMEM, "out of host memory");
  c->rank = rank;
  c->world = world;
  const int rc = a.CommInitRank(&c->nccl, world, uid, rank);  // collective over the `world` ranks (current HIP device)
  if (rc) {
    delete c;
    return rccl_fail("ncclCommInitRank", rc);
  }
  *out = c;
  return TAPCLIP_OK;
}

// The asynchronous state of a communicator, without blocking ("RCCL async-error query after the
// all-gather").  ncclAllGather's own return value is the ENQUEUE status: a peer that died, a link error or a failed kernel shows
// only here -- and a stream that waits on such a collective never finishes by itself.  On an error the communicator is ABORTED
// (ncclCommAbort: its pending work is torn down so the caller's stream can drain) and is dead from then on.
int tapclip_comm_check(tapclip_comm_t* comm) {
  if (!comm) return fail(TAPCLIP_EINVAL, "null communicator");
  if (!comm->nccl) return fail(TAPCLIP_ESTATE, "communicator of rank %d / %d was aborted after an RCCL error", comm->rank, comm->world);
  const RcclApi& a = rccl();
  if (!a.CommGetAsyncError) return TAPCLIP_OK;  // (nothing to ask)
  int async = 0;
  const int rc = a.CommGetAsyncError(comm->nccl, &async);
  if (rc) return rccl_fail("ncclCommGetAsyncError", rc);
  constexpr int kNcclInProgress = 7;  // ncclInProgress: a non-blocking communicator still working -- not an error
  if (async == 0 || async == kNcclInProgress) return TAPCLIP_OK;
  const char* what = a.GetErrorString ? a.GetErrorString(async) : "RCCL error";
  if (a.CommAbort) (void)a.CommAbort(comm->nccl);
  comm->nccl = nullptr;
  return fail(TAPCLIP_EHIP, "RCCL reported an asynchronous error on rank %d of %d: %s (communicator aborted; destroy it and rebuild the group)",
              comm->rank, comm->world, what);
}

int tapclip_allgather(tapclip_comm_t* comm, const void* send, void* recv, size_t bytes_per_rank, tapclip_stream_t stream) {
  if (!comm || !send || !recv) return fail(TAPCLIP_EINVAL, "null argument");
  if (const int st = tapclip_comm_check(comm)) return st;  // an error of an EARLIER collective: do not queue behind it
  if (bytes_per_rank == 0) return TAPCLIP_OK;
  const int rc = rccl().AllGather(send, recv, bytes_per_rank, /* ncclChar */ 0, comm->nccl, static_cast<hipStream_t>(stream));
  return rc ? rccl_fail("ncclAllGather", rc) : TAPCLIP_OK;
}

void tapclip_comm_destroy(tapclip_comm_t* comm) {
  if (!comm) return;
  if (comm->nccl) (void)rccl().CommDestroy(comm->nccl);  // (an aborted communicator was released by ncclCommAbort)
  delete comm;
}

int tapclip_tower_set_flag(tapclip_tower_t* t, int32_t flag, int32_t value) {
  if (!t) return fail(TAPCLIP_EINVAL, "null tower");
  switch (flag) {
    case TAPCLIP_FLAG_PRUNE_LAST_BLOCK:
      // fp8 towers keep 16-bit copies of the last block's weights for the pooled rows; a tower created with
      // TAPCLIP_PRUNE_LAST=0 in the environment did not make them
      if (value != 0 && t->fp8 && !t->layers.empty() && t->layers.back().wqkv.hi == nullptr && !t->loaded.empty())
        return fail(TAPCLIP_ESTATE, "this fp8 tower was created with TAPCLIP_PRUNE_LAST=0: it holds no 16-bit copy of the last block for the CLS-only path");
      t->prune_last = value != 0;
      return TAPCLIP_OK;
    case TAPCLIP_FLAG_KSPLIT: t->ksplit = value != 0; return TAPCLIP_OK;
    default: return fail(TAPCLIP_EINVAL, "unknown tower flag %d", flag);
  }
}

int tapclip_tower_get_flag(const tapclip_tower_t* t, int32_t flag, int32_t* value) {
  if (!t || !value) return fail(TAPCLIP_EINVAL, "null argument");
  switch (flag) {
    case TAPCLIP_FLAG_PRUNE_LAST_BLOCK: *value = t->prune_last ? 1 : 0; return TAPCLIP_OK;
    case TAPCLIP_FLAG_KSPLIT: *value = t->ksplit ? 1 : 0; return TAPCLIP_OK;
    default: return fail(TAPCLIP_EINVAL, "unknown tower flag %d", flag);
  }
}

int tapclip_profile_enable(tapclip_tower_t* t, int32_t on) {
  if (!t) return fail(TAPCLIP_EINVAL, "null tower");
  t->prof_on = on != 0;
  return TAPCLIP_OK;
}

int tapclip_profile_read(tapclip_tower_t* t, float* ms_out, int64_t* launches_out) {
  if (!t || !ms_out || !launches_out) return fail(TAPCLIP_EINVAL, "null argument");
  for (size_t i = 0; i < t->prof_used; ++i) {
    ProfRec& r = t->prof[i];
    HIP_TRY(hipEventSynchronize(r.stop));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, r.start, r.stop));
    t->prof_ms[r.slot] += ms;
    t->prof_n[r.slot] += 1;
  }
  t->prof_used = 0;
  for (int i = 0; i < TAPCLIP_PROFILE_SLOTS; ++i) {
    ms_out[i] = (float)t->prof_ms[i];
    launches_out[i] = t->prof_n[i];
    t->prof_ms[i] = 0;
    t->prof_n[i] = 0;
  }
  return TAPCLIP_OK;
}

}  // extern "C"
