// Host-callable launchers of the gfx950 kernels (internal; the public surface is include/tapclip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tapclip {

typedef uint16_t bf16_t;

enum Epilogue {
  EPI_BIAS_BF16 = 0,       // out_hi(/lo)[m,n] = bf16(acc + bias[n])                      (QKV, K3)
  EPI_BIAS_GELU_BF16 = 1,  // out_hi(/lo)[m,n] = bf16(act(acc + bias[n]))                 (c_fc, K6)
  EPI_BIAS_RESID_F32 = 2,  // out_f32[m,n] += acc + bias[n]   (in-place residual add)     (out_proj K5, c_proj K7)
  EPI_PATCH_F32 = 3,       // out_f32[b*(G2+1)+1+p, n] = acc + add_table[(1+p), n]        (patch embed, K1)
  EPI_BIAS_F32 = 4,        // out_f32[m,n] = acc + bias[n]   (unit API / literal hook attn_out)
  EPI_GELU_BWD_BF16 = 5,   // out_hi(/lo)[m,n] = bf16(act'(acc + bias[n]) * aux[m,n])   (backward of c_fc's GELU)
  EPI_BIAS_GELU_MX8 = 6,   // gemm_mx8.hip only: out_q/out_q_scale = MXFP8(act(acc + bias[n]))          (c_fc of the fp8 path)
};

struct GemmArgs {
  const bf16_t* A_hi;  // [M, K] activations, row stride lda
  const bf16_t* A_lo;  // bf16x3 only
  int64_t lda;
  const bf16_t* W_hi;  // [N, K] weights (nn.Linear layout), row stride K
  const bf16_t* W_lo;
  const float* bias;   // [N] or nullptr
  int64_t M;
  int32_t N, K;
  bf16_t* out_hi;
  bf16_t* out_lo;
  float* out_f32;
  int64_t ldo;
  const float* add_table;  // EPI_PATCH_F32: positional embedding [(G2+1), N]
  int32_t rows_per_group;  // EPI_PATCH_F32: G2 patches per image
  int32_t act;             // TAPCLIP_ACT_*
  const bf16_t* aux_hi = nullptr;  // EPI_GELU_BWD_BF16: upstream gradient dL/dh [M, N] (row stride ldo); may alias out
  const bf16_t* aux_lo = nullptr;
  int32_t group_m = 8;             // m-tiles per group of the grouped tile order (gemm256.hip)
  // gemm256.hip tail split: tiles >= split_from are each computed by split_parts workgroups over 1/split_parts
  // of K as raw fp32 partial tiles [256][BN] in split_ws; a fix-up kernel sums them and applies the epilogue
  float* split_ws = nullptr;       // caller-provided scratch, >= gemm256_split_ws_bytes()
  int32_t split_from = 0, split_parts = 0;  // filled by the launcher
  int32_t n_cu = 0;  // gemm256.hip: size the persistent grid for this many CUs (0 = the whole device); for launches on a CU-masked stream
};
size_t gemm256_split_ws_bytes();

hipError_t launch_gemm(const GemmArgs& a, int epilogue, bool split, hipStream_t s);
// gemm_skinny.hip: a few hundred rows (M <= 1024), bf16 outputs (EPI_BIAS_BF16 / EPI_BIAS_GELU_BF16): (N / 64) x SPLITK
// workgroups of all rows x 64 columns x one K slice, fp32 partial slabs in `ws`, summed in a fixed order by a finalize kernel
bool gemm_skinny_supports(const GemmArgs& a, int epilogue);
size_t gemm_skinny_ws_bytes(int64_t M, int32_t N, int32_t K);
hipError_t launch_gemm_skinny(const GemmArgs& a, int epilogue, bool split, float* ws, size_t ws_bytes, hipStream_t s);

// ---- MX-fp8 GEMM (gemm_mx8.hip): e4m3 elements, one e8m0 scale (2^(s-127)) per 32 consecutive k.
// Scale arrays are k-step major: scale of (row, 32-block b) at [(b >> 1) * rows_pad + row] * 2 + (b & 1).
struct Mx8GemmArgs {
  const uint8_t* A;        // [M, K] e4m3, row stride lda bytes (multiple of 16)
  const uint8_t* A_scale;  // [K/64][m_pad][2]
  int64_t lda, m_pad;      // m_pad: multiple of 8, >= M
  const uint8_t* W;        // [N, K] e4m3, row stride K
  const uint8_t* W_scale;  // [K/64][N][2]  (a row range of a larger weight: [K/64][w_scale_rows][2], see w_scale_rows)
  int32_t w_scale_rows = 0; // row count of the scale plane W_scale points into (0 = N): lets W / W_scale address rows r0 .. r0 + N - 1 of a taller matrix
  const float* bias;       // [N] or nullptr
  int64_t M;
  int32_t N, K;
  bf16_t* out_bf16 = nullptr;  // EPI_BIAS_BF16: [M, N] row stride ldo
  float* out_f32 = nullptr;    // EPI_BIAS_F32 (unit API)
  uint8_t* out_q = nullptr;        // EPI_BIAS_GELU_MX8: e4m3 [M, N], row stride ldo bytes
  uint8_t* out_q_scale = nullptr;  //   its scales [N/64][out_m_pad][2]
  int64_t out_m_pad = 0;
  int64_t ldo;
  int32_t act = 0;
  int32_t group_m = 8;
  unsigned long long* stamps = nullptr;  // diagnostic build (-DMX8_STAMP) only
};
// fp32 [rows, K] (row stride ldx; the first scale_rows rows multiplied by scale) -> e4m3 [rows, ldq] + scales [K/64][rows_pad][2]
hipError_t launch_quantize_mx8(const float* x, int64_t rows, int32_t K, int64_t ldx, int64_t scale_rows, float scale, uint8_t* q,
                               int64_t ldq, uint8_t* sc, int64_t rows_pad, hipStream_t s);
bool gemm_mx8_supports(const Mx8GemmArgs& a);
hipError_t launch_gemm_mx8(const Mx8GemmArgs& a, int epilogue, hipStream_t s);

// LayerNorm over rows; one wave per row.  out_hi/out_lo (bf16) or out_f32; x may alias out_f32.
hipError_t launch_layernorm(const float* x, int64_t ldx, const float* gamma, const float* beta,
                            int64_t rows, int32_t d, bf16_t* out_hi, bf16_t* out_lo, float* out_f32,
                            hipStream_t s, bool stream_x = false);

// x += delta (bf16 hi [+ lo]) written back in fp32, then LayerNorm(x) -> bf16 hi [+ lo].  x_wb != nullptr: the updated
// rows go to x_wb ([rows, d]) instead and x is left as it was
hipError_t launch_add_layernorm(float* x, const bf16_t* delta_hi, const bf16_t* delta_lo, const float* gamma,
                                const float* beta, int64_t rows, int32_t d, bf16_t* out_hi, bf16_t* out_lo,
                                hipStream_t s, float* x_wb = nullptr);
// add: 1 = x += d1 (written back); 2 = LayerNorm(x + d1), x NOT written back; 3 = x += d1 + d2 (written back)
hipError_t launch_add_layernorm_ex(int add, float* x, const bf16_t* d1_hi, const bf16_t* d1_lo, const bf16_t* d2_hi,
                                   const bf16_t* d2_lo, const float* gamma, const float* beta, int64_t rows, int32_t d,
                                   bf16_t* out_hi, bf16_t* out_lo, hipStream_t s, bool stream_x = false);
// LayerNorm with MXFP8 output: out_q [rows, d] e4m3 + out_sc [d/64][rows_pad][2]; add = 0 (none) or as above.
// x16 != nullptr: the residual stream is 16-bit ([rows, d]) and x is ignored.
hipError_t launch_layernorm_mx8(int add, float* x, bf16_t* x16, const bf16_t* d1_hi, const bf16_t* d2_hi, const float* gamma,
                                const float* beta, int64_t rows, int32_t d, uint8_t* out_q, uint8_t* out_sc, int64_t rows_pad, hipStream_t s);
// out[b, :] = x16[b * tokens, :] + delta[b * tokens, :]  (the CLS rows of a 16-bit residual stream, as fp32 [B, D])
// 24-bit residual stream of the image tower's 16-bit modes (layernorm.hip XF = 2): planes xhi [rows, d] u16 + xlo [rows, d] u8
bool layernorm_x24_supports(int32_t d);
hipError_t launch_layernorm_x24(int add, int pre, const float* src_f32, int64_t ld_src, bf16_t* xhi, uint8_t* xlo, const bf16_t* d1,
                                const bf16_t* d2, const float* gamma_pre, const float* beta_pre, const float* gamma, const float* beta,
                                int64_t rows, int32_t d, bf16_t* out, hipStream_t s);
hipError_t launch_gather_cls24(const bf16_t* xhi, const uint8_t* xlo, const bf16_t* delta, int32_t B, int32_t tokens, int32_t D, float* out,
                               hipStream_t s);
hipError_t launch_gather_cls16(const bf16_t* x16, const bf16_t* delta, int32_t B, int32_t tokens, int32_t D, float* out, hipStream_t s);
// Resize(size, bicubic) + CenterCrop(size) + ToTensor + Normalize of B packed uint8 RGB images (preprocess.hip).
// desc [B][4] int64 (device): byte offset of the image in `pixels`, height, width, byte offset of its height*size*3
// scratch bytes in `ws`.  mean_std: host, mean[3] then std[3].
hipError_t launch_preprocess_u8(const uint8_t* pixels, const int64_t* desc, int32_t B, int32_t size, const float* mean_std,
                                uint8_t* ws, float* out, hipStream_t s);
// x[i] += delta_hi[i] (+ delta_lo[i])
hipError_t launch_add_delta(float* x, const bf16_t* delta_hi, const bf16_t* delta_lo, int64_t n, hipStream_t s);

struct AttnArgs {
  const bf16_t* qkv_hi;  // [n*T, 3D]: q | k | v, head h at columns h*64 .. h*64+63 of each third
  const bf16_t* qkv_lo;
  bf16_t* out_hi;        // [n*T, D]
  bf16_t* out_lo;
  float* probs;          // nullable [n, H, T, T] fp32 softmax probabilities
  int32_t n_seq, T, H, D;
  int32_t causal;
  // tied padding (tied.hip): ln(m) added to the score of the LAST key of every sequence, i.e. that key counts m times in
  // every softmax (0 = an ordinary key).  Not with causal, not in the flash kernel (T > 256).
  float last_key_bias = 0.f;
  // q (and only q) carries log2(e) besides 1/sqrt(64): folded into Wq, bq at pack time by towers whose attention runs in
  // attention_long.hip (image towers of more than 256 tokens, 16-bit operands).  Only that kernel takes it.
  int32_t q_log2 = 0;
  // fp8 path: when out_q is set the output leaves as MXFP8 (e4m3 [n*T, D] + scales [D/64][out_m_pad][2]; a head's
  // 64 columns are one k-step of the out_proj GEMM) instead of bf16
  uint8_t* out_q = nullptr;
  uint8_t* out_q_scale = nullptr;
  int64_t out_m_pad = 0;
  unsigned long long* stamps = nullptr;  // diagnostic builds only (-DATTN_STAMP, tools/): [workgroup][8] s_memrealtime stamps
};
hipError_t launch_attention(const AttnArgs& a, bool split, hipStream_t s);
// attention_long.hip: T > 256, 16-bit operands, no mask (launch_attention routes there)
hipError_t launch_flash2(const AttnArgs& a, hipStream_t s);
int flash2_cfg();                // TAPCLIP_FLASH2_CFG: 0 default, 1 the first flash kernel, WAVES * 10 + QT a geometry of the second
void flash2_set_cfg(int cfg);    // tools only
// softmax(q K^T) V for ONE query row per sequence (the pooled token of a tower's last block): q [n_seq, D] (hi [+ lo]),
// k / v taken from the q|k|v buffer [n_seq * T, 3 D] at columns D + 64 h / 2 D + 64 h, out [n_seq, D]
hipError_t launch_attention_pooled(const bf16_t* q_hi, const bf16_t* q_lo, const bf16_t* qkv_hi, const bf16_t* qkv_lo, bf16_t* out_hi,
                                   bf16_t* out_lo, int32_t n_seq, int32_t T, int32_t H, int32_t D, bool split, hipStream_t s, bool q_log2 = false);

// probs [n,H,T,T] -> mean over H -> [n,T,T]
hipError_t launch_head_mean(const float* probs, int32_t n, int32_t H, int32_t T, float* out, hipStream_t s);

// images [B,3,S,S] fp32 -> patches [B*G*G, Kp] bf16 (Kp = 3*p*p rounded up to 64, zero padded)
hipError_t launch_im2col(const float* img, int32_t B, int32_t S, int32_t p, int32_t Kp, bf16_t* hi,
                         bf16_t* lo, hipStream_t s);
// x[b, 0, :] = class_embedding + pos[0]  (row stride: tokens*D)
hipError_t launch_class_token(const float* cls, const float* pos, int32_t B, int32_t tokens, int32_t D,
                              float* x, hipStream_t s);

// fp32 -> bf16 hi (+lo); the first scale_rows rows (of row length cols) are multiplied by scale first.
// src row stride = src_ld, dst row stride = dst_ld (>= cols, padding zero-filled).
hipError_t launch_pack(const float* src, int64_t rows, int32_t cols, int64_t src_ld, int32_t dst_ld,
                       int64_t scale_rows, float scale, bf16_t* hi, bf16_t* lo, hipStream_t s);
hipError_t launch_scale_copy(const float* src, int64_t n, int64_t scale_n, float scale, float* dst, hipStream_t s);

// out[i,:] = normalize?( LN?(src[i, idx_i, :]) @ proj[K, E] )
// delta_hi/lo (nullable): pending residual branch [n*tokens, K] added to the gathered row first
hipError_t launch_pool_project(const float* src, const bf16_t* delta_hi, const bf16_t* delta_lo, int64_t n,
                               int32_t tokens, int32_t K, const int64_t* index, int32_t fixed_token,
                               const float* ln_g, const float* ln_b, const float* proj, int32_t E,
                               int32_t normalize, float* out, hipStream_t s);

// ---- backward (prompt tuning): see backward.hip
struct AttnBwdArgs {
  const bf16_t* qkv_hi;   // saved forward q|k|v [n*T, 3D]
  const bf16_t* qkv_lo;
  const bf16_t* out_hi;   // saved forward attention output [n*T, D]
  const bf16_t* out_lo;
  const bf16_t* dout_hi;  // dL/d(attention output) [n*T, D]
  const bf16_t* dout_lo;
  bf16_t* dqkv_hi;        // [n*T, 3D]
  bf16_t* dqkv_lo;
  int32_t n_seq, T, H, D, causal;
  float last_key_bias = 0.f;  // as AttnArgs.last_key_bias: the forward's softmax is recomputed with it
};
size_t attn_bwd_lds_bytes(int T);
hipError_t launch_attention_bwd(const AttnBwdArgs& a, hipStream_t s);
hipError_t launch_ln_bwd(const float* x, const float* gamma, const float* dy, int64_t rows, int32_t d, float* dres,
                         bf16_t* pk_hi, bf16_t* pk_lo, hipStream_t s);  // pk_*: optional 16-bit planes of the updated dres
hipError_t launch_pool_project_bwd(const float* hidden, int64_t n, int32_t tokens, int32_t K, int32_t tok,
                                   const float* proj, int32_t E, int32_t normalize, const float* dt, float* d_hidden,
                                   hipStream_t s);
hipError_t launch_logits_bwd(const float* dl, const float* logits, const float* img, float scale, int32_t B, int32_t C,
                             int32_t E, float* d_txt, float* d_logscale, hipStream_t s);
hipError_t launch_pack_transpose(const float* src, int64_t N, int32_t K, int64_t scale_rows, float scale, bf16_t* hi,
                                 bf16_t* lo, hipStream_t s);

// ---- tied padding rows (tied.hip): the last `run` rows of every sequence of x [n, T, D] are one row repeated.
// Tc = T - run + 1 below; `flag` is a device int the kernels raise when a row of the run differs (bitwise) from the others,
// and every expand kernel writes NaN instead of results while it is raised.
hipError_t launch_tail_run(const float* x, int32_t n, int32_t T, int32_t D, int32_t* run_min_dev, hipStream_t s);  // *run_min_dev = min over sequences (preset to T by the caller)
hipError_t launch_tied_compact(const float* x, int32_t n, int32_t T, int32_t Tc, int32_t D, float* xc, int* flag, hipStream_t s);
hipError_t launch_tied_sum_tail(const float* g, int32_t n, int32_t T, int32_t Tc, int32_t D, float* gc, hipStream_t s);
hipError_t launch_tied_expand_rows(const float* hc, int32_t n, int32_t T, int32_t Tc, int32_t D, int32_t zero_tail, float* h, const int* flag, hipStream_t s);
// pc [nb, Hm, Tc, Tc] -> out [nb, T, T]: mean over Hm, rows / columns >= Tc - 1 replicate the last compact row / column, the
// replicated columns divided by run
hipError_t launch_tied_expand_map(const float* pc, int32_t nb, int32_t Hm, int32_t T, int32_t Tc, float* out, const int* flag, hipStream_t s);

hipError_t launch_embed_tokens(const float* table, int32_t vocab, const float* pos, const int64_t* tokens,
                               int32_t n, int32_t L, int32_t D, int32_t add_pos, float* out, int* bad_flag, hipStream_t s);
hipError_t launch_attribution(const float* amap, int32_t n, int32_t T, int32_t T2, int32_t P, int32_t normalize,
                              float* out, hipStream_t s);
hipError_t launch_build_prompts(const float* ctx, const float* tok, const float* attr, int32_t attr_cols,
                                int32_t n, int32_t P, int32_t L, int32_t D, float* out, hipStream_t s);
hipError_t launch_build_prompts_backward(const float* d_out, const float* attr, int32_t attr_cols, int32_t n, int32_t P, int32_t L,
                                         int32_t D, float* d_ctx, hipStream_t s);
// PromptAdjustor 'gate' (method 1: w2 [64], b2 [1]) / 'residual' (method 2: w2 [D, 64], b2 [D]) + the concatenations
hipError_t launch_build_prompts_mlp(int method, const float* ctx, const float* tok, const float* attr, int32_t attr_cols, const float* w1,
                                    const float* b1, const float* w2, const float* b2, int32_t n, int32_t P, int32_t L, int32_t D, float* out,
                                    hipStream_t s);
hipError_t launch_logits(const float* img, const float* txt, float scale, int32_t B, int32_t C, int32_t E,
                         float* out, hipStream_t s);

}  // namespace tapclip
