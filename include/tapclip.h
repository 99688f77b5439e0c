/*
 * tapclip.h -- C ABI of the MI355X (gfx950) CLIP dual-encoder hot path.
 *
 * This is the drop-in boundary for the ONE path of 3300786/TAP-CLIP that this
 * project accelerates: `FullModel.forward` (reference models/model_wrapper.py:28-100)
 * -> `CLIPWrapper.encode_image` / `clip.model.transformer` (reference
 * models/clip_wrapper.py:46-51, models/model_wrapper.py:58,72) -> the open_clip
 * towers.  The reference is pure Python on top of `open_clip`; the binding a
 * maintainer adds is a ctypes stub (shown in INTEGRATION.md, shipped in
 * tap-clip_amd/_lib.py).  Every entry point below names the reference
 * interface it replaces.
 *
 * Conventions
 *  - plain C: raw DEVICE pointers, sizes and a HIP stream; no torch types.
 *  - all user-visible tensors are caller-owned, fp32, contiguous, row-major.
 *    The library never allocates or frees them.  Weights are copied+packed
 *    (bf16 hi/lo) into memory owned by the opaque tower handle; scratch is a
 *    caller-provided workspace (size from tapclip_tower_workspace_bytes).
 *  - every call is asynchronous w.r.t. the host and ordered on `stream`
 *    (pass torch's current stream).  No internal host threads.  A handle is
 *    NOT thread-safe (the reference's CLIPWrapper is not re-entrant either:
 *    `attention_maps` is shared mutable state, clip_wrapper.py:23,42-44).
 *  - return value: 0 = TAPCLIP_OK, negative = error; the message is in a
 *    thread-local buffer read by tapclip_last_error().  No exception crosses
 *    this boundary (the Python shim raises RuntimeError / ValueError).
 */
#ifndef TAPCLIP_H
#define TAPCLIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TAPCLIP_ABI_VERSION 1

enum {
  TAPCLIP_OK = 0,
  TAPCLIP_EINVAL = -1,       /* bad argument / unsupported shape          */
  TAPCLIP_ENOMEM = -2,       /* device allocation for packed weights failed */
  TAPCLIP_EHIP = -3,         /* a HIP runtime call or launch failed       */
  TAPCLIP_ESTATE = -4,       /* weights missing / handle not ready        */
  TAPCLIP_EWORKSPACE = -5    /* workspace too small                       */
};

enum { TAPCLIP_TOWER_VISION = 0, TAPCLIP_TOWER_TEXT = 1 };
enum { TAPCLIP_ACT_GELU_ERF = 0, TAPCLIP_ACT_QUICK_GELU = 1 };
/* arithmetic of the MFMA GEMMs / attention:
 *   BF16   : operands rounded to bf16, fp32 accumulate (fast path, benchmarked)
 *   BF16X3 : each operand split hi+lo bf16, 3 MFMA products (a_hi*b_hi + a_lo*b_hi
 *            + a_hi*b_lo), ~2^-16 relative: the parity mode vs the fp32 reference */
enum {
  TAPCLIP_PREC_BF16 = 0,   /* bf16 MFMA operands, fp32 accumulate (fast path)                         */
  TAPCLIP_PREC_BF16X3 = 1, /* split-bf16 (hi + lo), three MFMA products: the 1e-3 parity mode         */
  TAPCLIP_PREC_FP8 = 2     /* image tower only: block GEMMs on MXFP8 (OCP e4m3 + e8m0 scale per 32 k)
                              MFMA, everything else as TAPCLIP_PREC_BF16 (BASELINE.json configs[4])   */
};

typedef struct tapclip_tower tapclip_tower_t; /* opaque */
typedef void* tapclip_stream_t;               /* hipStream_t */

typedef struct tapclip_tower_cfg {
  int32_t kind;       /* TAPCLIP_TOWER_*                                       */
  int32_t width;      /* d: 768 (ViT-B), 512 (text B), multiple of 128         */
  int32_t layers;
  int32_t heads;      /* width / heads must be 64                               */
  int32_t mlp_dim;    /* 4 * width                                             */
  int32_t embed_dim;  /* E: output projection width (512)                      */
  int32_t image_size; /* vision: 224                                           */
  int32_t patch;      /* vision: 16 / 32                                       */
  int32_t ctx_len;    /* text: positional table length (77)                    */
  int32_t vocab;      /* text: token-embedding rows (49408)                    */
  int32_t act;        /* TAPCLIP_ACT_*                                         */
  int32_t precision;  /* TAPCLIP_PREC_*                                        */
} tapclip_tower_cfg;

/* ---- handle lifetime: replaces open_clip.create_model_and_transforms +
 * load_state_dict + .to(device).eval() in CLIPWrapper.__init__
 * (reference models/clip_wrapper.py:10-20). ---------------------------------- */
int tapclip_tower_create(const tapclip_tower_cfg* cfg, tapclip_tower_t** out);
void tapclip_tower_destroy(tapclip_tower_t* tower);

/* Copy + pack one state-dict tensor (fp32, device memory, open_clip key name with
 * the tower prefix stripped: "conv1.weight", "resblocks.3.attn.in_proj_weight",
 * "ln_post.weight", "proj", "text_projection", "token_embedding.weight", ...).
 * Replaces `self.model.load_state_dict(state_dict, strict=True)`
 * (reference models/clip_wrapper.py:14-15).  Unknown keys -> TAPCLIP_EINVAL. */
int tapclip_tower_load_weight(tapclip_tower_t* tower, const char* key, const float* dev_ptr,
                              const int64_t* shape, int32_t ndim, tapclip_stream_t stream);
/* 0 when every required tensor has been loaded (strict=True semantics). */
int tapclip_tower_ready(const tapclip_tower_t* tower);

/* Scratch bytes needed for `rows` = n_sequences * tokens_per_sequence rows. */
size_t tapclip_tower_workspace_bytes(const tapclip_tower_t* tower, int64_t n_seq, int32_t tokens);

/* ---- `CLIPWrapper.encode_image(image_tensor)` (reference models/clip_wrapper.py:46-47):
 * images [B,3,S,S] fp32 NCHW -> out [B,E] fp32.  normalize != 0 fuses the
 * `image_feat / image_feat.norm(dim=-1, keepdim=True)` of model_wrapper.py:41. */
int tapclip_encode_image(tapclip_tower_t* vision, const float* images, int32_t batch, float* out,
                         int32_t normalize, void* workspace, size_t workspace_bytes,
                         tapclip_stream_t stream);

/* ---- `clip.model.transformer(x)` as FullModel drives it (reference
 * models/model_wrapper.py:58,72): x [n,T,D] fp32 -> out_hidden [n,T,D] fp32, no
 * positional embedding, no ln_final; causal != 0 adds open_clip's causal mask
 * (the `encode_text` path, clip_wrapper.py:49-51).
 * Optional write-back of the LAST block's attention, i.e. what the forward hook of
 * clip_wrapper.py:29-40 is documented to capture:
 *   attn_heads [n,H,T,T] fp32 softmax probabilities per head      (nullable)
 *   attn_mean  [n,T,T]   fp32 head mean                            (nullable)
 *   attn_out   [n,T,D]   fp32 output of the attention module (post out_proj, pre
 *              residual): what the hook LITERALLY captures (`output[0]`)  (nullable)
 * out_hidden == NULL: the call is for the capture only (model_wrapper.py:58 discards the
 * transformer's output), and the last block stops after its attention. */
int tapclip_text_forward(tapclip_tower_t* text, const float* x, int32_t n_seq, int32_t tokens,
                         int32_t causal, float* out_hidden, float* attn_heads, float* attn_mean,
                         float* attn_out, void* workspace, size_t workspace_bytes,
                         tapclip_stream_t stream);

/* ---- token gather + projection + L2 norm (reference models/model_wrapper.py:73-75:
 * `text_feat[arange(B), -1, :] @ text_projection`, `/ norm`).  index == NULL picks
 * token T-1; otherwise index[i] (int64, device) is the token of sequence i (EOT pool of
 * encode_text).  apply_ln_final != 0 applies ln_final to the picked row first. */
int tapclip_text_pool_project(tapclip_tower_t* text, const float* hidden, int32_t n_seq,
                              int32_t tokens, const int64_t* index, int32_t apply_ln_final,
                              int32_t normalize, float* out, tapclip_stream_t stream);

/* The same backward without the recomputation: tapclip_text_forward_saved runs the blocks (no attention
 * write-back) and keeps, in caller memory of tapclip_text_saved_bytes(), the residual stream before each
 * LayerNorm, q|k|v and the attention output of every block; tapclip_text_backward_saved consumes them.
 * grad_x may alias grad_hidden.  The pair replaces tapclip_text_forward + tapclip_text_backward inside a
 * training step (reference train.py:99-105), where the forward is otherwise run twice. */
size_t tapclip_text_saved_bytes(const tapclip_tower_t* tower, int64_t n_seq, int32_t tokens);
int tapclip_text_forward_saved(tapclip_tower_t* tower, const float* x_in, int32_t n_seq, int32_t tokens,
                               int32_t causal, float* out_hidden, void* saved, size_t saved_bytes,
                               void* workspace, size_t workspace_bytes, tapclip_stream_t stream);
int tapclip_text_backward_saved(tapclip_tower_t* tower, const void* saved, size_t saved_bytes,
                                const float* grad_hidden, int32_t n_seq, int32_t tokens, int32_t causal,
                                float* grad_x, void* workspace, size_t workspace_bytes,
                                tapclip_stream_t stream);

/* ---- tied padding rows.  FullModel builds every text sequence as [P context rows | token_embedding(tokenizer(text))]
 * (reference models/prompt_learner.py:31-34,62-65) and runs the transformer on it without positional embedding and
 * without mask (reference models/model_wrapper.py:58,72).  A tokenised prompt is zero-padded to 77 ids, so the trailing
 * rows of every sequence are ONE embedding row repeated (68-70 of 93 rows at BASELINE.json configs[2]); identical rows
 * stay identical through every block, and the whole pass is the same function of the DISTINCT rows when the attention
 * counts the repeated key `tail_run` times.  The *_tied entry points run exactly that: Tc = tokens - tail_run + 1 rows
 * per sequence inside, full [n, tokens, ...] tensors outside, results equal to the untied calls to fp32 round-off (split
 * modes) / to the mode's own rounding (16-bit modes).  tail_run = 1 is the untied computation.  No causal mask (a
 * position-free notion); attention write-back / backward limits apply to Tc.
 *   tapclip_text_tail_run        largest r such that the last r rows of EVERY sequence of x [n, tokens, width] are
 *                                bit-identical (>= 1); synchronous (waits for `stream`) -- call it once per token bank.
 *   tapclip_text_forward_tied    as tapclip_text_forward (causal = 0): rows / columns of the outputs that belong to the
 *                                run are filled in (each tied column gets 1/tail_run of the group's probability).
 *   tapclip_text_forward_saved_tied / tapclip_text_backward_saved_tied   the training pair; `saved` needs
 *                                tapclip_text_saved_bytes(tower, n_seq, tokens - tail_run + 1).  The tied rows are ONE
 *                                variable: the backward sums grad_hidden over the run on entry and returns the group's
 *                                input gradient in the run's FIRST row (zeros in the others); rows before the run (the
 *                                context rows FullModel differentiates) get exactly the untied gradients.
 * The claim "the last tail_run rows are identical" is checked on the device, without a host round trip: if it is false
 * the handle is poisoned -- every output of every later *_tied call on it is NaN -- until tapclip_text_tied_violations
 * (synchronous) has reported and cleared the condition. */
int tapclip_text_tail_run(const float* x, int32_t n_seq, int32_t tokens, int32_t width, int32_t* run_out,
                          tapclip_stream_t stream);
size_t tapclip_text_tied_workspace_bytes(const tapclip_tower_t* text, int64_t n_seq, int32_t tokens, int32_t tail_run);
int tapclip_text_forward_tied(tapclip_tower_t* text, const float* x, int32_t n_seq, int32_t tokens, int32_t tail_run,
                              float* out_hidden, float* attn_heads, float* attn_mean, float* attn_out,
                              void* workspace, size_t workspace_bytes, tapclip_stream_t stream);
int tapclip_text_forward_saved_tied(tapclip_tower_t* text, const float* x, int32_t n_seq, int32_t tokens,
                                    int32_t tail_run, float* out_hidden, void* saved, size_t saved_bytes,
                                    void* workspace, size_t workspace_bytes, tapclip_stream_t stream);
int tapclip_text_backward_saved_tied(tapclip_tower_t* text, const void* saved, size_t saved_bytes,
                                     const float* grad_hidden, int32_t n_seq, int32_t tokens, int32_t tail_run,
                                     float* grad_x, void* workspace, size_t workspace_bytes, tapclip_stream_t stream);
int tapclip_text_tied_violations(tapclip_tower_t* text, int32_t* violated_out, tapclip_stream_t stream);

/* ---- prompt-tuning backward (reference train.py:99-105: `loss.backward()`; only
 * `prompt_learner.context_bank.*` and `logit_scale` receive gradients, every CLIP weight is frozen,
 * clip_wrapper.py:19-20, so these are dX-only).  Stateless: tapclip_text_backward recomputes the forward
 * from x (the tensor given to tapclip_text_forward) inside its own workspace.
 *   grad_hidden [n,T,D] = dL/d(out_hidden)  ->  grad_x [n,T,D] = dL/dx   (may alias grad_hidden) */
size_t tapclip_text_backward_workspace_bytes(const tapclip_tower_t* text, int64_t n_seq, int32_t tokens);
int tapclip_text_backward(tapclip_tower_t* text, const float* x, const float* grad_hidden, int32_t n_seq,
                          int32_t tokens, int32_t causal, float* grad_x, void* workspace,
                          size_t workspace_bytes, tapclip_stream_t stream);
/* backward of tapclip_text_pool_project with index == NULL (token T-1), no ln_final (model_wrapper.py:73-75):
 * grad_out [n,E] -> grad_hidden [n,T,D] (zero except row T-1). */
int tapclip_text_pool_project_backward(tapclip_tower_t* text, const float* hidden, int32_t n_seq,
                                       int32_t tokens, int32_t normalize, const float* grad_out,
                                       float* grad_hidden, tapclip_stream_t stream);
/* backward of tapclip_logits w.r.t. txt and log(scale) (model_wrapper.py:26,79; img carries no gradient):
 * grad_txt[c,e] = scale * sum_b grad_logits[b,c] img[b,e];  *grad_log_scale = sum grad_logits * logits
 * (grad_log_scale nullable). */
int tapclip_logits_backward(const float* grad_logits, const float* logits, const float* img, float scale,
                            int32_t B, int32_t C, int32_t E, float* grad_txt, float* grad_log_scale,
                            tapclip_stream_t stream);

/* ---- token_embedding(tokens) + positional_embedding (open_clip encode_text prologue;
 * reference models/prompt_learner.py:32-33 calls token_embedding alone: add_pos = 0).
 * tokens [n,L] int64 -> out [n,L,D] fp32.  An id outside [0, vocab) returns TAPCLIP_EINVAL (torch's
 * embedding raises there too); to report it this entry point WAITS for the stream -- it is off the
 * hot path (prompt construction, encode_text) and must not be captured into a graph.
 * ONE caller at a time per tower handle: the out-of-range flag is a single word owned by the handle, so two
 * streams calling this on the same handle concurrently would share (and clear) each other's flag. */
int tapclip_embed_tokens(tapclip_tower_t* text, const int64_t* tokens, int32_t n_seq, int32_t len,
                         int32_t add_pos, float* out, tapclip_stream_t stream);

/* ---- `AttributionMonitor.forward` (reference models/attribution_monitor.py:17-36):
 * attn_map [n,T,T2] -> out [n,P] = softmax_p(attn_map[:, :P, T-1]) (normalize != 0)
 * or the raw column.  T2 is the trailing dim (== T for a real map; the literal hook
 * hands over [n,1,D]: T = 1, T2 = D). */
int tapclip_attribution(const float* attn_map, int32_t n, int32_t T, int32_t T2, int32_t P,
                        int32_t normalize, float* out, tapclip_stream_t stream);

/* ---- `PromptAdjustor('scale')` + the two torch.cat of reference
 * models/prompt_adjustor.py:35-36 and models/model_wrapper.py:51,68-69:
 * out[n, :P] = ctx[n] * attribution[n,:,None] (attribution NULL -> plain copy),
 * out[n, P:] = tok[n].   ctx [n,P,D], tok [n,L,D], out [n,P+L,D].
 * attr_cols = trailing size of attribution (P, or 1 for the literal hook: broadcast). */
int tapclip_build_prompts(const float* ctx, const float* tok, const float* attribution,
                          int32_t attr_cols, int32_t n, int32_t P, int32_t L, int32_t D, float* out,
                          tapclip_stream_t stream);

/* Backward of tapclip_build_prompts towards the context tokens -- the last torch arithmetic of the training forward
 * (reference train.py:99-105 differentiates models/prompt_adjustor.py:35-36 and the torch.cat of model_wrapper.py:69 by
 * autograd): d_ctx[n, t, :] = d_out[n, t, :] * attribution[n, t] for t < P (attribution NULL: a plain copy of those rows).
 * The attribution is a constant of the step (the reference's hook detaches it, models/clip_wrapper.py:36) and the token rows
 * belong to the frozen bank.  d_out [n,P+L,D], d_ctx [n,P,D]. */
int tapclip_build_prompts_backward(const float* d_out, const float* attribution, int32_t attr_cols, int32_t n, int32_t P,
                                   int32_t L, int32_t D, float* d_ctx, tapclip_stream_t stream);

/* ---- `PromptAdjustor('gate' | 'residual')` (reference models/prompt_adjustor.py:13-25,38-44; no reference script selects them)
 * fused with the same two concatenations: a = attribution[n, t]; h = relu(w1 a + b1) with w1, b1 [64] (nn.Linear(1, 64));
 *   TAPCLIP_ADJUST_GATE:     out[n, t] = ctx[n, t] * sigmoid(w2 . h + b2)      w2 [64] (nn.Linear(64, 1).weight), b2 [1]
 *   TAPCLIP_ADJUST_RESIDUAL: out[n, t] = ctx[n, t] + (W2 h + b2)               W2 [D, 64] (nn.Linear(64, D).weight), b2 [D]
 * out[n, P:] = tok[n].  Forward only: a training step that optimises the adjustor's own weights keeps the torch modules. */
#define TAPCLIP_ADJUST_GATE 1
#define TAPCLIP_ADJUST_RESIDUAL 2
int tapclip_build_prompts_mlp(int32_t method, const float* ctx, const float* tok, const float* attribution, int32_t attr_cols,
                              const float* w1, const float* b1, const float* w2, const float* b2, int32_t n, int32_t P,
                              int32_t L, int32_t D, float* out, tapclip_stream_t stream);

/* ---- cosine logits (reference models/model_wrapper.py:79,83):
 * out[b,c] = scale * sum_e img[b,e] * txt[c,e]; img, txt already L2-normalised. */
int tapclip_logits(const float* img, const float* txt, float scale, int32_t B, int32_t C, int32_t E,
                   float* out, tapclip_stream_t stream);

/* ---- the eval transform of `clip.get_preprocess()` (reference models/clip_wrapper.py:56-59 returns open_clip's
 * image_transform(is_train=False); dataset.py:29-35 applies it per sample): Resize(size, BICUBIC) ->
 * CenterCrop(size) -> ToTensor -> Normalize(mean, std), for B decoded uint8 RGB images of any sizes.
 * The resize reproduces Pillow's 8-bit bicubic resampling bit for bit (horizontal pass, then vertical, each
 * rounded to uint8), with torchvision's size rules (shorter side -> size, longer side int(size * long / short);
 * crop origin int(round((n - size) / 2.0))).
 * pixels: device, [h, w, 3] uint8 images.  desc: device [B][4] int64 = {byte offset of the image from `pixels`
 * (any sign: images in separate allocations are addressed by their distance to `pixels`), height, width, byte
 * offset of its scratch in workspace}; image i needs height_i * size * 3 scratch bytes.  mean_std: HOST, mean[3] then std[3].  out: device [B, 3, size, size] fp32. */
int tapclip_preprocess_u8(const uint8_t* pixels, const int64_t* desc, int32_t B, int32_t size,
                          const float* mean_std, void* workspace, float* out, tapclip_stream_t stream);

/* ---- unit entry points for the per-kernel parity tests and roofline micro-benches
 * (SURVEY.md section 2.1 K2 and K3/K5/K6/K7).  Row-major fp32 in/out. -------- */
/* y = LayerNorm(x) * gamma + beta, eps 1e-5; rows x d. */
int tapclip_layernorm_f32(const float* x, const float* gamma, const float* beta, int64_t rows,
                          int32_t d, float* y, tapclip_stream_t stream);
/* C[M,N] = A[M,K] @ W[N,K]^T + bias[N] (bias nullable); precision = TAPCLIP_PREC_*.
 * scratch must hold 2*(M*K + N*K + ...) bf16: use tapclip_gemm_scratch_bytes. */
size_t tapclip_gemm_scratch_bytes(int64_t M, int32_t N, int32_t K);
int tapclip_gemm_f32(const float* A, const float* W, const float* bias, int64_t M, int32_t N,
                     int32_t K, int32_t precision, float* C, void* scratch, size_t scratch_bytes,
                     tapclip_stream_t stream);

/* ---- MXFP8 unit entry points (the fp8 path's operand format, for its parity tests).
 * Elements: OCP e4m3 (saturating at +-448, round to nearest even).  Scales: one e8m0 byte per 32
 * consecutive k, value 2^(byte - 127) = 2^(floor(log2(block amax)) - 8), stored k-step major:
 * scale of (row, block b) at scales[((b >> 1) * rows_pad + row) * 2 + (b & 1)].  K % 64 == 0. */
int tapclip_mx8_quantize(const float* x, int64_t rows, int32_t K, uint8_t* q /* [rows, K] */,
                         uint8_t* scales /* [K/64, rows_pad, 2] */, int64_t rows_pad,
                         tapclip_stream_t stream);
/* C = dequant(A) @ dequant(W)^T + bias on the MXFP8 MFMA path.  N % 256 == 0, K % 64 == 0, K >= 256,
 * m_pad % 8 == 0.  epilogue 0: out_f32 [M, N] fp32.  epilogue 1: act(C) re-quantised to MXFP8 into
 * out_q [M, N] / out_q_scale [N/64, m_pad, 2] (act = TAPCLIP_ACT_*; the c_fc epilogue of the fp8 path). */
int tapclip_mx8_gemm(const uint8_t* a_q, const uint8_t* a_scale, int64_t M, int64_t m_pad,
                     const uint8_t* w_q, const uint8_t* w_scale, const float* bias, int32_t N, int32_t K,
                     int32_t epilogue, int32_t act, float* out_f32, uint8_t* out_q, uint8_t* out_q_scale,
                     tapclip_stream_t stream);

/* ---- the exchange step of the data-parallel path (BASELINE.json north_star: "an RCCL all-gather over xGMI of the image
 * embeddings before the logit-scale cosine-similarity matrix"; nothing of it exists in the single-process reference,
 * /root/reference/train.py:30) for hosts WITHOUT torch.distributed.  One process per GPU; rank 0 draws an id and hands
 * its TAPCLIP_COMM_ID_BYTES bytes to the other ranks out of band; every rank then calls tapclip_comm_create (collective,
 * on the HIP device it will use) and tapclip_allgather(send [bytes_per_rank] -> recv [world * bytes_per_rank], rank-major)
 * on its stream.  A thin layer over RCCL, which is opened at the first call (no link-time dependency).  The Python side
 * of this repository uses torch.distributed for the same step (tap-clip_amd/dist.py).
 * tapclip_allgather returns the ENQUEUE status.  tapclip_comm_check(comm) is the non-blocking query of what happened since
 * (ncclCommGetAsyncError): TAPCLIP_OK while the communicator is healthy or still working; on an asynchronous RCCL error -- a
 * peer that died, a link error -- TAPCLIP_EHIP with the rank and RCCL's text in tapclip_last_error(), after ABORTING the
 * communicator (ncclCommAbort) so that the stream queued behind the collective can drain; every later call on it returns
 * TAPCLIP_ESTATE.  Call it when the event recorded behind a gather has completed -- or while polling that event with a
 * deadline: a collective whose peer is gone never completes by itself.  tapclip_allgather checks before it enqueues. */
typedef struct tapclip_comm tapclip_comm_t;
#define TAPCLIP_COMM_ID_BYTES 128
int tapclip_comm_unique_id(void* id_out);
int tapclip_comm_create(const void* id, int32_t rank, int32_t world, tapclip_comm_t** out);
int tapclip_allgather(tapclip_comm_t* comm, const void* send, void* recv, size_t bytes_per_rank, tapclip_stream_t stream);
int tapclip_comm_check(tapclip_comm_t* comm);
void tapclip_comm_destroy(tapclip_comm_t* comm);

/* ---- behaviour switches of a tower handle.
 * TAPCLIP_FLAG_PRUNE_LAST_BLOCK (image towers, default 1): `encode_image` returns the CLS row only
 * (open_clip pools token 0 before ln_post / proj; reference call site models/clip_wrapper.py:46-47), so in the LAST block
 * every other row's query, attention output, out_proj and MLP are dead work -- the reference computes them and throws
 * them away.  With the flag on, the last block computes K and V for every token and the rest for the CLS rows only
 * (16-bit and split modes: the same results to rounding -- the CLS row's softmax and P.V run in fp32 there; measured against
 * the full computation 3.6e-4 in bf16, whose own rounding is 2.2e-3, 4.4e-5 in fp16, 6e-7 in bf16x3.  fp8 precision: NOT
 * the same pipeline -- the pooled rows' last block runs on 16-bit copies of that block's weights (14-25 MB per tower, made at
 * load time unless TAPCLIP_PRUNE_LAST=0 is in the environment) instead of MXFP8, so those rows see one block less of MXFP8
 * rounding: 9.6e-3 (ViT-B/16) / 6.5e-3 (ViT-L/14@336) from the all-MXFP8 computation, test bound 3e-2; parity of the fp8 mode
 * is unpinned either way, the reference has no fp8).  0 = compute every row of every block
 * (what bench.py's headline `value` times: the full 35.127 GFLOP per ViT-B/16 image of SURVEY.md section 8d). */
#define TAPCLIP_FLAG_PRUNE_LAST_BLOCK 1
/* TAPCLIP_FLAG_KSPLIT (both towers, default 1): when a GEMM launch ends in a partial round of tiles (c_proj at ViT-B/16,
 * batch 256: 2.31 rounds of 256 workgroups) or has fewer tiles than half the CUs (the text tower at 65 classes), those
 * tiles are K-split over the idle CUs and summed by a fix-up kernel: the launch finishes sooner, at MORE CU-time in total
 * (more workgroups + the fix-up).  That is the right trade when the tower has the GPU to itself (the image encoder alone,
 * the text backward) and the wrong one when another stream wants the idle CUs: with both towers of FullModel.forward in
 * flight (reference models/model_wrapper.py:40-75) the whole forward is 12.7 ms with the splits and 12.2 ms without, so
 * the Python FullModel switches it off for its forward and back on for the backward.  Results differ between the two
 * settings by fp32 summation order only (the rounding-level dependence documented for the tail split). */
#define TAPCLIP_FLAG_KSPLIT 2
int tapclip_tower_set_flag(tapclip_tower_t* tower, int32_t flag, int32_t value);
/* the current value of a flag (a caller that changes one for a while restores what it found, not the default) */
int tapclip_tower_get_flag(const tapclip_tower_t* tower, int32_t flag, int32_t* value);

/* ---- per-stage timing (HIP events on `stream`) for bench.py's roofline object.
 * When enabled, tapclip_encode_image records events around each kernel family;
 * tapclip_profile_read (after a stream sync) returns accumulated ms and launch counts.
 * slots: 0 patch-embed, 1 layernorm, 2 gemm_qkv, 3 attention, 4 gemm_out_proj,
 *        5 gemm_fc_gelu, 6 gemm_proj, 7 pool_proj (pool + ln_post + proj + L2 norm), 8 pooled_tail (the CLS rows' Q,
 *        attention, out_proj, LN2 and MLP of the last block when TAPCLIP_FLAG_PRUNE_LAST_BLOCK is on). */
#define TAPCLIP_PROFILE_SLOTS 9
int tapclip_profile_enable(tapclip_tower_t* tower, int32_t on);
int tapclip_profile_read(tapclip_tower_t* tower, float* ms_out, int64_t* launches_out);

const char* tapclip_last_error(void);
int tapclip_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* TAPCLIP_H */
