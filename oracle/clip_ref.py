"""CPU fp32 ORACLE for the CLIP towers -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this file; the product path (`tap-clip_amd/`) never does and fails loudly
when its HIP library is missing.

What it restates
----------------
The reference (3300786/TAP-CLIP) holds none of the encoder arithmetic itself: it
calls the un-vendored, version-unpinned PyPI package `open_clip_torch`
(reference call sites: models/clip_wrapper.py:5,13,27,39,47,51;
models/model_wrapper.py:58,72,74; models/prompt_learner.py:11-13,32-33).
`open_clip` is absent from this container, so this file restates its published
architecture (batch-first `Transformer` release) in plain
PyTorch fp32, with open_clip's state-dict key names:

* `VisionTransformer`: conv1 (stride = patch, no bias) -> class token concat ->
  + positional_embedding -> ln_pre -> L pre-LN residual blocks -> ln_post ->
  CLS pool -> @ proj.
* text `Transformer`: L pre-LN residual blocks over `[n, T, D]` (batch first).
  `FullModel` drives it WITHOUT positional embedding, causal mask or ln_final
  (models/model_wrapper.py:58,72); `encode_text` (models/clip_wrapper.py:49-51)
  is the full open_clip text path with all three.
* block: x = x + out_proj(MHA(ln_1(x))); x = x + c_proj(gelu(c_fc(ln_2(x)))),
  MHA = nn.MultiheadAttention math with explicit q/k/v so per-head softmax
  probabilities are observable (the "intended" attention map of
  models/clip_wrapper.py:35-36).

PARITY PIN STATUS: the encoder arithmetic is **parity unpinned** against
open_clip itself (the reference tree holds no test, fixture or golden vector for
it).  It is cross-checked in tests against an independent
implementation (HF `transformers` CLIP built from a config, random weights) and
against `torch.nn.MultiheadAttention`.  The reference's own L2/L3 modules
(FullModel, PromptLearner, AttributionMonitor, PromptAdjustor) ARE run unmodified
on top of this file by `oracle/make_golden.py` to produce `tests/golden/*.npz`.

`emulate` option: when set to "bf16" every GEMM operand is rounded to bf16 at the
same points where the HIP kernels round (LN outputs, q/k/v, softmax
probabilities, attention output, GELU output, weights), everything else stays
fp32.  That gives a reference for the fast bf16 path whose only difference from
the kernels is accumulation order.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# model configurations (open_clip model_configs/*.json, restated)
# --------------------------------------------------------------------------
@dataclass(frozen=True)
class TowerDims:
    width: int
    layers: int
    heads: int
    mlp: int


@dataclass(frozen=True)
class ClipDims:
    name: str
    embed_dim: int
    image_size: int
    patch: int
    vision: TowerDims
    text: TowerDims
    vocab: int = 49408
    ctx: int = 77
    quick_gelu: bool = False

    @property
    def grid(self) -> int:
        return self.image_size // self.patch

    @property
    def n_tokens(self) -> int:
        return self.grid * self.grid + 1


CONFIGS: Dict[str, ClipDims] = {
    "ViT-B-32": ClipDims("ViT-B-32", 512, 224, 32, TowerDims(768, 12, 12, 3072), TowerDims(512, 12, 8, 2048)),
    "ViT-B-16": ClipDims("ViT-B-16", 512, 224, 16, TowerDims(768, 12, 12, 3072), TowerDims(512, 12, 8, 2048)),
    "ViT-L-14-336": ClipDims("ViT-L-14-336", 768, 336, 14, TowerDims(1024, 24, 16, 4096), TowerDims(768, 12, 12, 3072)),
    # tiny configurations used by the golden fixtures (not open_clip models)
    "tiny": ClipDims("tiny", 64, 32, 8, TowerDims(128, 2, 2, 256), TowerDims(128, 2, 2, 256), vocab=97, ctx=77),
}


def _rb(x: torch.Tensor, emulate: Optional[str]) -> torch.Tensor:
    """Round-trip through the emulated operand dtype (identity for fp32)."""
    if emulate is None:
        return x
    if emulate == "bf16":  # (back to x's own dtype: the emulation also runs in float64, see emulation_floor)
        return x.to(torch.bfloat16).to(x.dtype)
    if emulate == "fp16":
        return x.to(torch.float16).to(x.dtype)
    if emulate == "mx8":  # the fp8 path keeps bf16 wherever it is not an MXFP8 GEMM operand
        return x.to(torch.bfloat16).to(x.dtype)
    raise ValueError(emulate)


def _rq(x: torch.Tensor, emulate: Optional[str]) -> torch.Tensor:
    """Rounding of a block-GEMM operand (k = last dim): MXFP8 on the fp8 path (oracle/mx8_ref.py), else as _rb."""
    if emulate == "mx8":
        from . import mx8_ref
        return mx8_ref.fake_quant(x).to(x.dtype)  # (exact in fp32; back to x's dtype: the emulation also runs in float64)
    return _rb(x, emulate)


def gelu_fit(x: torch.Tensor) -> torch.Tensor:
    """The GELU of the 16-/8-bit fast paths (tap-clip_amd/csrc/common.h gelu_erf_fast): x * sigmoid(x * (a + b x^2 +
    c x^4)) fitted to the exact-erf GELU, max |error| 2.5e-5.  Only used under `emulate`: that error is 1-6 % of a
    bf16 ulp of the stored value, so against the exact form a few per cent of the MLP-hidden elements round the
    other way (each a full 2^-8 relative step) -- the 3e-3 per-block rel-max of the round-1 log came from there."""
    xc = x.clamp(-10.0, 10.0)
    x2 = xc * xc
    p = 1.0142631e-3 * x2 - 1.0677573e-1
    p = p * x2 - 2.3011213
    return x / (1.0 + torch.exp2(p * xc))


def _act(x: torch.Tensor, quick: bool, emulate: Optional[str] = None) -> torch.Tensor:
    if quick:
        return x * torch.sigmoid(1.702 * x)
    if emulate is not None:
        return gelu_fit(x)
    return F.gelu(x)  # exact erf form, nn.GELU() default


# --------------------------------------------------------------------------
# one residual attention block
# --------------------------------------------------------------------------
def block_forward(
    x: torch.Tensor,  # [n, T, D] fp32
    sd: Dict[str, torch.Tensor],
    prefix: str,  # e.g. "visual.transformer.resblocks.3."
    heads: int,
    attn_mask: Optional[torch.Tensor] = None,  # additive [T, T]
    quick_gelu: bool = False,
    emulate: Optional[str] = None,
    want_probs: bool = False,
    taps: Optional[dict] = None,
) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    n, T, D = x.shape
    hd = D // heads
    g = lambda k: sd[prefix + k]

    h = F.layer_norm(x, (D,), g("ln_1.weight"), g("ln_1.bias"), 1e-5)
    if taps is not None:
        taps["ln_1"] = h
    w_in = g("attn.in_proj_weight")
    b_in = g("attn.in_proj_bias")
    if emulate:
        # the kernels fold 1/sqrt(hd) (a power of two for hd = 64) into Wq, bq
        qkv = F.linear(_rq(h, emulate), _rq(w_in, emulate), b_in)
    else:
        qkv = F.linear(h, w_in, b_in)
    if taps is not None:
        taps["qkv"] = qkv
    q, k, v = qkv.split(D, dim=-1)
    q = q.reshape(n, T, heads, hd).transpose(1, 2)  # [n, H, T, hd]
    k = k.reshape(n, T, heads, hd).transpose(1, 2)
    v = v.reshape(n, T, heads, hd).transpose(1, 2)
    scale = 1.0 / math.sqrt(hd)
    q = q * scale
    s = _rb(q, emulate) @ _rb(k, emulate).transpose(-1, -2)  # [n, H, T, T]
    if attn_mask is not None:
        s = s + attn_mask
    p = torch.softmax(s, dim=-1)
    if emulate:
        # the attention kernel rounds the UN-normalised exp(s - max) to bf16 for the PV product and
        # divides the 64 outputs by the fp32 row sum afterwards (same function, other rounding point)
        e = torch.exp(s - s.amax(dim=-1, keepdim=True))
        o = (_rb(e, emulate) @ _rb(v, emulate)) / e.sum(dim=-1, keepdim=True)
    else:
        o = p @ v  # [n, H, T, hd]
    o = o.transpose(1, 2).reshape(n, T, D)
    if taps is not None:
        taps["probs"] = p
        taps["attn_ctx"] = o
    a = F.linear(_rq(o, emulate), _rq(g("attn.out_proj.weight"), emulate), g("attn.out_proj.bias"))
    if taps is not None:
        taps["attn_out"] = a
    x = x + _rb(a, emulate)  # the kernels hand the branch to the next LayerNorm kernel as bf16

    h = F.layer_norm(x, (D,), g("ln_2.weight"), g("ln_2.bias"), 1e-5)
    h = F.linear(_rq(h, emulate), _rq(g("mlp.c_fc.weight"), emulate), g("mlp.c_fc.bias"))
    h = _act(h, quick_gelu, emulate)
    if taps is not None:
        taps["mlp_hidden"] = h
    h = F.linear(_rq(h, emulate), _rq(g("mlp.c_proj.weight"), emulate), g("mlp.c_proj.bias"))
    x = x + _rb(h, emulate)
    if emulate == "mx8":
        x = _rb(x, emulate)  # the fp8 path keeps the residual stream of the blocks in 16 bits (written once per block)
    if taps is not None:
        taps["out"] = x
    return x, (p if want_probs else None)


def emulation_floor(x: torch.Tensor, sd: Dict[str, torch.Tensor], prefix: str, heads: int, emulate: str = "bf16"):
    """How closely ANY two implementations that round at the same points can agree on one block: the emulation with
    fp32 accumulation against the same emulation with fp64 accumulation, per tap, as (rel_l2, rel_max).

    Two pipelines whose pre-rounding values differ by a relative eps disagree on a fraction ~eps/u of the rounded
    elements by one whole step u (2^-8 relative for bf16), i.e. by sqrt(eps * u) in rms -- far more than eps -- and the
    next rounding stage amplifies again: 1e-7 -> 2e-5 (q|k|v) -> 2e-4 (attention output) -> 6e-4 (out_proj) -> 1.1e-3
    rel-L2 / 3e-3 rel-max at the block output (ViT-B widths).  Round 1's kernels measured exactly these numbers against
    the emulation (gpurun_out/t9.log): that gap is this floor, not a rounding point that differs."""
    t32, t64 = {}, {}
    block_forward(x, sd, prefix, heads, emulate=emulate, taps=t32)
    block_forward(x.double(), {k: v.double() for k, v in sd.items() if k.startswith(prefix)}, prefix, heads, emulate=emulate, taps=t64)
    out = {}
    for k in t32:
        a, b = t32[k].double(), t64[k]
        out[k] = (float((a - b).norm() / b.norm()), float((a - b).abs().max() / b.abs().max()))
    return out


def transformer_forward(
    x: torch.Tensor,
    sd: Dict[str, torch.Tensor],
    prefix: str,  # "visual.transformer." or "transformer."
    layers: int,
    heads: int,
    attn_mask: Optional[torch.Tensor] = None,
    quick_gelu: bool = False,
    emulate: Optional[str] = None,
    want_last_probs: bool = False,
    want_last_attn_out: bool = False,
):
    """L residual blocks.  Returns (hidden, last-layer per-head probs [n,H,T,T] | None,
    last-layer attention-module output (post out_proj) [n,T,D] | None)."""
    probs = None
    attn_out = None
    for i in range(layers):
        last = i == layers - 1
        taps = {} if (last and want_last_attn_out) else None
        x, p = block_forward(
            x, sd, f"{prefix}resblocks.{i}.", heads, attn_mask, quick_gelu, emulate,
            want_probs=(last and want_last_probs), taps=taps,
        )
        if last:
            probs = p
            if taps is not None:
                attn_out = taps["attn_out"]
    return x, probs, attn_out


# --------------------------------------------------------------------------
# towers
# --------------------------------------------------------------------------
def encode_image(images: torch.Tensor, sd: Dict[str, torch.Tensor], cfg: ClipDims,
                 emulate: Optional[str] = None, normalize: bool = False) -> torch.Tensor:
    """open_clip `model.encode_image` (reference call site models/clip_wrapper.py:46-47)."""
    B = images.shape[0]
    w = sd["visual.conv1.weight"]  # [width, 3, p, p]
    width = w.shape[0]
    if emulate:
        # the kernels do patch-embed as an im2col GEMM with rounded operands
        x = F.conv2d(_rb(images, emulate), _rb(w, emulate), None, stride=cfg.patch)
    else:
        x = F.conv2d(images, w, None, stride=cfg.patch)
    x = x.reshape(B, width, -1).permute(0, 2, 1)  # [B, grid^2, width]
    cls = sd["visual.class_embedding"].reshape(1, 1, width).expand(B, -1, -1)
    x = torch.cat([cls, x], dim=1)
    x = x + sd["visual.positional_embedding"]
    x = F.layer_norm(x, (width,), sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"], 1e-5)
    if emulate == "mx8":
        x = _rb(x, emulate)
    x, _, _ = transformer_forward(x, sd, "visual.transformer.", cfg.vision.layers, cfg.vision.heads,
                                  None, cfg.quick_gelu, emulate)
    pooled = F.layer_norm(x[:, 0], (width,), sd["visual.ln_post.weight"], sd["visual.ln_post.bias"], 1e-5)
    out = pooled @ sd["visual.proj"]
    if normalize:
        out = out / out.norm(dim=-1, keepdim=True)
    return out


def causal_mask(T: int) -> torch.Tensor:
    return torch.full((T, T), float("-inf")).triu_(1)


def encode_text(tokens: torch.Tensor, sd: Dict[str, torch.Tensor], cfg: ClipDims,
                emulate: Optional[str] = None, normalize: bool = False) -> torch.Tensor:
    """open_clip `model.encode_text` (reference call site models/clip_wrapper.py:49-51):
    token-emb + pos-emb -> causal transformer -> ln_final -> EOT(argmax) pool -> @ text_projection."""
    x = sd["token_embedding.weight"][tokens]  # [n, 77, D]
    x = x + sd["positional_embedding"][: tokens.shape[1]]
    T = tokens.shape[1]
    x, _, _ = transformer_forward(x, sd, "transformer.", cfg.text.layers, cfg.text.heads,
                                  causal_mask(T), cfg.quick_gelu, emulate)
    D = x.shape[-1]
    x = F.layer_norm(x, (D,), sd["ln_final.weight"], sd["ln_final.bias"], 1e-5)
    pooled = x[torch.arange(x.shape[0]), tokens.argmax(dim=-1)]
    out = pooled @ sd["text_projection"]
    if normalize:
        out = out / out.norm(dim=-1, keepdim=True)
    return out


def text_transformer_raw(x: torch.Tensor, sd: Dict[str, torch.Tensor], cfg: ClipDims,
                         emulate: Optional[str] = None, want_probs: bool = False,
                         want_attn_out: bool = False):
    """`clip.model.transformer(x)` exactly as FullModel calls it
    (models/model_wrapper.py:58,72): no pos-emb, no mask, no ln_final."""
    return transformer_forward(x, sd, "transformer.", cfg.text.layers, cfg.text.heads, None,
                               cfg.quick_gelu, emulate, want_last_probs=want_probs,
                               want_last_attn_out=want_attn_out)
