"""CPU fp32 ORACLE for `FullModel.forward` -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this file.

Restates reference models/model_wrapper.py:28-100 on top of `oracle/clip_ref.py`:

* `forward_literal`  -- the reference loop as written: per class, B batch-1
  attribution passes (model_wrapper.py:55-63) + one batch-B pass (:72); only for
  small cases (it is O(n_cls * B) transformer passes).
* `forward_collapsed` -- the exactly equivalent form:
  text features do not depend on the image, so each class is run once.
  `tests/test_oracle.py` checks literal == collapsed.

`attn_semantics`:
  "literal"  -- what the reference's hook really captures (clip_wrapper.py:34-37):
                `output[0]` of nn.MultiheadAttention is the attention OUTPUT
                [1,T,D]; `.mean(dim=1)` -> [1,D]; model_wrapper.py:60-61 unsqueezes
                to [1,1,D]; AttributionMonitor (attribution_monitor.py:24-32) then
                slices `[:, :P, 0]` -> [1,1] and softmaxes one element -> 1.0;
                PromptAdjustor 'scale' (prompt_adjustor.py:35-36) multiplies by 1.
  "intended" -- what the comments document (clip_wrapper.py:35-36,
                attribution_monitor.py:19-29): per-head weights [n,H,T,T] ->
                head mean [n,T,T] -> column T-1, rows :P -> softmax over P ->
                scale each context token.

PARITY PIN STATUS: pinned for both semantics by `tests/golden/fullmodel_*.npz`,
which `oracle/make_golden.py` produced by running the reference's own
FullModel / PromptLearner / AttributionMonitor / PromptAdjustor classes,
unmodified, in this container (tower arithmetic underneath: see clip_ref.py,
"parity unpinned" against open_clip).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from . import clip_ref


def attribution_from_map(attn_map: torch.Tensor, prompt_len: int, normalize: bool = True) -> torch.Tensor:
    """reference models/attribution_monitor.py:17-36 on [n, T, T']."""
    T = attn_map.shape[1]
    raw = attn_map[:, :prompt_len, T - 1]
    return torch.softmax(raw, dim=-1) if normalize else raw


def adjust_scale(ctx: torch.Tensor, attribution: torch.Tensor) -> torch.Tensor:
    """reference models/prompt_adjustor.py:35-36."""
    return ctx * attribution.unsqueeze(-1)


def text_features(prompts: torch.Tensor, prompt_len: int, sd: Dict[str, torch.Tensor],
                  cfg: clip_ref.ClipDims, attn_semantics: str = "intended",
                  emulate: Optional[str] = None, return_aux: bool = False):
    """Collapsed text side: prompts [n_cls, P+77, D] -> normalised text features [n_cls, E].

    pass 1 (model_wrapper.py:58): raw transformer for the hook -> attribution;
    pass 2 (:72-75): adjusted prompt -> token -1 -> @ text_projection -> L2 norm."""
    ctx = prompts[:, :prompt_len]
    cls_tok = prompts[:, prompt_len:]
    aux = {}
    if attn_semantics == "intended":
        _, probs, _ = clip_ref.text_transformer_raw(prompts, sd, cfg, emulate, want_probs=True)
        attn_map = probs.mean(dim=1)  # [n, T, T]
        attr = attribution_from_map(attn_map, prompt_len)
        aux["attn_probs"] = probs
        aux["attn_map"] = attn_map
    elif attn_semantics == "literal":
        _, _, attn_out = clip_ref.text_transformer_raw(prompts, sd, cfg, emulate, want_attn_out=True)
        m = attn_out.mean(dim=1)  # [n, D]   (hook: output[0].mean(dim=1))
        attn_map = m.unsqueeze(1)  # per-sample [1, D] -> unsqueeze(0) -> [1,1,D]
        attr = attribution_from_map(attn_map, prompt_len)  # [n, 1] == 1.0
        aux["attn_map"] = attn_map
    else:
        raise ValueError(attn_semantics)
    aux["attribution"] = attr
    adjusted = torch.cat([adjust_scale(ctx, attr), cls_tok], dim=1)
    hidden, _, _ = clip_ref.text_transformer_raw(adjusted, sd, cfg, emulate)
    feat = hidden[:, -1, :] @ sd["text_projection"]
    feat = feat / feat.norm(dim=-1, keepdim=True)
    aux["hidden_last"] = hidden[:, -1, :]
    return (feat, aux) if return_aux else feat


def forward_collapsed(images: torch.Tensor, prompts: torch.Tensor, prompt_len: int,
                      sd: Dict[str, torch.Tensor], cfg: clip_ref.ClipDims,
                      logit_scale: float = math.log(1 / 0.07), labels: Optional[torch.Tensor] = None,
                      attn_semantics: str = "intended", emulate: Optional[str] = None):
    img = clip_ref.encode_image(images, sd, cfg, emulate, normalize=True)
    txt, aux = text_features(prompts, prompt_len, sd, cfg, attn_semantics, emulate, return_aux=True)
    logits = math.exp(logit_scale) * img @ txt.t()
    out = {"logits": logits, "image_features": img, "text_features": txt, **aux}
    if labels is not None:
        out["loss"] = F.cross_entropy(logits, labels)
    return out


def forward_literal(images: torch.Tensor, prompts: torch.Tensor, prompt_len: int,
                    sd: Dict[str, torch.Tensor], cfg: clip_ref.ClipDims,
                    logit_scale: float = math.log(1 / 0.07), labels: Optional[torch.Tensor] = None,
                    attn_semantics: str = "intended"):
    """The reference loop nest as written (model_wrapper.py:47-83)."""
    B = images.shape[0]
    img = clip_ref.encode_image(images, sd, cfg, None, normalize=True)
    sims = []
    for i in range(prompts.shape[0]):
        ctx = prompts[i, :prompt_len].unsqueeze(0).expand(B, -1, -1)
        cls_tok = prompts[i, prompt_len:].unsqueeze(0).expand(B, -1, -1)
        full = torch.cat([ctx, cls_tok], dim=1)
        attrs = []
        for b in range(B):
            single = full[b].unsqueeze(0)
            if attn_semantics == "intended":
                _, probs, _ = clip_ref.text_transformer_raw(single, sd, cfg, None, want_probs=True)
                amap = probs.mean(dim=1)
            else:
                _, _, ao = clip_ref.text_transformer_raw(single, sd, cfg, None, want_attn_out=True)
                amap = ao.mean(dim=1).unsqueeze(0)
            attrs.append(attribution_from_map(amap, prompt_len))
        attribution = torch.cat(attrs, dim=0)
        adjusted = torch.cat([adjust_scale(ctx, attribution), cls_tok], dim=1)
        hidden, _, _ = clip_ref.text_transformer_raw(adjusted, sd, cfg, None)
        tf = hidden[torch.arange(B), -1, :] @ sd["text_projection"]
        tf = tf / tf.norm(dim=-1, keepdim=True)
        sims.append(math.exp(logit_scale) * (img * tf).sum(dim=-1, keepdim=True))
    logits = torch.cat(sims, dim=1)
    out = {"logits": logits}
    if labels is not None:
        out["loss"] = F.cross_entropy(logits, labels)
    return out
