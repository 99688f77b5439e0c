"""Harness that lets the REFERENCE's own L2/L3 classes run in this container -- TEST
INFRASTRUCTURE, NOT PRODUCT CODE (only tests/ and oracle/make_golden.py import it).

The reference's `models/model_wrapper.py` (FullModel), `models/prompt_learner.py`,
`models/attribution_monitor.py` and `models/prompt_adjustor.py` import fine here (torch only);
`models/clip_wrapper.py` does not (`open_clip` is absent and stays absent).  `RefClip` below is the
object handed to the reference's `FullModel(class_names, clip_wrapper, ...)` in its place.  It
exposes exactly the attribute surface those classes touch and is built from
REAL `torch.nn.MultiheadAttention` / `nn.LayerNorm` / `nn.Linear` / `nn.GELU` modules -- the same
torch modules open_clip's ResidualAttentionBlock is made of -- so it is an implementation
independent of `oracle/clip_ref.py`'s explicit-q/k/v restatement, and the two are compared in
tests/test_oracle.py.

Hook semantics (reference models/clip_wrapper.py:29-40):
  "literal"  : MHA is called like open_clip calls it (`need_weights=False`), the forward hook
               captures `output[0].detach().mean(dim=1)` -- the reference's hook body.
  "intended" : MHA is called with `need_weights=True, average_attn_weights=False`; the hook
               captures `output[1].detach().mean(dim=1)`, the head-mean [n,T,T] map the
               reference's comments describe.
"""
from __future__ import annotations

import contextlib
import sys
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import clip_ref

REFERENCE_ROOT = "/root/reference"


class _Block(nn.Module):
    def __init__(self, d: int, heads: int, mlp: int, quick_gelu: bool):
        super().__init__()
        self.ln_1 = nn.LayerNorm(d)
        self.attn = nn.MultiheadAttention(d, heads, batch_first=True)
        self.ln_2 = nn.LayerNorm(d)
        act = (lambda: _QuickGELU()) if quick_gelu else nn.GELU
        self.mlp = nn.Sequential()
        self.mlp.add_module("c_fc", nn.Linear(d, mlp))
        self.mlp.add_module("gelu", act())
        self.mlp.add_module("c_proj", nn.Linear(mlp, d))
        self.need_weights = False

    def forward(self, x, attn_mask=None):
        h = self.ln_1(x)
        a = self.attn(h, h, h, need_weights=self.need_weights, average_attn_weights=False, attn_mask=attn_mask)[0]
        x = x + a
        return x + self.mlp(self.ln_2(x))


class _QuickGELU(nn.Module):
    def forward(self, x):
        return x * torch.sigmoid(1.702 * x)


class _Transformer(nn.Module):
    def __init__(self, d: int, layers: int, heads: int, mlp: int, quick_gelu: bool):
        super().__init__()
        self.resblocks = nn.ModuleList([_Block(d, heads, mlp, quick_gelu) for _ in range(layers)])

    def forward(self, x, attn_mask=None):
        for b in self.resblocks:
            x = b(x, attn_mask)
        return x


class _TextModel(nn.Module):
    """The part of open_clip's CLIP module the reference touches on the text side."""

    def __init__(self, cfg: clip_ref.ClipDims):
        super().__init__()
        t = cfg.text
        self.transformer = _Transformer(t.width, t.layers, t.heads, t.mlp, cfg.quick_gelu)
        self.token_embedding = nn.Embedding(cfg.vocab, t.width)
        self.text_projection = nn.Parameter(torch.empty(t.width, cfg.embed_dim))


class RefClip(nn.Module):
    """Stand-in for the reference's CLIPWrapper instance (NOT for the open_clip library)."""

    def __init__(self, cfg: clip_ref.ClipDims, sd: Dict[str, torch.Tensor], attn_semantics: str, tokenizer):
        super().__init__()
        self.cfg, self.sd, self.attn_semantics = cfg, sd, attn_semantics
        self.device = "cpu"
        self.model = _TextModel(cfg)
        own = self.model.state_dict()
        missing = [k for k in own if k not in sd]
        assert not missing, missing
        self.model.load_state_dict({k: sd[k] for k in own}, strict=True)
        self.model.eval()
        for p in self.model.parameters():
            p.requires_grad = False
        self.attention_maps: List[torch.Tensor] = []
        self.tokenizer = tokenizer
        last = self.model.transformer.resblocks[-1]
        if attn_semantics == "intended":
            last.need_weights = True

            def hook_fn(module, input, output):
                self.attention_maps.append(output[1].detach().mean(dim=1))
        else:

            def hook_fn(module, input, output):  # the reference's hook body (clip_wrapper.py:34-37)
                self.attention_maps.append(output[0].detach().mean(dim=1))

        last.attn.register_forward_hook(hook_fn)

    def reset(self):
        self.attention_maps.clear()

    def encode_image(self, images):
        return clip_ref.encode_image(images, self.sd, self.cfg)

    def get_attention_map(self):
        return self.attention_maps[-1] if self.attention_maps else None

    def get_tokenizer(self):
        return self.tokenizer


class FixedTokenizer:
    """Returns committed synthetic token ids ([1,77] int64) per prompt text: the BPE vocabulary
    is not available offline."""

    def __init__(self, table: Dict[str, torch.Tensor]):
        self.table = table

    def __call__(self, text):
        return self.table[text].clone()


@contextlib.contextmanager
def reference_modules():
    """Import the reference's L2/L3 modules from /root/reference with PromptLearner's
    device default patched to 'cpu' (prompt_learner.py:7 defaults to 'cuda' and FullModel never
    passes one, so construction would raise on a CPU-only box)."""
    def ours(k):  # the reference's top-level package names (models/, utils/) may already be taken in this process
        return k in ("models", "utils") or k.startswith("models.") or k.startswith("utils.")

    saved = {k: v for k, v in sys.modules.items() if ours(k)}
    for k in saved:
        del sys.modules[k]
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        import models.attribution_monitor as am
        import models.model_wrapper as mw
        import models.prompt_adjustor as pa
        import models.prompt_learner as pl
        import utils.eval_metrics as em  # reference utils/eval_metrics.py:6-96 (torch only)

        old = pl.PromptLearner.__init__.__defaults__
        pl.PromptLearner.__init__.__defaults__ = old[:-1] + ("cpu",)
        try:
            yield {"FullModel": mw.FullModel, "PromptLearner": pl.PromptLearner,
                   "AttributionMonitor": am.AttributionMonitor, "PromptAdjustor": pa.PromptAdjustor,
                   "eval_metrics": em}
        finally:
            pl.PromptLearner.__init__.__defaults__ = old
    finally:
        sys.path.remove(REFERENCE_ROOT)
        for k in [k for k in sys.modules if ours(k)]:
            del sys.modules[k]
        sys.modules.update(saved)
