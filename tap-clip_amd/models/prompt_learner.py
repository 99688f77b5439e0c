"""Per-class learnable context tokens + frozen class-token embeddings.

Drop-in for reference models/prompt_learner.py:5-70 (same constructor arguments, `context_bank`
ParameterDict keyed by class name in class order, `token_bank`, `add_class_prompt`, `forward() ->
[n_cls, prompt_len + 77, D]`, `n_cls`).  Differences, all host-side:
  * `device=None` follows the CLIP wrapper's device (the reference defaults to 'cuda' and
    `FullModel` never forwards one: prompt_learner.py:7, model_wrapper.py:19);
  * the token embedding lookup runs through the HIP `tapclip_embed_tokens` kernel;
  * the stacked `[n_cls, P, D]` / `[n_cls, 77, D]` views used by the fused text path are cached.
"""
from typing import Dict, Optional

import torch
import torch.nn as nn


def host_tail_run(tok: torch.Tensor) -> int:
    """Largest r such that the last r rows of every sequence of tok [n, L, D] are identical (>= 1), counted on the host."""
    same = (tok == tok[:, -1:, :]).all(dim=-1).flip(1).long().cumprod(dim=1).sum(dim=1)
    return int(same.min())


class PromptLearner(nn.Module):
    def __init__(self, class_names, clip_model, prompt_len=5, class_specific=True, use_init_prompt=True, device=None):
        super().__init__()
        self.prompt_len = prompt_len
        self.class_specific = class_specific
        self.use_init_prompt = use_init_prompt
        self.device = device if device is not None else clip_model.device
        self.tokenizer = clip_model.get_tokenizer()
        self.token_embedding = clip_model.model.token_embedding  # frozen, shared with the CLIP wrapper
        self.ctx_dim = self.token_embedding.embedding_dim

        self.context_bank = nn.ParameterDict()          # class name -> [P, D], trainable
        self.token_bank: Dict[str, torch.Tensor] = {}   # class name -> [1, 77, D], frozen
        self._tok_cache: Optional[torch.Tensor] = None
        self._tail_run: Optional[int] = None

        print(f"cls_specific: {class_specific}, use_init_prompt: {use_init_prompt}")
        for name in class_names:
            self.add_class_prompt(name)

    @torch.no_grad()
    def add_class_prompt(self, class_name: str) -> None:
        """Register a (possibly unseen) class: reference prompt_learner.py:26-43,
        used by test_cross_domain.py:65-67."""
        if class_name in self.context_bank:
            return
        ids = self.tokenizer(f"a photo of a {class_name}").to(self.device)      # [1, 77]
        emb = self.token_embedding(ids.unsqueeze(0)).squeeze(0)                 # [1, 77, D]
        self.token_bank[class_name] = emb
        # The reference tests `token_emb.shape[0] >= 5 + prompt_len` on this [1, 77, D] tensor, i.e.
        # 1 >= 5 + P, which never holds: the context is therefore ALWAYS Gaussian-initialised
        # (prompt_learner.py:37-41).  Kept as is.
        if self.use_init_prompt and emb.shape[0] >= 5 + self.prompt_len:
            init = emb[5: 5 + self.prompt_len].clone()
        else:
            init = torch.randn(self.prompt_len, self.ctx_dim).to(self.device)
        self.context_bank[class_name] = nn.Parameter(init)
        self._tok_cache = None
        self._tail_run = None

    @torch.no_grad()
    def refresh_token_bank(self) -> None:
        """Re-embed every class prompt (after the CLIP weights were replaced by a load_state_dict)."""
        for class_name in self.context_bank:
            ids = self.tokenizer(f"a photo of a {class_name}").to(self.device)
            self.token_bank[class_name] = self.token_embedding(ids.unsqueeze(0)).squeeze(0)
        self._tok_cache = None
        self._tail_run = None

    # ---- stacked views for the fused path ----------------------------------------------------
    def stacked_context(self) -> torch.Tensor:
        """[n_cls, P, D] in class order (differentiable w.r.t. every context_bank entry)."""
        return torch.stack([self.context_bank[c] for c in self.context_bank], dim=0)

    def stacked_tokens(self) -> torch.Tensor:
        """[n_cls, 77, D]"""
        if self._tok_cache is None:
            rows = []
            for c in self.context_bank:
                t = self.token_bank[c]
                if t.dim() == 2:
                    t = t.unsqueeze(0)
                elif t.dim() == 4:
                    t = t.squeeze(0)
                elif t.dim() != 3:
                    raise ValueError(f"Unexpected token shape: {t.shape}")
                rows.append(t)
            self._tok_cache = torch.cat(rows, dim=0)
        return self._tok_cache

    def tail_run(self) -> int:
        """How many trailing rows of EVERY class prompt are one and the same row: the tokenizer pads each prompt with zeros
        to 77 ids (reference prompt_learner.py:31-33) and `token_embedding` maps them all to one row.  FullModel feeds these
        sequences to a transformer that adds neither position nor mask (reference model_wrapper.py:58,72), so the run is
        merged there (`_TextTransformer.forward(_tail_run=)`).  Measured on the device once per token bank; at most
        76, so that the run never reaches into the context rows."""
        if self._tail_run is None:
            tok = self.stacked_tokens()
            if tok.is_cuda and hasattr(self.token_embedding, "_owner"):
                r = self.token_embedding._owner._text.tail_run(tok)
            else:  # (a token bank that does not live on the GPU: count on the host)
                r = host_tail_run(tok)
            self._tail_run = max(1, min(r, tok.shape[1] - 1))
        return self._tail_run

    def forward(self) -> torch.Tensor:
        """[n_cls, P + 77, D]: context tokens first, then the class prompt's token embeddings."""
        return torch.cat([self.stacked_context(), self.stacked_tokens()], dim=1)

    @property
    def n_cls(self) -> int:
        return len(self.context_bank)
