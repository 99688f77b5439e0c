"""Attribution-guided adjustment of the context tokens.

Drop-in for reference models/prompt_adjustor.py:6-47.  'scale' (the only method any reference
script selects: train.py:61, test_cross_domain.py:40, test_cross_domain2.py:78) multiplies every
context token by its attribution score; in `FullModel` it is fused with the prompt concatenation
into one HIP kernel (`tapclip_build_prompts`).  'gate' and 'residual' keep the reference's small
MLPs (1->64->1 sigmoid gate; 1->64->512 residual) as torch modules -- they hold trainable weights -- and
`FullModel`'s no-grad forward evaluates them in one HIP kernel too (`tapclip_build_prompts_mlp`); a pass
that differentiates goes through the modules."""
import torch
import torch.nn as nn

_METHODS = ("scale", "gate", "residual")


class PromptAdjustor(nn.Module):
    def __init__(self, method="scale"):
        super().__init__()
        self.method = method
        if method == "gate":
            self.gate_net = nn.Sequential(nn.Linear(1, 64), nn.ReLU(), nn.Linear(64, 1), nn.Sigmoid())
        elif method == "residual":
            # output width 512 is hard-coded in the reference (prompt_adjustor.py:24)
            self.residual_net = nn.Sequential(nn.Linear(1, 64), nn.ReLU(), nn.Linear(64, 512))

    def forward(self, prompt_embed: torch.Tensor, attribution_score: torch.Tensor) -> torch.Tensor:
        """prompt_embed [B, P, D], attribution_score [B, P] (or [B, 1]) -> [B, P, D]."""
        if self.method not in _METHODS:
            raise ValueError(f"Unknown method: {self.method}")
        score = attribution_score.to(prompt_embed.device).unsqueeze(-1)
        if self.method == "scale":
            return prompt_embed * score
        if self.method == "gate":
            return prompt_embed * self.gate_net(score)
        return prompt_embed + self.residual_net(score)
