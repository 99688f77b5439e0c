"""Attribution scores of the learnable context tokens, computed by a HIP kernel.

Drop-in for reference models/attribution_monitor.py:7-36: takes the (head-mean) attention map
`[B, T, T]`, reads how strongly each of the first `prompt_len` rows attends to the last token
(column `T - 1`) and soft-maxes those scores over the context tokens."""
import torch
import torch.nn as nn

from .. import engine


class AttributionMonitor(nn.Module):
    def __init__(self, prompt_len, normalize=True):
        super().__init__()
        self.prompt_len = int(prompt_len)
        self.normalize = bool(normalize)

    def forward(self, attn_map: torch.Tensor) -> torch.Tensor:
        """attn_map [B, T, T'] -> [B, min(prompt_len, T)] (`tapclip_attribution`)."""
        return engine.attribution(attn_map, self.prompt_len, self.normalize)

    def extra_repr(self) -> str:
        return f"prompt_len={self.prompt_len}, normalize={self.normalize}"
