"""CLIP model dimensions served by the HIP towers (open_clip model_configs, restated).

The reference selects a model by open_clip name (reference models/clip_wrapper.py:10,13:
`CLIPWrapper(model_name='ViT-B-32', ...)`)."""
from dataclasses import dataclass
from typing import Dict


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
    "ViT-B-32-quickgelu": ClipDims("ViT-B-32-quickgelu", 512, 224, 32, TowerDims(768, 12, 12, 3072),
                                   TowerDims(512, 12, 8, 2048), quick_gelu=True),
    "ViT-B-16-quickgelu": ClipDims("ViT-B-16-quickgelu", 512, 224, 16, TowerDims(768, 12, 12, 3072),
                                   TowerDims(512, 12, 8, 2048), quick_gelu=True),
    "ViT-L-14-336": ClipDims("ViT-L-14-336", 768, 336, 14, TowerDims(1024, 24, 16, 4096), TowerDims(768, 12, 12, 3072)),
    # small test model (kernel-legal dims: width % 128 == 0, head dim 64); not an open_clip model
    "tiny": ClipDims("tiny", 64, 32, 8, TowerDims(128, 2, 2, 256), TowerDims(128, 2, 2, 256), vocab=97, ctx=77),
}


def get_config(name: str) -> ClipDims:
    try:
        return CONFIGS[name]
    except KeyError:
        raise ValueError(f"unknown model '{name}'; known: {sorted(CONFIGS)}") from None
