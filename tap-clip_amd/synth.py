"""Deterministic synthetic weights / inputs in open_clip's state-dict key layout.

There is no network and no CLIP checkpoint in this project (the reference loads a local
`open_clip_pytorch_model.bin`, reference models/clip_wrapper.py:14-15), so the benchmark, the
smoke test and the large-shape fixtures use seeded random weights of the right architecture.

The generator is a counter-based integer hash (splitmix64) turned into an approximately normal
variate by summing eight 16-bit uniforms (Irwin-Hall): integer arithmetic plus one correctly
rounded float64 multiply, so the same (seed, key) gives bit-identical tensors on any host -- the
build container that writes `tests/golden/` and the GPU box that replays it.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from .configs import ClipDims

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


_IH_MEAN = 8 * 32767.5
_IH_STD = math.sqrt(8 * (65536.0**2 - 1.0) / 12.0)


def normal(shape: Sequence[int], seed: int, key: str, std: float = 1.0, mean: float = 0.0) -> torch.Tensor:
    """Approximately N(mean, std^2) fp32 tensor, a pure function of (seed, key, shape)."""
    n = int(np.prod(shape)) if len(shape) else 1
    stream = np.uint64(zlib.crc32(key.encode()) & 0xFFFFFFFF) << np.uint64(32)
    base = _splitmix64(np.array([seed], dtype=np.uint64))[0] ^ stream
    out = np.empty(n, dtype=np.float32)
    step = 1 << 22
    with np.errstate(over="ignore"):
        for lo in range(0, n, step):
            hi = min(n, lo + step)
            ctr = (np.arange(lo, hi, dtype=np.uint64) * np.uint64(2) + base) & _M64
            a = _splitmix64(ctr)
            b = _splitmix64(ctr + np.uint64(1))
            tot = np.zeros(hi - lo, dtype=np.int64)
            for w in (a, b):
                for sh in (0, 16, 32, 48):
                    tot += ((w >> np.uint64(sh)) & np.uint64(0xFFFF)).astype(np.int64)
            z = (tot.astype(np.float64) - _IH_MEAN) * (1.0 / _IH_STD)
            out[lo:hi] = (z * std + mean).astype(np.float32)
    return torch.from_numpy(out.reshape(tuple(shape)))


def integers(shape: Sequence[int], seed: int, key: str, high: int) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    stream = np.uint64(zlib.crc32(key.encode()) & 0xFFFFFFFF) << np.uint64(32)
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([seed], dtype=np.uint64))[0] ^ stream
        v = _splitmix64((np.arange(n, dtype=np.uint64) + base) & _M64)
    return torch.from_numpy((v % np.uint64(high)).astype(np.int64).reshape(tuple(shape)))


def _tower(sd: Dict[str, torch.Tensor], prefix: str, width: int, layers: int, mlp: int, seed: int) -> None:
    attn_std = width**-0.5
    proj_std = (width**-0.5) * ((2 * layers) ** -0.5)
    fc_std = (2 * width) ** -0.5
    for i in range(layers):
        p = f"{prefix}resblocks.{i}."
        # LN affine perturbed away from (1, 0) and non-zero biases so every epilogue term is exercised
        sd[p + "ln_1.weight"] = normal([width], seed, p + "ln_1.weight", 0.1, 1.0)
        sd[p + "ln_1.bias"] = normal([width], seed, p + "ln_1.bias", 0.05)
        sd[p + "attn.in_proj_weight"] = normal([3 * width, width], seed, p + "attn.in_proj_weight", attn_std)
        sd[p + "attn.in_proj_bias"] = normal([3 * width], seed, p + "attn.in_proj_bias", 0.02)
        sd[p + "attn.out_proj.weight"] = normal([width, width], seed, p + "attn.out_proj.weight", proj_std)
        sd[p + "attn.out_proj.bias"] = normal([width], seed, p + "attn.out_proj.bias", 0.02)
        sd[p + "ln_2.weight"] = normal([width], seed, p + "ln_2.weight", 0.1, 1.0)
        sd[p + "ln_2.bias"] = normal([width], seed, p + "ln_2.bias", 0.05)
        sd[p + "mlp.c_fc.weight"] = normal([mlp, width], seed, p + "mlp.c_fc.weight", fc_std)
        sd[p + "mlp.c_fc.bias"] = normal([mlp], seed, p + "mlp.c_fc.bias", 0.02)
        sd[p + "mlp.c_proj.weight"] = normal([width, mlp], seed, p + "mlp.c_proj.weight", proj_std)
        sd[p + "mlp.c_proj.bias"] = normal([width], seed, p + "mlp.c_proj.bias", 0.02)


def make_state_dict(cfg: ClipDims, seed: int = 2, vision: bool = True, text: bool = True) -> Dict[str, torch.Tensor]:
    """Seeded random CLIP state dict with open_clip key names."""
    sd: Dict[str, torch.Tensor] = {}
    if vision:
        w, p = cfg.vision.width, cfg.patch
        scale = w**-0.5
        sd["visual.conv1.weight"] = normal([w, 3, p, p], seed, "visual.conv1.weight", (3 * p * p) ** -0.5)
        sd["visual.class_embedding"] = normal([w], seed, "visual.class_embedding", scale)
        sd["visual.positional_embedding"] = normal([cfg.n_tokens, w], seed, "visual.positional_embedding", scale)
        sd["visual.ln_pre.weight"] = normal([w], seed, "visual.ln_pre.weight", 0.1, 1.0)
        sd["visual.ln_pre.bias"] = normal([w], seed, "visual.ln_pre.bias", 0.05)
        _tower(sd, "visual.transformer.", w, cfg.vision.layers, cfg.vision.mlp, seed)
        sd["visual.ln_post.weight"] = normal([w], seed, "visual.ln_post.weight", 0.1, 1.0)
        sd["visual.ln_post.bias"] = normal([w], seed, "visual.ln_post.bias", 0.05)
        sd["visual.proj"] = normal([w, cfg.embed_dim], seed, "visual.proj", scale)
    if text:
        w = cfg.text.width
        sd["token_embedding.weight"] = normal([cfg.vocab, w], seed, "token_embedding.weight", 0.02)
        sd["positional_embedding"] = normal([cfg.ctx, w], seed, "positional_embedding", 0.01)
        _tower(sd, "transformer.", w, cfg.text.layers, cfg.text.mlp, seed)
        sd["ln_final.weight"] = normal([w], seed, "ln_final.weight", 0.1, 1.0)
        sd["ln_final.bias"] = normal([w], seed, "ln_final.bias", 0.05)
        sd["text_projection"] = normal([w, cfg.embed_dim], seed, "text_projection", w**-0.5)
        sd["logit_scale"] = torch.tensor(math.log(1 / 0.07), dtype=torch.float32)
    return sd


def make_state_dict_device(cfg: ClipDims, seed: int = 2, device="cuda", vision: bool = True, text: bool = True) -> Dict[str, torch.Tensor]:
    """The same keys, shapes and scales as `make_state_dict`, drawn by torch's generator ON `device`: for throughput legs
    that need a large model quickly and compare nothing with a golden (ViT-L/14@336: 428 M parameters are 45 s of the
    host generator, 0.3 s here).  NOT the weights the goldens were made with."""
    g = torch.Generator(device=device).manual_seed(seed)
    global normal
    real = normal
    try:  # run the host layout with a stand-in generator that records (shape, std, mean) only
        normal = lambda shape, seed_, key, std=1.0, mean=0.0: ("spec", tuple(shape), float(std), float(mean))
        host = make_state_dict(cfg, seed, vision, text)
    finally:
        normal = real
    out = {}
    for k, v in host.items():
        if isinstance(v, tuple) and v and v[0] == "spec":
            _, shape, std, mean = v
            out[k] = torch.randn(shape, generator=g, device=device, dtype=torch.float32) * std + mean
        else:
            out[k] = v.to(device)
    return out


def make_images(batch: int, cfg: ClipDims, seed: int = 0) -> torch.Tensor:
    """[B,3,S,S] fp32 ~ N(0,1): stands in for CLIP-normalised pixels."""
    return normal([batch, 3, cfg.image_size, cfg.image_size], seed, "images")


def make_prompts(n_cls: int, prompt_len: int, cfg: ClipDims, seed: int = 1):
    """ctx [n_cls,P,D] ~ N(0,1) (reference models/prompt_learner.py:41 torch.randn) and class-token
    embeddings [n_cls,77,D] ~ N(0,0.02^2) (token-embedding scale)."""
    d = cfg.text.width
    ctx = normal([n_cls, prompt_len, d], seed, "ctx")
    tok = normal([n_cls, cfg.ctx, d], seed, "tok", 0.02)
    return ctx, tok


def make_labels(batch: int, n_cls: int, seed: int = 3) -> torch.Tensor:
    return integers([batch], seed, "labels", n_cls)
