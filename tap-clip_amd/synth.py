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


def make_stress_state_dict(cfg: ClipDims, seed: int = 7) -> Dict[str, torch.Tensor]:
    """The seeded state dict of `make_state_dict` bent towards what TRAINED CLIP weights look like and N(0, d^-1/2) draws do
    not (the 1e-3 claim of the IEEE-half mode had met one benign distribution only):
      * LayerNorm gains log-normal (sigma 0.5) with six channels per norm at x10 .. x30;
      * six residual-stream channels per tower driven to |x| = 100 .. 300 by the c_proj bias of block 1 (the "massive
        activations" of trained ViTs: every later LayerNorm is dominated by them);
      * the q and k rows of the LAST block's in_proj scaled so that its scores spread over +-30 (near one-hot softmax rows);
      * c_fc rows of two blocks scaled x4 (pre-activations far into both GELU tails);
      * the read-outs (ln_post gain, text_projection rows) all but ignore the six outlier channels, as trained ones do.
    Same keys and shapes; a pure function of (cfg, seed)."""
    sd = make_state_dict(cfg, seed)
    outliers = {}
    for prefix, dims in (("visual.transformer.", cfg.vision), ("transformer.", cfg.text)):
        w, L = dims.width, dims.layers
        for i in range(L):
            for ln in ("ln_1", "ln_2"):
                k = f"{prefix}resblocks.{i}.{ln}.weight"
                g = torch.exp(normal([w], seed, k + ".stress", 0.5))
                hot = integers([6], seed, k + ".hot", w)
                g[hot] = g[hot] * (10.0 + 20.0 * (integers([6], seed, k + ".hotx", 1000).float() / 999.0))
                sd[k] = g
        hot = integers([6], seed, prefix + "outlier.channels", w)
        mag = 100.0 + 200.0 * (integers([6], seed, prefix + "outlier.mag", 1000).float() / 999.0)
        sign = torch.where(integers([6], seed, prefix + "outlier.sign", 2) == 0, -1.0, 1.0)
        b = sd[f"{prefix}resblocks.1.mlp.c_proj.bias"].clone()
        b[hot] = mag * sign
        sd[f"{prefix}resblocks.1.mlp.c_proj.bias"] = b
        k = f"{prefix}resblocks.{L - 1}.attn.in_proj_weight"
        wq = sd[k].clone()
        wq[: 2 * w] *= 3.2  # q and k rows: scores x 10
        sd[k] = wq
        for i in (2, L - 2):
            k = f"{prefix}resblocks.{i}.mlp.c_fc.weight"
            sd[k] = sd[k] * 4.0
        outliers[prefix] = hot
    for k in ("visual.ln_pre.weight", "visual.ln_post.weight", "ln_final.weight"):
        w = sd[k].numel()
        g = torch.exp(normal([w], seed, k + ".stress", 0.5))
        hot = integers([6], seed, k + ".hot", w)
        g[hot] = g[hot] * 15.0
        sd[k] = g
    # what training does with such channels: the read-outs ignore them (left at full weight, the six constants of +-300 ARE the
    # embedding and every image / class looks alike: a stress case that tests nothing).  The towers' internals keep them.
    sd["visual.ln_post.weight"][outliers["visual.transformer."]] = 0.01
    sd["visual.ln_post.bias"][outliers["visual.transformer."]] = 0.0
    sd["text_projection"][outliers["transformer."]] *= 0.01
    return sd


def make_stress_images(batch: int, cfg: ClipDims, seed: int = 0) -> torch.Tensor:
    """[B,3,S,S]: N(0,1) pixels with a third of the patches SATURATED -- whole 16 x 16 (patch-sized) squares at the CLIP-normalised
    value of pure white (+1.93 / +2.07 / +2.15 per channel) or pure black (-1.79 / -1.75 / -1.48)."""
    x = normal([batch, 3, cfg.image_size, cfg.image_size], seed, "images.stress")
    g = cfg.image_size // cfg.patch
    pick = integers([batch, g, g], seed, "images.stress.pick", 6)  # 0: white, 1: black, else untouched
    white = torch.tensor([1.93, 2.07, 2.15]).view(1, 3, 1, 1)
    black = torch.tensor([-1.79, -1.75, -1.48]).view(1, 3, 1, 1)
    m = pick.repeat_interleave(cfg.patch, 1).repeat_interleave(cfg.patch, 2).unsqueeze(1)
    s_ = g * cfg.patch
    x[:, :, :s_, :s_] = torch.where(m == 0, white, torch.where(m == 1, black, x[:, :, :s_, :s_]))
    return x


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
