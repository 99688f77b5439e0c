"""Thin Python handles over the C-ABI towers (include/tapclip.h).  PyTorch only supplies device
memory and the current HIP stream; all arithmetic happens in libtapclip.so.

`VisionTower` / `TextTower` replace the open_clip model that reference
models/clip_wrapper.py:13-16 builds and loads."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import torch

from . import _lib
from .configs import ClipDims, TowerDims


def _stream_ptr(device: torch.device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _dev_f32(t: torch.Tensor, device: torch.device) -> torch.Tensor:
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


def _ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


class _Tower:
    kind = -1
    prefix = ""

    def __init__(self, cfg: ClipDims, dims: TowerDims, state_dict: Dict[str, torch.Tensor], device, precision: str):
        if precision not in _lib.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_lib.PRECISIONS)}, got {precision!r}")
        self.lib = _lib.load("fp16" if precision == "fp16" else "bf16")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("tapclip towers run on an AMD GPU only (device must be 'cuda[:i]'); there is no CPU path")
        self.cfg, self.dims, self.precision = cfg, dims, precision
        c = _lib.TowerCfg(
            kind=self.kind, width=dims.width, layers=dims.layers, heads=dims.heads, mlp_dim=dims.mlp,
            embed_dim=cfg.embed_dim, image_size=cfg.image_size, patch=cfg.patch, ctx_len=cfg.ctx, vocab=cfg.vocab,
            act=_lib.ACT_QUICK_GELU if cfg.quick_gelu else _lib.ACT_GELU_ERF, precision=_lib.PRECISIONS[precision],
        )
        h = C.c_void_p()
        self._check(self.lib.tapclip_tower_create(C.byref(c), C.byref(h)))
        self.handle = h
        self._ws: Optional[torch.Tensor] = None
        self._load(state_dict)

    def _check(self, rc: int) -> None:
        _lib.check(rc, self.lib)  # the error text lives in the library variant that produced it

    def set_ksplit(self, on: bool) -> None:
        """`TAPCLIP_FLAG_KSPLIT`: K-split the tiles of partial GEMM rounds over idle CUs (on: the launch finishes sooner --
        a tower that has the GPU to itself; off: fewest CU-seconds -- a tower that shares it with another stream)."""
        self._check(self.lib.tapclip_tower_set_flag(self.handle, _lib.FLAG_KSPLIT, int(bool(on))))

    def get_ksplit(self) -> bool:
        v = C.c_int32(0)
        self._check(self.lib.tapclip_tower_get_flag(self.handle, _lib.FLAG_KSPLIT, C.byref(v)))
        return bool(v.value)

    # -- weights -----------------------------------------------------------------------------
    def _wanted(self, key: str) -> Optional[str]:
        raise NotImplementedError

    def _load(self, state_dict: Dict[str, torch.Tensor]) -> None:
        with torch.cuda.device(self.device):
            stream = _stream_ptr(self.device)
            for key, t in state_dict.items():
                name = self._wanted(key)
                if name is None:
                    continue
                d = _dev_f32(t, self.device)
                shape = (C.c_int64 * max(d.dim(), 1))(*d.shape)
                self._check(self.lib.tapclip_tower_load_weight(self.handle, name.encode(), _ptr(d), shape, d.dim(), stream))
            torch.cuda.current_stream(self.device).synchronize()  # staging tensors may now be freed
        self._check(self.lib.tapclip_tower_ready(self.handle))  # strict=True semantics

    # -- scratch -----------------------------------------------------------------------------
    def workspace(self, n_seq: int, tokens: int, tail_run: int = 1) -> Tuple[torch.Tensor, int]:
        if tail_run > 1:
            need = int(self.lib.tapclip_text_tied_workspace_bytes(self.handle, n_seq, tokens, tail_run))
        else:
            need = int(self.lib.tapclip_tower_workspace_bytes(self.handle, n_seq, tokens))
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws, self._ws.numel()

    # -- per-stage HIP-event timing ------------------------------------------------------------
    def profile(self, on: bool) -> None:
        self._check(self.lib.tapclip_profile_enable(self.handle, int(on)))

    def profile_read(self) -> Dict[str, Tuple[float, int]]:
        ms = (C.c_float * len(_lib.PROFILE_SLOTS))()
        n = (C.c_int64 * len(_lib.PROFILE_SLOTS))()
        self._check(self.lib.tapclip_profile_read(self.handle, ms, n))
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(_lib.PROFILE_SLOTS)}

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.tapclip_tower_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class VisionTower(_Tower):
    kind = _lib.TOWER_VISION

    def __init__(self, cfg: ClipDims, state_dict, device="cuda", precision: str = "bf16", prune_last_block: Optional[bool] = None):
        """prune_last_block: None keeps the library default (on, TAPCLIP_PRUNE_LAST=0 switches it off): the last block
        computes K / V for every token and everything else for the CLS rows only -- `encode_image` returns nothing but
        the pooled CLS row (include/tapclip.h TAPCLIP_FLAG_PRUNE_LAST_BLOCK).  False computes every row of every block."""
        super().__init__(cfg, cfg.vision, state_dict, device, precision)
        if prune_last_block is not None:
            self.set_prune_last_block(prune_last_block)

    def set_prune_last_block(self, on: bool) -> None:
        self._check(self.lib.tapclip_tower_set_flag(self.handle, _lib.FLAG_PRUNE_LAST_BLOCK, int(bool(on))))

    def _wanted(self, key):
        return key[len("visual."):] if key.startswith("visual.") else None

    def encode_image(self, images: torch.Tensor, normalize: bool = False) -> torch.Tensor:
        """`CLIPWrapper.encode_image` (reference models/clip_wrapper.py:46-47): [B,3,S,S] -> [B,E]."""
        s = self.cfg.image_size
        if images.dim() != 4 or tuple(images.shape[1:]) != (3, s, s):
            raise ValueError(f"expected images [B,3,{s},{s}], got {tuple(images.shape)}")
        x = _dev_f32(images, self.device)
        B = x.shape[0]
        out = torch.empty(B, self.cfg.embed_dim, dtype=torch.float32, device=self.device)
        if B == 0:  # an empty batch encodes to an empty [0, E] tensor, as open_clip would
            return out
        with torch.cuda.device(self.device):
            ws, nbytes = self.workspace(B, self.cfg.n_tokens)
            self._check(self.lib.tapclip_encode_image(self.handle, _ptr(x), B, _ptr(out), int(normalize), _ptr(ws),
                                                     nbytes, _stream_ptr(self.device)))
        return out


class TextTower(_Tower):
    kind = _lib.TOWER_TEXT
    _TOP = ("token_embedding.weight", "positional_embedding", "ln_final.weight", "ln_final.bias", "text_projection")

    def __init__(self, cfg: ClipDims, state_dict, device="cuda", precision: str = "bf16"):
        # "fp8" and "fp16" are image-tower precisions (frozen weights, forward only).  The text tower carries the
        # prompt gradients, which need bf16's exponent range: beside an fp8 image tower (a throughput mode) it runs
        # plain bf16; beside the IEEE-half image tower (the mode that holds BASELINE.json's 1e-3 on embeddings AND
        # logits) it runs split-bf16 (three MFMA products, ~2^-16 relative, forward and backward) -- it is a tenth
        # of the image tower's work, and the IEEE-half text tower of round 1 left cfg-1 logits at 1.3e-3.
        precision = {"fp8": "bf16", "fp16": "bf16x3"}.get(precision, precision)
        super().__init__(cfg, cfg.text, state_dict, device, precision)

    def _wanted(self, key):
        if key.startswith("transformer.resblocks.") or key in self._TOP:
            return key
        return None

    @staticmethod
    def _check_tail_run(tail_run: int, T: int, causal: bool) -> int:
        tail_run = int(tail_run)
        if tail_run < 1 or tail_run > T:
            raise ValueError(f"tail_run must be in [1, {T}], got {tail_run}")
        if tail_run > 1 and causal:
            raise ValueError("tail_run > 1 (tied padding rows) has no meaning under the causal mask")
        return tail_run

    def tail_run(self, x: torch.Tensor) -> int:
        """Largest r such that the last r rows of EVERY sequence of x [n,T,D] are bit-identical (>= 1): the run of tied
        padding rows that `forward(..., tail_run=r)` may merge.  Synchronous -- once per token bank, not per step."""
        xin = _dev_f32(x, self.device)
        n, T, D = xin.shape
        r = C.c_int32(0)
        with torch.cuda.device(self.device):
            self._check(self.lib.tapclip_text_tail_run(_ptr(xin), n, T, D, C.byref(r), _stream_ptr(self.device)))
        return int(r.value)

    def tied_violations(self) -> bool:
        """True when a `tail_run` claim made to this tower since the last call was false (its outputs were NaN); clears it."""
        v = C.c_int32(0)
        with torch.cuda.device(self.device):
            self._check(self.lib.tapclip_text_tied_violations(self.handle, C.byref(v), _stream_ptr(self.device)))
        return bool(v.value)

    def forward(self, x: torch.Tensor, causal: bool = False, want_hidden: bool = True, want_heads: bool = False,
                want_mean: bool = False, want_attn_out: bool = False, tail_run: int = 1):
        """`clip.model.transformer(x)` (reference models/model_wrapper.py:58,72) on [n,T,D].
        Returns dict with any of hidden [n,T,D], attn_heads [n,H,T,T], attn_mean [n,T,T], attn_out [n,T,D].
        tail_run > 1: the caller knows the last `tail_run` rows of every sequence to be identical (zero-padded prompts
        without positional embedding); the tower then runs on the distinct rows only (include/tapclip.h "tied padding")."""
        D = self.dims.width
        if x.dim() != 3 or x.shape[-1] != D:
            raise ValueError(f"expected x [n,T,{D}], got {tuple(x.shape)}")
        xin = _dev_f32(x, self.device)
        n, T, _ = xin.shape
        mk = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=self.device)
        hidden = mk(n, T, D) if want_hidden else None
        heads = mk(n, self.dims.heads, T, T) if want_heads else None
        mean = mk(n, T, T) if want_mean else None
        aout = mk(n, T, D) if want_attn_out else None
        tail_run = self._check_tail_run(tail_run, T, causal)
        with torch.cuda.device(self.device):
            ws, nbytes = self.workspace(n, T, tail_run)
            if tail_run > 1:
                self._check(self.lib.tapclip_text_forward_tied(self.handle, _ptr(xin), n, T, tail_run, _ptr(hidden), _ptr(heads),
                                                              _ptr(mean), _ptr(aout), _ptr(ws), nbytes, _stream_ptr(self.device)))
            else:
                self._check(self.lib.tapclip_text_forward(self.handle, _ptr(xin), n, T, int(causal), _ptr(hidden), _ptr(heads),
                                                         _ptr(mean), _ptr(aout), _ptr(ws), nbytes, _stream_ptr(self.device)))
        return {"hidden": hidden, "attn_heads": heads, "attn_mean": mean, "attn_out": aout}

    def pool_project(self, hidden: torch.Tensor, index: Optional[torch.Tensor] = None, ln_final: bool = False,
                     normalize: bool = False) -> torch.Tensor:
        """token pick (-1 or index) [-> ln_final] -> @ text_projection [-> L2 norm]
        (reference models/model_wrapper.py:73-75; encode_text tail)."""
        h = _dev_f32(hidden, self.device)
        n, T, _ = h.shape
        idx = None if index is None else index.to(device=self.device, dtype=torch.int64).contiguous()
        out = torch.empty(n, self.cfg.embed_dim, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.lib.tapclip_text_pool_project(self.handle, _ptr(h), n, T, _ptr(idx), int(ln_final),
                                                          int(normalize), _ptr(out), _stream_ptr(self.device)))
        return out

    # -- prompt-tuning backward (dX only; reference train.py:99-105) ---------------------------------
    def backward(self, x: torch.Tensor, grad_hidden: torch.Tensor, causal: bool = False) -> torch.Tensor:
        """dL/dx of `forward(x)["hidden"]` given dL/d(hidden); recomputes the forward internally."""
        xin = _dev_f32(x, self.device)
        g = _dev_f32(grad_hidden, self.device)
        n, T, D = xin.shape
        out = torch.empty_like(xin)
        with torch.cuda.device(self.device):
            need = int(self.lib.tapclip_text_backward_workspace_bytes(self.handle, n, T))
            if self._ws is None or self._ws.numel() < need:
                self._ws = None
                self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._check(self.lib.tapclip_text_backward(self.handle, _ptr(xin), _ptr(g), n, T, int(causal), _ptr(out),
                                                      _ptr(self._ws), self._ws.numel(), _stream_ptr(self.device)))
        return out

    def forward_saved(self, x: torch.Tensor, causal: bool = False, tail_run: int = 1):
        """Training forward: hidden [n,T,D] plus an opaque buffer of saved activations for `backward_saved`
        (no attention write-back; `tapclip_text_forward_saved` / `_tied`: the same `tail_run` goes to `backward_saved`)."""
        xin = _dev_f32(x, self.device)
        n, T, D = xin.shape
        tail_run = self._check_tail_run(tail_run, T, causal)
        hidden = torch.empty_like(xin)
        with torch.cuda.device(self.device):
            saved = torch.empty(int(self.lib.tapclip_text_saved_bytes(self.handle, n, T - tail_run + 1)), dtype=torch.uint8,
                                device=self.device)
            ws, nbytes = self.workspace(n, T, tail_run)
            if tail_run > 1:
                self._check(self.lib.tapclip_text_forward_saved_tied(self.handle, _ptr(xin), n, T, tail_run, _ptr(hidden), _ptr(saved),
                                                                    saved.numel(), _ptr(ws), nbytes, _stream_ptr(self.device)))
            else:
                self._check(self.lib.tapclip_text_forward_saved(self.handle, _ptr(xin), n, T, int(causal), _ptr(hidden), _ptr(saved),
                                                               saved.numel(), _ptr(ws), nbytes, _stream_ptr(self.device)))
        return hidden, saved

    def backward_saved(self, saved: torch.Tensor, grad_hidden: torch.Tensor, causal: bool = False, tail_run: int = 1) -> torch.Tensor:
        """dL/dx from the activations kept by `forward_saved` (no recomputation; `tapclip_text_backward_saved` / `_tied`).
        tail_run > 1: the tied rows are one variable -- its gradient comes back in the run's first row, zeros behind it."""
        g = _dev_f32(grad_hidden, self.device)
        n, T, D = g.shape
        tail_run = self._check_tail_run(tail_run, T, causal)
        out = torch.empty_like(g)
        with torch.cuda.device(self.device):
            ws, nbytes = self.workspace(n, T, tail_run)
            if tail_run > 1:
                self._check(self.lib.tapclip_text_backward_saved_tied(self.handle, _ptr(saved), saved.numel(), _ptr(g), n, T, tail_run,
                                                                     _ptr(out), _ptr(ws), nbytes, _stream_ptr(self.device)))
            else:
                self._check(self.lib.tapclip_text_backward_saved(self.handle, _ptr(saved), saved.numel(), _ptr(g), n, T, int(causal),
                                                                _ptr(out), _ptr(ws), nbytes, _stream_ptr(self.device)))
        return out

    def pool_project_backward(self, hidden: torch.Tensor, grad_out: torch.Tensor, normalize: bool = True) -> torch.Tensor:
        """backward of `pool_project(hidden, index=None, ln_final=False, normalize)`: [n,E] -> [n,T,D]."""
        h = _dev_f32(hidden, self.device)
        g = _dev_f32(grad_out, self.device)
        n, T, _ = h.shape
        out = torch.empty_like(h)
        with torch.cuda.device(self.device):
            self._check(self.lib.tapclip_text_pool_project_backward(self.handle, _ptr(h), n, T, int(normalize), _ptr(g),
                                                                   _ptr(out), _stream_ptr(self.device)))
        return out

    def embed_tokens(self, tokens: torch.Tensor, add_pos: bool) -> torch.Tensor:
        t = tokens.to(device=self.device, dtype=torch.int64).contiguous()
        if t.dim() != 2:
            raise ValueError(f"expected tokens [n,L], got {tuple(t.shape)}")
        n, L = t.shape
        out = torch.empty(n, L, self.dims.width, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.lib.tapclip_embed_tokens(self.handle, _ptr(t), n, L, int(add_pos), _ptr(out), _stream_ptr(self.device)))
        return out


# ---- standalone ops ------------------------------------------------------------------------------
def attribution(attn_map: torch.Tensor, prompt_len: int, normalize: bool = True) -> torch.Tensor:
    """`AttributionMonitor.forward` (reference models/attribution_monitor.py:17-36)."""
    if attn_map.dim() != 3:
        raise ValueError(f"expected attn_map [B,T,T], got {tuple(attn_map.shape)}")
    a = attn_map.detach().to(torch.float32).contiguous()
    n, T, T2 = a.shape
    rows = min(prompt_len, T)
    out = torch.empty(n, rows, dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        _lib.check(_lib.load().tapclip_attribution(_ptr(a), n, T, T2, prompt_len, int(normalize), _ptr(out), _stream_ptr(a.device)))
    return out


def build_prompts(ctx: torch.Tensor, tok: torch.Tensor, attr: Optional[torch.Tensor] = None) -> torch.Tensor:
    """cat([ctx * attr[..., None], tok], dim=1) (reference models/prompt_adjustor.py:35-36,
    models/model_wrapper.py:51,68-69)."""
    c = ctx.detach().to(torch.float32).contiguous()
    t = tok.detach().to(device=c.device, dtype=torch.float32).contiguous()
    n, P, D = c.shape
    L = t.shape[1]
    a = None if attr is None else attr.detach().to(device=c.device, dtype=torch.float32).contiguous()
    out = torch.empty(n, P + L, D, dtype=torch.float32, device=c.device)
    with torch.cuda.device(c.device):
        _lib.check(_lib.load().tapclip_build_prompts(_ptr(c), _ptr(t), _ptr(a), 0 if a is None else a.shape[1], n, P, L, D,
                                                     _ptr(out), _stream_ptr(c.device)))
    return out


def build_prompts_backward(grad_out: torch.Tensor, prompt_len: int, attr: Optional[torch.Tensor] = None) -> torch.Tensor:
    """d(loss)/d(ctx) of `build_prompts`: grad_out[:, :P] * attr[..., None] (reference autograd through
    models/prompt_adjustor.py:35-36 and the torch.cat of models/model_wrapper.py:69; the attribution is detached there)."""
    g = grad_out.detach().to(torch.float32).contiguous()
    n, T, D = g.shape
    P = int(prompt_len)
    a = None if attr is None else attr.detach().to(device=g.device, dtype=torch.float32).contiguous()
    out = torch.empty(n, P, D, dtype=torch.float32, device=g.device)
    with torch.cuda.device(g.device):
        _lib.check(_lib.load().tapclip_build_prompts_backward(_ptr(g), _ptr(a), 0 if a is None else a.shape[1], n, P, T - P, D,
                                                              _ptr(out), _stream_ptr(g.device)))
    return out


def build_prompts_mlp(ctx: torch.Tensor, tok: torch.Tensor, attr: torch.Tensor, adjustor) -> torch.Tensor:
    """cat([PromptAdjustor('gate' | 'residual')(ctx, attr), tok], dim=1) in one kernel (reference
    models/prompt_adjustor.py:38-44, models/model_wrapper.py:68-69); forward only, the adjustor's weights as they are."""
    method = {"gate": _lib.ADJUST_GATE, "residual": _lib.ADJUST_RESIDUAL}[adjustor.method]
    net = adjustor.gate_net if adjustor.method == "gate" else adjustor.residual_net
    c = ctx.detach().to(torch.float32).contiguous()
    dev = c.device
    f = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
    t, a = f(tok), f(attr)
    w1, b1, w2, b2 = f(net[0].weight).view(-1), f(net[0].bias), f(net[2].weight), f(net[2].bias)
    n, P, D = c.shape
    if adjustor.method == "residual" and w2.shape[0] != D:
        raise ValueError(f"PromptAdjustor('residual') produces {w2.shape[0]} columns, the context tokens have {D}")  # (torch would fail to broadcast)
    out = torch.empty(n, P + t.shape[1], D, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().tapclip_build_prompts_mlp(method, _ptr(c), _ptr(t), _ptr(a), a.shape[1], _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2),
                                                         n, P, t.shape[1], D, _ptr(out), _stream_ptr(dev)))
    return out


def logits(img: torch.Tensor, txt: torch.Tensor, scale: float) -> torch.Tensor:
    """scale * img @ txt.T (reference models/model_wrapper.py:79,83)."""
    i = img.detach().to(torch.float32).contiguous()
    t = txt.detach().to(device=i.device, dtype=torch.float32).contiguous()
    B, E = i.shape
    Cn = t.shape[0]
    out = torch.empty(B, Cn, dtype=torch.float32, device=i.device)
    if B == 0 or Cn == 0:
        return out
    with torch.cuda.device(i.device):
        _lib.check(_lib.load().tapclip_logits(_ptr(i), _ptr(t), float(scale), B, Cn, E, _ptr(out), _stream_ptr(i.device)))
    return out


def logits_backward(grad_logits: torch.Tensor, logits_out: torch.Tensor, img: torch.Tensor, scale: float):
    """(d_txt [C,E], d_log_scale []) of `logits(img, txt, scale)` with scale = exp(log_scale)."""
    gl = grad_logits.detach().to(torch.float32).contiguous()
    lo = logits_out.detach().to(torch.float32).contiguous()
    i = img.detach().to(torch.float32).contiguous()
    B, Cn = gl.shape
    E = i.shape[1]
    d_txt = torch.empty(Cn, E, dtype=torch.float32, device=gl.device)
    d_ls = torch.empty((), dtype=torch.float32, device=gl.device)
    with torch.cuda.device(gl.device):
        _lib.check(_lib.load().tapclip_logits_backward(_ptr(gl), _ptr(lo), _ptr(i), float(scale), B, Cn, E, _ptr(d_txt),
                                                       _ptr(d_ls), _stream_ptr(gl.device)))
    return d_txt, d_ls


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor) -> torch.Tensor:
    xx = x.detach().to(torch.float32).contiguous()
    d = xx.shape[-1]
    rows = xx.numel() // d
    y = torch.empty_like(xx)
    g = gamma.detach().to(device=xx.device, dtype=torch.float32).contiguous()
    b = beta.detach().to(device=xx.device, dtype=torch.float32).contiguous()
    with torch.cuda.device(xx.device):
        _lib.check(_lib.load().tapclip_layernorm_f32(_ptr(xx), _ptr(g), _ptr(b), rows, d, _ptr(y), _stream_ptr(xx.device)))
    return y


def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, precision: str = "bf16") -> torch.Tensor:
    """a[M,K] @ w[N,K].T + bias through the MFMA GEMM kernel (unit parity tests / roofline bench)."""
    aa = a.detach().to(torch.float32).contiguous()
    ww = w.detach().to(device=aa.device, dtype=torch.float32).contiguous()
    M, K = aa.shape
    N = ww.shape[0]
    bb = None if bias is None else bias.detach().to(device=aa.device, dtype=torch.float32).contiguous()
    lib = _lib.load()
    nbytes = int(lib.tapclip_gemm_scratch_bytes(M, N, K))
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=aa.device)
    out = torch.empty(M, N, dtype=torch.float32, device=aa.device)
    with torch.cuda.device(aa.device):
        _lib.check(lib.tapclip_gemm_f32(_ptr(aa), _ptr(ww), _ptr(bb), M, N, K, _lib.PRECISIONS[precision], _ptr(out),
                                        _ptr(scratch), nbytes, _stream_ptr(aa.device)))
    return out


CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def preprocess_u8(images, size: int = 224, device="cuda", mean=CLIP_MEAN, std=CLIP_STD) -> torch.Tensor:
    """CLIP's eval transform on the GPU (include/tapclip.h tapclip_preprocess_u8; the reference applies
    `clip.get_preprocess()` per sample on the CPU, dataset.py:29-35): a list of decoded RGB images -- uint8
    [h, w, 3] tensors / arrays or PIL images, any sizes -> [B, 3, size, size] fp32 on `device`.  Bit-identical
    to Pillow's bicubic resize + torchvision's CenterCrop / ToTensor / Normalize."""
    import numpy as np

    dev = torch.device(device)
    flat, dims = [], []
    for im in images:
        if not torch.is_tensor(im):
            im = torch.from_numpy(np.array(im.convert("RGB") if hasattr(im, "convert") else im))
        if im.dtype != torch.uint8 or im.dim() != 3 or im.shape[2] != 3:
            raise ValueError(f"preprocess_u8 takes uint8 [h, w, 3] images, got {im.dtype} {tuple(im.shape)}")
        if im.shape[0] == 0 or im.shape[1] == 0:
            raise ValueError("preprocess_u8: empty image")
        dims.append((int(im.shape[0]), int(im.shape[1])))
        flat.append(im.contiguous().view(-1))
    if not flat:
        raise ValueError("preprocess_u8: no images")
    on_dev = all(f.device == dev for f in flat)
    desc, pix_off, ws_off = [], 0, 0
    if on_dev:  # no packing copy: the descriptor's offsets are address differences to the first image
        pixels = flat[0]
        offs = [f.data_ptr() - pixels.data_ptr() for f in flat]
    else:
        pixels = torch.cat([f.cpu() for f in flat]).to(dev, non_blocking=True)
        offs = []
        for h, w in dims:
            offs.append(pix_off)
            pix_off += h * w * 3
    for (h, w), o in zip(dims, offs):
        desc.append((o, h, w, ws_off))
        ws_off += h * size * 3
    desc_t = torch.tensor(desc, dtype=torch.int64).to(dev, non_blocking=True)
    ws = torch.empty(ws_off, dtype=torch.uint8, device=dev)
    out = torch.empty(len(dims), 3, size, size, dtype=torch.float32, device=dev)
    ms = (C.c_float * 6)(*[float(v) for v in mean], *[float(v) for v in std])
    with torch.cuda.device(dev):
        _lib.check(_lib.load().tapclip_preprocess_u8(_ptr(pixels), _ptr(desc_t), len(dims), size, ms, _ptr(ws), _ptr(out),
                                                     _stream_ptr(dev)))
    return out


def mx8_quantize(x: torch.Tensor):
    """fp32 [rows, K] -> (e4m3 bytes [rows, K], e8m0 scale bytes [K/64, rows_pad, 2]): the fp8 path's operand
    format (include/tapclip.h tapclip_mx8_quantize)."""
    xx = x.detach().to(torch.float32).contiguous()
    rows, K = xx.shape
    rows_pad = (rows + 7) // 8 * 8
    q = torch.empty(rows, K, dtype=torch.uint8, device=xx.device)
    sc = torch.zeros(K // 64, rows_pad, 2, dtype=torch.uint8, device=xx.device)
    with torch.cuda.device(xx.device):
        _lib.check(_lib.load().tapclip_mx8_quantize(_ptr(xx), rows, K, _ptr(q), _ptr(sc), rows_pad, _stream_ptr(xx.device)))
    return q, sc


def mx8_gemm(a_q: torch.Tensor, a_scale: torch.Tensor, w_q: torch.Tensor, w_scale: torch.Tensor,
             bias: Optional[torch.Tensor] = None, epilogue: str = "f32", act: int = 0):
    """dequant(a) @ dequant(w).T + bias on the MXFP8 MFMA kernel.  epilogue "f32": fp32 [M, N];
    "gelu_mx8": (e4m3 bytes [M, N], scale bytes [N/64, m_pad, 2]) of act(.) re-quantised (the c_fc epilogue)."""
    M, K = a_q.shape
    N = w_q.shape[0]
    m_pad = a_scale.shape[1]
    dev = a_q.device
    bb = None if bias is None else bias.detach().to(device=dev, dtype=torch.float32).contiguous()
    lib = _lib.load()
    with torch.cuda.device(dev):
        if epilogue == "f32":
            out = torch.empty(M, N, dtype=torch.float32, device=dev)
            _lib.check(lib.tapclip_mx8_gemm(_ptr(a_q), _ptr(a_scale), M, m_pad, _ptr(w_q), _ptr(w_scale), _ptr(bb), N, K, 0, act,
                                            _ptr(out), None, None, _stream_ptr(dev)))
            return out
        if epilogue != "gelu_mx8":
            raise ValueError(f"unknown mx8 epilogue {epilogue!r}")
        oq = torch.empty(M, N, dtype=torch.uint8, device=dev)
        osc = torch.zeros(N // 64, m_pad, 2, dtype=torch.uint8, device=dev)
        _lib.check(lib.tapclip_mx8_gemm(_ptr(a_q), _ptr(a_scale), M, m_pad, _ptr(w_q), _ptr(w_scale), _ptr(bb), N, K, 1, act,
                                        None, _ptr(oq), _ptr(osc), _stream_ptr(dev)))
        return oq, osc
