"""Experiment (CPU, oracle only; NOT collected by pytest, not product code): which 16-bit roundings of the text tower move
FullModel's logits, at BASELINE configs[2] (ViT-B/16 dims, 65 classes, P = 16, T = 93).

    python tests/probe_text_precision.py [model] [n_cls]

The oracle's `emulate` rounds every operand at the kernels' rounding points; here the rounding is switched per SITE and
per PASS, so that the cost of running a pass / a GEMM family on ONE 16-bit MFMA product (instead of the three products of
the split-bf16 mode) is known before a kernel is written.  Round-4 result: DESIGN.md section 2 "what the text tower's
three products buy".
"""
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tap_clip_amd  # noqa: E402,F401
from oracle import clip_ref, full_model_ref  # noqa: E402
from tap_clip_amd import synth  # noqa: E402

# the 14 rounding sites of clip_ref.block_forward, in call order
SITES = ["qkv_A", "qkv_W", "q", "k", "p", "v", "out_A", "out_W", "branch_a", "fc_A", "fc_W", "proj_A", "proj_W", "branch_m"]
_state = {"i": 0, "sites": frozenset(), "dtype": torch.float16}


def _round(x, emulate):
    site = SITES[_state["i"] % len(SITES)]
    _state["i"] += 1
    if emulate is None or site not in _state["sites"]:
        return x
    return x.to(_state["dtype"]).to(x.dtype)


clip_ref._rb = _round
clip_ref._rq = _round
clip_ref._act = lambda x, quick, emulate=None: (x * torch.sigmoid(1.702 * x)) if quick else torch.nn.functional.gelu(x)


def text_features(prompts, P, sd, cfg, sites1, sites2, dtype):
    """collapsed text side with rounding sites per pass (oracle/full_model_ref.py::text_features, restated for the probe)"""
    _state["dtype"] = dtype
    _state["i"], _state["sites"] = 0, frozenset(sites1)
    _, probs, _ = clip_ref.text_transformer_raw(prompts, sd, cfg, "x" if sites1 else None, want_probs=True)
    amap = probs.mean(dim=1)
    attr = full_model_ref.attribution_from_map(amap, P)
    adjusted = torch.cat([full_model_ref.adjust_scale(prompts[:, :P], attr), prompts[:, P:]], dim=1)
    _state["i"], _state["sites"] = 0, frozenset(sites2)
    hidden, _, _ = clip_ref.text_transformer_raw(adjusted, sd, cfg, "x" if sites2 else None)
    feat = hidden[:, -1, :] @ sd["text_projection"]
    return feat / feat.norm(dim=-1, keepdim=True), amap, attr


def main():
    model = sys.argv[1] if len(sys.argv) > 1 else "ViT-B-16"
    n_cls = int(sys.argv[2]) if len(sys.argv) > 2 else 65
    P = 16
    cfg = clip_ref.CONFIGS[model]
    sd = synth.make_state_dict(cfg, seed=2)
    torch.manual_seed(0)
    ctx, tok = synth.make_prompts(n_cls, P, cfg, seed=1)
    prompts = torch.cat([ctx, tok], dim=1)
    img = torch.randn(32, cfg.embed_dim)
    img = img / img.norm(dim=-1, keepdim=True)
    scale = math.exp(math.log(1 / 0.07))
    ALL = set(SITES)
    GEMM_A = {"qkv_A", "out_A", "fc_A", "proj_A"}
    GEMM_W = {"qkv_W", "out_W", "fc_W", "proj_W"}
    ATT = {"q", "k", "p", "v"}
    BR = {"branch_a", "branch_m"}
    with torch.no_grad():
        t0 = time.time()
        ref, amap0, attr0 = text_features(prompts, P, sd, cfg, (), (), torch.float16)
        print(f"fp32 reference: {time.time() - t0:.1f} s per text side", flush=True)
        lref = scale * img @ ref.t()

        def report(name, s1, s2, dtype=torch.float16):
            f, amap, attr = text_features(prompts, P, sd, cfg, s1, s2, dtype)
            lg = scale * img @ f.t()
            e_l = float((lg - lref).abs().max() / lref.abs().max())
            e_f = float((f - ref).abs().max() / ref.abs().max())
            e_m = float((amap - amap0).abs().max() / amap0.abs().max())
            e_a = float((attr - attr0).abs().max() / attr0.abs().max())
            print(f"{name:58s} logits {e_l:.2e}  text_feat {e_f:.2e}  map {e_m:.2e}  attribution {e_a:.2e}", flush=True)

        report("fp16 everywhere (the half text tower)", ALL, ALL)
        report("bf16 everywhere", ALL, ALL, torch.bfloat16)
        report("pass 1 fp16, pass 2 exact", ALL, ())
        report("pass 1 bf16, pass 2 exact", ALL, (), torch.bfloat16)
        report("pass 1 exact, pass 2 fp16", (), ALL)
        for label, s in (("GEMM A operands", GEMM_A), ("GEMM W operands", GEMM_W), ("attention q k p v", ATT), ("branches", BR),
                         ("qkv GEMM (A+W)", {"qkv_A", "qkv_W"}), ("out_proj GEMM", {"out_A", "out_W"}), ("c_fc GEMM", {"fc_A", "fc_W"}),
                         ("c_proj GEMM", {"proj_A", "proj_W"}), ("everything but the GEMM operands", ATT | BR),
                         ("W operands + attention + branches (A exact)", GEMM_W | ATT | BR),
                         ("A operands + attention + branches (W exact)", GEMM_A | ATT | BR)):
            report(f"pass 2 fp16 at: {label}", (), s)


if __name__ == "__main__":
    main()
