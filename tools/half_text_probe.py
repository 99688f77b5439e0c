"""Experiment: FullModel logits error against the reference goldens when the text tower of the fp16 mode is IEEE half
(one MFMA product) instead of split-bf16 (three).  Not part of the library."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import tap_clip_amd
from tap_clip_amd import engine
from conftest import golden, rel_max, rel_l2
from test_gpu_parity import _build_full, DEV


class HalfText(engine.TextTower):
    def __init__(self, cfg, sd, dev):
        engine._Tower.__init__(self, cfg, cfg.text, sd, dev, "fp16")


for name, cfgname in (("fullmodel_intended_vitb16_c65", "ViT-B-16"), ("fullmodel_intended_tiny", "tiny"), ("fullmodel_intended_vitb32", "ViT-B-32"), ("fullmodel_intended_vitl14", "ViT-L-14-336")):
    try:
        g = golden(name)
    except Exception as e:
        print("no golden", name, e); continue
    ref = torch.from_numpy(g["logits"])
    for mode in ("bf16x3-text", "half-text"):
        model, images = _build_full(cfgname, g, "intended", "fp16")
        if mode == "half-text":
            sd = {k: v.detach() for k, v in model.clip.model.state_dict().items()}
            model.clip._text = HalfText(model.clip.cfg, sd, DEV)
        with torch.no_grad():
            out = model(images, torch.from_numpy(g["labels"]).to(DEV))
        lg = out["logits"].cpu()
        amap = model.clip.attention_maps[0].cpu()
        extra = ""
        if "attn_map_last_col" in g.files:
            extra = f" map_last_col={rel_max(amap[:, :, -1], torch.from_numpy(g['attn_map_last_col'])):.2e} attribution={rel_max(model.last_attribution.cpu(), torch.from_numpy(g['attribution'])):.2e}"
        print(f"{name} {mode}: logits rel_max={rel_max(lg, ref):.3e} rel_l2={rel_l2(lg, ref):.3e}{extra}", flush=True)
        del model
        torch.cuda.empty_cache()
