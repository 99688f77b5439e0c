"""Generate tests/golden/*.npz by running the REFERENCE's own classes in this container.

Run from the repo root (needs /root/reference, so only in the build container):
    python -m oracle.make_golden [--only NAME]

What runs: the reference's FullModel / PromptLearner / AttributionMonitor / PromptAdjustor,
imported unmodified from /root/reference (oracle/ref_harness.py explains the harness: `RefClip`
stands where the reference's CLIPWrapper instance stands, built from real torch.nn modules;
PromptLearner's device default is patched to 'cpu').  Weights come from the build's own
deterministic generator (tap-clip_amd/synth.py), so the large cases commit only seeds + outputs.
The reference never travels to the GPU box; these fixtures and this script do.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import tap_clip_amd  # noqa: E402
from oracle import clip_ref, ref_harness  # noqa: E402
from tap_clip_amd import synth  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def _save(name, **arrays):
    path = os.path.join(GOLDEN, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrays.items()})
    print(f"  wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def class_list(n):
    base = ["Backpack", "Alarm_Clock", "Laptop", "Pen", "Mug", "Bike", "Bottle", "Chair", "Desk_Lamp", "Keyboard"]
    return [base[i % len(base)] + ("" if i < len(base) else f"_{i}") for i in range(n)]


def token_table(class_names, cfg, seed=5):
    """synthetic [1,77] token ids per prompt text: SOT, a few 'word' ids, EOT (= vocab-1), zero pad"""
    table = {}
    for i, c in enumerate(class_names):
        ids = torch.zeros(1, cfg.ctx, dtype=torch.long)
        n_words = 5 + (i % 3)
        body = synth.integers([n_words], seed, f"tok.{c}", cfg.vocab - 3) + 1
        ids[0, 0] = cfg.vocab - 2
        ids[0, 1: 1 + n_words] = body
        ids[0, 1 + n_words] = cfg.vocab - 1
        table[f"a photo of a {c}"] = ids
    return table


# ---------------------------------------------------------------------------------------------
def g_attribution_monitor(ref):
    torch.manual_seed(11)
    am = ref["AttributionMonitor"]
    a = torch.softmax(torch.randn(4, 93, 93), dim=-1)
    lit = torch.randn(4, 1, 512)
    _save("attribution_monitor", attn_map=a, out_p16=am(16)(a), out_p16_raw=am(16, normalize=False)(a),
          out_p5=am(5)(a), literal_in=lit, literal_out_p5=am(5)(lit))


def g_prompt_adjustor(ref):
    torch.manual_seed(12)
    pa = ref["PromptAdjustor"]("scale")
    p = torch.randn(3, 16, 512)
    a = torch.softmax(torch.randn(3, 16), dim=-1)
    a1 = torch.ones(3, 1)
    _save("prompt_adjustor", prompt=p, attribution=a, out=pa(p, a), attribution_b1=a1, out_b1=pa(p, a1))


def _full_model_case(ref, cfg, sd, class_names, prompt_len, B, semantics, seed_img, with_grads=True):
    table = token_table(class_names, cfg)
    clip = ref_harness.RefClip(cfg, sd, semantics, ref_harness.FixedTokenizer(table))
    torch.manual_seed(1234)  # PromptLearner draws its context with torch.randn (prompt_learner.py:41)
    model = ref["FullModel"](class_names, clip, prompt_len=prompt_len, adjustor_method="scale", class_specific=True)
    images = synth.make_images(B, cfg, seed_img)
    labels = synth.make_labels(B, len(class_names))
    out = model(images, labels)
    arrays = dict(
        logits=out["logits"], loss=out["loss"], labels=labels,
        token_ids=torch.cat([table[f"a photo of a {c}"] for c in class_names], 0),
        context=torch.stack([model.prompt_learner.context_bank[c].detach() for c in class_names], 0),
        prompts=model.prompt_learner().detach(),
        last_attention_capture=clip.get_attention_map(),
    )
    if with_grads:
        out["loss"].backward()
        arrays["context_grad"] = torch.stack([model.prompt_learner.context_bank[c].grad for c in class_names], 0)
        arrays["logit_scale_grad"] = model.logit_scale.grad
    arrays["state_dict_keys"] = np.array(sorted(model.state_dict().keys()))
    return arrays, model, clip


def g_fullmodel_tiny(ref):
    cfg = clip_ref.CONFIGS["tiny"]
    sd = synth.make_state_dict(cfg, seed=2)
    names = class_list(3)
    for semantics in ("literal", "intended"):
        arrays, model, clip = _full_model_case(ref, cfg, sd, names, 5, 4, semantics, seed_img=0)
        # attention map / attribution of a batch pass over the raw prompts (intended only: [n,T,T])
        if semantics == "intended":
            clip.reset()
            clip.model.transformer(model.prompt_learner().detach())
            amap = clip.get_attention_map()
            arrays["attn_map"] = amap
            arrays["attribution"] = model.attribution_monitor(amap)
        _save(f"fullmodel_{semantics}_tiny", seed_weights=2, seed_images=0, batch=4, prompt_len=5,
              class_names=np.array(names), **arrays)


def g_fullmodel_b32(ref):
    """BASELINE.json configs[0]: ViT-B/32, batch 8, 10 classes, prompt_len 5, CPU, literal loop."""
    cfg = clip_ref.CONFIGS["ViT-B-32"]
    t0 = time.time()
    sd = synth.make_state_dict(cfg, seed=2)
    names = class_list(10)
    for semantics in ("literal", "intended"):
        arrays, model, clip = _full_model_case(ref, cfg, sd, names, 5, 8, semantics, seed_img=0, with_grads=False)
        arrays.pop("prompts")  # [10,82,512]: regenerated from context + token ids
        _save(f"fullmodel_{semantics}_vitb32", seed_weights=2, seed_images=0, batch=8, prompt_len=5,
              class_names=np.array(names), **arrays)
        print(f"  {semantics}: {time.time() - t0:.0f}s")


def g_block_real_dims(ref):
    """One residual block at the real widths through the harness's torch.nn modules."""
    for tag, d, heads, mlp, n, T in (("vision", 768, 12, 3072, 1, 197), ("text", 512, 8, 2048, 2, 93)):
        cfg = clip_ref.ClipDims("blk", 512, 224, 16, clip_ref.TowerDims(d, 1, heads, mlp), clip_ref.TowerDims(d, 1, heads, mlp))
        sd = {}
        synth._tower(sd, "transformer.", d, 1, mlp, seed=7)
        blk = ref_harness._Block(d, heads, mlp, False)
        blk.load_state_dict({k[len("transformer.resblocks.0."):]: v for k, v in sd.items()}, strict=True)
        blk.eval()
        blk.need_weights = True
        x = synth.normal([n, T, d], 8, f"block.{tag}.x")
        cap = {}
        blk.attn.register_forward_hook(lambda m, i, o: cap.update(attn_out=o[0].detach(), probs=o[1].detach()))
        with torch.no_grad():
            y = blk(x)
        _save(f"block_{tag}", seed_weights=7, seed_x=8, n=n, T=T, d=d, heads=heads, mlp=mlp, out=y,
              attn_out=cap["attn_out"], probs_head_mean=cap["probs"].mean(dim=1),
              probs_head0_rows=cap["probs"][:, 0, :8, :])


def g_image_tower(ref):
    """encode_image of the oracle restatement at ViT-B/16 and ViT-B/32 (seeded weights, B=2):
    regression pin of oracle/clip_ref.py itself (parity unpinned against open_clip)."""
    for name in ("ViT-B-16", "ViT-B-32"):
        cfg = clip_ref.CONFIGS[name]
        sd = synth.make_state_dict(cfg, seed=2, text=False)
        images = synth.make_images(2, cfg, 0)
        with torch.no_grad():
            emb = clip_ref.encode_image(images, sd, cfg)
        _save(f"image_tower_{name}", seed_weights=2, seed_images=0, batch=2, embeddings=emb)


ALL = {"attribution_monitor": g_attribution_monitor, "prompt_adjustor": g_prompt_adjustor,
       "fullmodel_tiny": g_fullmodel_tiny, "block_real_dims": g_block_real_dims, "image_tower": g_image_tower,
       "fullmodel_b32": g_fullmodel_b32}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    os.makedirs(GOLDEN, exist_ok=True)
    torch.set_num_threads(os.cpu_count())
    with ref_harness.reference_modules() as ref:
        for name, fn in ALL.items():
            if args.only and args.only != name:
                continue
            print(name)
            fn(ref)


if __name__ == "__main__":
    main()
