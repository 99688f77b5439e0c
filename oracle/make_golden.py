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
    # the two methods no reference script selects (prompt_adjustor.py:13-25,38-44): their small MLPs as the reference's own
    # module initialises and evaluates them -- parameters and outputs committed
    extra = {}
    for method, net in (("gate", "gate_net"), ("residual", "residual_net")):
        torch.manual_seed(13)
        m = ref["PromptAdjustor"](method)
        with torch.no_grad():
            extra[f"{method}_out"] = m(p, a)
            extra[f"{method}_out_b1"] = m(p, a1.expand(3, 16).contiguous())
        seq = getattr(m, net)
        extra[f"{method}_w1"], extra[f"{method}_b1"] = seq[0].weight.detach(), seq[0].bias.detach()
        extra[f"{method}_w2"], extra[f"{method}_b2"] = seq[2].weight.detach(), seq[2].bias.detach()
    _save("prompt_adjustor_mlp", prompt=p, attribution=a, **extra)


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
        extra = {}
        with torch.no_grad():
            emb = clip_ref.encode_image(images, sd, cfg)
            if name == "ViT-B-16":
                extra = _mx8_pair(images, sd, cfg)
        _save(f"image_tower_{name}", seed_weights=2, seed_images=0, batch=2, embeddings=emb, **extra)


def _mx8_pair(images, sd, cfg):
    """The oracle with the block GEMMs' operands rounded to MXFP8 at the kernels' rounding points, accumulating in fp32
    and in fp64: the distance between the two is the floor of agreement for ANY two implementations of that pipeline
    (as clip_ref.emulation_floor is for bf16) -- the fp8 tests hold the HIP tower to 1.5x it instead of a loose constant."""
    e32 = clip_ref.encode_image(images, sd, cfg, emulate="mx8")
    e64 = clip_ref.encode_image(images.double(), {k: v.double() for k, v in sd.items()}, cfg, emulate="mx8")
    floor = float((e32.double() - e64).norm() / e64.norm())
    print(f"  mx8 emulation floor (fp32 vs fp64 accumulation) {floor:.3e}")
    return dict(embeddings_mx8=e32, embeddings_mx8_f64=e64.float(), mx8_floor_rel_l2=floor)


def _seed_context(model, class_names, cfg, prompt_len, seed):
    """Replace the torch.randn context draw by the build's own seeded generator, so that large cases commit a seed
    instead of the [n_cls, P, D] tensor (any N(0,1) sample is the same workload: prompt_learner.py:41)."""
    ctx = synth.make_prompts(len(class_names), prompt_len, cfg, seed=seed)[0]
    with torch.no_grad():
        for i, c in enumerate(class_names):
            model.prompt_learner.context_bank[c].copy_(ctx[i])
    return ctx


def g_fullmodel_b16_c65(ref):
    """BASELINE.json configs[2] at the reference's own loop: ViT-B/16 image + text towers, 65 classes, 16 context
    tokens (T = 93), attention-map capture on, batch 4 (the literal loop is 65 x (B + 1) text passes)."""
    cfg = clip_ref.CONFIGS["ViT-B-16"]
    sd = synth.make_state_dict(cfg, seed=2)
    names = class_list(65)
    P, B, KEEP = 16, 4, 8
    table = token_table(names, cfg)
    images = synth.make_images(B, cfg, 0)
    labels = synth.make_labels(B, len(names))
    for semantics in ("literal", "intended"):
        t0 = time.time()
        clip = ref_harness.RefClip(cfg, sd, semantics, ref_harness.FixedTokenizer(table))
        torch.manual_seed(1234)
        model = ref["FullModel"](names, clip, prompt_len=P, adjustor_method="scale", class_specific=True)
        _seed_context(model, names, cfg, P, seed=1)
        out = model(images, labels)
        out["loss"].backward()
        grads = torch.stack([model.prompt_learner.context_bank[c].grad for c in names], 0)
        arrays = dict(logits=out["logits"], loss=out["loss"], labels=labels,
                      token_ids=torch.cat([table[f"a photo of a {c}"] for c in names], 0),
                      context_grad_head=grads[:KEEP], context_grad_norms=grads.flatten(1).norm(dim=1),
                      logit_scale_grad=model.logit_scale.grad)
        if semantics == "intended":
            # one batch pass over the raw prompts, as the collapsed path does: head-mean map, per-head rows, attribution
            clip.reset()
            cap = {}
            last = clip.model.transformer.resblocks[-1].attn
            h = last.register_forward_hook(lambda m, i, o: cap.update(probs=o[1].detach()))
            with torch.no_grad():
                clip.model.transformer(model.prompt_learner().detach())
            h.remove()
            amap = clip.get_attention_map()                       # [65, 93, 93]
            arrays["attn_map_head"] = amap[:KEEP]                  # the first KEEP classes in full
            arrays["attn_map_last_col"] = amap[:, :, -1]           # column T-1 of every class (what attribution reads)
            arrays["attn_row_sums"] = amap.sum(-1)
            arrays["probs_per_head_rows"] = cap["probs"][:4, :, :4, :]   # [4 classes, 8 heads, 4 rows, 93]
            arrays["attribution"] = model.attribution_monitor(amap)
        _save(f"fullmodel_{semantics}_vitb16_c65", seed_weights=2, seed_images=0, seed_context=1, batch=B, prompt_len=P,
              class_names=np.array(names), **arrays)
        print(f"  {semantics}: {time.time() - t0:.0f}s")


def g_hf_clip_vitb16(ref):
    """An INDEPENDENT implementation at BASELINE's real dimensions: HF `transformers` CLIP built from a config at ViT-B/16
    dims, carrying the seeded weights (oracle/hf_harness.py).  Its image embeddings, its encode_text features and its raw text
    encoder on FullModel-style sequences ([16 context rows | token_embedding(zero-padded ids)], no position / mask /
    ln_final: reference models/model_wrapper.py:58,72) are what the HIP towers are held to in tests/test_gpu_hf.py --
    outputs of code that shares nothing with oracle/clip_ref.py."""
    from oracle import hf_harness

    cfg = clip_ref.CONFIGS["ViT-B-16"]
    sd = synth.make_state_dict(cfg, seed=2)
    t0 = time.time()
    model = hf_harness.build_hf_clip(cfg, sd)
    n, P, KEEP, B = 6, 16, 3, 4
    names = class_list(65)[:n]
    table = token_table(class_list(65), cfg)
    tokens = torch.cat([table[f"a photo of a {c}"] for c in names], 0)        # [6, 77]
    ctx = synth.make_prompts(65, P, cfg, seed=1)[0][:n]
    prompts = torch.cat([ctx, sd["token_embedding.weight"][tokens]], dim=1)   # [6, 93, 512]
    images = synth.make_images(B, cfg, 0)
    img = hf_harness.image_features(model, images)
    txt = hf_harness.text_features(model, tokens)
    hidden, probs = hf_harness.raw_text_transformer(model, prompts)
    print(f"  HF transformers {__import__('transformers').__version__}: {time.time() - t0:.0f}s")
    _save("hf_clip_vitb16", seed_weights=2, seed_images=0, seed_context=1, batch=B, prompt_len=P, token_ids=tokens,
          transformers_version=np.array(__import__("transformers").__version__),
          image_embeddings=img, text_features=txt, raw_hidden=hidden[:KEEP], raw_hidden_last=hidden[:, -1],
          raw_attn_mean=probs[:KEEP].mean(1), raw_probs_rows=probs[:KEEP, :, :4])


def g_image_tower_l14(ref):
    """BASELINE.json configs[4] image side: the full-depth (24-block) ViT-L/14@336 tower at batch 2, fp32 and with the
    GEMM operands rounded to MXFP8 at the kernels' rounding points (regression pins of oracle/clip_ref.py: parity
    unpinned against open_clip)."""
    cfg = clip_ref.CONFIGS["ViT-L-14-336"]
    sd = synth.make_state_dict(cfg, seed=2, text=False)
    images = synth.make_images(2, cfg, 0)
    with torch.no_grad():
        t0 = time.time()
        emb = clip_ref.encode_image(images, sd, cfg)
        pair = _mx8_pair(images, sd, cfg)
    print(f"  {time.time() - t0:.0f}s")
    _save("image_tower_ViT-L-14-336", seed_weights=2, seed_images=0, batch=2, embeddings=emb, **pair)


def g_fullmodel_l14(ref):
    """BASELINE.json configs[4] text side + loss: the reference FullModel on ViT-L/14@336 dims (text d = 768, H = 12),
    3 classes, 16 context tokens, batch 2, with the gradients its training loop uses (train.py:99-105)."""
    cfg = clip_ref.CONFIGS["ViT-L-14-336"]
    sd = synth.make_state_dict(cfg, seed=2)
    names = class_list(3)
    for semantics in ("literal", "intended"):
        t0 = time.time()
        arrays, model, clip = _full_model_case(ref, cfg, sd, names, 16, 2, semantics, seed_img=0)
        arrays.pop("prompts")
        arrays.pop("last_attention_capture")
        _save(f"fullmodel_{semantics}_vitl14", seed_weights=2, seed_images=0, batch=2, prompt_len=16,
              class_names=np.array(names), **arrays)
        print(f"  {semantics}: {time.time() - t0:.0f}s")


def g_checkpoint(ref):
    """what `torch.save(model.state_dict())` of the reference holds (train.py:131-132) and what
    test_cross_domain.py:43-61 loads back -- tensors of the reference FullModel's own state_dict() on the tiny config,
    the legacy single-tensor context layout (`prompt_learner.context_emb`), and the logits that model produces.
    RefClip is functional on the image side, so the `clip.model.visual.*` (and the unused text-side) entries an
    open_clip module would contribute are added from the same seeded state dict under open_clip's key names."""
    cfg = clip_ref.CONFIGS["tiny"]
    sd = synth.make_state_dict(cfg, seed=2)
    names = class_list(3)
    arrays, model, clip = _full_model_case(ref, cfg, sd, names, 5, 4, "intended", seed_img=0, with_grads=False)
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    for k, v in sd.items():
        state.setdefault("clip.model." + k, v)
    legacy = {k: v for k, v in state.items() if "prompt_learner.context_bank" not in k}
    legacy["prompt_learner.context_emb"] = torch.stack([state[f"prompt_learner.context_bank.{c}"] for c in names], 0)
    out = {"logits": arrays["logits"], "token_ids": arrays["token_ids"], "class_names": np.array(names),
           "keys": np.array(sorted(state.keys())), "legacy_keys": np.array(sorted(legacy.keys()))}
    # tensors that are the seeded CLIP weights themselves are committed as (key, float64 sum) and regenerated from
    # the seed by the test; everything else the reference's state_dict() holds is committed in full
    seeded, sums = [], []
    for k, v in state.items():
        base = k[len("clip.model."):] if k.startswith("clip.model.") else None
        if base in sd and torch.equal(v, sd[base]):
            seeded.append(k)
            sums.append(float(v.double().sum()))
        else:
            out["sd/" + k] = v
    out["seeded_keys"], out["seeded_sums"] = np.array(seeded), np.array(sums)
    out["legacy/prompt_learner.context_emb"] = legacy["prompt_learner.context_emb"]
    _save("checkpoint_tiny", seed_weights=2, seed_images=0, batch=4, prompt_len=5, **out)


def g_eval_metrics(ref):
    """the reference's own utils/eval_metrics.py functions (:6-96) run on the reference FullModel
    (tiny config) over a 3-batch loader, and on that model's attribution vectors."""
    import contextlib
    import io

    em = ref["eval_metrics"]
    cfg = clip_ref.CONFIGS["tiny"]
    sd = synth.make_state_dict(cfg, seed=2)
    names = class_list(6)
    arrays, model, clip = _full_model_case(ref, cfg, sd, names, 5, 4, "intended", seed_img=0, with_grads=False)
    model.eval()
    images = synth.make_images(12, cfg, 41)
    with torch.no_grad():
        logits = model(images)["logits"]
    # labels: the model's own prediction for two thirds of the samples, seeded random classes for the rest
    pred = logits.argmax(1)
    labels = pred.clone()
    labels[::3] = synth.make_labels(12, len(names), seed=9)[::3]
    top2 = logits.topk(2, dim=1).values
    margin = float((top2[:, 0] - top2[:, 1]).min() / logits.abs().max())
    loader = [(images[i:i + 4], labels[i:i + 4]) for i in range(0, 12, 4)]
    with contextlib.redirect_stdout(io.StringIO()):
        acc = em.evaluate_accuracy(model, loader, "cpu")
        per = em.evaluate_per_class_accuracy(model, loader, "cpu", names)
    clip.reset()
    clip.model.transformer(model.prompt_learner().detach())
    attr = model.attribution_monitor(clip.get_attention_map())     # [n_cls, P]
    # the variance helper on fixed scores (this randomly initialised model's attribution is uniform to 1e-4, so its
    # own variance would be a difference of round-off): 12 softmax rows in 3 label groups
    scores = torch.softmax(synth.normal([12, 5], 10, "eval.scores"), dim=-1)
    groups = synth.integers([12], 10, "eval.groups", 3)
    _save("eval_metrics_tiny", seed_weights=2, seed_images=41, n_images=12, prompt_len=5, class_names=np.array(names),
          token_ids=arrays["token_ids"], context=arrays["context"], labels=labels, logits=logits, min_top2_margin=margin,
          accuracy=acc, per_class_names=np.array(list(per.keys())), per_class_acc=np.array(list(per.values())),
          attribution=attr, attribution_entropy=em.attribution_entropy(attr), variance_scores=scores, variance_groups=groups,
          attribution_variance=em.attribution_variance(scores, groups), scores_entropy=em.attribution_entropy(scores))
    print(f"  accuracy {acc:.2f}  per-class {per}  min top-2 margin {margin:.3e}")


def g_fullmodel_b16_b32c5(ref):
    """The reference scripts' OWN operating point (train.py:29-39,75-81; test_cross_domain.py:30): batch 32, 5 classes, 5
    context tokens (T = 82) -- on ViT-B/16 (BASELINE's model; the scripts name no architecture beyond a checkpoint path).
    The literal loop: 5 x (32 + 1) text passes and one 32-image pass of the image tower, with the training loop's gradients."""
    cfg = clip_ref.CONFIGS["ViT-B-16"]
    sd = synth.make_state_dict(cfg, seed=2)
    names = class_list(5)
    P, B = 5, 32
    table = token_table(names, cfg)
    images = synth.make_images(B, cfg, 0)
    labels = synth.make_labels(B, len(names))
    for semantics in ("literal", "intended"):
        t0 = time.time()
        clip = ref_harness.RefClip(cfg, sd, semantics, ref_harness.FixedTokenizer(table))
        torch.manual_seed(1234)
        model = ref["FullModel"](names, clip, prompt_len=P, adjustor_method="scale", class_specific=True)
        _seed_context(model, names, cfg, P, seed=1)
        out = model(images, labels)
        out["loss"].backward()
        grads = torch.stack([model.prompt_learner.context_bank[c].grad for c in names], 0)
        arrays = dict(logits=out["logits"], loss=out["loss"], labels=labels,
                      token_ids=torch.cat([table[f"a photo of a {c}"] for c in names], 0),
                      context_grad=grads, logit_scale_grad=model.logit_scale.grad)
        if semantics == "intended":
            clip.reset()
            with torch.no_grad():
                clip.model.transformer(model.prompt_learner().detach())
            amap = clip.get_attention_map()  # [5, 82, 82]
            arrays["attn_map"] = amap
            arrays["attribution"] = model.attribution_monitor(amap)
        _save(f"fullmodel_{semantics}_vitb16_b32_c5", seed_weights=2, seed_images=0, seed_context=1, batch=B, prompt_len=P,
              class_names=np.array(names), **arrays)
        print(f"  {semantics}: {time.time() - t0:.0f}s")


def g_stress_vitb16(ref):
    """Hostile statistics at ViT-B/16 dims (synth.make_stress_state_dict / make_stress_images): LayerNorm gains with x10-x30
    channels, residual-stream outliers of |x| = 100-300, near one-hot last-block softmax rows, saturated patches -- what the
    IEEE-half default mode must hold 1e-3 on.  Goldens from THREE sources: the fp32 oracle, HF `transformers` CLIP carrying the
    same weights (independent code), and the reference's own FullModel (8 classes, 16 context tokens, batch 4, literal loop)."""
    from oracle import hf_harness

    cfg = clip_ref.CONFIGS["ViT-B-16"]
    sd = synth.make_stress_state_dict(cfg, seed=7)
    names = class_list(8)
    P, B, KEEP = 16, 4, 3
    table = token_table(names, cfg)
    images = synth.make_stress_images(B, cfg, 0)
    labels = synth.make_labels(B, len(names))
    t0 = time.time()
    with torch.no_grad():
        emb = clip_ref.encode_image(images, sd, cfg)
    hf = hf_harness.build_hf_clip(cfg, sd)
    emb_hf = hf_harness.image_features(hf, images)
    print(f"  image tower: oracle vs HF rel-max {float((emb - emb_hf).abs().max() / emb_hf.abs().max()):.3e}  ({time.time() - t0:.0f}s)")
    tokens = torch.cat([table[f"a photo of a {c}"] for c in names], 0)
    ctx = synth.make_prompts(len(names), P, cfg, seed=1)[0]
    prompts = torch.cat([ctx, sd["token_embedding.weight"][tokens]], dim=1)  # [8, 93, 512]
    with torch.no_grad():
        hidden_hf, probs_hf = hf_harness.raw_text_transformer(hf, prompts)
    arrays = dict(image_embeddings=emb, image_embeddings_hf=emb_hf, token_ids=tokens,
                  raw_hidden_last_hf=hidden_hf[:, -1], raw_hidden_hf=hidden_hf[:KEEP], raw_attn_mean_hf=probs_hf[:KEEP].mean(1),
                  raw_attn_max_prob_hf=probs_hf.max())
    for semantics in ("intended",):
        clip = ref_harness.RefClip(cfg, sd, semantics, ref_harness.FixedTokenizer(table))
        torch.manual_seed(1234)
        model = ref["FullModel"](names, clip, prompt_len=P, adjustor_method="scale", class_specific=True)
        _seed_context(model, names, cfg, P, seed=1)
        # the images as the stressed batch: RefClip.encode_image is the fp32 oracle
        out = model(images, labels)
        out["loss"].backward()
        grads = torch.stack([model.prompt_learner.context_bank[c].grad for c in names], 0)
        clip.reset()
        with torch.no_grad():
            hid = clip.model.transformer(model.prompt_learner().detach())
        amap = clip.get_attention_map()
        arrays.update(logits=out["logits"], loss=out["loss"], labels=labels, context_grad=grads, logit_scale_grad=model.logit_scale.grad,
                      attn_map=amap[:KEEP], attn_map_last_col=amap[:, :, -1], attribution=model.attribution_monitor(amap),
                      raw_hidden_last=hid[:, -1])
    print(f"  text side: max probability of the last block {float(probs_hf.max()):.4f}; oracle vs HF hidden(last) rel-max "
          f"{float((arrays['raw_hidden_last'] - hidden_hf[:, -1]).abs().max() / hidden_hf[:, -1].abs().max()):.3e}")
    _save("stress_vitb16", seed_weights=7, seed_images=0, seed_context=1, batch=B, prompt_len=P, class_names=np.array(names), **arrays)
    print(f"  {time.time() - t0:.0f}s")


ALL = {"attribution_monitor": g_attribution_monitor, "prompt_adjustor": g_prompt_adjustor,
       "fullmodel_tiny": g_fullmodel_tiny, "block_real_dims": g_block_real_dims, "image_tower": g_image_tower,
       "fullmodel_b32": g_fullmodel_b32, "fullmodel_b16_c65": g_fullmodel_b16_c65, "image_tower_l14": g_image_tower_l14,
       "fullmodel_l14": g_fullmodel_l14, "checkpoint": g_checkpoint, "eval_metrics": g_eval_metrics,
       "hf_clip_vitb16": g_hf_clip_vitb16, "fullmodel_b16_b32c5": g_fullmodel_b16_b32c5, "stress_vitb16": g_stress_vitb16}


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
