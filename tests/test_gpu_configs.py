"""GPU parity at the BASELINE.json configurations that round 1 left without a test, against goldens produced by the
REFERENCE's own classes (oracle/make_golden.py):

  configs[2]  ViT-B/16 image + text towers, 65 classes, 16 context tokens (T = 93), attention-map write-back on:
              the reference FullModel's logits / loss / gradients, the [65, 93, 93] head-mean map, per-head rows and
              the attribution -- at the reference's batch 4 -- plus the batch-256 run through size-independent properties.
  configs[4]  ViT-L/14@336: the full-depth (24-block) image tower in every precision incl. fp8 (MXFP8 GEMMs), the
              d = 768 / H = 12 text tower forward and backward, and the reference FullModel with the gradients of its
              training loop (train.py:99-105).
  SURVEY 8f   rows 2 and 4: a state dict as the reference's FullModel produces it (train.py:131-132) loaded the way
              test_cross_domain.py:43-61 does; the reference's utils/eval_metrics.py results on fixed inputs.
  N > 1       one AdamW step on two ranks == the same step in one process on the global batch.

Tolerances as in test_gpu_parity.py: TOL = 1e-3 for the modes that claim BASELINE.json's bound (bf16x3, and fp16 =
IEEE-half image tower + split-bf16 text tower), TOL_BF16 = 2e-2 for plain bf16 (reported, not the parity claim)."""
import contextlib
import math
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

import tap_clip_amd  # noqa: F401
from conftest import golden, rel_l2, rel_max
from oracle import clip_ref
from tap_clip_amd import configs, synth
from test_gpu_parity import DEV, TOL, TOL_BF16, TOL_TAIL_SPLIT, _bf16_floor, _build_full, _report

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def eng():
    from tap_clip_amd import engine

    return engine


# ---- BASELINE configs[2] -------------------------------------------------------------------------------------
@pytest.mark.parametrize("semantics", ["literal", "intended"])
def test_fullmodel_vitb16_65_classes_vs_reference(semantics):
    g = golden(f"fullmodel_{semantics}_vitb16_c65")
    ref = torch.from_numpy(g["logits"])
    labels = torch.from_numpy(g["labels"]).to(DEV)
    fl = _bf16_floor(g, "ViT-B-16", semantics)  # the bf16 mode is held to 1.5x what the format itself costs on this case
    for precision, tol in (("bf16x3", TOL), ("fp16", TOL), ("bf16", 1.5 * fl["logits"])):
        model, images = _build_full("ViT-B-16", g, semantics, precision)
        with torch.no_grad():
            out = model(images, labels)
        _report(f"FullModel ViT-B/16 65 classes {semantics} {precision} logits", out["logits"], ref)
        assert out["logits"].shape == (4, 65)
        assert rel_max(out["logits"].cpu(), ref) < tol
        assert abs(float(out["loss"]) - float(g["loss"])) < (1.5 * fl["loss"] if precision == "bf16" else tol * max(1.0, abs(float(g["loss"]))))
        if semantics == "intended":
            tol_map = 1.5 * fl["map"] if precision == "bf16" else tol
            amap = model.clip.attention_maps[0].cpu()                        # pass 1's capture: [65, 93, 93]
            assert amap.shape == (65, 93, 93)
            _report(f"  head-mean map {precision}", amap[:8], torch.from_numpy(g["attn_map_head"]))
            assert rel_max(amap[:8], torch.from_numpy(g["attn_map_head"])) < tol_map
            assert rel_max(amap[:, :, -1], torch.from_numpy(g["attn_map_last_col"])) < tol_map
            assert torch.allclose(amap.sum(-1), torch.from_numpy(g["attn_row_sums"]), atol=1e-4)
            assert rel_max(model.last_attribution.cpu(), torch.from_numpy(g["attribution"])) < (max(1.5 * fl["attribution"], TOL) if precision == "bf16" else tol)
        if precision == "bf16x3":
            # the training step of the same model: context gradients of the reference's autograd
            model.train()
            o = model(images, labels)
            o["loss"].backward()
            names = g["class_names"].tolist()
            grad = torch.stack([model.prompt_learner.context_bank[c].grad for c in names], 0).cpu()
            _report(f"  context grad {precision}", grad[:8], torch.from_numpy(g["context_grad_head"]))
            assert rel_max(grad[:8], torch.from_numpy(g["context_grad_head"])) < tol
            assert rel_max(grad.flatten(1).norm(dim=1), torch.from_numpy(g["context_grad_norms"])) < tol
            assert abs(float(model.logit_scale.grad) - float(g["logit_scale_grad"])) < tol * max(1.0, abs(float(g["logit_scale_grad"])))
        del model
        torch.cuda.empty_cache()


# ---- the reference scripts' own operating point ----------------------------------------------------------------
@pytest.mark.parametrize("semantics", ["literal", "intended"])
def test_fullmodel_vitb16_batch_32_five_classes_vs_reference(semantics):
    """What the reference's train.py / test_cross_domain.py really run (train.py:29-39,75-81): batch 32, 5 classes, 5 context
    tokens (T = 82) -- on ViT-B/16 -- against the reference FullModel's own logits, loss, gradients, map and attribution.
    At 6 304 image rows the tower is in another regime than at BASELINE's batch 256 (1.2 rounds of GEMM tiles, every c_proj tile
    K-split): this is the shape a user who drops the library into those scripts gets."""
    g = golden(f"fullmodel_{semantics}_vitb16_b32_c5")
    ref = torch.from_numpy(g["logits"])
    labels = torch.from_numpy(g["labels"]).to(DEV)
    names = g["class_names"].tolist()
    for precision, tol in (("bf16x3", TOL), ("fp16", TOL)):
        model, images = _build_full("ViT-B-16", g, semantics, precision)
        assert images.shape[0] == 32
        model.train()
        out = model(images, labels)
        out["loss"].backward()
        _report(f"FullModel ViT-B/16 batch 32, 5 classes, P=5 {semantics} {precision} logits", out["logits"].detach(), ref)
        assert out["logits"].shape == (32, 5)
        assert rel_max(out["logits"].detach().cpu(), ref) < tol
        assert abs(float(out["loss"]) - float(g["loss"])) < tol * max(1.0, abs(float(g["loss"])))
        grad = torch.stack([model.prompt_learner.context_bank[c].grad for c in names], 0).cpu()
        _report(f"  context grad {precision}", grad, torch.from_numpy(g["context_grad"]))
        assert rel_max(grad, torch.from_numpy(g["context_grad"])) < tol
        assert abs(float(model.logit_scale.grad) - float(g["logit_scale_grad"])) < tol * max(1.0, abs(float(g["logit_scale_grad"])))
        if semantics == "intended":
            amap = model.clip.attention_maps[0].cpu()
            assert amap.shape == (5, 82, 82)
            assert rel_max(amap, torch.from_numpy(g["attn_map"])) < tol
            assert rel_max(model.last_attribution.cpu(), torch.from_numpy(g["attribution"])) < tol
        del model
        torch.cuda.empty_cache()


def test_per_head_write_back_vitb16_65_classes(eng):
    """The per-head probabilities [n, 8, 93, 93] that a hook on resblocks[-1].attn receives (reference
    clip_wrapper.py:29-40, intended semantics) against the reference run's own rows."""
    g = golden("fullmodel_intended_vitb16_c65")
    model, _ = _build_full("ViT-B-16", g, "intended", "bf16x3")
    got = {}
    model.clip.model.transformer.resblocks[-1].attn.register_forward_hook(lambda m, i, o: got.update(p=o[0]))
    with torch.no_grad():
        model.clip.reset()
        model.clip.model.transformer(model.prompt_learner().detach())
    p = got["p"].cpu()
    assert p.shape == (65, 8, 93, 93)
    assert rel_max(p[:4, :, :4, :], torch.from_numpy(g["probs_per_head_rows"])) < TOL
    assert torch.allclose(p.sum(-1), torch.ones(65, 8, 93), atol=1e-5)


def test_fullmodel_vitb16_65_classes_batch_256_properties():
    """configs[2] at its full size (batch 256): finite, run-to-run bit-identical, and the rows of the 4 golden images
    inside the big batch equal the batch-4 rows (text features do not depend on the images)."""
    g = golden("fullmodel_intended_vitb16_c65")
    model, images4 = _build_full("ViT-B-16", g, "intended", "bf16")
    cfg = configs.get_config("ViT-B-16")
    big = torch.cat([images4, synth.make_images(252, cfg, 77).to(DEV)], 0)
    with torch.no_grad():
        a = model(big)["logits"]
        b = model(big)["logits"]
        small = model(images4)["logits"]
    assert a.shape == (256, 65) and bool(torch.isfinite(a).all())
    assert torch.equal(a, b)
    assert torch.equal(a[:4], small)
    assert rel_max(a[:4].cpu(), torch.from_numpy(g["logits"])) < TOL_BF16
    assert model.clip.attention_maps[0].shape == (65, 93, 93)


# ---- BASELINE configs[4] -------------------------------------------------------------------------------------
def test_encode_image_vit_l14_336_full_depth(eng):
    """All 24 blocks of ViT-L/14@336 (577 tokens: the flash-style attention kernel; K = 1024 / 4096 GEMMs) at batch 2
    against the fp32 oracle's embeddings, in every precision; fp8 also against the oracle with MXFP8 operand rounding."""
    g = golden("image_tower_ViT-L-14-336")
    cfg = configs.get_config("ViT-L-14-336")
    sd = synth.make_state_dict(cfg, seed=int(g["seed_weights"]), text=False)
    images = synth.make_images(int(g["batch"]), cfg, int(g["seed_images"])).to(DEV)
    ref, ref8 = torch.from_numpy(g["embeddings"]), torch.from_numpy(g["embeddings_mx8_f64"])
    floor8 = float(g["mx8_floor_rel_l2"])  # the MXFP8 emulation against itself, fp32 vs fp64 accumulation (make_golden._mx8_pair)
    for precision, tol in (("bf16x3", TOL), ("fp16", TOL), ("bf16", TOL_BF16)):
        tower = eng.VisionTower(cfg, sd, DEV, precision)
        emb = tower.encode_image(images).cpu()
        _report(f"encode_image ViT-L/14@336 24 blocks {precision}", emb, ref)
        assert rel_l2(emb, ref) < tol and rel_max(emb, ref) < (tol if precision != "bf16" else 2 * TOL_BF16)
        tower.close()
        del tower
        torch.cuda.empty_cache()
    tower = eng.VisionTower(cfg, sd, DEV, "fp8")
    emb = tower.encode_image(images).cpu()
    cos = torch.nn.functional.cosine_similarity(emb, ref, dim=-1)
    _report("encode_image ViT-L/14@336 24 blocks fp8 vs fp32", emb, ref)
    _report("encode_image ViT-L/14@336 24 blocks fp8 vs MXFP8-rounding oracle", emb, ref8)
    # a throughput mode (3 mantissa bits per operand element), parity unpinned against the reference.  Bounds from the
    # format's own floor, not loose constants: 1.5x the emulation's disagreement with itself (2.5e-2 over 24 blocks)
    # against the fp64-accumulating MXFP8 emulation, 1.5x that emulation's distance from the fp32 oracle (3.5e-2)
    # against the fp32 golden -- one layer with a wrong scale plane lands far outside either
    fmt = rel_l2(ref8, ref)
    assert rel_l2(emb, ref8) < 1.5 * floor8, (rel_l2(emb, ref8), floor8)
    assert rel_l2(emb, ref) < 1.5 * fmt and float(cos.min()) > 0.998, (rel_l2(emb, ref), fmt, float(cos.min()))
    assert torch.equal(emb, tower.encode_image(images).cpu())


@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_text_tower_l14_dims_forward_backward_vs_oracle(eng, precision):
    """The ViT-L/14 text tower (d = 768, 12 heads, mlp 3072, 12 blocks) at T = 16 + 77 = 93: hidden states against
    `clip_ref.text_transformer_raw`, dL/dx against torch autograd through the oracle."""
    cfg = configs.get_config("ViT-L-14-336")
    ocfg = clip_ref.CONFIGS["ViT-L-14-336"]
    sd = synth.make_state_dict(cfg, seed=2, vision=False)
    tower = eng.TextTower(cfg, sd, DEV, precision)
    n, T, D = 5, 93, 768
    x = torch.cat([synth.normal([n, 16, D], 4, "l14.ctx"), synth.normal([n, 77, D], 4, "l14.tok", 0.02)], dim=1)
    gfeat = synth.normal([n, cfg.embed_dim], 4, "l14.g")
    xr = x.clone().requires_grad_(True)
    hidden, probs, _ = clip_ref.text_transformer_raw(xr, sd, ocfg, want_probs=True)
    feat = hidden[:, -1, :] @ sd["text_projection"]
    feat = feat / feat.norm(dim=-1, keepdim=True)
    (feat * gfeat).sum().backward()
    tol = TOL if precision == "bf16x3" else 5e-2
    r = tower.forward(x.to(DEV), want_heads=True, want_mean=True)
    _report(f"L/14 text tower {precision} hidden", r["hidden"], hidden.detach())
    assert rel_max(r["hidden"].cpu(), hidden.detach()) < (TOL if precision == "bf16x3" else TOL_BF16)
    assert rel_max(r["attn_heads"].cpu(), probs.detach()) < (TOL if precision == "bf16x3" else TOL_BF16)
    assert rel_max(r["attn_mean"].cpu(), probs.detach().mean(1)) < (TOL if precision == "bf16x3" else TOL_BF16)
    got = tower.pool_project(r["hidden"], normalize=True)
    assert rel_max(got.cpu(), feat.detach()) < (TOL if precision == "bf16x3" else TOL_BF16)
    g_hidden = tower.pool_project_backward(r["hidden"], gfeat.to(DEV), normalize=True)
    gx = tower.backward(x.to(DEV), g_hidden).cpu()
    _report(f"L/14 text tower {precision} dL/dx", gx, xr.grad)
    assert rel_l2(gx, xr.grad) < tol
    assert rel_max(gx[:, :16], xr.grad[:, :16]) < tol


@pytest.mark.parametrize("semantics", ["literal", "intended"])
def test_fullmodel_vitl14_vs_reference(semantics):
    """The reference FullModel on ViT-L/14@336 (image tower 24 blocks, text d = 768): logits, loss and the gradients
    its training loop uses, bf16x3 at 1e-3; the fp8 image tower of configs[4] beside it, bounded."""
    g = golden(f"fullmodel_{semantics}_vitl14")
    ref = torch.from_numpy(g["logits"])
    labels = torch.from_numpy(g["labels"]).to(DEV)
    names = g["class_names"].tolist()
    model, images = _build_full("ViT-L-14-336", g, semantics, "bf16x3")
    model.train()
    out = model(images, labels)
    _report(f"FullModel ViT-L/14@336 {semantics} bf16x3 logits", out["logits"], ref)
    assert rel_max(out["logits"].detach().cpu(), ref) < TOL
    assert abs(float(out["loss"]) - float(g["loss"])) < TOL * max(1.0, abs(float(g["loss"])))
    out["loss"].backward()
    grad = torch.stack([model.prompt_learner.context_bank[c].grad for c in names], 0).cpu()
    gref = torch.from_numpy(g["context_grad"])
    _report(f"FullModel ViT-L/14@336 {semantics} bf16x3 context grad", grad, gref)
    assert rel_max(grad, gref) < TOL and rel_l2(grad, gref) < TOL
    assert abs(float(model.logit_scale.grad) - float(g["logit_scale_grad"])) < TOL * max(1.0, abs(float(g["logit_scale_grad"])))
    del model
    torch.cuda.empty_cache()
    model, images = _build_full("ViT-L-14-336", g, semantics, "fp8")  # configs[4]: fp8 image tower, bf16 text tower
    with torch.no_grad():
        lg = model(images)["logits"].cpu()
    _report(f"FullModel ViT-L/14@336 {semantics} fp8 logits", lg, ref)
    # A logit is 14.29 x the cosine of two unit vectors; MXFP8 moves the image one by d, |d| <= 1.5 x 3.5e-2 (the bound
    # of test_encode_image_vit_l14_336_full_depth), in a direction unrelated to the text feature: its component along a
    # fixed unit vector in E = 768 dimensions has standard deviation |d| / sqrt(E), so 4 sigma is 14.29 x 5.25e-2 x 4 /
    # 27.7 = 0.108 in ABSOLUTE terms.  (Relative to this randomly initialised model's largest logit, 0.30-0.41, that is
    # the 0.3 the test used before: small logits, not a loose bound -- which is why it is now stated in absolute terms.)
    bound = 14.2857 * 1.5 * 3.5e-2 * 4.0 / math.sqrt(768)
    assert float((lg - ref).abs().max()) < bound and bool(torch.isfinite(lg).all()), (float((lg - ref).abs().max()), bound)


# ---- real-dims encode_text (causal mask, positional embedding, ln_final, EOT pool) -----------------------
@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_encode_text_vitb16_dims(eng, precision):
    from tap_clip_amd.models import CLIPWrapper

    cfg = configs.get_config("ViT-B-16")
    sd = synth.make_state_dict(cfg, seed=2)
    clip = CLIPWrapper("ViT-B-16", None, DEV, precision=precision, state_dict=sd)
    tokens = torch.zeros(6, cfg.ctx, dtype=torch.long)
    for i in range(6):
        L = 3 + 11 * i
        tokens[i, 0] = cfg.vocab - 2
        tokens[i, 1:1 + L] = synth.integers([L], 3, f"t16.{i}", cfg.vocab - 3) + 1
        tokens[i, 1 + L] = cfg.vocab - 1          # EOT = the largest id: argmax finds it (open_clip's pooling)
    out = clip.encode_text(tokens.to(DEV))
    with torch.no_grad():
        ref = clip_ref.encode_text(tokens, sd, clip_ref.CONFIGS["ViT-B-16"])
    _report(f"encode_text ViT-B/16 dims {precision}", out, ref)
    assert out.shape == (6, 512)
    assert rel_max(out.cpu(), ref) < (TOL if precision == "bf16x3" else TOL_BF16)


# ---- a reference-produced state dict ---------------------------------------------------------------
def _reference_state_dict(g):
    """The reference FullModel's state_dict() of oracle/make_golden.py g_checkpoint: seeded CLIP tensors regenerated
    (and checked against their committed sums), everything else from the fixture."""
    cfg = configs.get_config("tiny")
    sd = synth.make_state_dict(cfg, seed=int(g["seed_weights"]))
    state = {}
    for k, want in zip(g["seeded_keys"].tolist(), g["seeded_sums"].tolist()):
        t = sd[k[len("clip.model."):]]
        assert abs(float(t.double().sum()) - want) <= 1e-9 * max(1.0, abs(want)), k
        state[k] = t
    for f in g.files:
        if f.startswith("sd/"):
            state[f[3:]] = torch.from_numpy(g[f])
    assert sorted(state) == g["keys"].tolist()
    return state


@pytest.mark.parametrize("legacy", [False, True])
def test_checkpoint_from_the_reference_loads(legacy):
    """reference test_cross_domain.py:43-67 against a state dict the REFERENCE's FullModel produced: filter / convert
    as the script does, load with strict=False into a model built from other weights and another context draw, get
    the reference's logits; then register an unseen class."""
    from tap_clip_amd.models import CLIPWrapper, FullModel

    g = golden("checkpoint_tiny")
    names = g["class_names"].tolist()
    state_dict = _reference_state_dict(g)
    if legacy:  # the older layout the script still converts: one [n_cls, P, D] tensor
        state_dict = {k: v for k, v in state_dict.items() if "prompt_learner.context_bank" not in k}
        state_dict["prompt_learner.context_emb"] = torch.from_numpy(g["legacy/prompt_learner.context_emb"])
        assert sorted(state_dict) == g["legacy_keys"].tolist()
    cfg = configs.get_config("tiny")
    torch.manual_seed(99)
    clip = CLIPWrapper("tiny", None, DEV, precision="bf16x3", state_dict=synth.make_state_dict(cfg, seed=3))
    table = {f"a photo of a {c}": torch.from_numpy(g["token_ids"][i:i + 1]) for i, c in enumerate(names)}
    table["a photo of a Bike"] = torch.from_numpy(g["token_ids"][0:1])
    clip.tokenizer = lambda text: table[text].clone()
    model = FullModel(names, clip, prompt_len=int(g["prompt_len"]), adjustor_method="scale", class_specific=True)
    images = synth.make_images(int(g["batch"]), cfg, int(g["seed_images"])).to(DEV)
    ref = torch.from_numpy(g["logits"])
    with torch.no_grad():
        assert rel_max(model.eval()(images)["logits"].cpu(), ref) > 1e-2   # other weights, other answer
    # --- the script's conversion, verbatim in structure (test_cross_domain.py:43-61)
    converted = {}
    if "prompt_learner.context_emb" in state_dict:
        old_ctx = state_dict["prompt_learner.context_emb"]
        for i, cls_name in enumerate(names):
            converted[f"prompt_learner.context_bank.{cls_name}"] = old_ctx[i]
    for k, v in state_dict.items():
        if "prompt_learner" not in k:
            converted[k] = v
    if not legacy:  # test_cross_domain2.py:81 loads the dict as saved
        converted = state_dict
    missing, unexpected = model.load_state_dict(converted, strict=False)
    assert not unexpected, unexpected
    model.eval()
    with torch.no_grad():
        got = model(images)["logits"].cpu()
    _report(f"logits after loading the reference's state dict (legacy={legacy})", got, ref)
    assert rel_max(got, ref) < TOL
    model.prompt_learner.add_class_prompt("Bike")
    with torch.no_grad():
        assert model(images)["logits"].shape == (4, len(names) + 1)


# ---- the reference's eval_metrics results ----------------------------------------------------------
def test_eval_metrics_vs_reference():
    from tap_clip_amd.models import CLIPWrapper, FullModel
    from tap_clip_amd.utils import eval_metrics

    g = golden("eval_metrics_tiny")
    names = g["class_names"].tolist()
    cfg = configs.get_config("tiny")
    clip = CLIPWrapper("tiny", None, DEV, precision="bf16x3", state_dict=synth.make_state_dict(cfg, seed=int(g["seed_weights"])))
    table = {f"a photo of a {c}": torch.from_numpy(g["token_ids"][i:i + 1]) for i, c in enumerate(names)}
    clip.tokenizer = lambda text: table[text].clone()
    model = FullModel(names, clip, prompt_len=int(g["prompt_len"]), adjustor_method="scale", class_specific=True)
    with torch.no_grad():
        for i, c in enumerate(names):
            model.prompt_learner.context_bank[c].copy_(torch.from_numpy(g["context"][i]))
    images = synth.make_images(int(g["n_images"]), cfg, int(g["seed_images"]))
    labels = torch.from_numpy(g["labels"])
    loader = [(images[i:i + 4], labels[i:i + 4]) for i in range(0, 12, 4)]
    with torch.no_grad():
        logits = model.eval()(images.to(DEV))["logits"].cpu()
    assert rel_max(logits, torch.from_numpy(g["logits"])) < TOL
    assert float(g["min_top2_margin"]) > 100 * TOL     # no prediction of the fixture is a near-tie
    assert eval_metrics.evaluate_accuracy(model, loader, DEV) == pytest.approx(float(g["accuracy"]), abs=1e-9)
    per = eval_metrics.evaluate_per_class_accuracy(model, loader, DEV, names)
    assert list(per.keys()) == g["per_class_names"].tolist()
    assert list(per.values()) == pytest.approx(g["per_class_acc"].tolist(), abs=1e-9)
    attr = model.last_attribution.cpu()                 # the HIP path's attribution of the same prompts
    assert rel_max(attr, torch.from_numpy(g["attribution"])) < TOL
    assert eval_metrics.attribution_entropy(attr) == pytest.approx(float(g["attribution_entropy"]), rel=1e-4)
    scores, groups = torch.from_numpy(g["variance_scores"]), torch.from_numpy(g["variance_groups"])
    assert eval_metrics.attribution_variance(scores, groups) == pytest.approx(float(g["attribution_variance"]), rel=1e-5)
    assert eval_metrics.attribution_entropy(scores) == pytest.approx(float(g["scores_entropy"]), rel=1e-5)


# ---- N > 1 training: two ranks (sharing the one GPU of the test box, gloo transport) -------------------------------
_TRAIN_RANKS = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
import tap_clip_amd
from tap_clip_amd import configs, synth
from tap_clip_amd.dist import shard_rows
from tap_clip_amd.models import CLIPWrapper, FullModel
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
cfg = configs.get_config("tiny")
sd = synth.make_state_dict(cfg, seed=2)
names = ["Backpack", "Laptop", "Mug"]
def build(gather):
    torch.manual_seed(7)   # same context draw on every rank
    clip = CLIPWrapper("tiny", None, "cuda:0", precision="bf16x3", state_dict=sd)
    return FullModel(names, clip, prompt_len=5, class_specific=True, gather_images=gather)
images, labels = synth.make_images(8, cfg, 0), synth.make_labels(8, 3)
lo, hi = shard_rows(8, rank, world)
def one_step(model, x, y):
    opt = torch.optim.AdamW(model.prompt_learner.parameters(), lr=2e-3, weight_decay=0.01)   # reference train.py:65-67
    model.train()
    out = model(x.cuda(), y.cuda())                                                            # train.py:99
    opt.zero_grad(); out["loss"].backward(); opt.step()                                        # train.py:103-105
    grads = torch.stack([model.prompt_learner.context_bank[c].grad for c in names]).cpu()
    ctx = torch.stack([model.prompt_learner.context_bank[c].detach() for c in names]).cpu()
    return out["logits"].detach().cpu(), float(out["loss"]), grads, ctx
lg_s, loss_s, g_s, ctx_s = one_step(build(True), images[lo:hi], labels[lo:hi])    # sharded: local images, local labels
lg_f, loss_f, g_f, ctx_f = one_step(build(False), images, labels)                 # one process, global batch
assert lg_s.shape == (8, 3)
rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
assert rel(lg_s, lg_f) < 1e-5 and abs(loss_s - loss_f) < 1e-5 * max(1.0, abs(loss_f)), (rel(lg_s, lg_f), loss_s, loss_f)
assert rel(g_s, g_f) < 1e-4, rel(g_s, g_f)
assert float((ctx_s - ctx_f).abs().max()) < 1e-5, float((ctx_s - ctx_f).abs().max())
# every rank holds the same updated context (no gradient all-reduce is needed: the loss is the global one everywhere)
parts = [torch.empty_like(ctx_s) for _ in range(world)]
dist.all_gather(parts, ctx_s)
assert all(torch.equal(p, parts[0]) for p in parts)
# the evaluation helpers count global totals on every rank
from tap_clip_amd.utils import eval_metrics
import contextlib, io
m = build(True).eval()
with contextlib.redirect_stdout(io.StringIO()):
    acc_s = eval_metrics.evaluate_accuracy(m, [(images[lo:hi], labels[lo:hi])], "cuda:0")
    acc_f = eval_metrics.evaluate_accuracy(build(False).eval(), [(images, labels)], "cuda:0")
assert acc_s == acc_f, (acc_s, acc_f)
# unequal shards end to end (ADVICE r03): rank 0 holds 5 images in batches of 2 (2 + 2 + 1), rank 1 holds 3 in ONE batch of
# 3 -- different batch counts AND different batch lengths; the loop must neither hang nor miscount, with a sized loader and
# with a bare generator (no __len__)
mine = [(images[0:2], labels[0:2]), (images[2:4], labels[2:4]), (images[4:5], labels[4:5])] if rank == 0 else [(images[5:8], labels[5:8])]
with contextlib.redirect_stdout(io.StringIO()):
    acc_u = eval_metrics.evaluate_accuracy(m, mine, "cuda:0")
    acc_g = eval_metrics.evaluate_accuracy(m, (b for b in mine), "cuda:0")
    per_u = eval_metrics.evaluate_per_class_accuracy(m, mine, "cuda:0", names)
    per_f = eval_metrics.evaluate_per_class_accuracy(build(False).eval(), [(images, labels)], "cuda:0", names)
assert acc_u == acc_f and acc_g == acc_f and per_u == per_f, (acc_u, acc_g, acc_f, per_u, per_f)
assert m.ragged_batches is False
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok", rel(g_s, g_f))
"""


def test_two_ranks_sharded_train_step_equals_single(tmp_path):
    script = tmp_path / "train_ranks.py"
    script.write_text(_TRAIN_RANKS % ROOT)
    port = 29900 + os.getpid() % 90
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.count("ok") == 2, r.stdout[-3000:] + r.stderr[-3000:]


_NCCL_ONE_RANK = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
import tap_clip_amd
from tap_clip_amd.dist import all_gather_rows
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))      # what bench.py does for N > 1
assert dist.get_backend() == "nccl"
x = torch.nn.functional.normalize(torch.randn(256, 512, device="cuda"), dim=-1)
y = all_gather_rows(x, force=True)                                     # dist.all_gather_into_tensor over RCCL
lab = torch.arange(256, device="cuda")
assert y.shape == (256, 512) and torch.equal(y, x) and torch.equal(all_gather_rows(lab, force=True), lab)
t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)                               # bench.py's max-over-ranks timing
assert float(t) == 1.5
dist.barrier(); dist.destroy_process_group()
print("nccl ok")
"""


def test_rccl_call_path_on_one_rank(tmp_path):
    """The `nccl` (= RCCL) branch of dist.py / bench.py with a group of ONE rank: a 1-GPU box cannot host two RCCL
    ranks, but the initialisation, the device-tensor all_gather_into_tensor and the all_reduce run as they do at N > 1."""
    script = tmp_path / "nccl1.py"
    script.write_text(_NCCL_ONE_RANK % ROOT)
    port = 29800 + os.getpid() % 90
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "nccl ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_bench_spawns_its_ranks_at_the_configs3_shape(tmp_path):
    """BASELINE configs[3] (ViT-B/16, the batch sharded 256 images per rank, all-gather of the embeddings before the 65-class
    logits) through the EXACT command line the driver uses -- `python bench.py --gpus N ...`, WORLD_SIZE unset -- so that
    bench.py must start its own ranks (it used to exit with "launch N>1 with ...").
    This box has ONE GPU, so the ranks share it and the collective is gloo through host memory (TAPCLIP_DIST_BACKEND);
    N = 4, not 8: the pool's process guard allows at most 6 processes on the card at once (this pytest process is one).
    Checks: the JSON line says 4 ranks / global batch 1024 / which collective ran; the gathered [1024, 65] logits are finite
    and equal, rank-major block by block, to a single-process encode of the same seeds (images seed = 100 + rank)."""
    import json

    out = tmp_path / "logits.npy"
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(TAPCLIP_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1",
                        "--dump-logits", str(out)], capture_output=True, text=True, timeout=900, env=env)  # (default precision: fp16)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 4 and res["config"]["global_batch"] == 1024 and res["config"]["batch_per_gpu"] == 256
    assert res["config"]["parallelism"] == "dp4" and res["scaling"] == "weak" and res["dtype"] == "fp16"
    assert "gloo" in res["config"]["collective"] and "REHEARSAL" in res["config"]["collective"]
    assert res["value"] > 0 and abs(res["value"] - 1024 * res["steps"] / (res["ms_per_step"] * 1e-3 * res["steps"])) < 1e-3 * res["value"]
    # what the first real multi-GPU run will be read by: ranks seen, each rank's device and step times,
    # the exchange step's own duration, and the second metric measured over the whole job
    rk = res["ranks"]
    assert rk["rccl_ranks_seen"] == 4 and rk["backend"] == "gloo" and len(rk["per_rank"]) == 4
    assert sorted(r_["rank"] for r_ in rk["per_rank"]) == [0, 1, 2, 3]
    for r_ in rk["per_rank"]:
        assert r_["device"].startswith("cuda") and 0 < r_["step_ms"]["min"] <= r_["step_ms"]["median"] <= r_["step_ms"]["max"]
        assert r_["allgather_us"]["median"] > 0
    ff = res["full_forward"]
    assert "4 ranks" in ff["workload"] and ff["cls_only_last_block"] is False and ff["default_path"]["cls_only_last_block"] is True
    assert abs(res["logits_per_sec"] - 1024 * 65 / (ff["ms_per_forward"] * 1e-3)) < 1e-3 * res["logits_per_sec"]
    assert ff["text_rows_per_sequence"] == {**ff["text_rows_per_sequence"], "input": 93, "computed": 24}
    assert ff["text_rows_per_sequence"]["ms_per_forward_every_row"] > 0
    assert res["train_step"]["ms_per_step"] > ff["default_path"]["ms_per_forward"]
    got = torch.from_numpy(np.load(out))
    assert got.shape == (1024, 65) and bool(torch.isfinite(got).all())
    # the same model in this process (what bench.py builds: seeds 2 / 1, its default precision), one rank's images at a time
    from tap_clip_amd import engine
    from tap_clip_amd.models import CLIPWrapper, FullModel
    cfg = configs.get_config("ViT-B-16")
    sd = synth.make_state_dict(cfg, seed=2)
    clip = CLIPWrapper("ViT-B-16", None, DEV, precision="fp16", attn_semantics="intended", state_dict=sd)
    names = [f"class_{i}" for i in range(65)]
    model = FullModel(names, clip, prompt_len=16, class_specific=True).eval()
    with torch.no_grad():
        ctx = synth.make_prompts(65, 16, cfg, seed=1)[0]
        for i, c in enumerate(names):
            model.prompt_learner.context_bank[c].copy_(ctx[i])
        text_feat = model.text_features()
        clip._vision.set_prune_last_block(False)  # the dumped logits are the headline's: every row of every block
        for rank in range(4):
            emb = clip._vision.encode_image(synth.make_images(256, cfg, seed=100 + rank).to(DEV), normalize=True)
            want = engine.logits(emb, text_feat, float(model.logit_scale.exp())).cpu()
            assert torch.equal(got[256 * rank: 256 * (rank + 1)], want), f"rows of rank {rank} differ from the single-process encode"


def test_bench_checks_do_not_depend_on_the_number_of_steps(tmp_path):
    """bench.py's precision table is evaluated at the SEEDED prompts (its prompt-tuning leg used to move
    them first, with a step count that follows --steps, so `precisions.*.logits_*` -- and the arg-max agreement DESIGN.md once
    quoted -- changed with the command line).  Two runs with different --steps: identical error fields in every precision, the
    headline's own tolerance flag TRUE (round 5: the headline is the compliant fp16 mode, bf16 an out-of-tolerance extra), logits/s
    made of the full forward; the sustained leg and the batch sweep of round 5 present in the first run (a short form of them)."""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    lines = []
    for steps in ("4", "24"):
        extra = ["--sustained-seconds", "2"] if steps == "4" else ["--no-sustained", "--no-batch-sweep"]
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", steps, "--warmup", "1", "--no-input-side", "--no-configs4"] + extra,
                           capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        lines.append(json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1]))
    a, b = lines
    assert set(a["precisions"]) == set(b["precisions"]) == {"bf16", "fp16", "bf16x3", "fp8"}
    for p in a["precisions"]:
        for k in ("logits_rel_max_vs_cpu_oracle", "logits_rel_l2_vs_cpu_oracle", "meets_1e-3", "logits_err_over_top2_margin", "embedding_rel_l2_vs_bf16x3"):
            assert a["precisions"][p][k] == b["precisions"][p][k], (p, k, a["precisions"][p][k], b["precisions"][p][k])
    assert a["dtype"] == "fp16" and a["headline_meets_tolerance"] is True and a["precisions"]["fp16"]["meets_1e-3"] and a["precisions"]["bf16x3"]["meets_1e-3"]
    assert a["parity_mode"]["precision"] == "fp16" and a["parity_mode"]["roofline"]["frac"] > 0.2
    assert a["bf16_mode"]["meets_1e-3"] is False and a["bf16_mode"]["img_per_s"] > 0
    su = a["sustained"]
    assert su["seconds"] >= 1.9 and su["steps"] > 50 and 0.8 < su["last_over_first"] < 1.25 and su["img_per_s"] > 0.8 * a["value"]
    bs = a["batch_sweep"]
    assert set(bs["rows"]) == {"8", "32", "64", "128", "256"} and "kernels" in bs["rows"]["32"]
    for row in bs["rows"].values():
        assert row["encode"]["img_per_s"] > 0 and row["c5_p5"]["train_step_ms"] > row["c5_p5"]["full_forward_ms"] > 0 and row["c65_p16"]["full_forward_ms"] > 0
    ff = a["full_forward"]
    assert ff["cls_only_last_block"] is False and abs(a["logits_per_sec"] - 256 * 65 / (ff["ms_per_forward"] * 1e-3)) < 1e-3 * a["logits_per_sec"]


def test_vit_l14_336_batch_128_properties(eng):
    """BASELINE configs[4] at its per-GPU size: ViT-L/14@336, batch 128, fp8 image tower -- finite, unit norms, run-to-run
    bit-identical, and the two golden images inside the big batch equal their batch-2 rows (to the mode's round-off: the
    K-split tail tiles associate differently, bitwise with the tail split off is not asserted here); bf16 and fp16 too."""
    g = golden("image_tower_ViT-L-14-336")
    cfg = configs.get_config("ViT-L-14-336")
    sd = synth.make_state_dict(cfg, seed=int(g["seed_weights"]), text=False)
    two = synth.make_images(2, cfg, int(g["seed_images"]))
    big = torch.cat([two, synth.make_images(126, cfg, 78)], 0).to(DEV)
    ref = torch.nn.functional.normalize(torch.from_numpy(g["embeddings"]), dim=-1)
    fmt = rel_l2(torch.from_numpy(g["embeddings_mx8_f64"]), torch.from_numpy(g["embeddings"]))
    for precision, tol in (("fp8", 1.5 * fmt), ("fp16", TOL), ("bf16", TOL_BF16)):
        tower = eng.VisionTower(cfg, sd, DEV, precision)
        a = tower.encode_image(big, normalize=True)
        b = tower.encode_image(big, normalize=True)
        assert a.shape == (128, cfg.embed_dim) and bool(torch.isfinite(a).all())
        assert torch.equal(a, b), f"{precision}: same input twice must be bit-identical"
        assert torch.allclose(a.norm(dim=-1), torch.ones(128, device=DEV), atol=1e-5)
        _report(f"ViT-L/14@336 batch 128 {precision}: golden rows in the big batch", a[:2].cpu(), ref)
        assert rel_l2(a[:2].cpu(), ref) < tol
        # the same two images alone (2 x 577 = 1 154 rows: since round 4 the persistent GEMM's shape too, where every tile of
        # so small a launch is K-split over the idle CUs).  Without the K-split the rows sum in the big batch's order (its
        # first row tiles are whole tiles); with it they differ by the mode's round-off, as documented for the tail split
        tower.set_ksplit(False)
        small = tower.encode_image(big[:2].clone(), normalize=True)
        assert rel_l2(a[:2].cpu(), small.cpu()) < 1e-6, precision
        tower.set_ksplit(True)
        small = tower.encode_image(big[:2].clone(), normalize=True)
        _report(f"ViT-L/14@336 {precision}: batch 2 (K-split tiles) against the same rows of batch 128", small.cpu(), a[:2].cpu())
        assert rel_l2(a[:2].cpu(), small.cpu()) < {"fp8": 1e-6, "fp16": TOL, "bf16": TOL_TAIL_SPLIT}[precision], precision
        tower.close()
        del tower
        torch.cuda.empty_cache()


def test_ksplit_flag_changes_summation_order_only_and_fullmodel_restores_it(eng):
    """`TAPCLIP_FLAG_KSPLIT` (include/tapclip.h): 1 K-splits the tiles of partial GEMM rounds over idle CUs, 0 does not.
    ViT-B/16 at batch 256 (BASELINE configs[1]): c_proj's 79 tail tiles are the only ones the image tower splits, so the
    two settings agree bit for bit on the rows of the whole rounds and to the bf16 mode's rounding on the rest; each
    setting is run-to-run bit-identical.  FullModel clears the flag for its forward passes (both towers in flight:
    CU-time, not latency -- reference models/model_wrapper.py:40-75) and must leave both towers with it set again."""
    from tap_clip_amd.models import CLIPWrapper, FullModel
    from test_gpu_parity import TOL_TAIL_SPLIT

    cfg = configs.get_config("ViT-B-16")
    sd = synth.make_state_dict(cfg, seed=2)
    clip = CLIPWrapper("ViT-B-16", None, DEV, precision="bf16", attn_semantics="intended", state_dict=sd)
    tower = clip._vision
    tower.set_prune_last_block(False)
    images = synth.make_images(256, cfg, 0).to(DEV)
    on = tower.encode_image(images, normalize=True).clone()
    tower.set_ksplit(False)
    off = tower.encode_image(images, normalize=True).clone()
    assert torch.equal(off, tower.encode_image(images, normalize=True))
    tower.set_ksplit(True)
    assert torch.equal(on, tower.encode_image(images, normalize=True))
    assert torch.equal(on[:128], off[:128]), "rows of the whole rounds must not depend on the flag"
    assert not torch.equal(on, off) and rel_l2(off.cpu(), on.cpu()) < TOL_TAIL_SPLIT
    with contextlib.redirect_stdout(sys.stderr):
        model = FullModel([f"class_{i}" for i in range(65)], clip, prompt_len=16, class_specific=True).eval()
    with torch.no_grad():
        a = model(images)["logits"].clone()
        assert torch.equal(a, model(images)["logits"])
    assert torch.equal(on, tower.encode_image(images, normalize=True)), "FullModel must hand the towers back with the K-split on"
    model.train()
    out = model(images, synth.make_labels(256, 65).to(DEV))
    out["loss"].backward()
    assert torch.equal(on, tower.encode_image(images, normalize=True))


@pytest.mark.parametrize("name,batch", [("ViT-B-16", 64), ("ViT-B-32", 8), ("ViT-B-32-quickgelu", 8), ("ViT-L-14-336", 3), ("tiny", 1100)])
def test_pruned_last_block_equals_the_full_computation(eng, name, batch):
    """The library default computes the image tower's LAST block for the CLS rows only (K and V for every token; Q, the
    attention core, out_proj, LN2, the MLP for the pooled row: include/tapclip.h TAPCLIP_FLAG_PRUNE_LAST_BLOCK) -- the rows
    the reference's pooling throws away (models/clip_wrapper.py:46-47).  Against the same tower computing every row of every
    block, in every precision that has the path: equal to the precision's own rounding (the pooled row's softmax and P.V
    run in fp32, the full kernel rounds P to 16 bits) -- bf16x3 to 1e-5, fp16 to 2e-4, bf16 to 1.5e-3 -- deterministic, and
    independent of the batch the image sits in (ragged batch sizes included).  (The QuickGELU config: the skinny GEMM's
    finalize kernel and the tiled epilogues apply the same activation variant per precision -- ADVICE r03.)  ("tiny" at batch 1100: more rows than one
    skinny-GEMM launch takes -- they go through it in chunks of 1024, K slices depending on (N, K) only, so the pooled rows
    stay on the skinny path; width 128, so the residual stream is fp32.)"""
    cfg = configs.get_config(name)
    sd = synth.make_state_dict(cfg, seed=2, text=False)
    images = synth.make_images(batch, cfg, 11).to(DEV)
    # fp8 (widths the MXFP8 GEMM takes): the pooled rows' last block runs on 16-bit copies of that block's weights, so
    # against the all-MXFP8 computation they differ by one block's MXFP8 rounding of the CLS row (4 % per GEMM: section 2)
    modes = [("bf16x3", 1e-5), ("fp16", 2e-4), ("bf16", 1.5e-3)] + ([("fp8", 3e-2)] if cfg.vision.width % 256 == 0 else [])
    for precision, tol in modes:
        full = eng.VisionTower(cfg, sd, DEV, precision, prune_last_block=False)
        pruned = eng.VisionTower(cfg, sd, DEV, precision)  # the default
        a = full.encode_image(images, normalize=True)
        b = pruned.encode_image(images, normalize=True)
        _report(f"{name} {precision}: CLS-only last block vs full computation", b.cpu(), a.cpu())
        assert rel_l2(b.cpu(), a.cpu()) < tol and rel_max(b.cpu(), a.cpu()) < 2 * tol
        assert torch.equal(b, pruned.encode_image(images, normalize=True))
        one = pruned.encode_image(images[:1].clone(), normalize=True)
        assert torch.equal(one, b[:1]), "a pruned embedding must not depend on its batch mates"
        pruned.set_prune_last_block(False)
        assert torch.equal(pruned.encode_image(images, normalize=True), a), "the flag must switch the same handle back to the full computation"
        for tw in (full, pruned):
            tw.close()
        del full, pruned
        torch.cuda.empty_cache()


def test_skinny_gemm_path_of_the_pooled_rows_vs_oracle(eng):
    """The M = batch GEMMs of the pooled last block (gemm_skinny.hip: split-K slabs + finalize) end to end: a ONE-block
    ViT-B/16-width tower, so the pooled path is the whole tower, at batch sizes that exercise one and several 256-row
    blocks and ragged row counts, bf16x3 against the fp32 oracle at 1e-3."""
    cfg = configs.ClipDims("blk1", 512, 224, 16, configs.TowerDims(768, 1, 12, 3072), configs.TowerDims(512, 1, 8, 2048), vocab=16, ctx=8)
    ocfg = clip_ref.ClipDims("blk1", 512, 224, 16, clip_ref.TowerDims(768, 1, 12, 3072), clip_ref.TowerDims(512, 1, 8, 2048), vocab=16, ctx=8)
    sd = synth.make_state_dict(cfg, seed=6, text=False)
    tower = eng.VisionTower(cfg, sd, DEV, "bf16x3")
    for batch in (1, 7, 300):
        images = synth.make_images(batch, cfg, 20 + batch)
        with torch.no_grad():
            ref = clip_ref.encode_image(images[:8], sd, ocfg)
        emb = tower.encode_image(images.to(DEV)).cpu()
        assert rel_max(emb[:8], ref) < TOL, (batch, rel_max(emb[:8], ref))
        assert bool(torch.isfinite(emb).all())


def test_c_abi_allgather_over_rccl_one_rank():
    """tapclip_comm_* / tapclip_allgather (include/tapclip.h: the exchange step for hosts without torch.distributed) on a
    communicator of ONE rank -- all a 1-GPU box can host: unique id, ncclCommInitRank, a 512-KiB all-gather of the size
    configs[3] moves per rank on the caller's stream, destroy.  In a child process, so that the library's own dlopen of
    librccl is what runs (not a copy a torch process group already initialised)."""
    code = r"""
import ctypes as C, sys, torch
sys.path.insert(0, %r)
import tap_clip_amd
from tap_clip_amd import _lib
lib = _lib.load()
torch.cuda.set_device(0)
idbuf = (C.c_char * 128)()
_lib.check(lib.tapclip_comm_unique_id(idbuf))
comm = C.c_void_p()
_lib.check(lib.tapclip_comm_create(idbuf, 0, 1, C.byref(comm)))
x = torch.nn.functional.normalize(torch.randn(256, 512, device="cuda"), dim=-1)
y = torch.zeros_like(x)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
_lib.check(lib.tapclip_comm_check(comm))
_lib.check(lib.tapclip_allgather(comm, x.data_ptr(), y.data_ptr(), x.numel() * 4, st))
done = torch.cuda.Event(); done.record()
import time
t0 = time.time()
while not done.query():  # the poll a real exchange step uses: the async state while the gather's event is pending, with a deadline
    _lib.check(lib.tapclip_comm_check(comm))
    assert time.time() - t0 < 60, "all-gather did not complete"
    time.sleep(1e-4)
_lib.check(lib.tapclip_comm_check(comm))
torch.cuda.synchronize()
assert torch.equal(x, y)
lib.tapclip_comm_destroy(comm)
print("abi allgather ok")
""" % ROOT
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "abi allgather ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_literal_replay_refuses_to_train():
    g = golden("fullmodel_intended_tiny")
    model, images = _build_full("tiny", g, "intended", "bf16", collapse=False)
    model.train()
    with pytest.raises(RuntimeError, match="no backward"):
        model(images, torch.from_numpy(g["labels"]).to(DEV))
    with torch.no_grad():
        assert model(images)["logits"].shape == (4, 3)


# ---- towers on two streams (FullModel(overlap_towers=True) runs the image tower beside the text tower) -----------------
@pytest.mark.parametrize("name,batch,precision", [("ViT-B-32", 8, "bf16"), ("ViT-B-32", 8, "fp16"), ("ViT-B-16", 64, "bf16"),
                                                  ("ViT-B-16", 64, "fp8")])
def test_image_tower_is_bit_stable_beside_a_busy_second_stream(eng, name, batch, precision):
    """`encode_image` on a side stream while the main stream runs (a) the split-bf16 text tower, (b) torch matmuls:
    every run must equal the solo run bit for bit.  (Round 2: the tower on 24-bit residual planes went wrong 19 times
    in 20 under exactly this load while every solo test passed -- its LayerNorms held v_pk_fma_f32 with op_sel:[0,1,0],
    which MI355X gets wrong in lanes 48..63 beside another kernel's MFMAs; csrc/common.h TAPCLIP_TU_NO_PK_F32,
    tests/test_abi.py::test_no_unsafe_packed_fp32_encodings, test_packed_fp32_probe_* below.)"""
    cfg = configs.get_config(name)
    sd = synth.make_state_dict(cfg, seed=2)
    images = synth.make_images(batch, cfg, 0).to(DEV)
    ctx, tok = synth.make_prompts(10, 5, cfg, seed=1)
    prompts = torch.cat([ctx, tok], 1).to(DEV)
    text = eng.TextTower(cfg, sd, DEV, "bf16x3")
    tower = eng.VisionTower(cfg, sd, DEV, precision)
    big = torch.randn(2048, 2048, device=DEV, dtype=torch.bfloat16)
    base = tower.encode_image(images, normalize=True).clone()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()

    def neighbour_text():
        text.forward(prompts, want_hidden=False, want_mean=True)
        text.forward(prompts)

    def neighbour_matmul():
        for _ in range(20):
            big @ big

    n_it = int(os.environ.get("TAPCLIP_STABILITY_ITERS", "10"))  # (soak runs: 200)
    for label, nb in (("text tower bf16x3", neighbour_text), ("torch matmul", neighbour_matmul)):
        differ = 0
        for _ in range(n_it):
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                e = tower.encode_image(images, normalize=True)
            nb()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            differ += int(not torch.equal(e, base))
        assert differ == 0, f"{name} batch {batch} {precision} beside {label}: {differ}/{n_it} runs differ from the solo run"


@pytest.mark.parametrize("precision", ["fp16", "bf16", "fp8"])
def test_long_sequence_tower_is_bit_stable_beside_a_busy_second_stream(eng, precision):
    """The same for the 577-token geometry (ViT-L/14@336 widths, 2 blocks, batch 16): the LDS-DMA attention kernel of round 5
    (csrc/attention_long.hip) keeps hand-counted vmcnt / lgkmcnt queues and runs VALU arithmetic on MFMA accumulators beside its
    own LDS-fed MFMAs; a timing-dependent fault of an earlier build (an asm statement that read MFMA results unpadded:
    profiles/r05_flash2_asm_hazard.txt) showed as other wrong rows every run.  Every run beside a busy neighbour must equal the
    solo run bit for bit."""
    cfg = configs.ClipDims("L14-336-2blocks", 768, 336, 14, configs.TowerDims(1024, 2, 16, 4096), configs.TowerDims(512, 2, 8, 2048), vocab=512)
    sd = synth.make_state_dict(cfg, seed=2)
    images = synth.make_images(16, cfg, 0).to(DEV)
    ctx, tok = synth.make_prompts(10, 5, cfg, seed=1)
    prompts = torch.cat([ctx, tok], 1).to(DEV)
    text = eng.TextTower(cfg, sd, DEV, "bf16x3")
    tower = eng.VisionTower(cfg, sd, DEV, precision, prune_last_block=False)
    big = torch.randn(2048, 2048, device=DEV, dtype=torch.bfloat16)
    base = tower.encode_image(images, normalize=True).clone()
    assert bool(torch.isfinite(base).all())
    torch.cuda.synchronize()
    side = torch.cuda.Stream()

    def neighbour_text():
        text.forward(prompts, want_hidden=False, want_mean=True)
        text.forward(prompts)

    def neighbour_matmul():
        for _ in range(20):
            big @ big

    n_it = int(os.environ.get("TAPCLIP_STABILITY_ITERS", "10"))  # (soak runs: 200)
    for label, nb in (("nothing", lambda: None), ("text tower bf16x3", neighbour_text), ("torch matmul", neighbour_matmul)):
        differ = 0
        for _ in range(n_it):
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                e = tower.encode_image(images, normalize=True)
            nb()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            differ += int(not torch.equal(e, base))
        assert differ == 0, f"577 tokens {precision} beside {label}: {differ}/{n_it} runs differ from the solo run"


def test_text_tower_is_bit_stable_beside_the_image_tower(eng):
    """The other direction: the text tower (bf16 and split-bf16; its LayerNorms run beside the image tower's GEMMs in
    FullModel(overlap_towers=True)) on the main stream while a side stream runs the image tower."""
    cfg = configs.get_config("ViT-B-16")
    sd = synth.make_state_dict(cfg, seed=4)
    images = synth.make_images(64, cfg, 0).to(DEV)
    ctx, tok = synth.make_prompts(65, 5, cfg, seed=1)
    prompts = torch.cat([ctx, tok], 1).to(DEV)
    tower = eng.VisionTower(cfg, sd, DEV, "bf16")
    side = torch.cuda.Stream()
    for precision in ("bf16", "bf16x3"):
        text = eng.TextTower(cfg, sd, DEV, precision)
        base = [t.clone() for t in text.forward(prompts, want_mean=True).values() if torch.is_tensor(t)]
        torch.cuda.synchronize()
        for it in range(int(os.environ.get("TAPCLIP_STABILITY_ITERS", "10"))):
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                tower.encode_image(images, normalize=True)
                tower.encode_image(images, normalize=True)
            got = [t for t in text.forward(prompts, want_mean=True).values() if torch.is_tensor(t)]
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            for g, b in zip(got, base):
                assert torch.equal(g, b), f"text tower ({precision}) beside the image tower, iteration {it}: {(g - b).abs().max().item():.3e}"


def _pk_probe(tmp_path):
    import shutil, subprocess
    exe = os.path.join(ROOT, "tools", "probes", "pk_opsel_table")
    if not os.path.exists(exe):
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        src = os.path.join(ROOT, "tools", "probes", "pk_opsel_table.hip")
        if not (os.path.exists(hipcc) and os.path.exists(src)):
            pytest.skip("tools/probes/pk_opsel_table is not built and cannot be built here")
        exe = str(tmp_path / "pk_opsel_table")
        subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", "-Wno-inline-asm", "-o", exe, src], check=True, capture_output=True)
    out = subprocess.run([exe, "30"], capture_output=True, text=True, timeout=300, check=True).stdout
    alone, beside = out.split("== beside")
    return alone, beside


def test_packed_fp32_probe_every_failing_encoding_is_one_the_library_check_rejects(tmp_path):
    """tools/probes/pk_opsel_table: all 96 op_sel / op_sel_hi encodings of v_pk_{fma,mul,add}_f32 against scalar ops.
    Alone none fails.  Beside a GEMM-like kernel on a second stream the op_sel = [0,1,...] ones fail (lanes 48..63):
    that is the set tests/test_abi.py keeps out of the library -- if ANY other encoding ever fails here, that check is
    no longer sufficient."""
    alone, beside = _pk_probe(tmp_path)
    assert re.search(r"\b0 of \d+ encodings returned wrong results", alone), alone
    bad = [l for l in beside.splitlines() if "UNSAFE" in l]
    outside = [l for l in bad if "op_sel:[0,1" not in l]
    assert not outside, "encodings outside op_sel = [0,1,...] failed:\n" + "\n".join(outside)
    for l in bad:  # and only in the last 16-lane quad
        q = [int(v) for v in l.split("quad:")[1].split()]
        assert q[0] == q[1] == q[2] == 0, l
    if not bad:
        # ONE probe run, as always (never looped or lengthened to provoke the erratum): when it shows no failing
        # encoding at all -- the victim's waves did not share a SIMD with the neighbour's on this box -- the run says
        # nothing about which encodings are unsafe, and a pass would overstate it.  A PASS of this test means: the
        # failing set was seen, and nothing outside op_sel = [0,1,...] failed.
        pytest.skip("probe silent on this box: no encoding failed beside the neighbour, so the failing set was not observed")


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_fullmodel_overlapped_towers_equal_the_serial_run_bit_for_bit(precision):
    """FullModel(overlap_towers=True) -- image tower on a second stream beside the text tower, the default -- against
    overlap_towers=False on the same weights: identical logits, every time (configs[2]'s shapes at batch 64; forward
    and one training step's context gradients)."""
    g = golden("fullmodel_intended_vitb16_c65")
    model, _ = _build_full("ViT-B-16", g, "intended", precision)
    cfg = configs.get_config("ViT-B-16")
    images = synth.make_images(64, cfg, 5).to(DEV)
    labels = (torch.arange(64) % 65).to(DEV)
    model.eval()
    model.overlap_towers = False
    with torch.no_grad():
        base = model(images)["logits"].clone()
    model.train()
    model(images, labels)["loss"].backward()
    names = g["class_names"].tolist()
    base_grad = torch.stack([model.prompt_learner.context_bank[c].grad.clone() for c in names], 0)
    model.overlap_towers = True
    for it in range(int(os.environ.get("TAPCLIP_STABILITY_ITERS", "10"))):
        model.eval()
        with torch.no_grad():
            got = model(images)["logits"]
        assert torch.equal(got, base), f"{precision}, forward {it}: overlapped towers differ from the serial run by {(got - base).abs().max().item():.3e}"
        model.train()
        for c in names:
            model.prompt_learner.context_bank[c].grad = None
        model(images, labels)["loss"].backward()
        grad = torch.stack([model.prompt_learner.context_bank[c].grad for c in names], 0)
        assert torch.equal(grad, base_grad), f"{precision}, training step {it}: context gradients differ by {(grad - base_grad).abs().max().item():.3e}"
