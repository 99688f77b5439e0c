"""The HIP towers against an INDEPENDENT implementation at BASELINE's real dimensions: HF `transformers` CLIP at ViT-B/16
dims with the seeded weights (tests/golden/hf_clip_vitb16.npz; oracle/make_golden.py::g_hf_clip_vitb16 through
oracle/hf_harness.py).  The reference's towers are open_clip's (reference models/clip_wrapper.py:13,47,51;
models/model_wrapper.py:58,72), absent here; these outputs come from code that shares nothing with oracle/clip_ref.py, so
the 1e-3 of BASELINE.json is held here against something other than this repository's own restatement."""
import pytest
import torch

import tap_clip_amd  # noqa: F401
from conftest import golden, rel_l2, rel_max
from tap_clip_amd import configs, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-3


@pytest.fixture(scope="module")
def case():
    g = golden("hf_clip_vitb16")
    cfg = configs.get_config("ViT-B-16")
    sd = synth.make_state_dict(cfg, seed=int(g["seed_weights"]))
    tokens = torch.from_numpy(g["token_ids"])
    n = tokens.shape[0]
    ctx = synth.make_prompts(65, int(g["prompt_len"]), cfg, seed=int(g["seed_context"]))[0][:n]
    prompts = torch.cat([ctx, sd["token_embedding.weight"][tokens]], dim=1)
    images = synth.make_images(int(g["batch"]), cfg, int(g["seed_images"]))
    return g, cfg, sd, tokens, prompts, images


@pytest.mark.parametrize("precision", ["bf16x3", "fp16"])
def test_image_embeddings_vs_hf(case, precision):
    from tap_clip_amd import engine

    g, cfg, sd, _, _, images = case
    ref = torch.from_numpy(g["image_embeddings"])
    for prune in (False, True):  # every row of every block, and the library's default CLS-only last block
        tower = engine.VisionTower(cfg, sd, DEV, precision, prune_last_block=prune)
        emb = tower.encode_image(images.to(DEV)).cpu()
        print(f"[hf] image {precision} prune={prune}: rel_max {rel_max(emb, ref):.2e} rel_l2 {rel_l2(emb, ref):.2e}")
        assert rel_max(emb, ref) < TOL and rel_l2(emb, ref) < TOL


def test_text_tower_vs_hf(case):
    """split-bf16 text tower (the text tower of the library's default mode): encode_text features, and the raw transformer
    as FullModel drives it -- hidden rows and last-layer probabilities -- on every row and on the distinct rows only."""
    from tap_clip_amd import engine

    g, cfg, sd, tokens, prompts, _ = case
    tower = engine.TextTower(cfg, sd, DEV, "bf16x3")
    x = tower.embed_tokens(tokens.to(DEV), add_pos=True)
    hid = tower.forward(x, causal=True)["hidden"]
    feat = tower.pool_project(hid, index=tokens.argmax(dim=-1), ln_final=True).cpu()
    ref = torch.from_numpy(g["text_features"])
    print(f"[hf] encode_text: rel_max {rel_max(feat, ref):.2e}")
    assert rel_max(feat, ref) < TOL
    k = g["raw_hidden"].shape[0]
    run = tower.tail_run(prompts.to(DEV))
    assert run >= 60
    for tail_run in (1, run):
        r = tower.forward(prompts.to(DEV), want_heads=True, want_mean=True, tail_run=tail_run)
        h, heads, mean = r["hidden"].cpu(), r["attn_heads"].cpu(), r["attn_mean"].cpu()
        errs = (rel_max(h[:k], torch.from_numpy(g["raw_hidden"])), rel_max(h[:, -1], torch.from_numpy(g["raw_hidden_last"])),
                rel_max(mean[:k], torch.from_numpy(g["raw_attn_mean"])), rel_max(heads[:k, :, :4], torch.from_numpy(g["raw_probs_rows"])))
        print(f"[hf] raw text transformer tail_run={tail_run}: hidden {errs[0]:.2e} last row {errs[1]:.2e} map {errs[2]:.2e} per-head rows {errs[3]:.2e}")
        assert max(errs) < TOL
    assert not tower.tied_violations()
