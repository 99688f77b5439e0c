"""Tied padding rows (include/tapclip.h "tied padding rows", csrc/tied.hip): the text tower on the DISTINCT rows of every
sequence against the same tower on every row.

Why the identity holds: reference models/prompt_learner.py:31-34 builds a class prompt's token rows as
`token_embedding(tokenizer(text))` -- zero-padded ids, so every padding position carries one embedding row -- and
reference models/model_wrapper.py:58,72 runs the transformer on [context | tokens] with neither positional embedding nor
mask.  Identical rows stay identical through every block; the tied entry points keep one of them and count its key
`tail_run` times in every softmax.  The FullModel goldens (tests/test_gpu_parity.py, test_gpu_configs.py: the reference's
own FullModel on zero-padded prompts) run through this path by default; here it is compared with the untied computation
directly, forward and backward, and the device-side check of the caller's claim is exercised."""
import numpy as np
import pytest
import torch

import tap_clip_amd  # noqa: F401
from conftest import golden, rel_l2, rel_max
from tap_clip_amd import configs, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _prompts(n, P, L, D, runs, seed=7):
    """[n, P + L, D]: P context rows ~ N(0, 1), L token rows ~ N(0, 0.02^2) whose last runs[i] rows repeat one row"""
    x = torch.cat([synth.normal([n, P, D], seed, "tied.ctx"), synth.normal([n, L, D], seed, "tied.tok", 0.02)], dim=1)
    for i in range(n):
        r = runs[i % len(runs)]
        x[i, P + L - r:] = x[i, -1]
    return x


def _text_tower(name, precision):
    from tap_clip_amd import engine

    cfg = configs.get_config(name)
    sd = synth.make_state_dict(cfg, seed=2, vision=False)
    return engine.TextTower(cfg, sd, DEV, precision), cfg


def test_tail_run_detection():
    tower, cfg = _text_tower("tiny", "bf16")
    D = cfg.text.width
    x = _prompts(5, 4, 20, D, [7, 9, 12])
    assert tower.tail_run(x.to(DEV)) == 7
    x[2, -3] += 1e-3                                   # one sequence's run is cut to 2
    assert tower.tail_run(x.to(DEV)) == 2
    assert tower.tail_run(synth.normal([3, 11, D], 1, "tied.none").to(DEV)) == 1
    same = torch.ones(2, 13, D) * 0.25                 # every row identical: the whole sequence is one run
    assert tower.tail_run(same.to(DEV)) == 13
    neg0 = same.clone()
    neg0[:, :, 0] = 0.0
    neg0[1, 5, 0] = -0.0                               # -0.0 == 0.0 in value, not in bits: the check is bitwise
    assert tower.tail_run(neg0.to(DEV)) == 13 - 5 - 1


@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
@pytest.mark.parametrize("name,n,P,runs", [("ViT-B-16", 9, 16, [68, 70, 69]), ("tiny", 4, 5, [3, 60])])
def test_tied_forward_equals_untied(precision, name, n, P, runs):
    tower, cfg = _text_tower(name, precision)
    x = _prompts(n, P, cfg.ctx, cfg.text.width, runs).to(DEV)
    run = tower.tail_run(x)
    assert run == min(runs)
    kw = dict(want_hidden=True, want_heads=True, want_mean=True, want_attn_out=True)
    full = tower.forward(x, **kw)
    tied = tower.forward(x, tail_run=run, **kw)
    assert not tower.tied_violations()
    # split-bf16: two evaluations of the same function on operands that carry 16 significand bits (hi + lo): they agree to
    # a few 2^-16 (measured 1.1e-5 hidden, 2.1e-5 gradients), a tenth of the 1e-3 parity bound; bf16: two roundings of the
    # same function (the merged key's un-normalised probability is rounded once instead of `run` times)
    tol = 1e-4 if precision == "bf16x3" else 1.5e-2
    for k in ("hidden", "attn_heads", "attn_mean", "attn_out"):
        a, b = tied[k].cpu(), full[k].cpu()
        assert a.shape == b.shape and torch.isfinite(a).all()
        print(f"[tied] {name} {precision} {k}: rel_max {rel_max(a, b):.2e} rel_l2 {rel_l2(a, b):.2e}")
        assert rel_l2(a, b) < tol and rel_max(a, b) < 4 * tol, k
    T = x.shape[1]
    assert torch.allclose(tied["attn_heads"].sum(-1).cpu(), torch.ones(n, cfg.text.heads, T), atol=1e-5)
    # the rows / columns of the run are copies of each other
    assert torch.equal(tied["hidden"][:, T - run:], tied["hidden"][:, T - 1:].expand(-1, run, -1))
    assert torch.equal(tied["attn_mean"][:, :, T - run:], tied["attn_mean"][:, :, T - 1:].expand(-1, -1, run))
    # capture-only call (pass 1 of FullModel): the map alone
    cap = tower.forward(x, want_hidden=False, want_mean=True, tail_run=run)
    assert cap["hidden"] is None and torch.equal(cap["attn_mean"], tied["attn_mean"])
    # a shorter run than the real one is a true claim too
    part = tower.forward(x, tail_run=run - 1, **kw)
    assert rel_l2(part["hidden"].cpu(), full["hidden"].cpu()) < tol


@pytest.mark.parametrize("run", [13, 12])
def test_tied_forward_at_the_degenerate_ends(run):
    """A sequence that is ONE row repeated (tail_run = T: a single distinct row, its own key counted T times) and one with a
    single other row in front: the merged tower still equals the untied one."""
    tower, cfg = _text_tower("tiny", "bf16x3")
    D, T = cfg.text.width, 13
    x = synth.normal([3, 1, D], 9, "tied.one").expand(3, T, D).clone()
    if run == 12:
        x[:, 0] = synth.normal([3, D], 9, "tied.first")
    x = x.to(DEV)
    assert tower.tail_run(x) == run
    full = tower.forward(x, want_mean=True, want_heads=True)
    tied = tower.forward(x, want_mean=True, want_heads=True, tail_run=run)
    for k in ("hidden", "attn_mean", "attn_heads"):
        assert rel_max(tied[k].cpu(), full[k].cpu()) < 1e-4, k
    assert not tower.tied_violations()


def test_false_claim_poisons_the_outputs_until_acknowledged():
    tower, cfg = _text_tower("tiny", "bf16x3")
    x = _prompts(3, 5, cfg.ctx, cfg.text.width, [10]).to(DEV)
    ok = tower.forward(x, want_mean=True, tail_run=10)
    assert torch.isfinite(ok["hidden"]).all() and not tower.tied_violations()
    bad = tower.forward(x, want_mean=True, want_heads=True, want_attn_out=True, tail_run=11)   # row T-11 differs
    for k in ("hidden", "attn_mean", "attn_heads", "attn_out"):
        assert torch.isnan(bad[k]).all(), k
    again = tower.forward(x, want_mean=True, tail_run=10)                                        # still poisoned
    assert torch.isnan(again["hidden"]).all()
    assert tower.tied_violations() is True          # reported ...
    assert tower.tied_violations() is False         # ... and cleared
    fine = tower.forward(x, want_mean=True, tail_run=10)
    assert torch.equal(fine["hidden"], ok["hidden"]) and torch.equal(fine["attn_mean"], ok["attn_mean"])
    with pytest.raises(ValueError):
        tower.forward(x, tail_run=x.shape[1] + 1)
    with pytest.raises(ValueError):
        tower.forward(x, causal=True, tail_run=4)


@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
@pytest.mark.parametrize("name,n,P,runs", [("ViT-B-16", 6, 16, [68, 70]), ("tiny", 4, 5, [40, 44])])
def test_tied_backward_equals_untied(precision, name, n, P, runs):
    """The training pair: dL/dx of the rows before the run (the context rows FullModel differentiates, reference
    train.py:99-105) equals the untied gradient; the run's gradient comes back summed in its first row."""
    tower, cfg = _text_tower(name, precision)
    D = cfg.text.width
    x = _prompts(n, P, cfg.ctx, D, runs).to(DEV)
    T = x.shape[1]
    run = min(runs)
    gfeat = synth.normal([n, cfg.embed_dim], 4, "tied.g").to(DEV)

    def step(tail_run):
        hidden, saved = tower.forward_saved(x, tail_run=tail_run)
        g_hidden = tower.pool_project_backward(hidden, gfeat, normalize=True)
        return hidden, tower.backward_saved(saved, g_hidden, tail_run=tail_run)

    h_full, g_full = step(1)
    h_tied, g_tied = step(run)
    assert not tower.tied_violations()
    tol = 1e-4 if precision == "bf16x3" else 3e-2
    print(f"[tied] {name} {precision} hidden {rel_l2(h_tied.cpu(), h_full.cpu()):.2e} "
          f"grad(ctx rows) {rel_l2(g_tied[:, :P].cpu(), g_full[:, :P].cpu()):.2e}")
    assert rel_l2(h_tied.cpu(), h_full.cpu()) < tol
    first = T - run
    assert rel_l2(g_tied[:, :first].cpu(), g_full[:, :first].cpu()) < tol
    assert rel_max(g_tied[:, :P].cpu(), g_full[:, :P].cpu()) < 4 * tol
    assert rel_l2(g_tied[:, first].cpu(), g_full[:, first:].sum(dim=1).cpu()) < tol
    assert float(g_tied[:, first + 1:].abs().max()) == 0.0


@pytest.mark.parametrize("semantics", ["intended", "literal"])
def test_fullmodel_with_and_without_tied_padding(semantics):
    """FullModel on the reference's zero-padded prompts (golden case): tie_padding on (the default) and off give the same
    logits, capture, attribution, loss and gradients -- and both match the reference's own FullModel."""
    from test_gpu_parity import _build_full

    g = golden(f"fullmodel_{semantics}_tiny")
    ref = torch.from_numpy(g["logits"])
    res = {}
    for tie in (True, False):
        model, images = _build_full("tiny", g, semantics, "bf16x3")
        model.tie_padding = tie
        run = model._tail_run()
        assert (run > 1) == tie
        model.train()
        out = model(images, torch.from_numpy(g["labels"]).to(DEV))
        out["loss"].backward()
        names = g["class_names"].tolist()
        res[tie] = dict(logits=out["logits"].detach().cpu(), loss=float(out["loss"]),
                        grad=torch.stack([model.prompt_learner.context_bank[c].grad for c in names], 0).cpu(),
                        amap=model.clip.attention_maps[0].cpu(), attr=model.last_attribution.cpu())
        assert rel_max(res[tie]["logits"], ref) < 1e-3
        assert rel_max(res[tie]["grad"], torch.from_numpy(g["context_grad"])) < 1e-3
        assert not model.clip._text.tied_violations()
    for k in ("logits", "grad", "amap", "attr"):
        assert res[True][k].shape == res[False][k].shape
        assert rel_max(res[True][k], res[False][k]) < 1e-4, k
    assert abs(res[True]["loss"] - res[False]["loss"]) < 1e-4


def test_token_bank_change_re_measures_the_run():
    from test_gpu_parity import _build_full

    g = golden("fullmodel_intended_tiny")
    model, _ = _build_full("tiny", g, "intended", "bf16")
    pl = model.prompt_learner
    r0 = pl.tail_run()
    assert 1 < r0 < 77
    ids = torch.from_numpy(g["token_ids"][:1]).clone()
    ids[0, 77 - (r0 - 3)] = 3                          # a longer prompt: four padding positions fewer
    pl.tokenizer = lambda text: ids.clone()            # (PromptLearner keeps the tokenizer it was built with)
    pl.add_class_prompt("Unseen_Thing")                # (reference test_cross_domain.py:65-67)
    assert pl.tail_run() == r0 - 4
    with torch.no_grad():
        images = synth.make_images(2, configs.get_config("tiny"), 0).to(DEV)
        out = model(images)["logits"]
    assert out.shape == (2, len(g["class_names"]) + 1) and torch.isfinite(out).all()
    assert not model.clip._text.tied_violations()


def test_direct_token_bank_edit_is_diagnosed_not_just_nan():
    """`token_bank` is a public dict; an edit that goes around add_class_prompt / refresh_token_bank leaves the cached
    run length stale, the library poisons its outputs (NaN, by design) -- and `check_tied_padding` (called by the evaluation loops
    at their end) names the cause and drops the cache, so that the next forward measures the bank again and is finite."""
    from test_gpu_parity import _build_full
    from tap_clip_amd.utils import eval_metrics

    g = golden("fullmodel_intended_tiny")
    model, images = _build_full("tiny", g, "intended", "bf16")
    with torch.no_grad():
        assert torch.isfinite(model(images)["logits"]).all()
    model.check_tied_padding()  # nothing to report
    pl = model.prompt_learner
    name = next(iter(pl.token_bank))
    edited = pl.token_bank[name].clone()
    edited[..., -3, :] += 1.0      # one of the "identical" padding rows is no longer identical
    pl.token_bank[name] = edited
    pl._tok_cache = None           # (the stacked copy is rebuilt from the dict; the run length measured before is now stale)
    with torch.no_grad():
        bad = model(images)["logits"]
    assert torch.isnan(bad).any()
    with pytest.raises(RuntimeError, match="tie_padding"):
        model.check_tied_padding()
    with torch.no_grad():
        again = model(images)["logits"]  # the run was re-measured: the edited row is outside it now
    assert torch.isfinite(again).all()
    labels = torch.zeros(images.shape[0], dtype=torch.int64)
    eval_metrics.evaluate_accuracy(model, [(images.cpu(), labels)], DEV)  # (and the loop's own check passes)
