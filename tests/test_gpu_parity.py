"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the committed
golden vectors (tests/golden/, produced from the reference's own classes by oracle/make_golden.py).

Tolerances (BASELINE.json north_star: 1e-3 relative):
  TOL        = 1e-3  bf16x3 (split-bf16, the parity mode) vs the fp32 oracle / goldens: the parity
                     claim.  Measured ~4e-6 on embeddings, ~2e-5 on logits.
  TOL_EMU    = 2e-3  (rel_l2) plain bf16 vs the oracle with operands rounded to bf16 at the same
                     points (`emulate="bf16"`).  Chained roundings amplify any eps to sqrt(eps * 2^-8):
                     two such pipelines cannot agree better than `clip_ref.emulation_floor` measures
                     (1.1e-3 rel-L2 / 3.0e-3 rel-max per ViT-B block); the block test bounds the kernels
                     at 1.5x that measured floor, rel-L2 and rel-max.
  TOL_BF16   = 2e-2  (unit kernels; the FullModel tests against the reference's goldens use 1.5x the MEASURED cost of the format
             instead: _bf16_floor)  plain bf16 vs the fp32 oracle: operand-quantisation noise of bf16 (8-bit
                     mantissa) through 12 layers; reported, bounded, not the parity claim.
`rel_max` = max|a-b| / max|b|, `rel_l2` = ||a-b|| / ||b||.
"""
import math

import numpy as np
import pytest
import torch

import tap_clip_amd  # noqa: F401
from conftest import golden, rel_l2, rel_max
from oracle import clip_ref, full_model_ref
from tap_clip_amd import configs, synth

pytestmark = pytest.mark.gpu

TOL = 1e-3
TOL_EMU = 2e-3
TOL_BF16 = 2e-2
# Two bf16-operand evaluations of ONE image that associate their fp32 sums differently (the K-split tail tiles of
# gemm256.hip): each sits one emulation floor -- 1.1e-3 rel-L2 per ViT-B block (clip_ref.emulation_floor,
# tests/test_oracle.py::test_emulation_floor_of_chained_bf16_rounding), 2.4e-3 on the L2-normalised embedding after 12
# blocks -- from the exact value with independent roundings, i.e. sqrt(2) x 2.4e-3 = 3.4e-3 from each other.  A fixed
# bound (it used to follow the run's own measured error); measured 2.1e-3.
TOL_TAIL_SPLIT = 3.5e-3
DEV = "cuda:0"


@pytest.fixture(scope="module")
def eng():
    from tap_clip_amd import engine

    return engine


def _report(tag, a, b):
    print(f"[parity] {tag}: rel_max={rel_max(a.cpu(), b):.3e} rel_l2={rel_l2(a.cpu(), b):.3e}")


# ---- unit kernels ---------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,d", [(50432 // 8, 768), (6045, 512), (37, 128), (5, 1024)])
def test_layernorm(eng, rows, d):
    x = synth.normal([rows, d], 1, "ln.x", 2.0, 0.3)
    g = synth.normal([d], 1, "ln.g", 0.1, 1.0)
    b = synth.normal([d], 1, "ln.b", 0.05)
    y = eng.layernorm(x.to(DEV), g.to(DEV), b.to(DEV)).cpu()
    ref = torch.nn.functional.layer_norm(x, (d,), g, b, 1e-5)
    assert rel_max(y, ref) < 1e-5


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 256, 192), (6045, 512, 512), (1000, 128, 2048)])
def test_gemm(eng, M, N, K):
    a = synth.normal([M, K], 2, "g.a")
    w = synth.normal([N, K], 2, "g.w", K**-0.5)
    bias = synth.normal([N], 2, "g.b", 0.1)
    ref = a.double() @ w.double().t() + bias.double()
    y3 = eng.gemm(a.to(DEV), w.to(DEV), bias.to(DEV), "bf16x3").cpu()
    assert rel_max(y3, ref.float()) < 1e-4, "bf16x3 GEMM must be fp32-grade"
    y1 = eng.gemm(a.to(DEV), w.to(DEV), bias.to(DEV), "bf16").cpu()
    ref16 = a.bfloat16().double() @ w.bfloat16().double().t() + bias.double()
    assert rel_max(y1, ref16.float()) < 1e-5, "bf16 GEMM must equal the product of bf16-rounded operands"


@pytest.mark.parametrize("tile,bn", [("256", "128"), ("256", "256")])
def test_gemm_large_tile_kernel(tile, bn):
    """The 256-row LDS-DMA kernel (used for the image tower's M = 50 432) on a ragged M, in a fresh
    process so the tile choice can be pinned through the environment."""
    import os, subprocess, sys
    code = r"""
import sys, torch
sys.path.insert(0, %r)
import tap_clip_amd
from tap_clip_amd import engine, synth
M, N, K = 8192 + 77, 512, 192
a = synth.normal([M, K], 2, "g.a"); w = synth.normal([N, K], 2, "g.w", K ** -0.5); b = synth.normal([N], 2, "g.b", 0.1)
ref = a.double() @ w.double().t() + b.double()
y3 = engine.gemm(a.cuda(), w.cuda(), b.cuda(), "bf16x3").cpu().double()
assert float((y3 - ref).abs().max() / ref.abs().max()) < 1e-4, "bf16x3"
ref16 = a.bfloat16().double() @ w.bfloat16().double().t() + b.double()
y1 = engine.gemm(a.cuda(), w.cuda(), b.cuda(), "bf16").cpu().double()
assert float((y1 - ref16).abs().max() / ref16.abs().max()) < 1e-5, "bf16"
eye = torch.eye(128); wa = (torch.arange(256 * 128, dtype=torch.float32).reshape(256, 128) %% 251) - 125.0
assert torch.equal(engine.gemm(eye.cuda(), wa.cuda(), None, "bf16").cpu(), wa.t())
print("ok")
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TAPCLIP_GEMM_TILE=tile, TAPCLIP_GEMM_BN=bn)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("tile", ["0", "1", "2", "", "off"])
def test_gemm_latency_kernel(tile):
    """The one-tile-per-CU LDS-DMA kernel of gemm_lat.hip (M < 2048: the text tower on its distinct rows) at each of its tile
    shapes (TAPCLIP_GEMM_LAT_TILE 0 / 1 / 2 = 128x128 / 128x64 / 64x64; "" = its own choice; "off" = the register-staged
    kernel it replaced), in a fresh process so the choice can be pinned: ragged M, K from 64 to 2048 (shorter and longer than
    its ring), one product against the exact product of the rounded operands, three products (ONE staging of hi and lo
    planes) against the fp64 product."""
    import os, subprocess, sys
    code = r"""
import sys, torch
sys.path.insert(0, %r)
import tap_clip_amd
from tap_clip_amd import engine, synth
for (M, N, K) in ((1560, 1536, 512), (777, 512, 2048), (70, 128, 64), (1, 128, 64), (2047, 2048, 128), (300, 256, 320)):
    a = synth.normal([M, K], 2, "g.a"); w = synth.normal([N, K], 2, "g.w", K ** -0.5); b = synth.normal([N], 2, "g.b", 0.1)
    ref = a.double() @ w.double().t() + b.double()
    y3 = engine.gemm(a.cuda(), w.cuda(), b.cuda(), "bf16x3").cpu().double()
    e3 = float((y3 - ref).abs().max() / ref.abs().max())
    assert e3 < 1e-4, ("bf16x3", M, N, K, e3)
    ref16 = a.bfloat16().double() @ w.bfloat16().double().t() + b.double()
    y1 = engine.gemm(a.cuda(), w.cuda(), b.cuda(), "bf16").cpu().double()
    e1 = float((y1 - ref16).abs().max() / ref16.abs().max())
    assert e1 < 1e-5, ("bf16", M, N, K, e1)
    y1n = engine.gemm(a.cuda(), w.cuda(), None, "bf16").cpu().double()
    assert float((y1n - (ref16 - b.double())).abs().max() / ref16.abs().max()) < 1e-5, "no bias"
eye = torch.eye(128); wa = (torch.arange(256 * 128, dtype=torch.float32).reshape(256, 128) %% 251) - 125.0
assert torch.equal(engine.gemm(eye.cuda(), wa.cuda(), None, "bf16").cpu(), wa.t())
print("ok")
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    if tile == "off":
        env["TAPCLIP_GEMM_LAT"] = "0"
    elif tile:
        env["TAPCLIP_GEMM_LAT_TILE"] = tile
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def test_gemm_identity_asymmetric(eng):
    """A = I with an asymmetric W catches a transposed C/D fragment map (guide section 3)."""
    K = N = 128
    w = (torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 251) - 125.0  # exact in bf16
    a = torch.eye(K)
    y = eng.gemm(a.to(DEV), w.to(DEV), None, "bf16").cpu()
    assert torch.equal(y, w.t())


# ---- one residual block at the real widths vs torch.nn.MultiheadAttention goldens ----------------
def _one_layer_tower(eng, d, heads, mlp, seed, precision, q_gain=1.0):
    cfg = configs.ClipDims("blk", 64, 224, 16, configs.TowerDims(d, 1, heads, mlp), configs.TowerDims(d, 1, heads, mlp), vocab=16, ctx=8)
    sd = {}
    synth._tower(sd, "transformer.", d, 1, mlp, seed=seed)
    if q_gain != 1.0:  # wider scores: q.k / 8 has sigma = q_gain on these weights
        sd["transformer.resblocks.0.attn.in_proj_weight"][:d] *= q_gain
        sd["transformer.resblocks.0.attn.in_proj_bias"][:d] *= q_gain
    sd["token_embedding.weight"] = torch.zeros(16, d)
    sd["positional_embedding"] = torch.zeros(8, d)
    sd["ln_final.weight"] = torch.ones(d)
    sd["ln_final.bias"] = torch.zeros(d)
    sd["text_projection"] = torch.zeros(d, 64)
    return eng.TextTower(cfg, sd, DEV, precision), sd


@pytest.mark.parametrize("tag", ["vision", "text"])
@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_block_vs_golden(eng, tag, precision):
    g = golden(f"block_{tag}")
    d, heads, mlp, n, T = (int(g[k]) for k in ("d", "heads", "mlp", "n", "T"))
    tower, sd = _one_layer_tower(eng, d, heads, mlp, int(g["seed_weights"]), precision)
    x = synth.normal([n, T, d], int(g["seed_x"]), f"block.{tag}.x")
    r = tower.forward(x.to(DEV), want_heads=True, want_mean=True, want_attn_out=True)
    tol = TOL if precision == "bf16x3" else TOL_BF16
    for name, got, key in (("hidden", r["hidden"], "out"), ("attn_out", r["attn_out"], "attn_out"),
                           ("attn_mean", r["attn_mean"], "probs_head_mean")):
        ref = torch.from_numpy(g[key])
        _report(f"block {tag} {precision} {name}", got, ref)
        assert rel_max(got.cpu(), ref) < tol, name
    assert rel_max(r["attn_heads"][:, 0, :8, :].cpu(), torch.from_numpy(g["probs_head0_rows"])) < tol
    assert torch.allclose(r["attn_heads"].sum(-1).cpu(), torch.ones(n, heads, T), atol=1e-5)
    if precision == "bf16":
        # Same rounding points and the same fitted GELU as the kernels.  What is left is NOT accumulation-order noise of
        # ~1e-6: pre-rounding values that differ by eps put a fraction eps/u of the rounded elements one whole bf16 step
        # u apart (sqrt(eps u) rms), and each further rounding stage amplifies again.  `emulation_floor` measures that
        # floor -- the emulation against ITSELF with fp64 accumulation: 5.6e-4 / 1.6e-3 at attn_out, 1.1e-3 / 3.0e-3 at
        # the block output for the vision widths -- and the kernels must sit on it (round 1 bounded rel-L2 only, after
        # a red run at 3.0e-3 rel-max; this is that number's cause).
        taps = {}
        y, _ = clip_ref.block_forward(x, sd, "transformer.resblocks.0.", heads, emulate="bf16", taps=taps)
        floor = clip_ref.emulation_floor(x, sd, "transformer.resblocks.0.", heads)
        for name, got, key in (("attn_out", r["attn_out"], "attn_out"), ("hidden", r["hidden"], "out")):
            l2, mx = rel_l2(got.cpu(), taps[key]), rel_max(got.cpu(), taps[key])
            print(f"[parity] block {tag} bf16 {name} vs emulated: rel_l2={l2:.3e} rel_max={mx:.3e}   "
                  f"floor (emulation fp32 vs fp64 accumulation): rel_l2={floor[key][0]:.3e} rel_max={floor[key][1]:.3e}")
            assert l2 < 1.5 * floor[key][0] and mx < 1.5 * floor[key][1], name
        assert floor["out"][0] < TOL_EMU and floor["out"][1] < 2 * TOL_EMU


def test_block_small_gemm_split_k(eng):
    """Text-tower dims at M = 30 x 77 = 2310 rows: c_proj (N = 512, K = 2048) has 40 tiles for 256 CUs, so the
    persistent GEMM K-splits every tile (gemm256.hip launch_t, split_from = 0) and the fix-up kernel sums the parts."""
    tower, sd = _one_layer_tower(eng, 512, 8, 2048, 17, "bf16")
    x = synth.normal([30, 77, 512], 18, "block.split.x")
    got = tower.forward(x.to(DEV))["hidden"].cpu()
    with torch.no_grad():
        emu, _ = clip_ref.block_forward(x, sd, "transformer.resblocks.0.", 8, emulate="bf16")
        ref, _ = clip_ref.block_forward(x, sd, "transformer.resblocks.0.", 8)
    _report("block split-K bf16 vs emulated", got, emu)
    assert rel_l2(got, emu) < TOL_EMU
    assert rel_max(got, ref) < TOL_BF16


@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
@pytest.mark.parametrize("causal", [False, True])
def test_long_sequence_flash_attention(eng, precision, causal):
    """T = 577 (ViT-L/14@336 tokens) > 256 keys: the flash-style attention kernel, one block at d = 1024, H = 16."""
    tower, sd = _one_layer_tower(eng, 1024, 16, 4096, 11, precision)
    x = synth.normal([2, 577, 1024], 12, "flash.x")
    r = tower.forward(x.to(DEV), causal=causal)
    mask = clip_ref.causal_mask(577) if causal else None
    y, _ = clip_ref.block_forward(x, sd, "transformer.resblocks.0.", 16, attn_mask=mask)
    _report(f"flash block T=577 {precision} causal={causal}", r["hidden"], y)
    assert rel_max(r["hidden"].cpu(), y) < (TOL if precision == "bf16x3" else TOL_BF16)
    with pytest.raises(ValueError):
        tower.forward(x.to(DEV), want_mean=True)  # write-back is only built for <= 256 tokens


@pytest.mark.parametrize("precision", ["fp16", "bf16"])
@pytest.mark.parametrize("size", [224, 280, 336, 448])
def test_long_sequence_attention_token_counts(eng, precision, size):
    """The LDS-DMA attention kernel of T > 256 (csrc/attention_long.hip) at token counts that put the ragged ends everywhere:
    patch 14 at 224 / 280 / 336 / 448 pixels = 257 / 401 / 577 / 1 025 tokens (last key block of 1, 17, 1 and 1 keys; last
    query chunk of 1, 10, 5 and 1 query tiles), one block at d = 256 / 4 heads, against the fp32 oracle; twice for bits."""
    d, heads, mlp = 256, 4, 512
    cfg = configs.ClipDims(f"long{size}", 64, size, 14, configs.TowerDims(d, 1, heads, mlp), configs.TowerDims(128, 1, 2, 256), vocab=16, ctx=8)
    ocfg = clip_ref.ClipDims(f"long{size}", 64, size, 14, clip_ref.TowerDims(d, 1, heads, mlp), clip_ref.TowerDims(128, 1, 2, 256), vocab=16, ctx=8)
    tokens = (size // 14) ** 2 + 1
    sd = {}
    synth._tower(sd, "visual.transformer.", d, 1, mlp, seed=21)
    g = torch.Generator().manual_seed(size)
    sd["visual.conv1.weight"] = torch.randn(d, 3, 14, 14, generator=g) * 0.03
    sd["visual.class_embedding"] = torch.randn(d, generator=g) * 0.3
    sd["visual.positional_embedding"] = torch.randn(tokens, d, generator=g) * 0.3
    for k in ("ln_pre", "ln_post"):
        sd[f"visual.{k}.weight"] = 1.0 + 0.1 * torch.randn(d, generator=g)
        sd[f"visual.{k}.bias"] = 0.05 * torch.randn(d, generator=g)
    sd["visual.proj"] = torch.randn(d, 64, generator=g) * d ** -0.5
    # (every row of the one block: with the CLS-only last block the long kernel would not run at all)
    tower = eng.VisionTower(cfg, sd, DEV, precision, prune_last_block=False)
    images = synth.make_images(3, cfg, 31)
    a = tower.encode_image(images.to(DEV)).cpu()
    b = tower.encode_image(images.to(DEV)).cpu()
    with torch.no_grad():
        ref = clip_ref.encode_image(images, sd, ocfg)
    _report(f"long attention {tokens} tokens {precision}", a, ref)
    assert torch.equal(a, b), "same input twice must be bit-identical"
    assert bool(torch.isfinite(a).all())
    assert rel_max(a, ref) < (TOL if precision == "fp16" else TOL_BF16)


@pytest.mark.parametrize("precision", ["fp16", "bf16"])
@pytest.mark.parametrize("T", [530, 576, 577, 641])
def test_long_sequence_attention_wide_scores(eng, precision, T):
    """EVERY row of one block through the T > 256 kernel with scores of sigma 6 (maxima near 20, softmax rows that rescale
    by e^10 from one key block to the next), at token counts whose last query chunk leaves waves with ONE query tile, with
    and without a partial last key block.  This is the shape that showed the round-5 hazard (profiles/r05_flash2_asm_hazard.txt:
    an asm statement read the last key tile's scores before their MFMA had landed -- hipcc does not pad asm operands -- so a
    block maximum missed its largest score: P beyond the half range, NaN rows; far-off rows in bf16).  With that asm statement
    put back, [530-bf16] and [576-bf16] fail (profiles/r05_asm_hazard_regression_tests.log).  "fp16" on a text-kind tower is the
    split-bf16 mode: the first flash kernel of attention.hip (measured 2e-5); "bf16" is the LDS-DMA kernel (measured 9.9e-3: the
    2^-9 rounding of q and k on sums of sigma 6)."""
    tower, sd = _one_layer_tower(eng, 256, 4, 512, 23, precision, q_gain=6.0)
    x = synth.normal([3, T, 256], 40 + T, "wide.x")
    a = tower.forward(x.to(DEV))["hidden"].cpu()
    b = tower.forward(x.to(DEV))["hidden"].cpu()
    with torch.no_grad():
        y, _ = clip_ref.block_forward(x, sd, "transformer.resblocks.0.", 4)
    _report(f"wide-score block T={T} {precision}", a, y)
    assert bool(torch.isfinite(a).all()), "non-finite rows"
    assert torch.equal(a, b), "same input twice must be bit-identical"
    row_err = (a - y).abs().amax(dim=-1) / y.abs().max()
    worst = int(row_err.argmax())
    assert float(row_err.max()) < (TOL if precision == "fp16" else 3e-2), f"row {worst % T} of sequence {worst // T}: {float(row_err.max()):.3e}"


@pytest.mark.parametrize("precision", ["fp16", "bf16"])
@pytest.mark.parametrize("size", [224, 322, 336])
def test_base2_scores_tower_wide_scores(eng, precision, size):
    """The same hazard shape through the path that folds log2(e) into Wq (image towers of more than 256 tokens: base-2 scores,
    AttnArgs::q_log2): two blocks at 257 / 530 / 577 tokens with every row computed, so a bad row of block 1 is a key / value of block 2's
    CLS query; q gains of 6 in both blocks.  Against the fp32 oracle, with the CLS-only last block and without (measured 0.8 - 1.6e-3 in
    IEEE half, 6.0 - 8.5e-3 in bf16; with the asm maximum put back the half build returns NaN: r05_asm_hazard_regression_tests.log)."""
    d, heads, mlp = 256, 4, 512
    cfg = configs.ClipDims("wide336", 64, size, 14, configs.TowerDims(d, 2, heads, mlp), configs.TowerDims(128, 1, 2, 256), vocab=16, ctx=8)
    ocfg = clip_ref.ClipDims("wide336", 64, size, 14, clip_ref.TowerDims(d, 2, heads, mlp), clip_ref.TowerDims(128, 1, 2, 256), vocab=16, ctx=8)
    tokens = (size // 14) ** 2 + 1
    sd = {}
    synth._tower(sd, "visual.transformer.", d, 2, mlp, seed=22)
    for li in range(2):
        sd[f"visual.transformer.resblocks.{li}.attn.in_proj_weight"][:d] *= 6.0
        sd[f"visual.transformer.resblocks.{li}.attn.in_proj_bias"][:d] *= 6.0
    g = torch.Generator().manual_seed(77)
    sd["visual.conv1.weight"] = torch.randn(d, 3, 14, 14, generator=g) * 0.03
    sd["visual.class_embedding"] = torch.randn(d, generator=g) * 0.3
    sd["visual.positional_embedding"] = torch.randn(tokens, d, generator=g) * 0.3
    for k in ("ln_pre", "ln_post"):
        sd[f"visual.{k}.weight"] = 1.0 + 0.1 * torch.randn(d, generator=g)
        sd[f"visual.{k}.bias"] = 0.05 * torch.randn(d, generator=g)
    sd["visual.proj"] = torch.randn(d, 64, generator=g) * d ** -0.5
    images = synth.make_images(4, cfg, 32)
    with torch.no_grad():
        ref = clip_ref.encode_image(images, sd, ocfg)
    tol = 3e-3 if precision == "fp16" else 3e-2  # (the rounding of q and k at this score width, not BASELINE's 1e-3)
    for prune in (False, True):
        emb = eng.VisionTower(cfg, sd, DEV, precision, prune_last_block=prune).encode_image(images.to(DEV)).cpu()
        _report(f"base-2 scores, wide, 2 blocks x {tokens} tokens {precision} prune={prune}", emb, ref)
        assert bool(torch.isfinite(emb).all())
        assert rel_max(emb, ref) < tol


@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_encode_image_vit_l14_336_geometry(eng, precision):
    """BASELINE configs[4] geometry (patch 14 -> K = 588 padded to 640, 577 tokens, width 1024, 16 heads,
    embed 768) with 2 blocks instead of 24 so the CPU oracle stays quick."""
    cfg = configs.ClipDims("L14-336-2blocks", 768, 336, 14, configs.TowerDims(1024, 2, 16, 4096), configs.TowerDims(768, 1, 12, 3072))
    ocfg = clip_ref.ClipDims("L14-336-2blocks", 768, 336, 14, clip_ref.TowerDims(1024, 2, 16, 4096), clip_ref.TowerDims(768, 1, 12, 3072))
    sd = synth.make_state_dict(cfg, seed=5, text=False)
    images = synth.make_images(2, cfg, 6)
    emb = eng.VisionTower(cfg, sd, DEV, precision).encode_image(images.to(DEV), normalize=True)
    with torch.no_grad():
        ref = clip_ref.encode_image(images, sd, ocfg, normalize=True)
    _report(f"encode_image L/14@336 geometry {precision}", emb, ref)
    assert rel_max(emb.cpu(), ref) < (TOL if precision == "bf16x3" else TOL_BF16)


def test_causal_mask(eng):
    tower, sd = _one_layer_tower(eng, 512, 8, 2048, 7, "bf16x3")
    x = synth.normal([3, 77, 512], 9, "causal.x")
    r = tower.forward(x.to(DEV), causal=True, want_heads=True)
    y, p = clip_ref.block_forward(x, sd, "transformer.resblocks.0.", 8, attn_mask=clip_ref.causal_mask(77), want_probs=True)
    assert rel_max(r["hidden"].cpu(), y) < TOL
    assert rel_max(r["attn_heads"].cpu(), p) < TOL
    assert float(r["attn_heads"].cpu().triu(1).abs().max()) == 0.0


# ---- image tower --------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def vitb16():
    cfg = configs.get_config("ViT-B-16")
    return cfg, synth.make_state_dict(cfg, seed=2, text=False)


@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_encode_image_tiny(eng, precision):
    cfg = configs.get_config("tiny")
    sd = synth.make_state_dict(cfg, seed=2)
    images = synth.make_images(5, cfg, 0)
    tower = eng.VisionTower(cfg, sd, DEV, precision)
    emb = tower.encode_image(images.to(DEV))
    ref = clip_ref.encode_image(images, sd, clip_ref.CONFIGS["tiny"])
    _report(f"encode_image tiny {precision}", emb, ref)
    assert rel_max(emb.cpu(), ref) < (TOL if precision == "bf16x3" else TOL_BF16)
    n = tower.encode_image(images.to(DEV), normalize=True).cpu()
    assert torch.allclose(n.norm(dim=-1), torch.ones(5), atol=1e-5)
    assert rel_max(n, ref / ref.norm(dim=-1, keepdim=True)) < (TOL if precision == "bf16x3" else TOL_BF16)


@pytest.mark.parametrize("name", ["ViT-B-16", "ViT-B-32"])
def test_encode_image_real_dims_vs_golden(eng, name):
    g = golden(f"image_tower_{name}")
    cfg = configs.get_config(name)
    sd = synth.make_state_dict(cfg, seed=int(g["seed_weights"]), text=False)
    images = synth.make_images(int(g["batch"]), cfg, int(g["seed_images"]))
    ref = torch.from_numpy(g["embeddings"])
    emb3 = eng.VisionTower(cfg, sd, DEV, "bf16x3").encode_image(images.to(DEV))
    _report(f"encode_image {name} bf16x3 vs fp32 golden", emb3, ref)
    assert rel_max(emb3.cpu(), ref) < TOL and rel_l2(emb3.cpu(), ref) < TOL
    emb1 = eng.VisionTower(cfg, sd, DEV, "bf16").encode_image(images.to(DEV))
    _report(f"encode_image {name} bf16 vs fp32 golden", emb1, ref)
    assert rel_l2(emb1.cpu(), ref) < TOL_BF16
    # (through 12 layers the bf16 round-off flips decorrelate: the bf16-emulating oracle is then no closer
    # to the kernels than the fp32 one -- measured 2.0e-3 vs 2.2e-3 -- so it is only used per block above)


@pytest.mark.parametrize("name,batch", [("ViT-B-32", 1), ("ViT-B-32", 11), ("ViT-B-32", 43), ("ViT-B-16", 3), ("ViT-B-16", 11), ("tiny", 300)])
def test_encode_image_ragged_batches(eng, name, batch):
    """Row counts that are not multiples of any tile (M = batch x tokens: 50, 550, 2150, 591, 2167, 5100) through
    both GEMM kernels (register-staged below 2048 rows, persistent LDS-DMA above), parity mode vs the fp32 oracle."""
    cfg = configs.get_config(name)
    sd = synth.make_state_dict(cfg, seed=2, text=False)
    images = synth.make_images(batch, cfg, 21)
    emb = eng.VisionTower(cfg, sd, DEV, "bf16x3").encode_image(images.to(DEV)).cpu()
    probe = [0, batch // 2, batch - 1] if batch > 3 else list(range(batch))
    with torch.no_grad():
        ref = clip_ref.encode_image(images[probe], sd, clip_ref.CONFIGS[name])
    assert rel_max(emb[probe], ref) < TOL
    assert torch.isfinite(emb).all()


def test_empty_batch(eng):
    from tap_clip_amd.models import CLIPWrapper, FullModel

    cfg = configs.get_config("tiny")
    sd = synth.make_state_dict(cfg, seed=2)
    clip = CLIPWrapper("tiny", None, DEV, state_dict=sd)
    assert clip.encode_image(torch.zeros(0, 3, 32, 32, device=DEV)).shape == (0, 64)
    model = FullModel(["Mug", "Pen"], clip, prompt_len=5).eval()
    with torch.no_grad():
        assert model(torch.zeros(0, 3, 32, 32, device=DEV))["logits"].shape == (0, 2)


def test_encode_image_full_batch_properties(eng, vitb16):
    """BASELINE.json configs[1] size (batch 256): size-independent properties -- unit norms,
    run-to-run determinism, and batch invariance (row i does not depend on its batch mates)."""
    cfg, sd = vitb16
    tower = eng.VisionTower(cfg, sd, DEV, "bf16")
    images = synth.make_images(256, cfg, 0).to(DEV)
    a = tower.encode_image(images, normalize=True)
    b = tower.encode_image(images, normalize=True)
    assert torch.equal(a, b), "same input twice must be bit-identical"
    assert torch.isfinite(a).all()
    assert torch.allclose(a.norm(dim=-1), torch.ones(256, device=DEV), atol=1e-5)
    small = tower.encode_image(images[:8].clone(), normalize=True)
    assert torch.equal(small, a[:8]), "embedding of an image must not depend on the rest of the batch"
    with torch.no_grad():
        ref = clip_ref.encode_image(images[:2].cpu(), sd, clip_ref.CONFIGS["ViT-B-16"], normalize=True)
    own = rel_l2(a[:2].cpu(), ref)  # this mode's rounding error against the fp32 oracle (2.4e-3)
    assert own < TOL_BF16
    # the last row tiles of the N = 768, K = 3072 GEMM are K-split over the CUs a partial round would idle
    # (gemm256.hip "tail split"): their fp32 sums associate differently, so those rows match a small batch
    # to the mode's round-off instead of bitwise (TOL_TAIL_SPLIT above; README "known properties of the bf16 mode")
    last = tower.encode_image(images[-8:].clone(), normalize=True)
    assert rel_l2(last.cpu(), a[-8:].cpu()) < TOL_TAIL_SPLIT


# ---- text side ----------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_encode_text_tiny(eng, precision):
    from tap_clip_amd.models import CLIPWrapper

    cfg = configs.get_config("tiny")
    sd = synth.make_state_dict(cfg, seed=2)
    clip = CLIPWrapper("tiny", None, DEV, precision=precision, state_dict=sd)
    tokens = torch.zeros(4, cfg.ctx, dtype=torch.long)
    for i in range(4):
        L = 3 + 2 * i
        tokens[i, 0] = cfg.vocab - 2
        tokens[i, 1:1 + L] = synth.integers([L], 3, f"t.{i}", cfg.vocab - 3) + 1
        tokens[i, 1 + L] = cfg.vocab - 1
    out = clip.encode_text(tokens.to(DEV))
    ref = clip_ref.encode_text(tokens, sd, clip_ref.CONFIGS["tiny"])
    _report(f"encode_text tiny {precision}", out, ref)
    assert rel_max(out.cpu(), ref) < (TOL if precision == "bf16x3" else TOL_BF16)


def test_small_ops_vs_reference_goldens(eng):
    g = golden("attribution_monitor")
    a = torch.from_numpy(g["attn_map"]).to(DEV)
    assert rel_max(eng.attribution(a, 16).cpu(), torch.from_numpy(g["out_p16"])) < 1e-6
    assert torch.equal(eng.attribution(a, 16, normalize=False).cpu(), torch.from_numpy(g["out_p16_raw"]))
    assert rel_max(eng.attribution(a, 5).cpu(), torch.from_numpy(g["out_p5"])) < 1e-6
    lit = eng.attribution(torch.from_numpy(g["literal_in"]).to(DEV), 5).cpu()
    assert torch.equal(lit, torch.from_numpy(g["literal_out_p5"]))
    g = golden("prompt_adjustor")
    p, at = torch.from_numpy(g["prompt"]).to(DEV), torch.from_numpy(g["attribution"]).to(DEV)
    tok = torch.zeros(3, 2, 512, device=DEV)
    out = eng.build_prompts(p, tok, at).cpu()
    assert torch.equal(out[:, :16], torch.from_numpy(g["out"])) and float(out[:, 16:].abs().max()) == 0.0
    out1 = eng.build_prompts(p, tok, torch.from_numpy(g["attribution_b1"]).to(DEV)).cpu()
    assert torch.equal(out1[:, :16], torch.from_numpy(g["out_b1"]))
    img = torch.nn.functional.normalize(synth.normal([9, 512], 1, "lg.i"), dim=-1)
    txt = torch.nn.functional.normalize(synth.normal([65, 512], 1, "lg.t"), dim=-1)
    lg = eng.logits(img.to(DEV), txt.to(DEV), 14.2857).cpu()
    assert rel_max(lg, 14.2857 * img @ txt.t()) < 1e-5


@pytest.mark.parametrize("method", ["gate", "residual"])
def test_prompt_adjustor_mlp_kernel_vs_reference(eng, method):
    """`tapclip_build_prompts_mlp` against the outputs of the reference's own PromptAdjustor('gate' | 'residual') (goldens carry
    the parameters its module drew): adjusted context rows, token rows copied behind them, a broadcast [n, 1] attribution too;
    and FullModel's no-grad forward takes the kernel while its training forward keeps the differentiable modules -- same logits."""
    from tap_clip_amd.models.prompt_adjustor import PromptAdjustor

    g = golden("prompt_adjustor_mlp")
    m = PromptAdjustor(method).to(DEV)
    net = m.gate_net if method == "gate" else m.residual_net
    with torch.no_grad():
        net[0].weight.copy_(torch.from_numpy(g[f"{method}_w1"])); net[0].bias.copy_(torch.from_numpy(g[f"{method}_b1"]))
        net[2].weight.copy_(torch.from_numpy(g[f"{method}_w2"])); net[2].bias.copy_(torch.from_numpy(g[f"{method}_b2"]))
    p, at = torch.from_numpy(g["prompt"]).to(DEV), torch.from_numpy(g["attribution"]).to(DEV)
    tok = synth.normal([3, 7, 512], 3, "adj.tok").to(DEV)
    out = eng.build_prompts_mlp(p, tok, at, m).cpu()
    assert rel_max(out[:, :16], torch.from_numpy(g[f"{method}_out"])) < 2e-6 and torch.equal(out[:, 16:], tok.cpu())
    out1 = eng.build_prompts_mlp(p, tok, torch.ones(3, 1, device=DEV), m).cpu()
    assert rel_max(out1[:, :16], torch.from_numpy(g[f"{method}_out_b1"])) < 2e-6
    if method == "gate":  # (the reference hard-codes 512 output columns for 'residual': only D = 512 models can run it; tiny has 128)
        gt = golden("fullmodel_intended_tiny")
        model, images = _build_full("tiny", gt, "intended", "bf16x3")
        model.prompt_adjustor = PromptAdjustor("gate").to(DEV)
        with torch.no_grad():
            a = model(images)["logits"]                       # kernel path
        b = model.train()(images)["logits"].detach()           # torch modules (autograd path)
        assert rel_max(a.cpu(), b.cpu()) < 1e-4


# ---- FullModel vs the reference's own FullModel (goldens) ----------------------------------------
def _build_full(cfg_name, g, semantics, precision, collapse=True):
    from tap_clip_amd.models import CLIPWrapper, FullModel

    cfg = configs.get_config(cfg_name)
    sd = synth.make_state_dict(cfg, seed=int(g["seed_weights"]))
    clip = CLIPWrapper(cfg_name, None, DEV, precision=precision, attn_semantics=semantics, state_dict=sd)
    names = g["class_names"].tolist()
    table = {f"a photo of a {c}": torch.from_numpy(g["token_ids"][i:i + 1]) for i, c in enumerate(names)}
    clip.tokenizer = lambda text: table[text].clone()
    model = FullModel(names, clip, prompt_len=int(g["prompt_len"]), adjustor_method="scale", class_specific=True,
                      collapse_text=collapse)
    # the context tokens: committed for the small cases, a seed of the build's own generator for the large ones
    ctx = (torch.from_numpy(g["context"]) if "context" in g.files else
           synth.make_prompts(len(names), int(g["prompt_len"]), cfg, seed=int(g["seed_context"]))[0])
    with torch.no_grad():
        for i, c in enumerate(names):
            model.prompt_learner.context_bank[c].copy_(ctx[i])
    images = synth.make_images(int(g["batch"]), cfg, int(g["seed_images"]))
    return model.eval(), images.to(DEV)


@pytest.mark.parametrize("semantics", ["literal", "intended"])
@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_fullmodel_tiny_vs_reference(semantics, precision):
    g = golden(f"fullmodel_{semantics}_tiny")
    model, images = _build_full("tiny", g, semantics, precision)
    with torch.no_grad():
        out = model(images, torch.from_numpy(g["labels"]).to(DEV))
    ref = torch.from_numpy(g["logits"])
    _report(f"FullModel tiny {semantics} {precision} logits", out["logits"], ref)
    fl = _bf16_floor(g, "tiny", semantics) if precision == "bf16" else None
    tol = TOL if fl is None else 1.5 * fl["logits"]
    assert rel_max(out["logits"].cpu(), ref) < tol
    assert abs(float(out["loss"]) - float(g["loss"])) < (TOL * max(1.0, abs(float(g["loss"]))) if fl is None else 1.5 * fl["loss"])
    assert torch.equal(model.prompt_learner().cpu(), torch.from_numpy(g["prompts"]))
    if semantics == "intended":
        assert rel_max(model.last_attribution.cpu(), torch.from_numpy(g["attribution"])) < (TOL if fl is None else max(1.5 * fl["attribution"], TOL))
        assert rel_max(model.clip.attention_maps[0].cpu(), torch.from_numpy(g["attn_map"])) < (TOL if fl is None else 1.5 * fl["map"])
    keys = set(model.state_dict().keys())
    assert set(g["state_dict_keys"].tolist()) <= keys, sorted(set(g["state_dict_keys"].tolist()) - keys)[:5]


def _bf16_floor(g, cfg_name, semantics):
    """What bf16 OPERANDS cost FullModel's outputs on a golden case, measured instead of chosen ("a bound the builder
    chose"): the CPU oracle with every MFMA operand rounded to bf16 at the kernels' rounding points (oracle/clip_ref.py `emulate`)
    against the reference's own outputs.  The bf16 mode of the HIP towers is held to 1.5x these figures -- the two differ by
    accumulation order and by which way ties of the chained roundings fall (clip_ref.emulation_floor)."""
    from oracle import full_model_ref

    cfg = clip_ref.CONFIGS[cfg_name]
    hcfg = configs.get_config(cfg_name)
    sd = synth.make_state_dict(hcfg, seed=int(g["seed_weights"]))
    P, n = int(g["prompt_len"]), len(g["class_names"])
    ctx = torch.from_numpy(g["context"]) if "context" in g.files else synth.make_prompts(n, P, hcfg, seed=int(g["seed_context"]))[0]
    tok = sd["token_embedding.weight"][torch.from_numpy(g["token_ids"])]
    images = synth.make_images(int(g["batch"]), hcfg, int(g["seed_images"]))
    with torch.no_grad():
        out = full_model_ref.forward_collapsed(images, torch.cat([ctx, tok], 1), P, sd, cfg, attn_semantics=semantics, emulate="bf16",
                                               labels=torch.from_numpy(g["labels"]))
    # (the loss: cross-entropy moves by at most twice the largest logit error -- bounded through the logits, not through the
    # emulation's own loss error, which is a signed sum that may cancel)
    floor = {"logits": rel_max(out["logits"], torch.from_numpy(g["logits"]))}
    floor["loss"] = 2.0 * floor["logits"] * float(torch.from_numpy(g["logits"]).abs().max())
    if semantics == "intended":
        amap = out["attn_map"]
        if "attn_map" in g.files:
            floor["map"] = rel_max(amap, torch.from_numpy(g["attn_map"]))
        elif "attn_map_head" in g.files:
            floor["map"] = max(rel_max(amap[:8], torch.from_numpy(g["attn_map_head"])), rel_max(amap[:, :, -1], torch.from_numpy(g["attn_map_last_col"])))
        if "attribution" in g.files:
            floor["attribution"] = rel_max(out["attribution"], torch.from_numpy(g["attribution"]))
    print("[parity] bf16 floor (emulating oracle vs the reference): " + ", ".join(f"{k} {v:.3e}" for k, v in floor.items()))
    return floor


def _oracle_context_grad(g, cfg_name, semantics, emulate):
    """d loss / d context of FullModel's collapsed forward through the CPU oracle (torch autograd), with the operand rounding
    of `emulate` at the kernels' rounding points; the attribution is a constant (reference clip_wrapper.py:36 detaches it)."""
    import math
    from oracle import full_model_ref

    cfg = clip_ref.CONFIGS[cfg_name]
    sd = synth.make_state_dict(configs.get_config(cfg_name), seed=int(g["seed_weights"]))
    P = int(g["prompt_len"])
    ctx = torch.from_numpy(g["context"]).clone().requires_grad_(True)
    tok = sd["token_embedding.weight"][torch.from_numpy(g["token_ids"])]
    images = synth.make_images(int(g["batch"]), configs.get_config(cfg_name), int(g["seed_images"]))
    with torch.no_grad():
        _, aux = full_model_ref.text_features(torch.cat([ctx.detach(), tok], 1), P, sd, cfg, semantics, emulate, return_aux=True)
        img = clip_ref.encode_image(images, sd, cfg, emulate, normalize=True)
    adjusted = torch.cat([full_model_ref.adjust_scale(ctx, aux["attribution"]), tok], dim=1)
    hidden, _, _ = clip_ref.text_transformer_raw(adjusted, sd, cfg, emulate)
    feat = hidden[:, -1, :] @ sd["text_projection"]
    feat = feat / feat.norm(dim=-1, keepdim=True)
    logits = math.exp(math.log(1 / 0.07)) * img @ feat.t()
    torch.nn.functional.cross_entropy(logits, torch.from_numpy(g["labels"])).backward()
    return ctx.grad.detach()


@pytest.mark.parametrize("semantics", ["literal", "intended"])
@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_fullmodel_backward_vs_reference(semantics, precision):
    """loss.backward() of the reference FullModel (train.py:99-105): gradients of every context_bank entry
    and of logit_scale, from the goldens."""
    g = golden(f"fullmodel_{semantics}_tiny")
    model, images = _build_full("tiny", g, semantics, precision)
    model.train()
    out = model(images, torch.from_numpy(g["labels"]).to(DEV))
    assert out["logits"].requires_grad
    out["loss"].backward()
    names = g["class_names"].tolist()
    grad = torch.stack([model.prompt_learner.context_bank[c].grad for c in names], 0).cpu()
    ref = torch.from_numpy(g["context_grad"])
    _report(f"FullModel tiny {semantics} {precision} context grad", grad, ref)
    if precision == "bf16x3":
        tol = TOL
    else:
        # what bf16 operands cost these gradients, measured: torch autograd through the oracle with bf16 rounding at the
        # kernels' rounding points (attribution a constant, as the reference's hook detaches it) against the reference's
        # own gradients; the HIP backward rounds its gradient operands too and is held to 2x that (it was a chosen 5e-2)
        e_grad = _oracle_context_grad(g, "tiny", semantics, "bf16")
        assert rel_max(_oracle_context_grad(g, "tiny", semantics, None), ref) < 1e-4
        tol = 2 * max(rel_max(e_grad, ref), rel_l2(e_grad, ref))
        print(f"[parity] bf16 context-gradient floor x2 = {tol:.3e}; HIP {rel_max(grad, ref):.3e} / {rel_l2(grad, ref):.3e}")
    assert rel_max(grad, ref) < tol and rel_l2(grad, ref) < tol
    tol = TOL if precision == "bf16x3" else TOL_BF16
    assert abs(float(model.logit_scale.grad) - float(g["logit_scale_grad"])) < tol * max(1.0, abs(float(g["logit_scale_grad"])))
    assert abs(float(out["loss"]) - float(g["loss"])) < tol * max(1.0, abs(float(g["loss"])))
    assert all(p.grad is None for p in model.clip.parameters())  # CLIP stays frozen


@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_text_backward_real_dims_vs_oracle_autograd(eng, precision):
    """dX through the ViT-B text tower (d=512, 12 blocks, T = 16 + 77 = 93) against torch autograd through
    the CPU oracle, for a gradient that enters at token -1 like FullModel's."""
    cfg = configs.get_config("ViT-B-16")
    sd = synth.make_state_dict(cfg, seed=2, vision=False)
    tower = eng.TextTower(cfg, sd, DEV, precision)
    n, T, D = 6, 93, 512
    x = torch.cat([synth.normal([n, 16, D], 4, "bwd.ctx"), synth.normal([n, 77, D], 4, "bwd.tok", 0.02)], dim=1)
    gfeat = synth.normal([n, 512], 4, "bwd.g")
    xr = x.clone().requires_grad_(True)
    hidden, _, _ = clip_ref.text_transformer_raw(xr, sd, clip_ref.CONFIGS["ViT-B-16"])
    feat = hidden[:, -1, :] @ sd["text_projection"]
    feat = feat / feat.norm(dim=-1, keepdim=True)
    (feat * gfeat).sum().backward()
    hid = tower.forward(x.to(DEV))["hidden"]
    g_hidden = tower.pool_project_backward(hid, gfeat.to(DEV), normalize=True)
    gx = tower.backward(x.to(DEV), g_hidden).cpu()
    _report(f"text backward real dims {precision} dL/dx", gx, xr.grad)
    if precision == "bf16x3":
        assert rel_l2(gx, xr.grad) < TOL and rel_max(gx[:, :16], xr.grad[:, :16]) < TOL
        return
    # bf16: what the FORMAT costs a gradient is measured, not chosen -- torch autograd through the oracle with the operands
    # rounded to bf16 at the kernels' rounding points (the casts pass gradients straight through) against the fp32 gradient;
    # the HIP backward also rounds the gradient operands of its own GEMMs, so it is held to 2x that figure (VERDICT r03 item 8:
    # the bound used to be a chosen 5e-2 against a measured 7e-3 .. 1.1e-2)
    xe = x.clone().requires_grad_(True)
    hidden_e, _, _ = clip_ref.text_transformer_raw(xe, sd, clip_ref.CONFIGS["ViT-B-16"], emulate="bf16")
    feat_e = hidden_e[:, -1, :] @ sd["text_projection"]
    feat_e = feat_e / feat_e.norm(dim=-1, keepdim=True)
    (feat_e * gfeat).sum().backward()
    floor_l2, floor_max = rel_l2(xe.grad, xr.grad), rel_max(xe.grad[:, :16], xr.grad[:, :16])
    print(f"[parity] bf16 gradient floor (emulated forward, exact backward): rel_l2 {floor_l2:.3e} rel_max(ctx rows) {floor_max:.3e}; "
          f"HIP: {rel_l2(gx, xr.grad):.3e} / {rel_max(gx[:, :16], xr.grad[:, :16]):.3e}")
    assert rel_l2(gx, xr.grad) < 2 * floor_l2
    assert rel_max(gx[:, :16], xr.grad[:, :16]) < 2 * floor_max  # the context-token rows FullModel uses


@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
@pytest.mark.parametrize("T", [5, 20, 40, 49, 70, 96])
def test_text_backward_every_sequence_length_class_vs_oracle_autograd(eng, precision, T):
    """The attention backward is instantiated per number of 16-key tiles (1..6): one sequence length in each class,
    including the odd ones (T = 5, 40, 70: the 16-bit variant's 32-deep MFMA steps run past the padded length there and
    must read zeros), through the 2-block tiny text tower against torch autograd through the CPU oracle.  Reference:
    the dX path of train.py:99-105 through models/model_wrapper.py:72-75."""
    cfg = configs.get_config("tiny")
    sd = synth.make_state_dict(cfg, seed=5, vision=False)
    tower = eng.TextTower(cfg, sd, DEV, precision)
    n, D = 3, cfg.text.width
    x = synth.normal([n, T, D], 6, f"bwdT.x{T}")
    g = synth.normal([n, T, D], 6, f"bwdT.g{T}")
    xr = x.clone().requires_grad_(True)
    hidden, _, _ = clip_ref.text_transformer_raw(xr, sd, clip_ref.CONFIGS["tiny"])
    (hidden * g).sum().backward()
    gx = tower.backward(x.to(DEV), g.to(DEV)).cpu()
    _report(f"text backward tiny T={T} {precision} dL/dx", gx, xr.grad)
    assert torch.isfinite(gx).all()
    assert rel_l2(gx, xr.grad) < (TOL if precision == "bf16x3" else 1e-2)  # measured 2.2e-3 .. 3.9e-3 (bf16), 4e-6 .. 6e-6 (bf16x3)


@pytest.mark.parametrize("quick", [False, True])
@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_text_tower_25_classes_all_tiles_k_split(eng, precision, quick):
    """25 classes x 93 tokens = 2 325 rows of the ViT-B text tower: 10 row tiles, so every GEMM of a block has fewer tiles
    than half the CUs.  In split-bf16 all four are then K-split over the idle CUs (48 / 192 k-steps in parts of >= 16): QKV,
    out_proj, c_fc -- the fix-up kernel applies bias + exact-erf GELU and writes hi + lo planes -- and c_proj; in bf16 only
    c_proj is long enough (64 steps).  Hidden states against the CPU oracle of `clip.model.transformer(x)` (reference
    models/model_wrapper.py:58,72)."""
    cfg = configs.get_config("ViT-B-16-quickgelu" if quick else "ViT-B-16")  # (QuickGELU: the fix-up kernel's activation variant)
    sd = synth.make_state_dict(cfg, seed=2, vision=False)
    tower = eng.TextTower(cfg, sd, DEV, precision)
    n, T, D = 25, 93, 512
    x = torch.cat([synth.normal([n, 16, D], 8, "ks.ctx"), synth.normal([n, 77, D], 8, "ks.tok", 0.02)], dim=1)
    import dataclasses
    with torch.no_grad():
        ref, _, _ = clip_ref.text_transformer_raw(x, sd, dataclasses.replace(clip_ref.CONFIGS["ViT-B-16"], quick_gelu=quick))
    r = tower.forward(x.to(DEV), want_mean=True)
    _report(f"text tower 25 classes {precision} hidden", r["hidden"], ref)
    assert torch.isfinite(r["hidden"]).all()
    assert rel_l2(r["hidden"].cpu(), ref) < (TOL if precision == "bf16x3" else TOL_BF16)
    again = tower.forward(x.to(DEV), want_mean=True)
    assert torch.equal(again["hidden"], r["hidden"])  # (fixed-order sums of the parts: run-to-run bit-identical)
    # a row's bits may depend on the launch's row count, never on which rows share the launch: same count, other mates
    x2 = x.clone()
    x2[1:] = x.flip(0)[:-1]
    assert torch.equal(tower.forward(x2.to(DEV))["hidden"][0], r["hidden"][0])


def test_text_backward_saved_equals_recompute(eng):
    """`forward_saved` + `backward_saved` (what a training step uses) against `forward` + the recomputing `backward`."""
    cfg = configs.get_config("tiny")
    sd = synth.make_state_dict(cfg, seed=2)
    tower = eng.TextTower(cfg, sd, DEV, "bf16")
    x = synth.normal([7, 21, cfg.text.width], 31, "saved.x").to(DEV)
    g = synth.normal([7, 21, cfg.text.width], 32, "saved.g").to(DEV)
    hidden, saved = tower.forward_saved(x)
    assert rel_max(hidden.cpu(), tower.forward(x)["hidden"].cpu()) < 1e-6
    assert torch.equal(tower.backward_saved(saved, g), tower.backward(x, g))


def test_train_step_reduces_loss():
    """A few AdamW steps on context_bank only (reference train.py:65-67,99-105) lower the loss."""
    g = golden("fullmodel_intended_tiny")
    model, images = _build_full("tiny", g, "intended", "bf16")
    labels = torch.from_numpy(g["labels"]).to(DEV)
    opt = torch.optim.AdamW(model.prompt_learner.parameters(), lr=2e-3, weight_decay=0.01)
    model.train()
    losses = []
    for _ in range(8):
        out = model(images, labels)
        opt.zero_grad()
        out["loss"].backward()
        opt.step()
        losses.append(float(out["loss"]))
    assert losses[-1] < losses[0], losses


@pytest.mark.parametrize("semantics", ["literal", "intended"])
def test_fullmodel_literal_loop_equals_collapsed(semantics):
    g = golden(f"fullmodel_{semantics}_tiny")
    model, images = _build_full("tiny", g, semantics, "bf16x3", collapse=False)
    with torch.no_grad():
        lit = model(images)["logits"]
    assert rel_max(lit.cpu(), torch.from_numpy(g["logits"])) < TOL


@pytest.mark.parametrize("semantics", ["literal", "intended"])
def test_fullmodel_vitb32_cfg1_vs_reference(semantics):
    """BASELINE.json configs[0]: ViT-B/32, batch 8, 10 classes, P=5 -- the reference FullModel's own logits."""
    g = golden(f"fullmodel_{semantics}_vitb32")
    ref = torch.from_numpy(g["logits"])
    # fp16 = IEEE-half image tower (2.8e-4 on the embeddings) + split-bf16 text tower: the fast mode inside the 1e-3
    # bound, on logits too (round 1's IEEE-half text tower left these small-magnitude logits at 1.3e-3 rel-max)
    fl = _bf16_floor(g, "ViT-B-32", semantics)
    for precision, tol in (("bf16x3", TOL), ("bf16", 1.5 * fl["logits"]), ("fp16", TOL)):
        model, images = _build_full("ViT-B-32", g, semantics, precision)
        with torch.no_grad():
            out = model(images, torch.from_numpy(g["labels"]).to(DEV))
        _report(f"FullModel ViT-B/32 cfg1 {semantics} {precision} logits", out["logits"], ref)
        assert rel_max(out["logits"].cpu(), ref) < tol
        assert abs(float(out["loss"]) - float(g["loss"])) < (1.5 * fl["loss"] if precision == "bf16" else tol * max(1.0, abs(float(g["loss"]))))
        assert torch.equal(out["logits"].argmax(1).cpu(), ref.argmax(1)) or precision == "bf16"
        del model
        torch.cuda.empty_cache()


# ---- checkpoint round trip and the evaluation contract -------------
def test_checkpoint_roundtrip_repacks_towers():
    """`torch.save(model.state_dict())` (train.py:131-132) -> `load_state_dict(..., strict=False)` into a model
    built from OTHER weights (test_cross_domain.py:43-61, incl. the legacy `context_emb` conversion) gives the
    saved model's logits: the `clip.model.*` tensors must reach the packed HIP weights, not only the torch copies."""
    from tap_clip_amd.models import CLIPWrapper, FullModel

    cfg = configs.get_config("tiny")
    names = ["Backpack", "Laptop", "Mug"]
    images = synth.make_images(4, cfg, 0).to(DEV)

    def build(seed):
        torch.manual_seed(seed)
        clip = CLIPWrapper("tiny", None, DEV, precision="bf16x3", state_dict=synth.make_state_dict(cfg, seed=seed))
        return FullModel(names, clip, prompt_len=5, class_specific=True).eval()

    a, b = build(2), build(3)
    with torch.no_grad():
        ref = a(images)["logits"]
        assert rel_max(b(images)["logits"].cpu(), ref.cpu()) > 1e-2  # different weights, different answer
    saved = {k: v.detach().cpu().clone() for k, v in a.state_dict().items()}
    # the script's legacy path: context stored as one [n_cls, P, D] tensor
    legacy = {k: v for k, v in saved.items() if "prompt_learner" not in k}
    legacy["prompt_learner.context_emb"] = torch.stack([saved[f"prompt_learner.context_bank.{c}"] for c in names])
    converted = {f"prompt_learner.context_bank.{c}": legacy["prompt_learner.context_emb"][i] for i, c in enumerate(names)}
    converted.update({k: v for k, v in legacy.items() if "prompt_learner" not in k})
    missing, unexpected = b.load_state_dict(converted, strict=False)
    assert not unexpected
    with torch.no_grad():
        assert rel_max(b(images)["logits"].cpu(), ref.cpu()) < 1e-5
    b.prompt_learner.add_class_prompt("Bike")  # unseen class at test time (test_cross_domain.py:65-67)
    with torch.no_grad():
        assert b(images)["logits"].shape == (4, 4)


def test_eval_metrics_contract():
    from tap_clip_amd.utils import eval_metrics

    class Fixed(torch.nn.Module):
        def forward(self, images, labels=None):
            return {"logits": torch.nn.functional.one_hot(images[:, 0, 0, 0].long(), 3).float()}

    x = torch.zeros(6, 1, 1, 1)
    x[:, 0, 0, 0] = torch.tensor([0, 1, 2, 2, 1, 0])
    y = torch.tensor([0, 1, 2, 0, 1, 1])
    loader = [(x[:3], y[:3]), (x[3:], y[3:])]
    assert abs(eval_metrics.evaluate_accuracy(Fixed(), loader, DEV) - 100.0 * 4 / 6) < 1e-9
    per = eval_metrics.evaluate_per_class_accuracy(Fixed(), loader, DEV, ["a", "b", "c"])
    assert per == {"a": 50.0, "b": pytest.approx(100.0 * 2 / 3), "c": 100.0}
    p = torch.tensor([[0.5, 0.5], [1.0, 0.0]])
    assert eval_metrics.attribution_entropy(p) == pytest.approx(0.5 * math.log(2), abs=1e-6)


# ---- error behaviour ----------------------------------------------------------------------------
def test_errors(eng):
    cfg = configs.get_config("tiny")
    sd = synth.make_state_dict(cfg, seed=2)
    tower = eng.VisionTower(cfg, sd, DEV, "bf16")
    with pytest.raises(ValueError):
        tower.encode_image(torch.zeros(2, 3, 16, 16, device=DEV))
    bad = dict(sd)
    del bad["visual.ln_post.bias"]
    with pytest.raises(RuntimeError, match="missing"):
        eng.VisionTower(cfg, bad, DEV, "bf16")
    bad = dict(sd)
    bad["visual.proj"] = torch.zeros(3, 3)
    with pytest.raises(ValueError, match="size mismatch"):
        eng.VisionTower(cfg, bad, DEV, "bf16")
    with pytest.raises(RuntimeError, match="no CPU path"):
        eng.VisionTower(cfg, sd, "cpu", "bf16")
    # a token id outside the embedding table is an error, as in torch (not a silent clamp)
    text = eng.TextTower(cfg, sd, DEV, "bf16")
    ok = torch.zeros(2, 7, dtype=torch.long)
    assert text.embed_tokens(ok, add_pos=False).shape == (2, 7, cfg.text.width)
    for bad_id in (cfg.vocab, -1):
        ids = ok.clone()
        ids[1, 3] = bad_id
        with pytest.raises(ValueError, match="token id"):
            text.embed_tokens(ids, add_pos=False)
    assert text.embed_tokens(ok, add_pos=True).shape == (2, 7, cfg.text.width)  # the flag was cleared


# ---- N > 1: two ranks (sharing the one GPU of the test box, gloo transport) ----------------------
_RANK_SCRIPT = r"""
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
    return FullModel(names, clip, prompt_len=5, class_specific=True, gather_images=gather).eval()
images = synth.make_images(8, cfg, 0)
lo, hi = shard_rows(8, rank, world)
with torch.no_grad():
    sharded = build(True)(images[lo:hi].cuda())["logits"].cpu()      # every rank: GLOBAL logits
    full = build(False)(images.cuda())["logits"].cpu()               # single-process answer
assert sharded.shape == (8, 3), sharded.shape
err = float((sharded - full).abs().max() / full.abs().max())
assert err < 1e-5, err
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok", err)
"""


def test_two_ranks_sharded_forward_equals_single(tmp_path):
    import os, subprocess, sys
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    port = 29600 + os.getpid() % 300
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.count("ok") == 2, r.stdout[-2000:] + r.stderr[-2000:]


# ---- precision "fp16": the same kernels compiled with IEEE-half operands (libtapclip_fp16.so) -----------------
def test_fp16_encode_image_meets_1e3(eng):
    """ViT-B/16 image tower with IEEE-half MFMA operands (same kernels, same speed as bf16): 11 significand bits
    instead of 8 bring the fast path inside the 1e-3 bound of BASELINE.json on the embeddings."""
    g = golden("image_tower_ViT-B-16")
    cfg = configs.get_config("ViT-B-16")
    sd = synth.make_state_dict(cfg, seed=int(g["seed_weights"]), text=False)
    images = synth.make_images(int(g["batch"]), cfg, int(g["seed_images"]))
    ref = torch.from_numpy(g["embeddings"])
    tower = eng.VisionTower(cfg, sd, DEV, "fp16")
    emb = tower.encode_image(images.to(DEV))
    _report("encode_image ViT-B-16 fp16 vs fp32 golden", emb, ref)
    assert rel_l2(emb.cpu(), ref) < TOL and rel_max(emb.cpu(), ref) < TOL
    with torch.no_grad():
        emu = clip_ref.encode_image(images[:2], sd, clip_ref.CONFIGS["ViT-B-16"], emulate="fp16")
    assert rel_l2(emb[:2].cpu(), emu) < TOL
    assert torch.equal(emb, tower.encode_image(images.to(DEV)))


def test_text_tower_precision_beside_the_image_tower_modes(eng):
    """fp16 / fp8 are image-tower precisions: the differentiated text tower runs split-bf16 beside the IEEE-half image
    tower (the 1e-3 mode) and plain bf16 beside the fp8 one (the throughput mode)."""
    cfg = configs.get_config("tiny")
    sd = synth.make_state_dict(cfg, seed=2)
    assert eng.TextTower(cfg, sd, DEV, "fp16").precision == "bf16x3"
    assert eng.TextTower(cfg, sd, DEV, "fp8").precision == "bf16"


def test_text_feature_cache_is_opt_in_and_invalidates():
    """`FullModel(cache_text_features=True)`: identical logits, and the cache follows parameter updates."""
    from tap_clip_amd.models import CLIPWrapper, FullModel
    cfg = configs.get_config("tiny")
    sd = synth.make_state_dict(cfg, seed=2)
    torch.manual_seed(5)
    clip = CLIPWrapper("tiny", None, DEV, state_dict=sd)
    model = FullModel(["Mug", "Pen", "Bag"], clip, prompt_len=5, class_specific=True, cache_text_features=True).eval()
    assert FullModel(["Mug"], clip, prompt_len=5).cache_text_features is False
    images = synth.make_images(4, cfg, 3).to(DEV)
    with torch.no_grad():
        a = model(images)["logits"]
        assert model._text_cache is not None
        b = model(images)["logits"]      # served from the cache
        assert torch.equal(a, b)
        model.cache_text_features = False
        assert torch.equal(model(images)["logits"], a)
        model.cache_text_features = True
        # like an optimiser step: bumps the version counter.  (Not a constant shift or a rescaling of the tokens: the
        # pre-LN blocks are invariant to both as far as the pooled last token is concerned.)
        ctx_p = model.prompt_learner.context_bank["Mug"]
        ctx_p.add_(synth.normal(list(ctx_p.shape), 77, "cache.noise").to(DEV))
        c = model(images)["logits"]
        assert not torch.equal(c, a)
        model.cache_text_features = False
        assert torch.equal(model(images)["logits"], c)


def test_build_prompts_backward_vs_autograd(eng):
    """tapclip_build_prompts_backward against torch autograd through the reference's two ops (prompt_adjustor.py:35-36 multiply,
    model_wrapper.py:69 torch.cat): bit-equal (one fp32 multiply per element either way); [n,P] and literal [n,1] attribution."""
    n, P, L, D = 5, 16, 77, 512
    ctx = synth.normal([n, P, D], 41, "bp.ctx").to(DEV).requires_grad_(True)
    tok = synth.normal([n, L, D], 41, "bp.tok", 0.02).to(DEV)
    g = synth.normal([n, P + L, D], 41, "bp.g").to(DEV)
    for cols in (P, 1):
        a = torch.softmax(synth.normal([n, cols], 41, f"bp.a{cols}"), dim=-1).to(DEV) if cols > 1 else torch.full((n, 1), 0.75, device=DEV)
        ref_out = torch.cat([ctx * a.unsqueeze(-1) if cols > 1 else ctx * a.view(n, 1, 1), tok], dim=1)
        (ref_grad,) = torch.autograd.grad(ref_out, ctx, g)
        got = eng.build_prompts_backward(g, P, a)
        assert torch.equal(got, ref_grad)
        assert torch.equal(eng.build_prompts(ctx, tok, a), ref_out.detach())
    assert torch.equal(eng.build_prompts_backward(g, P, None), g[:, :P])
