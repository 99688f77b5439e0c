"""MXFP8 ("fp8" precision) kernels against oracle/mx8_ref.py (the OCP MX v1.0 conversion restated on the
CPU).  The reference has no fp8 path: these results are "parity unpinned" against it (see the oracle's
header); the kernel-level bar here is bit-exact quantisation and fp32-grade accumulation."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mx8_ref  # noqa: E402  (test infrastructure: the checker)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def eng():
    import tap_clip_amd  # noqa: F401
    from tap_clip_amd import engine
    return engine


def _data(rows, K, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(rows, K, generator=g)
    x *= torch.exp2(torch.randint(-12, 12, (rows, K // 32, 1), generator=g).float()).repeat_interleave(32, 1).reshape(rows, K)
    x[0, :32] = 0.0            # an all-zero block
    x[1, 32:64] = 2.0 ** -140  # a denormal block
    x[2, 0] = 3.0e38           # near the fp32 maximum
    return x


@pytest.mark.parametrize("rows,K", [(37, 64), (1000, 768), (197 * 3, 3072)])
def test_quantize_bit_exact(eng, rows, K):
    x = _data(rows, K, 1)
    q, sc = eng.mx8_quantize(x.to(DEV))
    q_ref, s_ref = mx8_ref.quantize(x)
    assert torch.equal(mx8_ref.scales_from_kstep_major(sc.cpu(), rows), s_ref), "e8m0 scale bytes"
    assert torch.equal(q.cpu(), q_ref), "e4m3 element bytes"


@pytest.mark.parametrize("M,N,K", [(300, 256, 256), (2167, 768, 768), (1024, 768, 3072)])
def test_gemm_fp32_out(eng, M, N, K):
    g = torch.Generator().manual_seed(2)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * K ** -0.5
    bias = torch.randn(N, generator=g) * 0.1
    aq, asc = eng.mx8_quantize(a.to(DEV))
    wq, wsc = eng.mx8_quantize(w.to(DEV))
    out = eng.mx8_gemm(aq, asc, wq, wsc, bias.to(DEV)).cpu().double()
    ad = mx8_ref.dequantize(*mx8_ref.quantize(a)).double()
    wd = mx8_ref.dequantize(*mx8_ref.quantize(w)).double()
    ref = ad @ wd.t() + bias.double()
    err = float((out - ref).abs().max() / ref.abs().max())
    # (the block-scaled MFMA does not keep every product bit of a 64-deep dot product: 2e-5 measured, against
    # 1e-7 for the bf16 MFMA; two orders below the format's own rounding)
    assert err < 1e-4, f"MXFP8 GEMM vs exact product of the quantised operands: {err:.2e}"
    # and what the quantisation itself costs against the fp32 product (informational bound: ~3 % of an output's scale)
    full = a.double() @ w.double().t() + bias.double()
    rel = float((out - full).norm() / full.norm())
    print(f"[mx8] M{M} N{N} K{K}: vs quantised operands {err:.2e}, vs fp32 operands rel_l2 {rel:.3e}")
    assert rel < 6e-2


def test_gemm_identity_asymmetric(eng):
    """A = I (exact in e4m3) with an asymmetric, exactly representable W: catches a swapped fragment / scale map."""
    K, N = 256, 256
    w = ((torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 13) - 6.0) * torch.exp2((torch.arange(N) % 5).float())[:, None]
    a = torch.eye(K)
    aq, asc = eng.mx8_quantize(a.to(DEV))
    wq, wsc = eng.mx8_quantize(w.to(DEV))
    out = eng.mx8_gemm(aq, asc, wq, wsc, None).cpu()
    assert torch.equal(out, w.t())


@pytest.mark.parametrize("act", [0, 1])
def test_gemm_gelu_requantised_epilogue(eng, act):
    """c_fc of the fp8 path: GELU and MXFP8 re-quantisation fused into the GEMM epilogue."""
    M, N, K = 777, 1024, 768
    g = torch.Generator().manual_seed(3)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * K ** -0.5 * 1.5
    bias = torch.randn(N, generator=g) * 0.2
    aq, asc = eng.mx8_quantize(a.to(DEV))
    wq, wsc = eng.mx8_quantize(w.to(DEV))
    oq, osc = eng.mx8_gemm(aq, asc, wq, wsc, bias.to(DEV), epilogue="gelu_mx8", act=act)
    got = mx8_ref.dequantize(oq.cpu(), mx8_ref.scales_from_kstep_major(osc.cpu(), M))
    z = (mx8_ref.dequantize(*mx8_ref.quantize(a)).double() @ mx8_ref.dequantize(*mx8_ref.quantize(w)).double().t() + bias.double()).float()
    y = torch.nn.functional.gelu(z) if act == 0 else z * torch.sigmoid(1.702 * z)
    q_ref, s_ref = mx8_ref.quantize(y)
    ref = mx8_ref.dequantize(q_ref, s_ref)
    # the kernel's GELU is a 2.5e-5-accurate fit and its accumulation is fp32: a value within that of a rounding
    # boundary may land on the neighbouring e4m3 code; everything else is bit-identical
    mism = float((oq.cpu() != q_ref).float().mean())
    assert mism < 2e-2, f"{mism:.2%} of the e4m3 bytes differ from the restatement"  # 0.6 % measured
    assert torch.equal(mx8_ref.scales_from_kstep_major(osc.cpu(), M), s_ref) or float((mx8_ref.scales_from_kstep_major(osc.cpu(), M) != s_ref).float().mean()) < 1e-3
    assert float((got - ref).norm() / ref.norm()) < 5e-3
    assert float((got - y).norm() / y.norm()) < 4e-2  # MXFP8 rounding of the activation itself


# ---- the fp8 precision of the image tower ---------------------------------------------------------
def _one_block_vision(eng, d, heads, mlp, seed, precision):
    from tap_clip_amd import configs, synth
    cfg = configs.ClipDims("blk8", 64, 32, 16, configs.TowerDims(d, 1, heads, mlp), configs.TowerDims(128, 1, 2, 256), vocab=16, ctx=8)
    sd = {}
    synth._tower(sd, "visual.transformer.", d, 1, mlp, seed=seed)
    g = torch.Generator().manual_seed(seed)
    sd["visual.conv1.weight"] = torch.randn(d, 3, 16, 16, generator=g) * 0.03
    sd["visual.class_embedding"] = torch.randn(d, generator=g) * 0.3
    sd["visual.positional_embedding"] = torch.randn(5, d, generator=g) * 0.3
    for k in ("ln_pre", "ln_post"):
        sd[f"visual.{k}.weight"] = 1.0 + 0.1 * torch.randn(d, generator=g)
        sd[f"visual.{k}.bias"] = 0.05 * torch.randn(d, generator=g)
    sd["visual.proj"] = torch.randn(d, 64, generator=g) * d ** -0.5
    # prune_last_block=False: this ONE block is the last block, and the library default would run its CLS rows through the
    # 16-bit pooled tail -- these tests are about the MXFP8 block itself
    return cfg, sd, eng.VisionTower(cfg, sd, DEV, precision, prune_last_block=False)


@pytest.mark.parametrize("d,heads,mlp", [(768, 12, 3072), (1024, 16, 4096)])
def test_fp8_block_vs_mx8_emulation(eng, d, heads, mlp):
    """One residual block at the real widths (ViT-B/16 and ViT-L/14), 32 x 32 images (5 tokens), batch 300 so that
    the GEMMs see several row tiles and a ragged last one: the fp8 tower against the oracle with MXFP8 rounding at
    the same operand points (emulate="mx8"), and against the fp32 oracle."""
    from oracle import clip_ref
    cfg, sd, tower = _one_block_vision(eng, d, heads, mlp, 5, "fp8")
    from tap_clip_amd import synth
    images = synth.make_images(300, cfg, 7)
    got = tower.encode_image(images.to(DEV)).cpu()
    ocfg = clip_ref.ClipDims("blk8", 64, 32, 16, clip_ref.TowerDims(d, 1, heads, mlp), clip_ref.TowerDims(128, 1, 2, 256), vocab=16, ctx=8)
    with torch.no_grad():
        emu = clip_ref.encode_image(images, sd, ocfg, emulate="mx8")
        ref = clip_ref.encode_image(images, sd, ocfg)
    e_emu = float((got - emu).norm() / emu.norm())
    e_ref = float((got - ref).norm() / ref.norm())
    print(f"[fp8 block d={d}] vs mx8 emulation rel_l2 {e_emu:.3e}; vs fp32 oracle rel_l2 {e_ref:.3e}")
    # The operand bytes agree except where a value sits within the kernels' bf16-level differences (patch embed,
    # GELU fit, accumulation order: ~1e-3) of an e4m3 rounding boundary; such an element moves by a whole e4m3 step
    # (6-12 %), so ~1-2 % flipped elements leave ~1 % rel_l2 (measured 1.1e-2), a quarter of the format's own error.
    # A misplaced scale or block shows up as tens of percent.
    assert e_emu < 2e-2 and e_emu < 0.5 * e_ref
    assert e_ref < 5e-2  # MXFP8: 3 mantissa bits per operand element
    assert torch.isfinite(got).all()


def test_fp8_encode_image_vitb16(eng):
    """Full ViT-B/16 image tower in fp8 against the fp32 golden embeddings and the bf16 path: what the throughput
    mode costs in accuracy (reported, bounded loosely), plus determinism and ragged batches."""
    import numpy as np
    from tap_clip_amd import configs, synth
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "image_tower_ViT-B-16.npz"))
    cfg = configs.get_config("ViT-B-16")
    sd = synth.make_state_dict(cfg, seed=int(g["seed_weights"]), text=False)
    images = synth.make_images(int(g["batch"]), cfg, int(g["seed_images"]))
    ref = torch.from_numpy(g["embeddings"])
    tower = eng.VisionTower(cfg, sd, DEV, "fp8")
    a = tower.encode_image(images.to(DEV)).cpu()
    b = tower.encode_image(images.to(DEV)).cpu()
    assert torch.equal(a, b), "same input twice must be bit-identical"
    rel = float((a - ref).norm() / ref.norm())
    cos = torch.nn.functional.cosine_similarity(a, ref, dim=-1)
    print(f"[fp8 ViT-B/16] vs fp32 golden: rel_l2 {rel:.3e}, cosine min {float(cos.min()):.5f} mean {float(cos.mean()):.5f}")
    # Floor-based bounds (the loose 0.15 / 0.99 would have passed a broken scale plane in one layer).
    # The oracle with MXFP8 operand rounding at the kernels' rounding points, run with fp32 and with fp64 accumulation,
    # disagrees with ITSELF by `mx8_floor_rel_l2` (2.4e-2 over 12 blocks: rounding-boundary flips of whole e4m3 steps,
    # amplified block by block) -- the closest any two implementations of this pipeline can agree.  The HIP tower is held
    # to 1.5x that floor against the fp64-accumulating emulation, and to 1.5x the FORMAT's own error (that emulation
    # against the fp32 oracle, 3.8e-2) against the fp32 golden.
    emu64, floor = torch.from_numpy(g["embeddings_mx8_f64"]), float(g["mx8_floor_rel_l2"])
    fmt = float((emu64 - ref).norm() / ref.norm())
    e_emu = float((a - emu64).norm() / emu64.norm())
    print(f"[fp8 ViT-B/16] vs MXFP8 emulation (fp64 acc): rel_l2 {e_emu:.3e}; floor {floor:.3e}; format error {fmt:.3e}")
    assert e_emu < 1.5 * floor, (e_emu, floor)
    assert rel < 1.5 * fmt and float(cos.min()) > 0.998, (rel, fmt, float(cos.min()))
    one = tower.encode_image(images[:1].to(DEV)).cpu()   # M = 197: a single ragged row tile
    assert float((one - a[:1]).norm() / a[:1].norm()) < 1e-6


def test_fp8_text_tower_stays_bf16(eng):
    from tap_clip_amd import configs, synth
    cfg = configs.get_config("tiny")
    sd = synth.make_state_dict(cfg, seed=2)
    t = eng.TextTower(cfg, sd, DEV, "fp8")
    assert t.precision == "bf16"
    with pytest.raises(ValueError):
        eng.VisionTower(cfg, sd, DEV, "fp8")  # width 128: the MXFP8 GEMM needs width % 256 == 0


@pytest.mark.parametrize("precision", ["fp8", "fp16"])
def test_full_batch_properties_other_precisions(eng, precision):
    """BASELINE configs[1] size (batch 256) in the fp8 and fp16 precisions: finite, unit norms, run-to-run identical,
    and an image's embedding does not depend on its batch mates (rows outside a K-split tail: bit-identical)."""
    from tap_clip_amd import configs, synth
    cfg = configs.get_config("ViT-B-16")
    sd = synth.make_state_dict(cfg, seed=2, text=False)
    tower = eng.VisionTower(cfg, sd, DEV, precision)
    images = synth.make_images(256, cfg, 0).to(DEV)
    a = tower.encode_image(images, normalize=True)
    assert torch.equal(a, tower.encode_image(images, normalize=True))
    assert torch.isfinite(a).all()
    assert torch.allclose(a.norm(dim=-1), torch.ones(256, device=DEV), atol=1e-5)
    small = tower.encode_image(images[:8].clone(), normalize=True)
    assert torch.equal(small, a[:8])
    last = tower.encode_image(images[-8:].clone(), normalize=True)
    assert float((last - a[-8:]).norm() / a[-8:].norm()) < (2e-3 if precision == "fp16" else 5e-2)


def test_fp8_flash_attention_path_vit_l14_geometry(eng):
    """ViT-L/14@336 geometry (577 tokens, d = 1024, 16 heads), one block, fp8: the flash-style attention kernel's
    MXFP8 store path, against the oracle with MXFP8 rounding at the same points and against the fp32 oracle."""
    from oracle import clip_ref
    from tap_clip_amd import configs, synth
    d, heads, mlp = 1024, 16, 4096
    cfg = configs.ClipDims("blk8l", 64, 336, 14, configs.TowerDims(d, 1, heads, mlp), configs.TowerDims(128, 1, 2, 256), vocab=16, ctx=8)
    sd = {}
    synth._tower(sd, "visual.transformer.", d, 1, mlp, seed=9)
    g = torch.Generator().manual_seed(9)
    sd["visual.conv1.weight"] = torch.randn(d, 3, 14, 14, generator=g) * 0.03
    sd["visual.class_embedding"] = torch.randn(d, generator=g) * 0.3
    sd["visual.positional_embedding"] = torch.randn(577, d, generator=g) * 0.3
    for k in ("ln_pre", "ln_post"):
        sd[f"visual.{k}.weight"] = 1.0 + 0.1 * torch.randn(d, generator=g)
        sd[f"visual.{k}.bias"] = 0.05 * torch.randn(d, generator=g)
    sd["visual.proj"] = torch.randn(d, 64, generator=g) * d ** -0.5
    tower = eng.VisionTower(cfg, sd, DEV, "fp8", prune_last_block=False)  # (one block = the last block: keep it on MXFP8)
    images = synth.make_images(3, cfg, 11)
    got = tower.encode_image(images.to(DEV)).cpu()
    ocfg = clip_ref.ClipDims("blk8l", 64, 336, 14, clip_ref.TowerDims(d, 1, heads, mlp), clip_ref.TowerDims(128, 1, 2, 256), vocab=16, ctx=8)
    with torch.no_grad():
        emu = clip_ref.encode_image(images, sd, ocfg, emulate="mx8")
        ref = clip_ref.encode_image(images, sd, ocfg)
    e_emu = float((got - emu).norm() / emu.norm())
    e_ref = float((got - ref).norm() / ref.norm())
    print(f"[fp8 flash block] vs mx8 emulation rel_l2 {e_emu:.3e}; vs fp32 oracle rel_l2 {e_ref:.3e}")
    assert e_emu < 2e-2 and e_ref < 5e-2
