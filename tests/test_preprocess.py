"""The input side: `clip.get_preprocess()`'s eval transform.

CPU tests pin oracle/preprocess_ref.py -- a restatement of Pillow's 8-bit bicubic resampling -- against Pillow
itself, BIT-EXACTLY (Pillow is the library the reference's transform calls for its PIL images, and it is installed
here), and the float tail against torch CPU ops.  GPU tests compare tapclip_preprocess_u8 (through the C ABI) with
the oracle: bit-exact, integer work and correctly rounded fp32 divisions."""
import numpy as np
import pytest
import torch

import tap_clip_amd  # noqa: F401
from oracle import preprocess_ref as P

Image = pytest.importorskip("PIL.Image")

# (h, w, size): down- and up-scales, odd sizes, square, 1-pixel-wide, taps beyond the kernel's LDS table (scale > 32)
CASES = [(375, 500, 224), (500, 375, 224), (224, 224, 224), (100, 80, 224), (37, 53, 48), (231, 229, 224), (64, 64, 224),
         (3, 5, 16), (1, 40, 8), (40, 1, 8), (600, 800, 336), (24, 1000, 24), (1000, 24, 24)]


def _img(h, w, seed):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if seed % 2:  # smooth images too: long runs exercise the rounding, noise the clipping
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack([(yy * 255 // max(h - 1, 1)), (xx * 255 // max(w - 1, 1)), ((yy + xx) % 256)], -1).astype(np.uint8)
    return base


@pytest.mark.parametrize("h,w,size", CASES)
def test_oracle_resize_is_pillow_bit_exact(h, w, size):
    img = _img(h, w, h * 7 + w)
    nh, nw = P.resized_size(h, w, size)
    assert min(nh, nw) == size
    ref = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BICUBIC))
    assert np.array_equal(P.resize_bicubic_u8(img, nh, nw), ref)


def test_oracle_transform_matches_pillow_plus_torch_ops():
    img = _img(375, 500, 3)
    nh, nw = P.resized_size(375, 500, 224)
    assert (nh, nw) == (224, 298)
    top, left = P.crop_origin(nh, nw, 224)
    assert (top, left) == (0, 37)
    assert P.crop_origin(299, 224, 224) == (38, 0)  # 37.5 rounds to even
    r = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BICUBIC))[top: top + 224, left: left + 224]
    x = torch.from_numpy(r.copy()).permute(2, 0, 1).to(torch.float32).div(255)
    x = x.sub(torch.tensor(P.CLIP_MEAN).view(3, 1, 1)).div(torch.tensor(P.CLIP_STD).view(3, 1, 1))
    assert np.array_equal(P.clip_preprocess(img, 224), x.numpy())


def test_wrapper_preprocess_follows_the_same_transform():
    """CLIPWrapper.get_preprocess() on a PIL image (the CPU path the dataset workers run, as the reference's)"""
    from tap_clip_amd.models.clip_wrapper import _make_preprocess

    img = _img(120, 90, 5)
    out = _make_preprocess(64)(Image.fromarray(img))
    assert out.shape == (3, 64, 64) and out.dtype == torch.float32
    assert np.array_equal(out.numpy(), P.clip_preprocess(img, 64))


@pytest.mark.gpu
def test_gpu_preprocess_bit_exact_mixed_batch():
    from tap_clip_amd import engine

    imgs = [_img(h, w, h * 7 + w) for h, w, _ in CASES]
    for size in (224, 48):
        out = engine.preprocess_u8(imgs, size=size).cpu().numpy()
        assert out.shape == (len(imgs), 3, size, size)
        for i, im in enumerate(imgs):
            ref = P.clip_preprocess(im, size)
            assert np.array_equal(out[i], ref), f"image {i} {im.shape} size {size}: max diff {np.abs(out[i] - ref).max()}"


@pytest.mark.gpu
def test_gpu_preprocess_matches_pillow_directly_and_feeds_the_tower():
    from tap_clip_amd import engine

    img = _img(333, 517, 11)
    pil = Image.fromarray(img)
    out = engine.preprocess_u8([pil, torch.from_numpy(img).cuda()], size=224)
    nh, nw = P.resized_size(333, 517, 224)
    top, left = P.crop_origin(nh, nw, 224)
    r = np.asarray(pil.resize((nw, nh), Image.BICUBIC))[top: top + 224, left: left + 224]
    x = torch.from_numpy(r.copy()).permute(2, 0, 1).to(torch.float32).div(255)
    x = x.sub(torch.tensor(P.CLIP_MEAN).view(3, 1, 1)).div(torch.tensor(P.CLIP_STD).view(3, 1, 1))
    assert torch.equal(out[0].cpu(), x) and torch.equal(out[1].cpu(), x)
    with pytest.raises(ValueError):
        engine.preprocess_u8([torch.zeros(4, 4, 3)])  # not uint8
    with pytest.raises(ValueError):
        engine.preprocess_u8([])


def _photo_folder(root):
    rng = np.random.default_rng(3)
    for cls in ["Backpack", "Mug", "Pen"]:
        (root / cls).mkdir(parents=True)
        for k in range(5):
            h, w = int(rng.integers(40, 200)), int(rng.integers(40, 200))
            Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(root / cls / f"{k}.png")


def test_raw_loader_hands_over_decoded_photos(tmp_path):
    """gpu_preprocess= leaves resize / crop / normalise to the device: workers return the decoded uint8 photos"""
    from tap_clip_amd.dataset import get_dataloaders

    _photo_folder(tmp_path / "rw")
    _, val = get_dataloaders(str(tmp_path / "rw"), ["Pen", "Mug"], batch_size=4, num_shots=1, gpu_preprocess=32, num_workers=0, seed=0)
    photos, labels = next(iter(val.loader))
    assert isinstance(photos, list) and len(photos) == 4 and labels.dtype == torch.int64
    assert all(p.dtype == torch.uint8 and p.dim() == 3 and p.shape[2] == 3 for p in photos)
    with pytest.raises(ValueError):
        get_dataloaders(str(tmp_path / "rw"), ["Pen"], preprocess=lambda x: x, gpu_preprocess=32, num_workers=0)


@pytest.mark.gpu
def test_gpu_loader_batches_equal_cpu_loader_batches(tmp_path):
    from tap_clip_amd.dataset import get_dataloaders
    from tap_clip_amd.models.clip_wrapper import _make_preprocess

    _photo_folder(tmp_path / "rw")
    names = ["Pen", "Backpack", "Mug"]
    _, val_cpu = get_dataloaders(str(tmp_path / "rw"), names, batch_size=5, num_shots=1, preprocess=_make_preprocess(64), num_workers=0, seed=1)
    _, val_gpu = get_dataloaders(str(tmp_path / "rw"), names, batch_size=5, num_shots=1, gpu_preprocess=64, num_workers=0, seed=1)
    assert len(val_cpu) == len(val_gpu)
    n = 0
    for (xc, yc), (xg, yg) in zip(val_cpu, val_gpu):
        assert xg.is_cuda and torch.equal(xc, xg.cpu()) and torch.equal(yc, yg.cpu())
        n += len(yc)
    assert n == 12


def test_oracle_resize_random_geometries_pillow_bit_exact():
    """40 seeded random (h, w, size) triples, aspect ratios up to 12:1, sizes 8..96: oracle == Pillow, every byte"""
    rng = np.random.default_rng(2024)
    for k in range(40):
        size = int(rng.integers(8, 97))
        h = int(rng.integers(1, 400))
        w = int(np.clip(h * float(rng.uniform(1 / 12, 12)), 1, 600))
        img = _img(h, w, k)
        nh, nw = P.resized_size(h, w, size)
        ref = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BICUBIC))
        assert np.array_equal(P.resize_bicubic_u8(img, nh, nw), ref), (h, w, size)


@pytest.mark.gpu
def test_gpu_preprocess_random_geometries_bit_exact():
    from tap_clip_amd import engine

    rng = np.random.default_rng(7)
    for size in (32, 96):
        imgs = []
        for k in range(24):
            h = int(rng.integers(1, 300))
            w = int(np.clip(h * float(rng.uniform(1 / 10, 10)), 1, 500))
            imgs.append(_img(h, w, k))
        out = engine.preprocess_u8(imgs, size=size).cpu().numpy()
        for i, im in enumerate(imgs):
            assert np.array_equal(out[i], P.clip_preprocess(im, size)), (im.shape, size)


@pytest.mark.gpu
def test_script_flow_loader_to_accuracy(tmp_path, capsys):
    """The flow of the reference's train.py (54-67, 99-116) end to end on the package: few-shot loaders with the
    transform on the GPU -> FullModel on the tiny towers -> AdamW on the context bank -> evaluate_accuracy.
    Each class folder holds one flat colour (plus noise), so a few prompt-tuning steps must fit the 2-shot set."""
    from tap_clip_amd import configs, synth
    from tap_clip_amd.dataset import get_dataloaders
    from tap_clip_amd.models import CLIPWrapper, FullModel
    from tap_clip_amd.utils.eval_metrics import evaluate_accuracy, evaluate_per_class_accuracy

    names = ["Backpack", "Mug", "Pen"]
    rng = np.random.default_rng(0)
    colours = {"Backpack": (220, 30, 30), "Mug": (30, 220, 30), "Pen": (30, 30, 220)}
    for cls in names:
        (tmp_path / "rw" / cls).mkdir(parents=True)
        for k in range(6):
            h, w = int(rng.integers(40, 90)), int(rng.integers(40, 90))
            img = np.clip(np.asarray(colours[cls])[None, None, :] + rng.integers(-25, 26, (h, w, 3)), 0, 255).astype(np.uint8)
            Image.fromarray(img).save(tmp_path / "rw" / cls / f"{k}.png")
    cfg = configs.get_config("tiny")
    clip = CLIPWrapper("tiny", None, "cuda", precision="bf16", state_dict=synth.make_state_dict(cfg, seed=3))
    train, val = get_dataloaders(str(tmp_path / "rw"), names, batch_size=6, num_shots=2, gpu_preprocess=cfg.image_size,
                                 num_workers=0, seed=0)
    model = FullModel(names, clip, prompt_len=4, class_specific=True)
    opt = torch.optim.AdamW(model.prompt_learner.parameters(), lr=5e-2)
    first = last = None
    for epoch in range(25):
        model.train()
        for images, labels in train:
            assert images.is_cuda and images.shape[1:] == (3, cfg.image_size, cfg.image_size)
            out = model(images, labels)
            opt.zero_grad()
            out["loss"].backward()
            opt.step()
            last = float(out["loss"].detach())
            first = last if first is None else first
    assert last < first, (first, last)
    model.eval()
    acc = evaluate_accuracy(model, val, "cuda")
    per = evaluate_per_class_accuracy(model, val, "cuda", names)
    assert 0.0 <= acc <= 100.0 and set(per) == set(names)
    assert "Overall Accuracy" in capsys.readouterr().out
    assert acc >= 66.0, (acc, per)  # colours are linearly separable: the tuned prompts must get most of 12 images right
