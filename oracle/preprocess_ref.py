"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the input side of the path: the eval
transform `clip.get_preprocess()` hands to the dataset (reference call sites: models/clip_wrapper.py:13,56-59 ->
open_clip's `image_transform(is_train=False)`; dataset.py:29-35 applies it per sample).  Only tests/, smoke() and
bench.py's cpu_baseline may import this module; the product path is tap-clip_amd/csrc/preprocess.hip.

The arithmetic lives in third-party code that is absent from /root/reference (open_clip_torch and torchvision,
neither pinned by the reference, neither installed here) and in Pillow, which IS installed here (12.2.0):

  Resize(size, BICUBIC)   torchvision `_compute_resized_output_size`: shorter side -> size, longer side ->
                          int(size * long / short) (truncation); then `PIL.Image.resize((nw, nh), BICUBIC)`
  CenterCrop(size)        top = int(round((nh - size) / 2.0)), left likewise (Python's round: half to even)
  ToTensor                uint8 HWC -> float32 CHW, x / 255 (float32 division)
  Normalize(mean, std)    (x - mean) / std in float32, CLIP's constants

Pillow's 8-bit resize (src/libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc,
ImagingResampleHorizontal_8bpc / Vertical_8bpc) is restated below in integer arithmetic: coefficient rows in
float64, normalised, converted to 22-bit fixed point, horizontal pass first, each pass rounded and clipped to
uint8.  Pinned: tests/test_oracle.py checks `resize_bicubic_u8` against Pillow itself BIT-EXACTLY over up- and
down-scales, odd sizes and 1-pixel-wide inputs, and the whole transform against Pillow + torch CPU ops.  The size
rules of torchvision are restated from its published source and have nothing here to be checked against:
"parity unpinned" for those two lines (resized_size / crop_origin)."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def bicubic_filter(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size: int, out_size: int):
    """-> (bounds [out][2] = (first input index, tap count), int32 coefficients [out][ksize])."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [bicubic_filter((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + k * (1 << PRECISION_BITS)) if k < 0 else int(0.5 + k * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    """one 8-bit resampling pass along `axis` of an [H, W, C] uint8 image"""
    in_size = img.shape[axis]
    bounds, kk = precompute_coeffs(in_size, out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], np.uint8)
    for xx in range(out_size):
        xmin, n = bounds[xx]
        acc = np.tensordot(kk[xx, :n].astype(np.int64), src[xmin:xmin + n], axes=(0, 0)) + (1 << (PRECISION_BITS - 1))
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bicubic_u8(img: np.ndarray, nh: int, nw: int) -> np.ndarray:
    """Pillow's Image.resize((nw, nh), BICUBIC) of an [H, W, C] uint8 image: horizontal pass, then vertical;
    a pass whose size does not change is skipped."""
    h, w = img.shape[:2]
    if nw != w:
        img = _pass(img, nw, 1)
    if nh != h:
        img = _pass(img, nh, 0)
    return img


def resized_size(h: int, w: int, size: int):
    """torchvision Resize(size): the shorter side becomes `size`, the longer int(size * long / short)"""
    if w <= h:
        return int(size * h / w), size
    return size, int(size * w / h)


def crop_origin(nh: int, nw: int, size: int):
    return int(round((nh - size) / 2.0)), int(round((nw - size) / 2.0))


def clip_preprocess(img: np.ndarray, size: int = 224) -> np.ndarray:
    """[H, W, 3] uint8 RGB -> [3, size, size] float32, CLIP-normalised"""
    h, w = img.shape[:2]
    nh, nw = resized_size(h, w, size)
    r = resize_bicubic_u8(img, nh, nw)
    top, left = crop_origin(nh, nw, size)
    c = r[top: top + size, left: left + size]
    x = c.astype(np.float32) / np.float32(255.0)
    mean = np.asarray(CLIP_MEAN, np.float32)
    std = np.asarray(CLIP_STD, np.float32)
    return np.ascontiguousarray(((x - mean) / std).transpose(2, 0, 1))
