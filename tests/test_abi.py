"""CPU: the C-ABI library loads and exports every symbol include/tapclip.h declares; argument
validation paths that need no GPU behave as documented."""
import ctypes as C
import os
import re

import pytest

import tap_clip_amd  # noqa: F401
from tap_clip_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "tapclip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tapclip_[a-z0-9_]+)\s*\(", text)))


@pytest.mark.parametrize("variant", ["bf16", "fp16"])
def test_library_exports_every_declared_symbol(variant):
    lib = _lib.load(variant)  # libtapclip.so / libtapclip_fp16.so (the IEEE-half build of the same sources)
    names = _declared()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"the {variant} library does not export {n}"
    assert sorted(s[0] for s in _lib.SYMBOLS) == names, "ctypes table and header disagree"


def test_abi_version_and_error_channel():
    lib = _lib.load()
    assert lib.tapclip_abi_version() == 1
    h = C.c_void_p()
    bad = _lib.TowerCfg(kind=0, width=100, layers=1, heads=1, mlp_dim=128, embed_dim=64, image_size=32, patch=8,
                        ctx_len=77, vocab=10, act=0, precision=0)
    assert lib.tapclip_tower_create(C.byref(bad), C.byref(h)) == _lib.EINVAL
    assert b"width" in lib.tapclip_last_error()
    with pytest.raises(ValueError):
        _lib.check(_lib.EINVAL)
    bad.width, bad.heads = 128, 3
    assert lib.tapclip_tower_create(C.byref(bad), C.byref(h)) == _lib.EINVAL
    assert b"head dim" in lib.tapclip_last_error()


def test_tower_handle_lifecycle_without_gpu():
    """create / workspace sizing / ready() / destroy touch no device memory."""
    lib = _lib.load()
    cfg = _lib.TowerCfg(kind=_lib.TOWER_VISION, width=768, layers=12, heads=12, mlp_dim=3072, embed_dim=512,
                        image_size=224, patch=16, ctx_len=77, vocab=49408, act=0, precision=0)
    h = C.c_void_p()
    assert lib.tapclip_tower_create(C.byref(cfg), C.byref(h)) == 0
    try:
        assert lib.tapclip_tower_ready(h) == _lib.ESTATE  # strict: nothing loaded yet
        assert b"missing" in lib.tapclip_last_error()
        n = lib.tapclip_tower_workspace_bytes(h, 256, 197)
        M = 256 * 197
        assert n >= M * 768 * 4 + M * 768 * 2 * 2 + M * 2304 * 2 + M * 3072 * 2
        assert n < 2 * (1 << 30)
    finally:
        lib.tapclip_tower_destroy(h)


def test_missing_library_is_a_hard_error(monkeypatch):
    monkeypatch.setattr(_lib, "_libs", {})
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libtapclip.so")
    monkeypatch.setattr(_lib, "LIB_PATH_FP16", "/nonexistent/libtapclip_fp16.so")
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load("fp16")
