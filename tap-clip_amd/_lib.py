"""ctypes binding of the C ABI in include/tapclip.h (libtapclip.so, built in-tree by
`make -C tap-clip_amd/csrc` or `__graft_entry__.build()`).

There is NO fallback: if the library is missing or a symbol is absent, importing the product path
raises.  This is the stub a maintainer of the reference would add."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libtapclip.so")
# the same sources compiled with IEEE-half operands instead of bf16 (csrc/common.h TAPCLIP_FP16): precision "fp16"
LIB_PATH_FP16 = os.path.join(_HERE, "csrc", "libtapclip_fp16.so")

# (name, restype, argtypes) -- must list every symbol include/tapclip.h declares
_i32, _i64, _f32 = C.c_int32, C.c_int64, C.c_float
_p, _sz = C.c_void_p, C.c_size_t


class TowerCfg(C.Structure):
    _fields_ = [(n, _i32) for n in ("kind", "width", "layers", "heads", "mlp_dim", "embed_dim", "image_size",
                                    "patch", "ctx_len", "vocab", "act", "precision")]


SYMBOLS = [
    ("tapclip_tower_create", _i32, [C.POINTER(TowerCfg), C.POINTER(_p)]),
    ("tapclip_tower_destroy", None, [_p]),
    ("tapclip_tower_load_weight", _i32, [_p, C.c_char_p, _p, C.POINTER(_i64), _i32, _p]),
    ("tapclip_tower_ready", _i32, [_p]),
    ("tapclip_tower_workspace_bytes", _sz, [_p, _i64, _i32]),
    ("tapclip_encode_image", _i32, [_p, _p, _i32, _p, _i32, _p, _sz, _p]),
    ("tapclip_text_forward", _i32, [_p, _p, _i32, _i32, _i32, _p, _p, _p, _p, _p, _sz, _p]),
    ("tapclip_text_pool_project", _i32, [_p, _p, _i32, _i32, _p, _i32, _i32, _p, _p]),
    ("tapclip_text_backward_workspace_bytes", _sz, [_p, _i64, _i32]),
    ("tapclip_text_backward", _i32, [_p, _p, _p, _i32, _i32, _i32, _p, _p, _sz, _p]),
    ("tapclip_text_saved_bytes", _sz, [_p, _i64, _i32]),
    ("tapclip_text_forward_saved", _i32, [_p, _p, _i32, _i32, _i32, _p, _p, _sz, _p, _sz, _p]),
    ("tapclip_text_backward_saved", _i32, [_p, _p, _sz, _p, _i32, _i32, _i32, _p, _p, _sz, _p]),
    ("tapclip_text_tail_run", _i32, [_p, _i32, _i32, _i32, C.POINTER(_i32), _p]),
    ("tapclip_text_tied_workspace_bytes", _sz, [_p, _i64, _i32, _i32]),
    ("tapclip_text_forward_tied", _i32, [_p, _p, _i32, _i32, _i32, _p, _p, _p, _p, _p, _sz, _p]),
    ("tapclip_text_forward_saved_tied", _i32, [_p, _p, _i32, _i32, _i32, _p, _p, _sz, _p, _sz, _p]),
    ("tapclip_text_backward_saved_tied", _i32, [_p, _p, _sz, _p, _i32, _i32, _i32, _p, _p, _sz, _p]),
    ("tapclip_text_tied_violations", _i32, [_p, C.POINTER(_i32), _p]),
    ("tapclip_text_pool_project_backward", _i32, [_p, _p, _i32, _i32, _i32, _p, _p, _p]),
    ("tapclip_logits_backward", _i32, [_p, _p, _p, _f32, _i32, _i32, _i32, _p, _p, _p]),
    ("tapclip_embed_tokens", _i32, [_p, _p, _i32, _i32, _i32, _p, _p]),
    ("tapclip_attribution", _i32, [_p, _i32, _i32, _i32, _i32, _i32, _p, _p]),
    ("tapclip_build_prompts", _i32, [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p, _p]),
    ("tapclip_build_prompts_backward", _i32, [_p, _p, _i32, _i32, _i32, _i32, _i32, _p, _p]),
    ("tapclip_build_prompts_mlp", _i32, [_i32, _p, _p, _p, _i32, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _p, _p]),
    ("tapclip_logits", _i32, [_p, _p, _f32, _i32, _i32, _i32, _p, _p]),
    ("tapclip_preprocess_u8", _i32, [_p, _p, _i32, _i32, _p, _p, _p, _p]),
    ("tapclip_layernorm_f32", _i32, [_p, _p, _p, _i64, _i32, _p, _p]),
    ("tapclip_gemm_scratch_bytes", _sz, [_i64, _i32, _i32]),
    ("tapclip_gemm_f32", _i32, [_p, _p, _p, _i64, _i32, _i32, _i32, _p, _p, _sz, _p]),
    ("tapclip_mx8_quantize", _i32, [_p, _i64, _i32, _p, _p, _i64, _p]),
    ("tapclip_mx8_gemm", _i32, [_p, _p, _i64, _i64, _p, _p, _p, _i32, _i32, _i32, _i32, _p, _p, _p, _p]),
    ("tapclip_comm_unique_id", _i32, [_p]),
    ("tapclip_comm_create", _i32, [_p, _i32, _i32, C.POINTER(_p)]),
    ("tapclip_allgather", _i32, [_p, _p, _p, _sz, _p]),
    ("tapclip_comm_check", _i32, [_p]),
    ("tapclip_comm_destroy", None, [_p]),
    ("tapclip_tower_set_flag", _i32, [_p, _i32, _i32]),
    ("tapclip_tower_get_flag", _i32, [_p, _i32, C.POINTER(_i32)]),
    ("tapclip_profile_enable", _i32, [_p, _i32]),
    ("tapclip_profile_read", _i32, [_p, C.POINTER(_f32), C.POINTER(_i64)]),
    ("tapclip_last_error", C.c_char_p, []),
    ("tapclip_abi_version", _i32, []),
]

TOWER_VISION, TOWER_TEXT = 0, 1
ACT_GELU_ERF, ACT_QUICK_GELU = 0, 1
PREC_BF16, PREC_BF16X3, PREC_FP8 = 0, 1, 2
# "fp16" is the bf16 code path of the IEEE-half build of the library
PRECISIONS = {"bf16": PREC_BF16, "bf16x3": PREC_BF16X3, "fp8": PREC_FP8, "fp16": PREC_BF16}
ADJUST_GATE, ADJUST_RESIDUAL = 1, 2
FLAG_PRUNE_LAST_BLOCK = 1
FLAG_KSPLIT = 2  # K-split of partial GEMM rounds over idle CUs: latency (1, default) against CU-time (0)
PROFILE_SLOTS = ("patch_embed", "layernorm", "gemm_qkv", "attention", "gemm_out_proj", "gemm_fc_gelu",
                 "gemm_proj", "pool_proj", "pooled_tail")

EINVAL, ENOMEM, EHIP, ESTATE, EWORKSPACE = -1, -2, -3, -4, -5

_libs = {}


def load(variant: str = "bf16") -> C.CDLL:
    """Load libtapclip.so (variant "bf16") or libtapclip_fp16.so ("fp16") and bind every symbol.  Raises (never
    falls back) when it is missing."""
    if variant in _libs:
        return _libs[variant]
    path = LIB_PATH_FP16 if variant == "fp16" else LIB_PATH
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: build the HIP extension first "
            "(`python -c 'import __graft_entry__ as g; g.build()'` or `make -C tap-clip_amd/csrc`). "
            "There is no CPU fallback."
        )
    lib = C.CDLL(path)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.tapclip_abi_version() != 1:
        raise ImportError(f"libtapclip ABI version {lib.tapclip_abi_version()} != 1")
    _libs[variant] = lib
    return lib


def check(rc: int, lib=None) -> None:
    """0 -> ok; TAPCLIP_EINVAL -> ValueError (the reference raises ValueError on bad shapes/methods,
    reference models/prompt_learner.py:60, models/prompt_adjustor.py:47); others -> RuntimeError."""
    if rc == 0:
        return
    msg = (lib or load()).tapclip_last_error().decode(errors="replace")
    if rc == EINVAL:
        raise ValueError(f"tapclip: {msg}")
    raise RuntimeError(f"tapclip error {rc}: {msg}")
