"""tapclip: MI355X-native (gfx950) CLIP dual-encoder forward + attribution hot path of
3300786/TAP-CLIP behind the reference's own Python interface.  The arithmetic lives in
csrc/libtapclip.so (C ABI: include/tapclip.h); importing `models` / `engine` without that library
raises -- there is no CPU fallback."""
from . import configs, synth  # noqa: F401  (pure host-side helpers, importable without the library)

__all__ = ["configs", "synth", "engine", "models", "dist"]


def install_as_models(host_side: bool = False) -> None:
    """Make the reference's own import lines resolve to this package:
    `from models.clip_wrapper import CLIPWrapper`, `from models.model_wrapper import FullModel`
    (reference train.py:3-4, test_cross_domain.py:4-5) then need no edit at all.
    `host_side=True` also takes over `from dataset import get_dataloaders` and `from utils.eval_metrics import ...`
    (train.py:5-6): the torchvision-free loader (with its `gpu_preprocess=` option) and the evaluation functions
    that count on the device."""
    import importlib
    import sys

    pkg = importlib.import_module(f"{__name__}.models")
    sys.modules["models"] = pkg
    for sub in ("clip_wrapper", "model_wrapper", "prompt_learner", "attribution_monitor", "prompt_adjustor"):
        sys.modules[f"models.{sub}"] = importlib.import_module(f"{__name__}.models.{sub}")
    if host_side:
        sys.modules["dataset"] = importlib.import_module(f"{__name__}.dataset")
        sys.modules["utils"] = importlib.import_module(f"{__name__}.utils")
        sys.modules["utils.eval_metrics"] = importlib.import_module(f"{__name__}.utils.eval_metrics")


def __getattr__(name):
    if name in ("engine", "models", "dist", "_lib"):
        import importlib

        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(name)
