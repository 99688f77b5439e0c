"""tapclip: MI355X-native (gfx950) CLIP dual-encoder forward + attribution hot path of
3300786/TAP-CLIP behind the reference's own Python interface.  The arithmetic lives in
csrc/libtapclip.so (C ABI: include/tapclip.h); importing `models` / `engine` without that library
raises -- there is no CPU fallback."""
from . import configs, synth  # noqa: F401  (pure host-side helpers, importable without the library)

__all__ = ["configs", "synth", "engine", "models", "dist"]


def __getattr__(name):
    if name in ("engine", "models", "dist", "_lib"):
        import importlib

        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(name)
