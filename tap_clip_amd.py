"""Import alias: the package directory is named `tap-clip_amd/` (not a valid Python identifier), so
`import tap_clip_amd` loads that directory as the package `tap_clip_amd`."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_pkg_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "tap-clip_amd")
_spec = _ilu.spec_from_file_location(
    "tap_clip_amd", _os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir]
)
_mod = _ilu.module_from_spec(_spec)
_sys.modules["tap_clip_amd"] = _mod
_spec.loader.exec_module(_mod)
