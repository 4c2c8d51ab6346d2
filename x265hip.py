"""Import shim: the package directory is named `x265-mod-by-patman_amd` (not an importable
identifier), so load it by path under the module name `x265hip_pkg` and re-export."""
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.join(_HERE, "x265-mod-by-patman_amd")
if "x265hip_pkg" not in sys.modules:
    _spec = importlib.util.spec_from_file_location("x265hip_pkg", os.path.join(_PKG, "__init__.py"),
                                                   submodule_search_locations=[_PKG])
    _mod = importlib.util.module_from_spec(_spec)
    sys.modules["x265hip_pkg"] = _mod
    _spec.loader.exec_module(_mod)
from x265hip_pkg import *  # noqa: F401,F403,E402
from x265hip_pkg import binding  # noqa: F401,E402
