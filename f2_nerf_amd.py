"""Import shim: the package directory is named `f2-nerf_amd` (not a valid Python identifier); this module
loads it under the importable name `f2_nerf_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "f2-nerf_amd")
_spec = importlib.util.spec_from_file_location("f2_nerf_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["f2_nerf_amd"] = _mod
_spec.loader.exec_module(_mod)
