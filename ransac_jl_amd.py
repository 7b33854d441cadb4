"""Import shim: the package directory is named `ransac.jl_amd/` (a dot cannot be
part of a Python module name), so `import ransac_jl_amd` loads that directory
as the package `ransac_jl_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ransac.jl_amd")
_spec = importlib.util.spec_from_file_location(
    "ransac_jl_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["ransac_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
