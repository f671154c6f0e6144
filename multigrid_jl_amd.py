"""Import shim: the package directory is ``multigrid.jl_amd/`` (a dot is not importable), so
``import multigrid_jl_amd`` loads it from there under this name."""
import importlib.util
import os
import sys

_d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "multigrid.jl_amd")
_spec = importlib.util.spec_from_file_location(
    "multigrid_jl_amd", os.path.join(_d, "__init__.py"), submodule_search_locations=[_d])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["multigrid_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
