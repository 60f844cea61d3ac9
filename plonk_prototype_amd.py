"""Import shim: the package directory is ``plonk-prototype_amd/`` (a hyphen is not a legal
module name), so ``import plonk_prototype_amd`` loads it from there under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "plonk-prototype_amd")
_spec = importlib.util.spec_from_file_location(
    "plonk_prototype_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["plonk_prototype_amd"] = _mod
_spec.loader.exec_module(_mod)
