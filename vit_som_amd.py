"""Import shim: the product package lives in the directory ``vit-som_amd/`` (a name Python
cannot import directly); ``import vit_som_amd`` loads it from there."""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg = os.path.join(_here, "vit-som_amd")
_spec = importlib.util.spec_from_file_location(
    "vit_som_amd", os.path.join(_pkg, "__init__.py"), submodule_search_locations=[_pkg])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["vit_som_amd"] = _mod
_spec.loader.exec_module(_mod)
