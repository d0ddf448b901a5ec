"""Import shim: the package lives in the directory ``n-hans_amd/`` (not an importable
identifier), so ``import nhans_amd`` loads that directory as the package ``nhans_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "n-hans_amd")
_spec = importlib.util.spec_from_file_location(
    "nhans_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["nhans_amd"] = _mod
_spec.loader.exec_module(_mod)
