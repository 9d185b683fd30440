"""Import alias: the package directory is `deep-prove_amd/` (not a valid Python identifier), so `import deep_prove_amd`
loads it from there."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "deep-prove_amd")
_spec = importlib.util.spec_from_file_location(
    "deep_prove_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["deep_prove_amd"] = _mod
_spec.loader.exec_module(_mod)
