"""Import shim: exposes the directory ``pop-net_amd/`` as the package ``popnet_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pop-net_amd")
_spec = importlib.util.spec_from_file_location(
    "popnet_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["popnet_amd"] = _mod
_spec.loader.exec_module(_mod)
