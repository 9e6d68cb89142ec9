"""Import helper: the product package directory is `near-light-client_amd/` (not a valid
Python identifier), so it is loaded by path and registered as module `nlx_amd`."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(_ROOT, "near-light-client_amd")


def load():
    mod = sys.modules.get("nlx_amd")
    if mod is not None:
        return mod
    spec = importlib.util.spec_from_file_location(
        "nlx_amd", os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["nlx_amd"] = mod
    spec.loader.exec_module(mod)
    return mod
