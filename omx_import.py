"""Import helper: the package directory is named `ominix-mlx_amd/` (not a valid Python
identifier), so it is registered under the module name `ominix_mlx_amd`."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "ominix-mlx_amd")


def load_package():
    if "ominix_mlx_amd" in sys.modules:
        return sys.modules["ominix_mlx_amd"]
    spec = importlib.util.spec_from_file_location(
        "ominix_mlx_amd", os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR]
    )
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ominix_mlx_amd"] = mod
    spec.loader.exec_module(mod)
    return mod
