"""Import shim: ``import saunet_amd`` loads the package that lives in ``shape-attentive-unet_amd/``
(a hyphenated directory name cannot be imported directly)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shape-attentive-unet_amd")
_spec = importlib.util.spec_from_file_location("saunet_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["saunet_amd"] = _mod
_spec.loader.exec_module(_mod)
