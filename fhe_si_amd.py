"""Import shim: the package directory is named ``fhe-si_amd`` (not a Python identifier); this module loads it
under the importable name ``fhe_si_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fhe-si_amd")
_spec = importlib.util.spec_from_file_location("fhe_si_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["fhe_si_amd"] = _mod
_spec.loader.exec_module(_mod)
