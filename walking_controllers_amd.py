"""Import shim: loads the package that lives in `walking-controllers_amd/` (a hyphen is
not importable) under the module name `walking_controllers_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "walking-controllers_amd")
_spec = importlib.util.spec_from_file_location(
    "walking_controllers_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["walking_controllers_amd"] = _mod
_spec.loader.exec_module(_mod)
