"""Import alias for the product package.

The package directory is named ``vision-basedsensor_amd`` (the repo contract), which is
not a valid Python identifier; ``import vbs_amd`` loads it under this name so that
``vbs_amd.marker_detection`` etc. resolve to files in that directory.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vision-basedsensor_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
